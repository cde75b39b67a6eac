"""`--dataset <name>` resolution like /root/reference/datasets/__init__.py:18-20."""
import importlib


def get_dataset(alias):
    return importlib.import_module('dvd_hip.datasets.' + alias.lower()).Dataset
