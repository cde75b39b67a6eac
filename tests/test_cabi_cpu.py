"""CPU-only checks of the drop-in boundary: the C-ABI library builds for
gfx950, loads, and exports every symbol include/dvd_hip.h declares.  No
compute is launched here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'dvd_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dvd_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    from dvd_hip import _lib, build
    path = build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), 'libdvd_hip.so does not export %s' % name
    assert set(declared) == set(_lib.SIGNATURES), 'ctypes binding and header disagree: %s' % (
        set(declared) ^ set(_lib.SIGNATURES))


def test_abi_version_and_error_channel():
    from dvd_hip import _lib
    lib = _lib.load()
    assert lib.dvd_abi_version() == _lib.ABI_VERSION
    assert lib.dvd_warp_loss_workspace_bytes(48, 384, 672) >= 48 * 252 * 16
    assert lib.dvd_warp_loss_workspace_bytes(0, 384, 672) == 0
    # argument validation happens before any HIP call, so it is testable without a GPU
    st = lib.dvd_loss_finalize(None, None, None, None)
    assert st == _lib.DVD_EINVAL
    assert b'null' in lib.dvd_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from dvd_hip import ops
    with pytest.raises(RuntimeError, match='GPU tensor'):
        ops.unproject(torch.ones(1, 1, 4, 4), torch.eye(3)[None], torch.zeros(1, 3), torch.eye(3)[None])


def test_missing_library_is_a_hard_error(monkeypatch):
    from dvd_hip import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setenv('DVD_HIP_LIB', '/nonexistent/libdvd_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()
