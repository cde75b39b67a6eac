"""Runs pieces of the REAL reference that cannot be imported as modules (test infrastructure; build container only).

scripts/preprocess/davis/generate_flows.py imports cv2, skimage and RAFT and loads a RAFT checkpoint at module level,
so `import` fails here -- but the functions the mask path needs (`get_oob_mask` :57-68, `backward_flow_warp` :71-82)
and the mask statements inside `generate_pair_data` (:139-148) use only torch / numpy.  They are cut out of the
reference's source with `ast` and executed unmodified: every line of arithmetic that runs is the reference's own.
"""
import ast
import os

REF = '/root/reference'
FLOWS_PY = os.path.join(REF, 'scripts/preprocess/davis/generate_flows.py')


def available():
    return os.path.isfile(FLOWS_PY)


def reference_mask_code():
    """-> (namespace with get_oob_mask / backward_flow_warp, code object of generate_pair_data's mask statements)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    src = open(FLOWS_PY).read()
    tree = ast.parse(src)
    want = ('get_oob_mask', 'backward_flow_warp')
    funcs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert sorted(f.name for f in funcs) == sorted(want)
    ns = {'torch': torch, 'np': np, 'F': F}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), FLOWS_PY, 'exec'), ns)
    gp = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'generate_pair_data'][0]
    targets = {'warp_flow_1_2', 'err_1', 'mask_1', 'oob_mask_1', 'warp_flow_2_1', 'err_2', 'mask_2', 'oob_mask_2'}
    stmts = [s for s in gp.body if isinstance(s, ast.Assign) and len(s.targets) == 1 and
             isinstance(s.targets[0], ast.Name) and s.targets[0].id in targets]
    assert len(stmts) == 10, 'generate_pair_data: expected the ten mask statements of :139-148, got %d' % len(stmts)
    return ns, compile(ast.Module(body=stmts, type_ignores=[]), FLOWS_PY, 'exec')


def reference_masks(flow_1_2, flow_2_1):
    """numpy float32 [H,W,2] x 2 -> (mask_1, mask_2) uint8 exactly as generate_pair_data stores them (:139-153).
    As in the script, the flows are numpy arrays (there: the output of cv2.resize); its helpers add them to torch
    tensors (`coord + flow_1_2[None, ...]`), which torch accepts."""
    import numpy as np
    ns, code = reference_mask_code()
    env = dict(ns)
    env['flow_1_2'] = flow_1_2
    env['flow_2_1'] = flow_2_1
    exec(code, env)
    return env['mask_1'].astype(np.uint8), env['mask_2'].astype(np.uint8)


class _Loggers(object):
    def add_logger(self, *a):
        pass

    def get_html_logger(self):
        return None


def reference_model(opt_dict):
    """The reference's `models.scene_flow_motion_field.Model(opt, loggers)` on CPU with the monkey-patches of
    SURVEY.md section 8c (no checkpoint files, no torch.hub, no visualiser worker pool); /root/reference must be on
    sys.path.  MiDaS: `torch.hub.load` returns oracle/resnext.py's encoder."""
    import unittest.mock as mock
    from types import SimpleNamespace
    import torch
    import third_party.hourglass as RH
    import third_party.MiDaS as RM
    import visualize.html_visualizer as HV
    from models.scene_flow_motion_field import Model
    from oracle import resnext
    real_load = torch.load
    with mock.patch.object(HV, 'Pool', lambda n: None), \
            mock.patch.object(torch.hub, 'load', lambda repo, entry, *a, **k: resnext.resnext101_32x8d()), \
            mock.patch.object(RM.BaseModel, 'load', lambda self, path: None), \
            mock.patch.object(torch, 'load', lambda path, *a, **k: RH.HourglassModel().state_dict()
                              if 'pretrained_depth_ckpt' in str(path) else real_load(path, *a, **k)):
        return Model(SimpleNamespace(**opt_dict), _Loggers())


class on_reference_path(object):
    """`with on_reference_path():` puts /root/reference first on sys.path and removes the modules it imported
    afterwards (its package names -- models, datasets, losses, networks, third_party, util -- shadow the product's)."""
    ROOTS = ('third_party', 'networks', 'models', 'losses', 'util', 'datasets', 'loggers', 'visualize', 'options', 'configs',
             'scripts')

    def _purge(self, keep_ref):
        import sys
        for m in list(sys.modules):
            if m.split('.')[0] in self.ROOTS:
                f = getattr(sys.modules[m], '__file__', '') or ''
                if (REF in f) != keep_ref or not f:
                    del sys.modules[m]

    def __enter__(self):
        import sys
        self._purge(keep_ref=True)         # drop product modules of the same names so the reference's are imported
        sys.path.insert(0, REF)
        return self

    def __exit__(self, *a):
        import sys
        sys.path.remove(REF)
        self._purge(keep_ref=False)
