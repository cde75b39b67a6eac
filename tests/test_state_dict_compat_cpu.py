"""Checkpoint interchange: the depth networks and the scene-flow MLP expose the reference's state_dict keys
and shapes.  Needs the reference checkout (build container only; skipped on the GPU box, where
/root/reference does not exist).  The MiDaS encoder comes from torch.hub in the reference (unreachable
here): `torch.hub.load` returns oracle/resnext.py's independent restatement of torchvision's ResNeXt-101 32x8d (the one
the golden generator uses), assembled by the reference's own `_make_resnet_backbone`; decoder, hourglass and MLP are
compared against the reference's own modules."""
import os
import sys
import unittest.mock as mock

import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')


def _same(a, b):
    assert list(a) == list(b)
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k


def test_state_dict_keys_and_shapes_match_the_reference():
    sys.path.insert(0, REF)
    try:
        import third_party.hourglass as RH
        import third_party.midas_blocks as RB
        import third_party.MiDaS as RM
        from networks.sceneflow_field import SceneFlowFieldNet as RefMLP
        from dvd_hip.networks.sceneflow_field import SceneFlowFieldNet
        from dvd_hip.third_party.hourglass import HourglassModel_Embed
        from dvd_hip.third_party.MiDaS import MidasNet
        from oracle import resnext
        import torch
        with mock.patch.object(torch.hub, 'load', lambda repo, entry, *a, **k: resnext.resnext101_32x8d()), \
                mock.patch.object(RM.BaseModel, 'load', lambda self, path: None):
            ref_midas = RM.MidasNet(path=None, non_negative=True)
        _same(MidasNet().state_dict(), ref_midas.state_dict())
        _same(HourglassModel_Embed(noexp=False, use_embedding=False).state_dict(),
              RH.HourglassModel_Embed(noexp=False, use_embedding=False).state_dict())
        kw = dict(net_width=256, n_layers=4, time_dependent=True, N_freq_xyz=16, N_freq_t=16)
        _same(SceneFlowFieldNet(**kw).state_dict(), RefMLP(**kw).state_dict())
    finally:
        sys.path.remove(REF)
        for m in [m for m in sys.modules if m.split('.')[0] in ('third_party', 'networks', 'models', 'losses', 'util')
                  and getattr(sys.modules[m], '__file__', '') and REF in (sys.modules[m].__file__ or '')]:
            del sys.modules[m]
