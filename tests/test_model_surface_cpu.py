"""Drop-in surface of `Model` that can be checked without a GPU: the flag set of
`add_arguments` (names, types, defaults) and the attribute contract after construction, compared with the
reference's own class when the reference checkout is present (build container), and against the recorded
expectations otherwise."""
import argparse
import os
import sys
import unittest.mock as mock
from types import SimpleNamespace

import pytest

import helpers

REF = '/root/reference'
EXPECTED_FLAGS = {
    'l1_mul', 'disp_mul', 'loss_type', 'scene_lr_mul', 'n_down', 'sf_min_mul', 'sf_quantile', 'static_mul', 'flow_mul',
    'acc_mul', 'si_mul', 'cos_mul', 'warm_mul', 'interp_steps', 'warm_sf', 'n_freq_xyz', 'n_freq_t', 'sf_mag_div',
    'one_way', 'weight_steps', 'static', 'motion_seg_hard', 'warm_static', 'use_disp', 'use_disp_ratio',
    'time_dependent', 'use_cnn', 'use_embedding', 'use_motion_seg', 'warm_reg', 'midas'}
OWN_FLAGS = {'mlp_stash_gb', 'mlp_whole_batch_gb', 'mlp_recompute', 'depth_chunk', 'depth_graphs', 'grad_buckets', 'depth_keep_gb', 'act_fp16', 'mlp_stash_fp16',
             'max_act_overflow_skips'}


def _flags(model_cls):
    parser = argparse.ArgumentParser()
    parser, unique = model_cls.add_arguments(parser)
    assert unique == set()
    return {a.dest: (a.type, a.default, type(a).__name__) for a in parser._actions if a.dest != 'help'}


def test_flag_set():
    from dvd_hip.models.scene_flow_motion_field import Model
    ours = _flags(Model)
    assert set(ours) == EXPECTED_FLAGS | OWN_FLAGS
    if not os.path.isdir(REF):
        return
    sys.path.insert(0, REF)
    try:
        import visualize.html_visualizer as HV
        with mock.patch.object(HV, 'Pool', lambda n: None):
            from models.scene_flow_motion_field import Model as RefModel
        ref = _flags(RefModel)
    finally:
        sys.path.remove(REF)
        for m in [m for m in list(sys.modules) if getattr(sys.modules[m], '__file__', None) and REF in sys.modules[m].__file__]:
            del sys.modules[m]
    assert set(ref) == EXPECTED_FLAGS                      # the recorded expectation is the reference's flag set
    for k, v in ref.items():                                  # same type, default and action kind for every shared flag
        assert ours[k] == v, (k, ours[k], v)


def test_attribute_contract_after_construction():
    from dvd_hip.models.scene_flow_motion_field import Model
    from dvd_hip.networks.sceneflow_field import SceneFlowFieldNet
    from dvd_hip.third_party.hourglass import HourglassModel_Embed
    o = dict(helpers.FULL_STEP_OPT)
    o.update(full_logdir='/tmp')
    with pytest.warns(UserWarning):                           # checkpoint absent: random weights announced
        m = Model(SimpleNamespace(**o), None)
    assert [type(n) for n in m._nets] == [HourglassModel_Embed, SceneFlowFieldNet]
    assert m._metrics == ['flow_loss_1_2', 'loss', 'disp_loss_1_2', 'data_time', 'acc_reg', 'sf_loss']
    for name in ('img_1', 'img_2', 'flow_1_2', 'mask_2', 'R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv',
                 'time_stamp_1', 'time_stamp_2', 'time_step', 'motion_seg_1', 'frame_id_1', 'frame_id_2'):
        assert name in m.input_names and name in m.requires and hasattr(m._input, name)
    assert m.num_parameters() == 5357730 + 297987              # hourglass + scene-flow MLP
    # the model is a GPU implementation: moving it to the CPU is an explicit error, not a silent fallback
    import torch
    with pytest.raises(RuntimeError, match='GPU only'):
        m.to(torch.device('cpu'))
    # --use_cnn: the U-Net scene-flow network of the reference (networks/FCNUnet.py:21-92), same state_dict keys
    from dvd_hip.networks.FCNUnet import FCNUnet
    with pytest.warns(UserWarning):
        mc = Model(SimpleNamespace(**dict(o, use_cnn=True)), None)
    assert type(mc.net_sceneflow) is FCNUnet and len(mc.net_sceneflow.state_dict()) == 30
    if os.path.isdir(REF):
        sys.path.insert(0, REF)
        try:
            from networks.FCNUnet import FCNUnet as RefUnet
            ref = RefUnet({'norm': 'none', 'activation': 'lrelu', 'pad_type': 'reflect', 'stride': 1}, n_down=3, feat=32,
                          block_type='double_conv', in_channel=4, out_channel=3)
        finally:
            sys.path.remove(REF)
            for mod in [mm for mm in list(sys.modules) if getattr(sys.modules[mm], '__file__', None) and REF in sys.modules[mm].__file__]:
                del sys.modules[mod]
        assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == \
            {k: tuple(v.shape) for k, v in mc.net_sceneflow.state_dict().items()}
        ref.load_state_dict(mc.net_sceneflow.state_dict())
        x = torch.randn(2, 4, 32, 48)
        assert torch.equal(ref(x), mc.net_sceneflow(x))              # CPU tensors: the reference's own ATen arithmetic


def test_automatic_depth_chunk_choice():
    """Model --depth_chunk 0 (what bench.py runs with): the largest of 48 / 24 / 16 images per slot for which every slot of the
    step fits beside phase 2's allocations, else 16 -- with the free memory / phase-2 figures of the round-6 bench lines
    (DVD_KEEP_DEBUG logs): the headline line keeps two 48-image slots, the hourglass, frame gap 2 and configs[4] above 24 pairs
    take 16-image slots, configs[4] at 24 pairs one slot per image set."""
    from types import SimpleNamespace
    from dvd_hip.models.scene_flow_motion_field import Model
    G = 2 ** 30

    def pick(B, HW, mlp_need_gb, free_gb, midas=True, fp16=False, budget=150.0, measured=0.0):
        fake = SimpleNamespace(opt=SimpleNamespace(depth_keep_gb=budget, midas=midas), _gscale=object() if fp16 else None,
                               _keep_per_px=measured, _free_hbm=lambda dev: (free_gb * G, 288 * G))
        fake._slot_bytes_per_px = lambda: Model._slot_bytes_per_px(fake)
        return Model._pick_depth_chunk(fake, B, HW, mlp_need_gb * G, None)
    hw, hw4 = 384 * 672, 768 * 1344
    assert pick(48, hw, 109.2, 286.4) == 48                              # headline: 96 images x 1.06 GB + two pools fit
    assert pick(48, hw, 109.2, 286.4, midas=False) == 16                 # hourglass: 6.2 KB per pixel, 154 GB for 96 images
    assert pick(48, hw, 155.6, 284.9) == 16                              # gap 2: two Euler stashes of the whole batch
    assert pick(48, hw, 63.4, 284.9) == 48                               # gap 4: the recompute schedule reserves one chunk
    assert pick(64, hw4, 62.5, 281.9, fp16=True, budget=160.0) == 16     # configs[4] at 64 pairs
    assert pick(32, hw4, 145.1, 283.8, fp16=True, budget=160.0) == 16    # ... at 32 pairs
    assert pick(24, hw4, 100.0, 284.0, fp16=True) == 24                  # ... at 24 pairs: one slot per image set
    assert pick(2, 32 * 48, 0.1, 280.0) == 2                             # tiny batches: one slot per image set


def test_kept_activation_slot_planning_arithmetic():
    """models.scene_flow_motion_field.keep_slot_fits with the numbers of the bench (288 GB device, MLP stashes 130 GB):
    two 58 GB slots fit, a third does not; with 82 GB slots (no BatchNorm fusion) only the first fits; the
    --depth_keep_gb budget caps regardless of free memory."""
    from dvd_hip.models.scene_flow_motion_field import keep_slot_fits
    G = 2 ** 30
    total, reserve = 288 * G, 130 * G
    assert keep_slot_fits(52 * G, 285 * G, total, reserve, 52 * G, 0, 150 * G)                 # first slot, a-priori size
    assert keep_slot_fits(60 * G, 225 * G, total, reserve, 0, 58 * G, 150 * G)                 # second (last) slot, measured size
    assert not keep_slot_fits(60 * G, 166 * G, total, reserve, 60 * G, 116 * G, 300 * G)       # a third would starve phase 2
    assert not keep_slot_fits(84 * G, 200 * G, total, reserve, 0, 82 * G, 150 * G)             # 82 GB slots: the second does not fit
    assert not keep_slot_fits(60 * G, 285 * G, total, 0, 0, 116 * G, 150 * G)                  # budget
    # data-parallel runs leave 10 % of the device free instead of 8 %: the benchmark's second slot still fits, a tighter one does not
    from dvd_hip.models.scene_flow_motion_field import head_room_fraction
    assert head_room_fraction(1) == 0.08 and head_room_fraction(8) == 0.10
    assert keep_slot_fits(60 * G, 225 * G, total, reserve, 0, 58 * G, 150 * G, head_room_fraction(8))
    assert keep_slot_fits(60 * G, 215 * G, total, reserve, 0, 58 * G, 150 * G, head_room_fraction(1))
    assert not keep_slot_fits(60 * G, 215 * G, total, reserve, 0, 58 * G, 150 * G, head_room_fraction(8))


def test_grouped_conv_modules_on_cpu_are_plain_convolutions():
    """The depth-net modules run on the CPU through the ATen ops the reference uses (the golden-fixture generator and the
    oracle instantiate them there): the 16-per-group module, strided or not, is nn.Conv2d with the same parameters."""
    import torch
    from dvd_hip import conv as C
    torch.manual_seed(0)
    for stride in (1, 2):
        m = C.GroupedConv3x3C16(64, stride=stride)
        ref = torch.nn.Conv2d(64, 64, 3, stride=stride, padding=1, groups=4, bias=False)
        ref.load_state_dict(m.state_dict())
        x = torch.randn(2, 64, 9, 14)
        assert torch.equal(m(x), ref(x))
    # the A/B switches are exactly these, and none is on by default (DVD_AB is an experimenter's tool, not a configuration)
    assert set(C.AB) == {'gconv32', 'no_c16', 'no_xwgrad3', 'no_xwgrad', 'no_bnfuse', 'no_xconv', 'no_alias', 'no_maskfuse',
                         'no_chansum', 'no_packplan', 'no_s2', 'rowsum'} and not any(C.AB.values())


def test_site_handover_detects_a_modified_or_replaced_gradient():
    """conv._Site: a BatchNorm+ReLU site may skip its mask pass only if the gradient it receives is EXACTLY the tensor its
    consumer's epilogue wrote -- the same tensor object, same version counter.  An in-place accumulation (what autograd does when a
    second consumer's gradient arrives) or a different tensor (a sum) must both be noticed."""
    import torch
    from dvd_hip.conv import _Site
    site = _Site()
    g = torch.zeros(8)
    assert not site.is_exactly(g)                 # nothing recorded yet
    site.wrote(g, amax=torch.ones(1))
    assert site.is_exactly(g) and site.amax is not None
    assert not site.is_exactly(g + 0.0)           # another tensor (autograd replaced it by a sum)
    # ABA: a NEW tensor at the recorded address with version 0 (what the caching allocator hands a sum after the recorded
    # tensor died) is another object; masking again is merely redundant, skipping the mask would be wrong
    alias = g.view(8)
    assert alias.data_ptr() == g.data_ptr() and alias._version == g._version and not site.is_exactly(alias)
    g.add_(1.0)                                   # accumulated into in place: same storage, version counter moved on
    assert not site.is_exactly(g)
