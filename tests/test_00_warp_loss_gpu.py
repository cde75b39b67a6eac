"""Parity of the fused HIP warp+loss kernel (through the C ABI) against the
CPU oracle and the golden fixtures.  Tolerances (fp32):
  loss sums            rtol 1e-5
  mask count S0        exact (bit-exact valid-pixel mask => identical count)
  gradients            rtol 1e-4, atol 1e-6 * max|g|   (fp32 atomics in g_depth_2)
"""
import numpy as np
import pytest
import torch

from helpers import golden_batch, golden_opt, load_golden, t
from oracle import losses as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['strips', 'strips64', 'strips64x16', 'strips96x16', 'stripsRy8', 'tiles', 'tile64x32', 'tile96x32x384', 'px4', 'direct'], autouse=True)
def warp_variant(request):
    """Every case runs on the production path -- since round 6 the strip kernel (csrc/warp_strip.hip: 96-column strips walked in
    16-row steps, LDS rings, LDS-direct window loads) for the shipped flag set and rows of whole quads, the tile kernel
    otherwise -- on strips cut into 64-row units (the shortest: most seams between units, every step a first or last step), on
    the other strip shapes (64-column strips in 16-row steps at 32-row units: one thread-step per barrier; 96 x 16; 96 x 32 with
    the 8-row vertical halo instead of 12),
    on the tile kernel of rounds 2-5 (auto tile shape; the smallest tile shape; the 384-thread blocks of the 96 x 32 tile;
    4 pixels per thread-step = the one-pixel loop of rounds 1-4 with the per-quad combine), and on the global-atomics
    reference variant.  All of them must reproduce the oracle's masks, counts and sub-gradient signs."""
    from dvd_hip import ops
    v = request.param
    ops.warp_loss_select(variant='direct' if v == 'direct' else ('tiled' if v.startswith('strips') else 'tiles'),
                         tile={'tile64x32': 3, 'tile96x32x384': 4}.get(v, -1), px=4 if v == 'px4' else 0,
                         strip_rows={'strips64': 64, 'strips64x16': 32}.get(v, 0),
                         strip_shape={'strips64x16': 1, 'strips96x16': 2, 'stripsRy8': 3}.get(v, 0))
    yield v
    ops.warp_loss_select()


CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def _cfg_from_opt(ops, opt, warm, B, H, W, steps=1):
    mul = steps if opt.weight_steps else 1
    disp_mode = 1 if opt.use_disp else (2 if opt.use_disp_ratio else 0)
    return ops.warp_cfg(B, H, W, midas_mask=opt.midas, crit_l2=warm, disp_mode=disp_mode,
                        loss_on_sf=not opt.use_disp, flow_mul=opt.flow_mul * mul, disp_mul=opt.disp_mul * mul)


def _run_hip(ops, cfg, batch_gpu, d1, d2, sf):
    cams = {k: batch_gpu[k] for k in CAM_KEYS}
    sums, g1, g2, gs = ops.warp_loss_fused(cfg, d1, d2, batch_gpu['flow_1_2'], batch_gpu['mask_2'], sf, cams)
    sc = ops.loss_finalize(cfg, sums)
    torch.cuda.synchronize()
    inv = float(sc[0])
    return sums.cpu().numpy(), sc.cpu().numpy(), g1.cpu().numpy() * inv, g2.cpu().numpy() * inv, gs.cpu().numpy() * inv


def _compare(ref, sums, sc, g1, g2, gs):
    rs = ref['sums'].numpy()
    assert sums[0] == rs[0], 'valid-pixel count differs: %r vs %r' % (sums[0], rs[0])
    np.testing.assert_allclose(sums[1:], rs[1:], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sc[1], float(ref['loss']), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sc[2], float(ref['parts']['flow_loss_1_2']), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sc[3], float(ref['parts']['disp_loss_1_2']), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sc[4], float(ref['parts']['sf_loss']), rtol=1e-5, atol=1e-7)
    for name, got in (('g_depth_1', g1), ('g_depth_2', g2), ('g_sf', gs)):
        want = ref[name].numpy()
        np.testing.assert_allclose(got.reshape(want.shape), want, rtol=1e-4,
                                   atol=1e-6 * max(1e-3, np.abs(want).max()), err_msg=name)


@pytest.mark.parametrize('name', ['step_b2_24x32_full', 'step_b2_24x32_warm', 'step_b3_16x40_behind_gap2',
                                  'step_b2_16x24_sfloss', 'step_b2_16x24_ratio'])
def test_against_golden_inputs(name):
    """Inputs (incl. the reference's own scene flow) from the golden fixture;
    loss scalars must match what the REAL reference logged."""
    from dvd_hip import ops
    gd = load_golden(name)
    opt, warm = golden_opt(gd), bool(gd['warm'])
    B, _, H, W = gd['in_depth_1'].shape
    batch = golden_batch(gd)
    d1, d2, sf = t(gd['in_depth_1']), t(gd['in_depth_2']), t(gd['pred_sf_1_2'])
    ref = L.warp_loss_leaf_sf(opt, warm, batch, d1, d2, sf)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, warm, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    _compare(ref, sums, sc, g1, g2, gs)
    # the reference's logged losses (fixture) -- independent of our oracle
    np.testing.assert_allclose(sc[2], float(gd['loss_flow_loss_1_2']), rtol=2e-5)
    np.testing.assert_allclose(sc[3], float(gd['loss_disp_loss_1_2']), rtol=2e-5)
    np.testing.assert_allclose(sc[4], float(gd['loss_sf_loss']), rtol=2e-5)
    np.testing.assert_allclose(sc[1], float(gd['loss_loss']), rtol=2e-5)


CASES = [
    # B, H, W, gap, behind, warm, opt overrides
    (3, 96, 160, 1, 0, False, {}),
    (2, 64, 100, 2, 1, False, {}),                       # W % 4 == 0, behind-camera pair
    (2, 40, 67, 1, 0, False, {}),                        # odd width -> scalar path
    (2, 33, 50, 1, 1, True, {}),                         # warm: L2 criterion
    (2, 48, 64, 1, 0, False, {'use_disp': False}),       # sf-loss mode
    (2, 48, 64, 1, 0, False, {'use_disp': False, 'use_disp_ratio': True}),
    (2, 48, 64, 1, 0, False, {'midas': False}),          # no depth masks
    (1, 32, 32, 1, 0, False, {'flow_mul': 2.5, 'disp_mul': 0.5}),
    (2, 160, 192, 1, 0, False, {}),                      # whole strips and steps, but the last unit of a column reaches below the image
    (2, 128, 192, 1, 1, False, {}),                      # whole strips, steps and units: the strip kernel's FULL instantiation
]


@pytest.mark.parametrize('B,H,W,gap,behind,warm,over', CASES)
def test_against_oracle(B, H, W, gap, behind, warm, over):
    from dvd_hip import ops, synthetic
    opt = L.default_opt(**over)
    batch = synthetic.make_batch(B, H, W, gap=gap, seed=100 + H, behind_camera_pairs=behind, with_images=False)
    batch['flow_1_2'][0, :4] *= 20.0          # some targets far outside the image (border clamp)
    batch['flow_2_1'] = -batch['flow_1_2']
    d1, d2 = synthetic.make_depths(B, H, W, seed=5 + W, far_depth_frac=0.01)
    sf = synthetic.make_scene_flow(B, H, W, seed=9)
    ref = L.warp_loss_leaf_sf(opt, warm, batch, d1, d2, sf)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, warm, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    _compare(ref, sums, sc, g1, g2, gs)
    if behind:
        assert bool(ref['behind'].any())


def test_all_masked_and_forward_only(warp_variant):
    from dvd_hip import ops, synthetic
    B, H, W = 2, 32, 48
    opt = L.default_opt()
    batch = synthetic.make_batch(B, H, W, with_images=False)
    batch['mask_2'].zero_()
    d1, d2 = synthetic.make_depths(B, H, W)
    sf = synthetic.make_scene_flow(B, H, W)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, False, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    assert not sums.any() and not g1.any() and not g2.any() and not gs.any()
    assert sc[1] == 0.0
    # forward-only entry gives the same sums as the fused one: the valid-pixel count exactly; the three error sums bit for bit
    # where both run the same kernel, to fp32 summation order where they do not (since round 6 the fused call of the production
    # path is the strip kernel, the forward-only call the tile kernel: each deterministic, per-block partial sums grouped
    # differently)
    batch['mask_2'].fill_(1.0)
    bg['mask_2'] = batch['mask_2'].cuda()
    cams = {k: bg[k] for k in CAM_KEYS}
    s_f, *_ = ops.warp_loss_fused(cfg, d1.cuda(), d2.cuda(), bg['flow_1_2'], bg['mask_2'], sf.cuda(), cams, grads=False)
    s_b, *_ = ops.warp_loss_fused(cfg, d1.cuda(), d2.cuda(), bg['flow_1_2'], bg['mask_2'], sf.cuda(), cams, grads=True)
    s_b2, *_ = ops.warp_loss_fused(cfg, d1.cuda(), d2.cuda(), bg['flow_1_2'], bg['mask_2'], sf.cuda(), cams, grads=True)
    assert torch.equal(s_b, s_b2)
    assert float(s_f[0]) == float(s_b[0]) and float(s_f[0]) > 0
    if warp_variant.startswith('strips'):
        np.testing.assert_allclose(s_f.cpu().numpy(), s_b.cpu().numpy(), rtol=2e-6)
    else:
        assert torch.equal(s_f, s_b)


def test_rejects_cpu_tensors_and_bad_shapes():
    from dvd_hip import ops, synthetic
    B, H, W = 1, 16, 16
    batch = synthetic.make_batch(B, H, W, with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W)
    sf = synthetic.make_scene_flow(B, H, W)
    cfg = ops.warp_cfg(B, H, W)
    cams = {k: batch[k] for k in CAM_KEYS}
    with pytest.raises(RuntimeError):
        ops.warp_loss_fused(cfg, d1, d2, batch['flow_1_2'], batch['mask_2'], sf, cams)       # CPU tensors
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    camg = {k: bg[k] for k in CAM_KEYS}
    with pytest.raises(RuntimeError):
        ops.warp_loss_fused(ops.warp_cfg(B, H, W + 4), d1.cuda(), d2.cuda(), bg['flow_1_2'], bg['mask_2'],
                            sf.cuda(), camg)
    bad = ops.warp_cfg(B, H, W, disp_mode=7)
    with pytest.raises(RuntimeError, match='disp_mode'):
        ops.warp_loss_fused(bad, d1.cuda(), d2.cuda(), bg['flow_1_2'], bg['mask_2'], sf.cuda(), camg)


def test_full_size_properties():
    """BASELINE config-2 size (48 x 384 x 672): size-independent properties --
    determinism of the sums, linearity of the gradients in the loss
    multipliers, per-pair additivity, and the all-static identity."""
    from dvd_hip import ops, synthetic
    B, H, W = 48, 384, 672
    batch = synthetic.make_batch(B, H, W, device='cuda', with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W, device='cuda')
    sf = synthetic.make_scene_flow(B, H, W, device='cuda')
    cams = {k: batch[k] for k in CAM_KEYS}

    def run(fm, dm, sl=slice(None), b=B):
        cfg = ops.warp_cfg(b, H, W, flow_mul=fm, disp_mul=dm)
        c = {k: v[sl].contiguous() for k, v in cams.items()}
        return ops.warp_loss_fused(cfg, d1[sl].contiguous(), d2[sl].contiguous(), batch['flow_1_2'][sl].contiguous(),
                                   batch['mask_2'][sl].contiguous(), sf[sl].contiguous(), c)
    s_a, g1_a, g2_a, gs_a = run(1.0, 0.0)
    s_b, g1_b, g2_b, gs_b = run(0.0, 1.0)
    s_c, g1_c, g2_c, gs_c = run(0.7, 1.9)
    s_c2, *_ = run(0.7, 1.9)
    assert torch.equal(s_c, s_c2)                                   # deterministic reduction
    assert torch.equal(s_a, s_c)                                    # sums do not depend on multipliers
    for a_, b_, c_ in ((g1_a, g1_b, g1_c), (g2_a, g2_b, g2_c), (gs_a, gs_b, gs_c)):
        lin = 0.7 * a_ + 1.9 * b_
        scale = float(c_.abs().max())
        assert float((lin - c_).abs().max()) <= 2e-5 * scale + 1e-6
    # additivity over pairs: pairs are independent units (this is what DP sharding relies on)
    s_lo, g1_lo, _, _ = run(0.7, 1.9, slice(0, 24), 24)
    s_hi, g1_hi, _, _ = run(0.7, 1.9, slice(24, 48), 24)
    assert float(s_lo[0] + s_hi[0]) == float(s_c[0])
    np.testing.assert_allclose((s_lo + s_hi).cpu().numpy(), s_c.cpu().numpy(), rtol=2e-6)
    assert torch.equal(torch.cat([g1_lo, g1_hi]), g1_c)
    g2_lo = run(0.7, 1.9, slice(0, 24), 24)[2]
    assert float((g2_lo - g2_c[:24]).abs().max()) <= 1e-5 * float(g2_c.abs().max())
    # identity: same camera, zero flow, zero scene flow, same depth -> every loss is exactly 0
    ident = dict(cams)
    ident['R_2'], ident['R_2_T'], ident['t_2'] = cams['R_1'], cams['R_1_T'], cams['t_1']
    cfg = ops.warp_cfg(B, H, W)
    z = torch.zeros_like
    s_i, *_ = ops.warp_loss_fused(cfg, d1, d1, z(batch['flow_1_2']), batch['mask_2'], z(sf), ident)
    s_i = s_i.cpu().numpy()
    assert s_i[0] > 0 and abs(s_i[1]) <= 2e-3 * s_i[0] and s_i[2] <= 1e-3 * s_i[0] and s_i[3] <= 1e-4 * s_i[0]


def test_oracle_parity_at_the_baseline_image_size():
    """The CPU oracle at BASELINE's image size (2 pairs x 384 x 672 -- what fits the oracle in seconds) on the
    production tile shape: 7 x 12 tiles of 96 x 32 per pair with their halo seams, flows up to tens of pixels
    (window overflow path), far depths, one behind-camera pair.  Valid-pixel count exact, per-pixel support of
    the gradients identical to the oracle's (no mask bit decided differently anywhere), sums rtol 1e-5,
    gradients rtol 1e-4."""
    from dvd_hip import ops, synthetic
    B, H, W = 2, 384, 672
    opt = L.default_opt()
    batch = synthetic.make_batch(B, H, W, gap=1, seed=2024, behind_camera_pairs=1, with_images=False)
    g = torch.Generator().manual_seed(7)
    big = (torch.rand(B, H, W, 1, generator=g) < 0.02).float()          # 2 % of the pixels: |flow| ~ 10-40 px
    batch['flow_1_2'] = batch['flow_1_2'] * (1.0 + 9.0 * big)
    batch['flow_1_2'][0, 100:140, 200:300] += torch.tensor([35.0, -22.0])   # a coherent fast-moving region
    batch['flow_2_1'] = -batch['flow_1_2']
    d1, d2 = synthetic.make_depths(B, H, W, seed=77, far_depth_frac=0.01)
    sf = synthetic.make_scene_flow(B, H, W, seed=9)
    ref = L.warp_loss_leaf_sf(opt, False, batch, d1, d2, sf)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, False, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    _compare(ref, sums, sc, g1, g2, gs)
    assert bool(ref['behind'].any()) and float(ref['occ'].sum()) == sums[0]
    # per-pixel: a pixel carries gradient iff the oracle's pixel does (valid-pixel mask, [d1<100], [W2.z<100])
    occ = ref['occ'].reshape(B, H, W).numpy() > 0
    for name, got in (('g_depth_1', g1), ('g_sf', gs)):
        want = ref[name].numpy()
        sup_got = np.abs(got.reshape(B, -1, H, W)).sum(1) > 0
        sup_want = np.abs(want.reshape(B, -1, H, W)).sum(1) > 0
        assert (sup_got != sup_want).sum() == 0, '%s: support differs at %d pixels' % (name, (sup_got != sup_want).sum())
        assert not (sup_got & ~occ).any(), name + ': gradient on a masked pixel'


@pytest.mark.parametrize('mean', [(25.0, -14.0), (-61.0, 37.0)])
def test_coherent_large_flow_against_oracle(mean):
    """A camera pan: every pixel of a pair moves by tens of pixels plus a small residual.  The LDS windows of the pair
    are shifted by its (sampled, rounded) mean flow, so these taps stay on chip; the result must be the oracle's."""
    from dvd_hip import ops, synthetic
    B, H, W = 2, 96, 160
    opt = L.default_opt()
    batch = synthetic.make_batch(B, H, W, gap=1, seed=321, with_images=False)
    batch['flow_1_2'] = batch['flow_1_2'] + torch.tensor(mean)
    batch['flow_1_2'][1] = batch['flow_1_2'][1] * 0.5 - torch.tensor(mean) * 1.2        # a different motion per pair
    batch['flow_2_1'] = -batch['flow_1_2']
    d1, d2 = synthetic.make_depths(B, H, W, seed=12, far_depth_frac=0.01)
    sf = synthetic.make_scene_flow(B, H, W, seed=9)
    ref = L.warp_loss_leaf_sf(opt, False, batch, d1, d2, sf)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, False, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    _compare(ref, sums, sc, g1, g2, gs)


@pytest.mark.parametrize('case', ['skew', 'r2t_not_transpose', 'k22_not_one'])
def test_cameras_outside_the_lockstep_loop_s_preconditions(case):
    """The two-pixel lockstep loop (round 5) runs for pairs whose intrinsics have the pinhole form of the data files
    (generate_frame_midas.py:135-139) and whose R_2 / R_2_T are bit-for-bit transposes (generate_sequence_midas.py:69-72); a
    block tests both (wave-uniform) and otherwise takes the general one-pixel path in the SAME launch.  Here: intrinsics with a
    skew term, an R_2_T that differs from R_2's transpose in the last bit of one entry, and a K whose [2][2] is not exactly 1 --
    one pair each, next to a regular pair in the same batch; the result must be the oracle's (which uses the tensors as given)."""
    from dvd_hip import ops, synthetic
    B, H, W = 2, 64, 96
    opt = L.default_opt()
    batch = synthetic.make_batch(B, H, W, gap=1, seed=77, with_images=False)
    if case == 'skew':
        K = batch['K'][1].reshape(3, 3).clone()          # stored transposed: K^T = [fx 0 0; s fy 0; cx cy 1]
        K[1, 0] = 0.7                                     # skew
        batch['K'][1] = K.reshape(batch['K'][1].shape)
        batch['K_inv'][1] = torch.linalg.inv(K.double()).float().reshape(batch['K_inv'][1].shape)
    elif case == 'r2t_not_transpose':
        r = batch['R_2_T'][0].reshape(-1)
        r[1] = torch.nextafter(r[1], torch.tensor(2.0))
    else:
        k = batch['K'][0].reshape(-1)
        k[8] = torch.nextafter(k[8], torch.tensor(2.0))
    d1, d2 = synthetic.make_depths(B, H, W, seed=3, far_depth_frac=0.01)
    sf = synthetic.make_scene_flow(B, H, W, seed=9)
    ref = L.warp_loss_leaf_sf(opt, False, batch, d1, d2, sf)
    bg = {k: (v.cuda() if k != 'time_step' else v) for k, v in batch.items()}
    cfg = _cfg_from_opt(ops, opt, False, B, H, W)
    sums, sc, g1, g2, gs = _run_hip(ops, cfg, bg, d1.cuda(), d2.cuda(), sf.cuda())
    _compare(ref, sums, sc, g1, g2, gs)


def test_outputs_are_bitwise_reproducible_run_to_run(warp_variant):
    """Two launches on the same inputs give the same BITS: the four sums, g_depth_1, g_sf and -- as long as no tap leaves the
    on-chip windows -- g_depth_2 (Q31.32 integer accumulation in LDS, slabs summed in a fixed order; the strip kernel's
    LDS-direct loads, counted waits and ring recycling included: a race there would show up here first).  Taps that do leave
    the windows are applied with hardware fp32 atomics: g_depth_2 then only to 1e-6 of its largest element.  The direct
    variant accumulates g_depth_2 with atomics everywhere."""
    from dvd_hip import ops, synthetic
    B, H, W = 3, 192, 288
    batch = synthetic.make_batch(B, H, W, device='cuda', with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W, device='cuda')
    sf = synthetic.make_scene_flow(B, H, W, device='cuda')
    cams = {k: batch[k] for k in CAM_KEYS}
    cfg = ops.warp_cfg(B, H, W)
    for scale, exact in ((0.25, True), (1.0, False)):
        flow = (batch['flow_1_2'] * scale).contiguous()
        runs = []
        for _ in range(4):
            o = ops.warp_loss_fused(cfg, d1, d2, flow, batch['mask_2'], sf, cams)
            torch.cuda.synchronize()
            runs.append([t.clone() for t in o])
        for o in runs[1:]:
            assert torch.equal(o[0], runs[0][0]) and torch.equal(o[1], runs[0][1]) and torch.equal(o[3], runs[0][3])
            if exact and warp_variant != 'direct':
                assert torch.equal(o[2], runs[0][2])
            else:
                assert float((o[2] - runs[0][2]).abs().max()) <= 1e-6 * float(runs[0][2].abs().max())
