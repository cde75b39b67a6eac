"""The module forms of the warp operators (flow_by_depth, scene_flow_projection_slack,
BackwardWarp, unproject_ptcld) on HIP against the fixtures written by the REAL reference
modules (tests/golden/make_golden.py) and against the CPU oracle at a larger size.

Tolerance: rtol/atol 2e-6 on the surfaces (same fp32 operation order as the reference; the
residual is FMA-vs-separate rounding inside ATen's own CPU kernels across hosts); the
behind-camera index mask and the bilinear tap indices are bit-exact."""
import numpy as np
import pytest
import torch

from helpers import CAM_KEYS, load_golden, t
from oracle import geometry as G

pytestmark = pytest.mark.gpu
TIGHT = dict(rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('name', ['geom_b2_24x32', 'geom_b3_16x40_behind'])
def test_module_forms_match_the_reference_fixtures(name):
    from dvd_hip.losses.scene_flow_projection import (BackwardWarp, flow_by_depth, scene_flow_projection_slack,
                                                      unproject_ptcld)
    gd = load_golden(name)
    cams = {k: t(gd['in_' + k]).cuda() for k in CAM_KEYS}
    d1, d2, flow = t(gd['in_depth_1']).cuda(), t(gd['in_depth_2']).cuda(), t(gd['in_flow_1_2']).cuda()
    sflow = t(gd['in_sf_1_2']).permute(0, 2, 3, 1)[..., None, :].contiguous().cuda()
    with torch.no_grad():
        st = flow_by_depth()(d1, d2, flow, **cams)
        dy = scene_flow_projection_slack()(d1, d2, flow, -flow, sflow_1_2=sflow, sflow_2_1=sflow, **cams)
        bw = BackwardWarp()(d2, flow)
        up = unproject_ptcld()(d1, cams['R_1'], cams['t_1'], cams['K_inv'])
    assert set(st) == {'dflow_1_2', 'sf_by_depth', 'warped_global_p2', 'global_p1'}
    assert set(dy) == {'dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'depth_1', 'depth_2', 'scenef_1_2',
                       'global_p1', 'staticflow_1_2', 'p1_camera_2', 'warped_p2_camera_2'}
    for k, v in st.items():
        assert tuple(v.shape) == gd['fbd_' + k].shape, k
        np.testing.assert_allclose(v.cpu().numpy(), gd['fbd_' + k], **TIGHT, err_msg='flow_by_depth.' + k)
    for k, v in dy.items():
        assert tuple(v.shape) == gd['slack_' + k].shape, k
        np.testing.assert_allclose(v.cpu().numpy(), gd['slack_' + k], **TIGHT, err_msg='slack.' + k)
    np.testing.assert_allclose(bw.cpu().numpy(), gd['bwarp_depth_2'], **TIGHT)
    np.testing.assert_allclose(up.cpu().numpy(), gd['unproject_global_p1'], **TIGHT)
    # index mask, bit exact: behind-camera pixels are exactly those the reference gave zero flow
    behind = (dy['depth_image_1_2'][:, 0] < 1e-3).cpu().numpy()
    zero_flow = (gd['slack_dflow_1_2'] == 0).all(-1)
    assert np.array_equal(behind, zero_flow & (gd['slack_depth_image_1_2'][:, 0] < 1e-3))
    assert np.array_equal(dy['dflow_1_2'].cpu().numpy()[behind], np.zeros_like(gd['slack_dflow_1_2'][behind]))
    if 'behind' in name:
        assert behind.any() and not behind.all()
    # backward of the module forms against the REAL reference's autograd: the fixture holds fixed upstream gradients of
    # every output of both modules and the resulting gradients of depth_1, depth_2 and the scene flow
    d1g, d2g = d1.clone().requires_grad_(True), d2.clone().requires_grad_(True)
    sfg = t(gd['in_sf_1_2']).cuda().requires_grad_(True)                     # planar leaf, as in make_golden.py
    sfl = sfg.permute(0, 2, 3, 1)[..., None, :]
    st = flow_by_depth()(d1g, d2g, flow, **cams)
    dy = scene_flow_projection_slack()(d1g, d2g, flow, -flow, sflow_1_2=sfl, sflow_2_1=sfl, **cams)
    total = 0.0
    for key, out in list(st.items()) + [('s_' + k, v) for k, v in dy.items()]:
        if 'up_' + key in gd:
            total = total + (out * t(gd['up_' + key]).cuda()).sum()
    total.backward()
    for got, key in ((d1g.grad, 'g_depth_1'), (d2g.grad, 'g_depth_2'), (sfg.grad, 'g_sf_1_2')):
        want = gd[key]
        err = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        assert err < 2e-5, '%s: %.2e of max|g|' % (key, err)


def test_surfaces_against_oracle_at_larger_size_and_autograd_contract():
    from dvd_hip import synthetic
    from dvd_hip.losses.scene_flow_projection import BackwardWarp, scene_flow_projection_slack
    B, H, W = 3, 96, 160
    batch = synthetic.make_batch(B, H, W, gap=2, seed=11, behind_camera_pairs=1, with_images=False)
    batch['flow_1_2'][0, :6] *= 15.0                       # targets far outside the image: border clamp
    d1, d2 = synthetic.make_depths(B, H, W, seed=3, far_depth_frac=0.01)
    sf = synthetic.make_scene_flow(B, H, W).permute(0, 2, 3, 1)[..., None, :].contiguous()
    cams = {k: batch[k] for k in CAM_KEYS}
    ref = G.dynamic_reprojection(d1, d2, batch['flow_1_2'], -batch['flow_1_2'], sflow_1_2=sf, sflow_2_1=sf, **cams)
    cg = {k: v.cuda() for k, v in cams.items()}
    with torch.no_grad():
        got = scene_flow_projection_slack()(d1.cuda(), d2.cuda(), batch['flow_1_2'].cuda(), None, sflow_1_2=sf.cuda(),
                                            sflow_2_1=None, **cg)
    for k in ('dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'global_p1', 'staticflow_1_2', 'p1_camera_2',
              'warped_p2_camera_2'):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k].numpy(), rtol=3e-6, atol=3e-5, err_msg=k)
    assert np.array_equal((got['depth_image_1_2'][:, 0] < 1e-3).cpu().numpy(), ref['_behind'][..., 0, 0].numpy())
    # the module forms are differentiable like the reference's: the oracle's autograd gradient of a random projection
    gref = {}
    for name, t0 in (('d1', d1), ('d2', d2), ('sf', sf)):
        gref[name] = t0.clone().requires_grad_(True)
    ref2 = G.dynamic_reprojection(gref['d1'], gref['d2'], batch['flow_1_2'], -batch['flow_1_2'], sflow_1_2=gref['sf'],
                                  sflow_2_1=gref['sf'], **cams)
    ggpu = {k: v.detach().clone().cuda().requires_grad_(True) for k, v in gref.items()}
    got2 = scene_flow_projection_slack()(ggpu['d1'], ggpu['d2'], batch['flow_1_2'].cuda(), None, sflow_1_2=ggpu['sf'],
                                         sflow_2_1=None, **cg)
    gen = torch.Generator().manual_seed(5)
    tot_ref = tot_gpu = 0.0
    for k in ('dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'global_p1', 'staticflow_1_2', 'p1_camera_2',
              'warped_p2_camera_2'):
        u = torch.randn(ref2[k].shape, generator=gen)
        tot_ref = tot_ref + (ref2[k] * u).sum()
        tot_gpu = tot_gpu + (got2[k] * u.cuda()).sum()
    tot_ref.backward()
    tot_gpu.backward()
    for name in ('d1', 'd2', 'sf'):
        want, got_g = gref[name].grad.numpy(), ggpu[name].grad.cpu().numpy()
        assert np.abs(got_g - want).max() <= 2e-5 * np.abs(want).max() + 1e-6, name
    # BackwardWarp is differentiable w.r.t. its buffer, like F.grid_sample
    buf = torch.randn(B, 3, H, W)
    up = torch.randn(B, 3, H, W)
    b_ref = buf.clone().requires_grad_(True)
    grid = G.pixel_grid(H, W)
    G.flow_sample(b_ref, batch['flow_1_2'].clone(), grid).backward(up)
    b_gpu = buf.cuda().requires_grad_(True)
    out = BackwardWarp()(b_gpu, batch['flow_1_2'].cuda())
    out.backward(up.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(),
                               G.flow_sample(buf, batch['flow_1_2'].clone(), grid).numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(b_gpu.grad.cpu().numpy(), b_ref.grad.numpy(), rtol=1e-5, atol=1e-5)
