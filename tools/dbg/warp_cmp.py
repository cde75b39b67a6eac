import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from dvd_hip import ops, synthetic
CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')
for (B, H, W) in ((2, 24, 32), (2, 96, 160), (2, 33, 50)):
    batch = synthetic.make_batch(B, H, W, device='cuda', with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W, device='cuda')
    sf = synthetic.make_scene_flow(B, H, W, device='cuda')
    cams = {k: batch[k] for k in CAM_KEYS}
    cfg = ops.warp_cfg(B, H, W)
    res = {}
    for name, kw in (('v5', dict()), ('px4', dict(px=4)), ('direct', dict(variant='direct'))):
        ops.warp_loss_select(**kw)
        out = ops.warp_loss_fused(cfg, d1, d2, batch['flow_1_2'], batch['mask_2'], sf, cams)
        torch.cuda.synchronize()
        res[name] = [o.clone() for o in out]
    ops.warp_loss_select()
    for name in ('px4', 'direct'):
        for i, nm in enumerate(('sums', 'g_d1', 'g_d2', 'g_sf')):
            a, b = res['v5'][i], res[name][i]
            print(B, H, W, 'v5 vs', name, nm, 'maxdiff %.3e' % float((a - b).abs().max()), 'max|ref| %.3e' % float(b.abs().max()),
                  'nonzero v5 %d ref %d' % (int((a != 0).sum()), int((b != 0).sum())))
    print('g_sf v5 sample', res['v5'][3][0, :, H // 2, W // 2 - 2:W // 2 + 2].tolist())
    print('g_sf px4 sample', res['px4'][3][0, :, H // 2, W // 2 - 2:W // 2 + 2].tolist())
