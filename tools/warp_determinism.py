"""Run-to-run reproducibility of dvd_warp_loss_fused: six launches per (shape, flow scale, kernel generation) on the same inputs,
how many differ from the first in each output (tests/test_00::test_outputs_are_bitwise_reproducible_run_to_run is the test form)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dynamic-video-depth_amd'))
from dvd_hip import ops, synthetic
CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')
def run(B,H,W,scale,variant,shape=0,n=6):
    ops.warp_loss_select(variant=variant, strip_shape=shape)
    batch = synthetic.make_batch(B, H, W, device='cuda', with_images=False)
    batch['flow_1_2'] = batch['flow_1_2']*scale
    d1, d2 = synthetic.make_depths(B, H, W, device='cuda')
    sf = synthetic.make_scene_flow(B, H, W, device='cuda')
    cams = {k: batch[k] for k in CAM_KEYS}
    cfg = ops.warp_cfg(B, H, W)
    outs=[]
    for i in range(n):
        o = ops.warp_loss_fused(cfg, d1, d2, batch['flow_1_2'], batch['mask_2'], sf, cams)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in o])
    bad=[0,0,0,0]
    for o in outs[1:]:
        for j in range(4):
            if not torch.equal(o[j], outs[0][j]): bad[j]+=1
    print(B,H,W,'flow x',scale,variant,shape,'mismatching runs per output [sums,g1,g2,gs]:',bad, 'max g2 diff', max(float((o[2]-outs[0][2]).abs().max()) for o in outs[1:]))
for (B,H,W) in ((48,384,672),(2,64,96),(2,32,48)):
    for scale in (0.25,1.0):
        for variant,shape in (('tiled',0),('tiled',1),('tiles',0)):
            run(B,H,W,scale,variant,shape)
