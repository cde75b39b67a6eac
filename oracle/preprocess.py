"""Oracle (test-only): the occlusion / out-of-bounds masks of a frame pair, restated from
/root/reference/scripts/preprocess/davis/generate_flows.py (the script itself imports cv2, skimage and RAFT at
module level and cannot be imported here):
  get_oob_mask        :57-68
  backward_flow_warp  :71-82    (F.grid_sample with the default 'zeros' padding, align_corners=True)
  mask_k              :139-148  clip([||warp + flow|| > 1] + oob, 0, 1), stored as uint8
Pinned: tests/golden/flow_masks.npz holds the masks the reference's OWN functions / statements produce (cut out of
the script with `ast` and executed unmodified: tests/ref_exec.py, tests/golden/make_golden.py); this restatement is
bit-identical to them on every case (tests/test_oracle_vs_golden.py, tests/test_reference_interop_cpu.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def oob_mask(flow):
    """flow [H,W,2] (torch) -> float numpy [H,W]: target pixel outside the image (:57-68)."""
    H, W, _ = flow.shape
    hh, ww = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    tx, ty = ww + flow[..., 0], hh + flow[..., 1]
    m = (tx < 0).float() + (tx > W - 1).float() + (ty < 0).float() + (ty > H - 1).float()
    return (m > 0).float().numpy()


def backward_flow_warp(im2, flow_1_2):
    """im2 numpy [H,W,C], flow torch [H,W,2] -> numpy [H,W,C] (:71-82)."""
    H, W, _ = im2.shape
    hh, ww = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    grid = torch.stack([ww, hh], -1)[None] + flow_1_2[None]
    grid[..., 0] /= (W - 1) / 2
    grid[..., 1] /= (H - 1) / 2
    grid -= 1
    im = torch.from_numpy(im2).float().permute(2, 0, 1)[None]
    return F.grid_sample(im, grid, align_corners=True)[0].permute(1, 2, 0).numpy()


def consistency_masks(flow_1_2, flow_2_1):
    """numpy float32 [H,W,2] x 2 -> (mask_1, mask_2) uint8 [H,W] (:139-148)."""
    out = []
    for fa, fb in ((flow_1_2, flow_2_1), (flow_2_1, flow_1_2)):
        warp = backward_flow_warp(fa, torch.from_numpy(fb))
        err = np.linalg.norm(warp + fb, axis=-1)
        m = np.where(err > 1, 1, 0) + oob_mask(torch.from_numpy(fb))
        out.append(np.clip(m, a_min=0, a_max=1).astype(np.uint8))
    return out[0], out[1]
