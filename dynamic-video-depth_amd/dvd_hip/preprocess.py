"""Mask preprocessing of a frame pair on the GPU.

Counterpart of /root/reference/scripts/preprocess/davis/generate_flows.py:57-82,139-148 (forward/backward
flow-consistency + out-of-bounds masks, computed there per pair in numpy after RAFT) and of the mask
conversion in scripts/preprocess/davis/generate_sequence_midas.py:144-147.  RAFT itself stays external: this
module starts from the two flow fields.
"""
import torch

from . import _lib
from .ops import _dev32, _p, _stream


def flow_consistency_masks(flow_1_2, flow_2_1):
    """flows [B,H,W,2] (or [H,W,2]) -> (mask_1, mask_2) uint8 [B,H,W], 1 = occluded or leaving the image:
    the arrays generate_flows.py stores as `mask_1` / `mask_2` (:139-153)."""
    single = flow_1_2.dim() == 3
    if single:
        flow_1_2, flow_2_1 = flow_1_2[None], flow_2_1[None]
    f12, f21 = _dev32(flow_1_2, 'flow_1_2'), _dev32(flow_2_1, 'flow_2_1')
    B, H, W, _ = f12.shape
    lib = _lib.load()
    out = []
    for a, b in ((f12, f21), (f21, f12)):         # mask_1 samples flow_1_2 along flow_2_1, mask_2 the reverse
        m = torch.empty(B, H, W, device=f12.device, dtype=torch.float32)
        _lib.check(lib.dvd_flow_consistency_mask(_p(a), _p(b), _p(m), B, H, W, _stream()), 'dvd_flow_consistency_mask')
        out.append(m.to(torch.uint8))
    return (out[0][0], out[1][0]) if single else (out[0], out[1])


def training_masks(mask_1, mask_2):
    """The `mask_1` / `mask_2` tensors of a training batch: 1 = valid, [B,H,W,1,1] float
    (generate_sequence_midas.py:144-147: 1 - ceil(mask))."""
    return tuple((1 - torch.ceil(m.float()))[..., None, None] for m in (mask_1, mask_2))
