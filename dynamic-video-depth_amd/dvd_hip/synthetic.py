"""Synthetic frame-pair batches with the reference's train-batch schema.

The dict this returns is what `Model._train_on_batch` receives after the
DataLoader dimension is stripped (reference: datasets/davis_sequence.py:98-115
on top of scripts/preprocess/davis/generate_sequence_midas.py:117-170; schema
table in SURVEY.md Appendix A, generator recipe in SURVEY.md section 8d).  There are
no datasets or checkpoints in this environment, so tests and bench.py run on
these tensors; `data: "synthetic"` in the bench line refers to this module.
"""

import math

import torch


def _intrinsics(H, W):
    f = 0.9 * W
    K = torch.tensor([[f, 0.0, (W - 1) / 2.0], [0.0, f, (H - 1) / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    return K, torch.linalg.inv(K)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float64)


def make_batch(B, H, W, gap=1, seed=1234, rank=0, device='cpu', n_frames=100,
               behind_camera_pairs=0, far_depth_frac=0.0, with_images=True):
    """Returns the batch dict (fp32 tensors on `device`).

    behind_camera_pairs: the last k pairs get a camera-2 translation along +z
      large enough that every reprojected point has z < 1e-3, which exercises
      the behind-camera index mask (losses/scene_flow_projection.py:253-256).
    far_depth_frac is consumed by `make_depths` (pixels with depth 150 to
      exercise the [depth < 100] masks); kept here so one seed describes a case.
    """
    g = torch.Generator().manual_seed(seed + rank)
    K, Kinv = _intrinsics(H, W)
    batch = {}
    if with_images:
        batch['img_1'] = torch.rand(B, 3, H, W, generator=g)
        batch['img_2'] = torch.rand(B, 3, H, W, generator=g)
    flow = 3.0 * torch.randn(B, H, W, 2, generator=g)
    batch['flow_1_2'] = flow
    batch['flow_2_1'] = -flow
    m1 = (torch.rand(B, H, W, 1, 1, generator=g) < 0.9).float()
    m2 = (torch.rand(B, H, W, 1, 1, generator=g) < 0.9).float()
    batch['mask_1'] = m1
    batch['mask_2'] = m2
    batch['motion_seg_1'] = m2.clone()

    R1 = torch.eye(3, dtype=torch.float64)                 # cam->world of frame 1
    R2 = _rot_y(0.01 * gap)
    t1 = torch.zeros(3, dtype=torch.float64)
    t2 = torch.tensor([0.05 * gap, 0.0, 0.0], dtype=torch.float64)

    def rep(m):
        return m.float().view(1, 1, 1, *m.shape).repeat(B, 1, 1, *([1] * m.dim())).contiguous()

    # stored transposed: p_row @ R_k applies R_c2w (generate_sequence_midas.py:69-76)
    batch['R_1'] = rep(R1.T.contiguous())
    batch['R_1_T'] = rep(R1)
    batch['R_2'] = rep(R2.T.contiguous())
    batch['R_2_T'] = rep(R2)
    batch['t_1'] = t1.float().view(1, 1, 1, 1, 3).repeat(B, 1, 1, 1, 1).contiguous()
    t2b = t2.float().view(1, 1, 1, 1, 3).repeat(B, 1, 1, 1, 1).contiguous()
    for k in range(behind_camera_pairs):
        t2b[B - 1 - k, 0, 0, 0, 2] = 50.0
    batch['t_2'] = t2b
    batch['K'] = rep(K.T.contiguous())
    batch['K_inv'] = rep(Kinv.T.contiguous())

    fid1 = torch.arange(B, dtype=torch.float32) % max(n_frames - gap, 1)
    fid2 = fid1 + gap
    batch['frame_id_1'] = fid1
    batch['frame_id_2'] = fid2
    batch['time_stamp_1'] = (fid1 / n_frames).view(B, 1, 1, 1).expand(B, 1, H, W).contiguous()
    batch['time_stamp_2'] = (fid2 / n_frames).view(B, 1, 1, 1).expand(B, 1, H, W).contiguous()
    batch['time_step'] = torch.tensor(1.0 / n_frames, dtype=torch.float64)
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device) if k != 'time_step' else v
    return out


def make_depths(B, H, W, seed=99, far_depth_frac=0.001, device='cpu'):
    """Leaf depth maps for the stand-alone warp+loss cases: U[1,6) with a
    fraction of pixels at 150 (both maps) so [d<100] / [W2.z<100] fire."""
    g = torch.Generator().manual_seed(seed)
    d1 = 1.0 + 5.0 * torch.rand(B, 1, H, W, generator=g)
    d2 = 1.0 + 5.0 * torch.rand(B, 1, H, W, generator=g)
    if far_depth_frac > 0:
        d1[torch.rand(B, 1, H, W, generator=g) < far_depth_frac] = 150.0
        d2[torch.rand(B, 1, H, W, generator=g) < far_depth_frac] = 150.0
    return d1.to(device), d2.to(device)


def make_scene_flow(B, H, W, seed=7, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    return (0.01 * torch.randn(B, 3, H, W, generator=g)).to(device)


def with_loader_dim(batch):
    """Add the leading DataLoader dimension (batch_size=1, experiments/davis/train_sequence.sh:34)
    that Model._train_on_batch strips again (models/scene_flow_motion_field.py:177-179)."""
    return {k: (v.unsqueeze(0) if torch.is_tensor(v) else v) for k, v in batch.items()}
