"""Oracle (test-only): unproject / flow-advect / reproject geometry on the CPU.

Restates, op for op, the arithmetic of the reference modules in
/root/reference/losses/scene_flow_projection.py so that results are
bit-identical to them on the same host:

  * pixel_grid            <- the `self.coord` cache, :116-121 / :224-229
  * flow_sample           <- `backward_warp`, :103-112 / :212-220 (F.grid_sample,
                             bilinear, border padding, align_corners=True)
  * unproject             <- `unproject_ptcld.forward`, :54-67
  * static_reprojection   <- `flow_by_depth.forward`, :114-153
  * dynamic_reprojection  <- `scene_flow_projection_slack.forward`, :222-278

Conventions (SURVEY.md section 8 header): row vectors, matrices stored transposed so that
`p @ M` applies the transform.  R_k = cam->world, R_k_T = world->cam,
K = intrinsics^T, K_inv = (intrinsics^-1)^T, all [B,1,1,3,3]; t_k [B,1,1,1,3].
"""

import torch
import torch.nn.functional as F

BEHIND_CAMERA_Z = 1e-3   # losses/scene_flow_projection.py:144,253,257,261
DIV_EPS = 1e-8           # :142,250-252


def pixel_grid(H, W, device=None):
    """[1,H,W,1,3] homogeneous pixel coordinates (x, y, 1)."""
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    g = torch.ones([1, H, W, 1, 3])
    g[0, :, :, 0, 0] = xs
    g[0, :, :, 0, 1] = ys
    return g if device is None else g.to(device)


def flow_sample(src, flow, grid):
    """Bilinear sample of `src` [B,C,H,W] at (x,y)+flow, border clamp.

    The normalise -> un-normalise round trip is kept exactly as the reference
    does it (in-place divides on the sum), because the five fp32 roundings
    decide the tap indices.
    """
    B, _, H, W = src.shape
    xy = grid[..., :2].view(1, H, W, 2).expand([B, H, W, 2])
    loc = xy + flow
    loc[..., 0] /= (W - 1) / 2
    loc[..., 1] /= (H - 1) / 2
    loc -= 1
    return F.grid_sample(src, loc, align_corners=True, padding_mode='border')


def camera_points(depth, grid, K_inv):
    """depth [B,1,H,W] -> camera-space points [B,H,W,1,3]."""
    B, _, H, W = depth.shape
    return depth.view([B, H, W, 1, 1]) * torch.matmul(grid, K_inv)


def unproject(depth_1, R_1, t_1, K_inv, grid=None):
    B, _, H, W = depth_1.shape
    grid = pixel_grid(H, W, depth_1.device) if grid is None else grid
    return torch.matmul(camera_points(depth_1, grid, K_inv), R_1) + t_1


def _project_with_fallback(p_cam, K, own_xy):
    """Perspective divide with the behind-camera index mask.

    Returns (xy [B,H,W,1,2], homogeneous image point, bool mask of the
    overwritten pixels).  Pixels with z < 1e-3 get their own pixel coordinates
    (zero flow) through index assignment, which also cuts their gradient.
    """
    p_img = torch.matmul(p_cam, K)
    xy = (p_img / (p_img[..., -1:] + DIV_EPS))[..., :-1]
    iB, iH, iW, iC, iF = torch.where(p_img[..., -1:] < BEHIND_CAMERA_Z)
    xy[iB, iH, iW, iC, iF] = own_xy[iB, iH, iW, iC, iF]
    xy[iB, iH, iW, iC, iF + 1] = own_xy[iB, iH, iW, iC, iF + 1]
    return xy, p_img, (p_img[..., -1:] < BEHIND_CAMERA_Z)


def static_reprojection(depth_1, depth_2, flow_1_2, R_1, R_2, R_1_T, R_2_T, t_1, t_2, K, K_inv):
    """flow_by_depth.forward: rigid-scene flow + scene flow implied by depth."""
    B, _, H, W = depth_1.shape
    grid = pixel_grid(H, W, depth_1.device)
    own = grid.expand([B, H, W, 1, 3])
    pc1 = camera_points(depth_1, grid, K_inv)
    pc2 = camera_points(depth_2, grid, K_inv)
    P1 = torch.matmul(pc1, R_1) + t_1
    P2 = torch.matmul(pc2, R_2) + t_2
    P2w = flow_sample(P2.squeeze(3).permute([0, 3, 1, 2]), flow_1_2, grid)
    P2w = P2w.permute([0, 2, 3, 1])[..., None, :]
    sf_by_depth = P2w - P1
    q = torch.matmul(P1 - t_2, R_2_T)
    xy, p_img, behind = _project_with_fallback(q, K, own[..., :-1])
    dflow = (xy - own[..., :-1])[..., 0, :]
    return {'dflow_1_2': dflow, 'sf_by_depth': sf_by_depth, 'warped_global_p2': P2w,
            'global_p1': P1, '_behind': behind}


def dynamic_reprojection(depth_1, depth_2, flow_1_2, flow_2_1, R_1, R_2, R_1_T, R_2_T,
                         t_1, t_2, K, K_inv, sflow_1_2, sflow_2_1):
    """scene_flow_projection_slack.forward (all ten returned surfaces)."""
    B, _, H, W = depth_1.shape
    grid = pixel_grid(H, W, depth_1.device)
    own = grid.expand([B, H, W, 1, 3])
    own_xy = own[..., :-1]
    pc1 = camera_points(depth_1, grid, K_inv)
    pc2 = camera_points(depth_2, grid, K_inv)
    P1 = torch.matmul(pc1, R_1) + t_1
    P2 = torch.matmul(pc2, R_2) + t_2

    W2 = flow_sample(pc2.squeeze(3).permute([0, 3, 1, 2]), flow_1_2, grid)
    W2 = W2.permute([0, 2, 3, 1])[..., None, :]

    q_dyn = torch.matmul(P1 + sflow_1_2 - t_2, R_2_T)
    q_sta = torch.matmul(P1 - t_2, R_2_T)
    q_rev = torch.matmul(P2 + sflow_2_1 - t_1, R_1_T)
    # same evaluation order as the reference (:247-263); the reverse branch
    # is dead but harmless.
    xy_dyn, img_dyn, behind_dyn = _project_with_fallback(q_dyn, K, own_xy)
    _xy_rev, _img_rev, behind_rev = _project_with_fallback(q_rev, K, own_xy)
    xy_sta, _img_sta, behind_sta = _project_with_fallback(q_sta, K, own_xy)

    dflow = (xy_dyn - own_xy)[..., 0, :]
    sflow_static = (xy_sta - own_xy)[..., 0, :]
    depth_image = img_dyn[..., -1].permute(0, 3, 1, 2)
    d1 = depth_1.view(B, 1, H, W)
    d2 = depth_2.view(B, 1, H, W)
    depth_warp = flow_sample(d2, flow_1_2, grid).view([B, 1, H, W])
    return {'dflow_1_2': dflow, 'depth_image_1_2': depth_image, 'depth_warp_1_2': depth_warp,
            'depth_1': d1, 'depth_2': d2, 'scenef_1_2': sflow_1_2, 'global_p1': P1,
            'staticflow_1_2': sflow_static, 'p1_camera_2': q_dyn, 'warped_p2_camera_2': W2,
            '_behind': behind_dyn, '_behind_static': behind_sta, '_behind_reverse': behind_rev}


def tap_indices(flow_1_2, H, W):
    """Integer (x0, y0) of the north-west bilinear tap, as torch's CPU
    grid_sample derives them (fp32, same rounding chain).  Used by the
    bit-exactness tests of the tap-index mask."""
    B = flow_1_2.shape[0]
    grid = pixel_grid(H, W)
    loc = grid[..., :2].view(1, H, W, 2).expand([B, H, W, 2]) + flow_1_2
    loc[..., 0] /= (W - 1) / 2
    loc[..., 1] /= (H - 1) / 2
    loc -= 1
    ix = (loc[..., 0] + 1) * ((W - 1) / 2)
    iy = (loc[..., 1] + 1) * ((H - 1) / 2)
    ix = torch.clamp(ix, 0, W - 1)
    iy = torch.clamp(iy, 0, H - 1)
    return torch.floor(ix).to(torch.int32), torch.floor(iy).to(torch.int32)
