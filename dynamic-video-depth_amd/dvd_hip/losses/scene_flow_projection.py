"""Geometry modules with the reference's names and forward signatures, on HIP kernels.

Counterparts of /root/reference/losses/scene_flow_projection.py:
  unproject_ptcld              :48-67    depth -> world points [B,H,W,1,3]
  flow_by_depth                :95-153   (forward signature kept; see below)
  scene_flow_projection_slack  :204-278  (forward signature kept; see below)
  BackwardWarp                 :281-307

The training step does not go through these modules: Model._train_on_batch calls the
fused kernel (dvd_warp_loss_fused), which never materialises the ten per-pixel
surfaces.  The module forms exist for the drop-in surface (the reference Model
discovers their argument names with inspect, scene_flow_motion_field.py:128-138), for
visualisation / export / inference code that wants the surfaces, and for swapping a
single operator into the reference's own Model: they return the reference's dict keys,
shapes and values (dvd_warp_surfaces, same fp32 operation order) and are differentiable
w.r.t. depth_1, depth_2 and the scene flow exactly like the reference's autograd graph
(dvd_warp_surfaces_bwd, including the gradient cut at behind-camera pixels); the flow and
the cameras are data.  BackwardWarp is differentiable w.r.t. its buffer.
"""
import torch
from torch import nn

from .. import ops


class _Unproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, R, t, K_inv):
        ctx.save_for_backward(R, K_inv)
        return ops.unproject(depth, R, t, K_inv, planar=False)

    @staticmethod
    def backward(ctx, g):
        R, K_inv = ctx.saved_tensors
        return ops.unproject_backward(g.contiguous(), False, R, K_inv), None, None, None


class unproject_ptcld(nn.Module):
    def __init__(self, is_one_way=True):
        super().__init__()

    def forward(self, depth_1, R_1, t_1, K_inv):
        return _Unproject.apply(depth_1, R_1, t_1, K_inv)


class _WarpSurfaces(torch.autograd.Function):
    """The surfaces in `want` as a tuple (forward: dvd_warp_surfaces; backward: dvd_warp_surfaces_bwd)."""

    @staticmethod
    def forward(ctx, depth_1, depth_2, sflow_1_2, flow_1_2, want, *cam_tensors):
        cams = dict(zip(ops.CAM_KEYS, cam_tensors))
        s = ops.warp_surfaces(depth_1, depth_2, flow_1_2, cams, sflow_1_2=sflow_1_2, want=want)
        ctx.save_for_backward(depth_1, depth_2, flow_1_2, sflow_1_2, *cam_tensors)
        ctx.want = want
        return tuple(s[k] for k in want)

    @staticmethod
    def backward(ctx, *grads):
        depth_1, depth_2, flow_1_2, sflow_1_2 = ctx.saved_tensors[:4]
        cams = dict(zip(ops.CAM_KEYS, ctx.saved_tensors[4:]))
        g = {k: (gk.contiguous() if gk is not None else None) for k, gk in zip(ctx.want, grads)}
        g1, g2, gs = ops.warp_surfaces_backward(depth_1, depth_2, flow_1_2, cams, g, sflow_1_2=sflow_1_2,
                                                want_sflow_grad=sflow_1_2 is not None and ctx.needs_input_grad[2])
        return (g1, g2, gs, None, None) + (None,) * len(ops.CAM_KEYS)


def _surfaces(depth_1, depth_2, flow_1_2, cams, want, sflow_1_2=None):
    out = _WarpSurfaces.apply(depth_1, depth_2, sflow_1_2, flow_1_2, tuple(want), *[cams[k] for k in ops.CAM_KEYS])
    return dict(zip(want, out))


class flow_by_depth(nn.Module):
    """losses/scene_flow_projection.py:95-153: rigid-scene flow and the scene flow implied by the two depth maps."""

    def __init__(self, is_one_way=True):
        super().__init__()
        self.one_way = is_one_way

    def forward(self, depth_1, depth_2, flow_1_2, R_1, R_2, R_1_T, R_2_T, t_1, t_2, K, K_inv):
        cams = dict(R_1=R_1, R_2=R_2, R_1_T=R_1_T, R_2_T=R_2_T, t_1=t_1, t_2=t_2, K=K, K_inv=K_inv)
        s = _surfaces(depth_1, depth_2, flow_1_2, cams, ('staticflow_1_2', 'sf_by_depth', 'warped_global_p2', 'global_p1'))
        return {'dflow_1_2': s['staticflow_1_2'], 'sf_by_depth': s['sf_by_depth'],
                'warped_global_p2': s['warped_global_p2'], 'global_p1': s['global_p1']}


class scene_flow_projection_slack(nn.Module):
    """losses/scene_flow_projection.py:204-278: reprojection of the scene-flow-advected points (ten surfaces)."""

    def __init__(self, is_one_way=False):
        super().__init__()
        self.is_one_way = is_one_way

    def forward(self, depth_1, depth_2, flow_1_2, flow_2_1, R_1, R_2, R_1_T, R_2_T, t_1, t_2, K, K_inv, sflow_1_2,
                sflow_2_1):
        cams = dict(R_1=R_1, R_2=R_2, R_1_T=R_1_T, R_2_T=R_2_T, t_1=t_1, t_2=t_2, K=K, K_inv=K_inv)
        B, _, H, W = depth_1.shape
        s = _surfaces(depth_1, depth_2, flow_1_2, cams, ('dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'global_p1',
                                                         'staticflow_1_2', 'p1_camera_2', 'warped_p2_camera_2'),
                      sflow_1_2=sflow_1_2)
        s.update(depth_1=depth_1.view(B, 1, H, W), depth_2=depth_2.view(B, 1, H, W), scenef_1_2=sflow_1_2)
        return s


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buffer, flow_1_2):
        ctx.save_for_backward(flow_1_2)
        return ops.flow_warp(buffer, flow_1_2)

    @staticmethod
    def backward(ctx, g):
        (flow_1_2,) = ctx.saved_tensors
        return ops.flow_warp_backward(g.contiguous(), flow_1_2), None


class BackwardWarp(nn.Module):
    """losses/scene_flow_projection.py:281-307: bilinear sample of `buffer` [B,C,H,W] at (x,y)+flow,
    border clamp, align_corners=True; differentiable w.r.t. the buffer (flow is data)."""

    def __init__(self, is_one_way=False):
        super().__init__()
        self.is_one_way = is_one_way

    def forward(self, buffer, flow_1_2):
        return _FlowWarp.apply(buffer, flow_1_2)
