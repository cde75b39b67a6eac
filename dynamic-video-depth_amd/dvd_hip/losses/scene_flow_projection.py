"""Geometry modules with the reference's names and forward signatures, on HIP kernels.

Counterparts of /root/reference/losses/scene_flow_projection.py:
  unproject_ptcld              :48-67    depth -> world points [B,H,W,1,3]
  flow_by_depth                :95-153   (forward signature kept; see below)
  scene_flow_projection_slack  :204-278  (forward signature kept; see below)
  BackwardWarp                 :281-307

The training step does not go through these modules: Model._train_on_batch calls the
fused kernel (dvd_warp_loss_fused), which never materialises the ten per-pixel
surfaces.  The module forms exist for the drop-in surface (the reference Model
discovers their argument names with inspect, scene_flow_motion_field.py:128-138) and
for visualisation / inference code that wants the surfaces.
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


class _SurfaceModule(nn.Module):
    what = ''

    def _unavailable(self):
        raise NotImplementedError(
            '%s: the per-pixel surface form is not built yet (round 1 ships the fused training kernel '
            'dvd_warp_loss_fused and unproject); see DESIGN.md "next"' % self.what)


class flow_by_depth(_SurfaceModule):
    what = 'flow_by_depth'

    def __init__(self, is_one_way=True):
        super().__init__()
        self.one_way = is_one_way

    def forward(self, depth_1, depth_2, flow_1_2, R_1, R_2, R_1_T, R_2_T, t_1, t_2, K, K_inv):
        self._unavailable()


class scene_flow_projection_slack(_SurfaceModule):
    what = 'scene_flow_projection_slack'

    def __init__(self, is_one_way=False):
        super().__init__()
        self.is_one_way = is_one_way

    def forward(self, depth_1, depth_2, flow_1_2, flow_2_1, R_1, R_2, R_1_T, R_2_T, t_1, t_2, K, K_inv, sflow_1_2,
                sflow_2_1):
        self._unavailable()


class BackwardWarp(_SurfaceModule):
    what = 'BackwardWarp'

    def __init__(self, is_one_way=False):
        super().__init__()
        self.is_one_way = is_one_way

    def forward(self, buffer, flow_1_2):
        self._unavailable()
