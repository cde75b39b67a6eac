"""Tensor-level wrappers over the C ABI (include/dvd_hip.h).

These functions only validate, allocate outputs with torch (so the caching
allocator owns every buffer) and enqueue the HIP kernels on torch's current
stream.  They never touch tensor contents on the host and never fall back to
PyTorch arithmetic: a tensor that is not on a GPU is an error.
"""
import ctypes
import os
import sys

import torch

from . import _lib

CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev32(t, name):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError('%s must be a GPU tensor (dvd_hip has no CPU path)' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be float32, got %s' % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def pack_cameras(cams):
    """dict of the eight camera tensors ([B,1,1,3,3] / [B,1,1,1,3] or flat)
    -> (ctypes struct, list of tensors kept alive)."""
    keep, st = [], _lib.Cameras()
    for k in CAM_KEYS:
        t = _dev32(cams[k], k)
        keep.append(t)
        setattr(st, k, t.data_ptr())
    return st, keep


def warp_cfg(B, H, W, midas_mask=True, crit_l2=False, disp_mode=1, loss_on_sf=False, flow_mul=1.0, disp_mul=1.0):
    return _lib.WarpCfg(int(B), int(H), int(W), int(bool(midas_mask)), int(bool(crit_l2)), int(disp_mode),
                        int(bool(loss_on_sf)), float(flow_mul), float(disp_mul))


def unproject(depth, R, t, K_inv, planar=True):
    """depth [B,1,H,W] -> world points, [B,3,H,W] (planar) or [B,H,W,1,3]."""
    depth = _dev32(depth, 'depth')
    R, t, K_inv = _dev32(R, 'R'), _dev32(t, 't'), _dev32(K_inv, 'K_inv')
    B, _, H, W = depth.shape
    out = torch.empty((B, 3, H, W) if planar else (B, H, W, 1, 3), device=depth.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.dvd_unproject_fwd(_p(depth), _p(R), _p(t), _p(K_inv), _p(out), int(planar), B, H, W, _stream()),
               'dvd_unproject_fwd')
    return out


def unproject_backward(g_points, planar, R, K_inv, scale=None, out=None, accumulate=False):
    """g_depth (+)= scale * ((g_points @ R^T) . ray);  scale: 1-element GPU tensor or None."""
    g_points = _dev32(g_points, 'g_points')
    R, K_inv = _dev32(R, 'R'), _dev32(K_inv, 'K_inv')
    if planar:
        B, _, H, W = g_points.shape
    else:
        B, H, W = g_points.shape[:3]
    if out is None:
        if accumulate:
            raise RuntimeError('accumulate=True needs an output tensor')
        out = torch.empty((B, 1, H, W), device=g_points.device, dtype=torch.float32)
    out = _dev32(out, 'g_depth')
    lib = _lib.load()
    _lib.check(lib.dvd_unproject_bwd(_p(g_points), int(planar), _p(R), _p(K_inv), _p(scale), _p(out),
                                     int(bool(accumulate)), B, H, W, _stream()), 'dvd_unproject_bwd')
    return out


# -- max|x| scalars ----------------------------------------------------------------------------------------------------
# Every tensor that feeds a matrix kernel carries max|x| (or an upper bound) in a 1-element GPU tensor: the kernels derive
# the power-of-two scale of their fp16 operand split from it on the device (csrc/dvd_split.h); the host never reads it.
# The scalar hangs on the tensor OBJECT together with the tensor's version counter (an in-place update invalidates it)
# and the capture context it was computed in.
_capture_state = [0, False, False]     # [capture generation, the previous _capture_gen() call was inside a capture, announced]


def begin_capture():
    """Call right before every `torch.cuda.graph(...)`: opens a new capture generation.  Scalars (and scalar-pool chunks)
    of one capture are never continued by another -- a chunk lives in the private pool of the graph that allocated it and
    its zero-fill node replays with THAT graph only (round 3 inferred the boundary from an eager call happening between two
    captures; back-to-back captures of two kept slots then shared a chunk: slot N+1's scalars were zeroed by slot N's
    replay, or never)."""
    _capture_state[0] += 1
    _capture_state[2] = True
    return _capture_state[0]


def _capture_gen():
    """0 outside a capture; inside one, the capture's generation.  A scalar computed eagerly must never be baked into a
    graph (the replay would scale new data with the warm-up pass's maximum), and a scalar that lives in a graph's private
    pool must not be used outside it.  The generation is opened by begin_capture() -- and, for a capture nobody announced
    (tests, tools, future capture sites: ADVICE round 4), by the first call made inside a capture after a call made outside
    one: the two mechanisms together leave only back-to-back unannounced captures with no eager call in between sharing
    a generation."""
    st = _capture_state
    if not torch.cuda.is_current_stream_capturing():
        st[1] = False
        return 0
    if not st[1] and not st[2]:         # first call inside a capture nobody announced: a generation of its own all the same
        st[0] += 1
    # the announcement is consumed by EVERY in-capture call (two announced captures back to back, with no eager
    # call in between, left it set -- the next unannounced capture then shared the second one's generation and pool chunk)
    st[2] = False
    st[1] = True
    return st[0]


_scalar_pool = {}        # (device index, capture generation) -> [zeroed chunk, next free element]


def new_scalar(device):
    """A zeroed 1-element GPU tensor for a max|.| scalar.  Scalars are handed out from zero-filled chunks of 512 (one fill
    launch per 512 instead of one per scalar: a MiDaS step needs ~800 of them); a chunk belongs to the capture context it
    was allocated in (inside a HIP graph the fill is part of the graph, so every replay starts from zeros again)."""
    gen = _capture_gen()
    key = (torch.device(device).index or 0, gen)
    for k in [k for k in _scalar_pool if k[1] not in (0, gen)]:       # chunks of finished captures: their graphs own them
        del _scalar_pool[k]
    ent = _scalar_pool.get(key)
    if ent is None or ent[1] >= ent[0].numel():
        ent = [torch.zeros(512, device=device, dtype=torch.float32), 0]
        _scalar_pool[key] = ent
    out = ent[0][ent[1]:ent[1] + 1]
    ent[1] += 1
    return out


_AMAX_LOG = {} if os.environ.get('DVD_AMAX_LOG') else None      # developer aid: who still needs a reduction pass (printed at exit)
if _AMAX_LOG is not None:
    import atexit
    import traceback

    def _amax_report():
        for k, n in sorted(_AMAX_LOG.items(), key=lambda kv: -kv[1]):
            print('amax x%d %s' % (n, k), file=sys.stderr)
    atexit.register(_amax_report)


def amax(t):
    """max|t| by the reduction kernel (one read of the tensor) -> 1-element GPU tensor."""
    t = _dev32(t, 'tensor')
    if _AMAX_LOG is not None:
        fr = traceback.extract_stack(limit=6)[:-1]
        key = (tuple(t.shape), ' < '.join('%s:%d' % (f.name, f.lineno) for f in reversed(fr)))
        _AMAX_LOG[key] = _AMAX_LOG.get(key, 0) + 1
    out = new_scalar(t.device)
    if t.numel():
        _lib.check(_lib.load().dvd_amax(_p(t), ctypes.c_longlong(t.numel()), _p(out), _stream()), 'dvd_amax')
    return out


def chansum(t, want_amax=False):
    """Per-channel sums of an NCHW fp32 tensor (a convolution's bias gradient) and, if want_amax, max|t| from the same read
    (attached to t like amax_of would) -> [C] tensor."""
    t = _dev32(t, 'tensor')
    N, C = t.shape[0], t.shape[1]
    HW = t.numel() // max(N * C, 1)
    lib = _lib.load()
    out = torch.empty(C, device=t.device, dtype=torch.float32)
    am = new_scalar(t.device) if want_amax else None
    nws = lib.dvd_chansum_workspace_bytes(N, C)
    ws = torch.empty(max(nws, 4), device=t.device, dtype=torch.uint8)
    _lib.check(lib.dvd_chansum(_p(t), N, C, ctypes.c_longlong(HW), _p(out), _p(am) if am is not None else None, _p(ws),
                               ctypes.c_size_t(nws), _stream()), 'dvd_chansum')
    if am is not None:
        t._dvd_amax = (t._version, am, _capture_gen())
    return out


def set_amax(t, am):
    """Attach a known max|t| (or upper bound: a sub-sampled / interpolated / ReLU'd view of a bounded tensor)."""
    if am is not None:
        t._dvd_amax = (t._version, am, _capture_gen())
    return t


def known_amax(t):
    """The scalar attached to the tensor if it is still valid (same contents, same capture context), else None."""
    hit = getattr(t, '_dvd_amax', None)
    if hit is not None and hit[0] == t._version and hit[2] == _capture_gen():
        return hit[1]
    return None


def amax_of(t):
    """The tensor's max|.| scalar: the one its producer attached, else computed now and remembered."""
    am = known_amax(t)
    if am is None:
        am = amax(t)
        t._dvd_amax = (t._version, am, _capture_gen())
    return am


_ws_cache = {}


def _workspace(nbytes, device):
    if torch.cuda.is_current_stream_capturing():
        # an allocation made while a HIP graph is being captured lives in THAT graph's private pool: it must stay private
        # to the graph (a cached one would outlive the pool -- a memory-access fault in whichever model reused it after
        # the first model's graphs were destroyed)
        return torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


def warp_loss_fused(cfg, depth_1, depth_2, flow_1_2, mask_2, sf_1_2, cams, grads=True, out=None):
    """One launch: the four un-normalised loss sums and, if grads, the
    un-normalised gradients w.r.t. depth_1, depth_2 and the scene flow.

    mask_2: [B,H,W] (or [B,H,W,1,1]); sf_1_2: planar [B,3,H,W].
    Returns (sums[4], g_depth_1, g_depth_2, g_sf) -- gradients None if not grads.
    `out` may pass pre-allocated (sums, g_depth_1, g_depth_2, g_sf) to avoid allocation.
    """
    depth_1, depth_2 = _dev32(depth_1, 'depth_1'), _dev32(depth_2, 'depth_2')
    flow_1_2, mask_2, sf_1_2 = _dev32(flow_1_2, 'flow_1_2'), _dev32(mask_2, 'mask_2'), _dev32(sf_1_2, 'sf_1_2')
    B, H, W = cfg.B, cfg.H, cfg.W
    if depth_1.numel() != B * H * W or depth_2.numel() != B * H * W or mask_2.numel() != B * H * W:
        raise RuntimeError('warp_loss_fused: depth/mask shape does not match cfg %dx%dx%d' % (B, H, W))
    if flow_1_2.numel() != 2 * B * H * W or sf_1_2.numel() != 3 * B * H * W:
        raise RuntimeError('warp_loss_fused: flow / scene-flow shape mismatch')
    cst, keep = pack_cameras(cams)
    dev = depth_1.device
    lib = _lib.load()
    nws = lib.dvd_warp_loss_workspace_bytes(B, H, W)
    ws = _workspace(nws, dev)
    if out is not None:
        sums, g1, g2, gs = out
    else:
        sums = torch.empty(4, device=dev, dtype=torch.float32)
        g1 = torch.empty_like(depth_1) if grads else None
        g2 = torch.empty_like(depth_2) if grads else None
        gs = torch.empty_like(sf_1_2) if grads else None
    if grads:
        st = lib.dvd_warp_loss_fused(ctypes.byref(cfg), _p(depth_1), _p(depth_2), _p(flow_1_2), _p(mask_2),
                                     _p(sf_1_2), ctypes.byref(cst), _p(ws), ws.numel(), _p(sums), _p(g1), _p(g2),
                                     _p(gs), _stream())
        _lib.check(st, 'dvd_warp_loss_fused')
    else:
        st = lib.dvd_warp_loss_fwd(ctypes.byref(cfg), _p(depth_1), _p(depth_2), _p(flow_1_2), _p(mask_2),
                                   _p(sf_1_2), ctypes.byref(cst), _p(ws), ws.numel(), _p(sums), _stream())
        _lib.check(st, 'dvd_warp_loss_fwd')
    del keep
    return sums, g1, g2, gs


FLOP_CLASSES = ('xconv_1x1_wide', 'xconv_wide', 'xconv_128', 'xconv_small', 'xwgrad3', 'xwgrad3g', 'xwgrad1b', 'xwgrad1s',
                'xwgradk', 'mlp_fwd', 'mlp_bwd_dx', 'mlp_bwd_dw')
# kernel-name fragments (mangled names of a rocprofv3 trace) of each class: tools/mfma_roofline.py joins a trace on these
FLOP_CLASS_KERNELS = {          # mangled (rocpd database) and demangled (rocprofv3 csv) spellings
    'xconv_1x1_wide': ('xconv_kernelILi4ELi2ELi2ELi2ELi11E', 'xconv_kernel<4, 2, 2, 2, 11,'),
    'xconv_wide': ('xconv_kernelILi4ELi2ELi2ELi4E', 'xconv_kernelILi4ELi2ELi2ELi2ELi0E', 'xconv_kernelILi4ELi2ELi2ELi2ELi1E',
                   'xconv_kernelILi4ELi2ELi2ELi2ELi2E', 'xconv_kernelILi4ELi2ELi2ELi2ELi3E', 'xconv_kernel<4, 2, 2, 4,',
                   'xconv_kernel<4, 2, 2, 2, 0,', 'xconv_kernel<4, 2, 2, 2, 1,', 'xconv_kernel<4, 2, 2, 2, 2,',
                   'xconv_kernel<4, 2, 2, 2, 3,'),
    'xconv_128': ('xconv_kernelILi2ELi2ELi2ELi2E', 'xconv_kernel<2, 2, 2, 2,'),
    'xconv_small': ('xconv_kernelILi1ELi2ELi1ELi4E', 'xconv_kernelILi2ELi2ELi1ELi4E', 'xconv_kernel<1, 2, 1, 4,',
                    'xconv_kernel<2, 2, 1, 4,'),
    'xwgrad3': ('xwgrad3_kernel',), 'xwgrad3g': ('xwgrad3g_kernel',), 'xwgrad1b': ('xwgrad1b_kernel',),
    'xwgrad1s': ('xwgrad1s_kernel',), 'xwgradk': ('xwgradk_kernel',),
    'mlp_fwd': ('mlp_fwd_kernel',), 'mlp_bwd_dx': ('mlp_bwd_dx_kernel',), 'mlp_bwd_dw': ('mlp_bwd_dw_kernel',),
}


# Memory-bound helper kernels (round 6; bench.py's roofline_helpers): algorithmic BYTES per class, counted by the library the
# same way (include/dvd_hip.h: dvd_byte_counters), and the kernel-name fragments tools/mfma_roofline.py joins a trace on.
# They travel through the same snapshot / replay bookkeeping as the matrix classes, under their own keys.
BYTE_CLASSES = ('bnrelu_fwd', 'bnrelu_bwd', 'upsample_fwd', 'upsample_bwd', 'amax', 'pack', 'pool', 'gconv_c8', 'elementwise',
                'adam', 'geometry')
BYTE_CLASS_KERNELS = {
    'bnrelu_fwd': ('bnrelu_fwd_kernel',), 'bnrelu_bwd': ('bnrelu_bwd', 'bn_mask', 'bnrelu_sum'),
    'upsample_fwd': ('upsample_bilinear_fwd',), 'upsample_bwd': ('upsample_bilinear_bwd', 'upsample_bwd'),
    'amax': ('amax_kernel', 'chansum_'), 'pack': ('xconv_wamax', 'xconv_pack_kernel'), 'pool': ('maxpool3s2', 'subsample2_', 'avgpool_'),
    'gconv_c8': ('gconv3x3_c8',), 'elementwise': ('mul_mask_kernel', 'scale_add_kernel', 'acc_reg_kernel', 'sum_partials_kernel',
                                                  'head1x1_', 'cast_scale_kernel'),
    'adam': ('adam_kernel',), 'geometry': ('unproject_',),
}
ALL_CLASSES = FLOP_CLASSES + BYTE_CLASSES


def flop_counters(reset=False):
    """Algorithmic work (2 x MACs) the matrix kernels were launched with since the last reset, per kernel class
    (include/dvd_hip.h: dvd_flop_counters) -- and, under the BYTE_CLASSES keys, the algorithmic bytes of the helper kernels
    (dvd_byte_counters).  Counted at launch / graph-capture time: read it after a step that CAPTURED (or ran eagerly)
    everything a step launches."""
    lib = _lib.load()
    buf = (ctypes.c_double * len(FLOP_CLASSES))()
    _lib.check(lib.dvd_flop_counters(ctypes.cast(buf, ctypes.c_void_p), len(FLOP_CLASSES), 1 if reset else 0), 'dvd_flop_counters')
    out = dict(zip(FLOP_CLASSES, [float(v) for v in buf]))
    bbuf = (ctypes.c_double * len(BYTE_CLASSES))()
    _lib.check(lib.dvd_byte_counters(ctypes.cast(bbuf, ctypes.c_void_p), len(BYTE_CLASSES), 1 if reset else 0), 'dvd_byte_counters')
    out.update(zip(BYTE_CLASSES, [float(v) for v in bbuf]))
    return out


# work of REPLAYED HIP graphs: a graph's launches are counted once, when it is captured (flops_since around the capture);
# whoever replays it adds that record here, so that (flop_counters + REPLAYED) over a region is the work the region executed
REPLAYED = {k: 0.0 for k in ALL_CLASSES}


# work that is a RE-computation (round 6; SURVEY 8d: "recompute is not counted"): the no-graph depth-net forward of a chunk whose
# autograd state is not kept (phase 3 runs that forward again inside its forward+backward graph) and the stash-free Euler
# chain of the scene-flow MLP's recompute schedule.  Included in executed_flops(); bench.py reports the step's work with and
# without it (roofline_mfma.needed_TFLOP_per_step, frac_of_needed_work).
RECOMPUTED = {k: 0.0 for k in ALL_CLASSES}


class counting_recomputed(object):
    """with ops.counting_recomputed(): ...   -- everything launched / replayed inside is a re-computation."""

    def __enter__(self):
        self.before = executed_flops()
        return self

    def __exit__(self, *exc):
        now = executed_flops()
        for k in ALL_CLASSES:
            RECOMPUTED[k] += now[k] - self.before[k]
        return False


def flops_since(snapshot):
    now = flop_counters()
    return {k: now[k] - snapshot.get(k, 0.0) for k in ALL_CLASSES}


def note_replay(flops):
    if flops:
        for k, v in flops.items():
            REPLAYED[k] += v


def executed_flops():
    """Per kernel class: work launched eagerly plus work of replayed graphs, since process start (take differences)."""
    now = flop_counters()
    return {k: now[k] + REPLAYED[k] for k in ALL_CLASSES}


def warp_loss_select(variant='tiled', tile=-1, px=0, strip_rows=0, strip_shape=0):
    """Test hook: run dvd_warp_loss_fused on another variant -- 'tiled' = production (the strip kernel where it applies, at
    tile = -1 and px = 0), 'direct' = global gathers + hardware atomics, 'tiles' = the tile kernel of rounds 2-5 whatever the
    call -- tile shape, pixels-per-step mapping, rows per strip unit or strip shape (csrc/warp_strip.hip kStripShapes).  Process wide; call warp_loss_select() to restore
    the production path."""
    lib = _lib.load()
    _lib.check(lib.dvd_warp_loss_select({'tiled': 0, 'direct': 1, 'tiles': 2}[variant], int(tile), int(px)), 'dvd_warp_loss_select')
    _lib.check(lib.dvd_warp_loss_strip_select(int(strip_rows), int(strip_shape)), 'dvd_warp_loss_strip_select')


def loss_finalize(cfg, sums, out=None):
    """scalars[8]: [0]=1/(S0+1e-8) [1]=loss [2]=flow [3]=disp [4]=sf [5]=S0."""
    sums = _dev32(sums, 'sums')
    if out is None:
        out = torch.empty(8, device=sums.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.dvd_loss_finalize(ctypes.byref(cfg), _p(sums), _p(out), _stream()), 'dvd_loss_finalize')
    return out


_SURFACE_SHAPES = {'global_p1': 3, 'warped_global_p2': 3, 'sf_by_depth': 3, 'p1_camera_2': 3, 'warped_p2_camera_2': 3,
                   'staticflow_1_2': 2, 'dflow_1_2': 2, 'depth_image_1_2': 1, 'depth_warp_1_2': 1}


def warp_surfaces(depth_1, depth_2, flow_1_2, cams, sflow_1_2=None, want=_lib.SURFACE_KEYS):
    """The per-pixel surfaces of flow_by_depth / scene_flow_projection_slack (forward values).
    Returns a dict name -> tensor: 3-vectors [B,H,W,1,3], flows [B,H,W,2], depths [B,1,H,W]."""
    depth_1, depth_2, flow_1_2 = _dev32(depth_1, 'depth_1'), _dev32(depth_2, 'depth_2'), _dev32(flow_1_2, 'flow_1_2')
    B, _, H, W = depth_1.shape
    if sflow_1_2 is not None:
        sflow_1_2 = _dev32(sflow_1_2, 'sflow_1_2')
        if sflow_1_2.numel() != 3 * B * H * W:
            raise RuntimeError('sflow_1_2 must be [B,H,W,1,3]')
    cst, keep = pack_cameras(cams)
    out, st = {}, _lib.Surfaces()
    for k in want:
        c = _SURFACE_SHAPES[k]
        shape = (B, H, W, 1, 3) if c == 3 else ((B, H, W, 2) if c == 2 else (B, 1, H, W))
        out[k] = torch.empty(shape, device=depth_1.device, dtype=torch.float32)
        setattr(st, k, out[k].data_ptr())
    lib = _lib.load()
    _lib.check(lib.dvd_warp_surfaces(_p(depth_1), _p(depth_2), _p(flow_1_2), _p(sflow_1_2), ctypes.byref(cst),
                                     ctypes.byref(st), B, H, W, _stream()), 'dvd_warp_surfaces')
    del keep
    return out


def warp_surfaces_backward(depth_1, depth_2, flow_1_2, cams, grads, sflow_1_2=None, want_sflow_grad=False):
    """VJP of warp_surfaces: `grads` maps surface name -> upstream gradient (same layout as the surface; missing / None =
    no gradient).  Returns (g_depth_1, g_depth_2, g_sflow_1_2 | None)."""
    depth_1, depth_2, flow_1_2 = _dev32(depth_1, 'depth_1'), _dev32(depth_2, 'depth_2'), _dev32(flow_1_2, 'flow_1_2')
    B, _, H, W = depth_1.shape
    if sflow_1_2 is not None:
        sflow_1_2 = _dev32(sflow_1_2, 'sflow_1_2')
    cst, keep = pack_cameras(cams)
    st = _lib.Surfaces()
    for k in _lib.SURFACE_KEYS:
        g = grads.get(k)
        if g is not None:
            g = _dev32(g, 'grad of ' + k)
            if g.numel() != _SURFACE_SHAPES[k] * B * H * W:
                raise RuntimeError('gradient of %s has the wrong size' % k)
            keep.append(g)
            setattr(st, k, g.data_ptr())
    g1, g2 = torch.empty_like(depth_1), torch.empty_like(depth_2)
    gs = torch.empty(B, H, W, 1, 3, device=depth_1.device, dtype=torch.float32) if want_sflow_grad else None
    _lib.check(_lib.load().dvd_warp_surfaces_bwd(_p(depth_1), _p(depth_2), _p(flow_1_2), _p(sflow_1_2), ctypes.byref(cst),
                                                 ctypes.byref(st), _p(g1), _p(g2), _p(gs), B, H, W, _stream()),
               'dvd_warp_surfaces_bwd')
    del keep
    return g1, g2, gs


def flow_warp(buffer, flow_1_2):
    buffer, flow_1_2 = _dev32(buffer, 'buffer'), _dev32(flow_1_2, 'flow_1_2')
    B, C, H, W = buffer.shape
    out = torch.empty_like(buffer)
    lib = _lib.load()
    _lib.check(lib.dvd_flow_warp_fwd(_p(buffer), _p(flow_1_2), _p(out), B, C, H, W, _stream()), 'dvd_flow_warp_fwd')
    return out


def flow_warp_backward(g_out, flow_1_2):
    g_out, flow_1_2 = _dev32(g_out, 'g_out'), _dev32(flow_1_2, 'flow_1_2')
    B, C, H, W = g_out.shape
    g = torch.empty_like(g_out)
    lib = _lib.load()
    _lib.check(lib.dvd_flow_warp_bwd(_p(g_out), _p(flow_1_2), _p(g), B, C, H, W, _stream()), 'dvd_flow_warp_bwd')
    return g


# ---------------------------------------------------------------------------------------
# scene-flow field MLP


def embed_frequencies(n_freq):
    """PeriodicEmbed(max_freq=N, N_freq=N).freqs, exactly as the reference builds them
    (torch.linspace(1, N + 1, N), networks/blocks.py:24)."""
    return torch.linspace(1, n_freq + 1, steps=n_freq)


class SceneFlowMLPKernels(object):
    """Host-side handle of the fused MLP kernels for one network configuration:
    owns the frequency tables, the packed-weight buffer and the scratch stashes."""

    def __init__(self, device, n_freq_xyz=16, n_freq_t=16, time_dependent=True, stash_f16=False):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('SceneFlowMLPKernels needs a GPU device (dvd_hip has no CPU path)')
        self.device = torch.device(device)
        self.n_freq_xyz, self.n_freq_t, self.time_dependent = int(n_freq_xyz), int(n_freq_t), bool(time_dependent)
        self.fx = embed_frequencies(self.n_freq_xyz).to(self.device) if self.n_freq_xyz > 0 else None
        self.ft = embed_frequencies(self.n_freq_t).to(self.device) if (self.time_dependent and self.n_freq_t > 0) else None
        # stash_f16: the five hidden activations of the stash as fp16 (3.2 instead of 5.7 KB per pixel-evaluation; only the
        # weight gradients see the rounding -- include/dvd_hip.h, dvd_mlp_desc)
        self.stash_f16 = bool(stash_f16)
        self.desc = _lib.MlpDesc(self.n_freq_xyz, self.n_freq_t, int(self.time_dependent),
                                 self.fx.data_ptr() if self.fx is not None else 0,
                                 self.ft.data_ptr() if self.ft is not None else 0, int(self.stash_f16), 0)
        self._monitor = None
        lib = _lib.load()
        self.c_in = lib.dvd_sf_mlp_in_channels(ctypes.byref(self.desc))
        self.packed = torch.empty(lib.dvd_sf_mlp_packed_bytes(ctypes.byref(self.desc)) // 4, device=self.device,
                                  dtype=torch.float32)

    def set_forward_monitor(self, monitor):
        """fp16 stash: the 1-element device tensor (slot [6] of a model's loss-scale state) that every stashing forward folds
        max |hidden activation| into -- an activation beyond fp16's range would be stashed as Inf and poison the weight
        gradients; the step's overflow guard (csrc/a16.hip dvd_gscale_end) then skips the step.  None: no guard."""
        self._monitor = monitor                      # (kept alive: the descriptor holds its address)
        self.desc.fwd_monitor = monitor.data_ptr() if monitor is not None else 0

    # -- sizes
    def stash_floats(self, n_pix):
        return _lib.load().dvd_sf_mlp_stash_bytes(ctypes.byref(self.desc), int(n_pix)) // 4

    def gstash_floats(self, n_pix):
        return _lib.load().dvd_sf_mlp_gstash_bytes(int(n_pix)) // 4

    def new_stash(self, n_pix):
        return torch.empty(self.stash_floats(n_pix), device=self.device, dtype=torch.float32)

    def new_gstash(self, n_pix):
        return torch.empty(self.gstash_floats(n_pix), device=self.device, dtype=torch.float32)

    # -- kernels
    def pack(self, weights, biases):
        """weights: six tensors [out,in,1,1] (or [out,in]); biases: six [out]."""
        ws = [_dev32(w.detach(), 'weight') for w in weights]
        bs = [_dev32(b.detach(), 'bias') for b in biases]
        if len(ws) != 6 or len(bs) != 6:
            raise RuntimeError('the fused MLP has exactly 6 conv layers (n_layers=4)')
        dims = [self.c_in] + [256] * 5
        for i, w in enumerate(ws):
            o = 256 if i < 5 else 3
            if w.numel() != o * dims[i] or bs[i].numel() != o:
                raise RuntimeError('layer %d: expected weight %dx%d, got %s' % (i, o, dims[i], tuple(w.shape)))
        W = _lib.PtrArr6(*[w.data_ptr() for w in ws])
        Bp = _lib.PtrArr6(*[b.data_ptr() for b in bs])
        lib = _lib.load()
        _lib.check(lib.dvd_sf_mlp_pack(ctypes.byref(self.desc), ctypes.byref(W), ctypes.byref(Bp), _p(self.packed),
                                       _stream()), 'dvd_sf_mlp_pack')

    def forward(self, p, t, t_offset=0.0, out_scale=1.0, sf_out=None, p_next=None, acc=None, stash=None):
        p = _dev32(p, 'p')
        B, C, H, W = p.shape
        if C != 3:
            raise RuntimeError('p must be [B,3,H,W]')
        if self.time_dependent:
            if t is None:
                raise ValueError('time dependent scene-flow field needs t')
            t = _dev32(t, 't')
            if t.numel() != B * H * W:
                raise RuntimeError('t must be [B,1,H,W]')
        n_pix = B * H * W
        if stash is not None and stash.numel() < self.stash_floats(n_pix):
            raise RuntimeError('stash too small')
        lib = _lib.load()
        _lib.check(lib.dvd_sf_mlp_fwd(ctypes.byref(self.desc), _p(self.packed), _p(p),
                                      _p(t) if self.time_dependent else ctypes.c_void_p(0), float(t_offset),
                                      float(out_scale), n_pix, H * W, _p(sf_out), _p(p_next), _p(acc), _p(stash),
                                      _stream()), 'dvd_sf_mlp_fwd')

    def backward_dx(self, stash, out_scale, g_out1, g_p, gstash, gW5, gb5, shape, gscale=1.0, scale_ptr=None,
                    g_out2=None, g_p_add=None):
        B, H, W = shape
        n_pix = B * H * W
        lib = _lib.load()
        _lib.check(lib.dvd_sf_mlp_bwd_dx(ctypes.byref(self.desc), _p(self.packed), _p(stash), float(out_scale),
                                         _p(g_out1), float(gscale), _p(scale_ptr), _p(g_out2), _p(g_p_add), n_pix,
                                         H * W, _p(g_p), _p(gstash), _p(gW5), _p(gb5), _stream()),
                   'dvd_sf_mlp_bwd_dx')

    def backward_dw(self, stash, gstash, n_pix, gW, gb):
        """Accumulates into gW[0..4] ([256,in]) and gb[0..4] ([256])."""
        W = _lib.PtrArr5(*[g.data_ptr() for g in gW])
        Bp = _lib.PtrArr5(*[g.data_ptr() for g in gb])
        lib = _lib.load()
        _lib.check(lib.dvd_sf_mlp_bwd_dw(ctypes.byref(self.desc), _p(stash), _p(gstash), int(n_pix), ctypes.byref(W),
                                         ctypes.byref(Bp), _stream()), 'dvd_sf_mlp_bwd_dw')


# ---------------------------------------------------------------------------------------
# elementwise helpers


def scale_add(out, a, scale=1.0, scale_ptr=None, b=None):
    """out = scale * (*scale_ptr) * a + b   (flat, any shape; out may alias a or b)."""
    lib = _lib.load()
    _lib.check(lib.dvd_scale_add(_p(out), _p(a), float(scale), _p(scale_ptr), _p(b), a.numel(), _stream()),
               'dvd_scale_add')
    return out


def mul_mask(out, a, mask):
    """out[b,c,h,w] = a[b,c,h,w] * mask[b,h,w]  (planar a, per-pixel mask; out may alias a)."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // (B * C)
    if mask.numel() != B * HW:
        raise RuntimeError('mul_mask: mask must have one value per pixel')
    lib = _lib.load()
    _lib.check(lib.dvd_mul_mask(_p(out), _p(_dev32(a, 'a')), _p(_dev32(mask, 'mask')), B, C, HW, _stream()), 'dvd_mul_mask')
    return out


def acc_reg(sf0, sf1, coef, g_sf1, abs_sum, accumulate=True):
    lib = _lib.load()
    ws = _workspace(lib.dvd_acc_reg_workspace_bytes(), sf0.device)
    _lib.check(lib.dvd_acc_reg(_p(sf0), _p(sf1), float(coef), _p(g_sf1), _p(ws), _p(abs_sum), int(accumulate),
                               sf0.numel(), _stream()), 'dvd_acc_reg')


# Bumped by every optimiser step: the fused Adam kernel rewrites parameters behind autograd's back
# (no `_version` change), and per-weight derived buffers (fragment-ordered conv weights) key on it.
WEIGHT_EPOCH = [0]


def adam_step(param, grad1, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps=1e-8, scale=1.0, scale_ptr=None,
              grad2=None, skip_ptr=None):
    """skip_ptr: device scalar; the update is skipped when it is non-zero (fp16 gradient overflow, csrc/a16.hip)."""
    WEIGHT_EPOCH[0] += 1
    lib = _lib.load()
    _lib.check(lib.dvd_adam_step_guarded(_p(param), _p(grad1), float(scale), _p(scale_ptr), _p(grad2), _p(exp_avg),
                                         _p(exp_avg_sq), param.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                         int(step), _p(skip_ptr), _stream()), 'dvd_adam_step')


def gscale_new(device, target_exponent=4.0):
    """The loss-scale state of the fp16 gradients (16 floats on the device; layout: include/dvd_hip.h, policy: csrc/a16.hip)."""
    st = torch.empty(16, device=device, dtype=torch.float32)
    _lib.check(_lib.load().dvd_gscale_init(_p(st), float(target_exponent), _stream()), 'dvd_gscale_init')
    return st


def gscale_step_begin(state):
    """First launch of a training step: clears the forward monitor (validation / warm-up / inference passes fold into it too)."""
    _lib.check(_lib.load().dvd_gscale_step_begin(_p(state), _stream()), 'dvd_gscale_step_begin')


def gscale_end(state):
    _lib.check(_lib.load().dvd_gscale_end(_p(state), _stream()), 'dvd_gscale_end')
