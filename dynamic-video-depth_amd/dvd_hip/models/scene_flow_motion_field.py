"""`--net scene_flow_motion_field` on MI355X: the per-video test-time-optimisation step.

Drop-in for /root/reference/models/scene_flow_motion_field.py:32-367 (class `Model`):
same `add_arguments` flag set (:33-67), constructor contract `Model(opt, loggers)`
(:78-138), attributes `_nets = [net_depth, net_sceneflow]`, `_optimizers`, `_metrics`,
`input_names`, `requires`, and `_train_on_batch(epoch, batch_ind, batch) -> batch_log`
with the keys of :226,321-323,195.

What is different is HOW a step runs.  The reference builds one autograd graph out of
~250 ATen launches per pair; here the step is three phases over device-resident data:

  1. depth nets forward, chunk by chunk (--depth_chunk images), every chunk WITH its autograd state kept in a slot of
     its own (a forward HIP graph and a backward HIP graph on one private memory pool) -> depth_1, depth_2; chunks that
     do not fit --depth_keep_gb take a no-graph forward and are recomputed in phase 3;
  2. everything downstream of the depths in hand-written HIP kernels, per chunk of
     pairs: unproject -> fused MLP (Euler steps, activations stashed) -> ONE fused
     warp+reprojection+loss forward/backward launch -> MLP backward (dX chain, dW) ->
     unproject backward; the acceleration regulariser shares the first MLP evaluation.  Gradients
     are produced UN-normalised; the batch-global 1/(sum(mask)+1e-8) is a device
     scalar applied at the very end, after the (data-parallel) all-reduce of the five
     loss sums, so there is no host synchronisation inside the step and N ranks
     reproduce the single-device result on the concatenated batch;
  3. (not in the warm-up phase) depth nets backward from the depth gradients of phase 2: a replay of each kept slot's
     backward graph (or a forward+backward recompute graph); then one RCCL all-reduce per net over its flat gradient
     buffer and one fused Adam launch per net.

Every convolution (the 7x7 stride-2 stem included, round 4), BatchNorm+ReLU, pooling, up-sampling and the MLP run on this
package's HIP kernels (csrc/); the matrix kernels evaluate fp32 products from two fp16 terms per operand (csrc/dvd_split.h).
`--act_fp16` (BASELINE configs[4]) stores the depth net's activations and their gradients as fp16 (csrc/a16.hip: the loss scale
lives on the device); `--use_cnn` swaps the MLP for the reference's U-Net scene-flow network (networks/FCNUnet.py) under autograd.
"""
import os
import sys
import warnings
from os import makedirs
from os.path import join

import numpy as np
import torch

from .. import configs, conv, flat, ops, parallel
from ..losses.scene_flow_projection import BackwardWarp, flow_by_depth, scene_flow_projection_slack, unproject_ptcld
from ..networks.FCNUnet import FCNUnet
from ..networks.sceneflow_field import SceneFlowFieldNet
from ..third_party.hourglass import HourglassModel_Embed
from ..third_party.MiDaS import MidasNet
from .netinterface import NetInterface

CAM_KEYS = ops.CAM_KEYS


def head_room_fraction(world, total_bytes=None):
    """Fraction of the device the memory planner leaves untouched: allocator fragmentation, and -- with several ranks --
    whatever RCCL's collectives allocate while the step runs (its communicator is created BEFORE the planner reads the free
    memory, parallel.init_from_env, so its channel / staging buffers are already counted as used).  Single process: 8 % (23 GB
    of 288); data parallel: 10 % (29 GB), derated automatically -- the benchmark configuration still keeps both of its slots
    (225 GB free - 60 GB slot >= 130 GB of stashes + 29 GB)."""
    frac = 0.08 if world <= 1 else 0.10
    # DVD_HEAD_ROOM_GB: an explicit head room in GB for runs that must leave more to other tenants of the device -- the 8-rank
    # run's RCCL buffers when the communicator is created late, or a memory-capped deployment.  Absolute: it is divided by the
    # device's REAL size (keep_slot_fits multiplies the fraction by that same total), not by a hard-coded 288 GB.
    gb = _head_room_gb_from_env()
    if gb:
        if total_bytes is None:
            total_bytes = torch.cuda.mem_get_info()[1] if torch.cuda.is_available() else 288 * 2 ** 30
        frac = max(frac, min(0.9, gb * 2 ** 30 / float(total_bytes)))
    return frac


def _head_room_gb_from_env():
    """DVD_HEAD_ROOM_GB, validated (also once at import: a malformed value must not first surface in the middle of step
    planning)."""
    raw = os.environ.get('DVD_HEAD_ROOM_GB')
    if not raw:
        return 0.0
    try:
        gb = float(raw)
    except ValueError:
        raise ValueError('DVD_HEAD_ROOM_GB=%r is not a number of gigabytes' % raw)
    if not (0.0 <= gb < 1e5):
        raise ValueError('DVD_HEAD_ROOM_GB=%r: expected a non-negative number of gigabytes' % raw)
    return gb


_head_room_gb_from_env()        # validate at import


def keep_slot_fits(est, free, total, reserve, spare, kept, budget, head_room=0.08):
    """May one more kept-activation slot of `est` bytes be captured?  `free` = HBM available to ordinary allocations,
    `reserve` = what phase 2 (the MLP stashes) will allocate, `spare` = room for the recompute graph of a chunk that is not
    kept (0 if this slot completes the step), `kept` = bytes already held by slots, `budget` = --depth_keep_gb.
    `head_room` (head_room_fraction) of the device stays free."""
    return kept + est <= budget and free - est >= reserve + head_room * total + spare


class Model(NetInterface):
    @classmethod
    def add_arguments(cls, parser):
        # the reference's flag set, verbatim names/defaults (scene_flow_motion_field.py:35-65)
        f, i, s = float, int, str
        for name, typ, default in (('l1_mul', f, 1e-4), ('disp_mul', f, 10), ('loss_type', s, 'l2'),
                                   ('scene_lr_mul', f, 1), ('n_down', i, 3), ('sf_min_mul', f, 0),
                                   ('sf_quantile', f, 0.5), ('static_mul', f, 1), ('flow_mul', f, 10),
                                   ('acc_mul', f, 100), ('si_mul', f, 0), ('cos_mul', f, 0), ('warm_mul', f, 1),
                                   ('interp_steps', i, 5), ('warm_sf', i, 0), ('n_freq_xyz', i, 16),
                                   ('n_freq_t', i, 16), ('sf_mag_div', f, 100)):
            parser.add_argument('--' + name, type=typ, default=default)
        for name in ('one_way', 'weight_steps', 'static', 'motion_seg_hard', 'warm_static', 'use_disp',
                     'use_disp_ratio', 'time_dependent', 'use_cnn', 'use_embedding', 'use_motion_seg', 'warm_reg',
                     'midas'):
            parser.add_argument('--' + name, action='store_true')
        # MI355X-specific knobs (new; defaults need no tuning)
        parser.add_argument('--mlp_stash_gb', type=float, default=48.0,
                            help='HBM budget for the scene-flow MLP activation stashes; sets pairs per chunk')
        parser.add_argument('--mlp_whole_batch_gb', type=float, default=160.0,
                            help='HBM ceiling for keeping the forward stashes of the whole batch alive so that the '
                                 'warp+loss kernel runs as ONE launch (falls back to one launch per chunk)')
        parser.add_argument('--mlp_recompute', type=int, default=1,
                            help='when the forward stashes of the whole batch do not fit --mlp_whole_batch_gb: 1 = one stash-free '
                                 'forward over all pairs, ONE warp+loss launch, then per chunk the stashed forward again + the '
                                 'merged backward (3 forward + 2 dX + 2 dW passes per pair at gap 1); 0 = the late-normaliser '
                                 'schedule of rounds 1-5 (one warp+loss launch per chunk, 3 + 3 + 3 passes)')
        parser.add_argument('--depth_chunk', type=int, default=48,
                            help='images per depth-net forward/backward chunk = per kept-activation graph slot (48: two slots per '
                                 '48-pair step).  0 = chosen at the first training step: the largest of 48 / 24 / 16 for which every '
                                 'slot is expected to fit beside the MLP stashes, else 16 (what bench.py runs with)')
        parser.add_argument('--depth_graphs', type=int, default=1,
                            help='1 (default): capture the depth net per chunk shape in HIP graphs (forward, forward+backward) '
                                 'and replay them: one launch per chunk instead of ~2 000, so the step does not depend on '
                                 'the host cores (2.8 s -> 2.0 s per step on a slow-host box, equal on a fast one; replay is '
                                 'bit-identical to eager execution, tests/test_30_full_step_gpu.py); falls back to eager '
                                 'execution if a capture fails.  0: eager launches')
        parser.add_argument('--depth_keep_gb', type=float, default=150.0,
                            help='HBM budget for keeping the depth net\'s autograd state of a step alive between its forward '
                                 '(phase 1) and its backward (phase 3), chunk by chunk, in HIP graphs: a kept chunk is not '
                                 'recomputed (MiDaS at 384x672: 1 GB per image, 95 GB for 48 pairs).  Chunks beyond the budget, '
                                 'or beyond what the MLP stashes leave free, take the no-graph forward + recompute path')
        parser.add_argument('--act_fp16', action='store_true',
                            help='BASELINE configs[4]: store the depth net\'s activations (and their gradients) as fp16 in HBM -- '
                                 'fp32 parameters, fp32 accumulation, fp32 loss sums; gradients carry a power-of-two loss scale kept on '
                                 'the device (csrc/a16.hip); a step whose fp16 gradients overflow skips its depth-net update.  MiDaS only')
        parser.add_argument('--max_act_overflow_skips', type=int, default=25,
                            help='--act_fp16: consecutive steps skipped because an fp16 ACTIVATION overflowed after which the run '
                                 'stops with an error (a warning from the third on): the loss scale cannot cure that')
        parser.add_argument('--mlp_stash_fp16', action='store_true',
                            help='store the hidden activations of the scene-flow MLP stash as fp16 (3.2 instead of 5.7 KB per pixel and '
                                 'Euler step; the MLP weight gradients then see fp16-rounded activations, losses and input gradients '
                                 'are unchanged).  Implied by --act_fp16')
        parser.add_argument('--grad_buckets', type=int, default=4,
                            help='data parallel: the depth-net gradient (421 MB for MiDaS) is all-reduced as this many large '
                                 'buckets in flight at once, the Adam launch of a bucket overlapping the reduction of the next')
        return parser, set()

    # ------------------------------------------------------------------------------------
    def __init__(self, opt, loggers=None):
        super().__init__(opt, loggers)
        self.input_names = ['img', 'img_1', 'img_2', 'pose', 'intrinsic', 'mask_1', 'mask_2', 'R_1', 'R_1_T', 'R_2',
                            'R_2_T', 't_1', 't_2', 'flow_1_2', 'flow_2_1', 'K', 'K_inv', 'motion_seg_1',
                            'time_stamp_1', 'time_stamp_2', 'frame_id_1', 'frame_id_2', 'time_step']
        self.gt_names = []
        self.requires = list(set().union(self.input_names, self.gt_names))
        if opt.midas:
            resize = [224, 384] if any(k in opt.dataset for k in ('real_video', 'korean', 'mctest', 'cube')) else None
            path = configs.midas_pretrain_path if os.path.exists(configs.midas_pretrain_path) else None
            if path is None:
                warnings.warn('MiDaS checkpoint %s not found: random weights' % configs.midas_pretrain_path)
            self.net_depth = MidasNet(path=path, non_negative=True, normalize_input=True, resize=resize)
        else:
            self.net_depth = HourglassModel_Embed(noexp=False, use_embedding=opt.use_embedding)
            if os.path.exists(configs.depth_pretrain_path):
                self.net_depth.net_depth.load_state_dict(torch.load(configs.depth_pretrain_path, map_location='cpu'))
            else:
                warnings.warn('depth checkpoint %s not found: random weights' % configs.depth_pretrain_path)
        if opt.use_cnn:      # the U-Net scene-flow network (:102-105), on the convolution kernels under autograd
            conv_setup = {'norm': 'none', 'activation': 'lrelu', 'pad_type': 'reflect', 'stride': 1}
            self.net_sceneflow = FCNUnet(conv_setup, n_down=getattr(opt, 'n_down', 3), feat=32, block_type='double_conv',
                                         in_channel=4 if opt.time_dependent else 3, out_channel=3)
        else:
            self.net_sceneflow = SceneFlowFieldNet(net_width=256, n_layers=4, time_dependent=opt.time_dependent,
                                                   N_freq_xyz=opt.n_freq_xyz, N_freq_t=opt.n_freq_t)
        self.bkwarp = BackwardWarp()
        self.unproject_points = unproject_ptcld()
        self.global_rank = getattr(opt, 'global_rank', 0)
        self._nets = [self.net_depth, self.net_sceneflow]
        self._metrics = ['flow_loss_1_2', 'loss', 'disp_loss_1_2', 'data_time', 'acc_reg', 'sf_loss']
        self.init_vars(add_path=False)
        self.init_weight(self.net_sceneflow, 'kaiming', 0.01, a=0.2)
        self.warp = scene_flow_projection_slack()
        self.depth_flow = flow_by_depth()
        self.visualizer = None
        self._flat_depth = self._flat_sf = None     # created by .to(device)
        self._optimizers = []
        self._depth_graphs = {}
        self._cnn_px_measured = 0.0     # --use_cnn: measured autograd bytes per pixel and U-Net evaluation
        self._graph_flops = {}          # id(CUDAGraph) -> algorithmic work per kernel class, counted at its capture
        self._keep_bytes = 0         # HBM held by kept-activation graph slots
        self._keep_per_px = 0.0      # measured bytes per image pixel of a captured slot
        self._auto_chunk = None      # --depth_chunk 0: images per slot, chosen at the first training step that keeps slots
        self._pool_bytes = 0         # HBM reserved by the private pools of all captured graphs
        self._keep_denied = {}       # slot key -> step at which it was last denied / trimmed (retried 16 steps later)
        self._step_no = 0
        self.warm = False

    # flat parameter buffers + fused Adam replace the two torch.optim.Adam objects (:113-115)
    def to(self, device):
        super().to(device)
        if self.device.type != 'cuda':
            raise RuntimeError('dvd_hip Model runs on a GPU only (the reference CPU path is the oracle)')
        betas = self.optim_params['betas']
        self._flat_depth = flat.FlatNet(self.net_depth, self.opt.lr, betas)
        self._flat_sf = flat.FlatNet(self.net_sceneflow, self.opt.lr * self.opt.scene_lr_mul, betas)
        self.optimizer_depth, self.optimizer_scene = self._flat_depth, self._flat_sf
        self._optimizers = [self._flat_depth, self._flat_sf]
        self._sf_grad_main = torch.zeros_like(self._flat_sf.grad)
        self._mlp = None if self.opt.use_cnn else self.net_sceneflow.kernels(
            self.device, stash_f16=bool(getattr(self.opt, 'mlp_stash_fp16', False) or getattr(self.opt, 'act_fp16', False)))
        self._gscale = None
        self._steps_skipped = 0
        if getattr(self.opt, 'act_fp16', False):
            if not self.opt.midas:
                raise NotImplementedError('--act_fp16 covers the MiDaS depth net (BASELINE configs[4])')
            self.net_depth.act_dtype = torch.float16
            self._gscale = ops.gscale_new(self.device)       # lives as long as the model: captured graphs hold its address
            # what each network's guarded Adam step subtracts from its step number (checkpoints carry the effective step)
            self._flat_depth.skip_count = self._gscale[5:6]
            self._flat_sf.skip_count = self._gscale[9:10]
            if self._mlp is not None:
                # the scene-flow MLP's fp16 stash shares the forward monitor: a hidden activation beyond fp16's range (stored
                # as Inf, contracted into the weight gradients) skips the step like a depth-net activation does
                self._mlp.set_forward_monitor(self._gscale[6:7])
        if self._pending_optimizer_state is not None:      # checkpoint restored before .to() (train.py:256,279)
            self._apply_optimizer_state(self._pending_optimizer_state)
            self._pending_optimizer_state = None
        if parallel.is_distributed():      # every rank starts from rank 0's weights (train.py:290-292)
            parallel.broadcast_(self._flat_depth.flat)
            parallel.broadcast_(self._flat_sf.flat)

    # ------------------------------------------------------------------------------------
    def _depth_forward(self, img, frame_ids):
        if self.opt.midas:
            return self.net_depth(img)
        return self.net_depth(img, frame_ids.long() if frame_ids is not None else None)

    # -- HIP graphs for the depth net -------------------------------------------------------
    # A MiDaS forward+backward of one chunk is ~2 000 kernel launches; 12 chunk passes per step make the
    # step launch-bound on hosts with slower cores (rocprofv3: 1.95 s of kernels in a 2.8 s step on one
    # box, 1.97 s wall on another).  With --depth_graphs 1 (default since the convolutions run on this
    # package's own kernels: round 1's replay through MIOpen was erratic)
    # each chunk shape is captured once (forward-only graph for phase 1, forward+backward graph for
    # phase 3, static input / output / output-gradient buffers; parameter gradients accumulate in place
    # into the flat gradient buffer) and replayed; anything that cannot be captured falls back to eager.
    def _graph_key(self, kind, img):
        return (kind, tuple(img.shape), bool(self.opt.midas))

    def _capture_depth_graph(self, kind, img, fid):
        """Returns (graph, static_in, static_out, static_gout) or None if capture is not possible."""
        key = self._graph_key(kind, img)
        if key in self._depth_graphs:
            return self._depth_graphs[key]
        entry = None
        try:
            import gc
            gc.collect()        # graphs of discarded models must not be destroyed while this capture is open
            static_in = img.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            grad_backup = self._flat_depth.grad.clone() if kind == 'fb' else None
            with torch.cuda.stream(side):                  # warm-up outside the capture (allocator, MIOpen handles)
                for _ in range(2):
                    if kind == 'f':
                        with torch.no_grad():
                            self._depth_forward(static_in, fid)
                    else:
                        self._flat_depth.detach_grads()
                        with torch.enable_grad():
                            d = self._depth_forward(static_in, fid)
                        d.backward(torch.zeros_like(d))
                        self._flat_depth.absorb_grads()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: calls made by other threads (the RCCL watchdog polling its events) do not invalidate
            # the capture; the step also keeps collectives out of flight while a graph is being captured
            mode = dict(capture_error_mode='thread_local')
            conv.PACK_PLAN.extend()        # the packings the warm-up passes asked for: persistent buffers, two launches per step
            ops.begin_capture()
            f0 = ops.flop_counters()       # (bench.py roofline_mfma: the graph's algorithmic work, added at every replay)
            if kind == 'f':
                with torch.no_grad(), torch.cuda.graph(graph, **mode):
                    static_out = self._depth_forward(static_in, fid)
                entry = (graph, static_in, static_out, None)
            else:
                static_g = torch.zeros(img.shape[0], 1, img.shape[2], img.shape[3], device=img.device)
                # parameter gradients: the engine hands over fresh tensors (no pre-attached .grad) and ONE multi-tensor add
                # per chunk, captured with the rest, folds them into the flat buffer -- ~620 tiny accumulate kernels
                # per replay otherwise
                self._flat_depth.detach_grads()
                with torch.cuda.graph(graph, **mode):
                    with torch.enable_grad():
                        static_out = self._depth_forward(static_in, fid)
                    static_out.backward(static_g)
                    self._flat_depth.absorb_grads()
                entry = (graph, static_in, static_out, static_g)
                self._flat_depth.grad.copy_(grad_backup)   # warm-up / capture passes used zero output gradients
            self._graph_flops[id(graph)] = ops.flops_since(f0)
            self._pool_bytes += self._pool_size(graph.pool(), img.device)
        except Exception as e:                             # noqa: BLE001 -- capture is an optimisation only
            warnings.warn('depth-net HIP graph capture failed (%s); running eagerly' % (str(e).splitlines()[0],))
            torch.cuda.synchronize()
            self._flat_depth.reattach_grads()
            entry = None
        self._depth_graphs[key] = entry
        return entry

    # -- kept activations -----------------------------------------------------------------------
    # With this package's kernels a MiDaS forward keeps ~1 GB of autograd state per 384x672 image (round 1, through
    # MIOpen/ATen: 3 GB), so the state of ALL chunks of a 48-pair step (95 GB) fits next to the MLP stashes: phase 1 runs
    # every chunk's forward WITH its graph state into a slot of its own (forward graph + backward graph on one private
    # memory pool), phase 3 replays the slot's backward graph -- the forward is computed once per step instead of twice.
    def _keep_slot(self, slot, chunk, fid, reserve_bytes, last_and_all_kept=False):
        key = ('keep', slot, tuple(chunk.shape), bool(self.opt.midas))
        if key in self._depth_graphs:
            entry = self._depth_graphs[key]
            # a slot that was denied (or trimmed) for lack of room is tried again every 16 steps: one transient
            # low-memory moment must not pin its chunk to the recompute path for the rest of the run
            if entry is not None or self._step_no - self._keep_denied.get(key, self._step_no) < 16:
                return entry
            del self._depth_graphs[key]
        # slots captured for another chunk shape at this position (the last, smaller batch of an epoch) hold HBM this
        # shape needs: release them
        for k in [k for k, v in self._depth_graphs.items() if k[0] == 'keep' and k[1] == slot and k != key and v is not None]:
            self._keep_bytes -= self._depth_graphs[k][5]
            self._pool_bytes -= self._depth_graphs[k][5]
            del self._depth_graphs[k]
        entry = None
        # bytes a slot will hold: measured on the slots captured so far (per image and pixel), a-priori figure (MiDaS with
        # fused epilogues: ~4.1 KB per pixel, ~2.2 KB with fp16 activations) for the first one, + packed weights
        n_px = chunk.shape[0] * chunk.shape[2] * chunk.shape[3]
        est = int(n_px * self._slot_bytes_per_px() + 1.5 * 2 ** 30)
        free, total = self._free_hbm(chunk.device)
        budget = float(getattr(self.opt, 'depth_keep_gb', 150.0)) * 2 ** 30
        # room that must stay free: the MLP stashes of phase 2 + 8 % head room + (unless this is the last slot of a step
        # whose other slots are all kept) the pool of the forward+backward recompute graph a non-kept chunk will need
        # (the recompute graph of a non-kept chunk frees its activations as its backward proceeds: its pool measures 0.26-0.31
        #  of a kept slot's -- 8.0 vs 25.6 GB for 16 hourglass images, 9.7 vs 37.9 GB for 16 MiDaS images at 768x1344 -- so half
        #  the slot's estimate is room enough; rounds 4-5 asked for all of it and kept one slot fewer)
        spare = 0 if last_and_all_kept else est // 2
        if os.environ.get('DVD_KEEP_DEBUG'):
            print('keep slot %d: est %.1f GB, free %.1f, reserve %.1f + %.1f + spare %.1f, kept so far %.1f, pools %.1f' % (
                slot, est / 2 ** 30, free / 2 ** 30, reserve_bytes / 2 ** 30,
                head_room_fraction(parallel.world_size(), total) * total / 2 ** 30, spare / 2 ** 30,
                self._keep_bytes / 2 ** 30, self._pool_bytes / 2 ** 30), file=sys.stderr, flush=True)
        hr = head_room_fraction(parallel.world_size(), total)
        if not keep_slot_fits(est, free, total, reserve_bytes, spare, self._keep_bytes, budget, hr):
            self._depth_graphs[key] = None
            self._keep_denied[key] = self._step_no
            return None
        try:
            import gc
            gc.collect()
            static_in = chunk.clone()
            grad_backup = self._flat_depth.grad.clone()
            if not any(k[0] == 'keep' and v is not None for k, v in self._depth_graphs.items()):
                side = torch.cuda.Stream()                 # warm-up outside the capture (allocator, handles), once
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._flat_depth.detach_grads()
                        with torch.enable_grad():
                            d = self._depth_forward(static_in, fid)
                        d.backward(torch.zeros_like(d))
                        self._flat_depth.absorb_grads()
                        del d
                torch.cuda.current_stream().wait_stream(side)
            mode = dict(capture_error_mode='thread_local')
            pool = torch.cuda.graph_pool_handle()
            g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            conv.PACK_PLAN.extend()
            ops.begin_capture()
            f0 = ops.flop_counters()
            with torch.cuda.graph(g_f, pool=pool, **mode):
                with torch.enable_grad():
                    static_out = self._depth_forward(static_in, fid)
            self._graph_flops[id(g_f)] = ops.flops_since(f0)
            static_g = torch.zeros(chunk.shape[0], 1, chunk.shape[2], chunk.shape[3], device=chunk.device)
            self._flat_depth.detach_grads()
            ops.begin_capture()         # its own generation: g_b's scalars are zero-filled by g_b's replay
            f0 = ops.flop_counters()
            with torch.cuda.graph(g_b, pool=pool, **mode):
                static_out.backward(static_g)
                self._flat_depth.absorb_grads()
            self._graph_flops[id(g_b)] = ops.flops_since(f0)
            self._flat_depth.grad.copy_(grad_backup)
            used = self._pool_size(pool, chunk.device)
            self._keep_bytes += used
            self._pool_bytes += used
            self._keep_per_px = max(self._keep_per_px, used / float(n_px))
            entry = (g_f, g_b, static_in, static_out, static_g, used)
        except Exception as e:                             # noqa: BLE001 -- an optimisation only
            warnings.warn('keeping the depth net\'s activations in HIP graphs failed (%s); recomputing' % (str(e).splitlines()[0],))
            torch.cuda.synchronize()
            self._flat_depth.reattach_grads()
            entry = None
        self._depth_graphs[key] = entry
        return entry

    @staticmethod
    def _pool_size(pool_id, device):
        """Bytes of the caching allocator's segments that belong to a graph's private pool (the reserved-bytes counter does
        not tell: a new pool may be carved from memory the process had reserved before)."""
        pid = tuple(pool_id)
        return sum(seg['total_size'] for seg in torch.cuda.memory_snapshot()
                   if tuple(seg.get('segment_pool_id', (0, 0))) == pid and seg.get('device', device.index) == device.index)

    def _free_hbm(self, device):
        """(bytes available to ordinary allocations: free on the device + cached by the allocator, total bytes).  The
        private pools of captured graphs are reserved but not `allocated` once the capture's temporaries are released, and
        they are NOT reusable: they are subtracted."""
        free, total = torch.cuda.mem_get_info(device)
        cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device) - self._pool_bytes
        return free + max(0, cached), total

    def _trim_keep_slots(self, device, need_bytes):
        """After phase 1: if the slots left less than phase 2 needs (a first slot larger than its a-priori estimate),
        give the newest slots back -- their chunks take the recompute path in phase 3, the step stays correct."""
        free, total = self._free_hbm(device)
        if os.environ.get('DVD_KEEP_DEBUG'):
            print('after phase 1: free %.1f GB, phase 2 needs %.1f, pools %.1f' % (free / 2 ** 30, need_bytes / 2 ** 30,
                                                                               self._pool_bytes / 2 ** 30), file=sys.stderr, flush=True)
        kept = [k for k, v in self._depth_graphs.items() if k[0] == 'keep' and v is not None]
        while kept and free < need_bytes + 0.5 * head_room_fraction(parallel.world_size(), total) * total:
            key = kept.pop()
            self._keep_bytes -= self._depth_graphs[key][5]
            self._pool_bytes -= self._depth_graphs[key][5]
            self._depth_graphs[key] = None
            self._keep_denied[key] = self._step_no
            import gc
            gc.collect()
            free, total = self._free_hbm(device)

    def _chunk(self):
        """Images per depth-net chunk: --depth_chunk, or (0 = auto) what _pick_depth_chunk chose at the first training step."""
        c = int(getattr(self.opt, 'depth_chunk', 48))
        return max(1, c if c > 0 else int(self._auto_chunk or 48))

    def _slot_bytes_per_px(self):
        """Autograd state a kept slot holds per image pixel: measured on the slots captured so far, else an a-priori figure
        (as measured in round 6: MiDaS with fused epilogues 4.8 KB -- 55.8 GB per 48 images at 384x672 --, 2.5 KB with fp16
        activations, the hourglass 6.7 KB; rounds 4-5 assumed 4.4 / 2.4 KB)."""
        apriori = (2500.0 if self._gscale is not None else 4900.0) if self.opt.midas else 6700.0
        return max(apriori, self._keep_per_px)

    def _pick_depth_chunk(self, B, HW, mlp_need, device):
        """--depth_chunk 0: the largest of 48 / 24 / 16 images per slot for which EVERY slot of the step is expected to fit
        beside phase 2's allocations (then nothing is recomputed, and larger launches are a little faster: 16 / 24 / 48
        measured 0.840 / 0.843 / 0.851 iters/s); if no size fits, the finest -- slots are kept one by one, so smaller slots
        keep more of the batch (hourglass at 384x672: one of two 48-image slots, or all six 16-image ones)."""
        free, total = self._free_hbm(device)
        budget = float(getattr(self.opt, 'depth_keep_gb', 150.0)) * 2 ** 30
        avail = min(budget, free - mlp_need - head_room_fraction(parallel.world_size(), total) * total)
        per_img = HW * self._slot_bytes_per_px()
        for c in (48, 24, 16):
            cc = min(c, B)
            if 2 * B * per_img + 2 * (-(-B // cc)) * 1.5 * 2 ** 30 <= avail:
                return cc
        return min(16, B)

    def _phase2_bytes(self, B, HW, steps, do_reg):
        """What phase 2 will allocate (the room phase 1 leaves free): the MLP stashes of the whole batch if they fit
        --mlp_whole_batch_gb, else those of ONE chunk (late-normaliser / recompute schedules).  (Rounds 1-5 reserved
        min(whole batch, the ceiling) in the chunked case too: 160 GB for 48 GB of stashes at BASELINE configs[4]'s 64 pairs,
        and not one depth-net slot was kept there.)"""
        Bc0 = self._pairs_per_chunk(B, HW, steps, do_reg)
        if self._mlp is None:        # --use_cnn: autograd state of the U-Net, per pixel and evaluation (measured)
            return B * HW * self._cnn_bytes_per_px() * (steps + (1 if do_reg else 0)) + 24 * B * HW * 4
        stash, gstash = self._mlp.stash_floats(HW) * 4, self._mlp.gstash_floats(HW) * 4
        if Bc0 >= B or self._whole_batch_fits(B, Bc0, HW, steps, do_reg):
            need = B * steps * stash + min(Bc0, B) * (gstash + (stash if (do_reg and steps == 1) else 0))
        else:
            need = Bc0 * (stash * max(steps, 2 if do_reg else 1) + gstash)
        return need + 24 * B * HW * 4

    def _depths_keep(self, img, frame_ids, slot0, reserve_bytes, n_slots_total):
        """Depth maps of phase 1 with the autograd state of as many chunks as fit kept for phase 3."""
        out = []
        c = self._chunk()
        for ci, b0 in enumerate(range(0, img.shape[0], c)):
            fid = frame_ids[b0:b0 + c] if frame_ids is not None else None
            chunk = img[b0:b0 + c]
            others_kept = all(v is not None for k, v in self._depth_graphs.items() if k[0] == 'keep') and \
                sum(1 for k in self._depth_graphs if k[0] == 'keep') == n_slots_total - 1
            e = self._keep_slot(slot0 + ci, chunk, fid, reserve_bytes, others_kept) if self._use_graphs(chunk, fid) else None
            if e is not None:
                e[2].copy_(chunk)
                conv.PACK_PLAN.ensure_current()      # (packed weights follow the optimiser: two launches after a step)
                e[0].replay()
                ops.note_replay(self._graph_flops.get(id(e[0])))
                out.append(e[3].detach().clone())
            else:
                with ops.counting_recomputed():      # phase 3 runs this chunk's forward again (forward+backward graph)
                    out.append(self._depths_nograd(chunk, fid))
        return torch.cat(out, 0).contiguous()

    def _use_graphs(self, img, frame_ids):
        return bool(getattr(self.opt, 'depth_graphs', 1)) and (frame_ids is None or not self.opt.use_embedding)

    def _depths_nograd(self, img, frame_ids):
        out = []
        c = self._chunk()
        with torch.no_grad():
            for b0 in range(0, img.shape[0], c):
                fid = frame_ids[b0:b0 + c] if frame_ids is not None else None
                chunk = img[b0:b0 + c]
                g = self._capture_depth_graph('f', chunk, fid) if self._use_graphs(chunk, fid) else None
                if g is not None:
                    g[1].copy_(chunk)
                    conv.PACK_PLAN.ensure_current()      # (packed weights follow the optimiser: two launches after a step)
                    g[0].replay()
                    ops.note_replay(self._graph_flops.get(id(g[0])))
                    out.append(g[2].clone())
                else:
                    out.append(self._depth_forward(chunk, fid))
        return torch.cat(out, 0).contiguous()

    def _depth_backward(self, img, frame_ids, g_depth, slot0=None):
        c = self._chunk()
        for ci, b0 in enumerate(range(0, img.shape[0], c)):
            fid = frame_ids[b0:b0 + c] if frame_ids is not None else None
            chunk = img[b0:b0 + c]
            kept = None if slot0 is None else self._depth_graphs.get(('keep', slot0 + ci, tuple(chunk.shape), bool(self.opt.midas)))
            if kept is not None:                        # the forward of phase 1 left this chunk's graph state in its slot
                kept[4].copy_(g_depth[b0:b0 + c])
                conv.PACK_PLAN.ensure_current()      # (packed weights follow the optimiser: two launches after a step)
                kept[1].replay()
                ops.note_replay(self._graph_flops.get(id(kept[1])))
                continue
            g = self._capture_depth_graph('fb', chunk, fid) if self._use_graphs(chunk, fid) else None
            if g is not None:
                g[1].copy_(chunk)
                g[3].copy_(g_depth[b0:b0 + c])
                conv.PACK_PLAN.ensure_current()      # (packed weights follow the optimiser: two launches after a step)
                g[0].replay()
                ops.note_replay(self._graph_flops.get(id(g[0])))
                continue
            self._flat_depth.detach_grads()          # one multi-tensor accumulation per chunk instead of ~620 adds
            with torch.enable_grad():
                d = self._depth_forward(chunk, fid)
            d.backward(g_depth[b0:b0 + c])
            self._flat_depth.absorb_grads()

    def _integer_steps(self, batch_or_input):
        """Euler steps of this batch = round(mean(ts2 - ts1) / time_step) (:248-250), recomputed every step:
        consecutive batches mix frame gaps and allocators reuse addresses, so nothing about the tensors'
        identity says the gap is unchanged (one tiny reduction + read-back per 2 s step)."""
        ts1, ts2 = batch_or_input['time_stamp_1'], batch_or_input['time_stamp_2']
        step = batch_or_input['time_step']
        time_step = float(step.squeeze().item()) if torch.is_tensor(step) else float(step)
        gap = torch.mean(ts2.float() - ts1.float())
        return int((gap / time_step).round().long().item()), time_step

    def _cnn_bytes_per_px(self):
        """Autograd state of ONE U-Net evaluation per pixel (--use_cnn): the measured value once a step has run; before that
        an a-priori figure that grows with the depth of the U-Net -- 2.6 KB per pixel was measured for n_down = 3 on the
        CPU path; the GPU path's Conv2dBlocks also keep a reflect-padded copy and a sliced .contiguous() copy of their
        inputs, hence the factor 1.5."""
        if self._cnn_px_measured > 0:
            return self._cnn_px_measured
        return 2600.0 * 1.5 * (1.0 + 0.15 * max(0, int(getattr(self.opt, 'n_down', 3)) - 3))

    def _pairs_per_chunk(self, B, HW, steps, with_reg):
        if self._mlp is None:            # --use_cnn: the whole batch goes through the U-Net at once
            return B
        per_pair = self._mlp.stash_floats(HW) * 4 * max(steps, 2 if with_reg else 1) + self._mlp.gstash_floats(HW) * 4
        return int(max(1, min(B, (self.opt.mlp_stash_gb * 2 ** 30) // per_pair)))

    def _whole_batch_fits(self, B, Bc, HW, steps, with_reg):
        """Forward stashes of all B pairs + the backward / regulariser scratch of one chunk."""
        if self._mlp is None:
            return True
        stash, gstash = self._mlp.stash_floats(HW) * 4, self._mlp.gstash_floats(HW) * 4
        # (merged backward: the regulariser's second evaluation needs a stash of its own only at gap 1; from gap 2 on it is
        #  Euler evaluation 1)
        need = B * steps * stash + Bc * (gstash + (stash if (with_reg and steps == 1) else 0))
        return need <= float(getattr(self.opt, 'mlp_whole_batch_gb', 160.0)) * 2 ** 30

    # ------------------------------------------------------------------------------------
    def _train_on_batch(self, epoch, batch_ind, batch):
        opt = self.opt
        self._step_no += 1
        conv.set_grad_scale_state(self._gscale)      # the loss-scale state of THIS model's fp16 gradients (None: fp32 storage)
        if self._gscale is not None:
            # the forward monitor counts for THIS step's forward passes only: warm-up epochs, validation and inference fold
            # into it too and never reach dvd_gscale_end -- a stale overflow of theirs skipped the next real step
            ops.gscale_step_begin(self._gscale)
        self.warm = warm = epoch <= opt.warm_sf
        self.net_depth.eval()                        # BN statistics are never updated (:157,168 / hourglass.py:200-208)
        for p in self.net_depth.parameters():
            p.requires_grad = not warm
        self._flat_sf.zero_grad()
        self._sf_grad_main.zero_()
        if not warm:
            self._flat_depth.zero_grad()
        for k, v in batch.items():                   # strip the DataLoader dimension (:177-179)
            if type(v) != list:
                batch[k] = v.squeeze(0)
        steps, time_step = self._integer_steps(batch)
        self.steps = steps
        self.load_batch(batch)
        inp = self._input
        B, _, H, W = inp.img_1.shape
        HW, dev = H * W, self.device
        fid1 = inp.frame_id_1 if not opt.midas else None
        fid2 = inp.frame_id_2 if not opt.midas else None

        # ---- phase 1: depth maps; the autograd state of as many chunks as fit stays alive for phase 3 (kept slots),
        #      the rest is a no-graph forward that phase 3 recomputes
        do_reg = opt.interp_steps > 0 and (not warm or opt.warm_reg) and opt.acc_mul > 0
        keeping = not (warm or not getattr(opt, 'depth_graphs', 1) or float(getattr(opt, 'depth_keep_gb', 150.0)) <= 0)
        if keeping and int(opt.depth_chunk) <= 0 and self._auto_chunk is None:
            # decided ONCE, at the first training step that keeps slots (graphs and slots are per chunk shape)
            self._auto_chunk = self._pick_depth_chunk(B, HW, self._phase2_bytes(B, HW, steps, do_reg), dev)
        n_slots = -(-B // self._chunk())
        if not keeping:
            import contextlib
            # (non-warm steps without kept slots: phase 3 recomputes every chunk's forward; a warm-up step has no depth-net backward)
            with (contextlib.nullcontext() if warm else ops.counting_recomputed()):
                depth_1 = self._depths_nograd(inp.img_1, fid1)
                depth_2 = self._depths_nograd(inp.img_2, fid2)
        else:
            mlp_need = self._phase2_bytes(B, HW, steps, do_reg)
            depth_1 = self._depths_keep(inp.img_1, fid1, 0, mlp_need, 2 * n_slots)
            depth_2 = self._depths_keep(inp.img_2, fid2, n_slots, mlp_need, 2 * n_slots)
            self._trim_keep_slots(dev, mlp_need)

        # ---- phase 2: geometry + scene-flow MLP + losses, forward and backward, in HIP
        mul = steps if opt.weight_steps else 1
        disp_mode = 1 if opt.use_disp else (2 if opt.use_disp_ratio else 0)
        sums = torch.zeros(8, device=dev)            # [S0..S3, sum|sf1-sf0|, 0, 0, 0]
        g_d1_main = torch.empty_like(depth_1)
        g_d2_main = torch.empty_like(depth_2)
        g_d1_reg = None                              # allocated only by the per-chunk fallback path
        mlp, k = self._mlp, self._flat_sf
        if opt.use_cnn:
            return self._finish_step_cnn(epoch, batch_ind, batch, inp, depth_1, depth_2, steps, time_step, warm, do_reg, n_slots,
                                         fid1, fid2, mul, disp_mode, sums, g_d1_main, g_d2_main)
        mlp.pack([p for p in self.net_sceneflow.parameter_list()[0::2]], self.net_sceneflow.parameter_list()[1::2])
        gW_main = [k.view(self._sf_grad_main, 2 * i) for i in range(6)]
        gb_main = [k.view(self._sf_grad_main, 2 * i + 1) for i in range(6)]
        gW_reg = [k.view(k.grad, 2 * i) for i in range(6)]
        gb_reg = [k.view(k.grad, 2 * i + 1) for i in range(6)]
        inv_div = 1.0 / opt.sf_mag_div
        Bc = self._pairs_per_chunk(B, HW, steps, do_reg)
        mask_2 = inp.mask_2.reshape(B, H, W)
        # --use_motion_seg (:253-254): the integrated scene flow is multiplied by motion_seg_1 before the
        # warp; the un-masked field still drives the Euler chain and the regulariser
        mseg = inp.motion_seg_1.reshape(B, H, W) if opt.use_motion_seg else None
        cfg_all = ops.warp_cfg(B, H, W, midas_mask=opt.midas, crit_l2=warm, disp_mode=disp_mode,
                               loss_on_sf=not opt.use_disp, flow_mul=opt.flow_mul * mul, disp_mul=opt.disp_mul * mul)
        # The MLP activation stashes bound how many pairs go through the MLP kernels at once
        # (Bc).  If the forward stashes of the WHOLE batch fit the budget next to one chunk of
        # backward scratch, the warp+loss kernel runs once over all pairs (one launch of
        # B*H*W pixels fills the chip far better than B/Bc smaller ones); otherwise every
        # chunk gets its own warp+loss launch.
        whole = Bc < B and self._whole_batch_fits(B, Bc, HW, steps, do_reg)
        # ranks may hold different batch sizes / frame gaps: agree on the schedule (early or late normaliser)
        # and on the size of the global batch before the first data-dependent collective
        # (also agreed across ranks: does ANY rank still have to capture a depth-net graph in phase 3 -- a chunk that is not
        #  kept and has no recompute graph yet?  Then every rank keeps the MLP-gradient all-reduce out of flight until after
        #  phase 3, so the order of collectives is the same everywhere)
        # over the REAL chunk list of both image sets (a ragged last chunk has its own slot / graph keys)
        may_capture = False
        if not warm and getattr(opt, 'depth_graphs', 1):
            cw = self._chunk()
            for s0, img in ((0, inp.img_1), (n_slots, inp.img_2)):
                for ci, b0 in enumerate(range(0, B, cw)):
                    chunk = img[b0:b0 + cw]
                    kept = self._depth_graphs.get(('keep', s0 + ci, tuple(chunk.shape), bool(opt.midas)))
                    if kept is None and self._graph_key('fb', chunk) not in self._depth_graphs:
                        may_capture = True
        # Round 6: when the forward stashes of the whole batch do NOT fit (BASELINE configs[4]: 64 pairs at 768 x 1344 = 211 GB of
        # stashes), the late-normaliser schedule below costs 3 forward + 3 dX + 3 dW passes per pair and Euler step -- the
        # regulariser cannot share the main path's backward while the normaliser is unknown.  The RECOMPUTE schedule runs the
        # Euler chain once WITHOUT stashes for every pair (the cheapest form of the forward kernel), one warp+loss launch over
        # the whole batch, and then per chunk the stashed forward again + the merged backward: 3 forward + 2 dX + 2 dW at
        # gap 1 (2k + k + k instead of (k + 2) x 3 at gap k), same arithmetic as the whole-batch schedule (the recomputed
        # evaluations are bit-identical: the forward kernel is deterministic).  --mlp_recompute 0 restores the late schedule.
        recompute = Bc < B and not whole and bool(int(getattr(opt, 'mlp_recompute', 1)))
        late, n_global, capturing = parallel.agree_on_step_plan(dev, not (whole or Bc >= B or recompute), B, may_capture)
        early_norm = not late
        reg_coef = opt.acc_mul / (3.0 * n_global * HW + 1e-6)
        chunks = [(b0, min(B, b0 + Bc)) for b0 in range(0, B, Bc)]
        P1_all = ops.unproject(depth_1, inp.R_1, inp.t_1, inp.K_inv, planar=True)
        sf_all = torch.zeros(B, 3, H, W, device=dev)
        g_sf_all = torch.empty_like(sf_all)

        def cams_of(b0, b1):
            return {kk: getattr(inp, kk)[b0:b1] for kk in CAM_KEYS}

        def mlp_forward_chunk(b0, b1, keep_first=None, acc=None, with_stash=True):
            """Euler integration of the scene flow over `steps` frames (:360-367).  keep_first: also
            return sf_0 and q = P1 + sf_0 of the first evaluation (the regulariser's sf_0, see below).
            acc: the tensor the integrated flow is accumulated into (default: this chunk of sf_all); with_stash=False: no
            activation stashes (the first pass of the recompute schedule)."""
            ts = inp.time_stamp_1[b0:b1] if opt.time_dependent else None
            n_pix = (b1 - b0) * HW
            stashes, p_cur, first = [], P1_all[b0:b1], None
            acc = sf_all[b0:b1] if acc is None else acc
            for i in range(steps):
                st = mlp.new_stash(n_pix) if with_stash else None
                want_next = i + 1 < steps or (keep_first and i == 0)
                p_next = torch.empty_like(p_cur) if want_next else None
                # the regulariser's two evaluations ARE Euler evaluations 0 and 1 (see merged_backward_chunk): keep their
                # outputs (evaluation 0's only when it is not the accumulated flow itself)
                sf_i = torch.empty_like(p_cur) if (keep_first and i < 2 and steps > 1) else None
                mlp.forward(p_cur, ts, t_offset=i * time_step, out_scale=inv_div, sf_out=sf_i, p_next=p_next,
                            acc=acc, stash=st)
                if keep_first and i == 0:
                    first = [sf_i, p_next, None]    # sf_i is None when steps == 1: sf_0 is the accumulated flow itself
                if keep_first and i == 1:
                    first[2] = sf_i                 # sf_1 = MLP(P1 + sf_0, t_1 + dt)
                stashes.append(st)
                p_cur = p_next
            return (stashes, first) if keep_first is not None else stashes

        def warp(b0, b1):
            sf_used = sf_all[b0:b1]
            if mseg is not None:
                sf_used = ops.mul_mask(torch.empty_like(sf_used), sf_used, mseg[b0:b1])
            cfg = cfg_all if (b0, b1) == (0, B) else ops.warp_cfg(
                b1 - b0, H, W, midas_mask=opt.midas, crit_l2=warm, disp_mode=disp_mode, loss_on_sf=not opt.use_disp,
                flow_mul=opt.flow_mul * mul, disp_mul=opt.disp_mul * mul)
            csum = torch.empty(4, device=dev)
            ops.warp_loss_fused(cfg, depth_1[b0:b1], depth_2[b0:b1], inp.flow_1_2[b0:b1], mask_2[b0:b1],
                                sf_used, cams_of(b0, b1),
                                out=(csum, g_d1_main[b0:b1], g_d2_main[b0:b1], g_sf_all[b0:b1]))
            if mseg is not None:
                ops.mul_mask(g_sf_all[b0:b1], g_sf_all[b0:b1], mseg[b0:b1])
            sums[:4] += csum

        def mlp_backward_chunk(b0, b1, stashes, gst):
            """Backward through the Euler chain: g_p_i = g_p_{i+1} + J_i^T (g_acc + g_p_{i+1})."""
            nb, n_pix = b1 - b0, (b1 - b0) * HW
            cams = cams_of(b0, b1)
            g_p = None
            for i in reversed(range(steps)):
                g_new = torch.empty_like(P1_all[b0:b1])
                mlp.backward_dx(stashes[i], inv_div, g_sf_all[b0:b1], g_new, gst, gW_main[5], gb_main[5], (nb, H, W),
                                g_out2=g_p, g_p_add=g_p)
                mlp.backward_dw(stashes[i], gst, n_pix, gW_main[:5], gb_main[:5])
                g_p = g_new
            ops.unproject_backward(g_p, True, cams['R_1'], cams['K_inv'], out=g_d1_main[b0:b1], accumulate=True)

        def reg_chunk(b0, b1, gst):
            """Acceleration regulariser (:326-344), P1 un-detached."""
            nb, n_pix = b1 - b0, (b1 - b0) * HW
            cams = cams_of(b0, b1)
            ts = inp.time_stamp_1[b0:b1] if opt.time_dependent else None
            P1 = P1_all[b0:b1]
            sa, sb = mlp.new_stash(n_pix), mlp.new_stash(n_pix)
            sf0, sf1, q = torch.empty_like(P1), torch.empty_like(P1), torch.empty_like(P1)
            mlp.forward(P1, ts, 0.0, inv_div, sf_out=sf0, p_next=q, stash=sa)
            mlp.forward(q, ts, time_step, inv_div, sf_out=sf1, stash=sb)
            g1 = torch.empty_like(P1)
            ops.acc_reg(sf0, sf1, reg_coef, g1, sums[4:5], accumulate=True)
            g_q = torch.empty_like(P1)
            mlp.backward_dx(sb, inv_div, g1, g_q, gst, gW_reg[5], gb_reg[5], (nb, H, W))
            mlp.backward_dw(sb, gst, n_pix, gW_reg[:5], gb_reg[:5])
            g_P = torch.empty_like(P1)
            mlp.backward_dx(sa, inv_div, g1, g_P, gst, gW_reg[5], gb_reg[5], (nb, H, W), gscale=-1.0, g_out2=g_q,
                            g_p_add=g_q)
            mlp.backward_dw(sa, gst, n_pix, gW_reg[:5], gb_reg[:5])
            ops.unproject_backward(g_P, True, cams['R_1'], cams['K_inv'], out=g_d1_reg[b0:b1])

        def merged_backward_chunk(b0, b1, stashes, first, gst, inv):
            """Main loss and acceleration regulariser through ONE backward of the first evaluation.

            The regulariser's sf_0 = MLP(P1, t_1) (:329-333) is the first Euler evaluation of the main
            path (:360-367: same points, same time, same weights), so its forward, stash, dX and dW
            passes are shared: with G = inv * g_sf (main, late normaliser already known) the first
            evaluation receives  G + g_p1 - g1 + g_q  where g1 = dR/dsf_1, g_q = J_b^T g1, and
            g_P1 = g_p1 + g_q + J_0^T(...).  One MLP evaluation per step less than the reference.
            From gap 2 on the regulariser's sf_1 = MLP(P1 + sf_0, t_1 + dt) (:335-338) is Euler evaluation 1 as well
            (:360-367), so that one is shared too (round 4): evaluation 1 receives G + g_p2 + g1, and a gap-k step costs k
            evaluations instead of the reference's k + 2."""
            nb, n_pix = b1 - b0, (b1 - b0) * HW
            cams = cams_of(b0, b1)
            ts = inp.time_stamp_1[b0:b1] if opt.time_dependent else None
            P1, g_sf = P1_all[b0:b1], g_sf_all[b0:b1]
            g_q = g1 = None
            shared1 = do_reg and steps > 1      # the regulariser's second evaluation is Euler evaluation 1 as well (:335-338 = :360-367)
            if do_reg:
                sf0 = first[0] if first[0] is not None else sf_all[b0:b1]
                g1 = torch.empty_like(P1)
                if shared1:
                    ops.acc_reg(sf0, first[2], reg_coef, g1, sums[4:5], accumulate=True)
                else:
                    sb, sf1 = mlp.new_stash(n_pix), torch.empty_like(P1)
                    mlp.forward(first[1], ts, time_step, inv_div, sf_out=sf1, stash=sb)
                    ops.acc_reg(sf0, sf1, reg_coef, g1, sums[4:5], accumulate=True)
                    g_q = torch.empty_like(P1)
                    mlp.backward_dx(sb, inv_div, g1, g_q, gst, gW_reg[5], gb_reg[5], (nb, H, W))
                    mlp.backward_dw(sb, gst, n_pix, gW_reg[:5], gb_reg[:5])
                    del sb, sf1
            g_p = None
            for i in reversed(range(1, steps)):
                g_new = torch.empty_like(P1)
                g_o2 = g_p
                if shared1 and i == 1:          # evaluation 1's output also feeds the regulariser: + dR/dsf_1 (not normalised)
                    g_o2 = g1 if g_p is None else ops.scale_add(torch.empty_like(P1), g1, b=g_p)
                mlp.backward_dx(stashes[i], inv_div, g_sf, g_new, gst, gW_reg[5], gb_reg[5], (nb, H, W), scale_ptr=inv,
                                g_out2=g_o2, g_p_add=g_p)
                mlp.backward_dw(stashes[i], gst, n_pix, gW_reg[:5], gb_reg[:5])
                g_p = g_new
            extra, add = g_p, g_p                    # gradient reaching sf_0 besides G, and reaching P1 directly
            if shared1:                              # g_p already holds everything that reaches q = P1 + sf_0
                extra = ops.scale_add(torch.empty_like(P1), g1, scale=-1.0, b=g_p)          # g_p - g1
            elif do_reg:
                extra = ops.scale_add(torch.empty_like(P1), g1, scale=-1.0, b=g_q)          # -g1 + g_q
                add = g_q
            g_P = torch.empty_like(P1)
            mlp.backward_dx(stashes[0], inv_div, g_sf, g_P, gst, gW_reg[5], gb_reg[5], (nb, H, W), scale_ptr=inv,
                            g_out2=extra, g_p_add=add)
            mlp.backward_dw(stashes[0], gst, n_pix, gW_reg[:5], gb_reg[:5])
            ops.unproject_backward(g_P, True, cams['R_1'], cams['K_inv'], out=g_d1_main[b0:b1], accumulate=True)

        gst = mlp.new_gstash(min(Bc, B) * HW)
        if early_norm and recompute and Bc < B and not whole:
            with ops.counting_recomputed():          # the stashed evaluations below are the ones the backward needs
                for b0, b1 in chunks:
                    mlp_forward_chunk(b0, b1, with_stash=False)
            warp(0, B)
            parallel.all_reduce_sum_(sums[:4])
            scalars = ops.loss_finalize(cfg_all, sums)
            inv = scalars[0:1]
            ops.scale_add(g_d1_main, g_d1_main, scale_ptr=inv)
            for b0, b1 in chunks:
                # the same evaluations again, stashed; their flow goes to a scratch accumulator (sf_all already holds it)
                st, first = mlp_forward_chunk(b0, b1, keep_first=bool(do_reg), acc=torch.zeros_like(sf_all[b0:b1]))
                merged_backward_chunk(b0, b1, st, first, gst, inv)
                del st[:], first
            parallel.all_reduce_sum_(sums[4:])
        elif early_norm:
            kept = [mlp_forward_chunk(b0, b1, keep_first=bool(do_reg)) for b0, b1 in chunks]
            warp(0, B)
            # the batch-global normaliser is known before the MLP backward: all-reduce the four loss sums
            # now and hand 1/(S0+1e-8) to the backward kernels as a device scalar
            parallel.all_reduce_sum_(sums[:4])
            scalars = ops.loss_finalize(cfg_all, sums)
            inv = scalars[0:1]
            ops.scale_add(g_d1_main, g_d1_main, scale_ptr=inv)
            for (b0, b1), (st, first) in zip(chunks, kept):
                merged_backward_chunk(b0, b1, st, first, gst, inv)
                del st[:]
            del kept
            parallel.all_reduce_sum_(sums[4:])
        else:
            g_d1_reg = torch.zeros_like(depth_1) if do_reg else None
            for b0, b1 in chunks:
                st = mlp_forward_chunk(b0, b1)
                warp(b0, b1)
                mlp_backward_chunk(b0, b1, st, gst)
                del st
                if do_reg:
                    reg_chunk(b0, b1, gst)
        del gst
        if not early_norm:
            # per-chunk launches: the batch-global normaliser only exists now.  All-reduce the loss sums
            # (SURVEY.md section 8e), then scene-flow MLP gradient = inv * main + reg
            parallel.all_reduce_sum_(sums)
            scalars = ops.loss_finalize(cfg_all, sums)
            inv = scalars[0:1]
            ops.scale_add(k.grad, self._sf_grad_main, scale_ptr=inv, b=k.grad)
        # the MLP gradient all-reduce overlaps the depth-net backward, except in a step that still has to
        # capture the depth net's forward+backward graph (no collective in flight during a capture)
        h_sf = None if capturing else k.all_reduce_grads(async_op=True)

        # ---- phase 3: depth-net backward from the depth gradients
        if not warm:
            if early_norm:
                g_d1 = g_d1_main                      # already normalised, regulariser included
            else:
                g_d1 = ops.scale_add(g_d1_main, g_d1_main, scale_ptr=inv, b=g_d1_reg)
            g_d2 = ops.scale_add(g_d2_main, g_d2_main, scale_ptr=inv)
            self._depth_backward(inp.img_1, fid1, g_d1, slot0=0)
            self._depth_backward(inp.img_2, fid2, g_d2, slot0=n_slots)
            # 421 MB (MiDaS) in --grad_buckets large all-reduces, each bucket's Adam launch overlapping the next
            # bucket's reduction
            skip = None
            if self._gscale is not None:
                # fp16 gradients: did this step stay inside fp16 range?  (device-side decision; with several ranks an
                # overflow anywhere skips the update everywhere)
                if parallel.is_distributed():          # both monitors: gradients [3], forward activations [6]
                    both = torch.stack([self._gscale[3], self._gscale[6]])
                    torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.MAX)
                    self._gscale[3:4].copy_(both[0:1])
                    self._gscale[6:7].copy_(both[1:2])
                ops.gscale_end(self._gscale)
                skip = self._gscale[4:5]
            self._flat_depth.all_reduce_and_adam_step(getattr(opt, 'grad_buckets', 4), skip_ptr=skip)
        if capturing:
            k.all_reduce_grads()
        if h_sf is not None:
            h_sf.wait()
        # (fp16 activations: an ACTIVATION that overflowed fp16 makes the depth map, and with it the scene-flow network's
        #  gradient, non-finite: that step skips BOTH updates.  A mere loss-scale overflow of the depth net's fp16 GRADIENTS
        #  leaves the depth maps finite and the scene-flow network's fp32 gradients valid: its update goes through -- the
        #  (flag, count) pair [8], [9] of the loss-scale state counts activation overflows only)
        k.adam_step(skip_ptr=None if (warm or self._gscale is None) else self._gscale[8:10])

        host = self._step_scalars_to_host(scalars, sums, warm)    # the only host synchronisation of the step
        acc_reg = opt.acc_mul * host[8] / (3.0 * n_global * HW + 1e-6) if do_reg else 0
        # --weight_steps scales the gradient only: the logged loss is the unweighted one (`**loss_data`
        # overwrites 'loss' at :226)
        batch_log = {'size': opt.batch_size, 'loss': host[1] / mul, 'total_loss': host[1] / mul, 'flow_loss_1_2': host[2],
                     'disp_loss_1_2': host[3], 'sf_loss': host[4], 'acc_reg': acc_reg}
        if self._gscale is not None:
            batch_log['steps_skipped'] = self._steps_skipped
        self._last = {'depth_1': depth_1, 'depth_2': depth_2, 'mask_sum': host[5], 'sf_all': sf_all, 'mseg': mseg}
        self._export_train_visuals(epoch, batch_ind, batch)
        return batch_log

    # -- --use_cnn: phases 2 and 3 with the U-Net scene-flow network ------------------------------------------------------
    def _sf_net_cnn(self, p, t):
        """Model.forward_sf_net with --use_cnn (:346-357)."""
        x = torch.cat([p, t], 1) if self.opt.time_dependent else p
        return self.net_sceneflow(x) / self.opt.sf_mag_div

    def _finish_step_cnn(self, epoch, batch_ind, batch, inp, depth_1, depth_2, steps, time_step, warm, do_reg, n_slots, fid1,
                         fid2, mul, disp_mode, sums, g_d1_main, g_d2_main):
        """Everything downstream of the depth maps when the scene-flow network is the U-Net (networks/FCNUnet.py): the
        network runs under PyTorch autograd on this package's convolution kernels, the geometry and the losses stay the ONE
        fused warp+loss launch -- its scene-flow gradient is what the U-Net's backward starts from, with the batch-global
        normaliser applied as a device scalar.  The regulariser's first evaluation is Euler evaluation 0, as in the MLP path."""
        opt, dev = self.opt, self.device
        B, _, H, W = inp.img_1.shape
        HW = H * W
        k = self._flat_sf
        cams = {kk: getattr(inp, kk) for kk in CAM_KEYS}
        mask_2 = inp.mask_2.reshape(B, H, W)
        mseg = inp.motion_seg_1.reshape(B, H, W) if opt.use_motion_seg else None
        cfg = ops.warp_cfg(B, H, W, midas_mask=opt.midas, crit_l2=warm, disp_mode=disp_mode, loss_on_sf=not opt.use_disp,
                           flow_mul=opt.flow_mul * mul, disp_mul=opt.disp_mul * mul)
        _late, n_global, capturing = parallel.agree_on_step_plan(dev, False, B, False)
        reg_coef = opt.acc_mul / (3.0 * n_global * HW + 1e-6)
        P1 = ops.unproject(depth_1, inp.R_1, inp.t_1, inp.K_inv, planar=True).requires_grad_(True)
        ts = inp.time_stamp_1
        # the whole batch goes through the U-Net under autograd: fail early, with the numbers, instead of somewhere inside it
        evals = steps + (1 if do_reg else 0)
        need = B * HW * self._cnn_bytes_per_px() * evals
        free, _total = self._free_hbm(dev)
        if need > free:
            msg = ('--use_cnn: the U-Net\'s autograd state for %d pairs at %dx%d (%d evaluations, %.0f B per pixel and '
                   'evaluation) needs %.1f GB, %.1f GB are free: use fewer pairs per step or a smaller --depth_keep_gb'
                   % (B, H, W, evals, self._cnn_bytes_per_px(), need / 2 ** 30, free / 2 ** 30))
            if self._cnn_px_measured > 0:        # a measured footprint: fail early, with the numbers
                raise RuntimeError(msg)
            import warnings                       # an a-priori estimate must not reject a configuration that may well run
            warnings.warn(msg + ' (a-priori estimate: nothing has been measured yet, trying anyway)')
        mem0 = torch.cuda.memory_allocated(dev)
        with torch.enable_grad():
            sf_acc, p, t, sf0 = None, P1, ts, None
            for i in range(steps):
                s_i = self._sf_net_cnn(p, t)
                sf0 = s_i if i == 0 else sf0
                sf_acc = s_i if sf_acc is None else sf_acc + s_i
                p, t = p + s_i, t + time_step
        # measured footprint of an evaluation (like _keep_per_px for the depth net's slots): the planner's next decisions
        # use it instead of the a-priori figure
        self._cnn_px_measured = max(self._cnn_px_measured, (torch.cuda.memory_allocated(dev) - mem0) / float(B * HW * steps))
        sf_all = sf_acc.detach().contiguous()
        sf_used = ops.mul_mask(torch.empty_like(sf_all), sf_all, mseg) if mseg is not None else sf_all
        csum, g_sf = torch.empty(4, device=dev), torch.empty_like(sf_all)
        ops.warp_loss_fused(cfg, depth_1, depth_2, inp.flow_1_2, mask_2, sf_used, cams, out=(csum, g_d1_main, g_d2_main, g_sf))
        if mseg is not None:
            ops.mul_mask(g_sf, g_sf, mseg)
        sums[:4] += csum
        parallel.all_reduce_sum_(sums[:4])
        scalars = ops.loss_finalize(cfg, sums)
        inv = scalars[0:1]
        ops.scale_add(g_d1_main, g_d1_main, scale_ptr=inv)
        ops.scale_add(g_sf, g_sf, scale_ptr=inv)
        sf_acc.backward(g_sf, retain_graph=bool(do_reg))        # parameter gradients land in the flat buffer's views
        if do_reg:
            with torch.enable_grad():
                sf1 = self._sf_net_cnn(P1 + sf0, ts + time_step)
                dsum = (sf1 - sf0).abs().sum()
            sums[4:5] += dsum.detach()
            (dsum * reg_coef).backward()
        parallel.all_reduce_sum_(sums[4:])
        ops.unproject_backward(P1.grad.contiguous(), True, cams['R_1'], cams['K_inv'], out=g_d1_main, accumulate=True)
        h_sf = None if capturing else k.all_reduce_grads(async_op=True)
        if not warm:
            g_d2 = ops.scale_add(g_d2_main, g_d2_main, scale_ptr=inv)
            self._depth_backward(inp.img_1, fid1, g_d1_main, slot0=0)
            self._depth_backward(inp.img_2, fid2, g_d2, slot0=n_slots)
            skip = None
            if self._gscale is not None:
                if parallel.is_distributed():          # both monitors: gradients [3], forward activations [6]
                    both = torch.stack([self._gscale[3], self._gscale[6]])
                    torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.MAX)
                    self._gscale[3:4].copy_(both[0:1])
                    self._gscale[6:7].copy_(both[1:2])
                ops.gscale_end(self._gscale)
                skip = self._gscale[4:5]
            self._flat_depth.all_reduce_and_adam_step(getattr(opt, 'grad_buckets', 4), skip_ptr=skip)
        if capturing:
            k.all_reduce_grads()
        if h_sf is not None:
            h_sf.wait()
        # (fp16 activations: an ACTIVATION that overflowed fp16 makes the depth map, and with it the scene-flow network's
        #  gradient, non-finite: that step skips BOTH updates.  A mere loss-scale overflow of the depth net's fp16 GRADIENTS
        #  leaves the depth maps finite and the scene-flow network's fp32 gradients valid: its update goes through -- the
        #  (flag, count) pair [8], [9] of the loss-scale state counts activation overflows only)
        k.adam_step(skip_ptr=None if (warm or self._gscale is None) else self._gscale[8:10])
        host = self._step_scalars_to_host(scalars, sums, warm)    # the only host synchronisation of the step
        acc_reg = opt.acc_mul * host[8] / (3.0 * n_global * HW + 1e-6) if do_reg else 0
        batch_log = {'size': opt.batch_size, 'loss': host[1] / mul, 'total_loss': host[1] / mul, 'flow_loss_1_2': host[2],
                     'disp_loss_1_2': host[3], 'sf_loss': host[4], 'acc_reg': acc_reg}
        if self._gscale is not None:
            batch_log['steps_skipped'] = self._steps_skipped
        self._last = {'depth_1': depth_1, 'depth_2': depth_2, 'mask_sum': host[5], 'sf_all': sf_all, 'mseg': mseg}
        self._export_train_visuals(epoch, batch_ind, batch)
        return batch_log

    def _step_scalars_to_host(self, scalars, sums, warm):
        """The step's ONE device -> host copy: the loss scalars, the regulariser sum and -- with fp16 activation storage --
        the loss-scale state's skip counters, so that a run whose updates are being skipped says so (batch_log
        'steps_skipped') instead of logging a loss that never moves.  An fp16 ACTIVATION overflow has no
        back-off (the loss scale is not at fault): a warning from the third consecutive skipped step on, an error after
        --max_act_overflow_skips (default 25) of them."""
        if self._gscale is None:
            return torch.cat([scalars, sums[4:5]]).tolist()
        host = torch.cat([scalars, sums[4:5], self._gscale[4:6], self._gscale[8:11]]).tolist()
        self._steps_skipped = int(host[10])
        run = 0 if warm else int(host[13])
        if run >= 3:
            import warnings
            limit = int(getattr(self.opt, 'max_act_overflow_skips', 25))
            msg = ('fp16 activation storage: an activation of the depth network left fp16\'s range in %d consecutive steps; '
                   'every one of them was skipped (%d skipped steps in all).  The loss scale cannot cure this: lower the '
                   'learning rate or run without --act_fp16' % (run, self._steps_skipped))
            if run >= limit:
                raise RuntimeError(msg)
            warnings.warn(msg)
        return host

    # -- the reference's `pred` dict and its export (scene_flow_motion_field.py:201-225, video_base.py:105-126) ----
    def _export_train_visuals(self, epoch, batch_ind, batch):
        opt = self.opt
        every = getattr(opt, 'vis_every_train', 0)
        if not every or np.mod(epoch, every) != 0:
            return
        indx = batch_ind if getattr(opt, 'vis_at_start', False) else getattr(opt, 'epoch_batches', 0) - batch_ind
        if indx > getattr(opt, 'vis_batches_train', 0):
            return
        pred = {k: v.data.cpu().numpy() for k, v in self._predict_on_batch(is_train=True).items()}
        outdir = join(self.full_logdir, 'visualize', 'epoch%04d_train' % epoch)
        makedirs(outdir, exist_ok=True)
        output = self.pack_output(pred, batch)
        if self.global_rank == 0 and self.visualizer is not None:
            self.visualizer.visualize(output, indx + (1000 * epoch), outdir)
        np.savez(join(outdir, 'rank%04d_batch%04d' % (self.global_rank, batch_ind)), **output)

    def pack_output(self, pred_all, batch):
        """video_base.py:105-126, key for key."""
        if 'pair_path' in batch:
            batch_size = len(batch['pair_path'])
        else:                                          # synthetic batches carry no paths
            batch_size = int((batch['img'] if 'img' in batch else batch['img_1']).shape[0])
        if 'img' not in batch:
            img_1, img_2 = batch['img_1'].cpu().numpy(), batch['img_2'].cpu().numpy()
        else:
            img_1 = img_2 = batch['img']
        output = {'batch_size': batch_size, 'img_1': img_1, 'img_2': img_2, **pred_all}
        if 'img' not in batch:
            output['flow_1_2'] = self._input.flow_1_2.cpu().numpy()
            output['flow_2_1'] = self._input.flow_2_1.cpu().numpy()
            if 'depth_pred_1' in batch:
                output['depth_nn_1'] = batch['depth_pred_1'].cpu().numpy()
        else:
            for src, dst in (('depth_pred', 'depth_nn'), ('depth_mvs', 'depth_gt'), ('cam_c2w', 'cam_c2w'), ('K', 'K')):
                if src in batch:
                    output[dst] = batch[src].cpu().numpy()
        output['pair_path'] = batch.get('pair_path', [])
        return output

    # ------------------------------------------------------------------------------------
    def _predict_on_batch(self, is_train=True):
        """Inference path (is_train=False, :265-275): depth + unprojection + one MLP evaluation."""
        if is_train:
            return self._train_pred()
        inp = self._input
        fid = inp.frame_id_1 if not self.opt.midas else None
        depth = self._depths_nograd(inp.img, fid)
        P = ops.unproject(depth, inp.R_1, inp.t_1, inp.K_inv, planar=True)
        if self.opt.use_cnn:
            with torch.no_grad():
                return {'depth': depth, 'sf_1_2': self._sf_net_cnn(P, inp.time_stamp_1)}
        sf = torch.empty_like(P)
        self._mlp.pack(self.net_sceneflow.parameter_list()[0::2], self.net_sceneflow.parameter_list()[1::2])
        ts = inp.time_stamp_1 if self.opt.time_dependent else None
        self._mlp.forward(P, ts, 0.0, 1.0 / self.opt.sf_mag_div, sf_out=sf)
        return {'depth': depth, 'sf_1_2': sf}

    def _train_pred(self):
        """The reference's train-time `pred` dict (:243-264 + `sf_loss_pp` of :308), materialised on demand from the
        state of the last `_train_on_batch` (the fused step itself never builds these thirteen surfaces): same keys,
        shapes and values, forward only."""
        if getattr(self, '_last', None) is None:
            raise RuntimeError('the training forward is fused into _train_on_batch: run a step first')
        inp, last = self._input, self._last
        cams = {k: getattr(inp, k) for k in CAM_KEYS}
        sf = last['sf_all']
        if last['mseg'] is not None:
            sf = ops.mul_mask(torch.empty_like(sf), sf, last['mseg'])
        sflow = sf.permute(0, 2, 3, 1)[..., None, :].contiguous()
        with torch.no_grad():
            dflow = self.depth_flow(last['depth_1'], last['depth_2'], inp.flow_1_2, **cams)
            pred = self.warp(last['depth_1'], last['depth_2'], inp.flow_1_2, inp.flow_2_1, sflow_1_2=sflow,
                             sflow_2_1=sflow, **cams)
        pred['sf_1_2'] = sf
        pred['global_p1'] = dflow['global_p1'].squeeze(3).permute(0, 3, 1, 2)
        pred['sf_by_dep_1_2'] = dflow['sf_by_depth']
        pred['sf_loss_pp'] = torch.abs(dflow['sf_by_depth'].squeeze(3).permute(0, 3, 1, 2) - sf).sum(1)
        return pred

    @staticmethod
    def depth2disp(depth):
        valid = (depth > 1e-2).float()
        return (1 / (depth + (1 - valid) * 1e-8)) * valid

    def _vali_on_batch(self, epoch, batch_idx, batch):
        """Disparity MSE against the MVS depth, optional export (models/video_base.py:72-103)."""
        self.eval()
        self.load_batch(batch)
        with torch.no_grad():
            pred = self._predict_on_batch(is_train=False)
            gt = batch['depth_mvs'].to(self.device)
            vali = gt > 1e-2
            loss = torch.nn.functional.mse_loss(self.depth2disp(pred['depth']) * vali, self.depth2disp(gt) * vali).item()
        every = getattr(self.opt, 'vis_every_vali', 0)
        if every and np.mod(epoch, every) == 0 and batch_idx < getattr(self.opt, 'vis_batches_vali', 0):
            pred = {k: v.cpu().numpy() for k, v in pred.items()}
            outdir = join(self.full_logdir, 'visualize', 'epoch%04d_vali' % epoch)
            makedirs(outdir, exist_ok=True)
            output = self.pack_output(pred, batch)
            if self.global_rank == 0 and self.visualizer is not None:
                self.visualizer.visualize(output, batch_idx + (1000 * epoch), outdir)
            np.savez(join(outdir, 'rank%04d_batch%04d' % (self.global_rank, batch_idx)), **output)
        return {'size': batch['img'].shape[0], 'loss': loss}

    def test_on_batch(self, batch_idx, batch):
        """models/video_base.py:128-155: predict, pack, cache, write `<output_dir>/epoch<e>_test/batch%04d.npz`."""
        if not hasattr(self, 'test_cache'):
            self.test_cache = []
        self.eval()
        self.load_batch(batch)
        with torch.no_grad():
            pred = self._predict_on_batch(is_train=False)
        if not hasattr(self, 'test_loss'):
            self.test_loss = 0
        pred = {k: v.cpu().numpy() for k, v in pred.items()}
        ep = getattr(self.opt, 'epoch', 0)
        outdir = join(getattr(self.opt, 'output_dir', self.full_logdir or '.'), 'epoch%s_test' % ('best' if ep < 0 else '%04d' % ep))
        if not hasattr(self, 'outdir'):
            self.outdir = outdir
        makedirs(outdir, exist_ok=True)
        output = self.pack_output(pred, batch)
        self.test_cache.append(output.copy())
        if self.global_rank == 0 and self.visualizer is not None:
            self.visualizer.visualize(output, batch_idx, outdir)
        np.savez(join(outdir, 'batch%04d' % batch_idx), **output)
        return output
