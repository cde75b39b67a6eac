/*
 * dvd_hip.h -- C ABI of libdvd_hip.so, the MI355X (gfx950) kernels behind the
 * dynamic-video-depth test-time-optimisation inner loop.
 *
 * The reference (google/dynamic-video-depth) has NO native boundary: its
 * operator API is Python classes.  Each entry point below therefore names the
 * reference Python interface it replaces (file:line under /root/reference) and
 * is what the host-side mirror in dynamic-video-depth_amd/dvd_hip binds with
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless noted;
 *     the caller (PyTorch's caching allocator) owns all memory, the library
 *     never allocates or frees; scratch is passed in (`*_workspace_bytes`).
 *   - all work is enqueued on `stream` (a hipStream_t) and returns at once.
 *   - return value: DVD_OK or a negative dvd_status; text via dvd_last_error()
 *     (thread local).  Nothing throws across the ABI.
 *   - tensors: depth [B,1,H,W]; flow [B,H,W,2] (x then y, pixels);
 *     mask [B,H,W]; scene flow / world points planar [B,3,H,W] unless a
 *     `*_interleaved` ([B,H,W,3]) variant is named; camera blocks as in the
 *     reference data files: row-vector convention, matrices stored transposed,
 *     each [B,3,3] or [B,3] (SURVEY.md section 8 header;
 *     scripts/preprocess/davis/generate_sequence_midas.py:69-76).
 */
#ifndef DVD_HIP_H
#define DVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVD_ABI_VERSION 7

typedef void* dvd_stream_t; /* hipStream_t */

typedef enum dvd_status {
  DVD_OK = 0,
  DVD_EINVAL = -1, /* bad shape / null pointer / unsupported option */
  DVD_EHIP = -2,   /* a HIP runtime call failed; see dvd_last_error() */
  DVD_ENOSPC = -3  /* workspace too small */
} dvd_status;

int dvd_abi_version(void);
const char* dvd_last_error(void);
/* Static properties the host needs for sizing (CU count etc.). */
int dvd_device_cu_count(void);

/* Algorithmic work (2 x multiply-accumulates) the matrix kernels were LAUNCHED with since the last reset, per kernel class --
 * the class is the template the dispatcher picked, so a profiler trace's per-kernel time and this table join on the kernel
 * name (bench.py `roofline_mfma`, tools/mfma_roofline.py).  Counted on the host at launch / graph-capture time; a replayed
 * graph does not count again.  Measurement aid with no counterpart in the reference; out[i] = class i, n <= DVD_FLOP_CLASSES. */
enum {
  DVD_FLOP_XCONV_1X1_WIDE = 0, /* xconv_kernel<4,2,2,2,FitOne>: 1x1, >= 256 rows, one position tile per block row       */
  DVD_FLOP_XCONV_WIDE = 1,     /* xconv_kernel<4,2,2,{2,4},*>: >= 256 output rows per group (3x3 and larger, other 1x1)   */
  DVD_FLOP_XCONV_128 = 2,      /* xconv_kernel<2,2,2,2,*>: 128 x 128 blocks                                                */
  DVD_FLOP_XCONV_SMALL = 3,    /* xconv_kernel<{1,2},2,1,4,*>: <= 64 rows per group (grouped 3x3 of ResNeXt, thin layers)  */
  DVD_FLOP_XWGRAD3 = 4,        /* xwgrad3_kernel: dense 3x3 weight gradient                                                */
  DVD_FLOP_XWGRAD3G = 5,       /* xwgrad3g_kernel: grouped 3x3 weight gradient                                             */
  DVD_FLOP_XWGRAD1B = 6,       /* xwgrad1b_kernel: wide 1x1 weight gradient                                                */
  DVD_FLOP_XWGRAD1S = 7,       /* xwgrad1s_kernel: other 1x1 weight gradients                                              */
  DVD_FLOP_XWGRADK = 8,        /* xwgradk_kernel: 5x5 / 7x7 / 11x11 weight gradients                                       */
  DVD_FLOP_MLP_FWD = 9,        /* mlp_fwd_kernel                                                                           */
  DVD_FLOP_MLP_DX = 10,        /* mlp_bwd_dx_kernel                                                                        */
  DVD_FLOP_MLP_DW = 11,        /* mlp_bwd_dw_kernel                                                                        */
  DVD_FLOP_CLASSES = 12
};
int dvd_flop_counters(double* out, int n, int reset);
/* The same accounting for the memory-bound helper kernels of a step (ABI 7; bench.py's roofline_helpers): every launch adds the
 * bytes it must move -- each operand tensor read once, each result written once -- to its class.  out[i] = class i. */
enum {
  DVD_BYTES_BNRELU_FWD = 0, /* bnrelu_fwd_kernel: x (+ residual) in, y out                                                       */
  DVD_BYTES_BNRELU_BWD = 1, /* bnrelu_bwd kernels: gy, y / x in, gx (+ g_residual) out, per-channel sums                          */
  DVD_BYTES_UPSAMPLE_FWD = 2,
  DVD_BYTES_UPSAMPLE_BWD = 3,
  DVD_BYTES_AMAX = 4,       /* amax_kernel: one read of the tensor                                                                */
  DVD_BYTES_PACK = 5,       /* weight maximum + fragment packing of the convolution weights                                       */
  DVD_BYTES_POOL = 6,       /* max-pool of the stem, forward and backward                                                         */
  DVD_BYTES_GCONV = 7,      /* grouped 3x3 with 8 channels per group on the vector unit (ResNeXt stage 1), all three passes       */
  DVD_BYTES_ELEMENTWISE = 8,/* scale_add, mul_mask, acc_reg, cast_scale, depth head                                               */
  DVD_BYTES_ADAM = 9,       /* the fused Adam step: param, grad, two moments in, param and moments out                            */
  DVD_BYTES_GEOMETRY = 10,  /* unproject forward / backward                                                                       */
  DVD_BYTES_CLASSES = 11
};
int dvd_byte_counters(double* out, int n, int reset);

/* Camera block of a batch of frame pairs (the eight tensors the reference
 * forwards take by name: losses/scene_flow_projection.py:114,222). */
typedef struct dvd_cameras {
  const float* R_1;   /* [B,3,3] cam1->world, stored transposed   */
  const float* R_2;   /* [B,3,3] cam2->world, stored transposed   */
  const float* R_1_T; /* [B,3,3] world->cam1                       */
  const float* R_2_T; /* [B,3,3] world->cam2                       */
  const float* t_1;   /* [B,3] camera centre 1 (world)             */
  const float* t_2;   /* [B,3]                                     */
  const float* K;     /* [B,3,3] intrinsics^T                      */
  const float* K_inv; /* [B,3,3] (intrinsics^-1)^T                 */
} dvd_cameras;

/* ------------------------------------------------------------------------
 * unproject: depth -> world points.
 * Replaces unproject_ptcld.forward (losses/scene_flow_projection.py:54-67)
 * and the `global_p1` half of flow_by_depth.forward (:127-131).
 * out_planar != 0: [B,3,H,W] (what the scene-flow MLP consumes,
 * models/scene_flow_motion_field.py:245); else [B,H,W,3].
 * bwd: g_depth (+)= (g_points @ R_1^T) . ray ;  accumulate != 0 adds. */
int dvd_unproject_fwd(const float* depth, const float* R, const float* t, const float* K_inv,
                      float* points, int out_planar, int B, int H, int W, dvd_stream_t stream);
int dvd_unproject_bwd(const float* g_points, int planar, const float* R, const float* K_inv,
                      const float* scale_or_null, float* g_depth, int accumulate,
                      int B, int H, int W, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused warp + reprojection + masked losses, forward AND backward in one
 * launch (the HBM-bound kernel of BASELINE.json).
 * Replaces, for the training step, the chain
 *   flow_by_depth.forward            losses/scene_flow_projection.py:114-153
 *   scene_flow_projection_slack.fwd  losses/scene_flow_projection.py:222-278
 *   F.grid_sample (bilinear, border) losses/scene_flow_projection.py:112,220
 *   Model._calc_loss / disp_loss     models/scene_flow_motion_field.py:140-150,285-324
 * and their autograd backward.
 *
 * sums[0..3] (device) receive  S0=sum m, S1=sum m*flow_err, S2=sum m*disp_err,
 * S3=sum m*sf_err  (un-normalised; deterministic two-stage reduction).
 * Gradients are written UN-NORMALISED, i.e. as if the loss were
 *   flow_mul*S1 + disp_mul*(use_disp ? S2 : S3);
 * the global 1/(S0+1e-8) (after the data-parallel all-reduce of the sums,
 * SURVEY.md section 8e) is applied by the consumers through a device scalar
 * (dvd_loss_finalize writes it).  g_depth_2 (bilinear scatter) is accumulated in Q31.32 fixed point in the LDS
 * windows of the tiles (ds_add_u64: integer adds commute, so the result is bitwise reproducible), cells only one
 * window covers are stored directly, the halo ring is summed over the <= 4 overlapping windows in a fixed order; only
 * taps that leave a window (|flow - pair mean| > 8 px) go through a record list and global fp32 atomics.  No memset of
 * g_depth_2 is needed. */
typedef struct dvd_warp_cfg {
  int B, H, W;
  int midas_mask;  /* 1: m *= [depth_1<100]*[W2.z<100]   (model:286-289)        */
  int crit_l2;     /* 1: squared flow error (warm phase, model:291), 0: L1      */
  int disp_mode;   /* 0:|z1-z2|  1:use_disp  2:use_disp_ratio  (model:140-150)  */
  int loss_on_sf;  /* 1: second loss term is sf_loss (not use_disp, model:310)  */
  float flow_mul;  /* already multiplied by `steps` when --weight_steps         */
  float disp_mul;
} dvd_warp_cfg;

size_t dvd_warp_loss_workspace_bytes(int B, int H, int W);
/* Test hook (process wide; not an environment switch): variant 0 = production -- since ABI 7 the strip kernel
 * (csrc/warp_strip.hip) for a backward call with the shipped flag set and W % 4 == 0 at tile = -1, px = 0, the tiled LDS
 * kernel otherwise; 1 = reference variant with global gathers + hardware atomics; 2 = the tiled LDS kernel of rounds 2-5
 * whatever the call; tile = -1 (auto) or an index of the tile-shape table; px = 0 (auto), 2 or 4 pixels per thread-step.
 * Every combination computes the same function. */
int dvd_warp_loss_select(int variant, int tile, int px);
/* Test hook (ABI 7): rows = image rows per unit of the strip kernel -- 0 = chosen from the shape and the device's block slots,
 * else a multiple of the shape's step height, at least two steps (more, shorter units: more seams between units); shape = 0
 * (production: 96-column strips in 32-row steps, 12 rows of vertical halo, 768-thread blocks), 1 (64 x 16, 512 threads),
 * 2 (96 x 16, 384 threads), 3 (as 0 with 8 rows of vertical halo).
 * dvd_warp_loss_workspace_bytes follows both. */
int dvd_warp_loss_strip_select(int rows, int shape);
int dvd_warp_loss_fused(const dvd_warp_cfg* cfg, const float* depth_1, const float* depth_2,
                        const float* flow_1_2, const float* mask_2, const float* sf_1_2,
                        const dvd_cameras* cams, void* workspace, size_t workspace_bytes,
                        float* sums /*[4]*/, float* g_depth_1, float* g_depth_2, float* g_sf_1_2,
                        dvd_stream_t stream);
/* Forward only (validation / logging): same sums, no gradients. */
int dvd_warp_loss_fwd(const dvd_warp_cfg* cfg, const float* depth_1, const float* depth_2,
                      const float* flow_1_2, const float* mask_2, const float* sf_1_2,
                      const dvd_cameras* cams, void* workspace, size_t workspace_bytes,
                      float* sums, dvd_stream_t stream);
/* scalars[0]=1/(S0+1e-8) [1]=loss [2]=flow_loss [3]=disp_loss [4]=sf_loss
 * [5]=S0; reads sums after the caller's (optional) all-reduce. */
int dvd_loss_finalize(const dvd_warp_cfg* cfg, const float* sums, float* scalars /*[8]*/,
                      dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Scene-flow field MLP (fused; fp16 matrix cores on two-term split, power-of-two scaled operands = fp32-class accuracy).
 * Replaces SceneFlowFieldNet.forward (networks/sceneflow_field.py:43-53):
 *   PeriodicEmbed of t and xyz (networks/blocks.py:19-34), 1x1 conv C_in->256,
 *   4 x (256->256), each + LeakyReLU(0.2), then 256->3 (blocks.py:50-102),
 * the division by sf_mag_div and one Euler step of
 * Model.forward_sf_net_multi_step (models/scene_flow_motion_field.py:346-367),
 * and their autograd backward.  Width 256 / 4 hidden layers are what the
 * reference Model builds (scene_flow_motion_field.py:107).
 *
 * Pixels are processed in tiles of 64; `n_pix` = B*H*W, p/sf tensors are planar
 * [B,3,H,W] (pix_per_img = H*W), t is [B,1,H,W].
 * Weights are the reference state_dict tensors convs.{0..5}.conv.{weight,bias}
 * (weight [out,in] row-major); dvd_sf_mlp_pack re-orders them into MFMA
 * fragment order (both orientations) inside `packed`.
 * `stash` (forward, optional) receives the embedding, the five hidden
 * activations (fp32, [channel][64 pixels] rows per tile) and one sign bit per
 * hidden unit, 5.72 KB per pixel; the backward kernels consume it. */
typedef struct dvd_mlp_desc {
  int n_freq_xyz;      /* 16 */
  int n_freq_t;        /* 16; ignored if !time_dependent */
  int time_dependent;  /* 1: input = [embed(t), embed(xyz)] */
  const float* freqs_xyz; /* device [n_freq_xyz] = linspace(1, n+1, n) as torch computes it */
  const float* freqs_t;   /* device [n_freq_t] */
  int stash_f16;          /* ABI 4: 1 = the five hidden activations of the stash are stored as _Float16 (3.2 KB per pixel instead of
                           * 5.7): the weight-gradient kernel then contracts fp32 gradients (two terms) against fp16 activations (one
                           * term, 2 MFMAs per product).  Losses and the input gradient do not depend on it. */
  float* fwd_monitor;     /* ABI 7, with stash_f16: a device float (slot [6] of the loss-scale state, dvd_gscale_*) that every
                           * stashing forward folds max |hidden activation| into (NaN counted as +Inf) -- an activation beyond
                           * fp16's range is stored as Inf and poisons the weight gradients: the step's guard skips it.  Or null. */
} dvd_mlp_desc;

/* Test / A-B hook (process wide, ABI 6): waves per workgroup of the forward / dX kernels.  4 (default) = two 256-thread
 * workgroups per CU, each wave two 32-channel row tiles; 8 = one 512-thread workgroup per CU (rounds 2-4).  Outputs, stashes
 * and input gradients are BIT-identical between the two (same products, same order); 0 = back to the default. */
int dvd_sf_mlp_select(int waves_per_workgroup);
int dvd_sf_mlp_in_channels(const dvd_mlp_desc* d);             /* 132 for the shipped config */
size_t dvd_sf_mlp_packed_bytes(const dvd_mlp_desc* d);
size_t dvd_sf_mlp_stash_bytes(const dvd_mlp_desc* d, long long n_pix);
size_t dvd_sf_mlp_gstash_bytes(long long n_pix);
int dvd_sf_mlp_pack(const dvd_mlp_desc* d, const float* const W[6], const float* const b[6],
                    void* packed, dvd_stream_t stream);
/* sf = MLP(p, t + t_offset) * out_scale;  optional fused Euler bookkeeping:
 * p_next = p + sf (may alias nothing), acc (in/out) += sf.  Any of sf_out,
 * p_next, acc may be null. */
int dvd_sf_mlp_fwd(const dvd_mlp_desc* d, const void* packed, const float* p, const float* t,
                   float t_offset, float out_scale, long long n_pix, int pix_per_img,
                   float* sf_out, float* p_next, float* acc, void* stash, dvd_stream_t stream);
/* Backward w.r.t. the input points and all pre-activations:
 *   g_out = gscale * (scale_ptr ? *scale_ptr : 1) * g_out1 + (g_out2 ? g_out2 : 0)
 *   g_p   = J^T (out_scale * g_out)  + (g_p_add ? g_p_add : 0)
 * writes the pre-activation gradients of layers 0..4 to `gstash` and
 * ACCUMULATES the last layer's dW (gW5 [3,256]) and db (gb5 [3]) with atomics. */
int dvd_sf_mlp_bwd_dx(const dvd_mlp_desc* d, const void* packed, const void* stash, float out_scale,
                      const float* g_out1, float gscale, const float* scale_ptr, const float* g_out2,
                      const float* g_p_add, long long n_pix, int pix_per_img, float* g_p, void* gstash,
                      float* gW5, float* gb5, dvd_stream_t stream);
/* dW_l += G_l . H_{l-1}^T and db_l += rowsum(G_l) for l = 0..4 (accumulating; per-workgroup partial
 * matrices in the tail of `gstash`, summed in fixed order: bitwise reproducible), gW[l] in the
 * reference layout [out,in]. */
int dvd_sf_mlp_bwd_dw(const dvd_mlp_desc* d, const void* stash, void* gstash, long long n_pix,
                      float* const gW[5], float* const gb[5], dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Elementwise helpers of the step.
 * dvd_scale_add: out = scale * (scale_ptr ? *scale_ptr : 1) * a + (b ? b : 0).
 *   Applies the late 1/(sum(mask)+1e-8) of Model._calc_loss
 *   (models/scene_flow_motion_field.py:297-306) from a device scalar.
 * dvd_acc_reg: the elementwise half of Model._opt_reg (:326-344):
 *   g_sf1 = coef * sign(sf1 - sf0),  abs_sum (+)= sum |sf1 - sf0|   (coef = acc_mul/(3N+1e-6)).
 * dvd_adam_step: torch.optim.Adam step (:113-115,212-213; betas from
 *   options/options_train.py:84-87) on a flat buffer with grad = s*grad1 + grad2. */
int dvd_scale_add(float* out, const float* a, float scale, const float* scale_ptr, const float* b,
                  long long n, dvd_stream_t stream);
/* The tail of the MiDaS depth head in one pass each way (ABI 7): depth = 10000 / clamp(relu(v), min = 1e-2) of
 * third_party/MiDaS.py:192-195,240-242 -- as torch evaluates it: reciprocal, then x 10000 (bit-identical); backward
 * g_v = -(g_depth * 10000) / v^2 where v >= 1e-2, else 0 (autograd's formula with clamp's mask).  16-byte aligned pointers, any n. */
int dvd_depth_tail_fwd(const float* v, float* depth, long long n, dvd_stream_t stream);
int dvd_depth_tail_bwd(const float* v, const float* g_depth, float* g_v, long long n, dvd_stream_t stream);
/* out[b,c,p] = a[b,c,p] * mask[b,p]: `sf_1_2 *= motion_seg_1` of --use_motion_seg
 * (models/scene_flow_motion_field.py:253-254) and the matching gradient mask; out may alias a. */
int dvd_mul_mask(float* out, const float* a, const float* mask, int B, int C, long long HW, dvd_stream_t stream);
size_t dvd_acc_reg_workspace_bytes(void);
int dvd_acc_reg(const float* sf0, const float* sf1, float coef, float* g_sf1, void* workspace,
                float* abs_sum, int accumulate, long long n, dvd_stream_t stream);
int dvd_adam_step(float* param, const float* grad1, float scale, const float* scale_ptr,
                  const float* grad2, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                  float beta1, float beta2, float eps, int step, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Per-pixel surfaces of the warp modules (forward values; the training gradient path is
 * dvd_warp_loss_fused, which never materialises them).  Replaces, for visualisation / export /
 * inference callers (models/scene_flow_motion_field.py:215-225, models/video_base.py:105-126),
 *   flow_by_depth.forward                losses/scene_flow_projection.py:114-153
 *   scene_flow_projection_slack.forward  losses/scene_flow_projection.py:222-278
 * Every output is optional (NULL = not wanted).  3-vectors are interleaved [B,H,W,3] (the
 * reference's [B,H,W,1,3]), flows [B,H,W,2], depths [B,1,H,W].  sflow_1_2 is the reference's
 * interleaved scene flow [B,H,W,(1,)3] or NULL for zero.  Same fp32 operation order as the
 * reference, so index masks and tap indices are bit-identical to its CPU path. */
typedef struct dvd_surfaces {
  float* global_p1;          /* P1 = (d1 (x,y,1) K_inv) R_1 + t_1                      (:127-131,235-237) */
  float* warped_global_p2;   /* bilinear sample of frame 2's world points at the flow target (:133-135) */
  float* sf_by_depth;        /* warped_global_p2 - global_p1                              (:136)         */
  float* staticflow_1_2;     /* rigid-scene flow (= flow_by_depth's dflow_1_2)           (:138-149,267) */
  float* dflow_1_2;          /* flow of the scene-flow-advected point                     (:244-265)     */
  float* depth_image_1_2;    /* z of the advected point in image 2                        (:268)         */
  float* depth_warp_1_2;     /* bilinear sample of depth_2                                (:274-276)     */
  float* p1_camera_2;        /* (P1 + sflow - t_2) R_2_T                                  (:244)         */
  float* warped_p2_camera_2; /* bilinear sample of frame 2's camera-space points          (:240-242)     */
} dvd_surfaces;
int dvd_warp_surfaces(const float* depth_1, const float* depth_2, const float* flow_1_2, const float* sflow_1_2,
                      const dvd_cameras* cams, const dvd_surfaces* out, int B, int H, int W, dvd_stream_t stream);
/* Vector-Jacobian product of dvd_warp_surfaces: g_surfaces holds the upstream gradient of every surface that has one
 * (NULL = none), in the surface's layout; writes d/d depth_1, d/d depth_2 (zeroed inside; the flow is data, so the taps
 * are a scatter-add) and, if not NULL, d/d sflow_1_2.  This is the autograd backward of the reference's module forms
 * flow_by_depth / scene_flow_projection_slack (losses/scene_flow_projection.py:114-153,222-278), including the gradient
 * cut at behind-camera pixels (:253-263). */
int dvd_warp_surfaces_bwd(const float* depth_1, const float* depth_2, const float* flow_1_2, const float* sflow_1_2,
                          const dvd_cameras* cams, const dvd_surfaces* g_surfaces, float* g_depth_1, float* g_depth_2,
                          float* g_sflow_1_2, int B, int H, int W, dvd_stream_t stream);

/* Stand-alone flow warp: BackwardWarp.forward (losses/scene_flow_projection.py:281-307) =
 * F.grid_sample(buffer, (x,y)+flow, bilinear, padding_mode='border', align_corners=True) on
 * buffer [B,C,H,W]; bwd = its gradient w.r.t. the buffer (bilinear scatter-add; g_buffer is
 * zeroed inside the call). */
int dvd_flow_warp_fwd(const float* buffer, const float* flow_1_2, float* out, int B, int C, int H, int W,
                      dvd_stream_t stream);
int dvd_flow_warp_bwd(const float* g_out, const float* flow_1_2, float* g_buffer, int B, int C, int H, int W,
                      dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Grouped 3x3 convolution, 8 channels per group, stride 1, pad 1, NCHW fp32.
 * Replaces the `conv2` of the ResNeXt-101 32x8d stage-1 bottlenecks inside the MiDaS encoder
 * (third_party/midas_blocks.py:35-50 -> torchvision ResNet(Bottleneck, groups=32,
 * width_per_group=8): stage 1 is 256 channels = 32 groups x 8) -- forward, backward-data and
 * backward-weight of nn.Conv2d(C, C, 3, padding=1, groups=C/8, bias=False).
 * x, y, gy, gx: [N,C,H,W]; w, gw: [C,8,3,3].  bwd_weight needs a workspace
 * (per-tile partial sums, reduced in a fixed order: deterministic); accumulate != 0 adds into gw. */
int dvd_gconv3x3_c8_fwd(const float* x, const float* w, float* y, int N, int C, int H, int W,
                        dvd_stream_t stream);
int dvd_gconv3x3_c8_bwd_data(const float* gy, const float* w, float* gx, int N, int C, int H, int W,
                             dvd_stream_t stream);
size_t dvd_gconv3x3_c8_wgrad_workspace_bytes(int N, int C, int H, int W);
int dvd_gconv3x3_c8_bwd_weight(const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                               size_t workspace_bytes, int N, int C, int H, int W, dvd_stream_t stream);

/* Same convolution with 32 channels per group (fp32 MFMA): the stride-1 bottlenecks of ResNeXt
 * stage 3 (width 1024 = 32 groups x 32).  w, gw: [C,32,3,3].  All three entry points take a workspace
 * of dvd_gconv3x3_c32_workspace_bytes (fragment-ordered weights for fwd / bwd_data, per-block
 * partial sums for bwd_weight). */
size_t dvd_gconv3x3_c32_workspace_bytes(int N, int C, int H, int W);
int dvd_gconv3x3_c32_fwd(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes, int N,
                         int C, int H, int W, dvd_stream_t stream);
int dvd_gconv3x3_c32_bwd_data(const float* gy, const float* w, float* gx, void* workspace, size_t workspace_bytes,
                              int N, int C, int H, int W, dvd_stream_t stream);
int dvd_gconv3x3_c32_bwd_weight(const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                                size_t workspace_bytes, int N, int C, int H, int W, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused eval-mode BatchNorm (+ residual add) (+ ReLU), NCHW fp32.  Replaces
 * relu(bn(x)) / relu(bn3(x) + skip) of the ResNeXt bottlenecks of the MiDaS encoder (torchvision
 * resnet.py Bottleneck via third_party/midas_blocks.py:35-50); the depth nets are in eval mode while
 * training (models/scene_flow_motion_field.py:157,168), so running statistics are used and gamma/beta
 * still get gradients.  y = max(0, x*s[c] + b[c] (+ residual)), s = gamma/sqrt(var+eps), b = beta - mean*s.
 * bwd: gx = g*s, g_residual = g, g_beta = sum g, g_gamma = sum g*(x-mean)/sqrt(var+eps), g = gy*[y>0];
 * any of gx, g_residual, g_gamma, g_beta may be NULL; channel sums are deterministic (two stages).  g_amax (optional):
 * device scalar into which max|g| is folded (atomic max; zero it first) -- the operand scale of the gradient kernels
 * that consume g (dvd_xconv_fwd / dvd_xwgrad*). */
int dvd_bnrelu_fwd(const float* x, const float* residual, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* y, int N, int C, int HW, int relu, dvd_stream_t stream);
size_t dvd_bnrelu_bwd_workspace_bytes(int N, int C, int HW);
int dvd_bnrelu_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* mean,
                   const float* var, float eps, float* gx, float* g_residual, float* g_gamma, float* g_beta,
                   void* workspace, size_t workspace_bytes, int N, int C, int HW, int relu, float* g_amax, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Bilinear up-sampling of [planes, H_in, W_in] -> [planes, H_out, W_out] (planes = N*C) and its backward.
 * Replaces F.interpolate(mode='bilinear') of the MiDaS decoder: align_corners=True at the end of every
 * FeatureFusionBlock (third_party/midas_blocks.py:164-166), align_corners=False in the output head
 * (midas_blocks.py:71-99, MiDaS.py:190).  Same source-index / weight formulas as ATen. */
int dvd_upsample_bilinear_fwd(const float* x, float* y, long long planes, int H_in, int W_in, int H_out, int W_out,
                              int align_corners, dvd_stream_t stream);
int dvd_upsample_bilinear_bwd(const float* gy, float* gx, long long planes, int H_in, int W_in, int H_out,
                              int W_out, int align_corners, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * max|x| of a tensor into a device scalar: out[0] = max(out[0], max|x|) (atomic; zero `out` first, or pass a running
 * bound).  The matrix kernels derive the power-of-two scale of their fp16 operand split from such scalars on the device;
 * nothing reads them on the host.  x: any 4-byte aligned address (a misaligned head is read by scalar loads). */
int dvd_amax(const float* x, long long n, float* out, dvd_stream_t stream);

/* out[c] = sum over n, p of x[n][c][p] of an NCHW fp32 tensor (HW pixels per plane) -- the bias gradient of a convolution,
 * conv.py _XConv.backward (the reference leaves it to autograd: torch.nn.Conv2d(bias=True), third_party/midas_blocks.py) --
 * and, when amax_out is not NULL, amax_out[0] = max(amax_out[0], max|x|) from the same read (dvd_amax's contract).
 * Deterministic: fixed summation order.  workspace: dvd_chansum_workspace_bytes(N, C) bytes of device memory. */
size_t dvd_chansum_workspace_bytes(int N, int C);
int dvd_chansum(const float* x, int N, int C, long long HW, float* out, float* amax_out, void* workspace, size_t workspace_bytes,
                dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense stride-1 "same" convolution (odd k x k up to 11, any channel counts), NCHW fp32, on the 16-bit
 * matrix cores with every fp32 operand, scaled by a power of two, split into two fp16 terms (three partial
 * products, fp32 accumulation: fp32-class accuracy, csrc/dvd_split.h, csrc/xconv.hip).  Replaces the dense
 * nn.Conv2d forward and backward-data of the depth networks:
 *   third_party/midas_blocks.py:102-168 (ResidualConvUnit / FeatureFusionBlock 3x3),
 *   third_party/MiDaS.py:186-195 (scratch.layerK_rn, output_conv),
 *   third_party/midas_blocks.py:35-50 (1x1 convolutions of the ResNeXt-101 32x8d bottlenecks),
 *   third_party/hourglass.py:21-57 (1x1 / k x k inception branches).
 * dvd_xconv_pack re-orders w [Cout,Cin,k,k] into MFMA fragment order, already split (once per weight
 * update); transposed != 0 packs the backward-data operator (channel roles swapped, taps flipped), which
 * is then run by the same dvd_xconv_fwd with Cin/Cout exchanged.
 * dvd_xconv_fwd: y = act_out( conv(act_in(x)) + bias + res' ) * [mask_src > 0]
 *   flags bit0: act_in = ReLU on the input as it is staged (conv(relu(x)) of ResidualConvUnit),
 *         bit1: act_out = ReLU, bit2: res' = relu(residual) instead of residual;
 *   bias [Cout], residual / mask_src [N,Cout,H,W] may be NULL.  mask_src is the ReLU mask of the
 *   backward-data pass (gx = dgrad(gy) * [x > 0]).
 *   Stride 2 (3x3 kernels, nn.Conv2d(.., 3, stride=2, padding=1): torchvision's Bottleneck.conv2 at the entry of ResNeXt
 *   stages 2-4, reached through third_party/midas_blocks.py:35-50).  H x W are the dimensions of the FULL-resolution tensor:
 *         bit3: stride-2 forward -- x is [N,Cin,H,W], y (and residual / mask_src) [N,Cout,(H+1)/2,(W+1)/2]; the haloed input
 *               tile is staged in LDS as four phase planes, every tap stays one constant LDS offset;
 *         bit4: backward-data of that convolution -- x is the gradient [N,Cin,(H+1)/2,(W+1)/2] of the strided output (Cin =
 *               the convolution's output channels, weights packed transposed), y [N,Cout,H,W]; the kernel stages the
 *               gradient AS IF zero-interleaved (v[2i][2j] = g[i][j]): the interleaved tensor is never written to HBM. */
/* Channel counts are totals; groups > 1 = grouped convolution with weight [Cout][Cin / groups][k][k] (nn.Conv2d's
 * layout): the 64-channels-per-group 3x3 convolutions of ResNeXt stage 4 run here with groups = 32. */
size_t dvd_xconv_packed_bytes(int Cout, int Cin, int KS, int groups, int transposed);
int dvd_xconv_pack(const float* w, void* packed, int Cout, int Cin, int KS, int groups, int transposed, dvd_stream_t stream);
/* Eval-mode BatchNorm of the convolution's output fused into the epilogue (the depth nets are always in eval mode while
 * training, models/scene_flow_motion_field.py:157,168):  y = act((conv(x) + bias - mean) / sqrt(var + eps) * gamma + beta
 * + residual).  gamma / beta null = BatchNorm2d(affine=False) (hourglass.py:21-57).  bn null = no BatchNorm. */
typedef struct dvd_bn_params {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
} dvd_bn_params;
/* x_amax: device scalar holding max|x| of the WHOLE input tensor, or an upper bound of it (written by dvd_amax or by the
 * producing kernel through its y_amax): the power-of-two operand scale of the two-term fp16 split is derived from it on the
 * device (csrc/dvd_split.h).  y_amax (optional): device scalar into which max|y| of this launch is folded with an atomic
 * max -- zero it (or pass a running bound) before the launch. */
int dvd_xconv_fwd(const float* x, const float* x_amax, const void* packed, const float* bias, const float* residual,
                  const float* mask_src, const dvd_bn_params* bn, float* y, float* y_amax, int N, int Cin, int Cout, int H,
                  int W, int KS, int groups, int flags, dvd_stream_t stream);
/* Test / A-B hook (process wide, like dvd_warp_loss_select): block shape of dvd_xconv_fwd for >= 256 output channels.
 * 0 = automatic (256 channels x 128 positions for 1x1 kernels, 256 x 256 for k >= 3), 1 = round 2's 128 x 128 blocks,
 * 2 / 3 = force 256 x 128 / 256 x 256, 4 = round 2's blocks on the generic pointer-addressed main loop (the path shapes
 * with Cin % 16 != 0 take), 5 = 128 x 128 blocks with one activation stage at three blocks per CU, 6 = 1x1 kernels without
 * their two-chunk loop, 7 = 1x1 kernels with the narrow (dword) epilogue instead of the LDS-transposed 16-byte one.
 * Results are identical in every setting (same products, same K order). */
int dvd_xconv_select(int cfg);
/* Transposed / forward packing with every weight of output channel co scaled by gamma[co] / sqrt(var[co] + eps): the
 * backward-data pass through a fused BatchNorm is dvd_xconv_fwd on the masked, UNSCALED output gradient. */
int dvd_xconv_pack_scaled(const float* w, void* packed, int Cout, int Cin, int KS, int groups, int transposed,
                          const float* bn_gamma, const float* bn_var, float bn_eps, dvd_stream_t stream);
/* All packings of a network in TWO launches (round 6).  items[i] describes one dvd_xconv_pack / dvd_xconv_pack_scaled call
 * (var == NULL: unscaled); `packed` buffers of dvd_xconv_packed_bytes each, all distinct.  `table`: device memory of
 * dvd_xconv_pack_table_bytes(n) that the launches read; upload != 0 (re)writes it from `items` with a blocking copy -- pass it
 * once per item list, OUTSIDE a graph capture -- upload == 0 reuses what the last upload left (`items` must be the same list).
 * The packed bytes are identical to n single calls.  The reference has no counterpart: nn.Conv2d consumes its fp32 weight
 * directly (third_party/midas_blocks.py:35-50); the packing is the operand format of this library's matrix kernels. */
typedef struct dvd_xpack_item {
  const float* w;
  void* packed;
  const float* gamma;   /* may be NULL (scale 1 / sqrt(var + eps)) */
  const float* var;     /* NULL: no BatchNorm scale */
  float eps;
  int Cout, Cin, KS, groups, transposed;
} dvd_xpack_item;
size_t dvd_xconv_pack_table_bytes(int n);
int dvd_xconv_pack_many(const dvd_xpack_item* items, int n, void* table, size_t table_bytes, int upload, dvd_stream_t stream);
/* After the weight-gradient kernel ran on the unscaled masked gradient (dW holds dWu [Cout][K]): scales dW in place and
 * derives the BatchNorm-gamma and convolution-bias gradients (csrc/bnrelu.hip); dbeta = per-channel sum of the masked
 * gradient (dvd_bnrelu_bwd with x = null). */
int dvd_convbn_finalize(const float* W, float* dW, const float* dbeta, const float* gamma, const float* mean, const float* var,
                        float eps, const float* conv_bias, int Cout, int K, float* dgamma, float* dconv_bias, dvd_stream_t stream);
/* Backward-weight of the same convolutions (k = 1 and 3), exact fp32 MFMA, deterministic two-stage sum
 * (csrc/xwgrad.hip): gw[Cout,Cin,k,k] = sum_{n,p} gy[n,co,p] * act(x)[n,ci,p + tap], act = ReLU if relu_in.
 * Replaces the autograd weight gradient of the nn.Conv2d named above (MIOpen accumulates it with atomics). */
size_t dvd_xwgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int KS);
int dvd_xwgrad(const float* x, const float* gy, float* gw, void* workspace, size_t workspace_bytes, int N, int Cin,
               int Cout, int H, int W, int KS, int relu_in, dvd_stream_t stream);

/* The 3x3 case on the 16-bit matrix cores with the two-term fp16 split of dvd_xconv_fwd (fp32-class accuracy,
 * deterministic; csrc/xwgrad3.hip): what the MiDaS decoder's weight gradients run on.  x_amax / gy_amax: device scalars
 * with max|x| / max|gy| (or upper bounds) of the whole tensors, see dvd_xconv_fwd. */
size_t dvd_xwgrad3_workspace_bytes(int N, int Cin, int Cout, int H, int W, int groups);
int dvd_xwgrad3(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int groups, int relu_in, dvd_stream_t stream);
/* ... and the 1x1 case (ResNeXt bottleneck convolutions), same arithmetic: a K-contiguous "NT" GEMM over the pixels. */
size_t dvd_xwgrad1s_workspace_bytes(int N, int Cin, int Cout, int H, int W);
int dvd_xwgrad1s(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                 size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in, dvd_stream_t stream);
/* The same, and additionally (gy_rowsum != NULL) gy_rowsum[co] = sum over images and pixels of gy[n][co][:] -- the bias
 * gradient of the convolution / the shift gradient of a BatchNorm fused behind it (nn.Conv2d bias, BatchNorm2d.bias of the
 * ResNeXt bottlenecks behind third_party/midas_blocks.py:35-50), by one extra pass over gy in fixed order (deterministic).
 * Workspace: dvd_xwgrad1s_workspace_bytes. */
int dvd_xwgrad1s_rowsum(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, float* gy_rowsum,
                        void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in,
                        dvd_stream_t stream);
/* Test / A-B hook (process wide): 0 = automatic (256 x 256-channel workgroups for wide 1x1 layers, 128 x 128 otherwise),
 * 1 = always 128 x 128, 2 = round 3's row step everywhere (no buffer-load / interleaved-staging instantiations of the 3x3 and
 * wide 1x1 kernels, grouped layers on the 64 x 64 channel blocks).  Same products and the same per-element summation order
 * within a slice; the number of slices (partial sums added at the end) differs, so results agree to fp32 rounding, not bitwise. */
int dvd_xwgrad_select(int variant);
/* The same for dense 5x5 / 7x7 / 11x11 stride-1 "same" convolutions (round 4: third_party/hourglass.py:21-57, the inception
 * branches; round 3 left their weight gradient to MIOpen): split-operand MFMA, 32 x 32 channels per block, one wave per kernel
 * row, deterministic.  gw [Cout][Cin][KS][KS]. */
size_t dvd_xwgradk_workspace_bytes(int N, int Cin, int Cout, int H, int W, int KS);
int dvd_xwgradk(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int KS, int relu_in, dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * Flow-consistency (occlusion) + out-of-bounds mask of one direction of a frame pair.
 * Replaces scripts/preprocess/davis/generate_flows.py:57-82,139-148:
 *   mask = clip([ || grid_sample(flow_a, (x,y)+flow_b, zeros padding, align_corners) + flow_b || > 1 ]
 *               + [ (x,y)+flow_b outside the image ], 0, 1)
 * mask_1 = f(flow_a = flow_1_2, flow_b = flow_2_1), mask_2 = f(flow_2_1, flow_1_2); flows [B,H,W,2], mask [B,H,W]
 * in {0,1} (1 = occluded / leaves the image; the training masks are 1 - mask, generate_sequence_midas.py:144-147).
 * Same fp32 operation order as the reference's numpy / ATen path: the integer masks are bit-identical. */
int dvd_flow_consistency_mask(const float* flow_a, const float* flow_b, float* mask, int B, int H, int W,
                              dvd_stream_t stream);

/* ------------------------------------------------------------------------
 * fp16 ACTIVATION STORAGE (round 4; BASELINE.json configs[4]: "fp16 activations with fp32 loss accumulation").
 * Reference: the same layers as the fp32 entry points above (third_party/MiDaS.py:186-246, third_party/midas_blocks.py:35-168);
 * the reference itself has no reduced-precision mode.  Every tensor between the encoder stem and the depth head is stored as
 * _Float16 in HBM; the image, the depth map, parameters, parameter gradients, loss sums and optimiser state stay fp32 and every
 * accumulation is fp32.  An fp16 activation IS the matrix operand (one term, no scale): activation x two-term weight = 2 MFMAs per
 * product instead of 3, activation x activation (weight gradients) = 1 instead of 3.
 *   "_h" entry points: activations / gradients are _Float16 (void*).
 *   "_t" entry points: `f16` selects the storage (0 = float, 1 = _Float16) of the activation / gradient tensors.
 *   out_scale: device scalar the PARAMETER gradients are multiplied by -- 1 / (loss scale of the fp16 gradients), state[1] of the
 *   loss-scale state below (null = 1). */
/* dvd_xconv_fwd with fp16 input; out_f16 = 0 writes fp32 (then residual / mask_src are fp32 too).  Needs Cin / groups % 16 == 0. */
int dvd_xconv_fwd_h(const void* x, const void* packed, const float* bias, const void* residual, const void* mask_src,
                    const dvd_bn_params* bn, void* y, float* y_amax, int N, int Cin_total, int Cout_total, int H, int W, int KS,
                    int groups, int flags, int out_f16, dvd_stream_t stream);
int dvd_xwgrad3_h(const void* x, const void* gy, const float* out_scale, float* gw, void* workspace, size_t workspace_bytes, int N,
                  int Cin_total, int Cout_total, int H, int W, int groups, int relu_in, dvd_stream_t stream);
int dvd_xwgrad1s_h(const void* x, const void* gy, const float* out_scale, float* gw, void* workspace, size_t workspace_bytes, int N,
                   int Cin, int Cout, int H, int W, int relu_in, dvd_stream_t stream);
int dvd_bnrelu_fwd_t(const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                     const float* var, float eps, void* y, int f16, int N, int C, int HW, int relu, dvd_stream_t stream);
int dvd_bnrelu_bwd_t(const void* gy, const void* y, const void* x, const float* gamma, const float* mean, const float* var,
                     float eps, void* gx, void* g_residual, float* g_gamma, float* g_beta, void* workspace,
                     size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int HW, int relu, float* g_amax,
                     dvd_stream_t stream);
/* The same with the maxima the consuming convolutions take their operand scale from (round 6; before: a dvd_amax pass over
 * the tensor): y_amax[0] = max(y_amax[0], max|y|); gx_amax[0] = max(gx_amax[0], max|gx|) (exact: |gamma / sqrt(var + eps)| *
 * max|g| per channel).  NULL: not wanted.  fp32 and fp16 storage alike. */
int dvd_bnrelu_fwd_m(const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                     const float* var, float eps, void* y, int f16, int N, int C, int HW, int relu, float* y_amax,
                     dvd_stream_t stream);
int dvd_bnrelu_bwd_m(const void* gy, const void* y, const void* x, const float* gamma, const float* mean, const float* var,
                     float eps, void* gx, void* g_residual, float* g_gamma, float* g_beta, void* workspace,
                     size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int HW, int relu, float* g_amax,
                     float* gx_amax, dvd_stream_t stream);
int dvd_upsample_bilinear_fwd_t(const void* x, void* y, int f16, long long planes, int H_in, int W_in, int H_out, int W_out,
                                int align_corners, dvd_stream_t stream);
int dvd_upsample_bilinear_bwd_t(const void* gy, void* gx, int f16, long long planes, int H_in, int W_in, int H_out, int W_out,
                                int align_corners, dvd_stream_t stream);
int dvd_gconv3x3_c8_fwd_t(const void* x, const float* w, void* y, int f16, int N, int C, int H, int W, dvd_stream_t stream);
int dvd_gconv3x3_c8_bwd_data_t(const void* gy, const float* w, void* gx, int f16, int N, int C, int H, int W,
                               dvd_stream_t stream);
int dvd_gconv3x3_c8_bwd_weight_t(const void* x, const void* gy, float* gw, int accumulate, void* workspace,
                                 size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int H, int W,
                                 dvd_stream_t stream);
/* The depth head `ReLU -> Conv2d(C, 1, 1)` (third_party/MiDaS.py:192-194), the fp16 / fp32 boundary: y fp32 [N,1,H,W] from
 * x [N,C,H,W] (C <= 64, H * W % 4 == 0); backward: gx = S * w[c] * gy * [x > 0] (storage of x), gw [C], gb [1] fp32 from the
 * unscaled fp32 gy; gscale_state: the loss-scale state (null: S = 1).  Deterministic (fixed-order partial sums). */
/* fwd_amax (optional): max |y| of the head output is folded into it (NaN counted as +Inf) -- slot [6] of the loss-scale state,
 * the forward monitor of the fp16 overflow guard. */
int dvd_head1x1_fwd(const void* x, int f16, const float* w, const float* bias, float* y, float* fwd_amax, int N, int C, int HW,
                    int relu_in, dvd_stream_t stream);
size_t dvd_head1x1_bwd_workspace_bytes(int C);
int dvd_head1x1_bwd(const void* x, int f16, const float* w, const float* gy, const float* gscale_state, void* gx, float* gw,
                    float* gb, void* workspace, size_t workspace_bytes, int N, int C, int HW, int relu_in, dvd_stream_t stream);
/* Loss scale of the fp16 gradients, kept on the device (state: 16 floats since ABI 7; [0] = S, [1] = 1 / S, [2] = target exponent,
 * [3] = observed max |S g| this step, [4] = skip flag, [5] = skipped steps, [6] = forward monitor: max |activation| the fp16-output
 * convolution epilogues and the depth head folded in this step -- every maximum counts a NaN as +Inf).  begin: S = 2^(target - ceil(log2(max|g| * max|w|)))
 * from the device scalar max|g_out| and the n_w head weights; end (once per step, before the optimiser): overflow -> skip flag +
 * back-off, small observed maximum -> raise the target; an ACTIVATION beyond fp16's range ([6] >= 65504) -> skip flag only.  Policy and thresholds: csrc/a16.hip.
 * ABI 7: [8] = this step was skipped for an activation overflow, [9] = steps skipped for that reason so far (the (flag, count) pair
 * for dvd_adam_step_guarded of a network whose fp32 gradients survive a mere loss-scale overflow), [10] = consecutive such skips;
 * step_begin (first launch of every TRAINING step) clears the forward monitor [6], which every fp16 forward -- validation and
 * inference included -- folds into. */
int dvd_gscale_init(float* state, float target_exponent, dvd_stream_t stream);
int dvd_gscale_step_begin(float* state, dvd_stream_t stream);
int dvd_gscale_begin(float* state, const float* g_amax, const float* w, int n_w, dvd_stream_t stream);
int dvd_gscale_end(float* state, dvd_stream_t stream);
/* out[i] = scale[0] * in[i] as fp32 (n % 4 == 0): the gradient leaving the fp16 region towards the fp32 stem. */
int dvd_cast_scale_f32(const void* in, int f16, float* out, long long n, const float* scale, dvd_stream_t stream);
/* dvd_adam_step that does nothing when skip_flag[0] != 0 (state[4] above): the GradScaler "skipped step".  skip_flag points INTO
 * the loss-scale state: skip_flag[1] (state[5], the steps skipped so far) is subtracted from `step` for Adam's bias correction,
 * so that skipped steps do not advance the optimiser (round 5). */
int dvd_adam_step_guarded(float* param, const float* grad1, float scale, const float* scale_ptr, const float* grad2,
                          float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2, float eps,
                          int step, const float* skip_flag, dvd_stream_t stream);

/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNeXt stem (third_party/midas_blocks.py:35-45), ATen's semantics (the first
 * maximum of a window takes the gradient).  x fp32 [planes][H][W] -> y [planes][Ho][Wo] (Ho = (H - 1) / 2 + 1) in fp32 or
 * _Float16 (y_f16: the fp32 -> fp16 boundary of fp16 activation storage), index: one byte per output (winner's window position).
 * Backward: gx fp32 = out_scale[0] * (gather of gy over the windows that contain the pixel); deterministic. */
int dvd_maxpool3s2_fwd(const float* x, void* y, int y_f16, unsigned char* index, long long planes, int H, int W,
                       dvd_stream_t stream);
int dvd_maxpool3s2_bwd(const void* gy, int gy_f16, const unsigned char* index, float* gx, const float* out_scale,
                       long long planes, int H, int W, dvd_stream_t stream);
/* x[:, :, ::2, ::2] as a contiguous tensor [planes][(H + 1) / 2][(W + 1) / 2] and its backward (zeros with gy at the even
 * positions), fp32 or _Float16 (ABI 7): the sub-sampling around the stride-2 convolutions of a ResNeXt stage's entry
 * (third_party/midas_blocks.py:35-50 via torchvision's Bottleneck.conv2 / downsample). */
int dvd_subsample2_fwd(const void* x, void* y, int f16, long long planes, int H, int W, dvd_stream_t stream);
int dvd_subsample2_bwd(const void* gy, void* gx, int f16, long long planes, int H, int W, dvd_stream_t stream);
/* nn.AvgPool2d(k, stride, pad) with PyTorch's defaults (count_include_pad, floor mode) on [planes, H, W] tensors, fp32 or fp16
 * (f16 != 0): AvgPool2d(2) of the hourglass (third_party/hourglass.py:60-158), AvgPool2d(3, 2, 1) of FCNUnet
 * (networks/FCNUnet.py:64); ATen's arithmetic (fp32 window sum / (k * k); backward: gy / (k * k) summed over the windows that
 * hold the pixel).  y / gy are [planes, (H + 2 pad - k) / stride + 1, (W + 2 pad - k) / stride + 1]. */
int dvd_avgpool_fwd(const void* x, void* y, int f16, long long planes, int H, int W, int k, int stride, int pad,
                    dvd_stream_t stream);
int dvd_avgpool_bwd(const void* gy, void* gx, int f16, long long planes, int H, int W, int k, int stride, int pad,
                    dvd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DVD_HIP_H */
