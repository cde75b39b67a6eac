// Fused scene-flow field MLP for gfx950 (MI355X): forward, backward-dX chain, backward-dW, on the 16-bit
// matrix cores with fp32-class accuracy.
//
// What it replaces (reference, /root/reference):
//   networks/sceneflow_field.py:43-53   SceneFlowFieldNet.forward
//   networks/blocks.py:19-34            PeriodicEmbed (64 separate sin/cos launches per eval)
//   networks/blocks.py:50-102           Conv2dBlock = 1x1 conv + LeakyReLU(0.2), x6
//   models/scene_flow_motion_field.py:346-367  forward_sf_net (/= sf_mag_div) and one Euler step
// and the autograd backward of all of it.  The unfused reference saves 11 168 B per
// pixel-evaluation for backward and launches ~80 kernels per evaluation.
//
// Arithmetic (csrc/dvd_split.h, as in csrc/xconv.hip): every fp32 operand, scaled by a power of two, is split into two
// fp16 terms (22 significant bits) and a product is three partial products on v_mfma_f32_32x32x16_f16 with fp32
// accumulation, unscaled exactly afterwards.  The scales are dynamic and never leave the device: weights per layer (max|W_l|,
// found by the pack step), activations / gradients PER TILE AND LAYER (the epilogue that produces a tile's 256 x 64 values
// exchanges the waves' maxima through LDS across the barrier it needs anyway), the weight-gradient kernel per launch (the
// forward / dX kernels fold their tile maxima into per-layer scalars at the end of the stash).  Earlier generations (fp32
// MFMAs: 121 / 94 / 91 TF/s forward / dX / dW; three bf16 terms and six products: 164 / 143 / 165) are in the history.
//
// 593 408 FLOP per pixel forward, the same again for dX and for dW.
//
// Forward and dX: a workgroup owns a tile of 64 pixels.  Round 5: 256 threads = 4 waves, two workgroups per CU (rounds 2-4:
// one of 8 waves; dvd_sf_mlp_select switches, the results are bit-identical).
//   * Activations live in LDS already split: X[term][k-octet][pixel] cells of 8 fp16 (64 KB).  A lane's B fragment
//     of one K step (16 channels) is one ds_read_b128 per term; 32 consecutive pixels = 32 consecutive cells.
//   * Weights never touch LDS: dvd_sf_mlp_pack writes them split and in fragment order (both orientations,
//     2.3 MB, resident in each XCD's 4 MB L2); a lane's A fragment is one 16-byte global load per term.
//   * Wave w computes output channels [64w, 64w+64) for all 64 pixels (2x2 tiles of 32x32), layer after layer in
//     place: 12 MFMAs per K step against 4 global + 4 LDS fragment loads (8 waves: 6 against 2 + 4 -- every B fragment
//     now feeds two row tiles, half the LDS reads per MFMA).
//   * The epilogue (bias from LDS, LeakyReLU, split, LDS write) also streams the fp32 activation to the stash in the T8
//     layout below -- pixel runs per channel, the K-contiguous operand layout of the dW GEMM -- and one SIGN BIT per unit:
//     the dX chain needs LeakyReLU' only, so it reads 4 bytes per lane, row tile and layer instead of the activations.
//   * The 256 -> 3 output layer is folded into the last epilogue (per-lane partial dot products of the fp32
//     values, reduced over the row tiles through LDS in fixed order); its weight gradient in the dX kernel is a
//     [3 x 64] x [64 x 256] product per tile, read from the h4 blocks 16 bytes per lane and accumulated in LDS.
// dW: dW_l = G_l H_{l-1}^T contracts over PIXELS.  A 512-thread workgroup owns the whole 256 x 256 matrix of one
//   layer for a run of tiles (each operand is read from HBM exactly once): per 16-pixel chunk 512 channel rows are
//   loaded as fp32, split and written to a double-buffered LDS stage while the MFMAs of the previous chunk run
//   (one barrier per chunk); wave (wr, wc) keeps rows [64wr,+64) x columns [128wc,+128) in 128 accumulators.
//   Per-workgroup partial matrices go to a workspace at the end of `gstash` and are summed in fixed order by a
//   second kernel: the weight gradients are bitwise reproducible (the first generation used float atomics).

#include "dvd_split.h"

// waves per SIMD the forward / dX kernels are compiled for: 2 = <= 256 VGPRs (forward 212, dX 225 with four waves per
// workgroup).  4 (<= 128 VGPRs) with 8-wave workgroups, two per CU: 7 / 25 spilled registers and no faster (round 5)
#ifndef DVD_MLP_FWD_OCC
#define DVD_MLP_FWD_OCC 2
#endif
#ifndef DVD_MLP_DX_PRE
#define DVD_MLP_DX_PRE 0        // 1: backward-data requests the next layer's first weight fragments in front of the stash stores
                                // (round 6, VERDICT round 5 item 4: measured 11.0 ms per call with it, 10.9 without -- the in-order
                                //  wait behind the stash stores is not what the kernel's waiting consists of; kept as an A/B knob)
#endif
#ifndef DVD_MLP_DX_OCC
#define DVD_MLP_DX_OCC 2
#endif
// waves per workgroup the forward / dX kernels start with (dvd_sf_mlp_select changes it at run time): see mlp_fwd_kernel
#ifndef DVD_MLP_NW
#define DVD_MLP_NW 4
#endif

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: arrays of it stay in registers
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTM = 64;        // pixels per tile
constexpr int kWidth = 256;    // hidden width (scene_flow_motion_field.py:107)
constexpr int kHidden = 5;     // layers with LeakyReLU: convs.0 .. convs.4
constexpr int kNT = 512;       // threads per workgroup (forward, dX, dW)
constexpr float kSlope = 0.2f;
constexpr int kTermStride = 32 * kTM * 16;   // bytes between the split terms of X: [32 k-octets][64 pixels][16 B]
constexpr int kXBytes = 2 * kTermStride;     // 65 536

struct Geometry {
  int n_freq_xyz, n_freq_t, time_dependent;
  int s16;               // stash_f16 of the descriptor
  int c_in, c_in16;      // input channels, padded to the K step
  int ks0, rt0;          // K steps of the input layer, row tiles of W_0^T
  int t_base, xyz_base;  // channel of t / of x in the input layer
};

static Geometry make_geometry(const dvd_mlp_desc* d) {
  Geometry g;
  g.n_freq_xyz = d->n_freq_xyz;
  g.n_freq_t = d->time_dependent ? d->n_freq_t : 0;
  g.time_dependent = d->time_dependent;
  g.s16 = d->stash_f16 ? 1 : 0;
  const int ct = d->time_dependent ? 1 + 2 * d->n_freq_t : 0;
  g.c_in = ct + 3 + 6 * d->n_freq_xyz;
  g.c_in16 = (g.c_in + 15) & ~15;
  g.ks0 = g.c_in16 / 16;
  g.rt0 = (g.c_in + 31) / 32;
  g.t_base = 0;
  g.xyz_base = ct;
  return g;
}

// ---- packed weight buffer ----------------------------------------------------------------
// fragments: [(row tile * nk + K step) * 2 + term][64 lanes] x 16 bytes; lane l holds A[32 rt + (l&31)][16 kc + 8 (l>>5) .. +7]
// of W_l * pow2_scale(max|W_l|)
//   forward  layer l: A[m][k] = W_l[out = m][in = k]    8 row tiles, nk = K_l / 16
//   backward layer l: A[m][k] = W_l[out = k][in = m]    ceil(K_l / 32) row tiles, 16 K steps
// then fp32: W_5 [3][256], biases, max|W_l| of the five hidden layers (8 floats).
struct PackLayout {
  size_t fwd[kHidden], bwd[kHidden];   // offsets in 16-byte units
  int nk[kHidden], rtb[kHidden];
  size_t w5, bias[6], wamax;           // offsets in floats
  size_t total_bytes;
};

static PackLayout make_pack_layout(const Geometry& g) {
  PackLayout L;
  size_t off = 0;
  for (int l = 0; l < kHidden; ++l) {
    L.nk[l] = l == 0 ? g.ks0 : kWidth / 16;
    L.rtb[l] = l == 0 ? g.rt0 : kWidth / 32;
    L.fwd[l] = off;
    off += (size_t)8 * L.nk[l] * 128;
    L.bwd[l] = off;
    off += (size_t)L.rtb[l] * 16 * 128;
  }
  size_t f = off * 4;
  L.w5 = f;
  f += 3 * kWidth;
  for (int l = 0; l < 6; ++l) {
    L.bias[l] = f;
    f += l < 5 ? kWidth : 4;
  }
  L.wamax = f;
  f += 8;
  L.total_bytes = f * 4;
  return L;
}

struct PackArgs {
  const float* W[6];
  const float* b[6];
  void* out;
  PackLayout L;
  int c_in;
};

// max|W_l| of the hidden layers -> packed tail (zeroed by a memset node before)
__global__ __launch_bounds__(256) void mlp_wamax_kernel(const PackArgs a) {
  const int l = blockIdx.y;
  const int n = kWidth * (l == 0 ? a.c_in : kWidth);
  float m = 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, fabsf(a.W[l][i]));
  wave_amax_to(m, static_cast<float*>(a.out) + a.L.wamax + l);
}

__global__ __launch_bounds__(256) void mlp_pack_kernel(const PackArgs a) {
  const int job = blockIdx.y;   // 0..4 forward layer, 5..9 backward layer, 10 fp32 tail
  const int gid = blockIdx.x * 256 + threadIdx.x;
  float* outf = static_cast<float*>(a.out);
  if (job == 10) {
    if (gid < 3 * kWidth) outf[a.L.w5 + gid] = a.W[5][gid];
    if (gid < 4) outf[a.L.bias[5] + gid] = gid < 3 ? a.b[5][gid] : 0.0f;
    for (int l = 0; l < kHidden; ++l)
      if (gid < kWidth) outf[a.L.bias[l] + gid] = a.b[l][gid];
    return;
  }
  const bool bwd = job >= kHidden;
  const int l = bwd ? job - kHidden : job;
  const int Kin = l == 0 ? a.c_in : kWidth;          // true fan-in of layer l (row length of W_l)
  const int nk = bwd ? 16 : a.L.nk[l], nrt = bwd ? a.L.rtb[l] : 8;
  if (gid >= nrt * nk * 64) return;
  const int lane = gid & 63, f = gid >> 6, kc = f % nk, rt = f / nk;
  const int m = 32 * rt + (lane & 31), k0 = 16 * kc + 8 * (lane >> 5);
  const float* W = a.W[l];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float val = 0.0f;
    if (!bwd) {
      if (k < Kin) val = W[(size_t)m * Kin + k];       // m < 256 always
    } else {
      if (m < Kin) val = W[(size_t)k * Kin + m];       // k < 256 always
    }
    v[e] = val;
  }
  const float sw = pow2_scale(outf[a.L.wamax + l]);
  unsigned hw[4], lw[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split_pair_f16(v[2 * e] * sw, v[2 * e + 1] * sw, hw[e], lw[e]);
  const u32x4 h = {hw[0], hw[1], hw[2], hw[3]}, lo = {lw[0], lw[1], lw[2], lw[3]};
  u32x4* dst = static_cast<u32x4*>(a.out) + (bwd ? a.L.bwd[l] : a.L.fwd[l]) + (size_t)f * 128 + lane;
  dst[0] = h;
  dst[64] = lo;
}

// ---- stash layouts (floats per tile) -------------------------------------------------------
// stash : embedding [c_in16][64], h_0 .. h_4 [256][64] each, sign words [5][512] (one bit per unit of the lane's
//         32 outputs of that layer, bit = 16 ct + r)
// gstash: pre-activation gradients G_0 .. G_4 [256][64] each; after all tiles: the dW workspace
// s16 (round 4, dvd_mlp_desc.stash_f16): h_0 .. h_4 are stored as _Float16 -- half the floats per layer, everything else as
// before.  The stash is what the WEIGHT-GRADIENT kernel contracts against (and h_4 what the dX kernel takes dW_5 from); the
// forward arithmetic keeps its two-term activations in LDS either way, so losses and dX are untouched by the stash's storage.
__host__ __device__ inline size_t stash_h_floats(bool s16) { return (size_t)kWidth * kTM / (s16 ? 2 : 1); }   // one h_l, in floats
__host__ __device__ inline size_t stash_floats_per_tile(int c_in16, bool s16) {
  return (size_t)c_in16 * kTM + kHidden * stash_h_floats(s16) + (size_t)kHidden * kNT;
}
__host__ __device__ inline size_t stash_h_off(int c_in16, int l, bool s16) { return (size_t)c_in16 * kTM + l * stash_h_floats(s16); }   // h_l
__host__ __device__ inline size_t stash_sign_off(int c_in16, int l, bool s16) {
  return (size_t)c_in16 * kTM + kHidden * stash_h_floats(s16) + (size_t)l * kNT;
}
__host__ __device__ inline size_t gstash_floats_per_tile() { return (size_t)kHidden * kWidth * kTM; }
// After the tiles of a stash: 16 floats of per-layer maxima over ALL tiles (atomic max by the forward / dX kernels, read by
// the weight-gradient kernel for its per-launch operand scales): [0] embedding, [1 + l] h_l, [8 + l] G_l.
constexpr int kStashTail = 16;
constexpr int kDwSlices = 51;                                  // workgroups per layer (5 x 51 = 255 of 256 CUs)
constexpr size_t kDwPartial = (size_t)kWidth * kWidth + kWidth;   // floats per workgroup: dW block + row sums

// ---- the GEMM core of forward and dX ---------------------------------------------------------
// acc[ct] += A (this wave's 32 rows, fragments streamed from global / L2) x B (X in LDS, column tile ct) over nk K steps.
// Ap = fragment base of the wave's row tile + lane; Xl = LDS base of X + (lane>>5) * 1024 + (lane&31) * 16.
template <int RT>
__device__ __forceinline__ void load_frags(const u32x4* __restrict__ Ap, int rt_stride, const unsigned char* Xl, int kc,
                                           u32x4 (&A)[RT][2], u32x4 (&B)[2][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < RT; ++r) A[r][t] = Ap[(size_t)r * rt_stride + (size_t)kc * 128 + t * 64];
    B[0][t] = *reinterpret_cast<const u32x4*>(Xl + t * kTermStride + kc * 2048);
    B[1][t] = *reinterpret_cast<const u32x4*>(Xl + t * kTermStride + kc * 2048 + 512);
  }
}

template <int RT>
__device__ __forceinline__ void mfma_step(const u32x4 (&A)[RT][2], const u32x4 (&B)[2][2], f32x16 (*acc)[2]) {
  // per accumulator the order of the three partial products is the same for every RT (smallest first): RT = 1 and RT = 2
  // give bit-identical sums
#define DVD_MLP_TERM(SA, SB)                                                                                    \
  _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                              \
    acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[r][SA]),                     \
                                                       __builtin_bit_cast(f16x8, B[0][SB]), acc[r][0], 0, 0, 0); \
    acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[r][SA]),                     \
                                                       __builtin_bit_cast(f16x8, B[1][SB]), acc[r][1], 0, 0, 0); \
  }
  DVD_MLP_TERM(1, 0)   // smallest partial products first
  DVD_MLP_TERM(0, 1)
  DVD_MLP_TERM(0, 0)
#undef DVD_MLP_TERM
}

// RT row tiles of 32 output channels per wave (RT = 1: eight waves per 64-pixel tile; RT = 2: four, every B fragment feeds
// two row tiles).  Ap = fragment base of the wave's FIRST row tile + lane, rt_stride = fragments (u32x4) per row tile.
template <int RT>
__device__ __forceinline__ void gemm_rows(const u32x4* __restrict__ Ap, int rt_stride, int nk, const unsigned char* Xl,
                                          f32x16 (*acc)[2]) {
  // two fragment sets, ping-pong: the loads of step kc+1 are issued before the MFMAs of step kc.  The scheduling
  // barriers keep them there (the scheduler otherwise sinks the loads to their first use and every K step waits a
  // full L2 round trip); steps past the end re-read the last one (no conditional loads).
  u32x4 A0[RT][2], B0[2][2], A1[RT][2], B1[2][2];
  load_frags<RT>(Ap, rt_stride, Xl, 0, A0, B0);
  int kc = 0;
#pragma unroll 1
  for (; kc + 1 < nk; kc += 2) {
    load_frags<RT>(Ap, rt_stride, Xl, kc + 1, A1, B1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step<RT>(A0, B0, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_frags<RT>(Ap, rt_stride, Xl, kc + 2 < nk ? kc + 2 : nk - 1, A0, B0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step<RT>(A1, B1, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kc < nk) mfma_step<RT>(A0, B0, acc);   // odd nk: set 0 already holds the last step
}

// The same with the weight fragments of K step 0 already REQUESTED by the caller (round 6: the backward-data kernel requests
// the next layer's first fragments in front of a layer's gradient-stash stores -- the one memory counter retires in order, so
// fragments requested behind the stores wait for every one of them: an exposed HBM write round trip per layer and tile).
template <int RT>
__device__ __forceinline__ void load_a_frags(const u32x4* __restrict__ Ap, int rt_stride, int kc, u32x4 (&A)[RT][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) A[r][t] = Ap[(size_t)r * rt_stride + (size_t)kc * 128 + t * 64];
}
template <int RT>
__device__ __forceinline__ void gemm_rows_pre(const u32x4* __restrict__ Ap, int rt_stride, int nk, const unsigned char* Xl,
                                              f32x16 (*acc)[2], const u32x4 (&Apre)[RT][2]) {
  u32x4 A0[RT][2], B0[2][2], A1[RT][2], B1[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < RT; ++r) A0[r][t] = Apre[r][t];
    B0[0][t] = *reinterpret_cast<const u32x4*>(Xl + t * kTermStride);
    B0[1][t] = *reinterpret_cast<const u32x4*>(Xl + t * kTermStride + 512);
  }
  int kc = 0;
#pragma unroll 1
  for (; kc + 1 < nk; kc += 2) {
    load_frags<RT>(Ap, rt_stride, Xl, kc + 1, A1, B1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step<RT>(A0, B0, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_frags<RT>(Ap, rt_stride, Xl, kc + 2 < nk ? kc + 2 : nk - 1, A0, B0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step<RT>(A1, B1, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kc < nk) mfma_step<RT>(A0, B0, acc);
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RT][2]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.0f;
}

// Four consecutive channels (n4 .. n4+3, the half `hh` of k-octet ko) of pixel m, times the tile's scale -> the two split
// terms in X.
__device__ __forceinline__ void store_split4(unsigned char* X, int ko, int m, int hh, float s, float v0, float v1, float v2,
                                             float v3) {
  unsigned h0, l0, h1, l1;
  split_pair_f16(v0 * s, v1 * s, h0, l0);
  split_pair_f16(v2 * s, v3 * s, h1, l1);
  unsigned char* dst = X + (ko * kTM + m) * 16 + hh * 8;
  *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
  *reinterpret_cast<u32x2*>(dst + kTermStride) = (u32x2){l0, l1};
}

// DPP exchanges inside a quad of lanes
__device__ __forceinline__ float quad_swap1(float v) {     // from lane ^ 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float quad_swap2(float v) {     // from lane ^ 2
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
}
// ---- layout of a layer's [256 channels][64 pixels] block in the stashes ("T8") -------------------------------------------
// Blocks of 8 channels; inside a block four chunks of 16 pixel POSITIONS; inside a chunk the 8 channels' runs of 16 positions:
//     element (channel n, pixel m) at (n >> 3) * 512 + (pos >> 4) * 128 + (n & 7) * 16 + (pos & 15),  pos = (m & 15) * 4 + (m >> 4).
// Why: the accumulator layout gives a lane ONE pixel (column lane & 31 of the column tile ct) of four consecutive channels,
// the weight-gradient GEMM contracts over pixels and wants runs of pixels per channel.  One v_permlane16_swap between the
// registers of channels 2 e1 and 2 e1 + 1 moves pixel bit 4 from the lane index into the register index (and channel bit 0
// the other way): with the two column tiles a lane then holds the pixels jl, jl + 16, jl + 32, jl + 48 (jl = lane & 15) of ONE
// channel -- the positions 4 jl .. 4 jl + 3 -- in four registers = one 16-byte store (fp16: 8), 4 exchanges per 32 values.
// (Rounds 3-4 transposed 4 x 4 blocks inside each quad of lanes with DPP moves and selects: 36 instructions per 32 values,
// a third of all vector instructions of the forward's epilogue.)  Which pixels share a run does not matter to the
// contraction as long as both operands agree: the gradient stash, the activation stash AND the embedding rows
// ([channel][64 positions]) use the same pixel -> position map.  The weight-gradient kernel reads 16-position chunks: 512
// contiguous bytes per 8-channel block.
__host__ __device__ inline int t8_pos(int m) { return (m & 15) * 4 + (m >> 4); }
__host__ __device__ inline size_t t8_off(int n, int pos) { return (size_t)(n >> 3) * 512 + (pos >> 4) * 128 + (n & 7) * 16 + (pos & 15); }

// x[e] / y[e] = channels n8 + 4 hh + e (e = 0..3; n8 a multiple of 8, hh = lane >> 5) of the pixels (lane & 31) / 32 + (lane & 31)
// -- an accumulator register quad of the two column tiles.  blk = the layer block + t8_off(n8, 0).  On return o[e1][k] =
// channel n8 + 4 hh + 2 e1 + ((lane >> 4) & 1), pixel (lane & 15) + 16 k.
__device__ __forceinline__ void t8_exchange(const float (&x)[4], const float (&y)[4], float (&o)[2][4]) {
#pragma unroll
  for (int e1 = 0; e1 < 2; ++e1) {
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[2 * e1]), __float_as_uint(x[2 * e1 + 1]), false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[2 * e1]), __float_as_uint(y[2 * e1 + 1]), false, false);
    o[e1][0] = __uint_as_float(s0[0]);
    o[e1][1] = __uint_as_float(s0[1]);
    o[e1][2] = __uint_as_float(s1[0]);
    o[e1][3] = __uint_as_float(s1[1]);
  }
}
// the lane's element offset inside an 8-channel block for e1 = 0 (e1 = 1: + 32)
__device__ __forceinline__ int t8_lane_off(int lane) {
  const int jl = lane & 15;
  return (jl >> 2) * 128 + (4 * (lane >> 5) + ((lane >> 4) & 1)) * 16 + (jl & 3) * 4;
}
__device__ __forceinline__ void store_t8(float* blk, int lane, const float (&x)[4], const float (&y)[4]) {
  float o[2][4];
  t8_exchange(x, y, o);
  float* dst = blk + t8_lane_off(lane);
  *reinterpret_cast<float4*>(dst) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
  *reinterpret_cast<float4*>(dst + 32) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
}
__device__ __forceinline__ void store_t8(_Float16* blk, int lane, const float (&x)[4], const float (&y)[4]) {
  float o[2][4];
  t8_exchange(x, y, o);
  _Float16* dst = blk + t8_lane_off(lane);
#pragma unroll
  for (int e1 = 0; e1 < 2; ++e1) {
    const f16x2 a0 = {(_Float16)o[e1][0], (_Float16)o[e1][1]}, a1 = {(_Float16)o[e1][2], (_Float16)o[e1][3]};
    *reinterpret_cast<u32x2*>(dst + 32 * e1) = (u32x2){__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1)};
  }
}

// max over the wave's lanes, in every lane
__device__ __forceinline__ float wave_max_all(float m) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  return m;
}
// tile maximum from the waves' maxima (eight LDS slots, the unused ones of a 4-wave workgroup stay 0; written before the
// barrier the caller has just passed)
__device__ __forceinline__ float tile_max(const float* tmx) {
  const float4 a = *reinterpret_cast<const float4*>(tmx), b = *reinterpret_cast<const float4*>(tmx + 4);
  return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
}
__device__ __forceinline__ void fold_amax(float* dst, float m) {      // (read first: see wave_amax_to)
  unsigned* p = reinterpret_cast<unsigned*>(dst);
  if (m > 0.0f && __float_as_uint(m) > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, __float_as_uint(m));
}

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, kSlope * v); }

// ==========================================================================================
// forward
struct FwdArgs {
  const void* packed;
  const float* p;
  const float* t;
  const float* freqs_xyz;
  const float* freqs_t;
  float* sf_out;
  float* p_next;
  float* acc;
  float* stash;
  float* monitor;     // fp16 stash: max |h| of the hidden activations is folded in here too (the step's overflow guard), or null
  PackLayout L;
  Geometry g;
  long long n_pix;
  int pix_per_img, n_tiles;
  float t_offset, out_scale;
};

// LDS: X (at the end of a tile: red [8 row tiles][3][64]) | psm [4][64] | w5 [3][256] + bias5 [4] | tmx [8] | bias [5][256]
constexpr size_t kFwdLds = (size_t)kXBytes + 4 * kTM * 4 + (3 * kWidth + 4) * 4 + 8 * 4 + kHidden * kWidth * 4;

// Input embedding of one tile: split into X, fp32 to the stash.  psm = [4][64] floats: x, y, z, t of the pixels.
// `sx` = the tile's operand scale (pow2_scale of the largest |input| of the tile; sin / cos are bounded by 1).
template <bool STASH, int NW>
__device__ __forceinline__ void build_embedding(const Geometry& g, const float* __restrict__ fx, const float* __restrict__ ft,
                                                const float* psm, unsigned char* X, float* st_emb, float sx) {
  const int m = threadIdx.x & 63, part = threadIdx.x >> 6;   // NW parts
  auto put = [&](int ch, float v) {
    unsigned h, l;
    split_pair_f16(v * sx, 0.0f, h, l);
    unsigned char* dst = X + ((ch >> 3) * kTM + m) * 16 + (ch & 7) * 2;
    *reinterpret_cast<unsigned short*>(dst) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(dst + kTermStride) = (unsigned short)l;
    if (STASH) st_emb[(size_t)ch * kTM + t8_pos(m)] = v;      // rows of 64 pixel POSITIONS, like the T8 blocks
  };
  const float x0 = psm[m], x1 = psm[kTM + m], x2 = psm[2 * kTM + m], tt = psm[3 * kTM + m];
  if (part == 0) {
    if (g.time_dependent) put(g.t_base, tt);
    put(g.xyz_base + 0, x0);
    put(g.xyz_base + 1, x1);
    put(g.xyz_base + 2, x2);
    for (int ch = g.c_in; ch < g.c_in16; ++ch) put(ch, 0.0f);
  }
  const int nt = g.n_freq_t, nx = g.n_freq_xyz;
  const int items = nt + 3 * nx;
  for (int e = part; e < items; e += NW) {
    float arg;
    int ch_cos, ch_sin;
    if (e < nt) {
      arg = ft[e] * tt;
      ch_cos = g.t_base + 1 + e;
      ch_sin = g.t_base + 1 + nt + e;
    } else {
      const int q = e - nt, i = q / 3, c = q - 3 * i;
      arg = fx[i] * (c == 0 ? x0 : (c == 1 ? x1 : x2));
      ch_cos = g.xyz_base + 3 + 3 * i + c;
      ch_sin = g.xyz_base + 3 + 3 * nx + 3 * i + c;
    }
    float sv, cv;
    sincosf(arg, &sv, &cv);  // accurate ocml path (arguments reach |17 x|)
    put(ch_cos, cv);
    put(ch_sin, sv);
  }
}

// NW = waves per workgroup: 8 (one 512-thread workgroup per CU, wave w = row tile w) or 4 (two 256-thread workgroups per CU,
// wave w = row tiles 2w and 2w + 1).  Both compute every value with the same operations in the same order: bit-identical
// outputs, stash and maxima (tests/test_02_sf_mlp_gpu.py).
template <bool STASH, bool S16, int NW>
__global__ __launch_bounds__(64 * NW, DVD_MLP_FWD_OCC) void mlp_fwd_kernel(const FwdArgs a) {
  constexpr int RT = 8 / NW, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* X = smem;
  float* psm = reinterpret_cast<float*>(smem + kXBytes);   // [4][64]
  float* w5 = psm + 4 * kTM;                               // [3][256] + bias5[4]
  float* tmx = w5 + 3 * kWidth + 4;                        // [8] (NW used): maxima of the values a layer's epilogue produced
  float* bl = tmx + 8;                                     // [5][256]: biases of the hidden layers
  float* red = reinterpret_cast<float*>(smem);             // [8 row tiles][3][64]: the output layer's partial sums, in X's place
                                                           // (written after the last layer's GEMM has read X, summed before
                                                           // the next tile's first barrier)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const float* pf = static_cast<const float*>(a.packed);
  const u32x4* P4 = static_cast<const u32x4*>(a.packed);
  for (int i = tid; i < 3 * kWidth + 4; i += NT) w5[i] = pf[i < 3 * kWidth ? a.L.w5 + i : a.L.bias[5] + (i - 3 * kWidth)];
  if (tid < 8) tmx[tid] = 0.0f;
  for (int i = tid; i < kHidden * kWidth; i += NT) bl[i] = pf[a.L.bias[i / kWidth] + (i % kWidth)];
  const unsigned char* Xl = X + hh * 1024 + j * 16;
  float* tail = STASH ? a.stash + (size_t)a.n_tiles * stash_floats_per_tile(a.g.c_in16, S16) : nullptr;   // per-layer maxima

  // pixel inputs of a tile: thread (c = tid >> 6, m = lane), requested one tile ahead (an HBM round trip per tile otherwise)
  auto pixel_input = [&](int tile) {
    float v = 0.0f;
    if (tid < 4 * kTM && tile < a.n_tiles) {
      const int c = tid >> 6;
      const long long n = (long long)tile * kTM + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        if (c < 3)
          v = a.p[(b * 3 + c) * a.pix_per_img + hw];
        else
          v = a.t ? a.t[n] + a.t_offset : 0.0f;
      }
    }
    return v;
  };
  float v_in = pixel_input(blockIdx.x);

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long n0 = (long long)tile * kTM;
    __syncthreads();  // previous tile's readers are done with X / psm / red
    if (tid < 4 * kTM) psm[(tid >> 6) * kTM + lane] = v_in;
    v_in = pixel_input(tile + gridDim.x);
    __syncthreads();
    float* st = STASH ? a.stash + (size_t)tile * stash_floats_per_tile(a.g.c_in16, S16) : nullptr;
    // operand scale of the embedding: the largest input magnitude of the tile (every wave sees all 64 pixels in its lanes)
    const float emax = wave_max_all(fmaxf(fmaxf(fabsf(psm[lane]), fabsf(psm[kTM + lane])),
                                          fmaxf(fmaxf(fabsf(psm[2 * kTM + lane]), fabsf(psm[3 * kTM + lane])), 1.0f)));
    float sx = pow2_scale(emax);                       // scale of what X currently holds (uniform over the workgroup)
    if (STASH && tid == 0) fold_amax(tail + 0, emax);
    build_embedding<STASH, NW>(a.g, a.freqs_xyz, a.freqs_t, psm, X, st, sx);
    __syncthreads();

#pragma unroll 1
    for (int l = 0; l < kHidden; ++l) {
      f32x16 acc[RT][2];
      zero_acc<RT>(acc);
      const int nk = a.L.nk[l];
      gemm_rows<RT>(P4 + a.L.fwd[l] + (size_t)(RT * w) * nk * 128 + lane, nk * 128, nk, Xl, acc);
      // activations in place of the accumulators (exact unscaling: a power of two), their maximum over the wave -> LDS
      const float unscale = 1.0f / (sx * pow2_scale(pf[a.L.wamax + l]));
      float vmax = 0.0f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(bl + l * kWidth + 32 * (RT * w + r) + 8 * q + 4 * hh);
            acc[r][ct][4 * q + 0] = lrelu(__builtin_fmaf(acc[r][ct][4 * q + 0], unscale, bv.x));
            acc[r][ct][4 * q + 1] = lrelu(__builtin_fmaf(acc[r][ct][4 * q + 1], unscale, bv.y));
            acc[r][ct][4 * q + 2] = lrelu(__builtin_fmaf(acc[r][ct][4 * q + 2], unscale, bv.z));
            acc[r][ct][4 * q + 3] = lrelu(__builtin_fmaf(acc[r][ct][4 * q + 3], unscale, bv.w));
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(acc[r][ct][4 * q + 0]), fabsf(acc[r][ct][4 * q + 1]))),
                         fmaxf(fabsf(acc[r][ct][4 * q + 2]), fabsf(acc[r][ct][4 * q + 3])));
          }
      vmax = wave_max_all(vmax);
      if (lane == 0) tmx[w] = vmax;
      __syncthreads();  // every wave has finished reading this layer's input; the waves' maxima are visible
      const float hmax = tile_max(tmx);
      sx = pow2_scale(hmax);                           // scale of the next layer's input
      if (STASH && tid == 0) {
        fold_amax(tail + 1 + l, hmax);
        // (an fp16 stash stores h as _Float16: beyond 65504 it holds Inf and the weight gradients contracted against it are
        //  not finite -- the step's forward monitor must see it, csrc/a16.hip state[6]; NaN counts as +Inf)
        if (S16 && a.monitor) fold_amax(a.monitor, hmax == hmax ? hmax : __builtin_inff());
      }
      float* sh = STASH ? st + stash_h_off(a.g.c_in16, l, S16) : nullptr;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int rt = RT * w + r;                     // the row tile: output channels [32 rt, 32 rt + 32)
        unsigned sw = 0;
        float po[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n4 = 32 * rt + 8 * q + 4 * hh;  // first of 4 consecutive output channels
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const float v0 = acc[r][ct][4 * q + 0], v1 = acc[r][ct][4 * q + 1], v2 = acc[r][ct][4 * q + 2], v3 = acc[r][ct][4 * q + 3];
            const int m = 32 * ct + j;
            if (l < kHidden - 1) store_split4(X, 4 * rt + q, m, hh, sx, v0, v1, v2, v3);   // the next layer's input
            if (STASH) {
              const int b0 = 16 * ct + 4 * q;
              sw |= (v0 > 0.f ? 1u : 0u) << b0 | (v1 > 0.f ? 1u : 0u) << (b0 + 1) | (v2 > 0.f ? 1u : 0u) << (b0 + 2) |
                    (v3 > 0.f ? 1u : 0u) << (b0 + 3);
            }
            if (l == kHidden - 1) {   // 256 -> 3 output layer: this lane's share of the dot products
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float4 ww = *reinterpret_cast<const float4*>(w5 + c * kWidth + n4);
                po[c][ct] = __builtin_fmaf(v0, ww.x, __builtin_fmaf(v1, ww.y, __builtin_fmaf(v2, ww.z, __builtin_fmaf(v3, ww.w, po[c][ct]))));
              }
            }
          }
          if (STASH) {
            const float x[4] = {acc[r][0][4 * q + 0], acc[r][0][4 * q + 1], acc[r][0][4 * q + 2], acc[r][0][4 * q + 3]};
            const float y[4] = {acc[r][1][4 * q + 0], acc[r][1][4 * q + 1], acc[r][1][4 * q + 2], acc[r][1][4 * q + 3]};
            if constexpr (S16) store_t8(reinterpret_cast<_Float16*>(sh) + t8_off(32 * rt + 8 * q, 0), lane, x, y);
            else store_t8(sh + t8_off(32 * rt + 8 * q, 0), lane, x, y);
          }
        }
        // one sign word per (row tile, lane): word 64 rt + lane, the same layout for every NW
        if (STASH) reinterpret_cast<unsigned*>(st + stash_sign_off(a.g.c_in16, l, S16))[64 * rt + lane] = sw;
        if (l == kHidden - 1) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              const float v = po[c][ct] + __shfl_xor(po[c][ct], 32, 64);
              if (hh == 0) red[(rt * 3 + c) * kTM + 32 * ct + j] = v;
            }
        }
      }
      __syncthreads();
    }
    if (tid < 3 * kTM) {  // thread (c, m): sum the 8 row tiles' shares in fixed order
      const int c = tid >> 6;
      float s = 0.0f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) s += red[(ww * 3 + c) * kTM + lane];
      const float sf = (s + w5[3 * kWidth + c]) * a.out_scale;
      const long long n = n0 + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + c) * a.pix_per_img + hw);
        if (a.sf_out) a.sf_out[o] = sf;
        if (a.p_next) a.p_next[o] = psm[c * kTM + lane] + sf;
        if (a.acc) a.acc[o] += sf;
      }
    }
  }
}

// ==========================================================================================
// backward, dX chain
struct BwdArgs {
  const void* packed;
  const float* stash;
  float* gstash;
  const float* g_out1;
  const float* g_out2;
  const float* scale_ptr;
  const float* g_p_add;
  const float* freqs_xyz;
  float* g_p;
  float* gW5;
  float* gb5;
  PackLayout L;
  Geometry g;
  long long n_pix;
  int pix_per_img, n_tiles;
  float out_scale, gscale;
};

// LDS: X (gradient tile, split; at the end fp32 g_in [<= 256][64]) | gz5 [4][64] | gz5p [4][64] | w5 [3][256] | tmx [8] | dw5s [3][256]
constexpr size_t kBwdLds = (size_t)kXBytes + 2 * 4 * kTM * 4 + 3 * kWidth * 4 + 8 * 4 + 3 * kWidth * 4;

// NW as in the forward.  The input gradient and the gradient stash are bit-identical for NW = 4 and 8 (the last layer's
// parameter gradients are float atomics over workgroups in both).
template <bool S16, int NW>
__global__ __launch_bounds__(64 * NW, DVD_MLP_DX_OCC) void mlp_bwd_dx_kernel(const BwdArgs a) {
  constexpr int RT = 8 / NW, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* X = smem;
  float* Xf = reinterpret_cast<float*>(smem);
  float* gz5 = reinterpret_cast<float*>(smem + kXBytes);   // [4][64]: g of the 3 outputs (already * out_scale)
  float* gz5p = gz5 + 4 * kTM;                             // [4][64]: the same in pixel-POSITION order (t8_pos), for dW5
  float* w5 = gz5p + 4 * kTM;                              // [3][256]
  float* tmx = w5 + 3 * kWidth;                            // [8] (NW used)
  float* dw5s = tmx + 8;                                   // [3][256]: this workgroup's share of dW5
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const float* pf = static_cast<const float*>(a.packed);
  const u32x4* P4 = static_cast<const u32x4*>(a.packed);
  for (int i = tid; i < 3 * kWidth; i += NT) w5[i] = pf[a.L.w5 + i];
  if (tid < 8) tmx[tid] = 0.0f;
  // per-layer maxima of the pre-activation gradients over all tiles (for the weight-gradient kernel): stash tail [8 + l]
  float* gtail = const_cast<float*>(a.stash) + (size_t)a.n_tiles * stash_floats_per_tile(a.g.c_in16, S16) + 8;
  const float s1 = a.gscale * (a.scale_ptr ? a.scale_ptr[0] : 1.0f);
  const unsigned char* Xl = X + hh * 1024 + j * 16;
  float sx = 1.0f;                                         // operand scale of what X currently holds
  // Last layer's parameter gradients dW5[c][n] = sum_m g_z5[c][m] h4[n][m], accumulated over this workgroup's tiles in LDS
  // (dw5s).  Wave w owns the channels [kCW w, kCW w + kCW) and reads their h4 blocks (T8 layout: contiguous) 16 bytes per
  // lane and load: one channel, 4 (fp16: 8) consecutive pixel positions -- the same positions of g_z5 are one (two)
  // ds_read_b128 from gz5p.  The lanes of a channel are summed by DPP exchanges inside the quad, the rest meets in LDS
  // atomics.  (Until round 5 this was done in the accumulator mapping -- a pixel per lane, 48 accumulator registers per
  // row tile that lived across the whole kernel.)
  constexpr int kCW = kWidth / NW;                         // channels per wave
  constexpr int kNL = kCW / (S16 ? 8 : 4);                 // 16-byte loads per lane and tile
  for (int i = tid; i < 3 * kWidth; i += NT) dw5s[i] = 0.0f;
  float db5 = 0.0f;
  const size_t spt = stash_floats_per_tile(a.g.c_in16, S16), gpt = gstash_floats_per_tile();

  // g_z5[c][m] of a tile: thread (c = tid >> 6, m = lane), requested one tile ahead (an HBM round trip per tile otherwise)
  auto out_grad = [&](int tile) {
    float v = 0.0f;
    if (tid < 3 * kTM && tile < a.n_tiles) {
      const int c = tid >> 6;
      const long long n = (long long)tile * kTM + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + c) * a.pix_per_img + hw);
        v = s1 * a.g_out1[o];
        if (a.g_out2) v += a.g_out2[o];
        v *= a.out_scale;
      }
    }
    return v;
  };
  float v_in = out_grad(blockIdx.x);

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long n0 = (long long)tile * kTM;
    const float* st = a.stash + (size_t)tile * spt;
    float* gs = a.gstash + (size_t)tile * gpt;
    __syncthreads();
    // this tile's h4 blocks (for dW5) and its sign words: requested here, consumed after the g_z4 passes below / in the layers
    const unsigned char* hb = reinterpret_cast<const unsigned char*>(st + stash_h_off(a.g.c_in16, 4, S16)) +
                              (size_t)kCW * w * kTM * (S16 ? 2 : 4) + lane * 16;
    u32x4 raw[kNL];
#pragma unroll
    for (int i = 0; i < kNL; ++i) raw[i] = *reinterpret_cast<const u32x4*>(hb + i * 1024);
    // (the sign words of ALL five layers: requested inside the layer loop they would sit in front of that layer's first weight
    //  fragments in the one in-order memory counter -- an HBM round trip before every GEMM)
    unsigned swl[kHidden][RT];
#pragma unroll
    for (int l = 0; l < kHidden; ++l)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        swl[l][r] = reinterpret_cast<const unsigned*>(st + stash_sign_off(a.g.c_in16, l, S16))[64 * (RT * w + r) + lane];
    if (tid < 4 * kTM) {
      const int c = tid >> 6;
      gz5[c * kTM + lane] = v_in;
      gz5p[c * kTM + t8_pos(lane)] = v_in;
      db5 += v_in;
    }
    v_in = out_grad(tile + gridDim.x);
    __syncthreads();
    {  // layer 5 (256 -> 3): g_z4 = (W5^T g_z5) * LeakyReLU'(h4) -- in the epilogue mapping
      float* g4 = gs + (size_t)4 * kWidth * kTM;
      // g_z4 of (pixel m, channels n4 .. n4 + 3): three FMAs and the slope per value -- evaluated twice (first for the
      // tile maximum and the stash, then for the split store) rather than kept in 32 registers per row tile across the barrier
      auto gz4 = [&](int r, int ct, int q, float (&v)[4]) {
        const int m = 32 * ct + j, n4 = 32 * (RT * w + r) + 8 * q + 4 * hh;
        const float ga = gz5[m], gb = gz5[kTM + m], gc = gz5[2 * kTM + m];
        const float4 wa = *reinterpret_cast<const float4*>(w5 + n4);
        const float4 wb = *reinterpret_cast<const float4*>(w5 + kWidth + n4);
        const float4 wc = *reinterpret_cast<const float4*>(w5 + 2 * kWidth + n4);
        const float war[4] = {wa.x, wa.y, wa.z, wa.w}, wbr[4] = {wb.x, wb.y, wb.z, wb.w}, wcr[4] = {wc.x, wc.y, wc.z, wc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool pos = (swl[4][r] >> (16 * ct + 4 * q + e)) & 1u;
          v[e] = (ga * war[e] + gb * wbr[e] + gc * wcr[e]) * (pos ? 1.0f : kSlope);
        }
      };
      float vmax = 0.0f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          float v0[4], v1[4];
          gz4(r, 0, q, v0);
          gz4(r, 1, q, v1);
#pragma unroll
          for (int e = 0; e < 4; ++e) vmax = fmaxf(vmax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
          store_t8(g4 + t8_off(32 * (RT * w + r) + 8 * q, 0), lane, v0, v1);
        }
      // the tile's operand scale: the waves' maxima through LDS (X is idle here: the previous tile is done with it)
      vmax = wave_max_all(vmax);
      if (lane == 0) tmx[w] = vmax;
      __syncthreads();
      const float gmax = tile_max(tmx);
      sx = pow2_scale(gmax);
      if (tid == 0) fold_amax(gtail + 4, gmax);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4];
            gz4(r, ct, q, v);
            store_split4(X, 4 * (RT * w + r) + q, 32 * ct + j, hh, sx, v[0], v[1], v[2], v[3]);
          }
    }
    {  // dW5 += g_z5 h4^T  (raw = the h4 blocks requested at the top of the tile)
#pragma unroll
      for (int i = 0; i < kNL; ++i) {
        if constexpr (S16) {
          // load i = the 8-channel block i of this wave: lane L holds channel (L >> 1) & 7, positions 16 (L >> 4) + 8 (L & 1) ..
          float hv[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f16x2 pr = __builtin_bit_cast(f16x2, (unsigned)raw[i][k]);
            hv[2 * k] = (float)pr[0];
            hv[2 * k + 1] = (float)pr[1];
          }
          const int p0 = 16 * (lane >> 4) + 8 * (lane & 1);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 ga = *reinterpret_cast<const float4*>(gz5p + c * kTM + p0), gb = *reinterpret_cast<const float4*>(gz5p + c * kTM + p0 + 4);
            float v = ga.x * hv[0];
            v = __builtin_fmaf(ga.y, hv[1], v); v = __builtin_fmaf(ga.z, hv[2], v); v = __builtin_fmaf(ga.w, hv[3], v);
            v = __builtin_fmaf(gb.x, hv[4], v); v = __builtin_fmaf(gb.y, hv[5], v); v = __builtin_fmaf(gb.z, hv[6], v); v = __builtin_fmaf(gb.w, hv[7], v);
            v += quad_swap1(v);                          // the other half of the chunk; the four chunks meet in the atomic
            if ((lane & 1) == 0) unsafeAtomicAdd(dw5s + c * kWidth + kCW * w + 8 * i + ((lane >> 1) & 7), v);
          }
        } else {
          // loads 2 b, 2 b + 1 = the 8-channel block b of this wave: lane L holds channel (L >> 2) & 7, positions
          // 16 (2 (i & 1) + (L >> 5)) + 4 (L & 3) ..
          const int p0 = 16 * (2 * (i & 1) + (lane >> 5)) + 4 * (lane & 3);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 ga = *reinterpret_cast<const float4*>(gz5p + c * kTM + p0);
            float v = ga.x * __uint_as_float(raw[i][0]);
            v = __builtin_fmaf(ga.y, __uint_as_float(raw[i][1]), v);
            v = __builtin_fmaf(ga.z, __uint_as_float(raw[i][2]), v);
            v = __builtin_fmaf(ga.w, __uint_as_float(raw[i][3]), v);
            v += quad_swap1(v);                          // the 16 positions of the chunk; the four chunks meet in the atomic
            v += quad_swap2(v);
            if ((lane & 3) == 0) unsafeAtomicAdd(dw5s + c * kWidth + kCW * w + 8 * (i >> 1) + ((lane >> 2) & 7), v);
          }
        }
      }
    }
    __syncthreads();
    // layers 4..1: g_z_{l-1} = (W_l^T g_z_l) * LeakyReLU'(h_{l-1})
    const int nv0 = a.g.rt0 - RT * w;                      // layer 0: this wave's row tiles that exist (wave uniform)
    u32x4 apre[RT][2];                                     // the NEXT GEMM's weight fragments of K step 0 (DVD_MLP_DX_PRE)
    if (DVD_MLP_DX_PRE) {
      load_a_frags<RT>(P4 + a.L.bwd[4] + (size_t)(RT * w) * 16 * 128 + lane, 16 * 128, 0, apre);
      // (consumed here: still pending at the loop's entry, with no store behind them, they would make the loop's wait for a
      //  layer's first fragments s_waitcnt vmcnt(0) -- the compiler merges the entry state with the back edge's -- which is the
      //  wait for every stash store this change is about)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(apre[r][t]));
    }
#pragma unroll 1
    for (int l = 4; l >= 1; --l) {
      f32x16 acc[RT][2];
      zero_acc<RT>(acc);
      if (DVD_MLP_DX_PRE)
        gemm_rows_pre<RT>(P4 + a.L.bwd[l] + (size_t)(RT * w) * 16 * 128 + lane, 16 * 128, 16, Xl, acc, apre);
      else
        gemm_rows<RT>(P4 + a.L.bwd[l] + (size_t)(RT * w) * 16 * 128 + lane, 16 * 128, 16, Xl, acc);
      const float unscale = 1.0f / (sx * pow2_scale(pf[a.L.wamax + l]));
      unsigned sw[RT];                                   // sign words of h_{l-1} (register selects: l is a run-time value)
#pragma unroll
      for (int r = 0; r < RT; ++r) sw[r] = l == 4 ? swl[3][r] : (l == 3 ? swl[2][r] : (l == 2 ? swl[1][r] : swl[0][r]));
      float vmax = 0.0f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool pos = (sw[r] >> (16 * ct + e)) & 1u;
            acc[r][ct][e] *= unscale * (pos ? 1.0f : kSlope);
            vmax = fmaxf(vmax, fabsf(acc[r][ct][e]));
          }
      vmax = wave_max_all(vmax);
      if (lane == 0) tmx[w] = vmax;
      __syncthreads();  // every wave has finished reading g_z_l; the waves' maxima are visible
      const float gmax = tile_max(tmx);
      sx = pow2_scale(gmax);
      if (tid == 0) fold_amax(gtail + (l - 1), gmax);
      float* gl = gs + (size_t)(l - 1) * kWidth * kTM;
      // the next GEMM's first fragments, in FRONT of this layer's stash stores (layer 0's only where the wave's rows exist)
      if (DVD_MLP_DX_PRE && (l > 1 || nv0 >= RT))
        load_a_frags<RT>(P4 + a.L.bwd[l - 1] + (size_t)(RT * w) * 16 * 128 + lane, 16 * 128, 0, apre);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            store_split4(X, 4 * (RT * w + r) + q, 32 * ct + j, hh, sx, acc[r][ct][4 * q + 0], acc[r][ct][4 * q + 1], acc[r][ct][4 * q + 2], acc[r][ct][4 * q + 3]);
          const float x[4] = {acc[r][0][4 * q + 0], acc[r][0][4 * q + 1], acc[r][0][4 * q + 2], acc[r][0][4 * q + 3]};
          const float y[4] = {acc[r][1][4 * q + 0], acc[r][1][4 * q + 1], acc[r][1][4 * q + 2], acc[r][1][4 * q + 3]};
          store_t8(gl + t8_off(32 * (RT * w + r) + 8 * q, 0), lane, x, y);
        }
      __syncthreads();
    }
    // layer 0: g_in = W_0^T g_z0  (c_in <= 256 rows = rt0 <= 8 row tiles; wave w takes the tiles RT w .. RT w + RT - 1 that
    // exist), as fp32 [channel][64]
    {
      f32x16 acc[RT][2];
      zero_acc<RT>(acc);
      const int nv = nv0;
      const u32x4* Ap = P4 + a.L.bwd[0] + (size_t)(RT * w) * 16 * 128 + lane;
      if (nv >= RT) {
        if (DVD_MLP_DX_PRE)
          gemm_rows_pre<RT>(Ap, 16 * 128, 16, Xl, acc, apre);
        else
          gemm_rows<RT>(Ap, 16 * 128, 16, Xl, acc);
      } else if (RT > 1 && nv == 1) {
        gemm_rows<1>(Ap, 16 * 128, 16, Xl, acc);
      }
      const float unscale = 1.0f / (sx * pow2_scale(pf[a.L.wamax + 0]));
      __syncthreads();  // all waves finished reading g_z0
#pragma unroll
      for (int r = 0; r < RT; ++r)
        if (r < nv) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              Xf[(32 * (RT * w + r) + 8 * (e >> 2) + 4 * hh + (e & 3)) * kTM + 32 * ct + j] = acc[r][ct][e] * unscale;
        }
      __syncthreads();
    }
    // embedding backward -> g_p[c][m]:  thread (c, m), waves 0..2
    if (tid < 3 * kTM) {
      const int c = tid >> 6;
      const int nx = a.g.n_freq_xyz, xb = a.g.xyz_base;
      auto gin = [&](int ch) { return Xf[ch * kTM + lane]; };
      auto emb = [&](int ch) { return st[(size_t)ch * kTM + t8_pos(lane)]; };
      float gx = gin(xb + c);
      for (int i = 0; i < nx; ++i) {
        const int cc = xb + 3 + 3 * i + c, cs = xb + 3 + 3 * nx + 3 * i + c;
        // d cos(f x)/dx = -f sin(f x) ; d sin(f x)/dx = f cos(f x)
        gx = __builtin_fmaf(a.freqs_xyz[i], __builtin_fmaf(emb(cc), gin(cs), -emb(cs) * gin(cc)), gx);
      }
      const long long n = n0 + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + c) * a.pix_per_img + hw);
        if (a.g_p_add) gx += a.g_p_add[o];
        a.g_p[o] = gx;
      }
    }
  }
  // flush the last layer's parameter gradients
  __syncthreads();
  for (int i = tid; i < 3 * kWidth; i += NT) unsafeAtomicAdd(a.gW5 + i, dw5s[i]);
  {
    const float v = wave_sum(db5);
    if (lane == 0 && w < 3) unsafeAtomicAdd(a.gb5 + w, v);
  }
}

// ==========================================================================================
// backward, dW:  dW_l[n][k] = sum_pixels G_l[n][m] H_{l-1}[k][m]   (H_{-1} = the embedding)
struct DwArgs {
  const float* stash;
  const float* gstash;
  float* partial;      // [5][S][kDwPartial]
  Geometry g;
  int n_tiles, S;
};

constexpr int kDwPitch = 48;                       // bytes per channel row of a 16-pixel chunk in LDS (32 + 16 pad)
constexpr int kDwTerm = 256 * kDwPitch;            // 12 288: one split term of one operand
constexpr int kDwBuf = 2 * 2 * kDwTerm;            // 49 152: G terms, then H terms
constexpr size_t kDwLds = 2 * (size_t)kDwBuf;      // double buffered

// FULL: every H row and every column tile is live (layers 1..4) -- no predicates in the chunk loop
// H16: the H operand (h_{layer-1}, layers 1..4) comes from an fp16 stash -- one term, no scale: G (two terms) x H = 2 MFMAs
template <bool FULL, bool H16>
__device__ __forceinline__ void dw_body(const DwArgs& a, unsigned char* smem) {
  static_assert(FULL || !H16, "the embedding (layer 0's H) is always fp32");
  const bool s16 = a.g.s16 != 0;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;               // rows [64 wr, +64), columns [128 wc, +128)
  const int i32 = lane & 31, hh = lane >> 5;
  const int layer = blockIdx.y, s = blockIdx.x;
  const int t0 = (int)((long long)a.n_tiles * s / a.S), t1 = (int)((long long)a.n_tiles * (s + 1) / a.S);
  const int n_it = (t1 - t0) * 4;                  // 16-pixel chunks
  const int hrows = FULL ? kWidth : a.g.c_in16;
  const int nct = FULL ? 4 : ((a.g.c_in16 + 31) / 32 - 4 * wc);   // column tiles of this wave that hold anything
  const size_t spt = stash_floats_per_tile(a.g.c_in16, s16), gpt = gstash_floats_per_tile();
  const size_t hoff = layer == 0 ? 0 : stash_h_off(a.g.c_in16, layer - 1, s16);
  const size_t goff = (size_t)layer * kWidth * kTM;

  f32x16 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.0f;
  float rs[2] = {0.f, 0.f};                        // row sums of G (bias gradient): rows row_of(0), row_of(1) of this thread's pixel quad
  // per-launch operand scales from the maxima the forward / dX kernels left behind the stash's tiles
  const float* tail = a.stash + (size_t)a.n_tiles * spt;
  const float scg = pow2_scale(tail[8 + layer]);                                  // G_layer
  const float sch = H16 ? 1.0f : pow2_scale(tail[layer]);                         // H = embedding (0) or h_{layer-1}

  // staging: 512 rows x 4 quads of 4 pixel positions -> 4 float4 per thread; q = i * 512 + tid, row = q >> 2, quad = q & 3
  // (i = 0, 1: G rows 0 .. 255; i = 2, 3: H rows).  Operands in the T8 layout (G, h_l): 32 consecutive threads read the 512
  // contiguous bytes one 8-channel block holds of this 16-position chunk (fp16: 256); the embedding (layer 0's H) is
  // [channel][64 positions] rows: four consecutive threads read a row's 64 bytes.
  // Two register sets: the chunk loaded during step `it` is split and stored during step it + 1 and consumed by the
  // MFMAs of step it + 2, so no wave ever waits for HBM.
  float4 sg0[4], sg1[4];
  auto row_of = [&](int i) { return i * 128 + (tid >> 2); };
  auto quad_of = [&](int) { return tid & 3; };
  auto stage_load = [&](int it, float4 (&sg)[4]) {
    const int tile = t0 + (it >> 2), chunk = it & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row_of(i), quad = quad_of(i);
      const bool isH = i >= 2;
      const int r = row & 255;
      if (H16 && isH) {      // four fp16 values = 8 bytes, carried in the first two lanes of the float4 as raw bits
        const _Float16* src = reinterpret_cast<const _Float16*>(a.stash + (size_t)tile * spt + hoff);
        const float2 raw = *reinterpret_cast<const float2*>(src + t8_off(r, chunk * 16 + quad * 4));
        sg[i] = make_float4(raw.x, raw.y, 0.f, 0.f);
        continue;
      }
      if (isH && !FULL) {    // the embedding
        const bool valid = r < hrows;
        const float4 v = *reinterpret_cast<const float4*>(a.stash + (size_t)tile * spt + (size_t)(valid ? r : 0) * kTM + chunk * 16 + quad * 4);
        sg[i] = valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const float* src = isH ? a.stash + (size_t)tile * spt + hoff : a.gstash + (size_t)tile * gpt + goff;
      sg[i] = *reinterpret_cast<const float4*>(src + t8_off(r, chunk * 16 + quad * 4));
    }
  };
  auto stage_store = [&](int buf, const float4 (&sg)[4], bool count) {
    unsigned char* base = smem + buf * kDwBuf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row_of(i), quad = quad_of(i);
      if (i < 2) rs[i] += count ? (sg[i].x + sg[i].y) + (sg[i].z + sg[i].w) : 0.0f;
      if (H16 && i >= 2) {
        unsigned char* dst16 = base + 2 * kDwTerm + (row & 255) * kDwPitch + quad * 8;
        *reinterpret_cast<u32x2*>(dst16) = (u32x2){__float_as_uint(sg[i].x), __float_as_uint(sg[i].y)};
        continue;
      }
      unsigned h0, l0, h1, l1;
      const float sc = i >= 2 ? sch : scg;
      split_pair_f16(sg[i].x * sc, sg[i].y * sc, h0, l0);
      split_pair_f16(sg[i].z * sc, sg[i].w * sc, h1, l1);
      unsigned char* dst = base + (i >= 2 ? 2 * kDwTerm : 0) + (row & 255) * kDwPitch + quad * 8;
      *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
      *reinterpret_cast<u32x2*>(dst + kDwTerm) = (u32x2){l0, l1};
    }
  };
  auto mfma_chunk = [&](int buf) {
    const unsigned char* gb = smem + buf * kDwBuf + (64 * wr + i32) * kDwPitch + hh * 16;
    const unsigned char* hb = smem + buf * kDwBuf + 2 * kDwTerm + (128 * wc + i32) * kDwPitch + hh * 16;
    u32x4 A[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) A[r][t] = *reinterpret_cast<const u32x4*>(gb + t * kDwTerm + r * 32 * kDwPitch);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (FULL || c < nct) {
        u32x4 B[2];
#pragma unroll
        for (int t = 0; t < (H16 ? 1 : 2); ++t) B[t] = *reinterpret_cast<const u32x4*>(hb + t * kDwTerm + c * 32 * kDwPitch);
#define DVD_DW_TERM(SA, SB)                                                                                     \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(             \
      __builtin_bit_cast(f16x8, A[r][SA]), __builtin_bit_cast(f16x8, B[SB]), acc[r][c], 0, 0, 0);
        DVD_DW_TERM(1, 0)
        if constexpr (!H16) { DVD_DW_TERM(0, 1) }
        DVD_DW_TERM(0, 0)
#undef DVD_DW_TERM
      }
    }
  };

  // n_it is a multiple of 4 (whole tiles), so the chunk loop runs in pairs with the two register sets / LDS buffers
  // in fixed roles.  One barrier per chunk: buffer b is complete, and every wave is done reading buffer b ^ 1.
  const int last = n_it - 1;
  stage_load(0, sg0);
  stage_store(0, sg0, true);
  stage_load(1 < last ? 1 : last, sg0);
#pragma unroll 1
  for (int it = 0; it < n_it; it += 2) {
    __syncthreads();
    stage_load(it + 2 < last ? it + 2 : last, sg1);
    __builtin_amdgcn_sched_barrier(0);   // the loads stay above the MFMAs
    mfma_chunk(0);
    stage_store(1, sg0, true);           // chunk it + 1 (always exists)
    __syncthreads();
    stage_load(it + 3 < last ? it + 3 : last, sg0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(1);
    // chunk it + 2; past the end this rewrites the idle buffer with the last chunk, not counted in the row sums
    // (a conditional store would drag its loads into the branch, behind the MFMAs)
    stage_store(0, sg1, it + 2 < n_it);
  }
  // this workgroup's partial matrix: element r of acc[rr][c] in lane l is row 64 wr + 32 rr + (r&3) + 8 (r>>2) + 4 (l>>5),
  // column 128 wc + 32 c + (l&31)
  float* dst = a.partial + ((size_t)layer * a.S + s) * kDwPartial;
  const float unscale = 1.0f / (scg * sch);          // exact power of two
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 64 * wr + 32 * rr + (r & 3) + 8 * (r >> 2) + 4 * hh, k = 128 * wc + 32 * c + i32;
        dst[(size_t)n * kWidth + k] = acc[rr][c][r] * unscale;
      }
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // G rows: thread = (row row_of(i), quad tid & 3)
    float v = rs[i];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if ((tid & 3) == 0) dst[(size_t)kWidth * kWidth + row_of(i)] = v;
  }
}

template <bool S16>
__global__ __launch_bounds__(kNT) void mlp_bwd_dw_kernel(const DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.y == 0)
    dw_body<false, false>(a, smem);
  else
    dw_body<true, S16>(a, smem);
}

// gW_l[n][k] += sum_s partial[l][s][n][k] (ascending s), gb_l[n] += sum_s rowsum[l][s][n]
struct DwReduceArgs {
  const float* partial;
  float* gW[kHidden];
  float* gb[kHidden];
  int S, c_in;
};

__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(const DwReduceArgs a) {
  const int layer = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // over [256][256] + [256]
  if (idx >= (int)kDwPartial) return;
  const float* p = a.partial + (size_t)layer * a.S * kDwPartial + idx;
  float s0 = 0.0f, s1 = 0.0f;
  int s = 0;
  for (; s + 1 < a.S; s += 2) {
    s0 += p[(size_t)s * kDwPartial];
    s1 += p[(size_t)(s + 1) * kDwPartial];
  }
  if (s < a.S) s0 += p[(size_t)s * kDwPartial];
  const float v = s0 + s1;
  if (idx < kWidth * kWidth) {
    const int K = layer == 0 ? a.c_in : kWidth;
    const int n = idx >> 8, k = idx & 255;
    if (k < K) a.gW[layer][(size_t)n * K + k] += v;
  } else {
    a.gb[layer][idx - kWidth * kWidth] += v;
  }
}

static int check_desc(const dvd_mlp_desc* d) {
  DVD_REQUIRE(d, "sf_mlp: null descriptor");
  DVD_REQUIRE(d->n_freq_xyz >= 0 && d->n_freq_xyz <= 20 && d->n_freq_t >= 0 && d->n_freq_t <= 20,
              "sf_mlp: unsupported frequency counts %d/%d", d->n_freq_xyz, d->n_freq_t);
  DVD_REQUIRE(d->n_freq_xyz == 0 || d->freqs_xyz, "sf_mlp: freqs_xyz is null");
  DVD_REQUIRE(!d->time_dependent || d->n_freq_t == 0 || d->freqs_t, "sf_mlp: freqs_t is null");
  return DVD_OK;
}

static int dw_slices(int n_tiles) { return n_tiles < kDwSlices ? n_tiles : kDwSlices; }

// waves per workgroup of the forward / dX kernels: 4 = two 256-thread workgroups per CU (default), 8 = one of 512 threads
// (rounds 2-4).  dvd_sf_mlp_select.
static int g_mlp_nw = DVD_MLP_NW;

static int persistent_grid(int n_tiles, int occ, int nw) {
  int cus = dvd_device_cu_count();
  if (cus <= 0) cus = 256;
  const int wgs = cus * (occ / 2) * (8 / nw);    // workgroups resident at once (8 waves per CU at occ = 2)
  return n_tiles < wgs ? n_tiles : wgs;
}

}  // namespace dvd

extern "C" {

int dvd_sf_mlp_select(int waves_per_workgroup) {
  DVD_REQUIRE(waves_per_workgroup == 0 || waves_per_workgroup == 4 || waves_per_workgroup == 8,
              "sf_mlp_select: waves per workgroup must be 4 or 8 (0 = default), got %d", waves_per_workgroup);
  dvd::g_mlp_nw = waves_per_workgroup ? waves_per_workgroup : DVD_MLP_NW;
  return DVD_OK;
}

int dvd_sf_mlp_in_channels(const dvd_mlp_desc* d) { return d ? dvd::make_geometry(d).c_in : -1; }

size_t dvd_sf_mlp_packed_bytes(const dvd_mlp_desc* d) {
  if (!d) return 0;
  return dvd::make_pack_layout(dvd::make_geometry(d)).total_bytes;
}

size_t dvd_sf_mlp_stash_bytes(const dvd_mlp_desc* d, long long n_pix) {
  if (!d || n_pix <= 0) return 0;
  const long long tiles = (n_pix + dvd::kTM - 1) / dvd::kTM;
  return ((size_t)tiles * dvd::stash_floats_per_tile(dvd::make_geometry(d).c_in16, d->stash_f16 != 0) + dvd::kStashTail) * 4;
}

size_t dvd_sf_mlp_gstash_bytes(long long n_pix) {
  if (n_pix <= 0) return 0;
  const long long tiles = (n_pix + dvd::kTM - 1) / dvd::kTM;
  return ((size_t)tiles * dvd::gstash_floats_per_tile() + (size_t)dvd::kHidden * dvd::kDwSlices * dvd::kDwPartial) * 4;
}

int dvd_sf_mlp_pack(const dvd_mlp_desc* d, const float* const W[6], const float* const b[6], void* packed,
                    dvd_stream_t stream) {
  using namespace dvd;
  if (int e = check_desc(d)) return e;
  DVD_REQUIRE(W && b && packed, "sf_mlp_pack: null pointer");
  PackArgs a;
  for (int l = 0; l < 6; ++l) {
    DVD_REQUIRE(W[l] && b[l], "sf_mlp_pack: null weight/bias %d", l);
    a.W[l] = W[l];
    a.b[l] = b[l];
  }
  const Geometry g = make_geometry(d);
  DVD_REQUIRE(g.c_in <= kWidth, "sf_mlp_pack: input layer wider than 256 channels");
  a.out = packed;
  a.L = make_pack_layout(g);
  a.c_in = g.c_in;
  // per-layer max|W_l| first (the packed buffer's tail), then the scaled split
  if (int e = zero_words(static_cast<float*>(packed) + a.L.wamax, 8, static_cast<hipStream_t>(stream))) return e;
  hipLaunchKernelGGL(mlp_wamax_kernel, dim3(16, kHidden), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  DVD_LAUNCH_OK();
  // the largest job has 8 row tiles x 16 K steps x 64 lanes = 8192 threads
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(32, 11), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_sf_mlp_fwd(const dvd_mlp_desc* d, const void* packed, const float* p, const float* t, float t_offset,
                   float out_scale, long long n_pix, int pix_per_img, float* sf_out, float* p_next, float* acc,
                   void* stash, dvd_stream_t stream) {
  using namespace dvd;
  if (int e = check_desc(d)) return e;
  DVD_REQUIRE(packed && p, "sf_mlp_fwd: null pointer");
  DVD_REQUIRE(n_pix > 0 && pix_per_img > 0 && n_pix % pix_per_img == 0, "sf_mlp_fwd: bad sizes %lld / %d", n_pix,
              pix_per_img);
  DVD_REQUIRE(!d->time_dependent || t, "sf_mlp_fwd: time-dependent model needs t");
  DVD_REQUIRE(n_pix < (1LL << 31) * 16, "sf_mlp_fwd: too many pixels");
  FwdArgs a;
  a.g = make_geometry(d);
  DVD_REQUIRE(a.g.c_in <= kWidth, "sf_mlp_fwd: input layer wider than 256 channels");
  a.L = make_pack_layout(a.g);
  a.packed = packed;
  a.p = p;
  a.t = d->time_dependent ? t : nullptr;
  a.freqs_xyz = d->freqs_xyz;
  a.freqs_t = d->freqs_t;
  a.sf_out = sf_out;
  a.p_next = p_next;
  a.acc = acc;
  a.stash = static_cast<float*>(stash);
  a.monitor = (stash && d->stash_f16) ? d->fwd_monitor : nullptr;
  a.n_pix = n_pix;
  a.pix_per_img = pix_per_img;
  a.n_tiles = (int)((n_pix + kTM - 1) / kTM);
  a.t_offset = t_offset;
  a.out_scale = out_scale;
  const int nw = g_mlp_nw;
  const int grid = persistent_grid(a.n_tiles, DVD_MLP_FWD_OCC, nw);
  hipStream_t s = static_cast<hipStream_t>(stream);
  flops_add(DVD_FLOP_MLP_FWD, 2.0 * ((double)a.g.c_in * kWidth + 4.0 * kWidth * kWidth + 3.0 * kWidth) * (double)n_pix);
  if (stash)   // per-layer maxima behind the tiles (embedding, h_0 .. h_4; the dX kernel zeroes its own half)
    if (int e = zero_words(a.stash + (size_t)a.n_tiles * stash_floats_per_tile(a.g.c_in16, a.g.s16 != 0), kStashTail, s)) return e;
  auto go = [&](auto kern) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), kFwdLds, s, a);
    return DVD_OK;
  };
  int e;
  if (stash && a.g.s16) e = nw == 4 ? go(mlp_fwd_kernel<true, true, 4>) : go(mlp_fwd_kernel<true, true, 8>);
  else if (stash) e = nw == 4 ? go(mlp_fwd_kernel<true, false, 4>) : go(mlp_fwd_kernel<true, false, 8>);
  else e = nw == 4 ? go(mlp_fwd_kernel<false, false, 4>) : go(mlp_fwd_kernel<false, false, 8>);
  if (e) return e;
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_sf_mlp_bwd_dx(const dvd_mlp_desc* d, const void* packed, const void* stash, float out_scale,
                      const float* g_out1, float gscale, const float* scale_ptr, const float* g_out2,
                      const float* g_p_add, long long n_pix, int pix_per_img, float* g_p, void* gstash, float* gW5,
                      float* gb5, dvd_stream_t stream) {
  using namespace dvd;
  if (int e = check_desc(d)) return e;
  DVD_REQUIRE(packed && stash && g_out1 && g_p && gstash && gW5 && gb5, "sf_mlp_bwd_dx: null pointer");
  DVD_REQUIRE(n_pix > 0 && pix_per_img > 0 && n_pix % pix_per_img == 0, "sf_mlp_bwd_dx: bad sizes");
  BwdArgs a;
  a.g = make_geometry(d);
  DVD_REQUIRE(a.g.c_in <= kWidth, "sf_mlp_bwd_dx: input layer wider than 256 channels");
  a.L = make_pack_layout(a.g);
  a.packed = packed;
  a.stash = static_cast<const float*>(stash);
  a.gstash = static_cast<float*>(gstash);
  a.g_out1 = g_out1;
  a.g_out2 = g_out2;
  a.scale_ptr = scale_ptr;
  a.g_p_add = g_p_add;
  a.freqs_xyz = d->freqs_xyz;
  a.g_p = g_p;
  a.gW5 = gW5;
  a.gb5 = gb5;
  a.n_pix = n_pix;
  a.pix_per_img = pix_per_img;
  a.n_tiles = (int)((n_pix + kTM - 1) / kTM);
  a.out_scale = out_scale;
  a.gscale = gscale;
  const int nw = g_mlp_nw;
  const int grid = persistent_grid(a.n_tiles, DVD_MLP_DX_OCC, nw);
  // maxima of G_0 .. G_4 over all tiles, folded in by the kernel: floats [8, 16) behind the stash's tiles
  if (int e = zero_words(const_cast<float*>(a.stash) + (size_t)a.n_tiles * stash_floats_per_tile(a.g.c_in16, a.g.s16 != 0) + 8, 8,
                         static_cast<hipStream_t>(stream)))
    return e;
  flops_add(DVD_FLOP_MLP_DX, 2.0 * ((double)a.g.c_in * kWidth + 4.0 * kWidth * kWidth + 3.0 * kWidth) * (double)a.n_pix);
  auto go = [&](auto kern) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), kBwdLds, static_cast<hipStream_t>(stream), a);
    return DVD_OK;
  };
  int e;
  if (a.g.s16) e = nw == 4 ? go(mlp_bwd_dx_kernel<true, 4>) : go(mlp_bwd_dx_kernel<true, 8>);
  else e = nw == 4 ? go(mlp_bwd_dx_kernel<false, 4>) : go(mlp_bwd_dx_kernel<false, 8>);
  if (e) return e;
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_sf_mlp_bwd_dw(const dvd_mlp_desc* d, const void* stash, void* gstash, long long n_pix, float* const gW[5],
                      float* const gb[5], dvd_stream_t stream) {
  using namespace dvd;
  if (int e = check_desc(d)) return e;
  DVD_REQUIRE(stash && gstash && gW && gb && n_pix > 0, "sf_mlp_bwd_dw: null pointer / size");
  DwArgs a;
  a.g = make_geometry(d);
  a.stash = static_cast<const float*>(stash);
  a.gstash = static_cast<const float*>(gstash);
  a.n_tiles = (int)((n_pix + kTM - 1) / kTM);
  a.S = dw_slices(a.n_tiles);
  a.partial = static_cast<float*>(gstash) + (size_t)a.n_tiles * gstash_floats_per_tile();
  DwReduceArgs r;
  for (int l = 0; l < kHidden; ++l) {
    DVD_REQUIRE(gW[l] && gb[l], "sf_mlp_bwd_dw: null gradient %d", l);
    r.gW[l] = gW[l];
    r.gb[l] = gb[l];
  }
  r.partial = a.partial;
  r.S = a.S;
  r.c_in = a.g.c_in;
  hipStream_t s = static_cast<hipStream_t>(stream);
  flops_add(DVD_FLOP_MLP_DW, 2.0 * ((double)a.g.c_in * kWidth + 4.0 * kWidth * kWidth + 3.0 * kWidth) * (double)n_pix);
  if (a.g.s16) {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_dw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kDwLds));
    hipLaunchKernelGGL(mlp_bwd_dw_kernel<true>, dim3(a.S, kHidden), dim3(kNT), kDwLds, s, a);
  } else {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_dw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kDwLds));
    hipLaunchKernelGGL(mlp_bwd_dw_kernel<false>, dim3(a.S, kHidden), dim3(kNT), kDwLds, s, a);
  }
  DVD_LAUNCH_OK();
  hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3((unsigned)((kDwPartial + 255) / 256), kHidden), dim3(256), 0, s, r);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
