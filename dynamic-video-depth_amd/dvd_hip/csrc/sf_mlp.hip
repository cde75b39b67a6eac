// Fused scene-flow field MLP for gfx950 (MI355X): forward, backward-dX chain, backward-dW.
//
// What it replaces (reference, /root/reference):
//   networks/sceneflow_field.py:43-53   SceneFlowFieldNet.forward
//   networks/blocks.py:19-34            PeriodicEmbed (64 separate sin/cos launches per eval)
//   networks/blocks.py:50-102           Conv2dBlock = 1x1 conv + LeakyReLU(0.2), x6
//   models/scene_flow_motion_field.py:346-367  forward_sf_net (/= sf_mag_div) and one Euler step
// and the autograd backward of all of it.  The unfused reference saves 11 168 B per
// pixel-evaluation for backward and launches ~80 kernels per evaluation.
//
// Roofline: fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TF peak, bit-exact fp32 FMA chains).
// 593 408 FLOP per pixel forward, 2x that backward.
//
// Structure (all three kernels): a 256-thread workgroup (4 waves, one per SIMD, two
// workgroups per CU) owns a tile of 64 pixels.
//   * Activations live in LDS as X[kq][m] float4 = channels 4kq..4kq+3 of pixel m
//     ("k-quad major").  One ds_read_b128 per lane feeds FOUR MFMA k-steps of a
//     32-pixel column tile, and the MFMA result registers (4 consecutive output channels
//     of one pixel per lane) go back with one ds_write_b128 - both conflict free.
//   * Weights never touch LDS: they are pre-packed (dvd_sf_mlp_pack) in MFMA fragment
//     order so that a lane's operands for four k-steps are one 16-byte global load; the
//     1.2 MB of packed weights stay in each XCD's 4 MB L2.
//   * forward: wave w computes output channels [64w, 64w+64) for all 64 pixels
//     (2x2 tiles of 32x32, 64 accumulator registers), layer after layer in place.
//   * backward dX: same loop with the transposed packing, multiplied by LeakyReLU'
//     (sign of the stashed activation); writes the pre-activation gradients G_l.
//   * backward dW: dW_l = G_l H_{l-1}^T contracts over PIXELS.  Both operands are read
//     straight from the stash in the layout above, where a lane's 16-byte load holds
//     four channels of one pixel = four MFMA row(col) tiles; a wave keeps a 128x128
//     block of dW_l in 256 accumulator registers across all of its tiles and adds it to
//     global memory once.
//
// The forward writes the embedding and the five hidden activations to the stash
// (5 664 B per pixel, streamed while the MFMAs run): at fp32-MFMA rates re-reading them
// is cheaper than recomputing the forward in the backward pass (+33 % MFMA work).

#include "dvd_common.h"

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 64;      // pixels per tile
constexpr int kWidth = 256;  // hidden width (scene_flow_motion_field.py:107)
constexpr int kKQ = kWidth / 4;
constexpr int kHidden = 5;   // layers with LeakyReLU: convs.0 .. convs.4
constexpr float kSlope = 0.2f;

struct Geometry {
  int n_freq_xyz, n_freq_t, time_dependent;
  int c_in, c_in_pad, kb0, kq0;  // input channels, padded to 8, k-blocks, k-quads
  int t_base, xyz_base;          // channel of t / of x in the input layer
};

static Geometry make_geometry(const dvd_mlp_desc* d) {
  Geometry g;
  g.n_freq_xyz = d->n_freq_xyz;
  g.n_freq_t = d->time_dependent ? d->n_freq_t : 0;
  g.time_dependent = d->time_dependent;
  const int ct = d->time_dependent ? 1 + 2 * d->n_freq_t : 0;
  g.c_in = ct + 3 + 6 * d->n_freq_xyz;
  g.c_in_pad = (g.c_in + 7) & ~7;
  g.kb0 = g.c_in_pad / 8;
  g.kq0 = g.c_in_pad / 4;
  g.t_base = 0;
  g.xyz_base = ct;
  return g;
}

// ---- packed weight buffer (floats) -------------------------------------------------
struct PackLayout {
  size_t fwd[kHidden];  // [8 nt][KB_l][64 lanes][4]
  size_t bwd[kHidden];  // [KT_l][32][64][4]      (W^T, for the dX chain)
  size_t w5, bias[6];
  int kb[kHidden], kt[kHidden];
  size_t total;
};

static PackLayout make_pack_layout(const Geometry& g) {
  PackLayout L;
  size_t off = 0;
  for (int l = 0; l < kHidden; ++l) {
    L.kb[l] = l == 0 ? g.kb0 : kWidth / 8;
    L.kt[l] = l == 0 ? (g.c_in_pad + 31) / 32 : kWidth / 32;
    L.fwd[l] = off;
    off += (size_t)8 * L.kb[l] * 256;
    L.bwd[l] = off;
    off += (size_t)L.kt[l] * 32 * 256;
  }
  L.w5 = off;
  off += 3 * kWidth;
  for (int l = 0; l < 6; ++l) {
    L.bias[l] = off;
    off += l < 5 ? kWidth : 4;
  }
  L.total = off;
  return L;
}

struct PackArgs {
  const float* W[6];
  const float* b[6];
  float* out;
  PackLayout L;
  int c_in;
};

__global__ __launch_bounds__(256) void mlp_pack_kernel(const PackArgs a) {
  const int l = blockIdx.y;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (l < kHidden) {
    const int K = l == 0 ? a.c_in : kWidth;  // true fan-in; padded positions get 0
    const float* W = a.W[l];
    const int kb = a.L.kb[l], kt = a.L.kt[l];
    const int nf = 8 * kb * 256;
    if (gid < nf) {
      const int s = gid & 3, lane = (gid >> 2) & 63, G = (gid >> 8) % kb, nt = (gid >> 8) / kb;
      const int n = 32 * nt + (lane & 31), k = 8 * G + 4 * (lane >> 5) + s;
      a.out[a.L.fwd[l] + gid] = k < K ? W[(size_t)n * K + k] : 0.0f;
    }
    const int nb = kt * 32 * 256;
    if (gid < nb) {
      const int s = gid & 3, lane = (gid >> 2) & 63, G = (gid >> 8) & 31, t = gid >> 13;
      const int n = 8 * G + 4 * (lane >> 5) + s, k = 32 * t + (lane & 31);
      a.out[a.L.bwd[l] + gid] = k < K ? W[(size_t)n * K + k] : 0.0f;
    }
    if (gid < kWidth) a.out[a.L.bias[l] + gid] = a.b[l][gid];
  } else {
    if (gid < 3 * kWidth) a.out[a.L.w5 + gid] = a.W[5][gid];
    if (gid < 4) a.out[a.L.bias[5] + gid] = gid < 3 ? a.b[5][gid] : 0.0f;
  }
}

// ---- stash layout --------------------------------------------------------------------
// per tile: float4 cells [kq0][64] (embedding) then 5 x [64][64] (hidden activations)
__host__ __device__ inline size_t stash_cells_per_tile(int kq0) { return (size_t)kq0 * kTM + (size_t)kHidden * kKQ * kTM; }
__host__ __device__ inline size_t stash_layer_off(int kq0, int slot) {  // slot 0 = embedding, 1..5 = h0..h4
  return slot == 0 ? 0 : (size_t)kq0 * kTM + (size_t)(slot - 1) * kKQ * kTM;
}
// gstash per tile: 5 x [64][64] cells (pre-activation gradients of layers 0..4) + [64] cells (g_z5, 3 used)
__host__ __device__ inline size_t gstash_cells_per_tile() { return (size_t)kHidden * kKQ * kTM + kTM; }

struct FwdArgs {
  const float* packed;
  const float* p;
  const float* t;
  const float* freqs_xyz;
  const float* freqs_t;
  float* sf_out;
  float* p_next;
  float* acc;
  float4* stash;
  PackLayout L;
  Geometry g;
  long long n_pix;
  int pix_per_img, n_tiles;
  float t_offset, out_scale;
};

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, kSlope * v); }

// Input embedding of one tile into X (k-quad major) ; psm = [4][64] floats: x,y,z,t of the pixels.
__device__ __forceinline__ void build_embedding(const Geometry& g, const float* __restrict__ fx,
                                                const float* __restrict__ ft, const float* psm, float* Xf) {
  const int m = threadIdx.x & 63, part = threadIdx.x >> 6;
  auto put = [&](int ch, float v) { Xf[((ch >> 2) * kTM + m) * 4 + (ch & 3)] = v; };
  const float x0 = psm[m], x1 = psm[kTM + m], x2 = psm[2 * kTM + m], tt = psm[3 * kTM + m];
  if (part == 0) {
    if (g.time_dependent) put(g.t_base, tt);
    put(g.xyz_base + 0, x0);
    put(g.xyz_base + 1, x1);
    put(g.xyz_base + 2, x2);
    for (int ch = g.c_in; ch < g.c_in_pad; ++ch) put(ch, 0.0f);
  }
  const int nt = g.n_freq_t, nx = g.n_freq_xyz;
  const int items = nt + 3 * nx;
  for (int e = part; e < items; e += 4) {
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

// One dense layer on the tile: acc[nt][mt] += A(packed weights) x B(X in LDS) over KB k-blocks of 8.
// Wave w owns row tiles 2w, 2w+1 of the packed matrix.
__device__ __forceinline__ void gemm_tile(const float4* __restrict__ Wp, int KB, int rowtile0, const float4* X4,
                                          int lane, f32x16 acc[2][2]) {
  const int j = lane & 31, h = lane >> 5;
  const float4* A0 = Wp + (size_t)(rowtile0 + 0) * KB * 64 + lane;
  const float4* A1 = Wp + (size_t)(rowtile0 + 1) * KB * 64 + lane;
  const float4* B0 = X4 + h * kTM + j;
  float4 a0 = A0[0], a1 = A1[0], b0 = B0[0], b1 = B0[32];
#pragma unroll 1
  for (int G = 0; G < KB; ++G) {
    float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
    if (G + 1 < KB) {
      na0 = A0[(size_t)(G + 1) * 64];
      na1 = A1[(size_t)(G + 1) * 64];
      nb0 = B0[(size_t)(G + 1) * 2 * kTM];
      nb1 = B0[(size_t)(G + 1) * 2 * kTM + 32];
    }
#define DVD_STEP(S)                                                                       \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.S, b0.S, acc[0][0], 0, 0, 0);       \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.S, b1.S, acc[0][1], 0, 0, 0);       \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.S, b0.S, acc[1][0], 0, 0, 0);       \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.S, b1.S, acc[1][1], 0, 0, 0);
    DVD_STEP(x)
    DVD_STEP(y)
    DVD_STEP(z)
    DVD_STEP(w)
#undef DVD_STEP
    a0 = na0;
    a1 = na1;
    b0 = nb0;
    b1 = nb1;
  }
}

// Same with a single row tile (acc[0][*] only).
__device__ __forceinline__ void gemm_tile_single(const float4* __restrict__ Wp, int KB, int rowtile, const float4* X4,
                                                 int lane, f32x16 acc[2][2]) {
  const int j = lane & 31, h = lane >> 5;
  const float4* A0 = Wp + (size_t)rowtile * KB * 64 + lane;
  const float4* B0 = X4 + h * kTM + j;
  float4 a0 = A0[0], b0 = B0[0], b1 = B0[32];
#pragma unroll 1
  for (int G = 0; G < KB; ++G) {
    float4 na0 = a0, nb0 = b0, nb1 = b1;
    if (G + 1 < KB) {
      na0 = A0[(size_t)(G + 1) * 64];
      nb0 = B0[(size_t)(G + 1) * 2 * kTM];
      nb1 = B0[(size_t)(G + 1) * 2 * kTM + 32];
    }
#define DVD_STEP(S)                                                                 \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.S, b0.S, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.S, b1.S, acc[0][1], 0, 0, 0);
    DVD_STEP(x)
    DVD_STEP(y)
    DVD_STEP(z)
    DVD_STEP(w)
#undef DVD_STEP
    a0 = na0;
    b0 = nb0;
    b1 = nb1;
  }
}

__device__ __forceinline__ void zero_acc(f32x16 acc[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
}

template <bool STASH>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* X4 = reinterpret_cast<float4*>(smem);  // [64 kq][64 m]
  float* Xf = smem;
  float* psm = smem + kKQ * kTM * 4;             // [4][64]
  float* w5 = psm + 4 * kTM;                     // [3][256] + bias5[4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 3 * kWidth + 4; i += 256) w5[i] = a.packed[a.L.w5 + (i < 3 * kWidth ? i : (a.L.bias[5] - a.L.w5) + (i - 3 * kWidth))];
  const float4* P4 = reinterpret_cast<const float4*>(a.packed);

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long n0 = (long long)tile * kTM;
    __syncthreads();  // previous tile's readers are done with X / psm
    {  // pixel inputs: thread (c = w, m = lane)
      const long long n = n0 + lane;
      float v = 0.0f;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        if (w < 3)
          v = a.p[(b * 3 + w) * a.pix_per_img + hw];
        else
          v = a.t ? a.t[n] + a.t_offset : 0.0f;
      }
      psm[w * kTM + lane] = v;
    }
    __syncthreads();
    build_embedding(a.g, a.freqs_xyz, a.freqs_t, psm, Xf);
    __syncthreads();
    float4* st = STASH ? a.stash + (size_t)tile * stash_cells_per_tile(a.g.kq0) : nullptr;
    if (STASH)
      for (int i = tid; i < a.g.kq0 * kTM; i += 256) st[i] = X4[i];

#pragma unroll 1
    for (int l = 0; l < kHidden; ++l) {
      f32x16 acc[2][2];
      zero_acc(acc);
      gemm_tile(P4 + a.L.fwd[l] / 4, a.L.kb[l], 2 * w, X4, lane, acc);
      __syncthreads();  // every wave has finished reading this layer's input
      const float* bias = a.packed + a.L.bias[l];
      float4* sl = STASH ? st + stash_layer_off(a.g.kq0, l + 1) : nullptr;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n4 = 64 * w + 32 * nt + 8 * q + 4 * hh;  // first of 4 consecutive output channels
          const float4 bv = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            float4 v;
            v.x = lrelu(acc[nt][mt][4 * q + 0] + bv.x);
            v.y = lrelu(acc[nt][mt][4 * q + 1] + bv.y);
            v.z = lrelu(acc[nt][mt][4 * q + 2] + bv.z);
            v.w = lrelu(acc[nt][mt][4 * q + 3] + bv.w);
            const int cell = (n4 >> 2) * kTM + 32 * mt + j;
            X4[cell] = v;
            if (STASH) sl[cell] = v;
          }
        }
      __syncthreads();
    }
    // output layer 256 -> 3 on the VALU: thread (c = w, m = lane), waves 0..2
    if (w < 3) {
      const float* wr = w5 + w * kWidth;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
      for (int kq = 0; kq < kKQ; ++kq) {
        const float4 x = X4[kq * kTM + lane];
        const float4 ww = *reinterpret_cast<const float4*>(wr + 4 * kq);
        s0 = __builtin_fmaf(x.x, ww.x, s0);
        s1 = __builtin_fmaf(x.y, ww.y, s1);
        s2 = __builtin_fmaf(x.z, ww.z, s2);
        s3 = __builtin_fmaf(x.w, ww.w, s3);
      }
      const float sf = (((s0 + s1) + (s2 + s3)) + w5[3 * kWidth + w]) * a.out_scale;
      const long long n = n0 + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + w) * a.pix_per_img + hw);
        if (a.sf_out) a.sf_out[o] = sf;
        if (a.p_next) a.p_next[o] = psm[w * kTM + lane] + sf;
        if (a.acc) a.acc[o] += sf;
      }
    }
  }
}

// ======================================================================================
// backward, dX chain
struct BwdArgs {
  const float* packed;
  const float4* stash;
  float4* gstash;
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

__global__ __launch_bounds__(256, 2) void mlp_bwd_dx_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* X4 = reinterpret_cast<float4*>(smem);  // gradient tile, k-quad major
  float* Xf = smem;
  float* gz5 = smem + kKQ * kTM * 4;             // [4][64]: g of the 3 outputs (already * out_scale)
  float* w5 = gz5 + 4 * kTM;                     // [3][256]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 3 * kWidth; i += 256) w5[i] = a.packed[a.L.w5 + i];
  const float4* P4 = reinterpret_cast<const float4*>(a.packed);
  const float s1 = a.gscale * (a.scale_ptr ? a.scale_ptr[0] : 1.0f);
  // last-layer weight/bias gradient, accumulated over this block's tiles:
  // thread t owns k-quad kq = t>>2 and pixels (t&3)*16 .. +15
  float dw5[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float db5 = 0.0f;

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long n0 = (long long)tile * kTM;
    const float4* st = a.stash + (size_t)tile * stash_cells_per_tile(a.g.kq0);
    float4* gs = a.gstash + (size_t)tile * gstash_cells_per_tile();
    __syncthreads();
    if (w < 3) {  // g_z5[c][m]
      const long long n = n0 + lane;
      float v = 0.0f;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + w) * a.pix_per_img + hw);
        v = s1 * a.g_out1[o];
        if (a.g_out2) v += a.g_out2[o];
        v *= a.out_scale;
      }
      gz5[w * kTM + lane] = v;
      db5 += v;
    } else {
      gz5[3 * kTM + lane] = 0.0f;
    }
    __syncthreads();
    {  // layer 5 (256 -> 3): g_h4 = W5^T g_z5, masked by LeakyReLU'(h4); dW5 += g_z5 h4^T
      const int kq = tid >> 2, mb = (tid & 3) * 16;
      const float4* h4 = st + stash_layer_off(a.g.kq0, 5) + kq * kTM;
      float4* g4 = gs + (size_t)4 * kKQ * kTM + kq * kTM;
      const float4 wa = *reinterpret_cast<const float4*>(w5 + 4 * kq);
      const float4 wb = *reinterpret_cast<const float4*>(w5 + kWidth + 4 * kq);
      const float4 wc = *reinterpret_cast<const float4*>(w5 + 2 * kWidth + 4 * kq);
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int m = mb + i;
        const float4 hv = h4[m];
        const float ga = gz5[m], gb = gz5[kTM + m], gc = gz5[2 * kTM + m];
        float4 v;
        v.x = (ga * wa.x + gb * wb.x + gc * wc.x) * (hv.x > 0.f ? 1.f : kSlope);
        v.y = (ga * wa.y + gb * wb.y + gc * wc.y) * (hv.y > 0.f ? 1.f : kSlope);
        v.z = (ga * wa.z + gb * wb.z + gc * wc.z) * (hv.z > 0.f ? 1.f : kSlope);
        v.w = (ga * wa.w + gb * wb.w + gc * wc.w) * (hv.w > 0.f ? 1.f : kSlope);
        X4[kq * kTM + m] = v;
        g4[m] = v;
        dw5[0][0] = __builtin_fmaf(ga, hv.x, dw5[0][0]);
        dw5[0][1] = __builtin_fmaf(ga, hv.y, dw5[0][1]);
        dw5[0][2] = __builtin_fmaf(ga, hv.z, dw5[0][2]);
        dw5[0][3] = __builtin_fmaf(ga, hv.w, dw5[0][3]);
        dw5[1][0] = __builtin_fmaf(gb, hv.x, dw5[1][0]);
        dw5[1][1] = __builtin_fmaf(gb, hv.y, dw5[1][1]);
        dw5[1][2] = __builtin_fmaf(gb, hv.z, dw5[1][2]);
        dw5[1][3] = __builtin_fmaf(gb, hv.w, dw5[1][3]);
        dw5[2][0] = __builtin_fmaf(gc, hv.x, dw5[2][0]);
        dw5[2][1] = __builtin_fmaf(gc, hv.y, dw5[2][1]);
        dw5[2][2] = __builtin_fmaf(gc, hv.z, dw5[2][2]);
        dw5[2][3] = __builtin_fmaf(gc, hv.w, dw5[2][3]);
      }
      if (tid < kTM) gs[(size_t)kHidden * kKQ * kTM + tid] = make_float4(gz5[tid], gz5[kTM + tid], gz5[2 * kTM + tid], 0.f);
    }
    __syncthreads();
    // layers 4..1: g_z_{l-1} = (W_l^T g_z_l) * LeakyReLU'(h_{l-1})
#pragma unroll 1
    for (int l = 4; l >= 1; --l) {
      f32x16 acc[2][2];
      zero_acc(acc);
      gemm_tile(P4 + a.L.bwd[l] / 4, 32, 2 * w, X4, lane, acc);
      __syncthreads();
      const float4* hprev = st + stash_layer_off(a.g.kq0, l);  // h_{l-1}
      float4* gl = gs + (size_t)(l - 1) * kKQ * kTM;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n4 = 64 * w + 32 * nt + 8 * q + 4 * hh;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int cell = (n4 >> 2) * kTM + 32 * mt + j;
            const float4 hv = hprev[cell];
            float4 v;
            v.x = acc[nt][mt][4 * q + 0] * (hv.x > 0.f ? 1.f : kSlope);
            v.y = acc[nt][mt][4 * q + 1] * (hv.y > 0.f ? 1.f : kSlope);
            v.z = acc[nt][mt][4 * q + 2] * (hv.z > 0.f ? 1.f : kSlope);
            v.w = acc[nt][mt][4 * q + 3] * (hv.w > 0.f ? 1.f : kSlope);
            X4[cell] = v;
            gl[cell] = v;
          }
        }
      __syncthreads();
    }
    // layer 0: g_in = W_0^T g_z0  (c_in_pad <= 256 rows = at most 8 row tiles: wave w takes tiles 2w, 2w+1)
    {
      const int KT = a.L.kt[0], rt = 2 * w;
      const bool has0 = rt < KT, has1 = rt + 1 < KT;
      f32x16 acc[2][2];
      zero_acc(acc);
      const float4* Wp = P4 + a.L.bwd[0] / 4;
      if (has1)
        gemm_tile(Wp, 32, rt, X4, lane, acc);
      else if (has0)
        gemm_tile_single(Wp, 32, rt, X4, lane, acc);
      __syncthreads();  // all waves finished reading g_z0
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (nt == 0 ? !has0 : !has1) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k4 = 32 * (rt + nt) + 8 * q + 4 * hh;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            X4[(k4 >> 2) * kTM + 32 * mt + j] = make_float4(acc[nt][mt][4 * q + 0], acc[nt][mt][4 * q + 1],
                                                            acc[nt][mt][4 * q + 2], acc[nt][mt][4 * q + 3]);
        }
      }
      __syncthreads();
    }
    // embedding backward -> g_p[c][m]:  thread (c = w, m = lane), waves 0..2
    if (w < 3) {
      const int nx = a.g.n_freq_xyz, xb = a.g.xyz_base;
      auto gin = [&](int ch) { return Xf[((ch >> 2) * kTM + lane) * 4 + (ch & 3)]; };
      const float* ef = reinterpret_cast<const float*>(st);  // embedding cells, same indexing
      auto emb = [&](int ch) { return ef[((ch >> 2) * kTM + lane) * 4 + (ch & 3)]; };
      float gx = gin(xb + w);
      for (int i = 0; i < nx; ++i) {
        const int cc = xb + 3 + 3 * i + w, cs = xb + 3 + 3 * nx + 3 * i + w;
        // d cos(f x)/dx = -f sin(f x) ; d sin(f x)/dx = f cos(f x)
        gx = __builtin_fmaf(a.freqs_xyz[i], __builtin_fmaf(emb(cc), gin(cs), -emb(cs) * gin(cc)), gx);
      }
      const long long n = n0 + lane;
      if (n < a.n_pix) {
        const long long b = n / a.pix_per_img, hw = n - b * a.pix_per_img;
        const size_t o = (size_t)((b * 3 + w) * a.pix_per_img + hw);
        if (a.g_p_add) gx += a.g_p_add[o];
        a.g_p[o] = gx;
      }
    }
  }
  // flush the last layer's parameter gradients
  {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = dw5[c][e];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if ((tid & 3) == 0) unsafeAtomicAdd(a.gW5 + c * kWidth + 4 * (tid >> 2) + e, v);
      }
    const float v = wave_sum(db5);
    if (lane == 0 && w < 3) unsafeAtomicAdd(a.gb5 + w, v);
  }
}

// ======================================================================================
// backward, dW:  dW_l[n][k] += sum_pixels G_l[n][m] H_{l-1}[k][m]
struct DwArgs {
  const float4* stash;
  const float4* gstash;
  float* gW[kHidden];
  float* gb[kHidden];
  Geometry g;
  int n_tiles, tiles_per_block;
};

// One layer for a run of tiles.  Wave (wr, wc) owns rows [128wr, +128) x cols [128wc, +128):
// accumulator tile (c, c') holds rows 4i+c (i = 0..31) and cols 4j+c'.
template <bool FIRST>
__device__ __forceinline__ void dw_layer(const DwArgs& a, int layer, int tile0, int tile1, int lane, int wr, int wc,
                                         f32x16 (&acc)[4][4], float (&rs)[4]) {
  const int i = lane & 31, h = lane >> 5;
  const int kq_cols = FIRST ? a.g.kq0 : kKQ;     // valid k-quads of H_{l-1}
  const int jq = 32 * wc + i;                    // this lane's k-quad of the B operand
  const bool bvalid = jq < kq_cols;
  const size_t spt = stash_cells_per_tile(a.g.kq0), gpt = gstash_cells_per_tile();
  for (int tile = tile0; tile < tile1; ++tile) {
    const float4* G = a.gstash + (size_t)tile * gpt + (size_t)layer * kKQ * kTM + (size_t)(32 * wr + i) * kTM;
    const float4* H = a.stash + (size_t)tile * spt + stash_layer_off(a.g.kq0, layer) + (size_t)jq * kTM;
#pragma unroll 2
    for (int mb = 0; mb < kTM; mb += 8) {
      float4 av[4], bv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        av[s] = G[mb + 4 * h + s];
        bv[s] = bvalid ? H[mb + 4 * h + s] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float ar[4] = {av[s].x, av[s].y, av[s].z, av[s].w};
        const float br[4] = {bv[s].x, bv[s].y, bv[s].z, bv[s].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          rs[c] += ar[c];
#pragma unroll
          for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[c], br[d], acc[c][d], 0, 0, 0);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_dw_kernel(const DwArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int tile0 = blockIdx.x * a.tiles_per_block;
  int tile1 = tile0 + a.tiles_per_block;
  if (tile1 > a.n_tiles) tile1 = a.n_tiles;
  if (tile0 >= tile1) return;
  const int layer = blockIdx.y;  // 0..4
  f32x16 acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.0f;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  if (layer == 0)
    dw_layer<true>(a, layer, tile0, tile1, lane, wr, wc, acc, rs);
  else
    dw_layer<false>(a, layer, tile0, tile1, lane, wr, wc, acc, rs);
  // flush: element r of acc[c][d] in lane l is row 4*i' + c, col 4*j' + d with
  // i' = (r&3) + 8(r>>2) + 4(l>>5), j' = l&31  (within this wave's 128x128 block)
  const int K = layer == 0 ? a.g.c_in : kWidth;
  float* gW = a.gW[layer];
  const int jp = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ip = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int n = 128 * wr + 4 * ip + c, k = 128 * wc + 4 * jp + d;
        if (k < K) unsafeAtomicAdd(gW + (size_t)n * K + k, acc[c][d][r]);
      }
  if (wc == 0) {
    const int i = lane & 31;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = rs[c];
      v += __shfl_xor(v, 32, 64);
      if (hh == 0) unsafeAtomicAdd(a.gb[layer] + 128 * wr + 4 * i + c, v);
    }
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

constexpr size_t kFwdLds = (size_t)kKQ * kTM * 16 + 4 * kTM * 4 + (3 * kWidth + 4) * 4;

}  // namespace dvd

extern "C" {

int dvd_sf_mlp_in_channels(const dvd_mlp_desc* d) { return d ? dvd::make_geometry(d).c_in : -1; }

size_t dvd_sf_mlp_packed_bytes(const dvd_mlp_desc* d) {
  if (!d) return 0;
  return dvd::make_pack_layout(dvd::make_geometry(d)).total * sizeof(float);
}

size_t dvd_sf_mlp_stash_bytes(const dvd_mlp_desc* d, long long n_pix) {
  if (!d || n_pix <= 0) return 0;
  const long long tiles = (n_pix + dvd::kTM - 1) / dvd::kTM;
  return (size_t)tiles * dvd::stash_cells_per_tile(dvd::make_geometry(d).kq0) * 16;
}

size_t dvd_sf_mlp_gstash_bytes(long long n_pix) {
  if (n_pix <= 0) return 0;
  const long long tiles = (n_pix + dvd::kTM - 1) / dvd::kTM;
  return (size_t)tiles * dvd::gstash_cells_per_tile() * 16;
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
  a.out = static_cast<float*>(packed);
  a.L = make_pack_layout(g);
  a.c_in = g.c_in;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(256, 6), dim3(256), 0, static_cast<hipStream_t>(stream), a);
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
  a.L = make_pack_layout(a.g);
  a.packed = static_cast<const float*>(packed);
  a.p = p;
  a.t = d->time_dependent ? t : nullptr;
  a.freqs_xyz = d->freqs_xyz;
  a.freqs_t = d->freqs_t;
  a.sf_out = sf_out;
  a.p_next = p_next;
  a.acc = acc;
  a.stash = static_cast<float4*>(stash);
  a.n_pix = n_pix;
  a.pix_per_img = pix_per_img;
  a.n_tiles = (int)((n_pix + kTM - 1) / kTM);
  a.t_offset = t_offset;
  a.out_scale = out_scale;
  int cus = dvd_device_cu_count();
  if (cus <= 0) cus = 256;
  const int grid = a.n_tiles < 2 * cus ? a.n_tiles : 2 * cus;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stash) {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
    hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(grid), dim3(256), kFwdLds, s, a);
  } else {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
    hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(grid), dim3(256), kFwdLds, s, a);
  }
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
  a.L = make_pack_layout(a.g);
  DVD_REQUIRE(a.L.kt[0] <= 8, "sf_mlp_bwd_dx: input layer wider than 256 channels");
  a.packed = static_cast<const float*>(packed);
  a.stash = static_cast<const float4*>(stash);
  a.gstash = static_cast<float4*>(gstash);
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
  int cus = dvd_device_cu_count();
  if (cus <= 0) cus = 256;
  const int grid = a.n_tiles < 2 * cus ? a.n_tiles : 2 * cus;
  DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_dx_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
  hipLaunchKernelGGL(mlp_bwd_dx_kernel, dim3(grid), dim3(256), kFwdLds, static_cast<hipStream_t>(stream), a);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_sf_mlp_bwd_dw(const dvd_mlp_desc* d, const void* stash, const void* gstash, long long n_pix,
                      float* const gW[5], float* const gb[5], dvd_stream_t stream) {
  using namespace dvd;
  if (int e = check_desc(d)) return e;
  DVD_REQUIRE(stash && gstash && gW && gb && n_pix > 0, "sf_mlp_bwd_dw: null pointer / size");
  DwArgs a;
  a.g = make_geometry(d);
  a.stash = static_cast<const float4*>(stash);
  a.gstash = static_cast<const float4*>(gstash);
  for (int l = 0; l < kHidden; ++l) {
    DVD_REQUIRE(gW[l] && gb[l], "sf_mlp_bwd_dw: null gradient %d", l);
    a.gW[l] = gW[l];
    a.gb[l] = gb[l];
  }
  a.n_tiles = (int)((n_pix + kTM - 1) / kTM);
  int cus = dvd_device_cu_count();
  if (cus <= 0) cus = 256;
  // one workgroup (4 waves x 256 accumulators) per CU per layer; 5 layers share the CUs in turn
  int blocks = a.n_tiles < cus ? a.n_tiles : cus;
  a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
  blocks = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
  hipLaunchKernelGGL(mlp_bwd_dw_kernel, dim3(blocks, kHidden), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
