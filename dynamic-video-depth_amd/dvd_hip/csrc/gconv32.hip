// Grouped 3x3 convolution with 32 channels per group (stride 1, pad 1, NCHW fp32) on fp32 MFMA for
// gfx950: forward, backward-data, backward-weight.
//
// What it replaces: the `conv2` of the stride-1 bottlenecks of ResNeXt-101 32x8d stage 3 inside the
// MiDaS encoder (reference: third_party/midas_blocks.py:35-50 -> torchvision ResNet(Bottleneck,
// [3,4,23,3], groups=32, width_per_group=8); stage 3 has width 1024 = 32 groups x 32; 22 of its 23
// blocks are stride 1).  MIOpen's immediate mode runs these per image as im2col + small GEMMs
// (+ col2im backward): profiles/r01_bench_kernel_trace_summary.txt shows ~16 000 launches of
// 15-20 us per step (Im2d2Col, Cijk_*MT32x32x64, MT16x32x128, Col2Im2dU: 14 % of the GPU time).
//
// Per group the convolution is nine 32x32 (co x ci) matrices applied to shifted copies of a
// 32 x pixels activation matrix: exactly the M = 32, N = 32 tile of v_mfma_f32_32x32x2_f32
// (A = W_tap[co][ci], lane l holds [l&31][l>>5]; B = X[ci][pixel + tap], lane l holds
// [l>>5][l&31]; 16 accumulator registers: row = (r&3) + 8 (r>>2) + 4 (l>>5), col = l&31).
// Roofline: fp32 MFMA (157 TF): 18 432 FLOP per pixel-group against 256 B moved.
//   * forward / backward-data (one kernel; backward-data swaps the channel roles and flips the
//     taps when the weights are packed): a 256-thread block owns a band of full rows of one
//     (image, group).  The 32 input planes of the band (+1 px halo) sit in LDS; each wave keeps
//     the group's 144 A fragments (9 taps x 16 k-steps) in registers, loaded as 36 x 16 B from the
//     fragment-ordered pack, and walks over 32-pixel tiles: 144 dependent MFMAs per tile, the B
//     operand of each one is a single ds_read_b32 at an immediate offset from a per-lane base.
//   * backward-weight: dW_tap[co][ci] = sum_pixels gy[co][p] x[ci][p + tap] is the same tile with
//     K = pixels: A = gy[co][p], B = x[ci][p + tap] (both read across channel planes, whose LDS
//     stride is odd, so the 32 lanes of a half-wave hit 32 different banks).  A 192-thread block
//     owns a band of one (image, group); wave w owns kernel row ky = w (three 32x32
//     accumulators) and walks over all pixel pairs of the band: 3 independent MFMAs per 4 LDS
//     reads, no cross-wave reduction.  Per-block partial sums go to a workspace and are reduced
//     in a fixed order (deterministic, no atomics).

#include "dvd_common.h"

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kG32 = 32;
constexpr int kFragsPerLane = 9 * 16;   // taps x k-steps

// packed[g][lane][t*16 + ks]:  forward:  W[g*32 + (lane&31)][2 ks + (lane>>5)][t]
//                              transposed (backward-data): W[g*32 + 2 ks + (lane>>5)][lane&31][8 - t]
__global__ __launch_bounds__(256) void gconv32_pack_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                                           int G, int transposed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G * 64 * kFragsPerLane) return;
  const int f = i % kFragsPerLane, lane = (i / kFragsPerLane) & 63, g = i / (kFragsPerLane * 64);
  const int t = f / 16, ks = f - t * 16;
  const int m = lane & 31, k = 2 * ks + (lane >> 5);
  const int co = transposed ? k : m, ci = transposed ? m : k, tap = transposed ? 8 - t : t;
  packed[i] = w[((size_t)(g * kG32 + co) * kG32 + ci) * 9 + tap];
}

constexpr int kFillU = 8;

// s_x[c][yy][xx] (plane stride PS, row stride WS) <- rows y0-1 .. y0+rows of the 32 planes at `src`, with a
// 1-pixel zero halo; NW waves, wave w takes row ids w*kFillU .. +kFillU-1, then strides by NW*kFillU
template <int NW>
__device__ __forceinline__ void fill_band(float* s_x, int PS, int WS, const float* __restrict__ src, size_t plane,
                                          int H, int W, int TH, int rows, int y0, int wave, int lane) {
  const int nrow = kG32 * (TH + 2);
  for (int rho0 = wave * kFillU; rho0 < nrow; rho0 += NW * kFillU) {
    for (int xx = lane; xx < WS; xx += 64) {
      const int gx = xx - 1;
      float v[kFillU];
#pragma unroll
      for (int u = 0; u < kFillU; ++u) {
        const int rho = rho0 + u;
        const int c = rho / (TH + 2), yy = rho - c * (TH + 2);
        const int gy = y0 + yy - 1;
        const bool ok = rho < nrow && yy < rows + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[u] = ok ? src[(size_t)c * plane + (size_t)gy * W + gx] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kFillU; ++u) {
        const int rho = rho0 + u;
        if (rho < nrow) {
          const int c = rho / (TH + 2), yy = rho - c * (TH + 2);
          s_x[c * PS + yy * WS + xx] = v[u];
        }
      }
    }
  }
}

// out[n, g*32 + m, p] = sum_{t, k} A_t[m][k] * in[n, g*32 + k, p + tap_t]
__global__ __launch_bounds__(256, 2) void gconv32_mfma_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ packed,
                                                              float* __restrict__ out, int C, int H, int W, int TH,
                                                              int bands) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];   // [32][TH + 2][WS], plane stride PS
  const int band = blockIdx.x % bands, g = blockIdx.y, n = blockIdx.z;
  const int WS = W + 2;
  const int PS = (TH + 2) * WS;
  const int y0 = band * TH;
  const int rows = (H - y0) < TH ? (H - y0) : TH;
  const size_t plane = (size_t)H * W;
  const float* inb = in + ((size_t)n * C + (size_t)g * kG32) * plane;
  float* outb = out + ((size_t)n * C + (size_t)g * kG32) * plane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // A fragments of the group: 144 registers per lane, 36 x 16-byte loads
  float a[kFragsPerLane];
  {
    const float4* src = reinterpret_cast<const float4*>(packed + ((size_t)g * 64 + lane) * kFragsPerLane);
#pragma unroll
    for (int i = 0; i < kFragsPerLane / 4; ++i) {
      const float4 v = src[i];
      a[4 * i] = v.x;
      a[4 * i + 1] = v.y;
      a[4 * i + 2] = v.z;
      a[4 * i + 3] = v.w;
    }
  }
  // input band with halo, zero outside the image
  // LDS fill: one wave per (channel, row) of the haloed band, lanes along x.  kFillU rows are requested
  // before the first one is stored, so a wave has kFillU loads in flight instead of one.
  fill_band<4>(s_x, PS, WS, inb, plane, H, W, TH, rows, y0, wave, lane);
  __syncthreads();

  const int npix = rows * W;
  const int ntiles = (npix + 31) >> 5;
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int p = tile * 32 + (lane & 31);
    const int pc = p < npix ? p : npix - 1;      // partial last tile: clamp the read, mask the store
    const int py = pc / W, px = pc - py * W;
    // B operand base: channel (lane>>5), centre pixel (py + 1, px + 1) of the haloed band
    const float* base = s_x + (lane >> 5) * PS + (py + 1) * WS + (px + 1);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int off = (t / 3 - 1) * WS + (t % 3 - 1);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float b = base[off + 2 * ks * PS];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t * 16 + ks], b, acc, 0, 0, 0);
      }
    }
    if (p < npix) {
      float* dst = outb + (size_t)y0 * W + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(size_t)m * plane] = acc[r];
      }
    }
  }
}

// partial[rec][t][co][ci]; rec = (n * bands + band) * G + g.  192 threads: wave w owns the taps of
// kernel row ky = w (three accumulators), so no cross-wave reduction is needed.
__global__ __launch_bounds__(192, 2) void gconv32_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                               float* __restrict__ partial, int C, int H, int W, int TH,
                                                               int bands, int G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int band = blockIdx.x % bands, g = blockIdx.y, n = blockIdx.z;
  const int WS = W + 2;
  const int PSX = ((TH + 2) * WS) | 1;   // odd plane strides: the 32 lanes of a half-wave (32 channels,
  const int PSG = (TH * W) | 1;          // same pixel) hit 32 different banks
  float* s_x = smem;
  float* s_g = smem + kG32 * PSX;
  const int y0 = band * TH;
  const int rows = (H - y0) < TH ? (H - y0) : TH;
  const size_t plane = (size_t)H * W;
  const float* xb = x + ((size_t)n * C + (size_t)g * kG32) * plane;
  const float* gb = gy + ((size_t)n * C + (size_t)g * kG32) * plane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  fill_band<3>(s_x, PSX, WS, xb, plane, H, W, TH, rows, y0, wave, lane);
  const int npix = rows * W;
  for (int c0 = wave * kFillU; c0 < kG32; c0 += 3 * kFillU) {   // the band's rows of a channel are contiguous in memory
    for (int r = lane; r < TH * W; r += 64) {
      float v[kFillU];
#pragma unroll
      for (int u = 0; u < kFillU; ++u) {
        const int c = c0 + u;
        v[u] = (c < kG32 && r < npix) ? gb[(size_t)c * plane + (size_t)y0 * W + r] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kFillU; ++u)
        if (c0 + u < kG32) s_g[(c0 + u) * PSG + r] = v[u];
    }
  }
  __syncthreads();

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float* ga = s_g + (lane & 31) * PSG;
  const float* xa = s_x + (lane & 31) * PSX + (wave - 1) * WS;   // kernel row ky = wave: dy = ky - 1
  // pixel pairs (px, px+1) of one row -> k = 0, 1 of one MFMA; an odd last column pairs with gy = 0
  const int pw = (W + 1) >> 1;
  const int npairs = rows * pw;
  for (int q = 0; q < npairs; ++q) {
    const int py = q / pw, px = (q - py * pw) * 2 + (lane >> 5);
    const bool in = px < W;
    const float av = in ? ga[py * W + px] : 0.0f;
    const float* xc = xa + (py + 1) * WS + (in ? px : W - 1) + 1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xc[kx - 1], acc[kx], 0, 0, 0);
  }
  float* dst = partial + (((size_t)n * bands + band) * G + g) * (9 * kG32 * kG32);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      dst[((wave * 3 + kx) * kG32 + co) * kG32 + (lane & 31)] = acc[kx][r];
    }
}

// gw[g*32 + co][ci][t] (+)= sum over records (n, band) of partial[rec][g][t][co][ci], ascending
__global__ __launch_bounds__(256) void gconv32_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ gw, int G, int n_outer,
                                                                   int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // over G * 9 * 32 * 32 in [g][t][co][ci] order
  const int per_g = 9 * kG32 * kG32;
  if (i >= G * per_g) return;
  const int g = i / per_g, r = i - g * per_g, t = r / (kG32 * kG32), cc = r - t * (kG32 * kG32);
  const int co = cc / kG32, ci = cc - co * kG32;
  float s0 = 0.0f, s1 = 0.0f;
  int o = 0;
  for (; o + 1 < n_outer; o += 2) {
    s0 += partial[((size_t)o * G + g) * per_g + r];
    s1 += partial[((size_t)(o + 1) * G + g) * per_g + r];
  }
  if (o < n_outer) s0 += partial[((size_t)o * G + g) * per_g + r];
  const size_t dst = ((size_t)(g * kG32 + co) * kG32 + ci) * 9 + t;
  const float s = s0 + s1;
  gw[dst] = accumulate ? gw[dst] + s : s;
}

constexpr int kLdsBudgetFloats = 72 * 1024 / 4;   // two blocks per CU
static int lds_floats(int th, int W, bool wgrad) {
  return wgrad ? kG32 * (((th + 2) * (W + 2)) | 1) + kG32 * ((th * W) | 1) : kG32 * (th + 2) * (W + 2);
}
// rows per band: the largest band that fits the LDS budget, then evened out over the bands
static int band_rows(int H, int W, bool wgrad) {
  int th = H < 16 ? H : 16;
  while (th > 1 && lds_floats(th, W, wgrad) > kLdsBudgetFloats) --th;
  const int bands = (H + th - 1) / th;
  return (H + bands - 1) / bands;
}

static int check32(int N, int C, int H, int W) {
  DVD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "gconv32: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  DVD_REQUIRE(C % kG32 == 0, "gconv32: C=%d is not a multiple of 32 (32 channels per group)", C);
  DVD_REQUIRE(C / kG32 <= 65535 && N <= 65535, "gconv32: too many groups / images for the grid");
  DVD_REQUIRE(lds_floats(1, W, true) <= 2 * kLdsBudgetFloats, "gconv32: rows of %d pixels do not fit the LDS band", W);
  return DVD_OK;
}

static int run_fwd(const float* in, const float* w, float* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                   int transposed, hipStream_t stream) {
  if (int e = check32(N, C, H, W)) return e;
  DVD_REQUIRE(in && w && out && ws, "gconv32: null pointer");
  const int G = C / kG32;
  const size_t need = (size_t)G * 64 * kFragsPerLane * sizeof(float);
  if (ws_bytes < need) {
    set_error("gconv32: workspace %zu < %zu bytes", ws_bytes, need);
    return DVD_ENOSPC;
  }
  float* packed = static_cast<float*>(ws);
  hipLaunchKernelGGL(gconv32_pack_kernel, dim3((G * 64 * kFragsPerLane + 255) / 256), dim3(256), 0, stream, w, packed,
                     G, transposed);
  DVD_LAUNCH_OK();
  const int TH = band_rows(H, W, false), bands = (H + TH - 1) / TH;
  const size_t lds = (size_t)kG32 * (TH + 2) * (W + 2) * sizeof(float);
  DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gconv32_mfma_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(gconv32_mfma_kernel, dim3(bands, G, N), dim3(256), lds, stream, in, packed, out, C, H, W, TH,
                     bands);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // namespace dvd

extern "C" {

size_t dvd_gconv3x3_c32_workspace_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || C % dvd::kG32) return 0;
  const size_t G = C / dvd::kG32;
  const size_t pack = G * 64 * dvd::kFragsPerLane * sizeof(float);
  const int TH = dvd::band_rows(H, W, true), bands = (H + TH - 1) / TH;
  const size_t part = (size_t)N * bands * G * 9 * dvd::kG32 * dvd::kG32 * sizeof(float);
  return pack > part ? pack : part;
}

int dvd_gconv3x3_c32_fwd(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes, int N,
                         int C, int H, int W, dvd_stream_t stream) {
  return dvd::run_fwd(x, w, y, workspace, workspace_bytes, N, C, H, W, 0, static_cast<hipStream_t>(stream));
}

int dvd_gconv3x3_c32_bwd_data(const float* gy, const float* w, float* gx, void* workspace, size_t workspace_bytes,
                              int N, int C, int H, int W, dvd_stream_t stream) {
  return dvd::run_fwd(gy, w, gx, workspace, workspace_bytes, N, C, H, W, 1, static_cast<hipStream_t>(stream));
}

int dvd_gconv3x3_c32_bwd_weight(const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                                size_t workspace_bytes, int N, int C, int H, int W, dvd_stream_t stream) {
  if (int e = dvd::check32(N, C, H, W)) return e;
  DVD_REQUIRE(x && gy && gw && workspace, "gconv32 bwd_weight: null pointer");
  const size_t need = dvd_gconv3x3_c32_workspace_bytes(N, C, H, W);
  if (workspace_bytes < need) {
    dvd::set_error("gconv32 bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  const int G = C / dvd::kG32;
  const int TH = dvd::band_rows(H, W, true), bands = (H + TH - 1) / TH;
  const size_t lds = ((size_t)dvd::kG32 * (((TH + 2) * (W + 2)) | 1) + (size_t)dvd::kG32 * ((TH * W) | 1)) * sizeof(float);
  hipStream_t stream_ = static_cast<hipStream_t>(stream);
  DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(dvd::gconv32_wgrad_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(dvd::gconv32_wgrad_kernel, dim3(bands, G, N), dim3(192), lds, stream_, x, gy,
                     static_cast<float*>(workspace), C, H, W, TH, bands, G);
  DVD_LAUNCH_OK();
  const int total = G * 9 * dvd::kG32 * dvd::kG32;
  hipLaunchKernelGGL(dvd::gconv32_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream_,
                     static_cast<const float*>(workspace), gw, G, N * bands, accumulate);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
