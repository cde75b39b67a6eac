// Bilinear up-sampling of NCHW fp32 feature maps and its backward (gfx950).
//
// What it replaces: the two interpolation flavours of the MiDaS decoder (reference:
// third_party/midas_blocks.py:71-99 `Interpolate` -> F.interpolate(scale_factor=2, mode='bilinear',
// align_corners=False) in the output head, third_party/MiDaS.py:190; and :164-166
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) at the end of every
// FeatureFusionBlock).  Source index and weights follow ATen's upsample_bilinear2d exactly:
//   align_corners: src = dst * (in - 1) / (out - 1);   else: src = max(0, (dst + 0.5) * scale - 0.5)
//   with scale = 1 / scale_factor when a scale factor was given (here: in / out),
//   i0 = floor(src), i1 = min(i0 + 1, in - 1), w1 = src - i0, w0 = 1 - w1.
//
// Roofline: HBM -- 4 B written per output element + 1 B read (x2 up-sampling); ATen's generic
// kernel reaches ~0.9 TB/s on the decoder's two big maps (1.5 ms for 16x256x96x168 -> 192x336,
// profiles/r01_bench_kernel_trace_summary.txt).  Forward: a thread produces 4 adjacent output
// pixels of one row (one 16-byte store); the two source rows it touches are L1/L2 resident.
// Backward: a GATHER over the (at most 3x3) output pixels that read an input pixel, so it is
// deterministic and needs neither atomics nor a memset.

#include "dvd_common.h"

namespace dvd {

struct Axis {
  int n_in, n_out;
  float scale;
  int align;
};
__device__ __forceinline__ void src_index(const Axis& a, int dst, int& i0, int& i1, float& w0, float& w1) {
  float s;
  if (a.align) {
    s = a.scale * (float)dst;
  } else {
    s = a.scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
  }
  i0 = (int)s;                       // s >= 0: truncation == floor
  i0 = i0 < a.n_in - 1 ? i0 : a.n_in - 1;
  i1 = i0 + (i0 < a.n_in - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.0f - w1;
}

// Linear index over (row group, quad); a thread produces 4 adjacent output pixels of kRows consecutive rows, so the
// column indices / weights are computed once per kRows rows (the kernel is VALU-bound otherwise: ~150 instructions per
// 16-byte store); row index = plane * n_out_y + oy, 32-bit arithmetic only (the first version decomposed a 64-bit linear
// index with three 64-bit divisions per thread and ran at 0.9 TB/s).
constexpr int kRows = 4;
__global__ __launch_bounds__(256) void upsample_bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    Axis ay, Axis ax, unsigned rows) {
  const unsigned quads = (unsigned)(ax.n_out + 3) >> 2;
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;          // (row group, quad), quads fastest: no idle lanes in
  const unsigned rg = idx / quads;                               // rows that are not a multiple of 64 quads wide
  const unsigned q = idx - rg * quads;
  const unsigned rb = rg * kRows;
  if (rb >= rows) return;
  int x0[4], x1[4];
  float wx0[4], wx1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = (int)q * 4 + j;
    src_index(ax, ox < ax.n_out ? ox : ax.n_out - 1, x0[j], x1[j], wx0[j], wx1[j]);
  }
  const bool vec = (ax.n_out & 3) == 0;
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const unsigned r = rb + k;
    if (r >= rows) break;
    const unsigned p = r / (unsigned)ay.n_out;
    const int oy = (int)(r - p * (unsigned)ay.n_out);
    int y0, y1;
    float wy0, wy1;
    src_index(ay, oy, y0, y1, wy0, wy1);
    const float* r0 = x + ((size_t)p * ay.n_in + y0) * ax.n_in;
    const float* r1 = x + ((size_t)p * ay.n_in + y1) * ax.n_in;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)   // ATen: w_y0 * (w_x0 * a + w_x1 * b) + w_y1 * (w_x0 * c + w_x1 * d)
      o[j] = wy0 * (wx0[j] * r0[x0[j]] + wx1[j] * r0[x1[j]]) + wy1 * (wx0[j] * r1[x0[j]] + wx1[j] * r1[x1[j]]);
    float* dst = y + (size_t)r * ax.n_out + q * 4;
    if (vec) {
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((int)q * 4 + j < ax.n_out) dst[j] = o[j];
    }
  }
}

// Outputs whose two source taps can include input index i lie in an interval of width < 2/scale + 1
// around i/scale: a bracket of kCand candidates from `lo` on covers it for scale >= 1/3; every
// candidate is then tested exactly (weight 0 if it does not touch i), so the bracket only has to
// be wide enough, never tight.
// A bracket of k candidates from lo = floor(L) covers it when frac(L) + 2 / scale <= k: k = 6 for scale >= 0.4 (the x2
// up-sampling of the decoder, either align_corners flavour), k = 8 for scale >= 1/3.
__device__ __forceinline__ int dst_lo(const Axis& a, int i) {
  if (a.scale <= 0.0f) return 0;
  const float inv = 1.0f / a.scale;
  const float l = a.align ? ((float)i - 1.0f) * inv : ((float)i - 0.5f) * inv - 0.5f;
  const int lo = (int)floorf(l);
  return lo < 0 ? 0 : lo;
}
template <int kCand>
__device__ __forceinline__ void axis_weights(const Axis& a, int i, int lo, float (&w)[kCand]) {
#pragma unroll
  for (int k = 0; k < kCand; ++k) {
    const int o = lo + k;
    int i0, i1;
    float w0, w1;
    src_index(a, o < a.n_out ? o : a.n_out - 1, i0, i1, w0, w1);
    const float v = (i0 == i ? w0 : 0.0f) + (i1 == i ? w1 : 0.0f);   // at the far edge i1 == i0: both taps land on i
    w[k] = o < a.n_out ? v : 0.0f;
  }
}

// Linear index over (row group of kRows input rows, input column): the column bracket and its weights are computed once
// per thread; row index = plane * n_in_y + iy.
template <int kCand>
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                    Axis ay, Axis ax, unsigned rows) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;          // (row group, input column), columns fastest
  const unsigned rg = idx / (unsigned)ax.n_in;
  const unsigned ixu = idx - rg * (unsigned)ax.n_in;
  const unsigned rb = rg * kRows;
  if (rb >= rows) return;
  const int ix = (int)ixu;
  const int xlo = dst_lo(ax, ix);
  float wx[kCand];
  axis_weights<kCand>(ax, ix, xlo, wx);
#pragma unroll 1
  for (int k = 0; k < kRows; ++k) {
    const unsigned r = rb + k;
    if (r >= rows) break;
    const unsigned p = r / (unsigned)ay.n_in;
    const int iy = (int)(r - p * (unsigned)ay.n_in);
    const int ylo = dst_lo(ay, iy);
    float wy[kCand];
    axis_weights<kCand>(ay, iy, ylo, wy);
    const float* g = gy + (size_t)p * ay.n_out * ax.n_out;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < kCand; ++a) {
      if (wy[a] == 0.0f) continue;
      const float* grow = g + (size_t)(ylo + a) * ax.n_out + xlo;
      float row = 0.0f;
#pragma unroll
      for (int b = 0; b < kCand; ++b)
        if (wx[b] != 0.0f) row = __builtin_fmaf(wx[b], grow[b], row);
      acc = __builtin_fmaf(wy[a], row, acc);
    }
    gx[(size_t)r * ax.n_in + ix] = acc;
  }
}

static Axis make_axis(int n_in, int n_out, int align) {
  Axis a;
  a.n_in = n_in;
  a.n_out = n_out;
  a.align = align;
  if (align)
    a.scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
  else
    a.scale = (float)n_in / (float)n_out;   // == 1 / scale_factor for the integer factors used here
  return a;
}

}  // namespace dvd

extern "C" {

int dvd_upsample_bilinear_fwd(const float* x, float* y, long long planes, int H_in, int W_in, int H_out, int W_out,
                              int align_corners, dvd_stream_t stream) {
  DVD_REQUIRE(x && y, "upsample fwd: null pointer");
  DVD_REQUIRE(planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "upsample fwd: bad shape");
  const dvd::Axis ay = dvd::make_axis(H_in, H_out, align_corners), ax = dvd::make_axis(W_in, W_out, align_corners);
  const long long rows = planes * H_out;
  DVD_REQUIRE(rows < (1LL << 32) - 4, "upsample fwd: too many rows");
  const unsigned quads = (unsigned)(W_out + 3) / 4;
  const long long total = (rows + dvd::kRows - 1) / dvd::kRows * quads;
  DVD_REQUIRE(total < (1LL << 32) - 256, "upsample fwd: too large");
  hipLaunchKernelGGL(dvd::upsample_bilinear_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, ay, ax, (unsigned)rows);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_upsample_bilinear_bwd(const float* gy, float* gx, long long planes, int H_in, int W_in, int H_out, int W_out,
                              int align_corners, dvd_stream_t stream) {
  DVD_REQUIRE(gy && gx, "upsample bwd: null pointer");
  DVD_REQUIRE(planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "upsample bwd: bad shape");
  const dvd::Axis ay = dvd::make_axis(H_in, H_out, align_corners), ax = dvd::make_axis(W_in, W_out, align_corners);
  const long long rows = planes * H_in;
  DVD_REQUIRE(rows < (1LL << 32) - 4, "upsample bwd: too many rows");
  const long long total = (rows + dvd::kRows - 1) / dvd::kRows * W_in;
  DVD_REQUIRE(total < (1LL << 32) - 256, "upsample bwd: too large");
  // the candidate bracket must cover every output that reads an input pixel (or simply all outputs of a short axis)
  const auto covered = [](const dvd::Axis& a, int k, float smin) { return a.scale >= smin || a.n_out <= k; };
  DVD_REQUIRE(covered(ay, 8, 1.0f / 3.0f) && covered(ax, 8, 1.0f / 3.0f), "upsample bwd: up-sampling factors above 3 are not covered");
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (covered(ay, 6, 0.4f) && covered(ax, 6, 0.4f))
    hipLaunchKernelGGL(dvd::upsample_bilinear_bwd_kernel<6>, grid, dim3(256), 0, s, gy, gx, ay, ax, (unsigned)rows);
  else
    hipLaunchKernelGGL(dvd::upsample_bilinear_bwd_kernel<8>, grid, dim3(256), 0, s, gy, gx, ay, ax, (unsigned)rows);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
