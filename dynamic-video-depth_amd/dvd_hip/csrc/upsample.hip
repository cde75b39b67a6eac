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

__global__ __launch_bounds__(256) void upsample_bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    Axis ay, Axis ax, long long planes) {
  const int quads = (ax.n_out + 3) >> 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = planes * ay.n_out * quads;
  if (gid >= total) return;
  const int q = (int)(gid % quads);
  const long long r = gid / quads;
  const int oy = (int)(r % ay.n_out);
  const long long p = r / ay.n_out;
  int y0, y1;
  float wy0, wy1;
  src_index(ay, oy, y0, y1, wy0, wy1);
  const float* r0 = x + (p * ay.n_in + y0) * ax.n_in;
  const float* r1 = x + (p * ay.n_in + y1) * ax.n_in;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = q * 4 + j;
    int x0, x1;
    float wx0, wx1;
    src_index(ax, ox < ax.n_out ? ox : ax.n_out - 1, x0, x1, wx0, wx1);
    // ATen: w_y0 * (w_x0 * a + w_x1 * b) + w_y1 * (w_x0 * c + w_x1 * d)
    o[j] = wy0 * (wx0 * r0[x0] + wx1 * r0[x1]) + wy1 * (wx0 * r1[x0] + wx1 * r1[x1]);
  }
  float* dst = y + (p * ay.n_out + oy) * ax.n_out + q * 4;
  if ((ax.n_out & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < ax.n_out) dst[j] = o[j];
  }
}

// Outputs whose two source taps can include input index i lie in an interval of width < 2/scale + 1
// around i/scale: a bracket of kCand candidates from `lo` on covers it for scale >= 1/3; every
// candidate is then tested exactly (weight 0 if it does not touch i), so the bracket only has to
// be wide enough, never tight.
constexpr int kCand = 8;
__device__ __forceinline__ int dst_lo(const Axis& a, int i) {
  if (a.scale <= 0.0f) return 0;
  const float inv = 1.0f / a.scale;
  const float l = a.align ? ((float)i - 1.0f) * inv : ((float)i - 0.5f) * inv - 0.5f;
  const int lo = (int)floorf(l);
  return lo < 0 ? 0 : lo;
}
__device__ __forceinline__ void axis_weights(const Axis& a, int i, int lo, float w[kCand]) {
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

__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                    Axis ay, Axis ax, long long planes) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = planes * ay.n_in * ax.n_in;
  if (gid >= total) return;
  const int ix = (int)(gid % ax.n_in);
  const long long r = gid / ax.n_in;
  const int iy = (int)(r % ay.n_in);
  const long long p = r / ay.n_in;
  const int ylo = dst_lo(ay, iy), xlo = dst_lo(ax, ix);
  float wy[kCand], wx[kCand];
  axis_weights(ay, iy, ylo, wy);
  axis_weights(ax, ix, xlo, wx);
  const float* g = gy + p * ay.n_out * ax.n_out;
  float acc = 0.0f;
#pragma unroll
  for (int a = 0; a < kCand; ++a) {
    if (wy[a] == 0.0f) continue;
    const float* grow = g + (long long)(ylo + a) * ax.n_out + xlo;
    float row = 0.0f;
#pragma unroll
    for (int b = 0; b < kCand; ++b)
      if (wx[b] != 0.0f) row = __builtin_fmaf(wx[b], grow[b], row);
    acc = __builtin_fmaf(wy[a], row, acc);
  }
  gx[gid] = acc;
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
  const long long total = planes * H_out * ((W_out + 3) / 4);
  DVD_REQUIRE((total + 255) / 256 < (1LL << 31), "upsample fwd: grid too large");
  hipLaunchKernelGGL(dvd::upsample_bilinear_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, ay, ax, planes);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_upsample_bilinear_bwd(const float* gy, float* gx, long long planes, int H_in, int W_in, int H_out, int W_out,
                              int align_corners, dvd_stream_t stream) {
  DVD_REQUIRE(gy && gx, "upsample bwd: null pointer");
  DVD_REQUIRE(planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "upsample bwd: bad shape");
  const dvd::Axis ay = dvd::make_axis(H_in, H_out, align_corners), ax = dvd::make_axis(W_in, W_out, align_corners);
  const long long total = planes * H_in * W_in;
  DVD_REQUIRE((total + 255) / 256 < (1LL << 31), "upsample bwd: grid too large");
  hipLaunchKernelGGL(dvd::upsample_bilinear_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gy, gx, ay, ax, planes);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
