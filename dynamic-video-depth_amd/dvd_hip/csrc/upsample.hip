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

#include "dvd_io.h"

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

// Linear index over (row group, quad): a thread produces 4 adjacent output pixels of kRows consecutive rows.  The column
// indices / weights are computed once per thread, and so is the HORIZONTAL blend of a source row: consecutive output rows
// share their source rows (x2 up-sampling: 8 output rows read 5 source rows), so a thread keeps the blended quads of the
// two source rows it last used and only loads a row it has not seen -- 5 x 8 instead of 8 x 16 tap loads per thread, and
// bit-identical results (ATen's association w_y0 * (w_x0 * a + w_x1 * b) + w_y1 * (w_x0 * c + w_x1 * d) IS "horizontal
// first").  Round 2's kernel re-loaded and re-blended both source rows for every output row and ran at 1.7 TB/s.
// Row index = plane * n_out_y + oy, 32-bit arithmetic only.
constexpr int kRows = 8;
template <class T>
__global__ __launch_bounds__(256) void upsample_bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
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
  unsigned p = rb / (unsigned)ay.n_out;
  int oy = (int)(rb - p * (unsigned)ay.n_out);
  unsigned c0 = 0xffffffffu, c1 = 0xffffffffu;   // source rows (plane * n_in_y + y) whose blends h0 / h1 hold
  float h0[4], h1[4];
  // The eight taps of a quad lie in four CONSECUTIVE source columns whenever the map is an up-sampling by about two (output
  // columns 4q .. 4q+3 read source columns 2q-1 .. 2q+2): one 16-byte (fp16: 8-byte) load of the window x0[0] .. x0[0]+3,
  // clamped into the row, instead of eight 4-byte gathers (round 6: the kernel sat at 2.75 TB/s with its load instructions,
  // not its bytes, as the limit).  Same values, same arithmetic: bit-identical.  Other maps keep the gathers.
  int xs = x0[0] < ax.n_in - 4 ? x0[0] : ax.n_in - 4;
  bool window = ax.n_in >= 4 && xs >= 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) window = window && x0[j] >= xs && x1[j] - xs <= 3;
  int d0[4], d1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d0[j] = x0[j] - xs;
    d1[j] = x1[j] - xs;
  }
  auto pick = [](const float (&w)[4], int d) { return d == 0 ? w[0] : (d == 1 ? w[1] : (d == 2 ? w[2] : w[3])); };
  auto blend = [&](unsigned g, float (&h)[4]) {
    const T* rp = x + (size_t)g * ax.n_in;
    if (window) {
      float w[4];
      if constexpr (sizeof(T) == 4) {
        float4 v;
        __builtin_memcpy(&v, rp + xs, 16);                 // 4-byte aligned: global loads need no more
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
        _Float16 v[4];
        __builtin_memcpy(v, rp + xs, 8);
        w[0] = (float)v[0]; w[1] = (float)v[1]; w[2] = (float)v[2]; w[3] = (float)v[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = wx0[j] * pick(w, d0[j]) + wx1[j] * pick(w, d1[j]);
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = wx0[j] * ldf(rp + x0[j]) + wx1[j] * ldf(rp + x1[j]);
  };
#pragma unroll 2
  for (int k = 0; k < kRows; ++k) {
    const unsigned r = rb + k;
    if (r >= rows) break;
    int y0, y1;
    float wy0, wy1;
    src_index(ay, oy, y0, y1, wy0, wy1);
    const unsigned g0 = p * (unsigned)ay.n_in + (unsigned)y0, g1 = p * (unsigned)ay.n_in + (unsigned)y1;
    if (g0 != c0) {
      if (g0 == c1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) h0[j] = h1[j];
      } else {
        blend(g0, h0);
      }
      c0 = g0;
    }
    if (g1 != c1) {
      if (g1 == c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) h1[j] = h0[j];
      } else {
        blend(g1, h1);
      }
      c1 = g1;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = wy0 * h0[j] + wy1 * h1[j];
    T* dst = y + (size_t)r * ax.n_out + q * 4;
    if (vec) {
      st4(dst, make_float4(o[0], o[1], o[2], o[3]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((int)q * 4 + j < ax.n_out) stf(dst + j, o[j]);
    }
    if (++oy == ay.n_out) {
      oy = 0;
      ++p;
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

// Linear index over (row group of kBRows input rows, input column): the column bracket and its weights are computed once
// per thread; row index = plane * n_in_y + iy.  (General gather: any factor up to 3, any width.  The decoder's x2 maps
// take upsample_bilinear_bwd_sep_kernel below.)
constexpr int kBRows = 4;
template <int kCand, class T>
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx,
                                                                    Axis ay, Axis ax, unsigned rows) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;          // (row group, input column), columns fastest
  const unsigned rg = idx / (unsigned)ax.n_in;
  const unsigned ixu = idx - rg * (unsigned)ax.n_in;
  const unsigned rb = rg * kBRows;
  if (rb >= rows) return;
  const int ix = (int)ixu;
  const int xlo = dst_lo(ax, ix);
  float wx[kCand];
  axis_weights<kCand>(ax, ix, xlo, wx);
#pragma unroll 1
  for (int k = 0; k < kBRows; ++k) {
    const unsigned r = rb + k;
    if (r >= rows) break;
    const unsigned p = r / (unsigned)ay.n_in;
    const int iy = (int)(r - p * (unsigned)ay.n_in);
    const int ylo = dst_lo(ay, iy);
    float wy[kCand];
    axis_weights<kCand>(ay, iy, ylo, wy);
    const T* g = gy + (size_t)p * ay.n_out * ax.n_out;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < kCand; ++a) {
      if (wy[a] == 0.0f) continue;
      const T* grow = g + (size_t)(ylo + a) * ax.n_out + xlo;
      float row = 0.0f;
#pragma unroll
      for (int b = 0; b < kCand; ++b)
        if (wx[b] != 0.0f) row = __builtin_fmaf(wx[b], ldf(grow + b), row);
      acc = __builtin_fmaf(wy[a], row, acc);
    }
    stf(gx + (size_t)r * ax.n_in + ix, acc);
  }
}


// Separable backward for the decoder's maps (n_out <= 2.5 n_in on both axes, W_out a multiple of 4): gx = Wy^T gy Wx.
// A thread owns TWO adjacent input columns and a segment of kSegRows input rows of one plane and walks DOWN the output
// rows that touch the segment: per output row three aligned 16-byte loads cover every output column that can read either
// input column (their exact weights -- zero for the ones that do not -- are computed once per thread), the two horizontal
// sums go into two running accumulators (the input rows y0 and y0 + 1 of that output row), and an input row is stored when
// the walk has passed its last reader.  Per input element ~3.3 16-byte loads instead of the gather's 16 scalar loads;
// the summation order (output rows ascending, columns ascending inside) is fixed: deterministic, no atomics, no memset.
constexpr int kSegRows = 16, kSepCand = 12;
template <class T>
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_sep_kernel(const T* __restrict__ gy, T* __restrict__ gx,
                                                                        Axis ay, Axis ax, unsigned segs, unsigned nseg) {
  const unsigned pairs = (unsigned)(ax.n_in + 1) >> 1;
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;          // (plane, segment, column pair), pairs fastest
  const unsigned sg = idx / pairs;
  if (sg >= segs) return;
  const int ix = (int)(idx - sg * pairs) * 2;
  const unsigned p = sg / nseg;
  const int m0 = (int)(sg - p * nseg) * kSegRows;
  const int m1 = m0 + kSegRows < ay.n_in ? m0 + kSegRows : ay.n_in;
  const bool two = ix + 1 < ax.n_in;
  // column bracket: kSepCand outputs from the 4-aligned column at or below the first reader of column ix
  const int a0 = dst_lo(ax, ix) & ~3;
  float wa[kSepCand], wb[kSepCand];
#pragma unroll
  for (int k = 0; k < kSepCand; ++k) {
    const int o = a0 + k;
    int i0, i1;
    float w0, w1;
    src_index(ax, o < ax.n_out ? o : ax.n_out - 1, i0, i1, w0, w1);
    const bool in = o < ax.n_out;
    wa[k] = in ? (i0 == ix ? w0 : 0.0f) + (i1 == ix ? w1 : 0.0f) : 0.0f;
    wb[k] = in && two ? (i0 == ix + 1 ? w0 : 0.0f) + (i1 == ix + 1 ? w1 : 0.0f) : 0.0f;
  }
  const T* g = gy + (size_t)p * ay.n_out * ax.n_out + a0;
  T* out = gx + (size_t)p * ay.n_in * ax.n_in + ix;
  const bool ld0 = a0 < ax.n_out, ld1 = a0 + 4 < ax.n_out, ld2 = a0 + 8 < ax.n_out;   // W_out % 4 == 0: all or nothing
  int oy = dst_lo(ay, m0);
  int A;                                   // the input row acc_a belongs to (acc_b: A + 1)
  {
    int y1;
    float w0, w1;
    src_index(ay, oy, A, y1, w0, w1);
  }
  float a_a = 0.0f, a_b = 0.0f, b_a = 0.0f, b_b = 0.0f;      // [column a / b]_[row A / A + 1]
  auto flush = [&]() {
    if (A >= m0 && A < m1) {
      stf(out + (size_t)A * ax.n_in, a_a);
      if (two) stf(out + (size_t)A * ax.n_in + 1, b_a);
    }
    a_a = a_b;
    b_a = b_b;
    a_b = b_b = 0.0f;
    ++A;
  };
  const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 n0 = z, n1 = z, n2 = z;           // the NEXT output row's three quads: requested one row ahead of their use
  auto request = [&](int r) {
    const T* row = g + (size_t)r * ax.n_out;
    n0 = ld0 ? ld4(row) : z;
    n1 = ld1 ? ld4(row + 4) : z;
    n2 = ld2 ? ld4(row + 8) : z;
  };
  if (oy < ay.n_out) request(oy);
  for (; oy < ay.n_out; ++oy) {
    int y0, y1;
    float wy0, wy1;
    src_index(ay, oy, y0, y1, wy0, wy1);
    if (y0 >= m1) break;
    const float4 v0 = n0, v1 = n1, v2 = n2;
    if (oy + 1 < ay.n_out) request(oy + 1);
    while (y0 > A) flush();
    const float v[kSepCand] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    float ta = 0.0f, tb = 0.0f;
#pragma unroll
    for (int k = 0; k < kSepCand; ++k) {
      ta = __builtin_fmaf(wa[k], v[k], ta);
      tb = __builtin_fmaf(wb[k], v[k], tb);
    }
    a_a = __builtin_fmaf(wy0, ta, a_a);
    b_a = __builtin_fmaf(wy0, tb, b_a);
    if (y1 != y0) {
      a_b = __builtin_fmaf(wy1, ta, a_b);
      b_b = __builtin_fmaf(wy1, tb, b_b);
    } else {                               // far edge: both taps are row y0
      a_a = __builtin_fmaf(wy1, ta, a_a);
      b_a = __builtin_fmaf(wy1, tb, b_a);
    }
  }
  while (A < m1) flush();
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
  return dvd_upsample_bilinear_fwd_t(x, y, 0, planes, H_in, W_in, H_out, W_out, align_corners, stream);
}

int dvd_upsample_bilinear_fwd_t(const void* x, void* y, int f16, long long planes, int H_in, int W_in, int H_out, int W_out,
                                int align_corners, dvd_stream_t stream) {
  DVD_REQUIRE(x && y, "upsample fwd: null pointer");
  DVD_REQUIRE(planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "upsample fwd: bad shape");
  dvd::bytes_add(DVD_BYTES_UPSAMPLE_FWD, (double)planes * ((double)H_in * W_in + (double)H_out * W_out) * (f16 ? 2 : 4));
  const dvd::Axis ay = dvd::make_axis(H_in, H_out, align_corners), ax = dvd::make_axis(W_in, W_out, align_corners);
  const long long rows = planes * H_out;
  DVD_REQUIRE(rows < (1LL << 32) - 4, "upsample fwd: too many rows");
  const unsigned quads = (unsigned)(W_out + 3) / 4;
  const long long total = (rows + dvd::kRows - 1) / dvd::kRows * quads;
  DVD_REQUIRE(total < (1LL << 32) - 256, "upsample fwd: too large");
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::upsample_bilinear_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<T*>(y), ay, ax,
                                         (unsigned)rows));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_upsample_bilinear_bwd(const float* gy, float* gx, long long planes, int H_in, int W_in, int H_out, int W_out,
                              int align_corners, dvd_stream_t stream) {
  return dvd_upsample_bilinear_bwd_t(gy, gx, 0, planes, H_in, W_in, H_out, W_out, align_corners, stream);
}

int dvd_upsample_bilinear_bwd_t(const void* gy, void* gx, int f16, long long planes, int H_in, int W_in, int H_out, int W_out,
                                int align_corners, dvd_stream_t stream) {
  DVD_REQUIRE(gy && gx, "upsample bwd: null pointer");
  DVD_REQUIRE(planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "upsample bwd: bad shape");
  dvd::bytes_add(DVD_BYTES_UPSAMPLE_BWD, (double)planes * ((double)H_in * W_in + (double)H_out * W_out) * (f16 ? 2 : 4));
  const dvd::Axis ay = dvd::make_axis(H_in, H_out, align_corners), ax = dvd::make_axis(W_in, W_out, align_corners);
  const long long rows = planes * H_in;
  DVD_REQUIRE(rows < (1LL << 32) - 4, "upsample bwd: too many rows");
  const long long total = (rows + dvd::kBRows - 1) / dvd::kBRows * W_in;
  DVD_REQUIRE(total < (1LL << 32) - 256, "upsample bwd: too large");
  // the candidate bracket must cover every output that reads an input pixel (or simply all outputs of a short axis)
  const auto covered = [](const dvd::Axis& a, int k, float smin) { return a.scale >= smin || a.n_out <= k; };
  DVD_REQUIRE(covered(ay, 8, 1.0f / 3.0f) && covered(ax, 8, 1.0f / 3.0f), "upsample bwd: up-sampling factors above 3 are not covered");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (covered(ay, 6, 0.4f) && covered(ax, 6, 0.4f) && ax.scale >= 0.4f && (W_out & 3) == 0 && H_out > H_in) {
    // the decoder's x2 maps: separable walk (a segment's first reader row comes from dst_lo, which needs scale > 0)
    const long long nseg = (H_in + dvd::kSegRows - 1) / dvd::kSegRows, segs = planes * nseg;
    const long long threads = segs * ((W_in + 1) / 2);
    DVD_REQUIRE(threads < (1LL << 32) - 256, "upsample bwd: too large");
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::upsample_bilinear_bwd_sep_kernel<T>, dim3((unsigned)((threads + 255) / 256)),
                                           dim3(256), 0, s, static_cast<const T*>(gy), static_cast<T*>(gx), ay, ax, (unsigned)segs,
                                           (unsigned)nseg));
    DVD_LAUNCH_OK();
    return DVD_OK;
  }
  const dim3 grid((unsigned)((total + 255) / 256));
  if (covered(ay, 6, 0.4f) && covered(ax, 6, 0.4f))
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::upsample_bilinear_bwd_kernel<6, T>), grid, dim3(256), 0, s,
                                           static_cast<const T*>(gy), static_cast<T*>(gx), ay, ax, (unsigned)rows));
  else
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::upsample_bilinear_bwd_kernel<8, T>), grid, dim3(256), 0, s,
                                           static_cast<const T*>(gy), static_cast<T*>(gx), ay, ax, (unsigned)rows));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
