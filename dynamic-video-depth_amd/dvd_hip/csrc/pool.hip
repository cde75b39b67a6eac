// 3x3 / stride 2 / padding 1 max-pooling of the ResNeXt stem, forward and backward (gfx950).
//
// What it replaces (reference, /root/reference): `pretrained.layer1[3]` = torchvision's nn.MaxPool2d(3, 2, 1) behind the 7x7
// stem (third_party/midas_blocks.py:35-45 builds `layer1 = Sequential(conv1, bn1, relu, maxpool, layer1)`); round 3 ran it
// on ATen (max_pool2d_with_indices / its backward, 6 ms of a step).
//
// Semantics = ATen's: the window of output (oy, ox) covers input rows 2 oy - 1 .. 2 oy + 1 (clipped to the image: padding
// never wins), scanned row-major; a later element replaces the maximum only if it is strictly greater (or NaN), so the FIRST
// maximum of a window takes the gradient -- after the stem's ReLU whole windows are 0 and the tie rule decides where the
// gradient goes.  The forward stores the winner's window position (0..8) in one byte per output; the backward is a gather
// over the (at most four) windows that contain an input pixel: deterministic, no atomics, no memset.
//
// HBM bound.  The forward writes the storage the network continues in (fp32, or _Float16 with fp16 activation storage -- the
// cast of BASELINE configs[4] is fused here); the backward reads that storage and writes the fp32 gradient of the stem times
// *out_scale (1 / loss scale of the fp16 gradients, csrc/a16.hip).
#include "dvd_io.h"

namespace dvd {

template <class T>
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                             unsigned char* __restrict__ idx, int H, int W, int Ho, int Wo,
                                                             long long total) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % Wo);
  const long long r = i / Wo;
  const int oy = (int)(r % Ho);
  const long long pl = r / Ho;
  const float* xp = x + pl * H * W;
  float best = -INFINITY;
  int bi = 0;
  bool any = false;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const float v = xp[(size_t)iy * W + ix];
        if (!any || v > best || v != v) {
          best = v;
          bi = ky * 3 + kx;
          any = true;
        }
      }
    }
  }
  stf(y + i, best);
  idx[i] = (unsigned char)bi;
}

template <class T>
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const T* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                             float* __restrict__ gx, int H, int W, int Ho, int Wo,
                                                             long long total, const float* __restrict__ out_scale) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long long r = i / W;
  const int y = (int)(r % H);
  const long long pl = r / H;
  const T* gp = gy + pl * Ho * Wo;
  const unsigned char* ip = idx + pl * Ho * Wo;
  const int oy0 = y >> 1, ox0 = x >> 1;              // windows: oy0 (and oy0 + 1 for odd y), likewise in x
  float s = 0.0f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int oy = oy0 + dy;
    if (oy >= Ho || (dy == 1 && !(y & 1))) continue;
    const int ky = y - (2 * oy - 1);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int ox = ox0 + dx;
      if (ox >= Wo || (dx == 1 && !(x & 1))) continue;
      const int kx = x - (2 * ox - 1);
      const size_t o = (size_t)oy * Wo + ox;
      if (ip[o] == ky * 3 + kx) s += ldf(gp + o);
    }
  }
  gx[i] = s * (out_scale ? out_scale[0] : 1.0f);
}

// Round 6: a thread owns the 2 x 2 input pixels (2 oy + {0, 1}, 2 ox + {0, 1}) -- together they are fed by the four windows
// (oy + {0, 1}, ox + {0, 1}), whose winner bytes and gradients it loads ONCE (the pixel-per-thread kernel above loads them per
// pixel: 9 byte + up to 9 dword loads per 2 x 2 block against 4 + 4, behind a 64-bit division per pixel; 1.2 ms per launch =
// 0.85 TB/s).  Every pixel adds its windows in the order of the kernel above: bit-identical.
template <class T>
__global__ __launch_bounds__(256) void maxpool3s2_bwd2_kernel(const T* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                              float* __restrict__ gx, int H, int W, int Ho, int Wo,
                                                              unsigned total_blocks2, const float* __restrict__ out_scale) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;      // over (plane, oy, ox) of the 2 x 2 blocks: Hb x Wb per plane
  if (i >= total_blocks2) return;
  const unsigned Wb = (unsigned)(W + 1) >> 1, Hb = (unsigned)(H + 1) >> 1;
  const unsigned ox = i % Wb, r = i / Wb, oy = r % Hb, pl = r / Hb;
  const T* gp = gy + (size_t)pl * Ho * Wo;
  const unsigned char* ip = idx + (size_t)pl * Ho * Wo;
  const bool r1 = (int)oy + 1 < Ho, c1 = (int)ox + 1 < Wo;          // windows of the next output row / column exist
  const size_t o00 = (size_t)oy * Wo + ox;
  // (oy < Ho and ox < Wo always: Hb == Ho and Wb == Wo for this pooling geometry)
  const int k00 = ip[o00], k01 = c1 ? ip[o00 + 1] : -1, k10 = r1 ? ip[o00 + Wo] : -1, k11 = (r1 && c1) ? ip[o00 + Wo + 1] : -1;
  const float g00 = ldf(gp + o00), g01 = c1 ? ldf(gp + o00 + 1) : 0.0f, g10 = r1 ? ldf(gp + o00 + Wo) : 0.0f,
              g11 = (r1 && c1) ? ldf(gp + o00 + Wo + 1) : 0.0f;
  const float sc = out_scale ? out_scale[0] : 1.0f;
  auto take = [](float s, int k, int tap, float g) { return k == tap ? s + g : s; };
  // pixel (even, even): window (oy, ox) tap 4 | (even, odd): (oy, ox) tap 5, (oy, ox + 1) tap 3
  // pixel (odd, even): (oy, ox) tap 7, (oy + 1, ox) tap 1 | (odd, odd): taps 8, 6, 2, 0 of the four windows
  const float p00 = take(0.0f, k00, 4, g00);
  const float p01 = take(take(0.0f, k00, 5, g00), k01, 3, g01);
  const float p10 = take(take(0.0f, k00, 7, g00), k10, 1, g10);
  const float p11 = take(take(take(take(0.0f, k00, 8, g00), k01, 6, g01), k10, 2, g10), k11, 0, g11);
  float* dst = gx + ((size_t)pl * H + 2 * oy) * W + 2 * ox;
  const bool xin = (int)(2 * ox + 1) < W, yin = (int)(2 * oy + 1) < H;
  if (xin && !(W & 1)) {                                               // even widths: 8-byte aligned pairs
    *reinterpret_cast<float2*>(dst) = make_float2(p00 * sc, p01 * sc);
    if (yin) *reinterpret_cast<float2*>(dst + W) = make_float2(p10 * sc, p11 * sc);
  } else {
    dst[0] = p00 * sc;
    if (xin) dst[1] = p01 * sc;
    if (yin) {
      dst[W] = p10 * sc;
      if (xin) dst[W + 1] = p11 * sc;
    }
  }
}

// x[:, :, ::2, ::2] as a contiguous tensor, and its backward (zeros with gy at the even positions): the stride-2 entry of a
// ResNeXt stage -- a strided 3x3 'same' convolution is evaluated as the stride-1 kernel's output sub-sampled, a strided 1x1
// shortcut as the 1x1 kernel on the sub-sampled input (dvd_hip/conv.py XConv2d; third_party/midas_blocks.py:35-50 via
// torchvision's Bottleneck).  Until round 6 ATen did both: a strided copy forward, a fill + a strided copy backward.
template <class T>
__global__ __launch_bounds__(256) void subsample2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int Ho, int Wo,
                                                             long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    y[i] = x[(pl * H + 2 * oy) * W + 2 * ox];
  }
}
// one thread = two horizontally adjacent input pixels (x even: the pair holds at most one gradient)
template <class T>
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int H, int W, int Ho, int Wo,
                                                             long long total_pairs) {
  const int Wp = (W + 1) >> 1;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total_pairs; i += (long long)gridDim.x * 256) {
    const int px = (int)(i % Wp);
    const long long r = i / Wp;
    const int y = (int)(r % H);
    const long long pl = r / H;
    T v = (T)0.0f;
    if (!(y & 1)) v = gy[(pl * Ho + (y >> 1)) * Wo + px];
    T* dst = gx + (pl * H + y) * W + 2 * px;
    dst[0] = v;
    if (2 * px + 1 < W) dst[1] = (T)0.0f;
  }
}

// nn.AvgPool2d(k, stride, padding) with PyTorch's defaults (count_include_pad: every window is divided by k * k, floor mode):
// AvgPool2d(2) = the hourglass's down-sampling (third_party/hourglass.py:60-158 `nn.AvgPool2d(2)` in front of every lower level)
// and AvgPool2d(3, 2, 1) = FCNUnet's (networks/FCNUnet.py:64, --use_cnn).  Until round 6 both ran on ATen
// (avg_pool2d_out_cuda_frame / avg_pool2d_backward_out_cuda_frame: 2.2 + 1.2 % of the hourglass step,
// profiles/r04_bench_kernel_trace_hourglass.txt).  Arithmetic as ATen's: the window is summed row by row in fp32 and divided
// by k * k (a division, not a reciprocal: 1 / 9 is not exact); backward adds gy / (k * k) of the windows that contain the
// pixel in (row, column) order of the windows.  One thread per output (forward) / input (backward) pixel.
template <class T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int Ho, int Wo,
                                                          int k, int st, int pad, long long total) {
  const float div = (float)(k * k);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const T* xp = x + (r / Ho) * H * W;
    const int y0 = oy * st - pad, x0 = ox * st - pad;
    float sum = 0.0f;
    for (int dy = 0; dy < k; ++dy) {
      const int yy = y0 + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int xx = x0 + dx;
        if (xx >= 0 && xx < W) sum += (float)xp[yy * W + xx];
      }
    }
    y[i] = (T)(sum / div);
  }
}
template <class T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int H, int W, int Ho, int Wo,
                                                          int k, int st, int pad, long long total) {
  const float div = (float)(k * k);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ix = (int)(i % W);
    const long long r = i / W;
    const int iy = (int)(r % H);
    const T* gp = gy + (r / H) * Ho * Wo;
    // windows oy with oy * st - pad <= iy < oy * st - pad + k
    int oy0 = (iy + pad - k + st) / st, ox0 = (ix + pad - k + st) / st;        // ceil((iy + pad - k + 1) / st) for a numerator >= -st + 1
    if (iy + pad - k + 1 <= 0) oy0 = 0;
    if (ix + pad - k + 1 <= 0) ox0 = 0;
    int oy1 = (iy + pad) / st, ox1 = (ix + pad) / st;
    if (oy1 > Ho - 1) oy1 = Ho - 1;
    if (ox1 > Wo - 1) ox1 = Wo - 1;
    float sum = 0.0f;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) sum += (float)gp[oy * Wo + ox] / div;
    gx[i] = (T)sum;
  }
}

}  // namespace dvd

extern "C" {

int dvd_subsample2_fwd(const void* x, void* y, int f16, long long planes, int H, int W, dvd_stream_t stream) {
  DVD_REQUIRE(x && y && planes > 0 && H > 0 && W > 0, "subsample2 fwd: bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = planes * Ho * Wo;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * (0.5 * H * W + (double)Ho * Wo) * (f16 ? 2 : 4));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::subsample2_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<T*>(y), H, W, Ho,
                                         Wo, total));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_subsample2_bwd(const void* gy, void* gx, int f16, long long planes, int H, int W, dvd_stream_t stream) {
  DVD_REQUIRE(gy && gx && planes > 0 && H > 0 && W > 0, "subsample2 bwd: bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = planes * H * ((W + 1) / 2);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * ((double)H * W + (double)Ho * Wo) * (f16 ? 2 : 4));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::subsample2_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(gy), static_cast<T*>(gx), H, W, Ho,
                                         Wo, total));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_avgpool_fwd(const void* x, void* y, int f16, long long planes, int H, int W, int k, int stride, int pad,
                    dvd_stream_t stream) {
  DVD_REQUIRE(x && y && planes > 0 && H > 0 && W > 0, "avgpool fwd: bad arguments");
  DVD_REQUIRE(k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && 2 * pad <= k, "avgpool fwd: kernel %d stride %d padding %d", k, stride, pad);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  DVD_REQUIRE(Ho > 0 && Wo > 0, "avgpool fwd: a %dx%d image is smaller than the window", H, W);
  const long long total = planes * Ho * Wo;
  long long blocks = (total + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * ((double)H * W + (double)Ho * Wo) * (f16 ? 2 : 4));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::avgpool_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<T*>(y), H, W, Ho,
                                         Wo, k, stride, pad, total));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_avgpool_bwd(const void* gy, void* gx, int f16, long long planes, int H, int W, int k, int stride, int pad,
                    dvd_stream_t stream) {
  DVD_REQUIRE(gy && gx && planes > 0 && H > 0 && W > 0, "avgpool bwd: bad arguments");
  DVD_REQUIRE(k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && 2 * pad <= k, "avgpool bwd: kernel %d stride %d padding %d", k, stride, pad);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  DVD_REQUIRE(Ho > 0 && Wo > 0, "avgpool bwd: a %dx%d image is smaller than the window", H, W);
  const long long total = planes * H * W;
  long long blocks = (total + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * ((double)H * W + (double)Ho * Wo) * (f16 ? 2 : 4));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::avgpool_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(gy), static_cast<T*>(gx), H, W, Ho,
                                         Wo, k, stride, pad, total));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_maxpool3s2_fwd(const float* x, void* y, int y_f16, unsigned char* index, long long planes, int H, int W,
                       dvd_stream_t stream) {
  DVD_REQUIRE(x && y && index && planes > 0 && H > 0 && W > 0, "maxpool fwd: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = planes * Ho * Wo;
  DVD_REQUIRE((total + 255) / 256 < (1LL << 31), "maxpool fwd: too large");
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * (4.0 * H * W + (double)Ho * Wo * ((y_f16 ? 2 : 4) + 1)));
  DVD_DISPATCH_T(y_f16, hipLaunchKernelGGL(dvd::maxpool3s2_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                                           static_cast<hipStream_t>(stream), x, static_cast<T*>(y), index, H, W, Ho, Wo, total));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_maxpool3s2_bwd(const void* gy, int gy_f16, const unsigned char* index, float* gx, const float* out_scale,
                       long long planes, int H, int W, dvd_stream_t stream) {
  DVD_REQUIRE(gy && index && gx && planes > 0 && H > 0 && W > 0, "maxpool bwd: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = planes * H * W;
  DVD_REQUIRE((total + 255) / 256 < (1LL << 31), "maxpool bwd: too large");
  dvd::bytes_add(DVD_BYTES_POOL, (double)planes * (4.0 * H * W + (double)Ho * Wo * ((gy_f16 ? 2 : 4) + 1)));
  const long long blocks2 = planes * Ho * Wo;            // 2 x 2 input blocks = outputs ((H + 1) / 2 == Ho for this geometry)
  if (blocks2 < (1LL << 31) && planes * (long long)Ho < (1LL << 31)) {
    DVD_DISPATCH_T(gy_f16, hipLaunchKernelGGL(dvd::maxpool3s2_bwd2_kernel<T>, dim3((unsigned)((blocks2 + 255) / 256)), dim3(256), 0,
                                              static_cast<hipStream_t>(stream), static_cast<const T*>(gy), index, gx, H, W, Ho, Wo,
                                              (unsigned)blocks2, out_scale));
  } else {
    DVD_DISPATCH_T(gy_f16, hipLaunchKernelGGL(dvd::maxpool3s2_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                                              static_cast<hipStream_t>(stream), static_cast<const T*>(gy), index, gx, H, W, Ho, Wo,
                                              total, out_scale));
  }
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
