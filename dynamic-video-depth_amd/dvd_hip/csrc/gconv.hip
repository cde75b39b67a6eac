// Grouped 3x3 convolution with 8 channels per group (stride 1, pad 1, NCHW fp32) for gfx950:
// forward, backward-data and backward-weight.
//
// What it replaces: the `conv2` of the three ResNeXt-101 32x8d stage-1 bottlenecks of the MiDaS
// encoder (reference: third_party/midas_blocks.py:35-50 builds torchvision's
// ResNet(Bottleneck,[3,4,23,3],groups=32,width_per_group=8); stage 1 has width 256 = 32 groups x 8).
// On MI355X MIOpen's immediate mode serves the BACKWARD of this shape with a per-image
// GEMM fallback: profiles/r01_depthnet_profile.txt shows 30 ms per convolution_backward call at
// 8 x 256 x 96 x 168 (9.5 GFLOP: 0.3 TFLOP/s), 38 % of the depth net's forward+backward time.
//
// Roofline: HBM.  A group is a tiny 8x8x3x3 convolution: 1152 FLOP per pixel-group for 64 B of
// activations moved, far below the MFMA ridge and awkward for MFMA tiles (M = 8), so this is a
// direct convolution on the packed-fp32 VALU with everything staged through LDS:
//   * forward / backward-data (one kernel; backward-data = forward with the taps flipped and the
//     channel roles swapped): a 256-thread block owns a 64x16 pixel tile of one (image, group);
//     the 8 input planes (+1 px halo) and the 576 weights sit in LDS; a thread produces
//     8 output channels x 4 adjacent pixels (32 accumulators), reading its 6-pixel input row
//     segment once per (channel, ky) and the weights as broadcast ds_read_b128.
//     Algorithmic bytes: 8 B per pixel-channel (read x, write y).
//   * backward-weight: gw[co,ci,ky,kx] = sum over pixels of gy[co] * x[ci] shifted.  A block owns a
//     64x8 tile of one (image, group); lane = (co, ci) pair, each of the 4 waves takes 2 rows;
//     per strip of 4 pixels a thread does 36 FMAs for 7 LDS vector reads.  Per-tile partial
//     sums go to a workspace and are reduced in a fixed order by a second kernel
//     (deterministic, no atomics).  Algorithmic bytes: 8 B per pixel-channel (read x, gy).

#include "dvd_io.h"

namespace dvd {

constexpr int kCPG = 8;          // channels per group
constexpr int kFT_W = 64, kFT_H = 16;   // forward / dgrad tile
constexpr int kWT_W = 64, kWT_H = 8;    // wgrad tile

// y[n, g*8+co, :, :] = sum_{ci,ky,kx} x[n, g*8+ci, y+ky-1, x+kx-1] * w[g*8+co, ci, ky, kx]      (TRANSPOSED = false)
// gx[n, g*8+ci, :, :] = sum_{co,ky,kx} gy[n, g*8+co, y+1-ky, x+1-kx] * w[g*8+co, ci, ky, kx]    (TRANSPOSED = true)
template <bool TRANSPOSED, class T>
__global__ __launch_bounds__(256) void gconv3x3_c8_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                          T* __restrict__ out, int C, int H, int W, int tiles_x) {
  constexpr int IW = kFT_W + 2 + 2;   // +2 halo, +2 pad: row stride 68 floats (16-byte multiple)
  constexpr int IH = kFT_H + 2;
  __shared__ __attribute__((aligned(16))) float s_in[kCPG][IH][IW];
  __shared__ __attribute__((aligned(16))) float s_w[kCPG][3][3][kCPG];   // [src channel][ky][kx][dst channel]
  const int tile = blockIdx.x, g = blockIdx.y, n = blockIdx.z;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * kFT_W, y0 = ty * kFT_H;
  const size_t plane = (size_t)H * W;
  const T* inb = in + ((size_t)n * C + (size_t)g * kCPG) * plane;
  T* outb = out + ((size_t)n * C + (size_t)g * kCPG) * plane;
  // weights of the group -> LDS in [src][ky][kx][dst] order
  for (int i = threadIdx.x; i < kCPG * kCPG * 9; i += 256) {
    const int co = i / (kCPG * 9), r = i - co * (kCPG * 9), ci = r / 9, t = r - ci * 9, ky = t / 3, kx = t - ky * 3;
    const float v = w[((size_t)(g * kCPG + co) * kCPG + ci) * 9 + t];
    if (!TRANSPOSED)
      s_w[ci][ky][kx][co] = v;              // src = ci, dst = co
    else
      s_w[co][2 - ky][2 - kx][ci] = v;      // src = co, dst = ci, taps flipped
  }
  // input tile with halo (zero outside the image)
  // kU loads are requested before the first one is stored (a thread has kU loads in flight, not one)
  {
    constexpr int kU = 8, kRow = kFT_W + 2, kTot = kCPG * IH * kRow;
    for (int i0 = threadIdx.x; i0 < kTot; i0 += 256 * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        const int c = i / (IH * kRow), r = i - c * (IH * kRow), yy = r / kRow, xx = r - yy * kRow;
        const int gy = y0 + yy - 1, gx = x0 + xx - 1;
        v[u] = (i < kTot && gy >= 0 && gy < H && gx >= 0 && gx < W) ? ldf(inb + (size_t)c * plane + (size_t)gy * W + gx) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        if (i < kTot) {
          const int c = i / (IH * kRow), r = i - c * (IH * kRow), yy = r / kRow, xx = r - yy * kRow;
          s_in[c][yy][xx] = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int sx = threadIdx.x & 15, sy = threadIdx.x >> 4;   // strip of 4 pixels, row
  float acc[kCPG][4];
#pragma unroll
  for (int d = 0; d < kCPG; ++d)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[d][j] = 0.0f;
#pragma unroll 1   // one source channel at a time: fully unrolled, the 72 row segments + 576 weights spill
  for (int s = 0; s < kCPG; ++s) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = &s_in[s][sy + ky][sx * 4];
      const float4 a = *reinterpret_cast<const float4*>(row);
      const float2 b = *reinterpret_cast<const float2*>(row + 4);
      const float seg[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w0 = *reinterpret_cast<const float4*>(&s_w[s][ky][kx][0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&s_w[s][ky][kx][4]);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int d = 0; d < kCPG; ++d)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[d][j] = __builtin_fmaf(seg[j + kx], wv[d], acc[d][j]);
      }
    }
  }
  const int oy = y0 + sy, ox = x0 + sx * 4;
  if (oy < H && ox < W) {
    const bool vec = ((W & 3) == 0) && (ox + 3 < W);
#pragma unroll
    for (int d = 0; d < kCPG; ++d) {
      T* dst = outb + (size_t)d * plane + (size_t)oy * W + ox;
      if (vec) {
        st4(dst, make_float4(acc[d][0], acc[d][1], acc[d][2], acc[d][3]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ox + j < W) stf(dst + j, acc[d][j]);
      }
    }
  }
}

// partial[tile][g][co][ci][9] = sum over the tile's pixels of gy[co, p] * x[ci, p + tap]
template <class T>
__global__ __launch_bounds__(256) void gconv3x3_c8_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                float* __restrict__ partial, int C, int H, int W,
                                                                int tiles_x, int tiles_per_img, int G) {
  constexpr int IW = kWT_W + 2 + 2;
  constexpr int IH = kWT_H + 2;
  // channel-plane strides padded so that the 8 distinct ds_read_b128 addresses of a wave (one per
  // co, or one per ci) fall on 8 disjoint groups of 4 banks
  constexpr int XP = IH * IW + 4;          // 684 floats: 684 % 32 == 12
  constexpr int GW = kWT_W + 4;
  constexpr int GP = kWT_H * GW + 4;       // 548 floats: 548 % 32 == 4
  __shared__ __attribute__((aligned(16))) float s_x[kCPG * XP];
  __shared__ __attribute__((aligned(16))) float s_g[kCPG * GP];
  __shared__ float s_red[3][64][9];
  const int tile = blockIdx.x, g = blockIdx.y, n = blockIdx.z;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * kWT_W, y0 = ty * kWT_H;
  const size_t plane = (size_t)H * W;
  const T* xb = x + ((size_t)n * C + (size_t)g * kCPG) * plane;
  const T* gb = gy + ((size_t)n * C + (size_t)g * kCPG) * plane;
  {
    constexpr int kU = 8, kRow = kWT_W + 2, kTot = kCPG * IH * kRow;
    for (int i0 = threadIdx.x; i0 < kTot; i0 += 256 * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        const int c = i / (IH * kRow), r = i - c * (IH * kRow), yy = r / kRow, xx = r - yy * kRow;
        const int py = y0 + yy - 1, px = x0 + xx - 1;
        v[u] = (i < kTot && py >= 0 && py < H && px >= 0 && px < W) ? ldf(xb + (size_t)c * plane + (size_t)py * W + px) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        if (i < kTot) {
          const int c = i / (IH * kRow), r = i - c * (IH * kRow), yy = r / kRow, xx = r - yy * kRow;
          s_x[c * XP + yy * IW + xx] = v[u];
        }
      }
    }
    constexpr int kTotG = kCPG * kWT_H * kWT_W;   // 4096: a multiple of 256 * kU
    for (int i0 = threadIdx.x; i0 < kTotG; i0 += 256 * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        const int c = i / (kWT_H * kWT_W), r = i - c * (kWT_H * kWT_W), yy = r / kWT_W, xx = r - yy * kWT_W;
        const int py = y0 + yy, px = x0 + xx;
        v[u] = (py < H && px < W) ? ldf(gb + (size_t)c * plane + (size_t)py * W + px) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        const int c = i / (kWT_H * kWT_W), r = i - c * (kWT_H * kWT_W), yy = r / kWT_W, xx = r - yy * kWT_W;
        s_g[c * GP + yy * GW + xx] = v[u];
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int co = lane >> 3, ci = lane & 7;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
  for (int r = wave; r < kWT_H; r += 4) {
#pragma unroll 4
    for (int sx = 0; sx < kWT_W / 4; ++sx) {
      const float4 g4 = *reinterpret_cast<const float4*>(&s_g[co * GP + r * GW + sx * 4]);
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* row = &s_x[ci * XP + (r + ky) * IW + sx * 4];
        const float4 a = *reinterpret_cast<const float4*>(row);
        const float2 b = *reinterpret_cast<const float2*>(row + 4);
        const float seg[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = __builtin_fmaf(gv[j], seg[j + kx], acc[ky * 3 + kx]);
      }
    }
  }
  // fixed-order sum over the 4 waves, then one record per (tile, image, group)
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) s_red[wave - 1][lane][t] = acc[t];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = partial + (((size_t)(n * tiles_per_img + tile) * G + g) * 64 + lane) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) dst[t] = ((acc[t] + s_red[0][lane][t]) + s_red[1][lane][t]) + s_red[2][lane][t];
  }
}

// gw[i] (+)= sum over records r of partial[r][i], r ascending; i over G*64*9 weights
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                 int n_records, int n_weights, int accumulate,
                                                                 const float* __restrict__ out_scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_weights) return;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int r = 0;
  for (; r + 3 < n_records; r += 4) {
    s0 += partial[(size_t)r * n_weights + i];
    s1 += partial[(size_t)(r + 1) * n_weights + i];
    s2 += partial[(size_t)(r + 2) * n_weights + i];
    s3 += partial[(size_t)(r + 3) * n_weights + i];
  }
  for (; r < n_records; ++r) s0 += partial[(size_t)r * n_weights + i];
  const float s = ((s0 + s1) + (s2 + s3)) * (out_scale ? out_scale[0] : 1.0f);     // fp16 gradients carry the loss scale
  gw[i] = accumulate ? gw[i] + s : s;
}

static int check_shape(int N, int C, int H, int W) {
  DVD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "gconv: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  DVD_REQUIRE(C % kCPG == 0, "gconv: C=%d is not a multiple of 8 (8 channels per group)", C);
  DVD_REQUIRE(C / kCPG <= 65535 && N <= 65535, "gconv: too many groups / images for the grid");
  return DVD_OK;
}

}  // namespace dvd

extern "C" {

int dvd_gconv3x3_c8_fwd(const float* x, const float* w, float* y, int N, int C, int H, int W, dvd_stream_t stream) {
  return dvd_gconv3x3_c8_fwd_t(x, w, y, 0, N, C, H, W, stream);
}
int dvd_gconv3x3_c8_bwd_data(const float* gy, const float* w, float* gx, int N, int C, int H, int W, dvd_stream_t stream) {
  return dvd_gconv3x3_c8_bwd_data_t(gy, w, gx, 0, N, C, H, W, stream);
}
int dvd_gconv3x3_c8_bwd_weight(const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                               size_t workspace_bytes, int N, int C, int H, int W, dvd_stream_t stream) {
  return dvd_gconv3x3_c8_bwd_weight_t(x, gy, gw, accumulate, workspace, workspace_bytes, 0, nullptr, N, C, H, W, stream);
}

int dvd_gconv3x3_c8_fwd_t(const void* x, const float* w, void* y, int f16, int N, int C, int H, int W, dvd_stream_t stream) {
  if (int e = dvd::check_shape(N, C, H, W)) return e;
  DVD_REQUIRE(x && w && y, "gconv fwd: null pointer");
  dvd::bytes_add(DVD_BYTES_GCONV, 2.0 * N * C * (double)H * W * (f16 ? 2 : 4));
  const int tx = (W + dvd::kFT_W - 1) / dvd::kFT_W, ty = (H + dvd::kFT_H - 1) / dvd::kFT_H;
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::gconv3x3_c8_kernel<false, T>), dim3(tx * ty, C / dvd::kCPG, N), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), w, static_cast<T*>(y), C, H, W,
                                         tx));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_gconv3x3_c8_bwd_data_t(const void* gy, const float* w, void* gx, int f16, int N, int C, int H, int W,
                               dvd_stream_t stream) {
  if (int e = dvd::check_shape(N, C, H, W)) return e;
  DVD_REQUIRE(gy && w && gx, "gconv bwd_data: null pointer");
  dvd::bytes_add(DVD_BYTES_GCONV, 2.0 * N * C * (double)H * W * (f16 ? 2 : 4));
  const int tx = (W + dvd::kFT_W - 1) / dvd::kFT_W, ty = (H + dvd::kFT_H - 1) / dvd::kFT_H;
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::gconv3x3_c8_kernel<true, T>), dim3(tx * ty, C / dvd::kCPG, N), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(gy), w, static_cast<T*>(gx), C, H,
                                         W, tx));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

size_t dvd_gconv3x3_c8_wgrad_workspace_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || C % dvd::kCPG) return 0;
  const size_t tiles = (size_t)((W + dvd::kWT_W - 1) / dvd::kWT_W) * ((H + dvd::kWT_H - 1) / dvd::kWT_H);
  return tiles * N * (size_t)C * dvd::kCPG * 9 * sizeof(float);
}

int dvd_gconv3x3_c8_bwd_weight_t(const void* x, const void* gy, float* gw, int accumulate, void* workspace,
                                 size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int H, int W,
                                 dvd_stream_t stream) {
  if (int e = dvd::check_shape(N, C, H, W)) return e;
  DVD_REQUIRE(x && gy && gw && workspace, "gconv bwd_weight: null pointer");
  dvd::bytes_add(DVD_BYTES_GCONV, 2.0 * N * C * (double)H * W * (f16 ? 2 : 4));
  const size_t need = dvd_gconv3x3_c8_wgrad_workspace_bytes(N, C, H, W);
  if (workspace_bytes < need) {
    dvd::set_error("gconv bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  const int tx = (W + dvd::kWT_W - 1) / dvd::kWT_W, ty = (H + dvd::kWT_H - 1) / dvd::kWT_H;
  const int G = C / dvd::kCPG;
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::gconv3x3_c8_wgrad_kernel<T>, dim3(tx * ty, G, N), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<const T*>(gy),
                                         static_cast<float*>(workspace), C, H, W, tx, tx * ty, G));
  DVD_LAUNCH_OK();
  const int n_weights = C * dvd::kCPG * 9;
  hipLaunchKernelGGL(dvd::gconv_wgrad_reduce_kernel, dim3((n_weights + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(workspace), gw, tx * ty * N, n_weights,
                     accumulate, out_scale);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
