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
constexpr int kWT_H = 8;                 // wgrad tile rows (its width is 56 or 64: wgrad_tile_w)

// y[n, g*8+co, :, :] = sum_{ci,ky,kx} x[n, g*8+ci, y+ky-1, x+kx-1] * w[g*8+co, ci, ky, kx]      (TRANSPOSED = false)
// gx[n, g*8+ci, :, :] = sum_{co,ky,kx} gy[n, g*8+co, y+1-ky, x+1-kx] * w[g*8+co, ci, ky, kx]    (TRANSPOSED = true)
//
// Round 6.  The 72 weights of one source channel are uniform over the block: they are read straight from the weight tensor into
// SGPRs (s_load) instead of through LDS -- rounds 1-5 spent 48 LDS clocks on weight broadcasts per 12 on pixels, and were
// LDS- and staging-bound at 24 % of the packed-FMA rate.  The 32 accumulators are 16 pixel PAIRS (v_pk_fma_f32: pixel pair x
// broadcast SGPR weight); every accumulator sees the products of rounds 1-5 in the same order, so results are bit-identical.
// The tile is staged with 16-byte loads when rows are 4-element aligned.  (TW, TH) = (84, 12) tiles a 168-wide plane exactly.
template <bool TRANSPOSED, class T, int TW, int TH>
__global__ __launch_bounds__(256) void gconv3x3_c8_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                          T* __restrict__ out, int C, int H, int W, int tiles_x) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int IW = TW + 2 + 2;      // +2 halo, +2 pad: the row stride is a 16-byte multiple
  constexpr int IH = TH + 2;
  constexpr int NSX = TW / 4;         // 4-pixel strips per row
  static_assert(TW % 4 == 0 && NSX * TH <= 256, "one thread per strip");
  __shared__ __attribute__((aligned(16))) float s_in[kCPG][IH][IW];
  const int tile = blockIdx.x, g = blockIdx.y, n = blockIdx.z;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  const size_t plane = (size_t)H * W;
  const T* inb = in + ((size_t)n * C + (size_t)g * kCPG) * plane;
  T* outb = out + ((size_t)n * C + (size_t)g * kCPG) * plane;
  // input tile with halo (zero outside the image)
  if ((W & 3) == 0) {
    // body: 8 * IH rows of NSX aligned quads; all loads of a thread are requested before the first one is stored
    constexpr int kQ = kCPG * IH * NSX, kU = (kQ + 255) / 256;
    float4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = threadIdx.x + u * 256;
      const int row = i / NSX, q = i - row * NSX, c = row / IH, yy = row - c * IH;
      const int gy = y0 + yy - 1, gx = x0 + q * 4;
      v[u] = (i < kQ && gy >= 0 && gy < H && gx < W) ? ld4(inb + (size_t)c * plane + (size_t)gy * W + gx)
                                                      : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // the two halo columns of every row
    constexpr int kE = kCPG * IH * 2, kUE = (kE + 255) / 256;
    float e[kUE];
#pragma unroll
    for (int u = 0; u < kUE; ++u) {
      const int i = threadIdx.x + u * 256;
      const int row = i >> 1, c = row / IH, yy = row - c * IH;
      const int gy = y0 + yy - 1, gx = (i & 1) ? x0 + TW : x0 - 1;
      e[u] = (i < kE && gy >= 0 && gy < H && gx >= 0 && gx < W) ? ldf(inb + (size_t)c * plane + (size_t)gy * W + gx) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < kQ) {
        const int row = i / NSX, q = i - row * NSX;
        float* d = &s_in[0][0][0] + row * IW + 1 + q * 4;
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < kUE; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < kE) (&s_in[0][0][0])[(i >> 1) * IW + ((i & 1) ? TW + 1 : 0)] = e[u];
    }
  } else {
    // kU loads are requested before the first one is stored (a thread has kU loads in flight, not one)
    constexpr int kU = 8, kRow = TW + 2, kTot = kCPG * IH * kRow;
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
  if (threadIdx.x >= NSX * TH) return;
  const int sy = threadIdx.x / NSX, sx = threadIdx.x - sy * NSX;   // row, strip of 4 pixels
  v2f acc[kCPG][2];
#pragma unroll
  for (int d = 0; d < kCPG; ++d) acc[d][0] = acc[d][1] = (v2f){0.0f, 0.0f};
  const float* wg = w + (size_t)g * (kCPG * kCPG * 9);
#pragma unroll 1   // one source channel at a time: its 72 weights fill the SGPRs
  for (int s = 0; s < kCPG; ++s) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = &s_in[s][sy + ky][sx * 4];
      const v4f a = *reinterpret_cast<const v4f*>(row);
      const v2f b = *reinterpret_cast<const v2f*>(row + 4);
      const v2f xp[3][2] = {{{a.x, a.y}, {a.z, a.w}}, {{a.y, a.z}, {a.w, b.x}}, {{a.z, a.w}, {b.x, b.y}}};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int d = 0; d < kCPG; ++d) {
          // src = ci, dst = co  |  src = co, dst = ci with the taps flipped
          const float wv = !TRANSPOSED ? wg[(d * kCPG + s) * 9 + ky * 3 + kx] : wg[(s * kCPG + d) * 9 + (2 - ky) * 3 + (2 - kx)];
          acc[d][0] = __builtin_elementwise_fma(xp[kx][0], (v2f){wv, wv}, acc[d][0]);
          acc[d][1] = __builtin_elementwise_fma(xp[kx][1], (v2f){wv, wv}, acc[d][1]);
        }
      }
    }
  }
  const int oy = y0 + sy, ox = x0 + sx * 4;
  if (oy < H && ox < W) {
    const bool vec = ((W & 3) == 0) && (ox + 3 < W);
#pragma unroll
    for (int d = 0; d < kCPG; ++d) {
      T* dst = outb + (size_t)d * plane + (size_t)oy * W + ox;
      const float o[4] = {acc[d][0].x, acc[d][0].y, acc[d][1].x, acc[d][1].y};
      if (vec) {
        st4(dst, make_float4(o[0], o[1], o[2], o[3]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ox + j < W) stf(dst + j, o[j]);
      }
    }
  }
}

// launch the instantiation whose tile wastes least of a W-wide row
template <bool TRANSPOSED>
static void launch_c8(const void* in, const float* w, void* out, int f16, int N, int C, int H, int W, hipStream_t stream) {
  if (W % 84 == 0) {
    const int tx = W / 84, ty = (H + 11) / 12;
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((gconv3x3_c8_kernel<TRANSPOSED, T, 84, 12>), dim3(tx * ty, C / kCPG, N), dim3(256), 0,
                                           stream, static_cast<const T*>(in), w, static_cast<T*>(out), C, H, W, tx));
  } else {
    const int tx = (W + 63) / 64, ty = (H + 15) / 16;
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((gconv3x3_c8_kernel<TRANSPOSED, T, 64, 16>), dim3(tx * ty, C / kCPG, N), dim3(256), 0,
                                           stream, static_cast<const T*>(in), w, static_cast<T*>(out), C, H, W, tx));
  }
}

// partial[band][g][co][ci][9] = sum over the pixels of a band of 8 rows of gy[co, p] * x[ci, p + tap]
//
// Round 6.  A lane owns one PAIR of destination channels and one source channel (4 pairs x 8 = 32 lanes), so a wave works on two
// rows at once (one per half) and the four waves cover the 8 rows.  gy lies in LDS interleaved by channel pair
// ([pair][row][x][2]): one ds_read_b128 hands a lane two pixels of both its channels, and the 9 taps accumulate with packed
// FMAs (v_pk_fma_f32: the gy pair times a broadcast x).  The round-1..5 kernel (one (co, ci) per lane, scalar FMAs, one 64x8
// tile per block) spent 44 LDS clocks per 36 FMAs per lane and most of its time staging 4-byte loads; this one spends 52 per 72,
// walks the TW-wide tiles of its band with the NEXT tile's 16-byte loads in flight under the current tile's FMAs, and writes one
// record per band (a third of the records the reduction reads).  TW = 56 tiles a 168-wide plane with no waste.
template <class T, int TW>
__global__ __launch_bounds__(256) void gconv3x3_c8_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                float* __restrict__ partial, int C, int H, int W,
                                                                int tiles_x, int G) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int IW = 68;                   // row stride of an x plane (TW + 2 halo, + pad): 68 % 32 == 4
  constexpr int IH = kWT_H + 2;
  // strides padded so that the distinct ds_read_b128 addresses of a wave fall on disjoint groups of 4 banks
  constexpr int XP = IH * IW + 4;          // 684 floats: 684 % 32 == 12
  constexpr int GW2 = 144;                 // floats per row of a gy channel PAIR (2 * TW, + pad): 144 % 32 == 16
  constexpr int GP2 = kWT_H * GW2 + 4;     // 1156 floats: 1156 % 32 == 4
  constexpr int kRedBytes = kWT_H * 32 * 18 * 4, kXBytes = kCPG * XP * 4;
  __shared__ __attribute__((aligned(16))) float s_x[(kRedBytes > kXBytes ? kRedBytes : kXBytes) / 4];
  __shared__ __attribute__((aligned(16))) float s_g[(kCPG / 2) * GP2];
  const int band = blockIdx.x, g = blockIdx.y, n = blockIdx.z;
  const int y0 = band * kWT_H;
  const size_t plane = (size_t)H * W;
  const T* xb = x + ((size_t)n * C + (size_t)g * kCPG) * plane;
  const T* gb = gy + ((size_t)n * C + (size_t)g * kCPG) * plane;
  const bool vec = (W & 3) == 0;

  // vector staging: aligned quads of x (8 channels x 10 rows), the two halo columns, and quads of BOTH channels of a gy pair
  constexpr int NQ = TW / 4;
  constexpr int kXQ = kCPG * IH * NQ, kUX = (kXQ + 255) / 256;
  constexpr int kXE = kCPG * IH * 2;                              // <= 256
  constexpr int kGQ = (kCPG / 2) * kWT_H * NQ, kUG = (kGQ + 255) / 256;
  float4 xv[kUX], g0[kUG], g1[kUG];
  float xe;
  auto fetch = [&](int x0) {
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < kUX; ++u) {
      const int i = threadIdx.x + u * 256;
      const int row = i / NQ, q = i - row * NQ, c = row / IH, yy = row - c * IH;
      const int py = y0 + yy - 1, px = x0 + q * 4;
      xv[u] = (i < kXQ && py >= 0 && py < H && px < W) ? ld4(xb + (size_t)c * plane + (size_t)py * W + px) : z;
    }
    {
      const int i = threadIdx.x, row = i >> 1, c = row / IH, yy = row - c * IH;
      const int py = y0 + yy - 1, px = (i & 1) ? x0 + TW : x0 - 1;
      xe = (i < kXE && py >= 0 && py < H && px >= 0 && px < W) ? ldf(xb + (size_t)c * plane + (size_t)py * W + px) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kUG; ++u) {
      const int i = threadIdx.x + u * 256;
      const int row = i / NQ, q = i - row * NQ, cp = row / kWT_H, yy = row - cp * kWT_H;
      const int py = y0 + yy, px = x0 + q * 4;
      const bool in = i < kGQ && py < H && px < W;
      const T* src = gb + (size_t)(2 * cp) * plane + (size_t)py * W + px;
      g0[u] = in ? ld4(src) : z;
      g1[u] = in ? ld4(src + plane) : z;
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int u = 0; u < kUX; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < kXQ) {
        const int row = i / NQ, q = i - row * NQ, c = row / IH, yy = row - c * IH;
        float* d = &s_x[c * XP + yy * IW + 1 + q * 4];
        d[0] = xv[u].x; d[1] = xv[u].y; d[2] = xv[u].z; d[3] = xv[u].w;
      }
    }
    if (threadIdx.x < kXE) {
      const int row = threadIdx.x >> 1, c = row / IH, yy = row - c * IH;
      s_x[c * XP + yy * IW + ((threadIdx.x & 1) ? TW + 1 : 0)] = xe;
    }
#pragma unroll
    for (int u = 0; u < kUG; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < kGQ) {
        const int row = i / NQ, q = i - row * NQ, cp = row / kWT_H, yy = row - cp * kWT_H;
        v4f* d = reinterpret_cast<v4f*>(&s_g[cp * GP2 + yy * GW2 + q * 8]);
        d[0] = (v4f){g0[u].x, g1[u].x, g0[u].y, g1[u].y};
        d[1] = (v4f){g0[u].z, g1[u].z, g0[u].w, g1[u].w};
      }
    }
  };
  // rows that are not 4-element aligned: 4-byte loads, no prefetch
  auto stage_scalar = [&](int x0) {
    constexpr int kU = 8, kRow = TW + 2, kTot = kCPG * IH * kRow;
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
    constexpr int kTotG = kCPG * kWT_H * TW;
    for (int i0 = threadIdx.x; i0 < kTotG; i0 += 256 * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        const int c = i / (kWT_H * TW), r = i - c * (kWT_H * TW), yy = r / TW, xx = r - yy * TW;
        const int py = y0 + yy, px = x0 + xx;
        v[u] = (i < kTotG && py < H && px < W) ? ldf(gb + (size_t)c * plane + (size_t)py * W + px) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * 256;
        if (i < kTotG) {
          const int c = i / (kWT_H * TW), r = i - c * (kWT_H * TW), yy = r / TW, xx = r - yy * TW;
          s_g[(c >> 1) * GP2 + yy * GW2 + xx * 2 + (c & 1)] = v[u];
        }
      }
    }
  };

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave * 2 + (lane >> 5), cp = (lane >> 3) & 3, ci = lane & 7;
  v2f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (v2f){0.0f, 0.0f};
  const float* grow = &s_g[cp * GP2 + r * GW2];
  const float* xrow = &s_x[ci * XP + r * IW];
  if (vec) fetch(0);
  for (int tx = 0; tx < tiles_x; ++tx) {
    if (vec) put(); else stage_scalar(tx * TW);
    __syncthreads();
    if (vec && tx + 1 < tiles_x) fetch((tx + 1) * TW);
#pragma unroll 2
    for (int sx = 0; sx < TW / 4; ++sx) {
      const v4f ga = *reinterpret_cast<const v4f*>(grow + sx * 8);
      const v4f gb4 = *reinterpret_cast<const v4f*>(grow + sx * 8 + 4);
      const v2f gv[4] = {{ga.x, ga.y}, {ga.z, ga.w}, {gb4.x, gb4.y}, {gb4.z, gb4.w}};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* row = xrow + ky * IW + sx * 4;
        const float4 a = *reinterpret_cast<const float4*>(row);
        const float2 b = *reinterpret_cast<const float2*>(row + 4);
        const float seg[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[ky * 3 + kx] = __builtin_elementwise_fma(gv[j], (v2f){seg[j + kx], seg[j + kx]}, acc[ky * 3 + kx]);
      }
    }
    __syncthreads();                       // the tile is consumed: the next one (or the row sums) may take its place
  }
  // the 8 row sums of every weight, added in row order, then one record per (image, band, group)
  float* s_red = s_x;                      // [row][cp][ci][tap][co & 1]
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<v2f*>(&s_red[((r * 32 + cp * 8 + ci) * 9 + t) * 2]) = acc[t];
  __syncthreads();
  float* dst = partial + ((size_t)(n * gridDim.x + band) * G + g) * 576;
  for (int i = threadIdx.x; i < 576; i += 256) {
    const int co = i / 72, rem = i - co * 72, ci2 = rem / 9, t = rem - ci2 * 9;
    const int src = (((co >> 1) * 8 + ci2) * 9 + t) * 2 + (co & 1);
    float sum = s_red[src];
#pragma unroll
    for (int rr = 1; rr < kWT_H; ++rr) sum += s_red[rr * 576 + src];
    dst[i] = sum;
  }
}

// gw[i] (+)= sum over records r of partial[r][i]; i over G*64*9 weights.  A block owns 64 weights; its four waves sum the
// records of one residue class r mod 4 each, in ascending order, and the classes are added as ((0 + 1) + (2 + 3)) -- the
// order of the one-thread-per-weight loop of rounds 1-5, on four times the threads.
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                 int n_records, int n_weights, int accumulate,
                                                                 const float* __restrict__ out_scale) {
  __shared__ float s_part[3][64];
  const int lane = threadIdx.x & 63, cls = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < n_weights;
  float s = 0.0f;
  if (live) {
    const int n4 = n_records & ~3;
    const float* p = partial + i;
#pragma unroll 8
    for (int r = cls; r < n4; r += 4) s += p[(size_t)r * n_weights];
    if (cls == 0)
      for (int r = n4; r < n_records; ++r) s += p[(size_t)r * n_weights];
  }
  if (cls > 0) s_part[cls - 1][lane] = s;
  __syncthreads();
  if (cls == 0 && live) {
    const float t = ((s + s_part[0][lane]) + (s_part[1][lane] + s_part[2][lane])) * (out_scale ? out_scale[0] : 1.0f);
    gw[i] = accumulate ? gw[i] + t : t;    // fp16 gradients carry the loss scale
  }
}

static inline int wgrad_tile_w(int W) { return W % 56 == 0 ? 56 : 64; }

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
  dvd::launch_c8<false>(x, w, y, f16, N, C, H, W, static_cast<hipStream_t>(stream));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_gconv3x3_c8_bwd_data_t(const void* gy, const float* w, void* gx, int f16, int N, int C, int H, int W,
                               dvd_stream_t stream) {
  if (int e = dvd::check_shape(N, C, H, W)) return e;
  DVD_REQUIRE(gy && w && gx, "gconv bwd_data: null pointer");
  dvd::bytes_add(DVD_BYTES_GCONV, 2.0 * N * C * (double)H * W * (f16 ? 2 : 4));
  dvd::launch_c8<true>(gy, w, gx, f16, N, C, H, W, static_cast<hipStream_t>(stream));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

size_t dvd_gconv3x3_c8_wgrad_workspace_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || C % dvd::kCPG) return 0;
  const size_t tiles = (size_t)((H + dvd::kWT_H - 1) / dvd::kWT_H);   // one record per band of 8 rows
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
  const int tw = dvd::wgrad_tile_w(W);
  const int tx = (W + tw - 1) / tw, bands = (H + dvd::kWT_H - 1) / dvd::kWT_H;
  const int G = C / dvd::kCPG;
  if (tw == 56) {
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::gconv3x3_c8_wgrad_kernel<T, 56>), dim3(bands, G, N), dim3(256), 0,
                                           static_cast<hipStream_t>(stream), static_cast<const T*>(x),
                                           static_cast<const T*>(gy), static_cast<float*>(workspace), C, H, W, tx, G));
  } else {
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::gconv3x3_c8_wgrad_kernel<T, 64>), dim3(bands, G, N), dim3(256), 0,
                                           static_cast<hipStream_t>(stream), static_cast<const T*>(x),
                                           static_cast<const T*>(gy), static_cast<float*>(workspace), C, H, W, tx, G));
  }
  DVD_LAUNCH_OK();
  const int n_weights = C * dvd::kCPG * 9;
  hipLaunchKernelGGL(dvd::gconv_wgrad_reduce_kernel, dim3((n_weights + 63) / 64), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(workspace), gw, bands * N, n_weights,
                     accumulate, out_scale);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
