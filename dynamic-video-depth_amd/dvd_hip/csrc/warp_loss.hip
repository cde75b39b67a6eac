// Fused unproject -> scene-flow advect -> reproject -> bilinear flow warp ->
// masked losses, forward + backward in ONE launch, for gfx950 (MI355X).
//
// What it replaces (reference, /root/reference):
//   losses/scene_flow_projection.py:114-153  flow_by_depth.forward
//   losses/scene_flow_projection.py:222-278  scene_flow_projection_slack.forward
//   losses/scene_flow_projection.py:103-112,212-220  backward_warp (F.grid_sample)
//   models/scene_flow_motion_field.py:140-150,285-324  disp_loss / _calc_loss
// plus the autograd backward of all of it (about 150 ATen launches and
// 2.5 GB of intermediates at B=48, 384x672 in the reference).
//
// Roofline: HBM.  Algorithmic bytes per pixel-pair: read depth_1 4, depth_2 4
// (gather), flow 8, mask 4, scene flow 12; write g_depth_1 4, g_depth_2 4
// (scatter-add), g_sf 12  => 52 B.  ~300 FLOP per pixel.
//
// Layout / mapping: one thread owns PX horizontally adjacent pixels so every
// streaming access is a 16-byte vector (PX=4); a 256-thread block owns 1024
// consecutive pixels of one pair, so the camera block of the pair is
// wave-uniform and lives in SGPRs.  That is the DIRECT reference variant (global gathers, depth_2 gradient scattered
// with global_atomic_add_f32; kept for A/B runs and as a second implementation); the production path is the TILED
// kernel further down: depth_2 window and a Q31.32 fixed-point gradient accumulator (ds_add_u64) in LDS per tile.
//
// Numerics: the forward follows the reference's fp32 operation order exactly
// (see dvd_common.h rowvec_mat3 and sample_coord/bilinear below), so the
// index masks [depth_1<100], [W2.z<100], [I.z<1e-3] and the tap indices are
// bit-identical to PyTorch's CPU path.  Build with -ffp-contract=off.

#include "warp_pixel.h"

namespace dvd {

struct TileArgs {
  float* slabs;
  Overflow ovf;
  int2* offs;         // per pair: window offset (multiple of 4 in x), written by the pair's first tile
  int ntx, nty;
  int direct;         // 1: window cells no neighbouring window covers go straight to g_depth_2 (needs W % 4 == 0)
  int tile0;          // global index of this launch's first tile (a launch covers a contiguous run of pairs)
};

// The windows of adjacent tiles overlap by the halo: cell (wx, wy) of a tile's window is covered by that tile ALONE when
// wx in [2R + 4, TW) and wy in [2R + 1, TH) (76 x 15 of the 116 x 49 cells of a 96 x 32 tile).  Those cells are final when
// the tile is done: the tile kernel converts and stores them into g_depth_2 itself, only the ring goes through the slab and
// the combine kernel (which skips the exclusive pixels).  Both kernels use this one predicate.
template <int TW, int TH, int R>
__device__ __forceinline__ bool tile_exclusive(int wx, int wy) {
  return wx >= 2 * R + 4 && wx < TW && wy >= 2 * R + 1 && wy < TH;
}

// g_depth_2[b,y,x .. x+3] = sum over the tiles whose window covers the quad, fixed order (dj outer, di inner): the ONE
// definition of the combine, used by combine_slabs_kernel and by the in-kernel combine of the tile kernel.
template <int TW, int TH, int R>
__device__ __forceinline__ void combine_quad(const float* __restrict__ slabs, int2 off, float* __restrict__ g_d2, int H, int W,
                                             int ntx, int nty, int b, int y, int x, int direct) {
  constexpr int WW = TW + 2 * R + 4;
  constexpr int WH = TH + 2 * R + 1;
  static_assert(2 * R + 4 <= TW && 2 * R + 1 <= TH, "only adjacent tiles may overlap a pixel");
  const int xs = x - off.x, ys = y - off.y;          // coordinates in the pair's shifted tile grid
  const int ti = xs >= 0 ? xs / TW : -((TW - 1 - xs) / TW), tj = ys >= 0 ? ys / TH : -((TH - 1 - ys) / TH);
  // written by the pixel's own tile (tile_exclusive; a quad is exclusive as a whole: every bound is a multiple of 4)
  if (direct && ti >= 0 && ti < ntx && tj >= 0 && tj < nty && tile_exclusive<TW, TH, R>(xs - ti * TW + R, ys - tj * TH + R)) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dj = -1; dj <= 1; ++dj) {
    const int j = tj + dj;
    const int wy = ys - (j * TH - R);
    if (j < 0 || j >= nty || wy < 0 || wy >= WH) continue;
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
      const int i = ti + di;
      const int wx = xs - (i * TW - R);
      if (i < 0 || i >= ntx || wx < 0 || wx + 3 >= WW) continue;
      const float4 v =
          *reinterpret_cast<const float4*>(slabs + ((size_t)(b * nty + j) * ntx + i) * (WW * WH) + wy * WW + wx);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
  float* dst = g_d2 + ((size_t)b * H + y) * W + x;
  if ((W & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = s;
  } else {
    dst[0] = s.x;
    if (x + 1 < W) dst[1] = s.y;
    if (x + 2 < W) dst[2] = s.z;
    if (x + 3 < W) dst[3] = s.w;
  }
}

constexpr int tile_lds_bytes(int tw, int th, int r) { return (tw + 2 * r + 4) * (th + 2 * r + 1) * 12 + 32 * 4 + 16; }
constexpr int tile_blocks_per_cu(int tw, int th, int r) { return 163840 / tile_lds_bytes(tw, th, r); }
// Measured on MI355X: this kernel is latency bound and its time falls steeply with resident
// waves (12 -> 16 waves/CU: 335 -> 233 us at 48x384x672), so take 4 waves/SIMD (128 VGPRs)
// whenever the LDS footprint admits it.
constexpr int tile_waves_per_simd(int tw, int th, int r, int nt) {
  return (tile_blocks_per_cu(tw, th, r) * nt + 255) / 256 > 4 ? 4 : (tile_blocks_per_cu(tw, th, r) * nt + 255) / 256;
}

// PX = pixels per thread-step (4: 16-byte vectors; 2: 8-byte vectors, which splits a 96x32 tile evenly
// over 512 threads -- 3 steps each instead of 1 or 2).
template <int TW, int TH, int R, int NT, bool GRADS, bool SHIPPED, int PX>
__global__ __launch_bounds__(NT, tile_waves_per_simd(TW, TH, R, NT)) void warp_loss_tiled_kernel(const WarpArgs a, const TileArgs ta) {
  constexpr int WW = TW + 2 * R + 4;  // multiple of 4: window rows are float4-aligned
  constexpr int WH = TH + 2 * R + 1;
  constexpr int QW = TW / PX;
  typedef float vecf __attribute__((ext_vector_type(PX)));
  static_assert(R % 4 == 0 && TW % 4 == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long* accw = reinterpret_cast<unsigned long long*>(smem);  // [WH][WW] u64 first (8-byte aligned)
  float* win = smem + 2 * WW * WH;
  float* camL = win + WW * WH;              // the pair's camera for the lockstep loop (kCamLdsFloats floats, 16-byte aligned)
  unsigned* lcount = reinterpret_cast<unsigned*>(camL + kCamLdsFloats);    // overflow records of this tile
  static_assert((3 * WW * WH) % 4 == 0 && kCamLdsFloats == 32, "camera quads must be 16-byte aligned");

  const int logical = ta.tile0 + xcd_contiguous_block(blockIdx.x, gridDim.x);      // global tile index
  const int tiles = ta.ntx * ta.nty;
  const int b = logical / tiles;
  const int t = logical - b * tiles;
  const int tj = t / ta.ntx, ti = t - tj * ta.ntx;
  const int tx0 = ti * TW, ty0 = tj * TH;
  // window offset of the pair (round 5: computed by every wave instead of a launch in front of this kernel -- 64 samples of
  // the flow on an 8 x 8 grid, the same shuffle tree in every wave of every tile, so all blocks of a pair agree bit for bit;
  // the requests fly next to the scalar loads of the camera below)
  const int2 off = pair_window_offset(a.flow, b, a.H, a.W);
  if (t == 0 && threadIdx.x == 0) ta.offs[b] = off;                 // for the combine and finish kernels
  const int wx0 = tx0 - R + off.x, wy0 = ty0 - R + off.y;
  Cam c0;
  load_cam(a, b, c0);
  Cam& c = c0;
  // pinhole intrinsics without skew (exact zeros / one in K^T and (K^-1)^T): block-uniform, selects pixel<.., PIN = true>
  const bool pinhole = c.Ki[1] == 0.0f && c.Ki[2] == 0.0f && c.Ki[3] == 0.0f && c.Ki[5] == 0.0f && c.Ki[8] == 1.0f &&
                       c.K[1] == 0.0f && c.K[2] == 0.0f && c.K[3] == 0.0f && c.K[5] == 0.0f && c.K[8] == 1.0f &&
                       DVD_WARP_PINHOLE;
  // The 51 camera scalars are wave-uniform; left alone they all land in SGPRs and push the
  // kernel past the 102-SGPR file (hundreds of v_readlane spill reloads).  Pin the three
  // matrices used mostly by the FMA-heavy parts into VGPRs instead.
  // (Round 5: the lockstep loop halves the uses per pixel and needs the VGPRs for two pixels' live values, so its
  //  instantiations keep the camera in SGPRs -- a packed instruction takes one SGPR pair as an operand; the pinhole
  //  form needs 41 of the 51 scalars.  DVD_WARP_CAM_VGPR selects matrices to pin for A/B builds.)
  constexpr int kCamV = (PX == 2 && SHIPPED && DVD_WARP_V5) ? DVD_WARP_CAM_VGPR : 15;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if (kCamV & 1) asm volatile("" : "+v"(c.R1[i]));
    if (kCamV & 2) asm volatile("" : "+v"(c.R2[i]));
    if (kCamV & 4) asm volatile("" : "+v"(c.K[i]));
    if (kCamV & 8) asm volatile("" : "+v"(c.R2T[i]));
  }
  const float* d2b = a.d2 + (size_t)b * a.HW;

  const bool wv = (a.W % PX) == 0;   // rows stay vector aligned
  auto locate = [&](int q, int& x, int& y) {
    const int ly = q / QW, lx = (q - ly * QW) * PX;
    y = ty0 + ly;
    x = tx0 + lx;
    return (q < QW * TH) && (y < a.H) && (x < a.W);
  };
  // ---- lockstep loop (round 5): per-pair input pointers.  DVD_WARP_V5_EARLY = 1 requests the FIRST thread-step's inputs
  //      before the window fill (their latency under the fill and the barrier): measured 231 us against 223 us for the
  //      launch sequence -- the requests queue in front of the window's and the registers they hold cost more; off.
  struct In2 {
    v2f d1, mk, s0, s1, s2;
    float4 fl;
  };
  // per-pair base pointers (wave-uniform: scalar registers) + one 32-bit byte offset per thread-step
  const char* d1b = reinterpret_cast<const char*>(a.d1 + (size_t)b * a.HW);
  const char* mkb = reinterpret_cast<const char*>(a.mask + (size_t)b * a.HW);
  const char* flb = reinterpret_cast<const char*>(a.flow + 2 * (size_t)b * a.HW);
  const char* sfb0 = reinterpret_cast<const char*>(a.sf + (size_t)b * 3 * a.HW);
  char* gd1b = reinterpret_cast<char*>(a.g_d1 + (size_t)b * a.HW);
  char* gsb0 = reinterpret_cast<char*>(a.g_sf + (size_t)b * 3 * a.HW);
  const unsigned plane = (unsigned)a.HW * 4u;
  // Branch free: a thread-step outside the tile / image reads the pair's first pixels instead (valid memory, never used),
  // so the requests are straight-line code that the scheduling barriers can hold in place -- behind a branch the block
  // was moved to the front of the step, and a wait for the rare global tap gathers (s_waitcnt vmcnt(0) at the merge) then
  // also waited for the prefetch.
  auto fetch2 = [&](int q) {
    int x, y;
    const bool ok = locate(q, x, y);
    const unsigned o = ok ? (unsigned)(y * a.W + x) * 4u : 0u;
    In2 r;
    r.d1 = *reinterpret_cast<const v2f*>(d1b + o);
    r.mk = *reinterpret_cast<const v2f*>(mkb + o);
    r.fl = *reinterpret_cast<const float4*>(flb + 2u * o);
    r.s0 = *reinterpret_cast<const v2f*>(sfb0 + o);
    r.s1 = *reinterpret_cast<const v2f*>(sfb0 + (o + plane));
    r.s2 = *reinterpret_cast<const v2f*>(sfb0 + (o + 2u * plane));
    return r;
  };
  In2 first = {};
  if constexpr (PX == 2 && SHIPPED && DVD_WARP_V5 && DVD_WARP_V5_EARLY) {
    if (wv) first = fetch2(threadIdx.x);
  }

  // ---- phase 0: fill the depth_2 window, clear the accumulator.  All of a thread's window loads are
  //      requested before the first one is consumed (the trip count is a compile-time constant).
  const bool w4 = (a.W & 3) == 0;
  {
    constexpr int kCells = (WW / 4) * WH;
    constexpr int kIt = (kCells + NT - 1) / NT;
    float4 v[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i = it * NT + threadIdx.x;
      const int wy = i / (WW / 4), wx = (i - wy * (WW / 4)) * 4;
      const int iy = wy0 + wy, ixx = wx0 + wx;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < kCells && iy >= 0 && iy < a.H && !DVD_WARP_KO_FILL) {
        if (w4) {
          if (ixx >= 0 && ixx < a.W) v[it] = *reinterpret_cast<const float4*>(d2b + (size_t)iy * a.W + ixx);
        } else {
          const float* row = d2b + (size_t)iy * a.W;
          if (ixx >= 0 && ixx < a.W) v[it].x = row[ixx];
          if (ixx + 1 >= 0 && ixx + 1 < a.W) v[it].y = row[ixx + 1];
          if (ixx + 2 >= 0 && ixx + 2 < a.W) v[it].z = row[ixx + 2];
          if (ixx + 3 >= 0 && ixx + 3 < a.W) v[it].w = row[ixx + 3];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i = it * NT + threadIdx.x;
      if (i < kCells) {
        const int wy = i / (WW / 4), wx = (i - wy * (WW / 4)) * 4;
        *reinterpret_cast<float4*>(win + wy * WW + wx) = v[it];
        if (GRADS) {
          uint4* z = reinterpret_cast<uint4*>(accw + wy * WW + wx);
          z[0] = make_uint4(0u, 0u, 0u, 0u);
          z[1] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
  }
  if (threadIdx.x == 0) *lcount = 0u;
  if constexpr (PX == 2 && SHIPPED && DVD_WARP_V5) {
    if (threadIdx.x < kCamLdsFloats) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < kCamLdsFloats; ++i) v = (int)threadIdx.x == i ? cam_lds_value(c, i) : v;
      camL[threadIdx.x] = v;
    }
  }
  __syncthreads();

  TileIO<WW, WH> io{d2b, win, accw, a.W, wx0, wy0, b * a.HW, logical, a.disp_mul, ta.ovf, lcount};
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  // ---- phase 1: the tile's pixels, PX per thread per step.  The inputs of step i+1 are requested before
  //      step i is evaluated: all waves of a block leave the barrier together, so without this every
  //      load latency of the block is exposed at the same time.  Build-time switch DVD_WARP_PREFETCH, off by
  //      default: measured 297 us with it (the 14 extra live registers spill) against 259 us without
  struct In {
    float d1[PX], mk[PX], fl[2 * PX], s0[PX], s1[PX], s2[PX];
  };
  auto fetch = [&](int q, In& r) {
    int x, y;
    if (!locate(q, x, y)) return;
    const int p0 = y * a.W + x;
    const size_t base = (size_t)b * a.HW + p0;
    const float* sfb = a.sf + (size_t)b * 3 * a.HW + p0;
    const int nvalid = (a.W - x) < PX ? (a.W - x) : PX;
    if (wv) {
      *reinterpret_cast<vecf*>(r.d1) = *reinterpret_cast<const vecf*>(a.d1 + base);
      *reinterpret_cast<vecf*>(r.mk) = *reinterpret_cast<const vecf*>(a.mask + base);
      *reinterpret_cast<vecf*>(r.fl) = *reinterpret_cast<const vecf*>(a.flow + 2 * base);
      *reinterpret_cast<vecf*>(r.fl + PX) = *reinterpret_cast<const vecf*>(a.flow + 2 * base + PX);
      *reinterpret_cast<vecf*>(r.s0) = *reinterpret_cast<const vecf*>(sfb);
      *reinterpret_cast<vecf*>(r.s1) = *reinterpret_cast<const vecf*>(sfb + a.HW);
      *reinterpret_cast<vecf*>(r.s2) = *reinterpret_cast<const vecf*>(sfb + 2 * a.HW);
    } else {
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        const bool ok = i < nvalid;
        r.d1[i] = ok ? a.d1[base + i] : 1.0f;
        r.mk[i] = ok ? a.mask[base + i] : 0.0f;
        r.fl[2 * i] = ok ? a.flow[2 * (base + i)] : 0.0f;
        r.fl[2 * i + 1] = ok ? a.flow[2 * (base + i) + 1] : 0.0f;
        r.s0[i] = ok ? sfb[i] : 0.0f;
        r.s1[i] = ok ? sfb[a.HW + i] : 0.0f;
        r.s2[i] = ok ? sfb[2 * a.HW + i] : 0.0f;
      }
    }
  };
  auto tile_pixels = [&](auto pin_tag) {
  constexpr bool PIN = decltype(pin_tag)::value;
  // In the instantiations that have the lockstep loop this one-pixel loop is the rare path (a camera with skew, odd-width
  // rows): it reads the camera AGAIN instead of keeping the 51 scalars of the prologue alive across the window fill -- they
  // pushed the whole kernel over its register budget (spills in the prologue and in the flush loop of every tile).
  Cam c;
  if constexpr (PX == 2 && SHIPPED && DVD_WARP_V5) {
    load_cam(a, b, c);
#pragma unroll
    for (int i = 0; i < 9; ++i) {                 // (as in the other instantiations: the FMA-heavy matrices in VGPRs)
      asm volatile("" : "+v"(c.R1[i]));
      asm volatile("" : "+v"(c.R2[i]));
      asm volatile("" : "+v"(c.K[i]));
      asm volatile("" : "+v"(c.R2T[i]));
    }
  } else {
    c = c0;
  }
  In cur, nxt;
  fetch(threadIdx.x, cur);
  for (int q = threadIdx.x; q < QW * TH; q += NT) {
    if (DVD_WARP_PREFETCH) fetch(q + NT, nxt);
    int x, y;
    if (locate(q, x, y)) {
      const int p0 = y * a.W + x;
      const size_t base = (size_t)b * a.HW + p0;
      const int nvalid = (a.W - x) < PX ? (a.W - x) : PX;
      float gd1[PX], g0[PX], g1[PX], g2[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        float gs[3] = {0.0f, 0.0f, 0.0f};
        gd1[i] = 0.0f;
        if (i < nvalid)
          pixel<GRADS, SHIPPED, PIN>(a, c, io, y, x + i, cur.d1[i], cur.fl[2 * i], cur.fl[2 * i + 1], cur.mk[i], cur.s0[i],
                                cur.s1[i], cur.s2[i], acc, gd1[i], gs);
        g0[i] = gs[0];
        g1[i] = gs[1];
        g2[i] = gs[2];
      }
      if (GRADS) {
        float* gsb = a.g_sf + (size_t)b * 3 * a.HW + p0;
        if (wv) {
          *reinterpret_cast<vecf*>(a.g_d1 + base) = *reinterpret_cast<const vecf*>(gd1);
          *reinterpret_cast<vecf*>(gsb) = *reinterpret_cast<const vecf*>(g0);
          *reinterpret_cast<vecf*>(gsb + a.HW) = *reinterpret_cast<const vecf*>(g1);
          *reinterpret_cast<vecf*>(gsb + 2 * a.HW) = *reinterpret_cast<const vecf*>(g2);
        } else {
          for (int i = 0; i < nvalid; ++i) {
            a.g_d1[base + i] = gd1[i];
            gsb[i] = g0[i];
            gsb[a.HW + i] = g1[i];
            gsb[2 * a.HW + i] = g2[i];
          }
        }
      }
    }
    if (DVD_WARP_PREFETCH)
      cur = nxt;
    else
      fetch(q + NT, cur);
  }
  };
  // ---- the lockstep loop (round 5): shipped flag set, pinhole pair, rows of whole pixel pairs
  auto tile_pixels2 = [&](auto crit_tag, const In2 first) {
    constexpr bool CRIT = decltype(crit_tag)::value;
    // (wave-uniform values computed on the vector unit: back into scalar registers, they are loop invariant)
    const float yhw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rcp_refined(a.half_w))));
    const float yhh = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rcp_refined(a.half_h))));
    In2 cur = DVD_WARP_V5_EARLY ? first : fetch2(threadIdx.x), nxt = cur;
    for (int q = threadIdx.x; q < QW * TH; q += NT) {
      int x, y;
      if (locate(q, x, y)) {
        const unsigned o = (unsigned)(y * a.W + x) * 4u;
        pixel2<GRADS, CRIT>(a, camL, io, y, x, cur.d1, (v2f){cur.fl.x, cur.fl.z}, (v2f){cur.fl.y, cur.fl.w}, cur.mk, cur.s0,
                            cur.s1, cur.s2, yhw, yhh, acc,
                            [&]() {
                              if (DVD_WARP_V5_PREFETCH) nxt = fetch2(q + NT);
                            },
                            [&](v2f gd1, v2f g0, v2f g1, v2f g2) {
                              *reinterpret_cast<v2f*>(gd1b + o) = gd1;
                              *reinterpret_cast<v2f*>(gsb0 + o) = g0;
                              *reinterpret_cast<v2f*>(gsb0 + (o + plane)) = g1;
                              *reinterpret_cast<v2f*>(gsb0 + (o + 2u * plane)) = g2;
                            });
      } else if (DVD_WARP_V5_PREFETCH) {
        nxt = fetch2(q + NT);
      }
      if (DVD_WARP_V5_PREFETCH)
        cur = nxt;
      else
        cur = fetch2(q + NT);
    }
  };
  if constexpr (PX == 2 && SHIPPED && DVD_WARP_V5) {
    // (a pinhole pair with odd-width rows takes the general instantiation: correct for every camera)
    bool r2t = true;            // R_2 and R_2_T are each other's transposes, bit for bit (wave-uniform)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) r2t = r2t && (c.R2[3 * i + j] == c.R2T[3 * j + i]);
    if (pinhole && wv && r2t) {
      if (a.crit_l2)
        tile_pixels2(std::true_type{}, first);
      else
        tile_pixels2(std::false_type{}, first);
    } else {
#ifndef DVD_WARP_NO_GENERAL
      tile_pixels(std::false_type{});
#endif
    }
  } else {
    if (pinhole)
      tile_pixels(std::true_type{});
    else
      tile_pixels(std::false_type{});
  }
  // ---- phase 2: accumulator window -> this tile's slab (coalesced), block sums
  __syncthreads();
  if (threadIdx.x == 0) ta.ovf.count[logical] = *lcount < ta.ovf.cap ? *lcount : ta.ovf.cap;
  if (GRADS) {
    float* slab = ta.slabs + (size_t)logical * (WW * WH);
    float* gb = a.g_d2 + (size_t)b * a.HW;
    const float back = a.disp_mul;
    for (int i = threadIdx.x; i < (WW * WH) / 4 && !DVD_WARP_KO_FLUSH; i += NT) {
      const longlong2 lo = reinterpret_cast<const longlong2*>(accw)[2 * i];
      const longlong2 hi = reinterpret_cast<const longlong2*>(accw)[2 * i + 1];
      const float4 v = make_float4(from_fixed(lo.x) * back, from_fixed(lo.y) * back, from_fixed(hi.x) * back, from_fixed(hi.y) * back);
      const int wy = i / (WW / 4), wx = (i - wy * (WW / 4)) * 4;
      if (ta.direct && tile_exclusive<TW, TH, R>(wx, wy)) {
        const int x = wx0 + wx, y = wy0 + wy;          // x is a multiple of 4 (R, the offset and the tile origin are)
        if (x >= 0 && x < a.W && y >= 0 && y < a.H) *reinterpret_cast<float4*>(gb + (size_t)y * a.W + x) = v;
      } else {
        reinterpret_cast<float4*>(slab)[i] = v;
      }
    }
  }
  float* red = win;  // window no longer needed
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) red[wave * 4 + k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.0f;
    for (int w = 0; w < NT / 64; ++w) v += red[w * 4 + threadIdx.x];
    a.partial[(size_t)logical * 4 + threadIdx.x] = v;
  }
}

template <int TW, int TH, int R>
__global__ __launch_bounds__(256) void combine_slabs_kernel(const float* __restrict__ slabs, const int2* __restrict__ offs,
                                                            float* __restrict__ g_d2, int H, int W, int ntx,
                                                            int nty, int total_quads, int direct, int b0) {
  const int qid = blockIdx.x * 256 + threadIdx.x;
  if (qid >= total_quads) return;
  const int qpr = (W + 3) >> 2;  // quads per row
  const int row = qid / qpr;
  const int x = (qid - row * qpr) * 4;
  const int b = b0 + row / H, y = row - (row / H) * H;      // this launch covers the pairs b0 ...
  combine_quad<TW, TH, R>(slabs, offs[b], g_d2, H, W, ntx, nty, b, y, x, direct);
}

// Round 5: the combine with one block per TILE.  The round-4 kernel above ran one thread per quad of every pixel: 37 % of its
// lanes belonged to quads the tile kernel had already written (exclusive cells) and left at once, every wave first waited
// for its pair's window offset, and a thread had at most four 16-byte loads in flight -- 35 us for 122 MB (3.5 TB/s).  Here a
// block owns one tile of the pair's (unshifted) tile grid; for a pair whose windows are not shifted (mean flow below 4 px:
// the common case) it enumerates ONLY the ring quads of its tile -- (R+1) + R full rows and (R+4)/4 + R/4 quads of every
// other row, 483 of 768 for a 96 x 32 tile -- two per thread with all their slab loads requested before the first sum, and
// knows from the quad's position which of the nine neighbouring windows cover it (same fixed order dj outer / di inner:
// bit-identical to combine_quad).  A pair with shifted windows takes combine_quad over the block's 96 x 32 image region.
template <int TW, int TH, int R>
__global__ __launch_bounds__(256) void combine_tiles_kernel(const float* __restrict__ slabs, const int2* __restrict__ offs,
                                                            float* __restrict__ g_d2, int H, int W, int ntx, int nty,
                                                            int direct, int b0) {
  constexpr int WW = TW + 2 * R + 4, WH = TH + 2 * R + 1;
  constexpr int QW = TW / 4;
  constexpr int kTop = R + 1, kBot = R, kLeft = (R + 4) / 4, kRight = R / 4;       // ring rows / ring quads of a middle row
  constexpr int kFull = (kTop + kBot) * QW, kRing = kFull + (TH - kTop - kBot) * (kLeft + kRight);
  static_assert(R % 4 == 0 && TW % 4 == 0 && 2 * R + 4 <= TW && 2 * R + 1 <= TH, "tile geometry");
  const int tiles = ntx * nty;
  const int b = b0 + blockIdx.x / tiles, t = blockIdx.x % tiles;
  const int tj = t / ntx, ti = t - tj * ntx;
  const int2 off = offs[b];
  if (off.x != 0 || off.y != 0 || !direct || (W & 3) != 0) {
    // shifted windows (or no exclusive cells): the general definition over this block's part of the image
    for (int q = threadIdx.x; q < QW * TH; q += 256) {
      const int hy = q / QW, hx = (q - hy * QW) * 4;
      const int x = ti * TW + hx, y = tj * TH + hy;
      if (x < W && y < H) combine_quad<TW, TH, R>(slabs, off, g_d2, H, W, ntx, nty, b, y, x, direct);
    }
    return;
  }
  const float* tile_slab = slabs + ((size_t)(b * nty + tj) * ntx + ti) * (WW * WH);
  float* gb = g_d2 + (size_t)b * H * W;
  constexpr int kPer = (kRing + 255) / 256;
#pragma unroll
  for (int it = 0; it < kPer; ++it) {
    const int idx = it * 256 + threadIdx.x;
    int hy, qx;
    if (idx < kFull) {
      const int r = idx / QW;
      hy = r < kTop ? r : r + (TH - kTop - kBot);
      qx = idx - r * QW;
    } else {
      const int m = idx - kFull, r = m / (kLeft + kRight), k = m - r * (kLeft + kRight);
      hy = kTop + r;
      qx = k < kLeft ? k : QW - (kLeft + kRight) + k;
    }
    const int hx = qx * 4, x = ti * TW + hx, y = tj * TH + hy;
    if (!(idx < kRing && x < W && y < H)) continue;
    // which neighbouring windows cover the quad (the own window always does): rows above / below, columns left / right
    const int dj = hy < kTop ? -1 : (hy >= TH - kBot ? 1 : 0);
    const int di = hx < R + 4 ? -1 : (hx >= TW - R ? 1 : 0);
    const bool okj = dj != 0 && (unsigned)(tj + dj) < (unsigned)nty, oki = di != 0 && (unsigned)(ti + di) < (unsigned)ntx;
    // window coordinates of the quad in tile (ti + a, tj + c): wx = hx + R - a TW, wy = hy + R - c TH.  A window that does
    // not cover the quad is read at the own window's address instead and multiplied by 0: no branch around a load
    const float* p_own = tile_slab + (hy + R) * WW + (hx + R);
    const float* p_hor = oki ? p_own + (ptrdiff_t)di * (WW * WH) - di * TW : p_own;
    const float* p_ver = okj ? p_own + (ptrdiff_t)dj * ntx * (WW * WH) - dj * TH * WW : p_own;
    const float* p_dia = (oki && okj) ? p_own + (ptrdiff_t)(dj * ntx + di) * (WW * WH) - dj * TH * WW - di * TW : p_own;
    // (ext-vector values: a select between two HIP float4 STRUCTS goes through the stack)
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const v4f own = *reinterpret_cast<const v4f*>(p_own);
#if DVD_WARP_COMBINE_MASKED
    v4f hor = zero, ver = zero, dia = zero;            // lane-masked loads: only the windows that cover the quad are read
    if (oki) hor = *reinterpret_cast<const v4f*>(p_hor);
    if (okj) ver = *reinterpret_cast<const v4f*>(p_ver);
    if (oki && okj) dia = *reinterpret_cast<const v4f*>(p_dia);
#else
    v4f hor = *reinterpret_cast<const v4f*>(p_hor);
    v4f ver = *reinterpret_cast<const v4f*>(p_ver);
    v4f dia = *reinterpret_cast<const v4f*>(p_dia);
    hor = oki ? hor : zero;
    ver = okj ? ver : zero;
    dia = (oki && okj) ? dia : zero;
#endif
    // fixed order of combine_quad: dj outer (-1, 0, 1), di inner (-1, 0, 1)
    const bool hfirst = di < 0, vfirst = dj < 0;
    const v4f r0a = hfirst ? dia : ver, r0b = hfirst ? ver : dia;      // the neighbouring row of tiles (dj != 0)
    const v4f r1a = hfirst ? hor : own, r1b = hfirst ? own : hor;      // the own row of tiles
    const v4f t0 = vfirst ? r0a : r1a, t1 = vfirst ? r0b : r1b, t2 = vfirst ? r1a : r0a, t3 = vfirst ? r1b : r0b;
    const v4f s4 = (((zero + t0) + t1) + t2) + t3;
    *reinterpret_cast<v4f*>(gb + y * W + x) = s4;
  }
}

// Second stage: fixed-order sum of the per-block partials (deterministic).
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int n,
                                                               float* __restrict__ sums) {
  __shared__ double sh[1024][4];
  double acc[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)i * 4);
    acc[0] += v.x;
    acc[1] += v.y;
    acc[2] += v.z;
    acc[3] += v.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] += sh[threadIdx.x + s][k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) {
    sums[threadIdx.x] = (float)sh[0][threadIdx.x];
  }
}

// Last launch of the tiled sequence: block 0 sums the per-tile partials (as reduce_partials_kernel),
// the other blocks apply the window-overflow records with hardware fp32 atomics -- the two are independent,
// so they share a launch.
__global__ __launch_bounds__(1024) void warp_finish_kernel(const float* __restrict__ partial, int n,
                                                           float* __restrict__ sums,
                                                           const unsigned* __restrict__ count,
                                                           const int2* __restrict__ rec, unsigned cap, float* g_d2) {
  if (blockIdx.x == 0) {
    __shared__ double sh[1024][4];
    double acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)i * 4);
      acc[0] += v.x;
      acc[1] += v.y;
      acc[2] += v.z;
      acc[3] += v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] += sh[threadIdx.x + s][k];
      }
      __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = (float)sh[0][threadIdx.x];
    return;
  }
  if (g_d2 == nullptr) return;
  // one wave per tile's list (round 5: per-tile lists, a few hundred records each at most in the benchmark's flow field)
  const int lane = threadIdx.x & 63, wave = (blockIdx.x - 1) * 16 + (threadIdx.x >> 6), nwaves = (gridDim.x - 1) * 16;
  for (int l = wave; l < n; l += nwaves) {
    unsigned m = count[l];
    if (m > cap) m = cap;
    const int2* lr = rec + (size_t)l * cap;
    for (unsigned i = lane; i < m; i += 64) {
      const int2 r = lr[i];
      unsafeAtomicAdd(g_d2 + r.x, __int_as_float(r.y));
    }
  }
}

// (shared with the strip generation, csrc/warp_strip.hip: n = lists / partial records of the launch sequence)
int launch_warp_finish(const float* partial, int n, float* sums, const unsigned* count, const int2* rec, unsigned cap,
                       float* g_d2, hipStream_t stream) {
  hipLaunchKernelGGL(warp_finish_kernel, dim3(g_d2 ? 129 : 1), dim3(1024), 0, stream, partial, n, sums, count, rec, cap, g_d2);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

__global__ void loss_finalize_kernel(const float* __restrict__ sums, float flow_mul, float disp_mul,
                                     int loss_on_sf, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float den = sums[0] + 1e-8f;
    const float fl = sums[1] / den, dl = sums[2] / den, sl = sums[3] / den;
    out[0] = 1.0f / den;
    out[1] = fl * flow_mul + (loss_on_sf ? sl : dl) * disp_mul;
    out[2] = fl;
    out[3] = dl;
    out[4] = sl;
    out[5] = sums[0];
    out[6] = 0.0f;
    out[7] = 0.0f;
  }
}

static int blocks_x(int HW, int px) { return (HW + 256 * px - 1) / (256 * px); }

// ---- tile selection -----------------------------------------------------------------
constexpr int kR = 8;  // LDS window halo: taps within |flow| <= 8 px stay on chip
struct TileShape {
  int tw, th, nt;
};
static const TileShape kShapes[] = {{96, 32, 512}, {64, 48, 512}, {64, 32, 512}, {64, 32, 256}, {96, 32, 384}};
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

// Variant selection: the production path is the tiled kernel with the auto-chosen tile shape.  The parity tests
// also drive the other shapes, the 4-pixels-per-step mapping and the global-atomics reference variant through
// dvd_warp_loss_select() -- a process-wide test hook, not an environment switch read on every call.
static int g_variant = 0;   // 0 production: strips (csrc/warp_strip.hip) when the call is eligible, else tiles; 1 direct (global
                            // gathers + hardware atomics); 2 tiles (the production kernel of rounds 2-5) whatever the call
static int g_tile = -1;     // -1 auto, else index into kShapes
static int g_px = 0;        // 0 auto, 2 or 4 pixels per thread-step
static int g_combine = 0;   // 0 per-tile combine (round 5), 1 the per-quad combine of rounds 2-4 (tests: px == 4 selects it too)

// Least padded area wins; ties go to the earlier (larger) shape.
static int choose_shape(int H, int W) {
  if (g_tile >= 0 && g_tile < kNumShapes) return g_tile;
  int best = 0;
  long long best_area = -1;
  for (int i = 0; i < kNumShapes; ++i) {
    const long long ntx = (W + kShapes[i].tw - 1) / kShapes[i].tw, nty = (H + kShapes[i].th - 1) / kShapes[i].th;
    const long long area = ntx * kShapes[i].tw * nty * kShapes[i].th;
    if (best_area < 0 || area < best_area) {
      best_area = area;
      best = i;
    }
  }
  return best;
}

struct Plan {
  int shape, ntx, nty, ww, wh;
  size_t n_partials, off_count, off_offs, off_slabs, off_ovf, ovf_cap, total;
};

static Plan make_plan(int B, int H, int W) {
  Plan p;
  p.shape = choose_shape(H, W);
  const TileShape& t = kShapes[p.shape];
  p.ntx = (W + t.tw - 1) / t.tw;
  p.nty = (H + t.th - 1) / t.th;
  p.ww = t.tw + 2 * kR + 4;
  p.wh = t.th + 2 * kR + 1;
  const size_t tiles = (size_t)p.ntx * p.nty * B;
  const size_t direct_blocks = (size_t)blocks_x(H * W, 1) * B;
  p.n_partials = tiles > direct_blocks ? tiles : direct_blocks;
  size_t off = p.n_partials * 4 * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  p.off_count = off;
  off += tiles * sizeof(unsigned);
  off = (off + 255) & ~(size_t)255;
  p.off_offs = off;
  off += (size_t)B * sizeof(int2);
  off = (off + 255) & ~(size_t)255;
  p.off_slabs = off;
  off += tiles * (size_t)p.ww * p.wh * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  p.off_ovf = off;
  // per tile: every tap of every pixel -- no list can overflow
  p.ovf_cap = (size_t)t.tw * t.th * 4;
  off += tiles * p.ovf_cap * sizeof(int2);
  p.total = off;
  return p;
}

// Launch sequence: tile kernel -> slab combine -> finish, on one stream (round 5: the prep launch -- counters, window offsets --
// is gone: the offsets are computed by the tiles themselves, the overflow lists are per tile with their counters in LDS).
// Round 4 tried to take the combine (34 us at 48 x 384 x 672) off the serial tail twice; both lost and are not kept:
//   * combine inside the tile kernel by the block that stores the LAST slab of a 3 x 3 tile neighbourhood (bit-identical
//     results, tests green): 2.9 ms instead of 0.24 -- the release / acquire fences the hand-over needs are agent-scope, and on
//     a multi-XCD part an agent-scope fence writes back / invalidates the XCD's whole L2;
//   * two tile launches cut where the full rounds of blocks end, the first cut's combine on a side stream under the second
//     launch: 0.252-0.263 ms against 0.234-0.238 -- the event record / wait pair costs more than the overlap returns.
template <int TW, int TH, int NT>
static int launch_tiled(const WarpArgs& a, const Plan& p, char* ws, bool grads, hipStream_t stream) {
  constexpr int WW = TW + 2 * kR + 4, WH = TH + 2 * kR + 1;
  TileArgs ta;
  ta.slabs = reinterpret_cast<float*>(ws + p.off_slabs);
  ta.ovf.count = reinterpret_cast<unsigned*>(ws + p.off_count);
  ta.ovf.rec = reinterpret_cast<int2*>(ws + p.off_ovf);
  ta.ovf.cap = (unsigned)p.ovf_cap;
  int2* offs = reinterpret_cast<int2*>(ws + p.off_offs);
  ta.offs = offs;
  ta.ntx = p.ntx;
  ta.nty = p.nty;
  ta.direct = ((a.W & 3) == 0 && DVD_WARP_DIRECT_INTERIOR) ? 1 : 0;
  ta.tile0 = 0;
  const int tiles = p.ntx * p.nty, nblocks = tiles * a.B;
  const size_t lds = (size_t)WW * WH * (sizeof(float) + sizeof(unsigned long long)) + kCamLdsFloats * sizeof(float) + 16;
  const bool shipped = a.midas_mask && a.disp_mode == 1 && !a.loss_on_sf;
  // 2 pixels per thread-step when that splits the tile evenly over the block and 4 does not
  constexpr bool kEven4 = ((TW / 4) * TH) % NT == 0, kEven2 = ((TW / 2) * TH) % NT == 0;
  const bool px2 = (g_px ? g_px : ((kEven2 && !kEven4) ? 2 : 4)) == 2;
  auto tile_launch = [&](int pair0, int npairs) -> int {
    TileArgs t2 = ta;
    t2.tile0 = pair0 * tiles;
    const int nb = npairs * tiles;
#define DVD_TILED_LAUNCH(G, S)                                                                            \
  do {                                                                                                    \
    auto k = px2 ? warp_loss_tiled_kernel<TW, TH, kR, NT, G, S, 2> : warp_loss_tiled_kernel<TW, TH, kR, NT, G, S, 4>; \
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                      \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
    hipLaunchKernelGGL(k, dim3(nb), dim3(NT), lds, stream, a, t2);                                        \
  } while (0)
#ifdef DVD_WARP_QUICK
    {
#ifndef DVD_WARP_QUICK_GRADS
#define DVD_WARP_QUICK_GRADS true
#endif
      auto k = warp_loss_tiled_kernel<TW, TH, kR, NT, DVD_WARP_QUICK_GRADS, true, 2>;
      hipLaunchKernelGGL(k, dim3(nb), dim3(NT), lds, stream, a, t2);
      return DVD_OK;
    }
#else
    if (grads) {
      if (shipped)
        DVD_TILED_LAUNCH(true, true);
      else
        DVD_TILED_LAUNCH(true, false);
    } else {
      if (shipped)
        DVD_TILED_LAUNCH(false, true);
      else
        DVD_TILED_LAUNCH(false, false);
    }
#endif
#undef DVD_TILED_LAUNCH
    DVD_LAUNCH_OK();
    return DVD_OK;
  };
  auto combine_launch = [&](int pair0, int npairs, hipStream_t st) -> int {
    const int qpr = (a.W + 3) / 4;
    const int total_quads = qpr * a.H * npairs;
    if (DVD_WARP_COMBINE_TILES && g_combine == 0)
      hipLaunchKernelGGL((combine_tiles_kernel<TW, TH, kR>), dim3(npairs * tiles), dim3(256), 0, st, ta.slabs,
                         (const int2*)offs, a.g_d2, a.H, a.W, p.ntx, p.nty, ta.direct, pair0);
    else
      hipLaunchKernelGGL((combine_slabs_kernel<TW, TH, kR>), dim3((total_quads + 255) / 256), dim3(256), 0, st, ta.slabs,
                         (const int2*)offs, a.g_d2, a.H, a.W, p.ntx, p.nty, total_quads, ta.direct, pair0);
    DVD_LAUNCH_OK();
    return DVD_OK;
  };
  if (int e = tile_launch(0, a.B)) return e;
  if (grads)
    if (int e = combine_launch(0, a.B, stream)) return e;
  // partial-sum reduction and overflow records in one launch
  hipLaunchKernelGGL(warp_finish_kernel, dim3(grads ? 129 : 1), dim3(1024), 0, stream, a.partial, nblocks, a.sums,
                     ta.ovf.count, ta.ovf.rec, ta.ovf.cap, grads ? a.g_d2 : (float*)nullptr);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

static int run(const dvd_warp_cfg* cfg, const float* depth_1, const float* depth_2, const float* flow_1_2,
               const float* mask_2, const float* sf_1_2, const dvd_cameras* cams, void* workspace,
               size_t workspace_bytes, float* sums, float* g_depth_1, float* g_depth_2, float* g_sf_1_2,
               bool grads, hipStream_t stream) {
  DVD_REQUIRE(cfg && cams, "warp_loss: null cfg/cameras");
  DVD_REQUIRE(cfg->B > 0 && cfg->H > 1 && cfg->W > 1, "warp_loss: bad shape B=%d H=%d W=%d", cfg->B, cfg->H,
              cfg->W);
  DVD_REQUIRE(cfg->B <= 65535, "warp_loss: B=%d exceeds grid.y limit", cfg->B);
  DVD_REQUIRE(depth_1 && depth_2 && flow_1_2 && mask_2 && sf_1_2 && sums && workspace,
              "warp_loss: null tensor pointer");
  DVD_REQUIRE(cams->R_1 && cams->R_2 && cams->R_2_T && cams->t_1 && cams->t_2 && cams->K && cams->K_inv,
              "warp_loss: null camera pointer");
  DVD_REQUIRE(cfg->disp_mode >= 0 && cfg->disp_mode <= 2, "warp_loss: disp_mode %d", cfg->disp_mode);
  if (grads) DVD_REQUIRE(g_depth_1 && g_depth_2 && g_sf_1_2, "warp_loss: null gradient pointer");
  const int HW = cfg->H * cfg->W;
  DVD_REQUIRE((long long)cfg->B * HW * 3 < (1LL << 31), "warp_loss: tensor too large for 32-bit indexing");
  const Plan plan = make_plan(cfg->B, cfg->H, cfg->W);
  const StripPlan splan = make_strip_plan(cfg->B, cfg->H, cfg->W);
  const size_t need = plan.total > splan.total ? plan.total : splan.total;
  if (workspace_bytes < need) {
    set_error("warp_loss: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  DVD_REQUIRE(((uintptr_t)workspace & 255) == 0, "warp_loss: workspace must be 256-byte aligned");
  WarpArgs a;
  a.d1 = depth_1;
  a.d2 = depth_2;
  a.flow = flow_1_2;
  a.mask = mask_2;
  a.sf = sf_1_2;
  a.R1 = cams->R_1;
  a.R2 = cams->R_2;
  a.R2T = cams->R_2_T;
  a.t1 = cams->t_1;
  a.t2 = cams->t_2;
  a.K = cams->K;
  a.Ki = cams->K_inv;
  a.partial = static_cast<float*>(workspace);
  a.sums = sums;
  a.g_d1 = g_depth_1;
  a.g_d2 = g_depth_2;
  a.g_sf = g_sf_1_2;
  a.B = cfg->B;
  a.H = cfg->H;
  a.W = cfg->W;
  a.HW = HW;
  a.midas_mask = cfg->midas_mask;
  a.crit_l2 = cfg->crit_l2;
  a.disp_mode = cfg->disp_mode;
  a.loss_on_sf = cfg->loss_on_sf;
  a.flow_mul = cfg->flow_mul;
  a.disp_mul = cfg->disp_mul;
  a.half_w = (float)((cfg->W - 1) / 2.0);
  a.half_h = (float)((cfg->H - 1) / 2.0);
  a.wmax = (float)(cfg->W - 1);
  a.hmax = (float)(cfg->H - 1);
  const bool all16 = (((uintptr_t)depth_1 | (uintptr_t)depth_2 | (uintptr_t)flow_1_2 | (uintptr_t)mask_2 |
                       (uintptr_t)sf_1_2 | (uintptr_t)g_depth_1 | (uintptr_t)g_depth_2 | (uintptr_t)g_sf_1_2) &
                      15) == 0;
  DVD_REQUIRE(all16 || (cfg->W & 3) != 0, "warp_loss: tensors must be 16-byte aligned when W %% 4 == 0");
  if (g_variant == 1) {
    // reference variant: global gathers + hardware atomics (kept for A/B runs and as a second implementation)
    const bool vec4 = (cfg->W % 4 == 0);
    const int nbx = blocks_x(HW, vec4 ? 4 : 1);
    dim3 grid(nbx, cfg->B), block(256);
    if (grads) DVD_HIP_OK(hipMemsetAsync(g_depth_2, 0, (size_t)cfg->B * HW * sizeof(float), stream));
    if (grads) {
      if (vec4)
        hipLaunchKernelGGL((warp_loss_kernel<4, true>), grid, block, 0, stream, a);
      else
        hipLaunchKernelGGL((warp_loss_kernel<1, true>), grid, block, 0, stream, a);
    } else {
      if (vec4)
        hipLaunchKernelGGL((warp_loss_kernel<4, false>), grid, block, 0, stream, a);
      else
        hipLaunchKernelGGL((warp_loss_kernel<1, false>), grid, block, 0, stream, a);
    }
    DVD_LAUNCH_OK();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, stream, a.partial, nbx * cfg->B, sums);
    DVD_LAUNCH_OK();
    return DVD_OK;
  }
  char* ws = static_cast<char*>(workspace);
  // Production path (round 6): the strip kernel -- backward pass, shipped flag set (--midas --use_disp), rows of whole
  // 16-byte quads.  Everything else (forward only, the other loss modes, odd widths, the test hooks' tile shapes) stays on
  // the tile kernel below.  Cameras outside the lockstep loop's preconditions are handled inside either kernel.
  if (g_variant == 0 && g_tile < 0 && g_px == 0 && grads && a.midas_mask && a.disp_mode == 1 && !a.loss_on_sf &&
      (cfg->W & 3) == 0 && all16)
    return launch_strips(a, splan, ws, stream);
#ifdef DVD_WARP_QUICK      // development: compile the production instantiation only (tools/isa_stats.py ... -DDVD_WARP_QUICK)
  return launch_tiled<96, 32, DVD_WARP_QUICK>(a, plan, ws, grads, stream);
#else
  switch (plan.shape) {
    case 0:
      return launch_tiled<96, 32, 512>(a, plan, ws, grads, stream);
    case 1:
      return launch_tiled<64, 48, 512>(a, plan, ws, grads, stream);
    case 2:
      return launch_tiled<64, 32, 512>(a, plan, ws, grads, stream);
    case 4:
      return launch_tiled<96, 32, 384>(a, plan, ws, grads, stream);
    default:
      return launch_tiled<64, 32, 256>(a, plan, ws, grads, stream);
  }
#endif
}

}  // namespace dvd

extern "C" {

int dvd_warp_loss_select(int variant, int tile, int px) {
  DVD_REQUIRE(variant >= 0 && variant <= 2 && tile >= -1 && tile < dvd::kNumShapes && (px == 0 || px == 2 || px == 4),
              "warp_loss_select: variant %d tile %d px %d", variant, tile, px);
  dvd::g_variant = variant;
  dvd::g_tile = tile;
  dvd::g_px = px;
  dvd::g_combine = px == 4 ? 1 : 0;      // the 4-pixel test variant also keeps the per-quad combine of rounds 2-4 covered
  return DVD_OK;
}

int dvd_warp_loss_strip_select(int rows, int shape) {
  DVD_REQUIRE(dvd::strip_select(rows, shape) == 0,
              "warp_loss_strip_select: rows %d shape %d (rows: 0 = automatic, else a multiple of the shape's step height, at least two steps)", rows, shape);
  return DVD_OK;
}

size_t dvd_warp_loss_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  // block partial sums + overflow counters + accumulator slabs + overflow lists, of whichever generation needs more
  const size_t t = dvd::make_plan(B, H, W).total, s = dvd::make_strip_plan(B, H, W).total;
  return t > s ? t : s;
}

int dvd_warp_loss_fused(const dvd_warp_cfg* cfg, const float* depth_1, const float* depth_2,
                        const float* flow_1_2, const float* mask_2, const float* sf_1_2,
                        const dvd_cameras* cams, void* workspace, size_t workspace_bytes, float* sums,
                        float* g_depth_1, float* g_depth_2, float* g_sf_1_2, dvd_stream_t stream) {
  return dvd::run(cfg, depth_1, depth_2, flow_1_2, mask_2, sf_1_2, cams, workspace, workspace_bytes, sums,
                  g_depth_1, g_depth_2, g_sf_1_2, true, static_cast<hipStream_t>(stream));
}

int dvd_warp_loss_fwd(const dvd_warp_cfg* cfg, const float* depth_1, const float* depth_2,
                      const float* flow_1_2, const float* mask_2, const float* sf_1_2,
                      const dvd_cameras* cams, void* workspace, size_t workspace_bytes, float* sums,
                      dvd_stream_t stream) {
  return dvd::run(cfg, depth_1, depth_2, flow_1_2, mask_2, sf_1_2, cams, workspace, workspace_bytes, sums,
                  nullptr, nullptr, nullptr, false, static_cast<hipStream_t>(stream));
}

int dvd_loss_finalize(const dvd_warp_cfg* cfg, const float* sums, float* scalars, dvd_stream_t stream) {
  DVD_REQUIRE(cfg && sums && scalars, "loss_finalize: null pointer");
  hipLaunchKernelGGL(dvd::loss_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums,
                     cfg->flow_mul, cfg->disp_mul, cfg->loss_on_sf, scalars);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
