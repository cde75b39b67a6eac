// Fused warp + losses, forward + backward: the STRIP kernel (round 6) -- the production path of dvd_warp_loss_fused for the
// shipped flag set.  Same arithmetic as csrc/warp_loss.hip (the per-pixel functions are shared: csrc/warp_pixel.h); what is
// new is how a block walks the image.
//
// What it replaces (reference, /root/reference): losses/scene_flow_projection.py:114-153 (flow_by_depth),
// :222-278 (scene_flow_projection_slack), :103-112,212-220 (backward_warp / F.grid_sample),
// models/scene_flow_motion_field.py:140-150,285-324 (disp_loss, _calc_loss), and their autograd backward.
//
// Why.  The tile kernel of rounds 2-5 gives every 96 x 32 tile to a block of its own: camera + window offset, fill of a
// 116 x 49 depth_2 window, barrier, three thread-steps, barrier, flush of the 116 x 49 accumulator window -- a serial chain per
// tile with two blocks per CU to overlap it.  Its counters (profiles/r05_warp_loss_sq_counters.txt) say 49 % VALU-busy, waves
// parked 57-59 % of their life, and 63 % of all pixels go through the slab round trip (flush -> combine kernel) because a
// 96 x 32 tile shares a ring of 8 + 9 rows and 12 + 8 columns with its neighbours.
//
// Here a block owns a UNIT = a strip of TW = 96 columns x SH rows (SH = 128 at 384 x 672) of one pair and walks DOWN it in
// steps of TH = 16 rows with both LDS windows as rings of C = 2 TH + 2 R + 1 = 49 window rows (+ 1 ghost row):
//   * step k reads / scatters window rows T_k = [k TH, k TH + TH + 2R + 1);
//   * the depth_2 rows step k + 1 adds, N_{k+1} = TH rows, are requested at the START of step k by LDS-direct loads
//     (global_load_lds_dwordx4: no VGPR holds them) into the ring slots of the rows step k - 1 was the last to read;
//   * the accumulator rows step k - 1 finished, F_{k-1} = TH rows, are converted, stored and cleared DURING step k (they are
//     disjoint from T_k) -- exclusive cells straight into g_depth_2, the others to the unit's slab;
//   => ONE barrier per step, no fill or flush phase on the critical path after the unit's first window, camera / window
//      offset / pinhole tests once per unit instead of once per tile, 116 x 16 window cells filled and flushed per 96 x 16
//      pixels instead of 116 x 49 per 96 x 32 (-59 %), and only the 12 + 8 border columns (and 17 rows per SH) go through
//      slabs: 31 % of the pixels at SH = 128 instead of 63 %.
//   * three waves per SIMD (384-thread blocks, two per CU: 2 x 69.8 KB of LDS) = 168 VGPRs: the two-pixel lockstep loop
//     needs no scratch and ~26 % fewer instructions than under the 128-register budget of four waves (DESIGN.md section 7.1
//     had measured that build as slower ON TILES: the per-tile chain was exposed at 12 waves per CU; there is none here).
// The ghost row: taps are read / scattered as row pairs (s, s + 1) of the ring; for s = C - 1 the lower row is ring row 0,
// which is kept a second time as row C (filled together with row 0; its accumulator cells are added to row 0's at the flush),
// so that the hot loop never wraps inside a tap quad.
//
// Roofline: HBM, 52 B per pixel-pair algorithmic.  Everything that decides an index or a mask is the
// reference's fp32 rounding sequence (EXACT class, csrc/warp_pixel.h); g_depth_2 is accumulated in Q31.32 fixed point in
// LDS (integer adds commute: bitwise reproducible), slabs are summed in a fixed order by combine_units_kernel.

#include "warp_pixel.h"

#ifndef DVD_STRIP_UNROLL
#define DVD_STRIP_UNROLL 0          // 1: the two thread-steps of a step as two code copies (precise waits, but loop invariants spill)
#endif
#ifndef DVD_STRIP_END_WAIT
#define DVD_STRIP_END_WAIT -1       // >= 0: vmcnt of the step's closing wait for A/B builds (default: what the walk guarantees)
#endif
#ifndef DVD_STRIP_CAM_REGS
#define DVD_STRIP_CAM_REGS 1        // 1: shapes with a 168-register budget keep the pair's camera in vector registers (0: LDS broadcasts)
#endif
#ifndef DVD_STRIP_COMBINE2
#define DVD_STRIP_COMBINE2 0        // 1: the unit combine requests two quads' slabs before it sums either (measured: 21 us against 16-18)
#endif
#ifndef DVD_STRIP_PERSISTENT
#define DVD_STRIP_PERSISTENT 0
#endif
#ifndef DVD_STRIP_PF_EARLY
#define DVD_STRIP_PF_EARLY 0        // 1: the next thread-step's inputs are requested at the START of a thread-step (0: between its
                                   // phases: 186.8 against 187.8 us, median of six interleaved runs -- the extra live registers cost
                                   // what the longer lead buys)
#endif
#ifndef DVD_STRIP_KO_BARRIER
#define DVD_STRIP_KO_BARRIER 0
#endif
#ifndef DVD_STRIP_KO_PIXEL
#define DVD_STRIP_KO_PIXEL 0
#endif
#ifndef DVD_STRIP_KO_FLUSH          // knock-out builds (timing studies only; results are wrong)
#define DVD_STRIP_KO_FLUSH 0
#endif
#ifndef DVD_STRIP_KO_FILL
#define DVD_STRIP_KO_FILL 0
#endif

namespace dvd {

// R: halo of the window in x (R columns to the left, R + 4 to the right: rows of whole quads); RY: halo in y (RY rows above,
// RY + 1 below).  RY may exceed R where the LDS has room: every row of a step sits RY rows from the window's edge at the
// step's first / last row, so with RY = 8 about one pixel pair per 96 x 32 step of the benchmark's sigma = 3 px flow field
// leaves the window -- one wave of the block then runs the global-gather path while eleven wait at the step's barrier (14 us of
// the kernel, profiles/r06_warp_strip_experiments.txt); RY = 12 is 4 sigma.
template <int TW, int TH, int R, int RY>
struct StripGeo {
  static constexpr int WW = TW + 2 * R + 4;      // cells per window row (multiple of 4: rows are 16-byte aligned)
  static constexpr int WH = TH + 2 * RY + 1;     // window rows one step touches
  static constexpr int C = 2 * TH + 2 * RY + 1;  // ring rows
  static constexpr int QR = WW / 4;              // 16-byte quads per window row
  static_assert(R % 4 == 0 && TW % 4 == 0 && 2 * RY + 1 <= 2 * TH, "strip geometry (a unit is at least two steps: only adjacent units overlap)");
};

constexpr int strip_lds_bytes(int tw, int th, int r, int ry, int nt) {
  return (tw + 2 * r + 4) * (2 * th + 2 * ry + 2) * 12 + kCamLdsFloats * 4 + 16 + (nt / 64) * 16;
}

// IO policy of a strip (the interface pixel() / pixel2() expect, like TileIO): the on-chip window of the CURRENT step is
// window rows [wyk, wyk + WH) of the image, window row wyk lives in ring slot sk.
template <int WW, int WH, int C>
struct RingIO {
  static constexpr int kWW = WW;
  const float* d2b;          // depth_2 of this pair
  float* win;                // LDS [C + 1][WW]
  unsigned long long* accw;  // LDS [C + 1][WW], Q31.32
  int W, wx0, wyk, sk, pair_base, list;
  float unit;
  Overflow ovf;
  unsigned* lcount;
  __device__ __forceinline__ int cell(int lx, int rr) const {
    unsigned s = (unsigned)(rr + sk);
    s = min(s, s - (unsigned)C);               // s >= C ? s - C : s   (s < 2 C)
    return (int)s * WW + lx;
  }
  __device__ __forceinline__ bool inside(int x0, int y0) const {
    return ((unsigned)(x0 - wx0) < (unsigned)(WW - 1)) & ((unsigned)(y0 - wyk) < (unsigned)(WH - 1));
  }
  __device__ __forceinline__ bool cells2(int x0A, int y0A, int x0B, int y0B, int& cellA, int& cellB) const {
    const int lxA = x0A - wx0, rA = y0A - wyk, lxB = x0B - wx0, rB = y0B - wyk;
    cellA = cell(lxA, rA);
    cellB = cell(lxB, rB);
    return ((unsigned)lxA < (unsigned)(WW - 1)) & ((unsigned)rA < (unsigned)(WH - 1)) &
           ((unsigned)lxB < (unsigned)(WW - 1)) & ((unsigned)rB < (unsigned)(WH - 1));
  }
  __device__ __forceinline__ void fetch(int o_n, int x0, int y0, bool in_e, bool in_s, float& dnw, float& dne,
                                        float& dsw, float& dse) const {
    if (inside(x0, y0)) {
      const float* p = win + cell(x0 - wx0, y0 - wyk);      // (the row below is p + WW also for the last slot: ghost row)
      dnw = p[0];
      dne = p[1];
      dsw = p[WW];
      dse = p[WW + 1];
    } else {
      DirectIO g{d2b, nullptr, W, 1.0f};
      g.fetch(o_n, x0, y0, in_e, in_s, dnw, dne, dsw, dse);
      asm volatile("" : "+v"(dnw), "+v"(dne), "+v"(dsw), "+v"(dse));      // consumed here (see TileIO::fetch)
    }
  }
  __device__ __forceinline__ void spill(int idx, float v) const {
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(lcount, (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    const unsigned i = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (i < ovf.cap) ovf.rec[(size_t)list * ovf.cap + i] = make_int2(pair_base + idx, __float_as_int(v * unit));
  }
  __device__ __forceinline__ void add_fixed(unsigned long long* p, int idx, float v) const {
    if (fabsf(v) < kFixMax)
      atomicAdd(p, to_fixed(v));
    else
      spill(idx, v);
  }
  __device__ __forceinline__ void scatter(int o_n, int x0, int y0, bool in_e, bool in_s, float tnw, float tne,
                                          float tsw, float tse) const {
    if (inside(x0, y0)) {
      unsigned long long* p = accw + cell(x0 - wx0, y0 - wyk);
      const float big = (fabsf(tnw) + fabsf(tne)) + (fabsf(tsw) + fabsf(tse));      // (NaN-aware: see TileIO::scatter)
      if (big < kFixMax) {
        atomicAdd(p, to_fixed(tnw));
        atomicAdd(p + 1, to_fixed(tne));
        atomicAdd(p + WW, to_fixed(tsw));
        atomicAdd(p + WW + 1, to_fixed(tse));
      } else {
        add_fixed(p, o_n, tnw);
        if (in_e) add_fixed(p + 1, o_n + 1, tne);
        if (in_s) add_fixed(p + WW, o_n + W, tsw);
        if (in_e && in_s) add_fixed(p + WW + 1, o_n + W + 1, tse);
      }
    } else {
      spill(o_n, tnw);
      if (in_e) spill(o_n + 1, tne);
      if (in_s) spill(o_n + W, tsw);
      if (in_e && in_s) spill(o_n + W + 1, tse);
    }
  }
};

struct StripArgs {
  float* slabs;
  Overflow ovf;
  int2* offs;         // per pair: window offset, written by the pair's first unit (for the combine and finish kernels)
  int ntx, nseg, SH;  // strips per image row, units per strip column, rows per unit (multiple of TH, >= 2 TH)
  int direct;         // 1: window cells no neighbouring unit covers go straight to g_depth_2
  unsigned slab_stride;   // floats per unit slab = WW * (SH + 2RY + 1)
  int n_units;
};

// Window cell (wx, r) of a unit is covered by that unit ALONE when wx in [2R + 4, TW) and r in [2RY + 1, SH): tile_exclusive
// with the unit's height.  The strip kernel and both combine forms use this one predicate.
template <int TW, int R, int RY>
__device__ __forceinline__ bool unit_exclusive(int wx, int r, int SH) {
  return wx >= 2 * R + 4 && wx < TW && r >= 2 * RY + 1 && r < SH;
}

// 16 bytes global -> LDS without a register in between (LDS-DMA).  lds_byte: wave-uniform LDS byte address; lane l's 16 bytes
// land at lds_byte + 16 l (lanes switched off by EXEC write nothing).  M0 is compiler-reserved: saved, set and restored in
// the statement that uses it (cdna_hip_programming.md, inline-asm section).  Not counted by the compiler's s_waitcnt
// bookkeeping: the caller waits vmcnt(0) before the barrier that publishes the rows.
__device__ __forceinline__ void glds16(const float* base, unsigned byte_off, unsigned lds_byte) {
  // (scalar base + one 32-bit lane offset: a 64-bit lane address costs a register pair that the walk's loop spilled)
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(byte_off), "s"(lds_byte), "s"(base)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// FULL: the image is whole strips wide, whole steps and whole units high (W % TW == 0, H % TH == 0, H % SH == 0: 384 x 672,
// 768 x 1344, 192 x 384) -- no
// thread-step lies outside the image, so the lockstep walk is straight-line code: its gradient stores are unconditional, and
// the compiler can count them when it places the wait for the next thread-step's inputs (behind CONDITIONAL stores that wait
// is s_waitcnt vmcnt(0): the in-order counter cannot tell whether the newest operations are the loads or stores behind them).
// BPC: blocks per CU the shape is meant for (register budget = 512 / (BPC * NT / 256) VGPRs).  A block's waves must spread EVENLY
// over the CU's four SIMDs (NT a multiple of 256): the first strip shape, 384-thread blocks at two per CU, measured 40 % of the
// expected wave residency -- the six waves of a block land 2, 2, 1, 1 on the SIMDs, the second block of a CU only fits next to
// the first when its doubled SIMDs are the other two, and the dispatcher does not arrange that (profiles/r06_warp_strip_experiments.txt).
template <int TW, int TH, int R, int RY, int NT, int BPC, bool FULL>
__global__ __launch_bounds__(NT, (BPC * NT + 255) / 256) void warp_loss_strip_kernel(const WarpArgs a, const StripArgs sa) {
  using G = StripGeo<TW, TH, R, RY>;
  constexpr int WW = G::WW, WH = G::WH, C = G::C, QR = G::QR;
  constexpr int QW = TW / 2;                     // pixel pairs per row of the strip
  constexpr int NW = NT / 64;
  static_assert((QW * TH) % NT == 0, "a step's pixel pairs must split evenly over the block");
  constexpr int NTS = (QW * TH) / NT;            // thread-steps per step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long* accw = reinterpret_cast<unsigned long long*>(smem);   // [C + 1][WW] u64
  float* win = smem + 2 * WW * (C + 1);                                      // [C + 1][WW]
  float* camL = win + WW * (C + 1);
  unsigned* lcount = reinterpret_cast<unsigned*>(camL + kCamLdsFloats);
  float* red = camL + kCamLdsFloats + 4;                                     // [NW][4] block sums
  static_assert((3 * WW * (C + 1)) % 4 == 0, "camera quads must be 16-byte aligned");

  // ---- DVD_STRIP_PERSISTENT = 1 (an experiment, off: measured 207 us against 171 us -- the outer loop's live values push the
  //      walk's loop invariants to scratch): persistent blocks (one or two per CU), a block evaluates several units.  Units are numbered so
  //      that each XCD (block b runs on XCD b % 8: speed only, never correctness) owns a contiguous run of them -- the blocks
  //      of an XCD take the run's units round robin, so the units in flight on an XCD at any time are neighbours (adjacent
  //      strips of the same pair share their halo columns of depth_2 in that XCD's L2).  What a unit's start costs in a grid of
  //      one block per unit -- block launch, clearing 76 KB of accumulator ring -- is paid once per block: every accumulator
  //      cell a unit touched has been flushed AND cleared when the unit ends.
  for (int i = threadIdx.x; i < (C + 1) * WW / 2; i += NT) reinterpret_cast<uint4*>(accw)[i] = make_uint4(0u, 0u, 0u, 0u);
#if DVD_STRIP_PERSISTENT
  const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3, per_xcd = ((int)gridDim.x - xcd + 7) >> 3;
  const int run_q = sa.n_units >> 3, run_r = sa.n_units & 7;
  const int run_lo = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
  const int run_n = run_q + (xcd < run_r ? 1 : 0);
  for (int run_i = in_xcd; run_i < run_n; run_i += per_xcd) {
  const int logical = run_lo + run_i;
#else
  {
  const int logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
#endif
  const int upp = sa.ntx * sa.nseg;              // units per pair
  const int b = logical / upp;
  const int t = logical - b * upp;
  const int sj = t / sa.ntx, ti = t - sj * sa.ntx;
  const int tx0 = ti * TW, uy0 = sj * sa.SH;
  const int nsteps = sa.SH / TH;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int2 off = pair_window_offset(a.flow, b, a.H, a.W);
  if (t == 0 && threadIdx.x == 0) sa.offs[b] = off;
  const int wx0 = tx0 - R + off.x, wy_top = uy0 - RY + off.y;
  const float* d2b = a.d2 + (size_t)b * a.HW;
  Cam c;
  load_cam(a, b, c);
  const bool pinhole = c.Ki[1] == 0.0f && c.Ki[2] == 0.0f && c.Ki[3] == 0.0f && c.Ki[5] == 0.0f && c.Ki[8] == 1.0f &&
                       c.K[1] == 0.0f && c.K[2] == 0.0f && c.K[3] == 0.0f && c.K[5] == 0.0f && c.K[8] == 1.0f;
  bool r2t = true;            // R_2 and R_2_T are each other's transposes, bit for bit (wave-uniform)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r2t = r2t && (c.R2[3 * i + j] == c.R2T[3 * j + i]);
  const bool lockstep = pinhole && r2t;

  // ---- depth_2 rows [r0, r0 + n) of the unit's window -> ring (n <= C).  Quads outside the image are written as zeros (what
  //      ATen's masked gather returns); a quad is inside or outside as a whole: W, the window origin and the quad are multiples
  //      of 4.  Consecutive rows are consecutive in the ring except at its end, so the rows are requested as (at most) two
  //      runs, each a sequence of 1 KB wave-instructions whose 64 lanes write 64 consecutive quads.
  const unsigned win_b = lds_addr(win);
  auto fill_run = [&](int lane, int r0, int s0, int nrows) {          // window rows r0.., ring slots s0.. (no wrap inside)
    const int quads = nrows * QR;
    for (int q0 = wave * 64; q0 < quads; q0 += NW * 64) {
      const int q = q0 + lane;
      const int j = q / QR, wxq = q - j * QR;
      const int y = wy_top + r0 + j, x = wx0 + wxq * 4;
      const unsigned dst = win_b + (unsigned)(s0 * WW + q0 * 4) * 4u;        // wave-uniform
      if (q < quads) {
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && !DVD_STRIP_KO_FILL)
          glds16(d2b, (unsigned)(y * a.W + x) * 4u, __builtin_amdgcn_readfirstlane(dst));
        else
          *reinterpret_cast<float4*>(win + s0 * WW + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // the row that lands in ring slot 0 is kept a second time as the ghost row C (one wave's 29 quads)
  auto fill_ghost = [&](int lane, int rg) {
    const int q = lane;
    const int y = wy_top + rg, x = wx0 + q * 4;
    const unsigned dst = win_b + (unsigned)(C * WW) * 4u;
    if (q < QR) {
      if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && !DVD_STRIP_KO_FILL)
        glds16(d2b, (unsigned)(y * a.W + x) * 4u, __builtin_amdgcn_readfirstlane(dst));
      else
        *reinterpret_cast<float4*>(win + C * WW + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // (`tid`: the thread index behind an empty asm -- opaque, so that the index arithmetic below is redone per step instead of
  //  hoisted out of the walk's loop, where it was spilled to scratch and reloaded behind s_waitcnt vmcnt(0))
  auto fill = [&](int tid, int r0, int n) {
    const int ln = tid & 63;
    const int s0 = r0 % C;
    const int nA = n < C - s0 ? n : C - s0;
    fill_run(ln, r0, s0, nA);
    if (n > nA) fill_run(ln, r0 + nA, 0, n - nA);
    if ((s0 == 0 || n > nA) && wave == NW - 1) fill_ghost(ln, s0 == 0 ? r0 : r0 + nA);
  };

  // ---- accumulator rows [r0, r0 + n) of the window: convert, store (exclusive cells -> g_depth_2, the others -> the unit's
  //      slab), clear.  The rows are final: no later step scatters into them.
  float* slab = sa.slabs + (size_t)logical * sa.slab_stride;
  float* gb = a.g_d2 + (size_t)b * a.HW;
  const float back = a.disp_mul;
  auto flush_quad = [&](int i, int r0, int s0) {
    const int j = i / QR, wx = (i - j * QR) * 4;
    int s = s0 + j;
    s = s >= C ? s - C : s;
    longlong2* cp = reinterpret_cast<longlong2*>(accw + s * WW + wx);
    longlong2 lo = cp[0], hi = cp[1];
    cp[0] = make_longlong2(0, 0);
    cp[1] = make_longlong2(0, 0);
    if (s == 0) {                                   // + what was scattered through the ghost row
      longlong2* gp = reinterpret_cast<longlong2*>(accw + C * WW + wx);
      const longlong2 glo = gp[0], ghi = gp[1];
      gp[0] = make_longlong2(0, 0);
      gp[1] = make_longlong2(0, 0);
      lo.x += glo.x;
      lo.y += glo.y;
      hi.x += ghi.x;
      hi.y += ghi.y;
    }
    const float4 v = make_float4(from_fixed(lo.x) * back, from_fixed(lo.y) * back, from_fixed(hi.x) * back, from_fixed(hi.y) * back);
    const int r = r0 + j;
    const int x = wx0 + wx, y = wy_top + r;
    // ONE unconditional store (an exclusive cell outside the image goes to its slab cell, which nobody reads): a store
    // count the compiler can see, see FULL above
    const bool to_image = sa.direct && unit_exclusive<TW, R, RY>(wx, r, sa.SH) && (unsigned)x < (unsigned)a.W && (unsigned)y < (unsigned)a.H;
    float* dst = to_image ? gb + (size_t)y * a.W + x : slab + r * WW + wx;
    if (!DVD_STRIP_KO_FLUSH) *reinterpret_cast<float4*>(dst) = v;
  };
  auto flush = [&](int r0, int n) {                  // any number of rows (the unit's last window)
    const int s0 = r0 % C;
    for (int i = threadIdx.x; i < n * QR; i += NT) flush_quad(i, r0, s0);
  };
  static_assert(TH * QR <= 2 * NT, "a step's TH rows are at most two quads per thread");
  auto flush_step = [&](int tid, int r0) {           // TH rows: one quad per thread (+ one more for the first threads)
    const int s0 = r0 % C;
    if (TH * QR >= NT || tid < TH * QR) flush_quad(tid, r0, s0);
    if (TH * QR > NT && tid + NT < TH * QR) flush_quad(tid + NT, r0, s0);
  };

  // ---- the unit's prologue: camera to LDS, first window (the accumulator ring is all zero: cleared before the first unit,
  //      by the flushes since)
  if (threadIdx.x == 0) *lcount = 0u;
  if (threadIdx.x < kCamLdsFloats) {
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < kCamLdsFloats; ++i) v = (int)threadIdx.x == i ? cam_lds_value(c, i) : v;
    camL[threadIdx.x] = v;
  }
  fill(threadIdx.x, 0, WH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  RingIO<WW, WH, C> io{d2b, win, accw, a.W, wx0, wy_top, 0, b * a.HW, logical, a.disp_mul, sa.ovf, lcount};
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  struct In2 {
    v2f d1, mk, s0, s1, s2;
    float4 fl;
  };
  const char* d1b = reinterpret_cast<const char*>(a.d1 + (size_t)b * a.HW);
  const char* mkb = reinterpret_cast<const char*>(a.mask + (size_t)b * a.HW);
  const char* flb = reinterpret_cast<const char*>(a.flow + 2 * (size_t)b * a.HW);
  const char* sfb0 = reinterpret_cast<const char*>(a.sf + (size_t)b * 3 * a.HW);
  char* gd1b = reinterpret_cast<char*>(a.g_d1 + (size_t)b * a.HW);
  char* gsb0 = reinterpret_cast<char*>(a.g_sf + (size_t)b * 3 * a.HW);
  const unsigned plane = (unsigned)a.HW * 4u;
  // thread-step q of the step whose first pixel row is y0: pixel pair (x, x + 1) of row y; false = outside the image
  auto locate = [&](int q, int y0, int& x, int& y) {
    const int ly = q / QW, lx = (q - ly * QW) * 2;
    y = y0 + ly;
    x = tx0 + lx;
    return (y < a.H) && (x < a.W);
  };
  // (branch free: a thread-step outside the image reads the pair's first pixels instead -- valid memory, never used)
  auto fetch2 = [&](int q, int y0) {
    int x, y;
    const bool ok = locate(q, y0, x, y);
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
  const float yhw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rcp_refined(a.half_w))));
  const float yhh = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rcp_refined(a.half_h))));

  // ---- the general one-pixel path (a camera with skew, an R_2_T that is not R_2's transpose): rare, correct for every camera
  auto step_pixels1 = [&](int y0) {
    Cam cc;
    load_cam(a, b, cc);
#pragma unroll
    for (int i = 0; i < 9; ++i) {                 // (the FMA-heavy matrices in VGPRs: 51 scalars do not fit the SGPR file)
      asm volatile("" : "+v"(cc.R1[i]));
      asm volatile("" : "+v"(cc.R2[i]));
      asm volatile("" : "+v"(cc.K[i]));
      asm volatile("" : "+v"(cc.R2T[i]));
    }
    for (int q = threadIdx.x; q < QW * TH; q += NT) {
      int x, y;
      if (!locate(q, y0, x, y)) continue;
      const In2 in = fetch2(q, y0);
      const unsigned o = (unsigned)(y * a.W + x) * 4u;
      const float d1[2] = {in.d1.x, in.d1.y}, mk[2] = {in.mk.x, in.mk.y}, fx[2] = {in.fl.x, in.fl.z}, fy[2] = {in.fl.y, in.fl.w};
      const float s0[2] = {in.s0.x, in.s0.y}, s1[2] = {in.s1.x, in.s1.y}, s2[2] = {in.s2.x, in.s2.y};
      float gd1[2], g0[2], g1[2], g2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float gs[3] = {0.0f, 0.0f, 0.0f};
        gd1[i] = 0.0f;
        pixel<true, true, false>(a, cc, io, y, x + i, d1[i], fx[i], fy[i], mk[i], s0[i], s1[i], s2[i], acc, gd1[i], gs);
        g0[i] = gs[0];
        g1[i] = gs[1];
        g2[i] = gs[2];
      }
      *reinterpret_cast<v2f*>(gd1b + o) = (v2f){gd1[0], gd1[1]};
      *reinterpret_cast<v2f*>(gsb0 + o) = (v2f){g0[0], g0[1]};
      *reinterpret_cast<v2f*>(gsb0 + (o + plane)) = (v2f){g1[0], g1[1]};
      *reinterpret_cast<v2f*>(gsb0 + (o + 2u * plane)) = (v2f){g2[0], g2[1]};
    }
  };

  // ---- the walk: one barrier per step.  Lockstep path (pinhole pair, R_2_T = R_2^T: what the data files hold): a software
  //      pipeline over the unit's thread-steps (two per step) -- the inputs of thread-step i + 1 are requested between the
  //      forward and the backward phase of thread-step i (by a thread whose own pixel pair lies outside the image: on the
  //      spot), ACROSS the step's barrier too, so no input latency is exposed.  That is why a step does not end in
  //      s_waitcnt vmcnt(0): what must have landed before the barrier are this wave's LDS-direct loads, issued FIRST in the
  //      step; behind them every wave issues the six input loads of the next thread-step (whatever its pixels are) and, in a FULL
  //      image, that thread-step's four gradient stores; the counter retires in order, so vmcnt(10) (vmcnt(6) without
  //      FULL) covers the LDS-direct loads and leaves exactly those newest requests in flight.  The barrier is the raw instruction
  //      behind an explicit lgkmcnt(0) (the step's LDS adds / clears / zero fills); __syncthreads() would add the vmcnt(0) of
  //      its workgroup-scope fence.
  // kEndWait: vector-memory instructions every wave is GUARANTEED to issue behind the step's LDS-direct loads -- the six input
  // loads of the next thread-step, and with FULL the four gradient stores of the thread-step they sit in.
  constexpr int kEndWait = DVD_STRIP_END_WAIT >= 0 ? DVD_STRIP_END_WAIT : (FULL ? 10 : 6);
  auto walk2 = [&](auto crit_tag) {
    constexpr bool CRIT = decltype(crit_tag)::value;
    // the camera of the lockstep loop: vector registers where the shape's budget has room for its 32 values (three waves per
    // SIMD), LDS broadcasts otherwise
    constexpr bool kCamRegs = DVD_STRIP_CAM_REGS && (BPC * NT) / 256 <= 3;
    typename std::conditional<kCamRegs, CamRegs, const float*>::type camv;
    if constexpr (kCamRegs)
      camv = cam_regs_from_lds(camL);
    else
      camv = camL;
    In2 cur = fetch2(threadIdx.x, uy0);
    // (consumed here: with these loads still pending at the loop's entry -- issued in another order than the loop's own, no
    //  store behind them -- the compiler's merged state makes the wait for a thread-step's inputs s_waitcnt vmcnt(0), which
    //  then also waits for the previous thread-step's gradient stores and the flush's stores, in EVERY thread-step)
    asm volatile("" : "+v"(cur.d1), "+v"(cur.mk), "+v"(cur.s0), "+v"(cur.s1), "+v"(cur.s2), "+v"(cur.fl.x), "+v"(cur.fl.y),
                 "+v"(cur.fl.z), "+v"(cur.fl.w));
    int sk = 0;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      if (k >= 1) flush_step(tid, (k - 1) * TH);                    // F_{k-1}: disjoint from what this step touches
      // N_{k+1}, the rows the NEXT step adds.  With one thread-step per step they are requested here, behind the flush's
      // stores (in front of them, the first counted wait of the step -- vmcnt(n) for the thread-step's inputs, n from the
      // compiler's count, which does not include the LDS-direct loads -- had to see them land: one exposed HBM latency per
      // step, 31 us of the kernel); with more thread-steps behind the first one, so that they land under the second.
      if (NTS == 1 && k + 1 < nsteps) fill(tid, (k + 1) * TH + 2 * RY + 1, TH);
      io.wyk = wy_top + k * TH;
      io.sk = sk;
      const int y0 = uy0 + k * TH;
#pragma unroll DVD_STRIP_UNROLL ? NTS : 1
      for (int ts = 0; ts < NTS; ++ts) {
        const int q = threadIdx.x + ts * NT;
        const bool last = ts == NTS - 1;
        const int qn = last ? (int)threadIdx.x : q + NT, yn = last ? y0 + TH : y0;
        In2 nxt;
        int x, y;
        const bool ok = locate(q, y0, x, y);
        // DVD_STRIP_PF_EARLY: the next thread-step's inputs requested FIRST (a whole thread-step for them to arrive) instead of
        // between the phases (the backward phase only).  With twelve waves per CU the kernel runs at the rate its memory
        // latency allows -- its time moves one for one with the bytes of a knocked-out stream
        // (profiles/r06_warp_strip_experiments.txt) -- but the longer lead did not pay: see the knob.  The scheduling
        // barrier keeps the early requests from sinking towards their use.
        constexpr bool kEarly = DVD_STRIP_PF_EARLY && kCamRegs;
        if (kEarly) {
          nxt = fetch2(qn, yn);
          __builtin_amdgcn_sched_barrier(0);
        }
        if ((FULL || ok) && !DVD_STRIP_KO_PIXEL) {
          const unsigned o = (unsigned)(y * a.W + x) * 4u;
          pixel2<true, CRIT>(a, camv, io, y, x, cur.d1, (v2f){cur.fl.x, cur.fl.z}, (v2f){cur.fl.y, cur.fl.w}, cur.mk, cur.s0,
                             cur.s1, cur.s2, yhw, yhh, acc,
                             [&]() {
                               if (!kEarly) nxt = fetch2(qn, yn);
                             },
                             [&](v2f gd1, v2f g0, v2f g1, v2f g2) {
                               *reinterpret_cast<v2f*>(gd1b + o) = gd1;
                               *reinterpret_cast<v2f*>(gsb0 + o) = g0;
                               *reinterpret_cast<v2f*>(gsb0 + (o + plane)) = g1;
                               *reinterpret_cast<v2f*>(gsb0 + (o + 2u * plane)) = g2;
                             });
        } else if (!kEarly) {
          nxt = fetch2(qn, yn);
        }
        cur = nxt;
        if (NTS > 1 && ts == 0 && k + 1 < nsteps) fill(tid, (k + 1) * TH + 2 * RY + 1, TH);
      }
      sk += TH;
      sk = sk >= C ? sk - C : sk;
      if (DVD_STRIP_KO_BARRIER)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kEndWait) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kEndWait) : "memory");
    }
  };
  // ---- the general walk (a camera with skew, an R_2_T that is not R_2's transpose: rare): no pipeline, everything waited for
  auto walk1 = [&]() {
    int sk = 0;
    for (int k = 0; k < nsteps; ++k) {
      if (k + 1 < nsteps) fill(threadIdx.x, (k + 1) * TH + 2 * RY + 1, TH);
      if (k >= 1) flush_step(threadIdx.x, (k - 1) * TH);
      io.wyk = wy_top + k * TH;
      io.sk = sk;
      if (uy0 + k * TH < a.H) step_pixels1(uy0 + k * TH);
      sk += TH;
      sk = sk >= C ? sk - C : sk;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  };
  if (lockstep) {
    if (a.crit_l2)
      walk2(std::true_type{});
    else
      walk2(std::false_type{});
  } else {
    walk1();
  }
  flush((nsteps - 1) * TH, WH);                                     // what the last step left in the ring
  if (threadIdx.x == 0) sa.ovf.count[logical] = *lcount < sa.ovf.cap ? *lcount : sa.ovf.cap;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const float v = wave_sum(acc[kk]);
    if (lane == 0) red[wave * 4 + kk] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.0f;
    for (int w = 0; w < NW; ++w) v += red[w * 4 + threadIdx.x];
    a.partial[(size_t)logical * 4 + threadIdx.x] = v;
  }
  }      // next unit of this block (everything of this one that lives in LDS was read before the barrier above)
}

// g_depth_2[b, y, x .. x+3] = sum over the units whose window covers the quad, fixed order (dj outer, di inner): combine_quad
// of warp_loss.hip with the unit height as a run-time value.  Used for pairs whose windows are shifted.
template <int TW, int R, int RY>
__device__ __forceinline__ void combine_quad_units(const float* __restrict__ slabs, unsigned slab_stride, int2 off,
                                                   float* __restrict__ g_d2, int H, int W, int ntx, int nseg, int SH, int b,
                                                   int y, int x, int direct) {
  constexpr int WW = TW + 2 * R + 4;
  const int WHu = SH + 2 * RY + 1;
  const int xs = x - off.x, ys = y - off.y;          // coordinates in the pair's shifted unit grid
  const int ti = xs >= 0 ? xs / TW : -((TW - 1 - xs) / TW), tj = ys >= 0 ? ys / SH : -((SH - 1 - ys) / SH);
  if (direct && ti >= 0 && ti < ntx && tj >= 0 && tj < nseg && unit_exclusive<TW, R, RY>(xs - ti * TW + R, ys - tj * SH + RY, SH)) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dj = -1; dj <= 1; ++dj) {
    const int j = tj + dj;
    const int wy = ys - (j * SH - RY);
    if (j < 0 || j >= nseg || wy < 0 || wy >= WHu) continue;
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
      const int i = ti + di;
      const int wx = xs - (i * TW - R);
      if (i < 0 || i >= ntx || wx < 0 || wx + 3 >= WW) continue;
      const float4 v = *reinterpret_cast<const float4*>(slabs + ((size_t)(b * nseg + j) * ntx + i) * slab_stride + wy * WW + wx);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(g_d2 + ((size_t)b * H + y) * W + x) = s;
}

// One block per unit: sums the slabs over the RING quads of the unit's home region (its 96 x SH pixels): RY + 1 rows at the top,
// RY at the bottom, (R + 4) / 4 + R / 4 quads of every other row -- 31 % of the quads at SH = 128; everything else was written
// by the strip kernel itself.  Which of the nine neighbouring windows cover a quad follows from its position; fixed order
// dj outer / di inner like combine_quad_units (bit-identical to it).  Pairs with shifted windows take combine_quad_units.
template <int TW, int R, int RY>
__global__ __launch_bounds__(256) void combine_units_kernel(const float* __restrict__ slabs, unsigned slab_stride,
                                                            const int2* __restrict__ offs, float* __restrict__ g_d2, int H, int W,
                                                            int ntx, int nseg, int SH, int direct) {
  constexpr int WW = TW + 2 * R + 4;
  constexpr int QW = TW / 4;
  constexpr int kTop = RY + 1, kBot = RY, kLeft = (R + 4) / 4, kRight = R / 4;
  const int upp = ntx * nseg;
  const int logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int b = logical / upp, t = logical - b * upp;
  const int sj = t / ntx, ti = t - sj * ntx;
  const int2 off = offs[b];
  if (off.x != 0 || off.y != 0 || !direct) {
    for (int q = threadIdx.x; q < QW * SH; q += 256) {
      const int hy = q / QW, hx = (q - hy * QW) * 4;
      const int x = ti * TW + hx, y = sj * SH + hy;
      if (x < W && y < H) combine_quad_units<TW, R, RY>(slabs, slab_stride, off, g_d2, H, W, ntx, nseg, SH, b, y, x, direct);
    }
    return;
  }
  const float* unit_slab = slabs + (size_t)logical * slab_stride;      // (logical = (b * nseg + sj) * ntx + ti)
  float* gb = g_d2 + (size_t)b * H * W;
  const int kFull = (kTop + kBot) * QW, kRing = kFull + (SH - kTop - kBot) * (kLeft + kRight);
  const ptrdiff_t row_of_units = (ptrdiff_t)ntx * slab_stride;
  // (DVD_STRIP_COMBINE2: two quads per pass, the loads of both requested before either is summed -- no gain, off)
  struct Quad {
    v4f own, hor, ver, dia;
    float* dst;
    int di, dj;
  };
  auto request = [&](int idx, Quad& q) {
    q.dst = nullptr;
    q.own = q.hor = q.ver = q.dia = (v4f){0.f, 0.f, 0.f, 0.f};
    q.di = q.dj = 0;
    if (idx >= kRing) return;
    int hy, qx;
    if (idx < kFull) {
      const int r = idx / QW;
      hy = r < kTop ? r : r + (SH - kTop - kBot);
      qx = idx - r * QW;
    } else {
      const int m = idx - kFull, r = m / (kLeft + kRight), k = m - r * (kLeft + kRight);
      hy = kTop + r;
      qx = k < kLeft ? k : QW - (kLeft + kRight) + k;
    }
    const int hx = qx * 4, x = ti * TW + hx, y = sj * SH + hy;
    if (!(x < W && y < H)) return;
    const int dj = hy < kTop ? -1 : (hy >= SH - kBot ? 1 : 0);
    const int di = hx < R + 4 ? -1 : (hx >= TW - R ? 1 : 0);
    const bool okj = dj != 0 && (unsigned)(sj + dj) < (unsigned)nseg, oki = di != 0 && (unsigned)(ti + di) < (unsigned)ntx;
    const float* p_own = unit_slab + (hy + RY) * WW + (hx + R);
    q.own = *reinterpret_cast<const v4f*>(p_own);
    // lane-masked loads: only the windows that cover the quad are read
    if (oki) q.hor = *reinterpret_cast<const v4f*>(p_own + (ptrdiff_t)di * slab_stride - di * TW);
    if (okj) q.ver = *reinterpret_cast<const v4f*>(p_own + dj * row_of_units - dj * SH * WW);
    if (oki && okj) q.dia = *reinterpret_cast<const v4f*>(p_own + dj * row_of_units + (ptrdiff_t)di * slab_stride - dj * SH * WW - di * TW);
    q.di = di;
    q.dj = dj;
    q.dst = gb + (size_t)y * W + x;
  };
  auto finish = [&](const Quad& q) {
    if (q.dst == nullptr) return;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const bool hfirst = q.di < 0, vfirst = q.dj < 0;
    const v4f r0a = hfirst ? q.dia : q.ver, r0b = hfirst ? q.ver : q.dia;      // the neighbouring row of units (dj != 0)
    const v4f r1a = hfirst ? q.hor : q.own, r1b = hfirst ? q.own : q.hor;      // the own row of units
    const v4f t0 = vfirst ? r0a : r1a, t1 = vfirst ? r0b : r1b, t2 = vfirst ? r1a : r0a, t3 = vfirst ? r1b : r0b;
    *reinterpret_cast<v4f*>(q.dst) = (((zero + t0) + t1) + t2) + t3;
  };
#if DVD_STRIP_COMBINE2
  for (int idx = threadIdx.x; idx < kRing; idx += 512) {
    Quad qa, qb;
    request(idx, qa);
    request(idx + 256, qb);
    finish(qa);
    finish(qb);
  }
#else
  for (int idx = threadIdx.x; idx < kRing; idx += 256) {
    Quad qa;
    request(idx, qa);
    finish(qa);
  }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------
constexpr int kSR = 8;
struct StripShape {
  int tw, th, ry, nt, bpc;
};
// 0: 96 x 32 steps, 12 rows of vertical halo, 768 threads, one block per CU (12 waves, three per SIMD, 168 VGPRs, 125 KB of
//    LDS) -- production;  3: the same with 8 rows of vertical halo (114 KB);
// 1: 64 x 16 steps, 512 threads, two blocks per CU (16 waves, four per SIMD, 128 VGPRs, 2 x 50 KB): one thread-step per barrier;
// 2: 96 x 16 steps, 384 threads, two blocks per CU -- the first strip shape, kept for the record of what uneven SIMD
//    placement costs (see BPC above).
static const StripShape kStripShapes[] = {{96, 32, 12, 768, 1}, {64, 16, 8, 512, 2}, {96, 16, 8, 384, 2}, {96, 32, 8, 768, 1}};
constexpr int kNumStripShapes = sizeof(kStripShapes) / sizeof(kStripShapes[0]);
static int g_strip_rows = 0;      // test hook: rows per unit (0 = chosen from the shape and the device's block slots)
static int g_strip_shape = 0;     // test hook: index into kStripShapes

int strip_select(int rows, int shape) {
  if (shape < 0 || shape >= kNumStripShapes) return 1;
  if (rows != 0 && (rows < 2 * kStripShapes[shape].th || rows % kStripShapes[shape].th != 0)) return 1;
  g_strip_rows = rows;
  g_strip_shape = shape;
  return 0;
}

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    }
  }
  return cus;
}

// Rows per unit: whole rounds of the device's block slots (1008 units on 256 slots at 48 x 384 x 672 with SH = 128) weighed
// against the share of rows that are ring rows (2R + 1 of every SH).
StripPlan make_strip_plan(int B, int H, int W) {
  StripPlan p;
  const StripShape& sh0 = kStripShapes[g_strip_shape];
  const int kSTW = sh0.tw, kSTH = sh0.th;
  p.shape = g_strip_shape;
  p.ntx = (W + kSTW - 1) / kSTW;
  int best_sh = 2 * kSTH;
  if (g_strip_rows >= 2 * kSTH) {
    best_sh = (g_strip_rows / kSTH) * kSTH;
  } else {
    double best = -1.0;
    const int slots = sh0.bpc * device_cus();
    for (int nseg = 1; nseg <= 64; ++nseg) {
      int sh = ((H + nseg - 1) / nseg + kSTH - 1) / kSTH * kSTH;
      if (sh < 2 * kSTH) sh = 2 * kSTH;
      const int segs = (H + sh - 1) / sh;
      const long long units = (long long)B * p.ntx * segs;
      const long long rounds = (units + slots - 1) / slots;
      const double fill = (double)units / (double)(rounds * slots);
      const double useful = (double)H / (double)(segs * (sh + 2 * sh0.ry + 1));
      const double score = fill * useful;
      if (score > best + 1e-9) {
        best = score;
        best_sh = sh;
      }
      if (sh == 2 * kSTH) break;
    }
  }
  p.SH = best_sh;
  p.nseg = (H + p.SH - 1) / p.SH;
  const int ww = kSTW + 2 * kSR + 4;
  p.slab_stride = (size_t)ww * (p.SH + 2 * sh0.ry + 1);
  p.n_units = (size_t)B * p.ntx * p.nseg;
  size_t off = p.n_units * 4 * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  p.off_count = off;
  off += p.n_units * sizeof(unsigned);
  off = (off + 255) & ~(size_t)255;
  p.off_offs = off;
  off += (size_t)B * sizeof(int2);
  off = (off + 255) & ~(size_t)255;
  p.off_slabs = off;
  off += p.n_units * p.slab_stride * sizeof(float);
  off = (off + 255) & ~(size_t)255;
  p.off_ovf = off;
  p.ovf_cap = (size_t)kSTW * p.SH * 4;          // every tap of every pixel of a unit: no list can overflow
  off += p.n_units * p.ovf_cap * sizeof(int2);
  p.total = off;
  return p;
}

// strip kernel -> unit combine -> finish (partial sums + overflow records: warp_finish_kernel of warp_loss.hip), one stream.
template <int TW, int TH, int RY, int NT, int BPC>
static int launch_strips_shape(const WarpArgs& a, const StripPlan& p, StripArgs& sa, hipStream_t stream) {
  constexpr int lds = strip_lds_bytes(TW, TH, kSR, RY, NT);
  // (whole strips, whole steps AND whole units: a unit that reaches below the image has steps without pixels, which only the
  //  instantiation with per-thread-step bounds checks skips)
  const bool full = (a.W % TW) == 0 && (a.H % TH) == 0 && (a.H % p.SH) == 0;
  auto k = full ? warp_loss_strip_kernel<TW, TH, kSR, RY, NT, BPC, true> : warp_loss_strip_kernel<TW, TH, kSR, RY, NT, BPC, false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[full]) {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set[full] = true;
  }
  const int units = (int)p.n_units;
  const int slots = DVD_STRIP_PERSISTENT ? BPC * device_cus() : units;
  hipLaunchKernelGGL(k, dim3(units < slots ? units : slots), dim3(NT), lds, stream, a, sa);
  DVD_LAUNCH_OK();
  hipLaunchKernelGGL((combine_units_kernel<TW, kSR, RY>), dim3(units), dim3(256), 0, stream, sa.slabs, sa.slab_stride,
                     (const int2*)sa.offs, a.g_d2, a.H, a.W, p.ntx, p.nseg, p.SH, sa.direct);
  DVD_LAUNCH_OK();
  return launch_warp_finish(a.partial, units, a.sums, sa.ovf.count, sa.ovf.rec, sa.ovf.cap, a.g_d2, stream);
}

int launch_strips(const WarpArgs& a, const StripPlan& p, char* ws, hipStream_t stream) {
  StripArgs sa;
  sa.slabs = reinterpret_cast<float*>(ws + p.off_slabs);
  sa.ovf.count = reinterpret_cast<unsigned*>(ws + p.off_count);
  sa.ovf.rec = reinterpret_cast<int2*>(ws + p.off_ovf);
  sa.ovf.cap = (unsigned)p.ovf_cap;
  sa.offs = reinterpret_cast<int2*>(ws + p.off_offs);
  sa.ntx = p.ntx;
  sa.nseg = p.nseg;
  sa.SH = p.SH;
  sa.direct = 1;
  sa.slab_stride = (unsigned)p.slab_stride;
  sa.n_units = (int)p.n_units;
  DVD_REQUIRE(p.n_units * p.slab_stride < (1ull << 32), "warp_loss: slab workspace too large for 32-bit unit strides");
  switch (p.shape) {
    case 0:
      return launch_strips_shape<96, 32, 12, 768, 1>(a, p, sa, stream);
    case 1:
      return launch_strips_shape<64, 16, 8, 512, 2>(a, p, sa, stream);
    case 2:
      return launch_strips_shape<96, 16, 8, 384, 2>(a, p, sa, stream);
    default:
      return launch_strips_shape<96, 32, 8, 768, 1>(a, p, sa, stream);
  }
}

}  // namespace dvd
