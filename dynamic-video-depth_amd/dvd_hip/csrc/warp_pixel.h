// Device code shared by the two generations of the fused warp+loss kernel (csrc/warp_loss.hip: tiles, rounds 2-5;
// csrc/warp_strip.hip: strips, round 6): kernel arguments, the per-pixel forward / backward (`pixel`, and `pixel2` = two
// pixels in lockstep on packed fp32 instructions), the Q31.32 LDS accumulation, the IO policies that supply the frame-2
// depth taps and take their gradient.  What the arithmetic replaces in the reference is stated at the top of
// warp_loss.hip; both translation units are built with -ffp-contract=off.
#pragma once

#include "dvd_common.h"

#ifndef DVD_WARP_PREFETCH
#define DVD_WARP_PREFETCH 0
#endif
#ifndef DVD_WARP_DIRECT_INTERIOR
#define DVD_WARP_DIRECT_INTERIOR 1
#endif
#ifndef DVD_WARP_PINHOLE
#define DVD_WARP_PINHOLE 1      // 0: A/B builds without the pinhole-intrinsics instantiation (tools/build_variant.sh)
#endif
#ifndef DVD_WARP_V5
#define DVD_WARP_V5 1           // 0: A/B builds without the two-pixel lockstep loop of round 5 (tools/build_variant.sh)
#endif
#ifndef DVD_WARP_V5_PREFETCH
#define DVD_WARP_V5_PREFETCH 0  // 1: inputs of thread-step i + 1 requested between the phases of step i (measured: 159.7 vs 156.1 us without)
#endif
#ifndef DVD_WARP_COMBINE_TILES
#define DVD_WARP_COMBINE_TILES 1
#endif
#ifndef DVD_WARP_KO_FILL           // knock-out builds (timing studies only; results are wrong): no window loads / no flush
#define DVD_WARP_KO_FILL 0
#endif
#ifndef DVD_WARP_KO_FLUSH
#define DVD_WARP_KO_FLUSH 0
#endif
#ifndef DVD_WARP_COMBINE_MASKED
#define DVD_WARP_COMBINE_MASKED 1
#endif
#ifndef DVD_WARP_V5_EARLY
#define DVD_WARP_V5_EARLY 0
#endif
#ifndef DVD_WARP_V5_SCHED
#define DVD_WARP_V5_SCHED 1     // scheduling barriers at the forward / backward boundary of the lockstep pixel pair
#endif
#ifndef DVD_WARP_CAM_VGPR
#define DVD_WARP_CAM_VGPR 0     // bit mask of the camera matrices pinned into VGPRs: 1 R1, 2 R2, 4 K, 8 R2T (round 4: 15)
#endif
#include <type_traits>


namespace dvd {


struct WarpArgs {
  const float* __restrict__ d1;
  const float* __restrict__ d2;
  const float* __restrict__ flow;
  const float* __restrict__ mask;
  const float* __restrict__ sf;
  const float* __restrict__ R1;
  const float* __restrict__ R2;
  const float* __restrict__ R2T;
  const float* __restrict__ t1;
  const float* __restrict__ t2;
  const float* __restrict__ K;
  const float* __restrict__ Ki;
  float* __restrict__ partial;
  float* sums;
  float* __restrict__ g_d1;
  float* g_d2;
  float* __restrict__ g_sf;
  int B, H, W, HW;
  int midas_mask, crit_l2, disp_mode, loss_on_sf;
  float flow_mul, disp_mul;
  float half_w, half_h, wmax, hmax;
};

struct Cam {
  float Ki[9], R1[9], R2[9], R2T[9], K[9], t1[3], t2[3];
};

__device__ __forceinline__ void load_cam(const WarpArgs& a, int b, Cam& c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    c.Ki[i] = a.Ki[b * 9 + i];
    c.R1[i] = a.R1[b * 9 + i];
    c.R2[i] = a.R2[b * 9 + i];
    c.R2T[i] = a.R2T[b * 9 + i];
    c.K[i] = a.K[b * 9 + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c.t1[i] = a.t1[b * 3 + i];
    c.t2[i] = a.t2[b * 3 + i];
  }
}

// IEEE division evaluated with the unscaled form of the hardware division sequence (rcp, one
// Newton step on the reciprocal, two fma corrections of the quotient: exactly what the compiler
// emits between v_div_scale and v_div_fixup, whose scaling is the identity for the operand
// ranges here: divisors (W-1)/2, (H-1)/2 and I.z + 1e-8 >= 1e-3, quotients far from the
// denormal range).  The reciprocal of a loop-invariant divisor is then computed once, and the
// two divisions by I.z share theirs.
__device__ __forceinline__ float rcp_refined(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_exact1(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

// (x + flow) -> normalised -> un-normalised -> border clamp: the five fp32
// roundings of backward_warp + torch's grid_sample (align_corners=True).
__device__ __forceinline__ float sample_coord(float pix, float fl, float half, float maxv) {
  float g = pix + fl;
  g = div_exact1(g, half, rcp_refined(half));   // == g / half (IEEE); the reciprocal is loop invariant
  g = g - 1.0f;
  float i = (g + 1.0f) * half;
  return fminf(maxv, fmaxf(i, 0.0f));
}

// mul + three chained FMAs in tap order nw, ne, sw, se: what ATen's
// vectorised CPU grid_sample evaluates.
__device__ __forceinline__ float bilinear(float vnw, float vne, float vsw, float vse, float wnw,
                                          float wne, float wsw, float wse) {
  float r = vnw * wnw;
  r = __builtin_fmaf(vne, wne, r);
  r = __builtin_fmaf(vsw, wsw, r);
  r = __builtin_fmaf(vse, wse, r);
  return r;
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }

// 8-byte load of two horizontally adjacent floats (4-byte aligned address).
__device__ __forceinline__ float2 load_pair(const float* p) {
  float2 r;
  __builtin_memcpy(&r, p, sizeof(float2));
  return r;
}

// `IO` supplies the frame-2 depth taps and takes the depth_2 gradient taps:
//   io.fetch(o_n, x0, y0, in_e, in_s, dnw, dne, dsw, dse)   (out-of-image taps -> 0)
//   io.scatter(o_n, x0, y0, in_e, in_s, t_nw, t_ne, t_sw, t_se)
//
// The arithmetic is split in two classes:
//   EXACT  -- everything that decides an index or a mask (tap indices, I.z < 1e-3,
//             W2.z < 100) follows the reference's fp32 rounding sequence: separate
//             multiplies/adds in torch's matmul order, IEEE division, the mul+3*fma
//             bilinear of ATen.  (The file is built with -ffp-contract=off.)
//   FAST   -- quantities only compared within a tolerance (sf_by_depth, the disparity
//             error, the whole backward) use explicit FMAs, v_rcp_f32 and the affine
//             structure of the tap rays, which cuts the VALU work per pixel by ~2x.
// SHIPPED=true folds the flag set of experiments/davis/train_sequence.sh
// (--midas --use_disp) at compile time; false reads the flags from the config.
#define DVD_FMA __builtin_fmaf
// PIN = true: the pair's intrinsics have the pinhole form without skew,
//   K^T = [fx 0 0; 0 fy 0; cx cy 1],  (K^-1)^T = [a 0 0; 0 b 0; c d 1]   (exact zeros, exact one)
// -- what generate_frame_midas.py:135-139 writes for every frame; the tile kernel tests the ten entries of the pair's
// matrices (wave-uniform) and takes this instantiation.  Products with an exact 0 are +-0 and adding them is exact, so every
// EXACT quantity below is bit-identical to the general expression; it is ~50 VALU instructions per pixel less.
template <bool GRADS, bool SHIPPED, bool PIN, class IO>
__device__ __forceinline__ void pixel(const WarpArgs& a, const Cam& c, IO& io, int y, int x,
                                      float d1, float fx, float fy, float mk, float s0, float s1,
                                      float s2, float acc[4], float& g_d1_out, float g_s_out[3]) {
  const bool midas_mask = SHIPPED ? true : (a.midas_mask != 0);
  const int disp_mode = SHIPPED ? 1 : a.disp_mode;
  const bool loss_on_sf = SHIPPED ? false : (a.loss_on_sf != 0);
  const float xf = (float)x, yf = (float)y;
  // --- EXACT: ray = (x,y,1) @ K_inv ; p1c = d1*ray ; P1 = p1c@R1 + t1
  float r0, r1, r2;
  if (PIN) {
    r0 = xf * c.Ki[0] + c.Ki[6];     // (x*a + y*0) + 1*c
    r1 = yf * c.Ki[4] + c.Ki[7];
    r2 = 1.0f;
  } else {
    rowvec_mat3(xf, yf, 1.0f, c.Ki, r0, r1, r2);
  }
  const float pc0 = d1 * r0, pc1 = d1 * r1, pc2 = PIN ? d1 : d1 * r2;
  float P0, P1, P2;
  rowvec_mat3(pc0, pc1, pc2, c.R1, P0, P1, P2);
  P0 = P0 + c.t1[0];
  P1 = P1 + c.t1[1];
  P2 = P2 + c.t1[2];

  // --- EXACT: bilinear taps of frame 2 at (x,y)+flow
  const float ix = sample_coord(xf, fx, a.half_w, a.wmax);
  const float iy = sample_coord(yf, fy, a.half_h, a.hmax);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float ww = ix - x0f, we = 1.0f - ww;
  const float wn = iy - y0f, ws = 1.0f - wn;
  const float w_nw = ws * we, w_ne = ws * ww, w_sw = wn * we, w_se = wn * ww;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const bool in_e = (x0 + 1) < a.W, in_s = (y0 + 1) < a.H;  // x0,y0 are always in range
  const int o_n = y0 * a.W + x0;
  float dnw, dne, dsw, dse;  // 0 for out-of-image taps, like ATen's masked gather
  io.fetch(o_n, x0, y0, in_e, in_s, dnw, dne, dsw, dse);
  // EXACT: z of the camera-2 points at the taps, W2.z = warped_p2_camera_2.z
  float W2z;
  if (PIN) {                                   // the four tap rays have z = (0 + 0) + 1
    W2z = bilinear(dnw, dne, dsw, dse, w_nw, w_ne, w_sw, w_se);
  } else {
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float zn0 = (x0f * c.Ki[2] + y0f * c.Ki[5]) + c.Ki[8];
    const float zn1 = (x1f * c.Ki[2] + y0f * c.Ki[5]) + c.Ki[8];
    const float zs0 = (x0f * c.Ki[2] + y1f * c.Ki[5]) + c.Ki[8];
    const float zs1 = (x1f * c.Ki[2] + y1f * c.Ki[5]) + c.Ki[8];
    W2z = bilinear(dnw * zn0, dne * zn1, dsw * zs0, dse * zs1, w_nw, w_ne, w_sw, w_se);
  }

  // --- EXACT: dynamic reprojection  Q = (P1 + s - t2) @ R2T ; I = Q @ K
  const float A0 = (P0 + s0) - c.t2[0], A1 = (P1 + s1) - c.t2[1], A2 = (P2 + s2) - c.t2[2];
  float Q0, Q1, Q2, I0, I1, I2;
  rowvec_mat3(A0, A1, A2, c.R2T, Q0, Q1, Q2);
  if (PIN) {
    I0 = Q0 * c.K[0] + Q2 * c.K[6];            // (Q0*fx + Q1*0) + Q2*cx
    I1 = Q1 * c.K[4] + Q2 * c.K[7];
    I2 = Q2;                                   // (0 + 0) + Q2*1
  } else {
    rowvec_mat3(Q0, Q1, Q2, c.K, I0, I1, I2);
  }
  const float den = I2 + 1e-8f;
  const bool behind = I2 < 1e-3f;
  const float yden = rcp_refined(den);     // IEEE-exact quotients: sign(dflow - flow) must match the reference
  const float u = behind ? xf : div_exact1(I0, den, yden);
  const float v = behind ? yf : div_exact1(I1, den, yden);
  const float ex = (u - xf) - fx, ey = (v - yf) - fy;  // dflow - flow

  // --- FAST: warped world point of frame 2, G = sum_k w_k (d2_k ray_k @ R2 + t2), via
  //     ray(x0+i, y0+j) = ray(x0,y0) + i*Ki[0,:] + j*Ki[1,:]
  const float q0 = PIN ? DVD_FMA(x0f, c.Ki[0], c.Ki[6]) : DVD_FMA(x0f, c.Ki[0], DVD_FMA(y0f, c.Ki[3], c.Ki[6]));
  const float q1 = PIN ? DVD_FMA(y0f, c.Ki[4], c.Ki[7]) : DVD_FMA(x0f, c.Ki[1], DVD_FMA(y0f, c.Ki[4], c.Ki[7]));
  const float q2 = PIN ? 1.0f : DVD_FMA(x0f, c.Ki[2], DVD_FMA(y0f, c.Ki[5], c.Ki[8]));
  const float a_nw = w_nw * dnw, a_ne = w_ne * dne, a_sw = w_sw * dsw, a_se = w_se * dse;
  const float sE = a_ne + a_se, sS = a_sw + a_se, sA = (a_nw + a_ne) + sS;
  const float V0 = PIN ? DVD_FMA(q0, sA, c.Ki[0] * sE) : DVD_FMA(q0, sA, DVD_FMA(c.Ki[0], sE, c.Ki[3] * sS));
  const float V1 = PIN ? DVD_FMA(q1, sA, c.Ki[4] * sS) : DVD_FMA(q1, sA, DVD_FMA(c.Ki[1], sE, c.Ki[4] * sS));
  const float V2 = PIN ? sA : DVD_FMA(q2, sA, DVD_FMA(c.Ki[2], sE, c.Ki[5] * sS));
  const float G0 = DVD_FMA(V0, c.R2[0], DVD_FMA(V1, c.R2[3], DVD_FMA(V2, c.R2[6], c.t2[0])));
  const float G1 = DVD_FMA(V0, c.R2[1], DVD_FMA(V1, c.R2[4], DVD_FMA(V2, c.R2[7], c.t2[1])));
  const float G2 = DVD_FMA(V0, c.R2[2], DVD_FMA(V1, c.R2[5], DVD_FMA(V2, c.R2[8], c.t2[2])));
  const float f0 = (G0 - P0) - s0, f1 = (G1 - P1) - s1, f2 = (G2 - P2) - s2;  // sf_by_depth - sf

  // --- mask (EXACT) and per-pixel errors
  float m = mk;
  if (midas_mask) {
    m = ((d1 < 100.0f) ? 1.0f : 0.0f) * m;
    m = ((W2z < 100.0f) ? 1.0f : 0.0f) * m;
  }
  const float flow_err = a.crit_l2 ? (ex * ex + ey * ey) : (fabsf(ex) + fabsf(ey));
  float disp_err, rca = 0.0f, rcb = 0.0f, ediff = 0.0f;
  if (disp_mode == 1) {
    rca = __builtin_amdgcn_rcpf(fmaxf(Q2, 1e-3f));
    rcb = __builtin_amdgcn_rcpf(fmaxf(W2z, 1e-3f));
    ediff = rca - rcb;
    disp_err = 100.0f * fabsf(ediff);
  } else if (disp_mode == 2) {
    const float ca = fmaxf(Q2, 1e-3f), cb = fmaxf(W2z, 1e-3f);
    disp_err = fmaxf(ca, cb) * __builtin_amdgcn_rcpf(fminf(ca, cb)) - 1.0f;
  } else {
    disp_err = fabsf(Q2 - W2z);
  }
  const float sf_err = fabsf(f0) + fabsf(f1) + fabsf(f2);
  acc[0] += m;
  acc[1] = DVD_FMA(m, flow_err, acc[1]);
  acc[2] = DVD_FMA(m, disp_err, acc[2]);
  acc[3] = DVD_FMA(m, sf_err, acc[3]);

  if (!GRADS) return;
  // ------------------------------ FAST: backward (un-normalised) ----------
  float gQ0 = 0.0f, gQ1 = 0.0f, gQ2 = 0.0f;
  const float fm = a.flow_mul * m;
  if (!behind && fm != 0.0f) {
    const float gu = a.crit_l2 ? fm * 2.0f * ex : fm * sgn(ex);
    const float gv = a.crit_l2 ? fm * 2.0f * ey : fm * sgn(ey);
    const float rden = __builtin_amdgcn_rcpf(den);
    const float gI0 = gu * rden, gI1 = gv * rden;
    const float gI2 = -DVD_FMA(gu, u, gv * v) * rden;
    if (PIN) {
      gQ0 = gI0 * c.K[0];
      gQ1 = gI1 * c.K[4];
      gQ2 = DVD_FMA(gI0, c.K[6], DVD_FMA(gI1, c.K[7], gI2));
    } else {
      gQ0 = DVD_FMA(gI0, c.K[0], DVD_FMA(gI1, c.K[1], gI2 * c.K[2]));
      gQ1 = DVD_FMA(gI0, c.K[3], DVD_FMA(gI1, c.K[4], gI2 * c.K[5]));
      gQ2 = DVD_FMA(gI0, c.K[6], DVD_FMA(gI1, c.K[7], gI2 * c.K[8]));
    }
  }
  // The second loss term only reaches depth_2 through W2.z / G, both linear in disp_mul:
  // keep those two in units of disp_mul (`u*`), the IO policy multiplies it back.
  float uW2z = 0.0f;                         // d loss / d W2.z        / disp_mul
  float uG0 = 0.0f, uG1 = 0.0f, uG2 = 0.0f;  // d loss / d warped point / disp_mul
  const float dm = a.disp_mul;
  if (!loss_on_sf) {
    if (disp_mode == 1 && m != 0.0f) {
      const float ue = m * 100.0f * sgn(ediff);
      if (Q2 >= 1e-3f) gQ2 = DVD_FMA(-ue * dm, rca * rca, gQ2);
      if (W2z >= 1e-3f) uW2z = ue * (rcb * rcb);
    }
  } else if (m != 0.0f) {
    uG0 = m * sgn(f0);
    uG1 = m * sgn(f1);
    uG2 = m * sgn(f2);
  }
  const float gG0 = dm * uG0, gG1 = dm * uG1, gG2 = dm * uG2;
  // scene flow enters A (+) and, in sf-loss mode, the error term (-); so does P1
  const float gA0 = DVD_FMA(gQ0, c.R2T[0], DVD_FMA(gQ1, c.R2T[1], gQ2 * c.R2T[2])) - gG0;
  const float gA1 = DVD_FMA(gQ0, c.R2T[3], DVD_FMA(gQ1, c.R2T[4], gQ2 * c.R2T[5])) - gG1;
  const float gA2 = DVD_FMA(gQ0, c.R2T[6], DVD_FMA(gQ1, c.R2T[7], gQ2 * c.R2T[8])) - gG2;
  g_s_out[0] = gA0;
  g_s_out[1] = gA1;
  g_s_out[2] = gA2;
  const float gp0 = DVD_FMA(gA0, c.R1[0], DVD_FMA(gA1, c.R1[1], gA2 * c.R1[2]));
  const float gp1 = DVD_FMA(gA0, c.R1[3], DVD_FMA(gA1, c.R1[4], gA2 * c.R1[5]));
  const float gp2 = DVD_FMA(gA0, c.R1[6], DVD_FMA(gA1, c.R1[7], gA2 * c.R1[8]));
  g_d1_out = PIN ? DVD_FMA(gp0, r0, DVD_FMA(gp1, r1, gp2)) : DVD_FMA(gp0, r0, DVD_FMA(gp1, r1, gp2 * r2));
  // depth_2 taps (units of disp_mul): d/d(d2_k) = w_k * (h . ray_k),  h = uG @ R2^T + (0,0,uW2z)
  float h0 = 0.0f, h1 = 0.0f, h2 = uW2z;
  if (loss_on_sf) {
    h0 = DVD_FMA(uG0, c.R2[0], DVD_FMA(uG1, c.R2[1], uG2 * c.R2[2]));
    h1 = DVD_FMA(uG0, c.R2[3], DVD_FMA(uG1, c.R2[4], uG2 * c.R2[5]));
    h2 += DVD_FMA(uG0, c.R2[6], DVD_FMA(uG1, c.R2[7], uG2 * c.R2[8]));
  }
  if (h0 != 0.0f || h1 != 0.0f || h2 != 0.0f) {
    if (PIN && !loss_on_sf) {                  // h = (0, 0, uW2z): every tap ray has z = 1
      io.scatter(o_n, x0, y0, in_e, in_s, w_nw * h2, w_ne * h2, w_sw * h2, w_se * h2);
    } else {
      const float hb = DVD_FMA(h0, q0, DVD_FMA(h1, q1, h2 * q2));
      const float hx = PIN ? h0 * c.Ki[0] : DVD_FMA(h0, c.Ki[0], DVD_FMA(h1, c.Ki[1], h2 * c.Ki[2]));
      const float hy = PIN ? h1 * c.Ki[4] : DVD_FMA(h0, c.Ki[3], DVD_FMA(h1, c.Ki[4], h2 * c.Ki[5]));
      io.scatter(o_n, x0, y0, in_e, in_s, w_nw * hb, w_ne * (hb + hx), w_sw * (hb + hy), w_se * ((hb + hx) + hy));
    }
  }
}

// ---- IO policy 1: straight to global memory (gather via L1/L2, hardware fp32 atomics).
struct DirectIO {
  const float* d2b;
  float* gb;
  int W;
  float unit;
  __device__ __forceinline__ void fetch(int o_n, int, int, bool in_e, bool in_s, float& dnw, float& dne,
                                        float& dsw, float& dse) const {
    const int o_s = o_n + W;
    if (in_e) {
      const float2 pn = load_pair(d2b + o_n);
      dnw = pn.x;
      dne = pn.y;
      if (in_s) {
        const float2 ps = load_pair(d2b + o_s);
        dsw = ps.x;
        dse = ps.y;
      } else {
        dsw = 0.0f;
        dse = 0.0f;
      }
    } else {
      dnw = d2b[o_n];
      dne = 0.0f;
      dsw = in_s ? d2b[o_s] : 0.0f;
      dse = 0.0f;
    }
  }
  __device__ __forceinline__ void scatter(int o_n, int, int, bool in_e, bool in_s, float tnw, float tne,
                                          float tsw, float tse) const {
    unsafeAtomicAdd(gb + o_n, tnw * unit);
    if (in_e) unsafeAtomicAdd(gb + o_n + 1, tne * unit);
    if (in_s) unsafeAtomicAdd(gb + o_n + W, tsw * unit);
    if (in_e && in_s) unsafeAtomicAdd(gb + o_n + W + 1, tse * unit);
  }
};

template <int PX, bool GRADS>
__global__ __launch_bounds__(256) void warp_loss_kernel(const WarpArgs a) {
  const int b = blockIdx.y;
  Cam c;
  load_cam(a, b, c);
  const int p0 = (blockIdx.x * 256 + threadIdx.x) * PX;  // first pixel of this thread in the pair
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (p0 < a.HW) {
    const size_t base = (size_t)b * a.HW + p0;
    float d1[PX], mk[PX], fl[2 * PX], s0[PX], s1[PX], s2[PX];
    const float* sfb = a.sf + (size_t)b * 3 * a.HW + p0;
    if (PX == 4) {
      *reinterpret_cast<float4*>(d1) = *reinterpret_cast<const float4*>(a.d1 + base);
      *reinterpret_cast<float4*>(mk) = *reinterpret_cast<const float4*>(a.mask + base);
      *reinterpret_cast<float4*>(fl) = *reinterpret_cast<const float4*>(a.flow + 2 * base);
      *reinterpret_cast<float4*>(fl + 4) = *reinterpret_cast<const float4*>(a.flow + 2 * base + 4);
      *reinterpret_cast<float4*>(s0) = *reinterpret_cast<const float4*>(sfb);
      *reinterpret_cast<float4*>(s1) = *reinterpret_cast<const float4*>(sfb + a.HW);
      *reinterpret_cast<float4*>(s2) = *reinterpret_cast<const float4*>(sfb + 2 * a.HW);
    } else {
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        d1[i] = a.d1[base + i];
        mk[i] = a.mask[base + i];
        fl[2 * i] = a.flow[2 * (base + i)];
        fl[2 * i + 1] = a.flow[2 * (base + i) + 1];
        s0[i] = sfb[i];
        s1[i] = sfb[a.HW + i];
        s2[i] = sfb[2 * a.HW + i];
      }
    }
    const int y = p0 / a.W;
    const int x = p0 - y * a.W;  // PX divides W, so the PX pixels share the row
    float gd1[PX], gs[PX][3];
    DirectIO io{a.d2 + (size_t)b * a.HW, a.g_d2 + (size_t)b * a.HW, a.W, a.disp_mul};
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      gd1[i] = 0.0f;
      gs[i][0] = gs[i][1] = gs[i][2] = 0.0f;
      pixel<GRADS, false, false>(a, c, io, y, x + i, d1[i], fl[2 * i], fl[2 * i + 1], mk[i], s0[i], s1[i], s2[i], acc,
                          gd1[i], gs[i]);
    }
    if (GRADS) {
      float* gsb = a.g_sf + (size_t)b * 3 * a.HW + p0;
      if (PX == 4) {
        *reinterpret_cast<float4*>(a.g_d1 + base) = make_float4(gd1[0], gd1[1], gd1[2], gd1[3]);
        *reinterpret_cast<float4*>(gsb) = make_float4(gs[0][0], gs[1][0], gs[2][0], gs[3][0]);
        *reinterpret_cast<float4*>(gsb + a.HW) = make_float4(gs[0][1], gs[1][1], gs[2][1], gs[3][1]);
        *reinterpret_cast<float4*>(gsb + 2 * a.HW) = make_float4(gs[0][2], gs[1][2], gs[2][2], gs[3][2]);
      } else {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
          a.g_d1[base + i] = gd1[i];
          gsb[i] = gs[i][0];
          gsb[a.HW + i] = gs[i][1];
          gsb[2 * a.HW + i] = gs[i][2];
        }
      }
    }
  }
  // block reduction of the four sums -> one partial record per block
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    a.partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;
  }
}

// ---------------------------------------------------------------------------
// Tiled variant (the production path).
//
// Global fp32 atomics run at only ~9e10 lane-ops/s on MI355X (measured: the
// direct kernel above spends 1.7 of its 1.9 ms in them at 48x384x672), and the
// per-lane gathers of depth_2 are TA-bound.  So a block owns a TW x TH tile of
// one pair and keeps two LDS windows that extend the tile by R pixels:
//   win  : depth_2 values, filled with coalesced 16-byte loads; the bilinear
//          taps are read from it (ds_read2_b32);
//   accw : depth_2-gradient accumulator, Q31.32 fixed point (ds_add_u64; see kFixScale below).
// At the end the accumulator window is stored, coalesced, to the tile's slab
// in the workspace; `combine_slabs_kernel` then sums, in a fixed order, the
// <= 4 slabs that cover each pixel and writes g_depth_2 once with plain
// stores: no global atomics, no memset of g_depth_2.  Taps that fall outside
// the window (|flow| > R) stay correct: they are gathered from global memory
// and their gradient goes to an overflow list applied after the combine.
// Blocks are numbered so that each XCD receives a contiguous run of tiles
// (neighbouring tiles share depth_2 halo lines in that XCD's L2).

// Window-overflow records (taps that no LDS window of their tile takes): one list PER TILE since round 5.  The tile's block
// counts its records in LDS (wave-aggregated ds_add with return) and stores the count when it is done; the finish kernel walks
// the lists.  No global counter is touched while the tiles run -- rounds 2-4 appended to 256 shared lists with returned global
// atomics (one list: ~88 appends per microsecond, 8 ms for a flow field that leaves the windows; 256 lists: tens of
// microseconds) and needed a launch in front of the tile kernel to zero the counters.
struct Overflow {
  unsigned* count;   // [tiles of the launch sequence]: written by each tile's block
  int2* rec;         // [tiles][cap]
  unsigned cap;      // records per list = every tap of every pixel of a tile: no list can overflow
};

// LDS accumulation is 64-bit fixed point (Q31.32): ds_add_u64 sustains ~6.5 lane-ops/clk/CU
// on gfx950 while ds_add_f32 manages 0.38 (tools/ubench/lds_atomics.hip), and integer adds
// commute, so g_depth_2 is bitwise reproducible.  Taps are accumulated in units of the loss
// multiplier (see `unit` in pixel()), which keeps the magnitudes O(100/depth^2) whatever the
// multipliers are; |value| >= 2^30 goes to the overflow list as a float.
constexpr float kFixScale = 4294967296.0f;          // 2^32
constexpr float kFixInv = 1.0f / 4294967296.0f;
constexpr float kFixMax = 1073741824.0f;            // 2^30

// Q31.32 from a float with |v| < 2^30: hi = floor(v), lo = (v - floor(v)) * 2^32 (both exact).
__device__ __forceinline__ unsigned long long to_fixed(float v) {
  const float fl = floorf(v);
  const unsigned lo = (unsigned)((v - fl) * kFixScale);
  const unsigned hi = (unsigned)(int)fl;
  return ((unsigned long long)hi << 32) | lo;
}

template <int WW, int WH>
struct TileIO {
  const float* d2b;          // depth_2 of this pair
  float* win;                // LDS [WH][WW]
  unsigned long long* accw;  // LDS [WH][WW], Q31.32
  int W, wx0, wy0, pair_base, list;
  float unit;                // accumulated values are multiplied by this at the end
  Overflow ovf;
  unsigned* lcount;          // LDS: records of this tile so far
  static constexpr int kWW = WW;
  __device__ __forceinline__ bool inside(int x0, int y0) const {
    const int lx = x0 - wx0, ly = y0 - wy0;
    return (lx >= 0) && (lx + 1 < WW) && (ly >= 0) && (ly + 1 < WH);
  }
  // both pixels of a lockstep pair: all four taps of each inside the window?  cell = index of the north-west tap
  __device__ __forceinline__ bool cells2(int x0A, int y0A, int x0B, int y0B, int& cellA, int& cellB) const {
    const int lxA = x0A - wx0, lyA = y0A - wy0, lxB = x0B - wx0, lyB = y0B - wy0;
    cellA = lyA * WW + lxA;
    cellB = lyB * WW + lxB;
    return ((unsigned)lxA < (unsigned)(WW - 1)) & ((unsigned)lyA < (unsigned)(WH - 1)) &
           ((unsigned)lxB < (unsigned)(WW - 1)) & ((unsigned)lyB < (unsigned)(WH - 1));
  }
  __device__ __forceinline__ void fetch(int o_n, int x0, int y0, bool in_e, bool in_s, float& dnw, float& dne,
                                        float& dsw, float& dse) const {
    if (inside(x0, y0)) {
      // window cells outside the image hold 0, exactly what ATen's masked gather returns
      const float* p = win + (y0 - wy0) * WW + (x0 - wx0);
      dnw = p[0];
      dne = p[1];
      dsw = p[WW];
      dse = p[WW + 1];
    } else {
      DirectIO g{d2b, nullptr, W, 1.0f};
      g.fetch(o_n, x0, y0, in_e, in_s, dnw, dne, dsw, dse);
      // consumed here: the wait for these rare gathers must not land at the merge with the LDS path, where it is
      // s_waitcnt vmcnt(0) for EVERY pixel and also waits for the previous step's stores (round 5)
      asm volatile("" : "+v"(dnw), "+v"(dne), "+v"(dsw), "+v"(dse));
    }
  }
  // (index, value) record for a tap the window cannot take; applied with global atomics after the slab combine.
  // The lanes of the wave that are here together reserve their slots with ONE atomic on the shared counter
  // (a flow field that leaves the windows used to serialise 30 M returned atomics on that counter: 8 ms).
  __device__ __forceinline__ void spill(int idx, float v) const {
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(lcount, (unsigned)__popcll(m));     // ds_add_rtn_u32
    base = __shfl(base, leader, 64);
    const unsigned i = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (i < ovf.cap) ovf.rec[(size_t)list * ovf.cap + i] = make_int2(pair_base + idx, __float_as_int(v * unit));
  }
  __device__ __forceinline__ void add_fixed(unsigned long long* p, int idx, float v) const {
    if (fabsf(v) < kFixMax)
      atomicAdd(p, to_fixed(v));  // ds_add_u64
    else
      spill(idx, v);
  }
  __device__ __forceinline__ void scatter(int o_n, int x0, int y0, bool in_e, bool in_s, float tnw, float tne,
                                          float tsw, float tse) const {
    if (inside(x0, y0)) {
      unsigned long long* p = accw + (y0 - wy0) * WW + (x0 - wx0);
      // A tap beyond the image's right / bottom edge has weight exactly 0 (the sampling coordinate is clamped to W-1 / H-1,
      // so its fractional part is 0) and its window cell exists (`inside`) and is never read by the combine: all four adds
      // are unconditional, and ONE magnitude test per pixel guards the fixed-point range (round 2: a branch per tap).
      // (a SUM of magnitudes, not a max: fmaxf drops NaNs, and a NaN tap must reach the spill list -- and g_depth_2 -- instead
      //  of being converted to 0; the sum is >= the largest magnitude, so the test is only stricter; same instruction count)
      const float big = (fabsf(tnw) + fabsf(tne)) + (fabsf(tsw) + fabsf(tse));
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

// ---------------------------------------------------------------------------
// Round 5: TWO horizontally adjacent pixels of a thread-step evaluated in lockstep, the lanes of a float2 holding the
// two pixels, so that every multiply / add / fma of pixel() issues ONCE as v_pk_{mul,add,fma}_f32 for both (gfx950's
// vector fp32 peak is the packed rate; the one-pixel loop ran 369 VALU instructions per pixel and was issue bound).
// Shipped flag set (--midas --use_disp) and pinhole intrinsics only -- every other case stays on pixel().
// Each lane of a packed instruction rounds like the scalar instruction, so the EXACT class is unchanged: the operation
// sequence below is pixel<GRADS, true, true>'s, statement by statement (masks and tap indices stay bit-identical; the
// parity tests run it against the oracle and against the one-pixel variants).  What differs, FAST class only:
//   * the sign of a residual is med3(e * 2^126, -1, 1) (two instructions instead of four; exact for |e| >= 2^-126);
//   * branches that only skipped work for masked / behind-camera pixels are arithmetic (a zeroed reciprocal / factor):
//     with finite inputs a masked pixel's gradients are products with an exact 0;
//   * the four sums are accumulated per lane and the lanes added at the end.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f fma2(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, (v2f){b, b}, c); }
__device__ __forceinline__ v2f fma2(v2f a, float b, float c) { return __builtin_elementwise_fma(a, (v2f){b, b}, (v2f){c, c}); }
__device__ __forceinline__ v2f abs2(v2f a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ v2f rcp2(v2f a) { return (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f sgn2(v2f e) {
  const v2f t = e * 0x1p126f;
  return (v2f){__builtin_amdgcn_fmed3f(t.x, -1.0f, 1.0f), __builtin_amdgcn_fmed3f(t.y, -1.0f, 1.0f)};
}
// div_exact1 / rowvec_mat3 / sample_coord on two pixels: the same operations in the same order, per lane
__device__ __forceinline__ v2f div_exact2(v2f a, v2f b, v2f y) {
  v2f q = a * y;
  v2f r = fma2(-b, q, a);
  q = fma2(r, y, q);
  r = fma2(-b, q, a);
  return fma2(r, y, q);
}
__device__ __forceinline__ void rowvec_mat3_2(v2f v0, v2f v1, v2f v2, const float* __restrict__ M, v2f& o0, v2f& o1, v2f& o2) {
  o0 = (v0 * M[0] + v1 * M[3]) + v2 * M[6];
  o1 = (v0 * M[1] + v1 * M[4]) + v2 * M[7];
  o2 = (v0 * M[2] + v1 * M[5]) + v2 * M[8];
}
__device__ __forceinline__ v2f sample_coord2(v2f pix, v2f fl, float half, float yhalf, float maxv) {
  v2f g = pix + fl;
  g = div_exact2(g, (v2f){half, half}, (v2f){yhalf, yhalf});
  g = g - 1.0f;
  const v2f i = (g + 1.0f) * half;
  return (v2f){fminf(maxv, fmaxf(i.x, 0.0f)), fminf(maxv, fmaxf(i.y, 0.0f))};
}
// Q31.32 from a float with |v| < 2^30 in four instructions: v_fract (v - floor(v), kept below 1), * 2^32, v_cvt_u32,
// v_cvt_flr_i32 (floor and convert in one).  Differs from to_fixed() only for -2^-24 < v < 0, where v_fract's clamp
// gives 1 - 2^-24 instead of 1: an absolute error of 6e-8 * 2^-24 in a gradient accumulator -- FAST class.
__device__ __forceinline__ unsigned long long to_fixed_fast(float v, float v_scaled_fract) {
  int hi;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi) : "v"(v));
  const unsigned lo = (unsigned)v_scaled_fract;
  return ((unsigned long long)(unsigned)hi << 32) | lo;
}
// accumulator cell -> float: float(hi) + float(lo) * 2^-32 in one fma (three instructions; the compiler's signed 64-bit
// conversion is a 13-instruction sequence with a 64-bit shift, and a tile converts 1.85 cells per pixel)
__device__ __forceinline__ float from_fixed(long long c) {
  return __builtin_fmaf((float)(unsigned)c, kFixInv, (float)(int)(c >> 32));
}

// The lockstep loop keeps the pair's camera in LDS (32 floats behind the windows, written once per block) and reads it
// as 16-byte BROADCASTS right where a phase needs it: a packed instruction takes a scalar from either half of a VGPR pair
// through op_sel, but a scalar REGISTER operand occupies an aligned SGPR pair of its own (the instruction selector builds
// {s, undef}), so 32 camera scalars in SGPRs cost 64 registers and spilled into vector lanes (217 spilled SGPRs, a
// v_readlane pair + s_nop in front of most packed instructions); pinned in VGPRs they cost 32 registers for the whole
// pixel pair.  Layout (float4 index): 0-2 columns of R1 | t1;  3 t2 | K[0];  4-6 columns of R2T | K[4], K[6], K[7];
// 7 Ki[0], Ki[4], Ki[6], Ki[7].
constexpr int kCamLdsFloats = 32;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile v4f* lds_quad_ptr;     // explicitly LDS: ds_read_b128, not a flat load
__device__ __forceinline__ v4f ldq(const float* cam, int i) {
  return ((lds_quad_ptr)cam)[i];                    // volatile: a read per phase, not one hoisted set
}
// The same eight quads held in VECTOR registers for a whole unit of the strip kernel (round 6: its 768-thread blocks have 168
// registers per lane, 40 more than the tile kernel's budget): no camera read, and no lgkmcnt wait for one, inside a thread-step
// (the LDS form issues 15 ds_read_b128 per thread-step).  Vector, not scalar, registers for the reason given above.
struct CamRegs {
  v4f q[8];
};
__device__ __forceinline__ v4f ldq(const CamRegs& cam, int i) { return cam.q[i]; }
__device__ __forceinline__ CamRegs cam_regs_from_lds(const float* cam) {
  CamRegs r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.q[i] = ((lds_quad_ptr)cam)[i];
    asm volatile("" : "+v"(r.q[i]));                // opaque: stays in VGPRs, is not re-derived as a scalar
  }
  return r;
}
__device__ __forceinline__ float cam_lds_value(const Cam& c, int i) {
  const int q = i >> 2, k = i & 3;
  if (q < 3) return k < 3 ? c.R1[3 * k + q] : c.t1[q];
  if (q == 3) return k < 3 ? c.t2[k] : c.K[0];
  if (q < 7) return k < 3 ? c.R2T[3 * k + (q - 4)] : (q == 4 ? c.K[4] : (q == 5 ? c.K[6] : c.K[7]));
  return k == 0 ? c.Ki[0] : (k == 1 ? c.Ki[4] : (k == 2 ? c.Ki[6] : c.Ki[7]));
}

// `IO`: TileIO (a tile's window, warp_loss.hip) or RingIO (a strip's ring of window rows, warp_strip.hip) -- kWW = cells per
// window row, win / accw, cells2() = are both pixels' tap quads on chip, and where; fetch / scatter for the taps that are not.
template <bool GRADS, bool CRIT_L2, class CAMT, class IO, class MID, class ST>
__device__ __forceinline__ void pixel2(const WarpArgs& a, const CAMT& cam, const IO& io, int y, int x, v2f d1,
                                       v2f fx, v2f fy, v2f mk, v2f s0, v2f s1, v2f s2, float yhw, float yhh, float acc[4],
                                       MID&& between_phases, ST&& store) {
  const float xf = (float)x, yf = (float)y;
  const v2f X = {xf, xf + 1.0f};
  // --- EXACT: ray, p1c, P1
  const v4f ki = ldq(cam, 7);                  // Ki[0], Ki[4], Ki[6], Ki[7]
  const v2f r0 = X * ki.x + ki.z;
  const float r1 = yf * ki.y + ki.w;
  const v2f pc0 = d1 * r0, pc1 = d1 * r1;
  v2f P0, P1, P2;
  {
    const v4f c0 = ldq(cam, 0), c1 = ldq(cam, 1), c2 = ldq(cam, 2);   // columns of R1 | t1
    P0 = ((pc0 * c0.x + pc1 * c0.y) + d1 * c0.z) + c0.w;
    P1 = ((pc0 * c1.x + pc1 * c1.y) + d1 * c1.z) + c1.w;
    P2 = ((pc0 * c2.x + pc1 * c2.y) + d1 * c2.z) + c2.w;
  }
  const v2f mk1 = {d1.x < 100.0f ? mk.x : 0.0f, d1.y < 100.0f ? mk.y : 0.0f};   // [depth_1 < 100] * mask_2
  // --- EXACT: bilinear taps of frame 2 at (x,y)+flow
  const v2f ix = sample_coord2(X, fx, a.half_w, yhw, a.wmax);
  const v2f iy = sample_coord2((v2f){yf, yf}, fy, a.half_h, yhh, a.hmax);
  const v2f x0f = {floorf(ix.x), floorf(ix.y)}, y0f = {floorf(iy.x), floorf(iy.y)};
  const v2f ww = ix - x0f, we = 1.0f - ww;
  const v2f wn = iy - y0f, ws = 1.0f - wn;
  const v2f w_nw = ws * we, w_ne = ws * ww, w_sw = wn * we, w_se = wn * ww;
  const int x0A = (int)x0f.x, x0B = (int)x0f.y, y0A = (int)y0f.x, y0B = (int)y0f.y;
  constexpr int WW = IO::kWW;
  int cellA, cellB;
  const bool inside = io.cells2(x0A, y0A, x0B, y0B, cellA, cellB);
  v2f dnw, dne, dsw, dse;
  if (inside) {
    const float* pA = io.win + cellA;
    const float* pB = io.win + cellB;
    dnw = (v2f){pA[0], pB[0]};
    dne = (v2f){pA[1], pB[1]};
    dsw = (v2f){pA[WW], pB[WW]};
    dse = (v2f){pA[WW + 1], pB[WW + 1]};
  } else {
#ifdef DVD_WARP_NO_SLOW
    dnw = dne = dsw = dse = (v2f){0.f, 0.f};
    return;
#endif
    float t0, t1, t2, t3, u0, u1, u2, u3;
    io.fetch(y0A * a.W + x0A, x0A, y0A, (x0A + 1) < a.W, (y0A + 1) < a.H, t0, t1, t2, t3);
    io.fetch(y0B * a.W + x0B, x0B, y0B, (x0B + 1) < a.W, (y0B + 1) < a.H, u0, u1, u2, u3);
    // the gathers are consumed HERE (an empty asm that uses the registers): otherwise the wait for them lands at the merge
    // with the LDS path as s_waitcnt vmcnt(0) -- for every pixel, and for everything else in flight (the previous
    // step's stores, the next step's inputs)
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    dnw = (v2f){t0, u0};
    dne = (v2f){t1, u1};
    dsw = (v2f){t2, u2};
    dse = (v2f){t3, u3};
  }
  // EXACT: W2.z (the four tap rays have z = 1): ATen's mul + 3 fma in tap order
  v2f W2z = dnw * w_nw;
  W2z = fma2(dne, w_ne, W2z);
  W2z = fma2(dsw, w_sw, W2z);
  W2z = fma2(dse, w_se, W2z);
  // --- EXACT: Q = (P1 + s - t2) @ R2T ; I = Q @ K
  const v4f tk = ldq(cam, 3);                  // t2[0], t2[1], t2[2], K[0]
  const v4f m0 = ldq(cam, 4), m1 = ldq(cam, 5), m2 = ldq(cam, 6);     // columns of R2T | K[4], K[6], K[7]
  const v2f Ps0 = P0 + s0, Ps1 = P1 + s1, Ps2 = P2 + s2;
  const v2f A0 = Ps0 - tk.x, A1 = Ps1 - tk.y, A2 = Ps2 - tk.z;
  const v2f Q0 = (A0 * m0.x + A1 * m0.y) + A2 * m0.z;
  const v2f Q1 = (A0 * m1.x + A1 * m1.y) + A2 * m1.z;
  const v2f Q2 = (A0 * m2.x + A1 * m2.y) + A2 * m2.z;
  const v2f I0 = Q0 * tk.w + Q2 * m1.w;
  const v2f I1 = Q1 * m0.w + Q2 * m2.w;
  const v2f den = Q2 + 1e-8f;
  const bool behindA = Q2.x < 1e-3f, behindB = Q2.y < 1e-3f;
  const v2f y0 = rcp2(den);
  const v2f yden = fma2(fma2(-den, y0, (v2f){1.0f, 1.0f}), y0, y0);
  v2f u = div_exact2(I0, den, yden), v = div_exact2(I1, den, yden);
  u.x = behindA ? xf : u.x;
  u.y = behindB ? X.y : u.y;
  v.x = behindA ? yf : v.x;
  v.y = behindB ? yf : v.y;
  const v2f ex = (u - X) - fx, ey = (v - yf) - fy;

  // --- FAST: warped world point of frame 2 (affine structure of the tap rays)
  const v2f q0 = fma2(x0f, ki.x, ki.z), q1 = fma2(y0f, ki.y, ki.w);
  const v2f a_nw = w_nw * dnw, a_ne = w_ne * dne, a_sw = w_sw * dsw, a_se = w_se * dse;
  const v2f sE = a_ne + a_se, sS = a_sw + a_se, sA = (a_nw + a_ne) + sS;
  const v2f V0 = fma2(q0, sA, ki.x * sE), V1 = fma2(q1, sA, ki.y * sS);
  // (R_2 read through its stored transpose R_2_T -- the block checked that the two ARE transposes of each other, bit for
  //  bit, as the data files write them (generate_sequence_midas.py:69-72): nine scalar registers less)
  const v2f G0 = fma2(V0, m0.x, fma2(V1, m1.x, fma2(sA, m2.x, tk.x)));
  const v2f G1 = fma2(V0, m0.y, fma2(V1, m1.y, fma2(sA, m2.y, tk.y)));
  const v2f G2 = fma2(V0, m0.z, fma2(V1, m1.z, fma2(sA, m2.z, tk.z)));
  const v2f f0 = G0 - Ps0, f1 = G1 - Ps1, f2 = G2 - Ps2;          // sf_by_depth - sf (FAST: one rounding regrouped)
  // --- mask (EXACT comparisons) and per-pixel errors
  const v2f m = {W2z.x < 100.0f ? mk1.x : 0.0f, W2z.y < 100.0f ? mk1.y : 0.0f};
  const v2f flow_err = CRIT_L2 ? (ex * ex + ey * ey) : (abs2(ex) + abs2(ey));
  const v2f rca = rcp2((v2f){fmaxf(Q2.x, 1e-3f), fmaxf(Q2.y, 1e-3f)});
  const v2f rcb = rcp2((v2f){fmaxf(W2z.x, 1e-3f), fmaxf(W2z.y, 1e-3f)});
  const v2f ediff = rca - rcb;
  const v2f disp_err = 100.0f * abs2(ediff);
  const v2f sf_err = (abs2(f0) + abs2(f1)) + abs2(f2);
  // (one accumulator per sum, the two pixels added first: four registers held across the pair instead of eight)
  {
    const v2f e1 = m * flow_err, e2 = m * disp_err, e3 = m * sf_err;
    acc[0] += m.x + m.y;
    acc[1] += e1.x + e1.y;
    acc[2] += e2.x + e2.y;
    acc[3] += e3.x + e3.y;
  }
  if (!GRADS) {
    between_phases();
    return;
  }
  // ------------------------------ FAST: backward (un-normalised) ----------
  const v2f fm = a.flow_mul * m;
  const v2f gu = CRIT_L2 ? fm * 2.0f * ex : fm * sgn2(ex);
  const v2f gv = CRIT_L2 ? fm * 2.0f * ey : fm * sgn2(ey);
  const v2f guv = fma2(gu, u, gv * v);           // (the backward needs u, v only through this dot product)
  const v2f ue = (m * 100.0f) * sgn2(ediff);
  const v2f rca2 = {Q2.x >= 1e-3f ? rca.x * rca.x : 0.0f, Q2.y >= 1e-3f ? rca.y * rca.y : 0.0f};
  const v2f h2 = {W2z.x >= 1e-3f ? ue.x * (rcb.x * rcb.x) : 0.0f, W2z.y >= 1e-3f ? ue.y * (rcb.y * rcb.y) : 0.0f};
  // ---- phase boundary: what the backward needs is (gu, gv, u, v, 1/den, ue, rca2, h2, the four weights, the cells, the
  //      ray); the scheduler may not mix the phases (it would hold both phases' values at once and spill), and the
  //      caller requests the NEXT thread-step's inputs here, so that they fly under the backward and the scatter
#if DVD_WARP_V5_SCHED
  __builtin_amdgcn_sched_barrier(0);
#endif
  between_phases();
#if DVD_WARP_V5_SCHED
  __builtin_amdgcn_sched_barrier(0);
#endif
  // behind the camera: the projection was replaced by the pixel's own coordinates, no gradient (a zero reciprocal:
  // u, v are finite there)
  const v2f rden = {behindA ? 0.0f : y0.x, behindB ? 0.0f : y0.y};
  const v2f gI0 = gu * rden, gI1 = gv * rden;
  const v2f gI2 = -guv * rden;
  // (the camera is read again for the backward: a second set of LDS broadcasts costs less than 32 registers held
  //  across the whole pixel pair)
  const v4f bk = ldq(cam, 3), b0 = ldq(cam, 4), b1 = ldq(cam, 5), b2 = ldq(cam, 6);
  const v2f gQ0 = gI0 * bk.w, gQ1 = gI1 * b0.w;
  v2f gQ2 = fma2(gI0, b1.w, fma2(gI1, b2.w, gI2));
  // disparity term: 100 |1/max(Q.z,1e-3) - 1/max(W2.z,1e-3)|; W2.z reaches depth_2 (units of disp_mul, like pixel())
  gQ2 = fma2((-ue) * a.disp_mul, rca2, gQ2);
  const v2f gA0 = fma2(gQ0, b0.x, fma2(gQ1, b1.x, gQ2 * b2.x));
  const v2f gA1 = fma2(gQ0, b0.y, fma2(gQ1, b1.y, gQ2 * b2.y));
  const v2f gA2 = fma2(gQ0, b0.z, fma2(gQ1, b1.z, gQ2 * b2.z));
  const v4f e0 = ldq(cam, 0), e1 = ldq(cam, 1), e2 = ldq(cam, 2);
  const v2f gp0 = fma2(gA0, e0.x, fma2(gA1, e1.x, gA2 * e2.x));
  const v2f gp1 = fma2(gA0, e0.y, fma2(gA1, e1.y, gA2 * e2.y));
  const v2f gp2 = fma2(gA0, e0.z, fma2(gA1, e1.z, gA2 * e2.z));
  const v2f g_d1 = fma2(gp0, r0, fma2(gp1, r1, gp2));
  auto store_grads = [&]() { store(g_d1, gA0, gA1, gA2); };
  // the pair's gradients leave before the scatter starts (the scatter then holds the weights, h2 and the two cells only)
  store_grads();
  // depth_2 taps: d/d(d2_k) = w_k * h2 (every tap ray has z = 1).  One tap PAIR at a time -- multiply, v_fract, * 2^32,
  // two conversions, two ds_add_u64 -- with scheduling barriers in between: evaluated all at once (what the scheduler does
  // for the instruction-level parallelism) the eight taps hold 40 registers and the pixel pair no longer fits 128.
  // Fixed-point range: the bilinear weights are >= 0 and sum to 1, so sum_k |w_k h2| = |h2| (a SUM-like test on h2 itself;
  // NaN fails it and takes the per-tap path, where it reaches the spill list and g_depth_2).
  if (h2.x != 0.0f || h2.y != 0.0f) {
    if (inside && (fabsf(h2.x) + fabsf(h2.y)) < kFixMax) {
      unsigned long long* pA = io.accw + cellA;
      unsigned long long* pB = io.accw + cellB;
      auto tap = [&](v2f w, int off) {
        const v2f t = w * h2;
        const v2f k = (v2f){__builtin_amdgcn_fractf(t.x), __builtin_amdgcn_fractf(t.y)} * kFixScale;
        atomicAdd(pA + off, to_fixed_fast(t.x, k.x));
        atomicAdd(pB + off, to_fixed_fast(t.y, k.y));
#if DVD_WARP_V5_SCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
      };
      tap(w_nw, 0);
      tap(w_ne, 1);
      tap(w_sw, WW);
      tap(w_se, WW + 1);
    } else {
#ifndef DVD_WARP_NO_SLOW
      const v2f t_nw = w_nw * h2, t_ne = w_ne * h2, t_sw = w_sw * h2, t_se = w_se * h2;
      if (h2.x != 0.0f)
        io.scatter(y0A * a.W + x0A, x0A, y0A, (x0A + 1) < a.W, (y0A + 1) < a.H, t_nw.x, t_ne.x, t_sw.x, t_se.x);
      if (h2.y != 0.0f)
        io.scatter(y0B * a.W + x0B, x0B, y0B, (x0B + 1) < a.W, (y0B + 1) < a.H, t_nw.y, t_ne.y, t_sw.y, t_se.y);
#endif
    }
  }
}

__device__ __forceinline__ int xcd_contiguous_block(int bid, int nb) {
  // dispatcher places block b on XCD b % 8 (speed only, never correctness)
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// The window offset of a pair: its mean flow, sampled on an 8 x 8 grid, rounded (x to a multiple of 4 so that window rows stay
// 16-byte aligned).  A coherent motion of tens of pixels (camera pan, the frame gaps 2-4 of the shipped schedule) then lands
// inside the LDS windows instead of on the overflow path; mean flows below 4 px keep the unshifted window.  Any offset is
// correct: it only moves where the on-chip window sits.  Called by all 64 lanes of a wave; the result is wave uniform.
__device__ __forceinline__ int2 pair_window_offset(const float* __restrict__ flow, int b, int H, int W) {
  const int lane = threadIdx.x & 63;
  const int gy = lane >> 3, gx = lane & 7;
  const int y = (int)(((2 * gy + 1) * (long long)H) / 16), x = (int)(((2 * gx + 1) * (long long)W) / 16);
  const float2 f = load_pair(flow + 2 * ((size_t)b * H * W + (size_t)y * W + x));
  // butterfly sum over the wave (every lane ends with the total, in the same order).  The lane index is made opaque: the
  // permute addresses would otherwise be shared with the wave_sum of the block sums at the END of the kernel, i.e. six
  // registers alive (spilled and reloaded) across the whole tile.
  int lid = lane;
  asm volatile("" : "+v"(lid));
  float sx = f.x, sy = f.y;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int src = (lid ^ o) << 2;
    sx += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, sx)));
    sy += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, sy)));
  }
  const float mx = sx * (1.0f / 64.0f), my = sy * (1.0f / 64.0f);
  int ox = 0, oy = 0;
  if (fabsf(mx) >= 4.0f || fabsf(my) >= 4.0f) {
    const float cx = fminf(fmaxf(mx, -(float)W), (float)W), cy = fminf(fmaxf(my, -(float)H), (float)H);   // (NaN -> bound)
    ox = ((int)rintf(cx * 0.25f)) * 4;
    oy = (int)rintf(cy);
  }
  return make_int2(__builtin_amdgcn_readfirstlane(ox), __builtin_amdgcn_readfirstlane(oy));
}


// ---- the strip generation (csrc/warp_strip.hip), called from dvd::run in csrc/warp_loss.hip
struct StripPlan {
  int shape, ntx, nseg, SH;
  size_t n_units, off_count, off_offs, off_slabs, slab_stride, off_ovf, ovf_cap, total;
};
StripPlan make_strip_plan(int B, int H, int W);
int strip_select(int rows, int shape);      // test hook; non-zero = rejected
int launch_strips(const WarpArgs& a, const StripPlan& p, char* ws, hipStream_t stream);
int launch_warp_finish(const float* partial, int n, float* sums, const unsigned* count, const int2* rec, unsigned cap,
                       float* g_d2, hipStream_t stream);

}  // namespace dvd
