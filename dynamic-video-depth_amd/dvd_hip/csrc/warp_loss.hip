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
// wave-uniform and lives in SGPRs.  The bilinear taps of depth_2 are gathered
// as two 8-byte loads (west/east taps are adjacent); the depth_2 gradient is
// scattered with hardware fp32 atomics (global_atomic_add_f32).
//
// Numerics: the forward follows the reference's fp32 operation order exactly
// (see dvd_common.h rowvec_mat3 and sample_coord/bilinear below), so the
// index masks [depth_1<100], [W2.z<100], [I.z<1e-3] and the tap indices are
// bit-identical to PyTorch's CPU path.  Build with -ffp-contract=off.

#include "dvd_common.h"

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

// (x + flow) -> normalised -> un-normalised -> border clamp: the five fp32
// roundings of backward_warp + torch's grid_sample (align_corners=True).
__device__ __forceinline__ float sample_coord(float pix, float fl, float half, float maxv) {
  float g = pix + fl;
  g = g / half;
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

template <bool GRADS>
__device__ __forceinline__ void pixel(const WarpArgs& a, const Cam& c, int b, int y, int x,
                                      float d1, float fx, float fy, float mk, float s0, float s1,
                                      float s2, float acc[4], float& g_d1_out, float g_s_out[3]) {
  const float xf = (float)x, yf = (float)y;
  // --- frame-1 point: ray = (x,y,1) @ K_inv ; p1c = d1*ray ; P1 = p1c@R1 + t1
  float r0, r1, r2;
  rowvec_mat3(xf, yf, 1.0f, c.Ki, r0, r1, r2);
  const float pc0 = d1 * r0, pc1 = d1 * r1, pc2 = d1 * r2;
  float P0, P1, P2;
  rowvec_mat3(pc0, pc1, pc2, c.R1, P0, P1, P2);
  P0 = P0 + c.t1[0];
  P1 = P1 + c.t1[1];
  P2 = P2 + c.t1[2];

  // --- bilinear taps of frame 2 at (x,y)+flow
  const float ix = sample_coord(xf, fx, a.half_w, a.wmax);
  const float iy = sample_coord(yf, fy, a.half_h, a.hmax);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float ww = ix - x0f, we = 1.0f - ww;
  const float wn = iy - y0f, ws = 1.0f - wn;
  const float w_nw = ws * we, w_ne = ws * ww, w_sw = wn * we, w_se = wn * ww;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const bool in_e = (x0 + 1) < a.W, in_s = (y0 + 1) < a.H;  // x0,y0 are always in range
  const float* d2b = a.d2 + (size_t)b * a.HW;
  const int o_n = y0 * a.W + x0;
  const int o_s = o_n + a.W;
  float dnw, dne, dsw, dse;
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
  // rays of the four tap pixels (same expression as for the own pixel)
  float rn0[2], rn1[2], rn2[2], rs0[2], rs1[2], rs2[2];
  const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
  rowvec_mat3(x0f, y0f, 1.0f, c.Ki, rn0[0], rn1[0], rn2[0]);
  rowvec_mat3(x1f, y0f, 1.0f, c.Ki, rn0[1], rn1[1], rn2[1]);
  rowvec_mat3(x0f, y1f, 1.0f, c.Ki, rs0[0], rs1[0], rs2[0]);
  rowvec_mat3(x1f, y1f, 1.0f, c.Ki, rs0[1], rs1[1], rs2[1]);
  // camera-2 points at the taps (0 for out-of-image taps, like ATen's masked gather)
  const float cnw0 = dnw * rn0[0], cnw1 = dnw * rn1[0], cnw2 = dnw * rn2[0];
  const float cne0 = in_e ? dne * rn0[1] : 0.0f, cne1 = in_e ? dne * rn1[1] : 0.0f,
              cne2 = in_e ? dne * rn2[1] : 0.0f;
  const float csw0 = in_s ? dsw * rs0[0] : 0.0f, csw1 = in_s ? dsw * rs1[0] : 0.0f,
              csw2 = in_s ? dsw * rs2[0] : 0.0f;
  const bool in_se = in_e && in_s;
  const float cse0 = in_se ? dse * rs0[1] : 0.0f, cse1 = in_se ? dse * rs1[1] : 0.0f,
              cse2 = in_se ? dse * rs2[1] : 0.0f;
  // W2 = warped_p2_camera_2 ; only z feeds masks/loss, x,y are not needed here
  const float W2z = bilinear(cnw2, cne2, csw2, cse2, w_nw, w_ne, w_sw, w_se);
  // warped world point of frame 2 (for sf_by_depth)
  float gnw0, gnw1, gnw2, gne0, gne1, gne2, gsw0, gsw1, gsw2, gse0, gse1, gse2;
  rowvec_mat3(cnw0, cnw1, cnw2, c.R2, gnw0, gnw1, gnw2);
  gnw0 += c.t2[0];
  gnw1 += c.t2[1];
  gnw2 += c.t2[2];
  rowvec_mat3(cne0, cne1, cne2, c.R2, gne0, gne1, gne2);
  gne0 = in_e ? gne0 + c.t2[0] : 0.0f;
  gne1 = in_e ? gne1 + c.t2[1] : 0.0f;
  gne2 = in_e ? gne2 + c.t2[2] : 0.0f;
  rowvec_mat3(csw0, csw1, csw2, c.R2, gsw0, gsw1, gsw2);
  gsw0 = in_s ? gsw0 + c.t2[0] : 0.0f;
  gsw1 = in_s ? gsw1 + c.t2[1] : 0.0f;
  gsw2 = in_s ? gsw2 + c.t2[2] : 0.0f;
  rowvec_mat3(cse0, cse1, cse2, c.R2, gse0, gse1, gse2);
  gse0 = in_se ? gse0 + c.t2[0] : 0.0f;
  gse1 = in_se ? gse1 + c.t2[1] : 0.0f;
  gse2 = in_se ? gse2 + c.t2[2] : 0.0f;
  const float G0 = bilinear(gnw0, gne0, gsw0, gse0, w_nw, w_ne, w_sw, w_se);
  const float G1 = bilinear(gnw1, gne1, gsw1, gse1, w_nw, w_ne, w_sw, w_se);
  const float G2 = bilinear(gnw2, gne2, gsw2, gse2, w_nw, w_ne, w_sw, w_se);
  const float sb0 = G0 - P0, sb1 = G1 - P1, sb2 = G2 - P2;  // sf_by_depth

  // --- dynamic reprojection: Q = (P1 + s - t2) @ R2T ; I = Q @ K
  const float A0 = (P0 + s0) - c.t2[0], A1 = (P1 + s1) - c.t2[1], A2 = (P2 + s2) - c.t2[2];
  float Q0, Q1, Q2, I0, I1, I2;
  rowvec_mat3(A0, A1, A2, c.R2T, Q0, Q1, Q2);
  rowvec_mat3(Q0, Q1, Q2, c.K, I0, I1, I2);
  const float den = I2 + 1e-8f;
  const bool behind = I2 < 1e-3f;
  const float u = behind ? xf : I0 / den;
  const float v = behind ? yf : I1 / den;
  const float ex = (u - xf) - fx, ey = (v - yf) - fy;  // dflow - flow

  // --- mask and per-pixel errors
  float m = mk;
  if (a.midas_mask) {
    m = ((d1 < 100.0f) ? 1.0f : 0.0f) * m;
    m = ((W2z < 100.0f) ? 1.0f : 0.0f) * m;
  }
  const float flow_err = a.crit_l2 ? (ex * ex + ey * ey) : (fabsf(ex) + fabsf(ey));
  float disp_err, ca = 0.0f, cb = 0.0f, ediff = 0.0f;
  if (a.disp_mode == 1) {
    ca = fmaxf(Q2, 1e-3f);
    cb = fmaxf(W2z, 1e-3f);
    ediff = (1.0f / ca) - (1.0f / cb);
    disp_err = 100.0f * fabsf(ediff);
  } else if (a.disp_mode == 2) {
    ca = fmaxf(Q2, 1e-3f);
    cb = fmaxf(W2z, 1e-3f);
    disp_err = fmaxf(ca, cb) / fminf(ca, cb) - 1.0f;
  } else {
    disp_err = fabsf(Q2 - W2z);
  }
  const float f0 = sb0 - s0, f1 = sb1 - s1, f2 = sb2 - s2;
  const float sf_err = fabsf(f0) + fabsf(f1) + fabsf(f2);
  acc[0] += m;
  acc[1] += m * flow_err;
  acc[2] += m * disp_err;
  acc[3] += m * sf_err;

  if (!GRADS) return;
  // ------------------------------ backward (un-normalised) ----------------
  float gQ0 = 0.0f, gQ1 = 0.0f, gQ2 = 0.0f;
  const float fm = a.flow_mul * m;
  if (!behind && fm != 0.0f) {
    const float gu = a.crit_l2 ? fm * 2.0f * ex : fm * sgn(ex);
    const float gv = a.crit_l2 ? fm * 2.0f * ey : fm * sgn(ey);
    const float gI0 = gu / den, gI1 = gv / den;
    const float gI2 = -(gu * u + gv * v) / den;
    rowvec_mat3_T(gI0, gI1, gI2, c.K, gQ0, gQ1, gQ2);
  }
  float gW2z = 0.0f;  // d loss / d W2.z
  float gG0 = 0.0f, gG1 = 0.0f, gG2 = 0.0f;  // d loss / d warped world point
  const float dm = a.disp_mul * m;
  if (!a.loss_on_sf) {
    if (a.disp_mode == 1 && dm != 0.0f) {
      const float ge = dm * 100.0f * sgn(ediff);
      if (Q2 >= 1e-3f) gQ2 += -ge / (ca * ca);
      if (W2z >= 1e-3f) gW2z = ge / (cb * cb);
    }
  } else if (dm != 0.0f) {
    gG0 = dm * sgn(f0);
    gG1 = dm * sgn(f1);
    gG2 = dm * sgn(f2);
  }
  float gA0, gA1, gA2;
  rowvec_mat3_T(gQ0, gQ1, gQ2, c.R2T, gA0, gA1, gA2);
  // scene flow enters A (+) and, in sf-loss mode, the error term (-)
  g_s_out[0] = gA0 - gG0;
  g_s_out[1] = gA1 - gG1;
  g_s_out[2] = gA2 - gG2;
  // P1 enters A (+) and sf_by_depth (-)
  float gp0, gp1, gp2;
  rowvec_mat3_T(gA0 - gG0, gA1 - gG1, gA2 - gG2, c.R1, gp0, gp1, gp2);
  g_d1_out = gp0 * r0 + gp1 * r1 + gp2 * r2;
  // depth_2 taps: d(W2)/d(d2_k) = w_k * ray_k ; d(G)/d(d2_k) = w_k * ray_k @ R2
  float h0, h1, h2;
  rowvec_mat3_T(gG0, gG1, gG2, c.R2, h0, h1, h2);
  h2 += gW2z;
  if (h0 != 0.0f || h1 != 0.0f || h2 != 0.0f) {
    float* gb = a.g_d2 + (size_t)b * a.HW;
    const float t_nw = w_nw * (h0 * rn0[0] + h1 * rn1[0] + h2 * rn2[0]);
    unsafeAtomicAdd(gb + o_n, t_nw);
    if (in_e) unsafeAtomicAdd(gb + o_n + 1, w_ne * (h0 * rn0[1] + h1 * rn1[1] + h2 * rn2[1]));
    if (in_s) unsafeAtomicAdd(gb + o_s, w_sw * (h0 * rs0[0] + h1 * rs1[0] + h2 * rs2[0]));
    if (in_se) unsafeAtomicAdd(gb + o_s + 1, w_se * (h0 * rs0[1] + h1 * rs1[1] + h2 * rs2[1]));
  }
}

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
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      gd1[i] = 0.0f;
      gs[i][0] = gs[i][1] = gs[i][2] = 0.0f;
      pixel<GRADS>(a, c, b, y, x + i, d1[i], fl[2 * i], fl[2 * i + 1], mk[i], s0[i], s1[i], s2[i], acc,
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

// Second stage: fixed-order sum of the per-block partials (deterministic).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int n,
                                                              float* __restrict__ sums) {
  __shared__ double sh[256][4];
  double acc[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)i * 4);
    acc[0] += v.x;
    acc[1] += v.y;
    acc[2] += v.z;
    acc[3] += v.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] += sh[threadIdx.x + s][k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) sums[threadIdx.x] = (float)sh[0][threadIdx.x];
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
  DVD_REQUIRE((long long)cfg->B * HW * 3 < (1LL << 40), "warp_loss: tensor too large");
  const size_t need = dvd_warp_loss_workspace_bytes(cfg->B, cfg->H, cfg->W);
  if (workspace_bytes < need) {
    set_error("warp_loss: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
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
  const bool vec4 = (cfg->W % 4 == 0) && (((uintptr_t)depth_1 | (uintptr_t)flow_1_2 | (uintptr_t)mask_2 |
                                            (uintptr_t)sf_1_2 | (uintptr_t)g_depth_1 | (uintptr_t)g_sf_1_2) %
                                               16 ==
                                           0);
  const int px = vec4 ? 4 : 1;
  const int nbx = blocks_x(HW, px);
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
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, a.partial, nbx * cfg->B, sums);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // namespace dvd

extern "C" {

size_t dvd_warp_loss_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  // one float4 per block; sized for the scalar (PX=1) tiling, the larger of the two
  return (size_t)dvd::blocks_x(H * W, 1) * (size_t)B * 4 * sizeof(float);
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
