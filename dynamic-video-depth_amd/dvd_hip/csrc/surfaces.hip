// Per-pixel surfaces of the warp modules and the stand-alone bilinear flow warp (gfx950).
//
// The training step never materialises these (dvd_warp_loss_fused consumes them in registers);
// they exist for the reference's operator surface -- visualisation, np.savez export and the
// inference path ask for them (models/scene_flow_motion_field.py:215-225, video_base.py:105-126):
//   flow_by_depth.forward                losses/scene_flow_projection.py:114-153
//       -> dflow_1_2 (static), sf_by_depth, warped_global_p2, global_p1
//   scene_flow_projection_slack.forward  losses/scene_flow_projection.py:222-278
//       -> dflow_1_2, depth_image_1_2, depth_warp_1_2, staticflow_1_2, p1_camera_2,
//          warped_p2_camera_2, global_p1 (+ pass-throughs the host returns as they are)
//   BackwardWarp.forward                 losses/scene_flow_projection.py:281-307 (F.grid_sample,
//       bilinear, padding_mode='border', align_corners=True) and its backward w.r.t. the buffer.
//
// Roofline: HBM (one thread per pixel, every requested surface written once: up to 92 B/pixel
// out for 24 B/pixel in).  Arithmetic follows the reference's fp32 operation order everywhere
// (rowvec_mat3, five-rounding sampling coordinate, ATen's mul + 3 fma bilinear, IEEE division),
// so the surfaces are bit-identical to the CPU reference up to libm-free fp32 semantics;
// built with -ffp-contract=off.

#include "dvd_common.h"

namespace dvd {

__device__ __forceinline__ float s_rcp_refined(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
// a / b (IEEE) through the unscaled division sequence; y = s_rcp_refined(b)
__device__ __forceinline__ float s_div(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float s_coord(float pix, float fl, float half, float maxv) {
  float g = pix + fl;
  g = s_div(g, half, s_rcp_refined(half));
  g = g - 1.0f;
  const float i = (g + 1.0f) * half;
  return fminf(maxv, fmaxf(i, 0.0f));
}
__device__ __forceinline__ float s_bilinear(float vnw, float vne, float vsw, float vse, float wnw, float wne,
                                            float wsw, float wse) {
  float r = vnw * wnw;
  r = __builtin_fmaf(vne, wne, r);
  r = __builtin_fmaf(vsw, wsw, r);
  return __builtin_fmaf(vse, wse, r);
}

struct Taps {
  int x0, y0;
  bool in_e, in_s;
  float w_nw, w_ne, w_sw, w_se;
};
__device__ __forceinline__ Taps make_taps(float xf, float yf, float fx, float fy, int H, int W) {
  Taps t;
  const float ix = s_coord(xf, fx, (float)((W - 1) / 2.0), (float)(W - 1));
  const float iy = s_coord(yf, fy, (float)((H - 1) / 2.0), (float)(H - 1));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float ww = ix - x0f, we = 1.0f - ww, wn = iy - y0f, ws = 1.0f - wn;
  t.w_nw = ws * we;
  t.w_ne = ws * ww;
  t.w_sw = wn * we;
  t.w_se = wn * ww;
  t.x0 = (int)x0f;
  t.y0 = (int)y0f;
  t.in_e = t.x0 + 1 < W;
  t.in_s = t.y0 + 1 < H;
  return t;
}

struct SurfArgs {
  const float *d1, *d2, *flow, *sflow;   // sflow: interleaved [B,H,W,3] or null (zero scene flow)
  const float *R1, *R2, *R2T, *t1, *t2, *K, *Ki;
  dvd_surfaces out;
  int H, W, HW;
};

__global__ __launch_bounds__(256) void warp_surfaces_kernel(const SurfArgs a) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.HW) return;
  float Ki[9], R1[9], R2[9], R2T[9], K[9], t1[3], t2[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Ki[i] = a.Ki[b * 9 + i];
    R1[i] = a.R1[b * 9 + i];
    R2[i] = a.R2[b * 9 + i];
    R2T[i] = a.R2T[b * 9 + i];
    K[i] = a.K[b * 9 + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t1[i] = a.t1[b * 3 + i];
    t2[i] = a.t2[b * 3 + i];
  }
  const int y = p / a.W, x = p - y * a.W;
  const float xf = (float)x, yf = (float)y;
  const size_t lin = (size_t)b * a.HW + p;
  const float d1 = a.d1[lin];
  const float fx = a.flow[2 * lin], fy = a.flow[2 * lin + 1];
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  if (a.sflow) {
    s0 = a.sflow[3 * lin];
    s1 = a.sflow[3 * lin + 1];
    s2 = a.sflow[3 * lin + 2];
  }
  // pixel 1: ray, camera point, world point
  float r0, r1, r2, P0, P1, P2;
  rowvec_mat3(xf, yf, 1.0f, Ki, r0, r1, r2);
  rowvec_mat3(d1 * r0, d1 * r1, d1 * r2, R1, P0, P1, P2);
  P0 = P0 + t1[0];
  P1 = P1 + t1[1];
  P2 = P2 + t1[2];
  // taps of frame 2 at (x, y) + flow: camera-2 points, world points and depth of the four taps
  const Taps t = make_taps(xf, yf, fx, fy, a.H, a.W);
  const float* d2b = a.d2 + (size_t)b * a.HW;
  float dk[4], pc[4][3], pw[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dx = k & 1, dy = k >> 1;
    const bool in = (dx == 0 || t.in_e) && (dy == 0 || t.in_s);
    const int tx = t.x0 + dx, ty = t.y0 + dy;
    const float d = in ? d2b[ty * a.W + tx] : 0.0f;
    float q0, q1, q2;
    rowvec_mat3((float)tx, (float)ty, 1.0f, Ki, q0, q1, q2);
    const float c0 = d * q0, c1 = d * q1, c2 = d * q2;
    float w0, w1, w2;
    rowvec_mat3(c0, c1, c2, R2, w0, w1, w2);
    // ATen's bilinear gather returns 0 for an out-of-image tap (its weight is 0 as well)
    dk[k] = d;
    pc[k][0] = in ? c0 : 0.0f;
    pc[k][1] = in ? c1 : 0.0f;
    pc[k][2] = in ? c2 : 0.0f;
    pw[k][0] = in ? w0 + t2[0] : 0.0f;
    pw[k][1] = in ? w1 + t2[1] : 0.0f;
    pw[k][2] = in ? w2 + t2[2] : 0.0f;
  }
  float W2c[3], G[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    W2c[c] = s_bilinear(pc[0][c], pc[1][c], pc[2][c], pc[3][c], t.w_nw, t.w_ne, t.w_sw, t.w_se);
    G[c] = s_bilinear(pw[0][c], pw[1][c], pw[2][c], pw[3][c], t.w_nw, t.w_ne, t.w_sw, t.w_se);
  }
  const float dwarp = s_bilinear(dk[0], dk[1], dk[2], dk[3], t.w_nw, t.w_ne, t.w_sw, t.w_se);
  // dynamic and static reprojection of pixel 1 into image 2
  float Qd0, Qd1, Qd2, Id0, Id1, Id2, Qs0, Qs1, Qs2, Is0, Is1, Is2;
  rowvec_mat3((P0 + s0) - t2[0], (P1 + s1) - t2[1], (P2 + s2) - t2[2], R2T, Qd0, Qd1, Qd2);
  rowvec_mat3(Qd0, Qd1, Qd2, K, Id0, Id1, Id2);
  rowvec_mat3(P0 - t2[0], P1 - t2[1], P2 - t2[2], R2T, Qs0, Qs1, Qs2);
  rowvec_mat3(Qs0, Qs1, Qs2, K, Is0, Is1, Is2);
  float ud = xf, vd = yf, us = xf, vs = yf;
  if (!(Id2 < 1e-3f)) {
    const float den = Id2 + 1e-8f, yd = s_rcp_refined(den);
    ud = s_div(Id0, den, yd);
    vd = s_div(Id1, den, yd);
  }
  if (!(Is2 < 1e-3f)) {
    const float den = Is2 + 1e-8f, yd = s_rcp_refined(den);
    us = s_div(Is0, den, yd);
    vs = s_div(Is1, den, yd);
  }
  const dvd_surfaces& o = a.out;
  if (o.global_p1) {
    o.global_p1[3 * lin] = P0;
    o.global_p1[3 * lin + 1] = P1;
    o.global_p1[3 * lin + 2] = P2;
  }
  if (o.warped_global_p2) {
    o.warped_global_p2[3 * lin] = G[0];
    o.warped_global_p2[3 * lin + 1] = G[1];
    o.warped_global_p2[3 * lin + 2] = G[2];
  }
  if (o.sf_by_depth) {
    o.sf_by_depth[3 * lin] = G[0] - P0;
    o.sf_by_depth[3 * lin + 1] = G[1] - P1;
    o.sf_by_depth[3 * lin + 2] = G[2] - P2;
  }
  if (o.staticflow_1_2) {
    o.staticflow_1_2[2 * lin] = us - xf;
    o.staticflow_1_2[2 * lin + 1] = vs - yf;
  }
  if (o.dflow_1_2) {
    o.dflow_1_2[2 * lin] = ud - xf;
    o.dflow_1_2[2 * lin + 1] = vd - yf;
  }
  if (o.depth_image_1_2) o.depth_image_1_2[lin] = Id2;
  if (o.depth_warp_1_2) o.depth_warp_1_2[lin] = dwarp;
  if (o.p1_camera_2) {
    o.p1_camera_2[3 * lin] = Qd0;
    o.p1_camera_2[3 * lin + 1] = Qd1;
    o.p1_camera_2[3 * lin + 2] = Qd2;
  }
  if (o.warped_p2_camera_2) {
    o.warped_p2_camera_2[3 * lin] = W2c[0];
    o.warped_p2_camera_2[3 * lin + 1] = W2c[1];
    o.warped_p2_camera_2[3 * lin + 2] = W2c[2];
  }
}

// Vector-Jacobian product of warp_surfaces_kernel: upstream gradients of any subset of the surfaces (null = none) ->
// gradients w.r.t. depth_1, depth_2 (scatter over the four bilinear taps; the flow is data) and the scene flow.  This is
// what makes the MODULE forms of flow_by_depth / scene_flow_projection_slack differentiable like the reference's
// (losses/scene_flow_projection.py:114-153,222-278 under autograd); the training step uses the fused kernel instead.
// The index_put of the behind-camera fallback cuts the gradient of the projected coordinates (losses:253-263).
struct SurfBwdArgs {
  const float *d1, *d2, *flow, *sflow;
  const float *R1, *R2, *R2T, *t1, *t2, *K, *Ki;
  dvd_surfaces g;                        // upstream gradients, same layouts as the surfaces
  float *g_d1, *g_d2, *g_sflow;          // g_d2 zeroed by the caller (atomic scatter); g_sflow may be null
  int H, W, HW;
};

__global__ __launch_bounds__(256) void warp_surfaces_bwd_kernel(const SurfBwdArgs a) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.HW) return;
  float Ki[9], R1[9], R2[9], R2T[9], K[9], t1[3], t2[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Ki[i] = a.Ki[b * 9 + i];
    R1[i] = a.R1[b * 9 + i];
    R2[i] = a.R2[b * 9 + i];
    R2T[i] = a.R2T[b * 9 + i];
    K[i] = a.K[b * 9 + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t1[i] = a.t1[b * 3 + i];
    t2[i] = a.t2[b * 3 + i];
  }
  const int y = p / a.W, x = p - y * a.W;
  const float xf = (float)x, yf = (float)y;
  const size_t lin = (size_t)b * a.HW + p;
  const float d1 = a.d1[lin];
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  if (a.sflow) {
    s0 = a.sflow[3 * lin];
    s1 = a.sflow[3 * lin + 1];
    s2 = a.sflow[3 * lin + 2];
  }
  auto g3 = [&](const float* q, float (&v)[3]) {
    v[0] = q ? q[3 * lin] : 0.0f;
    v[1] = q ? q[3 * lin + 1] : 0.0f;
    v[2] = q ? q[3 * lin + 2] : 0.0f;
  };
  float gP[3], gG[3], gS[3], gQd[3], gW2[3];
  g3(a.g.global_p1, gP);
  g3(a.g.warped_global_p2, gG);
  g3(a.g.sf_by_depth, gS);
  g3(a.g.p1_camera_2, gQd);
  g3(a.g.warped_p2_camera_2, gW2);
#pragma unroll
  for (int c = 0; c < 3; ++c) {          // sf_by_depth = warped_global_p2 - global_p1
    gG[c] += gS[c];
    gP[c] -= gS[c];
  }
  // forward values needed by the projections
  float r0, r1, r2, P0, P1, P2;
  rowvec_mat3(xf, yf, 1.0f, Ki, r0, r1, r2);
  rowvec_mat3(d1 * r0, d1 * r1, d1 * r2, R1, P0, P1, P2);
  P0 += t1[0];
  P1 += t1[1];
  P2 += t1[2];
  // dynamic / static reprojection: gradient of (u - x, v - y) and of I.z w.r.t. A = P (+ s) - t2
  auto reproject_bwd = [&](float A0, float A1, float A2, const float* gflow, float gz, float gQ_in0, float gQ_in1,
                           float gQ_in2, float (&gA)[3]) {
    float Q0, Q1, Q2, I0, I1, I2;
    rowvec_mat3(A0, A1, A2, R2T, Q0, Q1, Q2);
    rowvec_mat3(Q0, Q1, Q2, K, I0, I1, I2);
    float gI0 = 0.0f, gI1 = 0.0f, gI2 = gz;
    if (gflow && !(I2 < 1e-3f)) {
      const float den = I2 + 1e-8f, rden = 1.0f / den;
      const float gu = gflow[2 * lin], gv = gflow[2 * lin + 1];
      gI0 = gu * rden;
      gI1 = gv * rden;
      gI2 -= (gu * I0 + gv * I1) * rden * rden;
    }
    float q0, q1, q2;
    rowvec_mat3_T(gI0, gI1, gI2, K, q0, q1, q2);
    rowvec_mat3_T(q0 + gQ_in0, q1 + gQ_in1, q2 + gQ_in2, R2T, gA[0], gA[1], gA[2]);
  };
  float gAd[3], gAs[3];
  reproject_bwd((P0 + s0) - t2[0], (P1 + s1) - t2[1], (P2 + s2) - t2[2], a.g.dflow_1_2,
                a.g.depth_image_1_2 ? a.g.depth_image_1_2[lin] : 0.0f, gQd[0], gQd[1], gQd[2], gAd);
  reproject_bwd(P0 - t2[0], P1 - t2[1], P2 - t2[2], a.g.staticflow_1_2, 0.0f, 0.0f, 0.0f, 0.0f, gAs);
  if (a.g_sflow) {
    a.g_sflow[3 * lin] = gAd[0];
    a.g_sflow[3 * lin + 1] = gAd[1];
    a.g_sflow[3 * lin + 2] = gAd[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) gP[c] += gAd[c] + gAs[c];
  {  // P = (d1 ray) @ R1 + t1
    float c0, c1, c2;
    rowvec_mat3_T(gP[0], gP[1], gP[2], R1, c0, c1, c2);
    a.g_d1[lin] = c0 * r0 + c1 * r1 + c2 * r2;
  }
  // taps of frame 2: d/d(d2_k) = w_k (g_depth_warp + ray_k . (g_W2c + g_G @ R2^T))
  float h0, h1, h2;
  rowvec_mat3_T(gG[0], gG[1], gG[2], R2, h0, h1, h2);
  h0 += gW2[0];
  h1 += gW2[1];
  h2 += gW2[2];
  const float gdw = a.g.depth_warp_1_2 ? a.g.depth_warp_1_2[lin] : 0.0f;
  const Taps t = make_taps(xf, yf, a.flow[2 * lin], a.flow[2 * lin + 1], a.H, a.W);
  float* g2 = a.g_d2 + (size_t)b * a.HW;
  const float wk[4] = {t.w_nw, t.w_ne, t.w_sw, t.w_se};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dx = k & 1, dy = k >> 1;
    if ((dx == 1 && !t.in_e) || (dy == 1 && !t.in_s)) continue;
    const int tx = t.x0 + dx, ty = t.y0 + dy;
    float q0, q1, q2;
    rowvec_mat3((float)tx, (float)ty, 1.0f, Ki, q0, q1, q2);
    const float v = wk[k] * (gdw + (q0 * h0 + q1 * h1 + q2 * h2));
    if (v != 0.0f) unsafeAtomicAdd(g2 + ty * a.W + tx, v);
  }
}

// out[b,c,y,x] = bilinear(src[b,c], (x,y) + flow[b,y,x])       (forward)
// g_src[b,c,tap] += w_tap * g_out[b,c,y,x]                      (backward w.r.t. the buffer)
template <bool BACKWARD>
__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                        float* __restrict__ dst, int C, int H, int W, int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const size_t lin = (size_t)b * HW + p;
  const Taps t = make_taps((float)x, (float)y, flow[2 * lin], flow[2 * lin + 1], H, W);
  const int o_n = t.y0 * W + t.x0;
  for (int c = 0; c < C; ++c) {
    const size_t plane = ((size_t)b * C + c) * HW;
    if (!BACKWARD) {
      const float* s = src + plane;
      const float vnw = s[o_n], vne = t.in_e ? s[o_n + 1] : 0.0f, vsw = t.in_s ? s[o_n + W] : 0.0f,
                  vse = (t.in_e && t.in_s) ? s[o_n + W + 1] : 0.0f;
      dst[plane + p] = s_bilinear(vnw, vne, vsw, vse, t.w_nw, t.w_ne, t.w_sw, t.w_se);
    } else {
      const float g = src[plane + p];   // src = g_out, dst = g_buffer (zeroed by the caller)
      float* d = dst + plane;
      unsafeAtomicAdd(d + o_n, g * t.w_nw);
      if (t.in_e) unsafeAtomicAdd(d + o_n + 1, g * t.w_ne);
      if (t.in_s) unsafeAtomicAdd(d + o_n + W, g * t.w_sw);
      if (t.in_e && t.in_s) unsafeAtomicAdd(d + o_n + W + 1, g * t.w_se);
    }
  }
}

}  // namespace dvd

extern "C" {

int dvd_warp_surfaces(const float* depth_1, const float* depth_2, const float* flow_1_2, const float* sflow_1_2,
                      const dvd_cameras* cams, const dvd_surfaces* out, int B, int H, int W, dvd_stream_t stream) {
  DVD_REQUIRE(depth_1 && depth_2 && flow_1_2 && cams && out, "warp_surfaces: null pointer");
  DVD_REQUIRE(B > 0 && H > 1 && W > 1 && B <= 65535, "warp_surfaces: bad shape B=%d H=%d W=%d", B, H, W);
  DVD_REQUIRE((long long)B * H * W * 3 < (1LL << 31), "warp_surfaces: tensor too large for 32-bit indexing");
  DVD_REQUIRE(cams->R_1 && cams->R_2 && cams->R_2_T && cams->t_1 && cams->t_2 && cams->K && cams->K_inv,
              "warp_surfaces: null camera pointer");
  dvd::SurfArgs a;
  a.d1 = depth_1;
  a.d2 = depth_2;
  a.flow = flow_1_2;
  a.sflow = sflow_1_2;
  a.R1 = cams->R_1;
  a.R2 = cams->R_2;
  a.R2T = cams->R_2_T;
  a.t1 = cams->t_1;
  a.t2 = cams->t_2;
  a.K = cams->K;
  a.Ki = cams->K_inv;
  a.out = *out;
  a.H = H;
  a.W = W;
  a.HW = H * W;
  hipLaunchKernelGGL(dvd::warp_surfaces_kernel, dim3((a.HW + 255) / 256, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_warp_surfaces_bwd(const float* depth_1, const float* depth_2, const float* flow_1_2, const float* sflow_1_2,
                          const dvd_cameras* cams, const dvd_surfaces* g_surfaces, float* g_depth_1, float* g_depth_2,
                          float* g_sflow_1_2, int B, int H, int W, dvd_stream_t stream) {
  DVD_REQUIRE(depth_1 && depth_2 && flow_1_2 && cams && g_surfaces && g_depth_1 && g_depth_2, "warp_surfaces_bwd: null pointer");
  DVD_REQUIRE(B > 0 && H > 1 && W > 1 && B <= 65535, "warp_surfaces_bwd: bad shape B=%d H=%d W=%d", B, H, W);
  DVD_REQUIRE((long long)B * H * W * 3 < (1LL << 31), "warp_surfaces_bwd: tensor too large for 32-bit indexing");
  DVD_REQUIRE(cams->R_1 && cams->R_2 && cams->R_2_T && cams->t_1 && cams->t_2 && cams->K && cams->K_inv,
              "warp_surfaces_bwd: null camera pointer");
  dvd::SurfBwdArgs a;
  a.d1 = depth_1;
  a.d2 = depth_2;
  a.flow = flow_1_2;
  a.sflow = sflow_1_2;
  a.R1 = cams->R_1;
  a.R2 = cams->R_2;
  a.R2T = cams->R_2_T;
  a.t1 = cams->t_1;
  a.t2 = cams->t_2;
  a.K = cams->K;
  a.Ki = cams->K_inv;
  a.g = *g_surfaces;
  a.g_d1 = g_depth_1;
  a.g_d2 = g_depth_2;
  a.g_sflow = g_sflow_1_2;
  a.H = H;
  a.W = W;
  a.HW = H * W;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DVD_HIP_OK(hipMemsetAsync(g_depth_2, 0, (size_t)B * H * W * sizeof(float), s));
  hipLaunchKernelGGL(dvd::warp_surfaces_bwd_kernel, dim3((a.HW + 255) / 256, B), dim3(256), 0, s, a);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_flow_warp_fwd(const float* buffer, const float* flow_1_2, float* out, int B, int C, int H, int W,
                      dvd_stream_t stream) {
  DVD_REQUIRE(buffer && flow_1_2 && out, "flow_warp_fwd: null pointer");
  DVD_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1 && B <= 65535, "flow_warp_fwd: bad shape");
  hipLaunchKernelGGL(dvd::flow_warp_kernel<false>, dim3((H * W + 255) / 256, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), buffer, flow_1_2, out, C, H, W, H * W);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_flow_warp_bwd(const float* g_out, const float* flow_1_2, float* g_buffer, int B, int C, int H, int W,
                      dvd_stream_t stream) {
  DVD_REQUIRE(g_out && flow_1_2 && g_buffer, "flow_warp_bwd: null pointer");
  DVD_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1 && B <= 65535, "flow_warp_bwd: bad shape");
  DVD_HIP_OK(hipMemsetAsync(g_buffer, 0, (size_t)B * C * H * W * sizeof(float), static_cast<hipStream_t>(stream)));
  hipLaunchKernelGGL(dvd::flow_warp_kernel<true>, dim3((H * W + 255) / 256, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g_out, flow_1_2, g_buffer, C, H, W, H * W);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
