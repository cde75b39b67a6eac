// Forward/backward flow-consistency (occlusion) and out-of-bounds masks of a frame pair, on the device.
//
// What it replaces (reference, /root/reference/scripts/preprocess/davis/generate_flows.py):
//   :57-68   get_oob_mask         target (x, y) + flow outside [0, W-1] x [0, H-1]
//   :71-82   backward_flow_warp   F.grid_sample(flow_a, (x, y) + flow_b, bilinear, padding 'zeros', align_corners=True)
//   :139-148 mask = clip([ || warp(flow_a; flow_b) + flow_b ||_2 > 1 ] + oob(flow_b), 0, 1)
// which the reference evaluates in numpy / ATen-CPU per pair and stores as uint8; the training masks are then
// 1 - ceil(mask) (generate_sequence_midas.py:144-147).  Same fp32 operation order as that path (grid
// normalisation by (W-1)/2, un-normalisation (g + 1) * (W-1)/2, weights w = x - floor(x), e = 1 - w, bilinear
// as mul + 3 fma, norm as sqrt(a*a + b*b) with separate roundings), so the integer masks are bit-identical.
// HBM bound: 16 B read + 4 B written per pixel and direction; built with -ffp-contract=off.
#include "dvd_common.h"

namespace dvd {

__device__ __forceinline__ float c_rcp_refined(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float c_div(float a, float b, float y) {   // IEEE a / b, y = c_rcp_refined(b)
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
// (pix + flow) / half - 1, then (g + 1) * half: no clamp (padding_mode 'zeros')
__device__ __forceinline__ float c_coord(float pix, float fl, float half) {
  float g = pix + fl;
  g = c_div(g, half, c_rcp_refined(half));
  g = g - 1.0f;
  return (g + 1.0f) * half;
}

// mask[b][y][x] in {0, 1}; flows [B,H,W,2]
__global__ __launch_bounds__(256) void flow_consistency_kernel(const float* __restrict__ flow_a, const float* __restrict__ flow_b,
                                                               float* __restrict__ mask, int H, int W, int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const size_t lin = (size_t)b * HW + p;
  const float2 fb = *reinterpret_cast<const float2*>(flow_b + 2 * lin);
  const float xf = (float)x, yf = (float)y;
  const float ix = c_coord(xf, fb.x, (float)((W - 1) / 2.0));
  const float iy = c_coord(yf, fb.y, (float)((H - 1) / 2.0));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float ww = ix - x0f, we = 1.0f - ww, wn = iy - y0f, ws = 1.0f - wn;
  const float w_nw = ws * we, w_ne = ws * ww, w_sw = wn * we, w_se = wn * ww;
  const float wm = (float)(W - 1), hm = (float)(H - 1);
  const bool xw = x0f >= 0.0f && x0f <= wm, xe = (x0f + 1.0f) >= 0.0f && (x0f + 1.0f) <= wm;
  const bool yn = y0f >= 0.0f && y0f <= hm, ysb = (y0f + 1.0f) >= 0.0f && (y0f + 1.0f) <= hm;
  const float* ab = flow_a + (size_t)b * HW * 2;
  float2 vnw = make_float2(0.f, 0.f), vne = vnw, vsw = vnw, vse = vnw;
  if (xw || xe) {   // (NaN / huge coordinates: every tap is out of bounds)
    const int x0 = (int)x0f, y0 = (int)y0f;
    if (xw && yn) vnw = *reinterpret_cast<const float2*>(ab + 2 * ((size_t)y0 * W + x0));
    if (xe && yn) vne = *reinterpret_cast<const float2*>(ab + 2 * ((size_t)y0 * W + x0 + 1));
    if (xw && ysb) vsw = *reinterpret_cast<const float2*>(ab + 2 * ((size_t)(y0 + 1) * W + x0));
    if (xe && ysb) vse = *reinterpret_cast<const float2*>(ab + 2 * ((size_t)(y0 + 1) * W + x0 + 1));
  }
  float s0 = vnw.x * w_nw;
  s0 = __builtin_fmaf(vne.x, w_ne, s0);
  s0 = __builtin_fmaf(vsw.x, w_sw, s0);
  s0 = __builtin_fmaf(vse.x, w_se, s0);
  float s1 = vnw.y * w_nw;
  s1 = __builtin_fmaf(vne.y, w_ne, s1);
  s1 = __builtin_fmaf(vsw.y, w_sw, s1);
  s1 = __builtin_fmaf(vse.y, w_se, s1);
  const float e0 = s0 + fb.x, e1 = s1 + fb.y;
  const float q0 = e0 * e0, q1 = e1 * e1;
  const float err = sqrtf(q0 + q1);
  const float tx = xf + fb.x, ty = yf + fb.y;
  const bool oob = (tx < 0.0f) || (tx > wm) || (ty < 0.0f) || (ty > hm);
  mask[lin] = (err > 1.0f || oob) ? 1.0f : 0.0f;
}

}  // namespace dvd

extern "C" {

int dvd_flow_consistency_mask(const float* flow_a, const float* flow_b, float* mask, int B, int H, int W,
                              dvd_stream_t stream) {
  DVD_REQUIRE(flow_a && flow_b && mask, "flow_consistency_mask: null pointer");
  DVD_REQUIRE(B > 0 && H > 1 && W > 1 && B <= 65535, "flow_consistency_mask: bad shape B=%d H=%d W=%d", B, H, W);
  hipLaunchKernelGGL(dvd::flow_consistency_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), flow_a, flow_b, mask, H, W, H * W);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
