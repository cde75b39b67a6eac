// depth -> world points and its backward (gfx950).
//
// Replaces unproject_ptcld.forward (/root/reference/losses/scene_flow_projection.py:54-67)
// and the global_p1 half of flow_by_depth.forward (:127-131):
//     ray = (x, y, 1) @ K_inv ;  p_cam = depth * ray ;  P = p_cam @ R + t
// Roofline: HBM -- 4 B read + 12 B written per pixel forward; 12 B read (+4 when
// accumulating) + 4 B written backward.  One thread per 4 horizontally adjacent
// pixels (16-byte accesses on the planar layout the scene-flow MLP consumes).
// Same fp32 operation order as torch's CPU matmul (dvd_common.h), built with
// -ffp-contract=off, so the points are bit-identical to the reference's.

#include "dvd_common.h"

namespace dvd {

template <int PX, bool PLANAR>
__global__ __launch_bounds__(256) void unproject_fwd_kernel(const float* __restrict__ depth,
                                                            const float* __restrict__ Rm,
                                                            const float* __restrict__ tv,
                                                            const float* __restrict__ Kinv,
                                                            float* __restrict__ out, int H, int W, int HW) {
  const int b = blockIdx.y;
  float Ki[9], R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Ki[i] = Kinv[b * 9 + i];
    R[i] = Rm[b * 9 + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = tv[b * 3 + i];
  const int p0 = (blockIdx.x * 256 + threadIdx.x) * PX;
  if (p0 >= HW) return;
  const size_t base = (size_t)b * HW + p0;
  float d[PX];
  if (PX == 4) {
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(depth + base);
  } else {
    d[0] = depth[base];
  }
  const int y = p0 / W, x = p0 - y * W;
  float P[3][PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    float r0, r1, r2, q0, q1, q2;
    rowvec_mat3((float)(x + i), (float)y, 1.0f, Ki, r0, r1, r2);
    rowvec_mat3(d[i] * r0, d[i] * r1, d[i] * r2, R, q0, q1, q2);
    P[0][i] = q0 + t[0];
    P[1][i] = q1 + t[1];
    P[2][i] = q2 + t[2];
  }
  if (PLANAR) {
    float* o = out + (size_t)b * 3 * HW + p0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (PX == 4)
        *reinterpret_cast<float4*>(o + (size_t)c * HW) = make_float4(P[c][0], P[c][1], P[c][2], P[c][3]);
      else
        o[(size_t)c * HW] = P[c][0];
    }
  } else {
    float* o = out + base * 3;
    if (PX == 4) {
      float4* o4 = reinterpret_cast<float4*>(o);
      o4[0] = make_float4(P[0][0], P[1][0], P[2][0], P[0][1]);
      o4[1] = make_float4(P[1][1], P[2][1], P[0][2], P[1][2]);
      o4[2] = make_float4(P[2][2], P[0][3], P[1][3], P[2][3]);
    } else {
      o[0] = P[0][0];
      o[1] = P[1][0];
      o[2] = P[2][0];
    }
  }
}

// g_depth (+)= scale * ((g_P @ R^T) . ray)
template <int PX, bool PLANAR>
__global__ __launch_bounds__(256) void unproject_bwd_kernel(const float* __restrict__ gP,
                                                            const float* __restrict__ Rm,
                                                            const float* __restrict__ Kinv,
                                                            const float* __restrict__ scale_ptr,
                                                            float* __restrict__ g_depth, int accumulate, int H,
                                                            int W, int HW) {
  const int b = blockIdx.y;
  float Ki[9], R[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Ki[i] = Kinv[b * 9 + i];
    R[i] = Rm[b * 9 + i];
  }
  const float scale = scale_ptr ? scale_ptr[0] : 1.0f;
  const int p0 = (blockIdx.x * 256 + threadIdx.x) * PX;
  if (p0 >= HW) return;
  const size_t base = (size_t)b * HW + p0;
  float g[3][PX];
  if (PLANAR) {
    const float* s = gP + (size_t)b * 3 * HW + p0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (PX == 4)
        *reinterpret_cast<float4*>(g[c]) = *reinterpret_cast<const float4*>(s + (size_t)c * HW);
      else
        g[c][0] = s[(size_t)c * HW];
    }
  } else {
    const float* s = gP + base * 3;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      g[0][i] = s[3 * i];
      g[1][i] = s[3 * i + 1];
      g[2][i] = s[3 * i + 2];
    }
  }
  const int y = p0 / W, x = p0 - y * W;
  float o[PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    float r0, r1, r2, q0, q1, q2;
    rowvec_mat3((float)(x + i), (float)y, 1.0f, Ki, r0, r1, r2);
    rowvec_mat3_T(g[0][i], g[1][i], g[2][i], R, q0, q1, q2);
    o[i] = scale * (q0 * r0 + q1 * r1 + q2 * r2);
  }
  if (PX == 4) {
    float4* dst = reinterpret_cast<float4*>(g_depth + base);
    float4 v = make_float4(o[0], o[1], o[2], o[3]);
    if (accumulate) {
      const float4 old = *dst;
      v.x += old.x;
      v.y += old.y;
      v.z += old.z;
      v.w += old.w;
    }
    *dst = v;
  } else {
    g_depth[base] = accumulate ? g_depth[base] + o[0] : o[0];
  }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace dvd

extern "C" {

int dvd_unproject_fwd(const float* depth, const float* R, const float* t, const float* K_inv, float* points,
                      int out_planar, int B, int H, int W, dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(depth && R && t && K_inv && points, "unproject_fwd: null pointer");
  DVD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, "unproject_fwd: bad shape B=%d H=%d W=%d", B, H, W);
  bytes_add(DVD_BYTES_GEOMETRY, 16.0 * (double)B * H * W);
  const int HW = H * W;
  const bool v4 = (W % 4 == 0) && aligned16(depth) && aligned16(points);
  const int px = v4 ? 4 : 1;
  dim3 grid((HW + 256 * px - 1) / (256 * px), B), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (v4 && out_planar)
    hipLaunchKernelGGL((unproject_fwd_kernel<4, true>), grid, block, 0, s, depth, R, t, K_inv, points, H, W, HW);
  else if (v4)
    hipLaunchKernelGGL((unproject_fwd_kernel<4, false>), grid, block, 0, s, depth, R, t, K_inv, points, H, W, HW);
  else if (out_planar)
    hipLaunchKernelGGL((unproject_fwd_kernel<1, true>), grid, block, 0, s, depth, R, t, K_inv, points, H, W, HW);
  else
    hipLaunchKernelGGL((unproject_fwd_kernel<1, false>), grid, block, 0, s, depth, R, t, K_inv, points, H, W, HW);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_unproject_bwd(const float* g_points, int planar, const float* R, const float* K_inv,
                      const float* scale_or_null, float* g_depth, int accumulate, int B, int H, int W,
                      dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(g_points && R && K_inv && g_depth, "unproject_bwd: null pointer");
  DVD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, "unproject_bwd: bad shape B=%d H=%d W=%d", B, H, W);
  bytes_add(DVD_BYTES_GEOMETRY, (accumulate ? 20.0 : 16.0) * (double)B * H * W);
  const int HW = H * W;
  const bool v4 = (W % 4 == 0) && aligned16(g_points) && aligned16(g_depth);
  const int px = v4 ? 4 : 1;
  dim3 grid((HW + 256 * px - 1) / (256 * px), B), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (v4 && planar)
    hipLaunchKernelGGL((unproject_bwd_kernel<4, true>), grid, block, 0, s, g_points, R, K_inv, scale_or_null,
                       g_depth, accumulate, H, W, HW);
  else if (v4)
    hipLaunchKernelGGL((unproject_bwd_kernel<4, false>), grid, block, 0, s, g_points, R, K_inv, scale_or_null,
                       g_depth, accumulate, H, W, HW);
  else if (planar)
    hipLaunchKernelGGL((unproject_bwd_kernel<1, true>), grid, block, 0, s, g_points, R, K_inv, scale_or_null,
                       g_depth, accumulate, H, W, HW);
  else
    hipLaunchKernelGGL((unproject_bwd_kernel<1, false>), grid, block, 0, s, g_points, R, K_inv, scale_or_null,
                       g_depth, accumulate, H, W, HW);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
