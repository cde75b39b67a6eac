// Internal helpers shared by the gfx950 kernels of libdvd_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "dvd_hip.h"

namespace dvd {

void set_error(const char* fmt, ...);
int zero_words(void* p, int n_words, hipStream_t stream);   // device-side clear by a kernel (core.hip)
void flops_add(int cls, double flops);
void bytes_add(int cls, double bytes);                       // algorithmic bytes of the memory-bound helper kernels per class (core.hip)                       // algorithmic-work accounting per kernel class (core.hip)

#define DVD_REQUIRE(cond, ...)             \
  do {                                     \
    if (!(cond)) {                         \
      ::dvd::set_error(__VA_ARGS__);       \
      return DVD_EINVAL;                   \
    }                                      \
  } while (0)

#define DVD_HIP_OK(expr)                                                                   \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      ::dvd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                       __LINE__);                                                          \
      return DVD_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

#define DVD_LAUNCH_OK()                                                                    \
  do {                                                                                     \
    hipError_t e__ = hipGetLastError();                                                    \
    if (e__ != hipSuccess) {                                                               \
      ::dvd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),         \
                       __FILE__, __LINE__);                                                \
      return DVD_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// Row-vector times 3x3 (row-major M), with the SAME rounding sequence as the
// reference's torch.matmul on CPU for [..,1,3]@[3,3]: ((v0*M0j) + v1*M1j) + v2*M2j,
// separate multiplies and adds (this translation unit is built with
// -ffp-contract=off so nothing is fused behind our back).
__device__ __forceinline__ void rowvec_mat3(const float v0, const float v1, const float v2,
                                            const float* __restrict__ M, float& o0, float& o1,
                                            float& o2) {
  o0 = (v0 * M[0] + v1 * M[3]) + v2 * M[6];
  o1 = (v0 * M[1] + v1 * M[4]) + v2 * M[7];
  o2 = (v0 * M[2] + v1 * M[5]) + v2 * M[8];
}
// g_v_i = sum_j g_o_j * M[i][j]   (backward of rowvec_mat3 w.r.t. v)
__device__ __forceinline__ void rowvec_mat3_T(const float g0, const float g1, const float g2,
                                              const float* __restrict__ M, float& o0, float& o1,
                                              float& o2) {
  o0 = g0 * M[0] + g1 * M[1] + g2 * M[2];
  o1 = g0 * M[3] + g1 * M[4] + g2 * M[5];
  o2 = g0 * M[6] + g1 * M[7] + g2 * M[8];
}

// max|.| accumulation that keeps a NaN: fmaxf (and an unsigned atomicMax of the bit pattern's magnitude order) DROP a NaN
// operand, so every observed-maximum reduction the fp16 overflow guard and the operand scales look at records a NaN as
// +Inf, which all later fmaxf / atomicMax steps keep (a NaN gradient used to leave the skip flag at 0).
__device__ __forceinline__ float amax_acc(float m, float v) {
  const float a = fabsf(v);
  return fmaxf(m, a == a ? a : __builtin_inff());
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

}  // namespace dvd
