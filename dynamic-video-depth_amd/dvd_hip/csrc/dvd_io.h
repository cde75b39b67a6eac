// Element access shared by the HBM-bound kernels that exist for both activation storages: fp32 (BASELINE configs[1]-[3]) and
// _Float16 (configs[4]: fp16 activations, fp32 arithmetic inside the kernel, fp32 parameters and sums).  The arithmetic of a
// kernel is written once on float values; T only decides how a value is read from / written to HBM.
#pragma once
#include "dvd_common.h"

namespace dvd {

typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <class T>
__device__ __forceinline__ float ldf(const T* p) {
  return (float)*p;
}
template <class T>
__device__ __forceinline__ void stf(T* p, float v) {
  *p = (T)v;
}
// four consecutive elements, p aligned to four elements
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const _Float16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const f32x2_t a = __builtin_convertvector(__builtin_bit_cast(h16x2_t, u.x), f32x2_t);
  const f32x2_t b = __builtin_convertvector(__builtin_bit_cast(h16x2_t, u.y), f32x2_t);
  return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(_Float16* p, const float4 v) {
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, h16x2_t)),
                                            __builtin_bit_cast(unsigned, __builtin_convertvector(b, h16x2_t)));
}
// two consecutive elements, p aligned to two elements
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const _Float16* p) {
  const f32x2_t a = __builtin_convertvector(*reinterpret_cast<const h16x2_t*>(p), f32x2_t);
  return make_float2(a.x, a.y);
}

// Launch `KERNEL<float>` or `KERNEL<_Float16>` according to the run-time storage flag of the *_t entry points.
#define DVD_DISPATCH_T(f16, ...)            \
  do {                                      \
    if (f16) {                              \
      typedef _Float16 T;                   \
      __VA_ARGS__;                          \
    } else {                                \
      typedef float T;                      \
      __VA_ARGS__;                          \
    }                                       \
  } while (0)

}  // namespace dvd
