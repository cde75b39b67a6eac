// The matrix arithmetic shared by the convolution / weight-gradient / MLP kernels of this package (round 3).
//
// gfx950 has no fast path for fp32 matrix operands (v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate, 1/16 of the 16-bit
// matrix rate), and the 16-bit matrix pipe is POWER limited, not issue limited: with random operands the chip clocks
// 1.6-1.7 GHz at 72-77 % pipe utilisation, zero-filled operands run the same instruction stream 35 % faster
// (profiles/r03_xconv_power_limit.txt).  What a kernel can change is the number of MFMAs per fp32 product.
//
//   round 2: three bf16 terms per operand (exact 24-bit split), the six largest of nine partial products;
//   round 3: TWO fp16 terms per operand of the operand SCALED by a power of two,
//                xs = x * 2^e,  h = fp16(xs),  l = fp16(xs - h)           (round to nearest even; xs - h is exact in fp32)
//            and THREE partial products  l*h' + h*l' + h*h'  on v_mfma_f32_32x32x16_f16 with fp32 accumulation, the result
//            multiplied by 2^-(e + e').  fp16 carries 11 significant bits, h + l carries 22: the dropped l*l' term and the
//            rounding of l are below 2^-21 |x w|, the class of a single fp32 rounding (2^-24) times a few -- and far below
//            the difference between two fp32 summation orders over K >= 256 terms, which is what "fp32" means for a
//            convolution.  Measured against float64: tests/test_06_xconv_gpu.py (<= 4e-6 of max|y|, the same bound the
//            six-product arithmetic met), emulated in numpy: tests/test_split_bf16_cpu.py.
//   The scale puts the tensor's largest magnitude in [2^13, 2^14): no overflow (fp16 max 65504), and elements down to
//   2^-17 of the largest keep all 22 bits (below that the absolute error stays <= 2^-25 * 2^-e, i.e. 2^-38 of the largest).
//   Every tensor that feeds a matrix kernel therefore carries max|x| (or an upper bound) in a device scalar: written by the
//   producing kernel's epilogue or by dvd_amax, read by the consumer -- never by the host.
#pragma once
#include "dvd_common.h"

namespace dvd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

// 2^e with max|x| * 2^e in [2^13, 2^14); 1 for max|x| = 0, Inf or NaN (those propagate through the products).
__device__ __forceinline__ float pow2_scale(float amax) {
  const unsigned u = __float_as_uint(amax);
  const int e = (int)((u >> 23) & 0xffu);
  if (!(amax > 0.0f) || e == 0xff) return 1.0f;
  int se = 267 - e;                            // biased exponent of 2^(13 - (e - 127))
  se = se < 1 ? 1 : (se > 254 ? 254 : se);
  return __uint_as_float((unsigned)se << 23);
}

// (a, b), already scaled -> h and l dwords of two fp16 each: a in the low half, b in the high half
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& h, unsigned& l) {
  const f32x2s v = {a, b};
  const f16x2 hb = __builtin_convertvector(v, f16x2);
  const f32x2s r = v - __builtin_convertvector(hb, f32x2s);
  const f16x2 lb = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(unsigned, hb);
  l = __builtin_bit_cast(unsigned, lb);
}

__device__ __forceinline__ void split8_f16(const float v[8], float s, uint4& h, uint4& l) {
  split_pair_f16(v[0] * s, v[1] * s, h.x, l.x);
  split_pair_f16(v[2] * s, v[3] * s, h.y, l.y);
  split_pair_f16(v[4] * s, v[5] * s, h.z, l.z);
  split_pair_f16(v[6] * s, v[7] * s, h.w, l.w);
}

// A pointer the compiler can see is wave-uniform (block indices divided by run-time values pass through VGPRs): buffer
// resources must sit in SGPRs, a resource of unknown uniformity costs a waterfall loop around every load.
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// max over the wave, then AT MOST one atomic per wave: |x| as an unsigned integer is monotonic in |x|.  The scalar is read
// first and the atomic only issued when it would raise it: tens of thousands of waves finishing together otherwise
// serialise on the one address (~88 atomics per microsecond: a BatchNorm mask pass of 49 000 blocks took 1.4 ms instead of
// 0.16).  The unordered read can only under-estimate the current value, and atomicMax resolves the race.
__device__ __forceinline__ void wave_amax_to(float m, float* out) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0 && m > 0.0f) {
    unsigned* p = reinterpret_cast<unsigned*>(out);
    if (__float_as_uint(m) > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, __float_as_uint(m));
  }
}

}  // namespace dvd
