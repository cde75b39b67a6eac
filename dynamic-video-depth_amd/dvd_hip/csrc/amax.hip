// max|x| of a tensor into a device scalar: the power-of-two operand scale of the matrix kernels (csrc/dvd_split.h) is derived
// from it on the device.  HBM bound: one read of the tensor; producers that can (the convolution epilogues) deliver the
// value themselves, this kernel serves the rest (network inputs, up-sampled / pooled / summed tensors, output gradients).
#include "dvd_split.h"

namespace dvd {

// `head` leading elements bring x to a 16-byte boundary (a contiguous batch slice of odd-sized planes is only 4-byte aligned)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int head, long long n, float* __restrict__ out) {
  float m = 0.0f;
  if (blockIdx.x == 0 && threadIdx.x < head) m = amax_acc(m, x[threadIdx.x]);
  x += head;
  n -= head;
  const long long nv = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = amax_acc(amax_acc(amax_acc(amax_acc(m, v.x), v.y), v.z), v.w);       // (NaN -> +Inf: see amax_acc)
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) m = amax_acc(m, x[(nv << 2) + threadIdx.x]);
  wave_amax_to(m, out);
}

}  // namespace dvd

extern "C" {

// out[0] = max(out[0], max|x|): the caller zeroes `out` (or passes a running bound).  x: any 4-byte aligned pointer.
int dvd_amax(const float* x, long long n, float* out, dvd_stream_t stream) {
  DVD_REQUIRE(x && out && n > 0, "amax: bad arguments");
  DVD_REQUIRE(((uintptr_t)x & 3) == 0, "amax: tensor must be 4-byte aligned");
  dvd::bytes_add(DVD_BYTES_AMAX, 4.0 * (double)n);
  long long head = (long long)(((16 - ((uintptr_t)x & 15)) & 15) >> 2);
  if (head > n) head = n;
  const long long nv = (n - head + 3) / 4;
  long long blocks = (nv + 255) / 256;
  if (blocks > 2048) blocks = 2048;            // 8 blocks per CU, grid-stride
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dvd::amax_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, (int)head, n,
                     out);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
