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

// Per-channel sums of an NCHW tensor (the bias gradient of a convolution) and, in the same read, its max|.| (the operand scale
// of that convolution's two gradient kernels): partial[s][c] = sum over the images of slice s and all pixels of x[n][c][p].
// Block (c, s); a thread adds its elements in ascending (image, pixel) order, the block adds its threads in a fixed tree, so
// the sums do not depend on scheduling.  HBM bound: one read of the tensor (rounds 1-5: an ATen sum AND dvd_amax, two reads).
template <bool VEC>
__global__ __launch_bounds__(256) void chansum_kernel(const float* __restrict__ x, int N, int C, int HW, int per_slice,
                                                      float* __restrict__ partial, float* __restrict__ amax_out) {
  __shared__ float s_part[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const int n0 = s * per_slice, n1 = min(N, n0 + per_slice);
  float sum = 0.0f, m = 0.0f;
  for (int n = n0; n < n1; ++n) {
    const float* p = x + ((size_t)n * C + c) * HW;
    if (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll 4
      for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
        const float4 v = p4[i];
        sum += (v.x + v.y) + (v.z + v.w);
        m = amax_acc(amax_acc(amax_acc(amax_acc(m, v.x), v.y), v.z), v.w);
      }
    } else {
#pragma unroll 4
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = p[i];
        sum += v;
        m = amax_acc(m, v);
      }
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)s * C + c] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  if (amax_out) wave_amax_to(m, amax_out);
}

// out[c] = sum over slices (ascending) of partial[s][c]
__global__ __launch_bounds__(256) void chansum_finish_kernel(const float* __restrict__ partial, int C, int S, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float t = partial[c];
  for (int s = 1; s < S; ++s) t += partial[(size_t)s * C + c];
  out[c] = t;
}

static int chansum_slices(int N, int C) {
  int S = 2048 / (C > 0 ? C : 1);             // about 8 blocks per CU
  if (S > N) S = N;
  return S < 1 ? 1 : S;
}

}  // namespace dvd

extern "C" {

size_t dvd_chansum_workspace_bytes(int N, int C) {
  if (N <= 0 || C <= 0) return 0;
  return (size_t)dvd::chansum_slices(N, C) * C * sizeof(float);
}

int dvd_chansum(const float* x, int N, int C, long long HW, float* out, float* amax_out, void* workspace, size_t workspace_bytes,
                dvd_stream_t stream) {
  DVD_REQUIRE(x && out && workspace && N > 0 && C > 0 && HW > 0 && HW < (1ll << 31), "chansum: bad arguments");
  DVD_REQUIRE(C <= 65535 * 256, "chansum: too many channels");
  const size_t need = dvd_chansum_workspace_bytes(N, C);
  if (workspace_bytes < need) {
    dvd::set_error("chansum: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  dvd::bytes_add(DVD_BYTES_AMAX, 4.0 * (double)N * C * (double)HW);
  const int S = dvd::chansum_slices(N, C), per = (N + S - 1) / S, S_used = (N + per - 1) / per;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  if ((HW & 3) == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(dvd::chansum_kernel<true>, dim3(C, S_used), dim3(256), 0, st, x, N, C, (int)HW, per, part, amax_out);
  else
    hipLaunchKernelGGL(dvd::chansum_kernel<false>, dim3(C, S_used), dim3(256), 0, st, x, N, C, (int)HW, per, part, amax_out);
  DVD_LAUNCH_OK();
  hipLaunchKernelGGL(dvd::chansum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, part, C, S_used, out);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

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
