// Micro-benchmark: throughput of LDS atomic adds on gfx950 (f32 vs u32 vs u64),
// lane-linear and pseudo-random addresses.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = 8192;  // slots
  float* f = reinterpret_cast<float*>(smem);
  unsigned* u = reinterpret_cast<unsigned*>(smem);
  unsigned long long* q = reinterpret_cast<unsigned long long*>(smem);
  for (int i = threadIdx.x; i < N * 2; i += 256) u[i] = 0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int idx = RANDOM ? (s >> 8) % N : ((threadIdx.x + it * 64) % N);
    if (MODE == 0) atomicAdd(f + idx, 1.0f);
    if (MODE == 1) atomicAdd(u + idx, 3u);
    if (MODE == 2) atomicAdd(q + idx, 3ull);
    if (MODE == 3) f[idx] += 1.0f;  // plain RMW (racy) for reference
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[0] + (float)u[1];
}

template <int MODE, bool RANDOM>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE, RANDOM><<<blocks, 256, 65536>>>(16, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE, RANDOM><<<blocks, 256, 65536>>>(iters, out);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double lane_ops = (double)blocks * 256 * iters;
  printf("%-22s %8.3f ms  %7.1f G lane-ops/s  (%.2f lane-ops/clk/CU @2.1GHz)\n", name, ms, lane_ops / ms / 1e6,
         lane_ops / ms / 1e6 * 1e9 / 2.1e9 / 256);
  hipFree(out);
}

int main() {
  run<0, false>("ds_add_f32 linear");
  run<0, true>("ds_add_f32 random");
  run<1, false>("ds_add_u32 linear");
  run<1, true>("ds_add_u32 random");
  run<2, false>("ds_add_u64 linear");
  run<2, true>("ds_add_u64 random");
  run<3, true>("plain rmw random");
  return 0;
}
