// fp16 activation storage (BASELINE configs[4]: "fp16 activations with fp32 loss accumulation"): the pieces that sit at the
// fp32 / fp16 boundary of the depth network, and the loss-scale policy of the fp16 gradients.
//
// What it belongs to (reference, /root/reference): third_party/MiDaS.py:186-195,206-246 -- the network's last layers
// `ReLU -> Conv2d(32, 1, 1) -> ReLU -> 10000 / clamp(.)`.  With fp16 activations every tensor between the encoder's first stage
// and that 1x1 convolution is stored as _Float16 (csrc/xconv.hip IN16 / OUT16, csrc/xwgrad3.hip H16, the *_t entry points of the
// helper kernels); the image, the depth map, all parameters, parameter gradients, loss sums and optimiser state stay fp32.
//
//   * dvd_head1x1_{fwd,bwd}: the 1x1 convolution onto ONE channel (the depth head) -- the boundary itself.  Forward reads the
//     fp16 features and writes the fp32 head output; backward reads the fp32 output gradient (the depth gradient of the
//     warp+loss kernel, arbitrary magnitude), writes the fp16 feature gradient MULTIPLIED BY THE STEP'S LOSS SCALE S, and the
//     weight / bias gradients (fp32, from the unscaled fp32 gradient).  HBM bound: 2 C + 4 bytes per pixel forward.
//   * the loss-scale policy, entirely on the device (no host read-back anywhere in the step):
//       state[0] = S, state[1] = 1 / S          set by dvd_gscale_begin at the start of every depth-net backward pass: the power
//                                               of two that puts max|g_out| * max|w_head| at 2^target
//       state[2] = target exponent              (initially 4: gradients may grow by 2^11 between the head and the deepest layer
//                                               before the first step overflows; re-centred after every step)
//       state[3] = observed max |S g| of the fp16 gradient tensors of this step (the backward-data epilogues fold it in)
//       state[4] = skip flag of this step, state[5] = steps skipped so far
//       state[6] = forward monitor (round 5): max |y| every fp16-OUTPUT convolution epilogue and the depth head stored / read
//                  this step (NaN counted as +Inf); >= 65504 means an activation overflowed fp16 -> the step is skipped too.
//                  Every fp16 forward folds into it -- warm-up epochs, validation, inference included -- so a TRAINING step
//                  starts by clearing it (dvd_gscale_step_begin, round 6: a stale overflow from a pass that never reaches
//                  dvd_gscale_end used to skip the next real step)
//       state[8] = this step was skipped because an ACTIVATION overflowed, state[9] = steps skipped for that reason so far:
//                  the (flag, count) pair the scene-flow network's guarded Adam step reads -- its fp32 gradients are valid
//                  when only the loss scale of the depth net's fp16 GRADIENTS overflowed (state[4] covers both reasons);
//       state[10] = consecutive steps skipped for an activation overflow (no back-off can cure that one: the host reads it
//                  with the step's loss scalars and warns / raises);  state[7], [11..15] unused (16 floats in all)
//     dvd_gscale_end (once per step, before the optimiser): an observed maximum of 2^15.5 or more means fp16 range was
//     exceeded somewhere -> the step's depth-net update is SKIPPED (dvd_adam_step_guarded); in either case the target moves
//     by the whole number of octaves that puts the observed maximum at 2^13 (upwards by at most 4 per step).  Every kernel that produces a PARAMETER gradient from fp16 gradients
//     multiplies by state[1] (`out_scale`), so the flat gradient buffers always hold true gradients.
#include "dvd_io.h"
#include "dvd_split.h"      // wave_amax_to

namespace dvd {

constexpr int kHeadMaxC = 64;
constexpr float kGsOverflow = 46340.95f;      // 2^15.5
constexpr float kF16Max = 65504.0f;           // largest finite _Float16: an activation at or beyond it was stored as Inf

__global__ void gscale_init_kernel(float* __restrict__ st, float target) {
  if (threadIdx.x < 16) st[threadIdx.x] = threadIdx.x == 0 || threadIdx.x == 1 ? 1.0f : (threadIdx.x == 2 ? target : 0.0f);
}

// First launch of a training step: the forward monitor only means something for the forward passes of THIS step.
__global__ void gscale_step_begin_kernel(float* __restrict__ st) {
  if (threadIdx.x == 0) st[6] = 0.0f;
}

// S = 2^(target - ceil(log2(max|g| * max|w|))); 1 when the gradient is zero / not finite (nothing to scale, or nothing to save)
__global__ __launch_bounds__(64) void gscale_begin_kernel(float* __restrict__ st, const float* __restrict__ g_amax,
                                                          const float* __restrict__ w, int nw) {
  float wm = 0.0f;
  for (int i = threadIdx.x; i < nw; i += 64) wm = fmaxf(wm, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_down(wm, off, 64));
  if (threadIdx.x == 0) {
    const float m = g_amax[0] * (nw > 0 ? wm : 1.0f);
    float S = 1.0f;
    if (m > 0.0f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);                                   // m = f * 2^e, f in [0.5, 1)  ->  m <= 2^e
      int se = (int)st[2] - e;
      se = se < -120 ? -120 : (se > 120 ? 120 : se);
      S = ldexpf(1.0f, se);
    }
    st[0] = S;
    st[1] = 1.0f / S;
  }
}

// Once per step.  obs = the largest |S g| any monitored fp16 gradient tensor was ABOUT to store (computed in fp32 before the
// conversion, so it is finite and exact even when the fp16 value became Inf).  The target is re-centred so that the same
// network state would put obs at 2^13 next step (three octaves below fp16's maximum): gradient growth between the head and
// the deepest layer is a property of the weights and changes slowly.
__global__ void gscale_end_kernel(float* __restrict__ st) {
  if (threadIdx.x != 0) return;
  const float obs = st[3];
  const float fwd = st[6];             // forward monitor: max |activation| the fp16-output epilogues and the depth head saw
  float target = st[2];
  if (!(fwd < kF16Max)) {              // an fp16 ACTIVATION left the format's range (or was NaN): the parameter gradients
    st[4] = 1.0f;                      // of this step are not trustworthy -- skip it; the loss scale is not at fault
    st[5] += 1.0f;
    st[8] = 1.0f;                      // ... and the scene-flow network's gradient (through the depth map) is not either
    st[9] += 1.0f;
    st[10] += 1.0f;
  } else if (!(obs < kGsOverflow)) {   // fp16 range exceeded (or Inf / NaN): skip this step's update, back off
    st[4] = 1.0f;
    st[5] += 1.0f;
    st[8] = 0.0f;                      // the depth maps were finite: the scene-flow network's fp32 gradients stand
    st[10] = 0.0f;
    target -= (obs < 3.0e38f) ? ceilf(log2f(obs) - 13.0f) : 8.0f;
  } else {
    st[4] = 0.0f;
    st[8] = 0.0f;
    st[10] = 0.0f;
    if (obs > 0.0f) {
      float d = rintf(13.0f - log2f(obs));     // (whole octaves; within half an octave of 2^13 nothing moves)
      d = d > 4.0f ? 4.0f : d;                 // rise by at most 4 octaves per step
      target += d;
    }
  }
  st[2] = fminf(fmaxf(target, -24.0f), 14.0f);
  st[3] = 0.0f;
  st[6] = 0.0f;
}

// y[n][p] = bias + sum_c w[c] * act(x[n][c][p]); a thread owns 4 consecutive pixels (HW % 4 == 0)
template <class T>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int C, int HW4,
                                                          long long total, int relu_in, float* fmax) {
  __shared__ float sw[kHeadMaxC];
  if (threadIdx.x < C) sw[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const float b = bias ? bias[0] : 0.0f;
  float ym = 0.0f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW4;
    const int q = (int)(i - n * HW4);
    const T* xp = x + ((size_t)n * C * HW4 + q) * 4;
    float4 acc = make_float4(b, b, b, b);
    for (int c = 0; c < C; ++c) {
      float4 v = ld4(xp + (size_t)c * HW4 * 4);
      if (relu_in) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      const float wc = sw[c];
      acc.x = __builtin_fmaf(wc, v.x, acc.x);
      acc.y = __builtin_fmaf(wc, v.y, acc.y);
      acc.z = __builtin_fmaf(wc, v.z, acc.z);
      acc.w = __builtin_fmaf(wc, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(y + i * 4) = acc;
    if (fmax) ym = amax_acc(amax_acc(amax_acc(amax_acc(ym, acc.x), acc.y), acc.z), acc.w);
  }
  if (fmax) wave_amax_to(ym, fmax);       // forward monitor of the fp16 overflow guard: an Inf / NaN feature shows up here
}

// gx[n][c][p] = S * w[c] * gy[n][p] * [x > 0];  partial[block][c] = sum gy * act(x[c]),  partial[block][C] = sum gy
// (CC: compile-time channel bound, 32 or 64 -- the per-channel sums must stay in registers, so every index is static)
template <class T, int CC>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ gy, const float* __restrict__ gstate,
                                                          T* __restrict__ gx, float* __restrict__ partial, float* gmax, int C,
                                                          int HW4, long long total, int relu_in) {
  __shared__ float sw[kHeadMaxC];
  __shared__ float red[4][CC + 1];
  const float S = gstate ? gstate[0] : 1.0f;
  if (threadIdx.x < C) sw[threadIdx.x] = w[threadIdx.x] * S;
  __syncthreads();
  float sums[CC + 1];
#pragma unroll
  for (int c = 0; c <= CC; ++c) sums[c] = 0.0f;
  float gm = 0.0f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW4;
    const int q = (int)(i - n * HW4);
    const size_t base = ((size_t)n * C * HW4 + q) * 4;
    const float4 g = *reinterpret_cast<const float4*>(gy + i * 4);
    sums[CC] += (g.x + g.y) + (g.z + g.w);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      if (c >= C) continue;                              // uniform
      float4 v = ld4(x + base + (size_t)c * HW4 * 4);
      const float wc = sw[c];
      float4 o = make_float4(wc * g.x, wc * g.y, wc * g.z, wc * g.w);
      if (relu_in) {
        o.x = v.x > 0.f ? o.x : 0.f;
        o.y = v.y > 0.f ? o.y : 0.f;
        o.z = v.z > 0.f ? o.z : 0.f;
        o.w = v.w > 0.f ? o.w : 0.f;
        v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      }
      st4(gx + base + (size_t)c * HW4 * 4, o);
      gm = amax_acc(amax_acc(amax_acc(amax_acc(gm, o.x), o.y), o.z), o.w);
      sums[c] += __builtin_fmaf(g.x, v.x, __builtin_fmaf(g.y, v.y, __builtin_fmaf(g.z, v.z, g.w * v.w)));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c <= CC; ++c) {
    if (c < C || c == CC) {
      const float v = wave_sum(sums[c]);
      if (lane == 0) red[wave][c] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x <= C) {
    const int c = threadIdx.x < C ? threadIdx.x : CC;
    partial[(size_t)blockIdx.x * (C + 1) + threadIdx.x] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  }
  if (gmax) {                                            // observed max of the scaled fp16 gradient (loss-scale policy)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_down(gm, off, 64));
    if (lane == 0 && gm > 0.0f) {
      unsigned* p = reinterpret_cast<unsigned*>(gmax);
      if (__float_as_uint(gm) > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, __float_as_uint(gm));
    }
  }
}

// gw[c] = sum over the blocks' partials, gb likewise, in a fixed order: block c of this launch, thread t adds partials t, t + 256,
// ... in double, then a fixed tree over the 256 threads.  (Rounds 4-5: ONE thread per channel walked all 2 048 partials -- 2 048
// dependent loads, 0.70 ms with the GPU idle, twice per step.)
__global__ __launch_bounds__(256) void head1x1_reduce_kernel(const float* __restrict__ partial, int blocks, int C,
                                                             float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ double s_sum[256];
  const int c = blockIdx.x;                       // 0 .. C (C: the bias)
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += partial[(size_t)b * (C + 1) + c];
  s_sum[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (c < C) {
      if (gw) gw[c] = (float)s_sum[0];
    } else if (gb) {
      gb[0] = (float)s_sum[0];
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void cast_scale_kernel(const T* __restrict__ in, float* __restrict__ out, long long n4,
                                                         const float* __restrict__ scale) {
  const float s = scale ? scale[0] : 1.0f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = ld4(in + i * 4);
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
  }
}

static int blocks_for(long long items) {
  long long b = (items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dvd

extern "C" {

int dvd_gscale_init(float* state, float target_exponent, dvd_stream_t stream) {
  DVD_REQUIRE(state, "gscale_init: null pointer");
  hipLaunchKernelGGL(dvd::gscale_init_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, target_exponent);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_gscale_step_begin(float* state, dvd_stream_t stream) {
  DVD_REQUIRE(state, "gscale_step_begin: null pointer");
  hipLaunchKernelGGL(dvd::gscale_step_begin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_gscale_begin(float* state, const float* g_amax, const float* w, int n_w, dvd_stream_t stream) {
  DVD_REQUIRE(state && g_amax && (w || n_w == 0), "gscale_begin: null pointer");
  hipLaunchKernelGGL(dvd::gscale_begin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, g_amax, w, n_w);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_gscale_end(float* state, dvd_stream_t stream) {
  DVD_REQUIRE(state, "gscale_end: null pointer");
  hipLaunchKernelGGL(dvd::gscale_end_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_head1x1_fwd(const void* x, int f16, const float* w, const float* bias, float* y, float* fwd_amax, int N, int C, int HW,
                    int relu_in, dvd_stream_t stream) {
  DVD_REQUIRE(x && w && y, "head1x1 fwd: null pointer");
  DVD_REQUIRE(N > 0 && C > 0 && C <= dvd::kHeadMaxC && HW > 0 && (HW & 3) == 0, "head1x1 fwd: bad shape N=%d C=%d HW=%d (C <= 64, HW %% 4 == 0)", N, C, HW);
  const long long total = (long long)N * (HW / 4);
  dvd::bytes_add(DVD_BYTES_ELEMENTWISE, (double)N * HW * ((f16 ? 2.0 : 4.0) * C + 4.0));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::head1x1_fwd_kernel<T>, dim3(dvd::blocks_for(total)), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), w, bias, y, C, HW / 4, total,
                                         relu_in, fwd_amax));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

size_t dvd_head1x1_bwd_workspace_bytes(int C) { return (size_t)2048 * (C + 1) * sizeof(float); }

int dvd_head1x1_bwd(const void* x, int f16, const float* w, const float* gy, const float* gscale_state, void* gx, float* gw,
                    float* gb, void* workspace, size_t workspace_bytes, int N, int C, int HW, int relu_in, dvd_stream_t stream) {
  DVD_REQUIRE(x && w && gy && gx && workspace, "head1x1 bwd: null pointer");
  DVD_REQUIRE(N > 0 && C > 0 && C <= dvd::kHeadMaxC && HW > 0 && (HW & 3) == 0, "head1x1 bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
  if (workspace_bytes < dvd_head1x1_bwd_workspace_bytes(C)) {
    dvd::set_error("head1x1 bwd: workspace too small");
    return DVD_ENOSPC;
  }
  const long long total = (long long)N * (HW / 4);
  dvd::bytes_add(DVD_BYTES_ELEMENTWISE, (double)N * HW * ((f16 ? 2.0 : 4.0) * 2 * C + 4.0));
  const int blocks = dvd::blocks_for(total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* gmax = gscale_state ? const_cast<float*>(gscale_state) + 3 : nullptr;
  if (C <= 32)
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::head1x1_bwd_kernel<T, 32>), dim3(blocks), dim3(256), 0, s, static_cast<const T*>(x),
                                           w, gy, gscale_state, static_cast<T*>(gx), static_cast<float*>(workspace), gmax, C,
                                           HW / 4, total, relu_in));
  else
    DVD_DISPATCH_T(f16, hipLaunchKernelGGL((dvd::head1x1_bwd_kernel<T, 64>), dim3(blocks), dim3(256), 0, s, static_cast<const T*>(x),
                                           w, gy, gscale_state, static_cast<T*>(gx), static_cast<float*>(workspace), gmax, C,
                                           HW / 4, total, relu_in));
  DVD_LAUNCH_OK();
  if (gw || gb) {
    hipLaunchKernelGGL(dvd::head1x1_reduce_kernel, dim3(C + 1), dim3(256), 0, s, static_cast<const float*>(workspace), blocks, C, gw, gb);
    DVD_LAUNCH_OK();
  }
  return DVD_OK;
}

int dvd_cast_scale_f32(const void* in, int f16, float* out, long long n, const float* scale, dvd_stream_t stream) {
  DVD_REQUIRE(in && out && n > 0 && (n & 3) == 0, "cast_scale: bad arguments (n %% 4 == 0)");
  dvd::bytes_add(DVD_BYTES_ELEMENTWISE, (double)n * ((f16 ? 2.0 : 4.0) + 4.0));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::cast_scale_kernel<T>, dim3(dvd::blocks_for(n >> 2)), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(in), out, n >> 2, scale));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
