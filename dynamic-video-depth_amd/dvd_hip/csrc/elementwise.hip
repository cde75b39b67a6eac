// Small HBM-bound helpers of the training step (gfx950): gradient combination with a
// device-side scalar, the acceleration regulariser's elementwise part, and fused Adam.
//
// Reference counterparts (/root/reference):
//   models/scene_flow_motion_field.py:326-344   _opt_reg: |sf1 - sf0| mean and its gradient
//   models/scene_flow_motion_field.py:113-115,212-213 + torch.optim.Adam  (betas 0.5/0.9,
//       options/options_train.py:84-87): one fused pass over a flat parameter buffer
//   the 1/(sum(mask)+1e-8) normaliser of _calc_loss (:297-306) applied late, as a device
//       scalar, so that no host synchronisation sits between the loss and the backward.
// All kernels: 16-byte accesses, grid-stride, <= 2048 blocks.

#include "dvd_common.h"

namespace dvd {

static int grid_for(long long n4) {
  long long b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

__global__ __launch_bounds__(256) void scale_add_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                        float sa, const float* __restrict__ sa_ptr,
                                                        const float* __restrict__ b, long long n) {
  const float s = sa * (sa_ptr ? sa_ptr[0] : 1.0f);
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(a)[i];
    v.x *= s;
    v.y *= s;
    v.z *= s;
    v.w *= s;
    if (b) {
      const float4 w = reinterpret_cast<const float4*>(b)[i];
      v.x += w.x;
      v.y += w.y;
      v.z += w.z;
      v.w += w.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    out[i] = s * a[i] + (b ? b[i] : 0.0f);
  }
}

// The tail of the MiDaS depth head, `10000 / clamp(relu(v), min=1e-2)` (third_party/MiDaS.py:192-195,240-242), in one pass each
// way instead of ATen's clamp_min / clamp / reciprocal / mul kernels and their four backward kernels.  Forward: torch evaluates
// `10000 / t` as reciprocal(t) * 10000 -- two roundings, an IEEE reciprocal and a multiply -- and so does this (bit-identical).
// Backward: autograd's -(g * 10000) * r * r with r = 1 / v where the clamp passes the value on (v >= 1e-2; the ReLU in front of
// it passes every such v), 0 elsewhere.
__device__ __forceinline__ float depth_tail_value(float x) {
  const float r = 1.0f / fmaxf(fmaxf(x, 0.0f), 1e-2f);      // IEEE division (this file is not built with fast-math)
  return r * 10000.0f;
}
__global__ __launch_bounds__(256) void depth_tail_fwd_kernel(const float* __restrict__ v, float* __restrict__ out, long long n) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(v)[i];
    float4 o;
    o.x = depth_tail_value(x.x);
    o.y = depth_tail_value(x.y);
    o.z = depth_tail_value(x.z);
    o.w = depth_tail_value(x.w);
    reinterpret_cast<float4*>(out)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    out[i] = depth_tail_value(v[i]);
  }
}
__device__ __forceinline__ float depth_tail_grad(float x, float g) {
  const float r = 1.0f / x;
  return x >= 1e-2f ? -(g * 10000.0f) * r * r : 0.0f;
}
__global__ __launch_bounds__(256) void depth_tail_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             float* __restrict__ gv, long long n) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(v)[i], y = reinterpret_cast<const float4*>(g)[i];
    reinterpret_cast<float4*>(gv)[i] =
        make_float4(depth_tail_grad(x.x, y.x), depth_tail_grad(x.y, y.y), depth_tail_grad(x.z, y.z), depth_tail_grad(x.w, y.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    gv[i] = depth_tail_grad(v[i], g[i]);
  }
}

// out[b, c, p] = a[b, c, p] * m[b, p]   (planar [B,C,HW] times a per-pixel mask [B,HW]; out may alias a)
__global__ __launch_bounds__(256) void mul_mask_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                       const float* __restrict__ m, int C, long long HW, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long plane = i / HW, p = i - plane * HW;
    out[i] = a[i] * m[(plane / C) * HW + p];
  }
}

// g1 = coef * sign(sf1 - sf0);  partial[block] = sum |sf1 - sf0|
__global__ __launch_bounds__(256) void acc_reg_kernel(const float* __restrict__ sf0, const float* __restrict__ sf1,
                                                      float coef, float* __restrict__ g1, float* __restrict__ partial,
                                                      long long n) {
  float acc = 0.0f;
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(sf0)[i], b = reinterpret_cast<const float4*>(sf1)[i];
    const float d0 = b.x - a.x, d1 = b.y - a.y, d2 = b.z - a.z, d3 = b.w - a.w;
    acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
    float4 g;
    g.x = d0 > 0.f ? coef : (d0 < 0.f ? -coef : 0.f);
    g.y = d1 > 0.f ? coef : (d1 < 0.f ? -coef : 0.f);
    g.z = d2 > 0.f ? coef : (d2 < 0.f ? -coef : 0.f);
    g.w = d3 > 0.f ? coef : (d3 < 0.f ? -coef : 0.f);
    reinterpret_cast<float4*>(g1)[i] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float d = sf1[i] - sf0[i];
    acc += fabsf(d);
    g1[i] = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
  }
  __shared__ float red[4];
  const float v = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)sh[0];
}

// torch.optim.Adam (no amsgrad, no weight decay), same operation order:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// with g = s * g1 + g2 (s = sa * *sa_ptr), so the loss normaliser never needs its own pass.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g1, float sa,
                                                   const float* __restrict__ sa_ptr, const float* __restrict__ g2,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float b1, float b2, float eps, float step_size, float inv_sqrt_bc2,
                                                   const float* __restrict__ skip, int step, float lr) {
  if (skip && skip[0] != 0.0f) return;       // fp16 gradient overflow in this step (csrc/a16.hip): no update, moments untouched
  // Steps skipped EARLIER (skip[1], the loss-scale state's count) did not touch the moments, so Adam's bias correction must
  // not count them either: the effective step is step - skipped, like torch.amp's GradScaler, which does not advance the
  // optimiser on a skipped step.  The host's double-precision factors are used unchanged when nothing
  // was skipped; otherwise one thread per block redoes that arithmetic.
  __shared__ float bc[2];
  const int skipped = skip ? (int)skip[1] : 0;
  if (skipped > 0) {
    if (threadIdx.x == 0) {
      const double se = (double)(step - skipped > 1 ? step - skipped : 1);
      bc[0] = (float)((double)lr / (1.0 - pow((double)b1, se)));
      bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, se)));
    }
    __syncthreads();
    step_size = bc[0];
    inv_sqrt_bc2 = bc[1];
  }
  const float s = sa * (sa_ptr ? sa_ptr[0] : 1.0f);
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 g = reinterpret_cast<const float4*>(g1)[i];
    g.x *= s;
    g.y *= s;
    g.z *= s;
    g.w *= s;
    if (g2) {
      const float4 w = reinterpret_cast<const float4*>(g2)[i];
      g.x += w.x;
      g.y += w.y;
      g.z += w.z;
      g.w += w.w;
    }
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i], pp = reinterpret_cast<float4*>(p)[i];
#define DVD_ADAM(C)                                             \
  mm.C = mm.C * b1 + (1.0f - b1) * g.C;                         \
  vv.C = vv.C * b2 + (1.0f - b2) * g.C * g.C;                   \
  pp.C = pp.C - step_size * (mm.C / (sqrtf(vv.C) * inv_sqrt_bc2 + eps));
    DVD_ADAM(x)
    DVD_ADAM(y)
    DVD_ADAM(z)
    DVD_ADAM(w)
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float g = s * g1[i] + (g2 ? g2[i] : 0.0f);
    const float mm = m[i] * b1 + (1.0f - b1) * g, vv = v[i] * b2 + (1.0f - b2) * g * g;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
#undef DVD_ADAM
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace dvd

extern "C" {

int dvd_mul_mask(float* out, const float* a, const float* mask, int B, int C, long long HW, dvd_stream_t stream) {
  DVD_REQUIRE(out && a && mask && B > 0 && C > 0 && HW > 0, "mul_mask: bad arguments");
  const long long n = (long long)B * C * HW;
  dvd::bytes_add(DVD_BYTES_ELEMENTWISE, 8.0 * (double)n + 4.0 * (double)B * HW);
  hipLaunchKernelGGL(dvd::mul_mask_kernel, dim3(dvd::grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), out,
                     a, mask, C, HW, n);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_scale_add(float* out, const float* a, float scale, const float* scale_ptr, const float* b, long long n,
                  dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(out && a && n > 0, "scale_add: null pointer / size");
  DVD_REQUIRE(al16(out) && al16(a) && al16(b), "scale_add: pointers must be 16-byte aligned");
  bytes_add(DVD_BYTES_ELEMENTWISE, 4.0 * (double)n * (b ? 3 : 2));
  hipLaunchKernelGGL(scale_add_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, static_cast<hipStream_t>(stream), out, a,
                     scale, scale_ptr, b, n);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_depth_tail_fwd(const float* v, float* depth, long long n, dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(v && depth && n > 0, "depth_tail_fwd: null pointer / size");
  DVD_REQUIRE(al16(v) && al16(depth), "depth_tail_fwd: pointers must be 16-byte aligned");
  bytes_add(DVD_BYTES_ELEMENTWISE, 8.0 * (double)n);
  hipLaunchKernelGGL(depth_tail_fwd_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, static_cast<hipStream_t>(stream), v, depth, n);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_depth_tail_bwd(const float* v, const float* g_depth, float* g_v, long long n, dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(v && g_depth && g_v && n > 0, "depth_tail_bwd: null pointer / size");
  DVD_REQUIRE(al16(v) && al16(g_depth) && al16(g_v), "depth_tail_bwd: pointers must be 16-byte aligned");
  bytes_add(DVD_BYTES_ELEMENTWISE, 12.0 * (double)n);
  hipLaunchKernelGGL(depth_tail_bwd_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, static_cast<hipStream_t>(stream), v, g_depth, g_v,
                     n);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

size_t dvd_acc_reg_workspace_bytes(void) { return 2048 * sizeof(float); }

int dvd_acc_reg(const float* sf0, const float* sf1, float coef, float* g_sf1, void* workspace, float* abs_sum,
                int accumulate, long long n, dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(sf0 && sf1 && g_sf1 && workspace && abs_sum && n > 0, "acc_reg: null pointer / size");
  DVD_REQUIRE(al16(sf0) && al16(sf1) && al16(g_sf1), "acc_reg: pointers must be 16-byte aligned");
  bytes_add(DVD_BYTES_ELEMENTWISE, 12.0 * (double)n);
  const int grid = grid_for(n >> 2);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(acc_reg_kernel, dim3(grid), dim3(256), 0, s, sf0, sf1, coef, g_sf1,
                     static_cast<float*>(workspace), n);
  DVD_LAUNCH_OK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), grid, abs_sum,
                     accumulate);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_adam_step(float* param, const float* grad1, float scale, const float* scale_ptr, const float* grad2,
                  float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2, float eps,
                  int step, dvd_stream_t stream) {
  return dvd_adam_step_guarded(param, grad1, scale, scale_ptr, grad2, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, nullptr,
                               stream);
}

int dvd_adam_step_guarded(float* param, const float* grad1, float scale, const float* scale_ptr, const float* grad2,
                          float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2, float eps,
                          int step, const float* skip_flag, dvd_stream_t stream) {
  using namespace dvd;
  DVD_REQUIRE(param && grad1 && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad argument");
  DVD_REQUIRE(al16(param) && al16(grad1) && al16(grad2) && al16(exp_avg) && al16(exp_avg_sq),
              "adam_step: pointers must be 16-byte aligned");
  bytes_add(DVD_BYTES_ADAM, 4.0 * (double)n * (grad2 ? 8 : 7));
  // bias corrections in double, like torch's python-scalar arithmetic
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad1,
                     scale, scale_ptr, grad2, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, inv_sqrt_bc2, skip_flag, step, lr);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
