// Fused eval-mode BatchNorm + (residual add) + ReLU for NCHW fp32, forward and backward (gfx950).
//
// What it replaces: `relu(bn(conv(x)))` and `relu(bn3(conv3(.)) + skip)` of the ResNeXt bottlenecks
// inside the MiDaS encoder (torchvision resnet.py Bottleneck, reached through
// third_party/midas_blocks.py:35-50).  The depth nets are ALWAYS in eval mode while training
// (models/scene_flow_motion_field.py:157,168): BatchNorm uses its running statistics but gamma and
// beta still receive gradients.  ATen runs this as 3-4 kernels per site and direction
// (batch_norm, add_, clamp_min_ / threshold_backward, batch_norm_backward): ~250 ms of a
// 2.2 s step (profiles/r01_bench_kernel_trace_summary.txt).
//
//   forward :  y = max(0, x * s[c] + b[c] (+ r)),  s = gamma / sqrt(var + eps),  b = beta - mean * s
//   backward:  g = gy * [y > 0];  gx = g * s[c];  gr = g;
//              gbeta[c] = sum g;   ggamma[c] = sum g * (x - mean[c]) / sqrt(var[c] + eps)
// Roofline: HBM -- forward 8 B (+4 with a residual) per element, backward 12 B read + 4 (+4)
// written.  One block = one (image, channel) plane segment, 16-byte accesses; the two channel
// sums of the backward are reduced per block, written as partials and summed in a fixed order by
// a second kernel (deterministic).

#include "dvd_io.h"
#include "dvd_split.h"

namespace dvd {

constexpr int kBnChunk = 4096;   // elements of one plane handled by a block (1024 float4)

template <class T>
__global__ __launch_bounds__(256) void bnrelu_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, T* __restrict__ y, int C, int HW, int chunks,
                                                         int relu, float* __restrict__ y_amax) {
  const int chunk = blockIdx.x % chunks;
  const long long pl = blockIdx.x / chunks;   // n * C + c
  const int c = (int)(pl % C);
  const float s = gamma[c] / sqrtf(var[c] + eps);
  const float b = beta[c] - mean[c] * s;
  const long long base = pl * HW;
  const int lo = chunk * kBnChunk, hi = min(HW, lo + kBnChunk);
  float m = 0.0f;        // max|y| of this thread (y_amax: the operand scale of the convolution that consumes y)
  if ((HW & 3) == 0) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      float4 v = ld4(x + base + i);
      v.x = __builtin_fmaf(v.x, s, b);
      v.y = __builtin_fmaf(v.y, s, b);
      v.z = __builtin_fmaf(v.z, s, b);
      v.w = __builtin_fmaf(v.w, s, b);
      if (res) {
        const float4 r = ld4(res + base + i);
        v.x += r.x;
        v.y += r.y;
        v.z += r.z;
        v.w += r.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f);
        v.w = fmaxf(v.w, 0.0f);
      }
      st4(y + base + i, v);
      m = amax_acc(amax_acc(amax_acc(amax_acc(m, v.x), v.y), v.z), v.w);
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      float v = __builtin_fmaf(ldf(x + base + i), s, b);
      if (res) v += ldf(res + base + i);
      v = relu ? fmaxf(v, 0.0f) : v;
      stf(y + base + i, v);
      m = amax_acc(m, v);
    }
  }
  if (y_amax) {                            // (uniform) one look at the scalar per BLOCK, an atomic only if it raises the value:
    __shared__ float s_m[4];               // a block has 4 096 elements -- a look per wave cost the pass 40 % (1.3 ms per step)
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
      unsigned* p = reinterpret_cast<unsigned*>(y_amax);
      if (m > 0.0f && __float_as_uint(m) > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(p, __float_as_uint(m));
    }
  }
}

// partial[(c * N + n) * chunks + chunk] = (sum g, sum g * (x - mean[c])) of the block
template <class T>
__global__ __launch_bounds__(256) void bnrelu_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y,
                                                         const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, T* __restrict__ gx, T* __restrict__ gres,
                                                         float2* __restrict__ partial, int N, int C, int HW,
                                                         int chunks, int relu, float* __restrict__ pmax, int npb) {
  // npb > 1 (small planes, chunks == 1): a block owns the planes of npb consecutive images of one channel -- the deep
  // levels of the encoder have 1 008 / 252 elements per plane, and a block per plane spent its time on the reductions
  int chunk, c, n, n_end;
  if (npb > 1) {
    chunk = 0;
    c = blockIdx.x % C;
    n = (blockIdx.x / C) * npb;
    n_end = min(N, n + npb);
  } else {
    chunk = blockIdx.x % chunks;
    const long long pl = blockIdx.x / chunks;
    c = (int)(pl % C);
    n = (int)(pl / C);
    n_end = n + 1;
  }
  const int n_first = n;
  const float s = gamma[c] / sqrtf(var[c] + eps);
  const float mu = mean[c];
  const int lo = chunk * kBnChunk, hi = min(HW, lo + kBnChunk);
  float sg = 0.0f, sgx = 0.0f, gmax = 0.0f;
  for (; n < n_end; ++n) {
  const long long base = ((long long)n * C + c) * HW;
  if ((HW & 3) == 0) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      float4 g = ld4(gy + base + i);
      if (relu) {
        const float4 o = ld4(y + base + i);
        g.x = o.x > 0.0f ? g.x : 0.0f;
        g.y = o.y > 0.0f ? g.y : 0.0f;
        g.z = o.z > 0.0f ? g.z : 0.0f;
        g.w = o.w > 0.0f ? g.w : 0.0f;
      }
      sg += (g.x + g.y) + (g.z + g.w);
      gmax = amax_acc(amax_acc(amax_acc(amax_acc(gmax, g.x), g.y), g.z), g.w);
      if (x) {
        const float4 v = ld4(x + base + i);
        sgx = __builtin_fmaf(g.x, v.x - mu,
                             __builtin_fmaf(g.y, v.y - mu, __builtin_fmaf(g.z, v.z - mu, __builtin_fmaf(g.w, v.w - mu, sgx))));
      }
      if (gres) st4(gres + base + i, g);
      if (gx) st4(gx + base + i, make_float4(g.x * s, g.y * s, g.z * s, g.w * s));
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      float g = ldf(gy + base + i);
      if (relu) g = ldf(y + base + i) > 0.0f ? g : 0.0f;
      sg += g;
      gmax = amax_acc(gmax, g);
      if (x) sgx = __builtin_fmaf(g, ldf(x + base + i) - mu, sgx);
      if (gres) stf(gres + base + i, g);
      if (gx) stf(gx + base + i, g * s);
    }
  }
  }
  __shared__ float red[3][4];
  sg = wave_sum(sg);
  sgx = wave_sum(sgx);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_down(gmax, off, kWave));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = sg;
    red[1][wave] = sgx;
    red[2][wave] = gmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t rec = npb > 1 ? (size_t)c * ((N + npb - 1) / npb) + n_first / npb : ((size_t)c * N + n_first) * chunks + chunk;
    partial[rec] = make_float2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    // max|g| of the block: reduced to the scalar by the second kernel (tens of thousands of blocks folding it in with
    // atomics on one address cost more than the whole pass)
    if (pmax) pmax[rec] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
  }
}

// ggamma[c] = sum g (x - mean) / sqrt(var + eps), gbeta[c] = sum g   (fixed order over the records)
__global__ __launch_bounds__(64) void bnrelu_param_grad_kernel(const float2* __restrict__ partial,
                                                               const float* __restrict__ var, float eps,
                                                               float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                               int C, int records, const float* __restrict__ pmax,
                                                               float* g_amax, const float* __restrict__ out_scale,
                                                               const float* __restrict__ gamma, float* gx_amax) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  double sg = 0.0, sgx = 0.0;
  float m = 0.0f, mx = 0.0f;
  if (c < C) {
    for (int r = 0; r < records; ++r) {
      const float2 v = partial[(size_t)c * records + r];
      sg += v.x;
      sgx += v.y;
      if (pmax) m = fmaxf(m, pmax[(size_t)c * records + r]);
    }
    // out_scale: fp16 gradient storage -- the sums carry the step's loss scale, the PARAMETER gradients must not
    const double os = out_scale ? (double)out_scale[0] : 1.0;
    if (gbeta) gbeta[c] = (float)(sg * os);
    if (ggamma) ggamma[c] = (float)(sgx * os / sqrt((double)var[c] + (double)eps));
    // max|gx| over the channel: gx = g * s with the s of bnrelu_bwd_kernel, and rounding is monotonic -- exactly |s| max|g|
    if (gx_amax) mx = m * fabsf(gamma[c] / sqrtf(var[c] + eps));
  }
  if (g_amax) wave_amax_to(m, g_amax);      // C / 64 waves: a handful of atomics
  if (gx_amax) wave_amax_to(mx, gx_amax);
}

// Convolution + fused eval-mode BatchNorm, backward bookkeeping of one site.  The weight-gradient kernels ran on the
// UNSCALED masked gradient g = gy * [y > 0]:  dWu[co][k] = sum g[co] x[k].  With z = W x + cb (pre-BN), s = gamma * rstd:
//   dW[co][k]  = s[co] * dWu[co][k]
//   dgamma[co] = sum g * (z - mean) * rstd = rstd * (sum_k W[co][k] dWu[co][k] + (cb[co] - mean[co]) * dbeta[co])
//   dcb[co]    = s[co] * dbeta[co]
// One 64-thread block per output channel, fixed-order sums (deterministic).
__global__ __launch_bounds__(64) void convbn_finalize_kernel(const float* __restrict__ W, float* __restrict__ dW,
                                                             const float* __restrict__ dbeta, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                             const float* __restrict__ cbias, int K, float* __restrict__ dgamma,
                                                             float* __restrict__ dcbias) {
  const int co = blockIdx.x, lane = threadIdx.x;
  const float rstd = 1.0f / sqrtf(var[co] + eps);
  const float s = (gamma ? gamma[co] : 1.0f) * rstd;
  const float* w = W + (size_t)co * K;
  float* d = dW + (size_t)co * K;
  float acc = 0.0f;
  for (int k = lane; k < K; k += 64) {
    const float u = d[k];
    acc = __builtin_fmaf(w[k], u, acc);
    d[k] = u * s;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float db = dbeta[co];
    if (dgamma) dgamma[co] = rstd * (acc + ((cbias ? cbias[co] : 0.0f) - mean[co]) * db);
    if (dcbias) dcbias[co] = s * db;
  }
}

}  // namespace dvd

extern "C" {

int dvd_bnrelu_fwd(const float* x, const float* residual, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* y, int N, int C, int HW, int relu, dvd_stream_t stream) {
  return dvd_bnrelu_fwd_t(x, residual, gamma, beta, mean, var, eps, y, 0, N, C, HW, relu, stream);
}

int dvd_bnrelu_fwd_t(const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                     const float* var, float eps, void* y, int f16, int N, int C, int HW, int relu, dvd_stream_t stream) {
  return dvd_bnrelu_fwd_m(x, residual, gamma, beta, mean, var, eps, y, f16, N, C, HW, relu, nullptr, stream);
}

int dvd_bnrelu_fwd_m(const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                     const float* var, float eps, void* y, int f16, int N, int C, int HW, int relu, float* y_amax,
                     dvd_stream_t stream) {
  DVD_REQUIRE(x && gamma && beta && mean && var && y, "bnrelu fwd: null pointer");
  DVD_REQUIRE(N > 0 && C > 0 && HW > 0, "bnrelu fwd: bad shape");
  const int chunks = (HW + dvd::kBnChunk - 1) / dvd::kBnChunk;
  const long long blocks = (long long)N * C * chunks;
  DVD_REQUIRE(blocks < (1LL << 31), "bnrelu fwd: grid too large");
  dvd::bytes_add(DVD_BYTES_BNRELU_FWD, (double)N * C * HW * (f16 ? 2 : 4) * (residual ? 3 : 2));
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::bnrelu_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0,
                                         static_cast<hipStream_t>(stream), static_cast<const T*>(x), static_cast<const T*>(residual),
                                         gamma, beta, mean, var, eps, static_cast<T*>(y), C, HW, chunks, relu, y_amax));
  DVD_LAUNCH_OK();
  return DVD_OK;
}

size_t dvd_bnrelu_bwd_workspace_bytes(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const size_t chunks = (HW + dvd::kBnChunk - 1) / dvd::kBnChunk;
  return (size_t)N * C * chunks * (sizeof(float2) + sizeof(float));      // channel-sum records + block maxima
}

int dvd_bnrelu_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* mean,
                   const float* var, float eps, float* gx, float* g_residual, float* g_gamma, float* g_beta,
                   void* workspace, size_t workspace_bytes, int N, int C, int HW, int relu, float* g_amax, dvd_stream_t stream) {
  return dvd_bnrelu_bwd_t(gy, y, x, gamma, mean, var, eps, gx, g_residual, g_gamma, g_beta, workspace, workspace_bytes, 0, nullptr, N,
                          C, HW, relu, g_amax, stream);
}

int dvd_bnrelu_bwd_t(const void* gy, const void* y, const void* x, const float* gamma, const float* mean, const float* var,
                     float eps, void* gx, void* g_residual, float* g_gamma, float* g_beta, void* workspace,
                     size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int HW, int relu, float* g_amax,
                     dvd_stream_t stream) {
  return dvd_bnrelu_bwd_m(gy, y, x, gamma, mean, var, eps, gx, g_residual, g_gamma, g_beta, workspace, workspace_bytes, f16,
                          out_scale, N, C, HW, relu, g_amax, nullptr, stream);
}

int dvd_bnrelu_bwd_m(const void* gy, const void* y, const void* x, const float* gamma, const float* mean, const float* var,
                     float eps, void* gx, void* g_residual, float* g_gamma, float* g_beta, void* workspace,
                     size_t workspace_bytes, int f16, const float* out_scale, int N, int C, int HW, int relu, float* g_amax,
                     float* gx_amax, dvd_stream_t stream) {
  DVD_REQUIRE(!gx_amax || gx, "bnrelu bwd: max|gx| without gx");
  DVD_REQUIRE(gy && gamma && mean && var && workspace, "bnrelu bwd: null pointer");
  DVD_REQUIRE(x || !g_gamma, "bnrelu bwd: the gamma gradient needs the BatchNorm input");
  DVD_REQUIRE(!relu || y, "bnrelu bwd: the ReLU mask needs the forward output");
  DVD_REQUIRE(N > 0 && C > 0 && HW > 0, "bnrelu bwd: bad shape");
  if (workspace_bytes < dvd_bnrelu_bwd_workspace_bytes(N, C, HW)) {
    dvd::set_error("bnrelu bwd: workspace too small");
    return DVD_ENOSPC;
  }
  dvd::bytes_add(DVD_BYTES_BNRELU_BWD, (double)N * C * HW * (f16 ? 2 : 4) * (1 + (y ? 1 : 0) + (x ? 1 : 0) + (gx ? 1 : 0) + (g_residual ? 1 : 0)));
  const int chunks = (HW + dvd::kBnChunk - 1) / dvd::kBnChunk;
  // small planes: several images of a channel per block (as many as make ~4 096 elements, while >= 1 024 blocks remain)
  int npb = 1;
  if (chunks == 1 && HW <= dvd::kBnChunk / 2) {
    npb = dvd::kBnChunk / HW;
    while (npb > 1 && (long long)C * ((N + npb - 1) / npb) < 1024) npb >>= 1;
    if (npb > N) npb = N;
    if (npb < 1) npb = 1;
  }
  const int records = npb > 1 ? (N + npb - 1) / npb : N * chunks;          // per channel (<= the workspace's N * chunks)
  const long long blocks = (long long)C * records;
  DVD_REQUIRE(blocks < (1LL << 31), "bnrelu bwd: grid too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* pmax = (g_amax || gx_amax) ? reinterpret_cast<float*>(static_cast<float2*>(workspace) + (size_t)N * C * chunks) : nullptr;
  DVD_DISPATCH_T(f16, hipLaunchKernelGGL(dvd::bnrelu_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s,
                                         static_cast<const T*>(gy), static_cast<const T*>(y), static_cast<const T*>(x), gamma, mean,
                                         var, eps, static_cast<T*>(gx), static_cast<T*>(g_residual),
                                         static_cast<float2*>(workspace), N, C, HW, chunks, relu, pmax, npb));
  DVD_LAUNCH_OK();
  if (g_gamma || g_beta || g_amax || gx_amax) {
    hipLaunchKernelGGL(dvd::bnrelu_param_grad_kernel, dim3((C + 63) / 64), dim3(64), 0, s,
                       static_cast<const float2*>(workspace), var, eps, g_gamma, g_beta, C, records, pmax, g_amax, out_scale,
                       gamma, gx_amax);
    DVD_LAUNCH_OK();
  }
  return DVD_OK;
}

int dvd_convbn_finalize(const float* W, float* dW, const float* dbeta, const float* gamma, const float* mean, const float* var,
                        float eps, const float* conv_bias, int Cout, int K, float* dgamma, float* dconv_bias, dvd_stream_t stream) {
  DVD_REQUIRE(W && dW && dbeta && mean && var, "convbn_finalize: null pointer");
  DVD_REQUIRE(Cout > 0 && K > 0, "convbn_finalize: bad shape");
  hipLaunchKernelGGL(dvd::convbn_finalize_kernel, dim3(Cout), dim3(64), 0, static_cast<hipStream_t>(stream), W, dW, dbeta, gamma,
                     mean, var, eps, conv_bias, K, dgamma, dconv_bias);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
