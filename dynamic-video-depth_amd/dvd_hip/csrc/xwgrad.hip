// Backward-weight of the dense stride-1 "same" convolutions (1x1, 3x3 and -- round 4 -- 5x5, 7x7, 11x11), NCHW fp32, on fp32
// MFMA for gfx950:
//   dW[co][ci][ky][kx] = sum_{n, r, c} gy[n][co][r][c] * act(x)[n][ci][r + ky - pad][c + kx - pad]
// (act = identity or the ReLU that csrc/xconv.hip applies to its input).  Deterministic: no atomics.
//
// What it replaces (reference, /root/reference): the autograd weight gradient of the dense nn.Conv2d of the
// MiDaS decoder (third_party/midas_blocks.py:102-168, MiDaS.py:186-195) and of the ResNeXt bottleneck 1x1
// convolutions (midas_blocks.py:35-50) -- MIOpen's igemm_wrw kernels, which accumulate with atomics (the
// run-to-run gradient noise VERDICT r01 measured) and run at 60-120 TF/s.
//
// Mapping (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains): D tile = 32 output channels x 32 input channels,
// K = pixels, two per MFMA: lane l holds A[co = l&31][pixel 2q + (l>>5)] = gy and
// B[pixel 2q + (l>>5)][ci = l&31] = x shifted by the tap.  A block owns COB = 32 PM output channels x
// CIB = 32 PN input channels and walks over a slice of the (image, tile) list; wave (pm, pn, ky) keeps the KS
// accumulators of kernel row ky.  Per tile the gy rows and the haloed x rows sit in LDS channel-major with odd
// pitches (the 32 lanes of a half wave read 32 channels at one pixel: 32 different banks); the next tile's rows
// are requested into registers before the MFMAs of the current one.  Every block writes its partial sums to
// the workspace; `xwgrad_reduce_kernel` adds them in slice order.
// The 3x3 / 1x1 convolutions of the depth nets run on the faster split-operand kernels of csrc/xwgrad3.hip; this exact-fp32
// kernel serves the large kernels of the hourglass's inception branches (third_party/hourglass.py:21-57: 5x5, 7x7, 11x11 --
// round 3 left those to MIOpen) and the 5x5 space-to-depth form of the ResNeXt stem (third_party/midas_blocks.py:35-45,
// dvd_hip/third_party/MiDaS.py): COB = CIB = 32 channels, one wave per kernel row, an 11-wide kernel row split over two
// blocks (KXN columns each) so that the accumulators fit the register file.
#include "dvd_common.h"

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgArgs {
  const float* __restrict__ x;
  const float* __restrict__ gy;
  float* __restrict__ partial;   // [S][T][Cout][Cin]
  int N, Cin, Cout, H, W;
  int TC, ntr, ntc, S;
  int relu_in;
  int nkxb;                      // blocks per kernel row (column ranges of KXN taps)
};

// KS = 3: TR = 3 rows x TC <= 62 columns (x rows of TC + 2 <= 64 pixels: one wave-wide load per row).
// KS = 1: the image is one row of H * W pixels; a tile is TR = 2 "rows" of TC = 64 consecutive pixels.
template <int KS, int PM, int PN, int TR, int KXN = KS, int UN = 8>
__global__ __launch_bounds__(64 * PM * PN * KS) void xwgrad_kernel(const WgArgs a) {
  constexpr int NW = PM * PN * KS;
  constexpr int PAD = KS / 2;
  constexpr int COB = 32 * PM, CIB = 32 * PN;
  constexpr int XR = TR + 2 * PAD;
  constexpr int GROWS = COB * TR, XROWS = CIB * XR;
  constexpr int NG = (GROWS + NW - 1) / NW, NX = (XROWS + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int TC = a.TC, XC = TC + 2 * PAD;
  const int PG = (TR * TC) | 1, PX = (XR * XC) | 1;
  float* sG = smem;                 // [COB][PG]
  float* sX = smem + COB * PG;      // [CIB][PX]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ky = wave % KS, pn = (wave / KS) % PN, pm = wave / (KS * PN);
  const int co0 = blockIdx.z * COB, ci0 = blockIdx.y * CIB;
  const size_t plane = (size_t)a.H * a.W;
  const int tiles_per_img = a.ntr * a.ntc;
  const int total = a.N * tiles_per_img;
  const int slice = KXN == KS ? blockIdx.x : blockIdx.x / a.nkxb;
  const int kx0 = KXN == KS ? 0 : (blockIdx.x - slice * a.nkxb) * KXN;     // first kernel column of this block

  float rg[NG], rx[NX];
  auto load_tile = [&](int t) {
    const int n = t / tiles_per_img, tt = t - n * tiles_per_img;
    const int tr = tt / a.ntc, tc = tt - tr * a.ntc;
    const int r0 = tr * TR, c0 = tc * TC;
    const float* gb = a.gy + (size_t)n * a.Cout * plane;
    const float* xb = a.x + (size_t)n * a.Cin * plane;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int rowid = i * NW + wave;              // wave-uniform: (channel, row)
      const int ch = rowid / TR, rr = rowid - ch * TR;
      // KS == 1: the "rows" of a tile are consecutive runs of TC pixels of the flattened image
      const int row = KS == 1 ? 0 : r0 + rr, col = KS == 1 ? tc * (TR * TC) + rr * TC + lane : c0 + lane;
      const bool ok = rowid < GROWS && (co0 + ch) < a.Cout && row < a.H && lane < TC && col < a.W;
      const int chc = (co0 + ch) < a.Cout ? (co0 + ch) : (a.Cout - 1);
      const float v = gb[(size_t)chc * plane + (ok ? row * a.W + col : 0)];
      rg[i] = ok ? v : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int rowid = i * NW + wave;
      const int ch = rowid / XR, rr = rowid - ch * XR;
      const int row = KS == 1 ? 0 : r0 - PAD + rr, col = KS == 1 ? tc * (TR * TC) + rr * TC + lane : c0 - PAD + lane;
      const bool ok = rowid < XROWS && (ci0 + ch) < a.Cin && row >= 0 && row < a.H && lane < XC && col >= 0 && col < a.W;
      const int chc = (ci0 + ch) < a.Cin ? (ci0 + ch) : (a.Cin - 1);
      const float v = xb[(size_t)chc * plane + (ok ? row * a.W + col : 0)];
      rx[i] = ok ? (a.relu_in ? fmaxf(v, 0.0f) : v) : 0.0f;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int rowid = i * NW + wave;
      const int ch = rowid / TR, rr = rowid - ch * TR;
      if (rowid < GROWS && lane < TC) sG[ch * PG + rr * TC + lane] = rg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int rowid = i * NW + wave;
      const int ch = rowid / XR, rr = rowid - ch * XR;
      if (rowid < XROWS && lane < XC) sX[ch * PX + rr * XC + lane] = rx[i];
    }
  };

  f32x16 acc[KXN];
#pragma unroll
  for (int k = 0; k < KXN; ++k) acc[k] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float* ga = sG + (pm * 32 + (lane & 31)) * PG + (lane >> 5);
  // (a column range past the kernel's last column reads in-range LDS cells of the next row; those accumulators are dropped)
  const float* xa = sX + (pn * 32 + (lane & 31)) * PX + ky * XC + (lane >> 5) + kx0;
  const int pw = TC >> 1;                             // pixel pairs per row (TC is even)

  int t = slice;
  if (t < total) load_tile(t);
  for (; t < total; t += a.S) {
    __syncthreads();                                  // the previous tile's MFMAs have read their operands
    store_tile();
    __syncthreads();
    if (t + a.S < total) load_tile(t + a.S);
#pragma unroll
    for (int rr = 0; rr < TR; ++rr) {
      const float* gr = ga + rr * TC;
      const float* xr = xa + rr * XC;
      // UN pixel pairs per trip: their LDS reads are issued together, ahead of the MFMAs that use them
      int q = 0;
      for (; q + UN <= pw; q += UN) {
        float av[UN], bv[UN][KXN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          av[u] = gr[2 * (q + u)];
#pragma unroll
          for (int kx = 0; kx < KXN; ++kx) bv[u][kx] = xr[2 * (q + u) + kx];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
          for (int kx = 0; kx < KXN; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][kx], acc[kx], 0, 0, 0);
      }
      for (; q < pw; ++q) {
        const float av = gr[2 * q];
#pragma unroll
        for (int kx = 0; kx < KXN; ++kx) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xr[2 * q + kx], acc[kx], 0, 0, 0);
      }
    }
  }
  // partial[s][tap][co][ci]
  float* dst = a.partial + (size_t)slice * (KS * KS) * a.Cout * a.Cin;
#pragma unroll
  for (int kx = 0; kx < KXN; ++kx) {
    if (kx0 + kx >= KS) break;
    const int tap = ky * KS + kx0 + kx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + pm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ci = ci0 + pn * 32 + (lane & 31);
      if (co < a.Cout && ci < a.Cin) dst[((size_t)tap * a.Cout + co) * a.Cin + ci] = acc[kx][r];
    }
  }
}

// gw[co][ci][tap] = sum_s partial[s][tap][co][ci], ascending s (two interleaved chains)
__global__ __launch_bounds__(256) void xwgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int S,
                                                            int T, int Cout, int Cin) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over [tap][co][ci]
  const long long per = (long long)T * Cout * Cin;
  if (i >= per) return;
  float s0 = 0.0f, s1 = 0.0f;
  int s = 0;
  for (; s + 1 < S; s += 2) {
    s0 += partial[(size_t)s * per + i];
    s1 += partial[(size_t)(s + 1) * per + i];
  }
  if (s < S) s0 += partial[(size_t)s * per + i];
  const int ci = (int)(i % Cin);
  const int co = (int)((i / Cin) % Cout);
  const int tap = (int)(i / ((long long)Cin * Cout));
  gw[((size_t)co * Cin + ci) * T + tap] = s0 + s1;
}

struct WgPlan {
  int KS, TR, TC, ntr, ntc, S, cob, cib, nco, nci, nkxb;
  size_t lds;
};
static bool wg_plan(int N, int Cin, int Cout, int H, int W, int KS, WgPlan& p) {
  p.KS = KS;
  int Hh = H, Ww = W;
  if (KS == 1) {
    Hh = 1;
    Ww = H * W;
    p.TR = 2;
    p.TC = 64;
    p.cob = p.cib = 128;
    p.ntr = 1;
    p.ntc = (Ww + 127) / 128;
  } else if (KS == 3) {
    p.TR = 3;
    const int nct = (Ww + 61) / 62;
    p.TC = ((Ww + nct - 1) / nct + 1) & ~1;
    p.cob = p.cib = 64;
    p.ntr = (Hh + p.TR - 1) / p.TR;
    p.ntc = (Ww + p.TC - 1) / p.TC;
  } else if (KS == 5 || KS == 7 || KS == 11) {
    // x rows of TC + 2 pad <= 64 pixels (one wave-wide load per row), TC even; 32 x 32 channels per block
    p.TR = KS == 11 ? 2 : 3;
    const int tcmax = (64 - 2 * (KS / 2)) & ~1;
    const int nct = (Ww + tcmax - 1) / tcmax;
    p.TC = ((Ww + nct - 1) / nct + 1) & ~1;
    if (p.TC > tcmax) p.TC = tcmax;
    p.cob = p.cib = 32;
    p.ntr = (Hh + p.TR - 1) / p.TR;
    p.ntc = (Ww + p.TC - 1) / p.TC;
  } else {
    return false;
  }
  p.nkxb = KS == 11 ? 2 : 1;
  p.nco = (Cout + p.cob - 1) / p.cob;
  p.nci = (Cin + p.cib - 1) / p.cib;
  const int total = N * p.ntr * p.ntc;
  // one block per CU is resident (LDS): make the grid a whole number of rounds over the 256 CUs
  const int pairs = p.nco * p.nci;
  int S = pairs >= 256 ? 1 : (512 + pairs - 1) / pairs;
  if (S > total) S = total;
  p.S = S;
  const int pad = KS / 2, XR = p.TR + 2 * pad, XC = p.TC + 2 * pad;
  p.lds = ((size_t)p.cob * ((p.TR * p.TC) | 1) + (size_t)p.cib * ((XR * XC) | 1) + 16) * sizeof(float);   // (+ slack: see kx0)
  return p.lds <= 160 * 1024;
}

}  // namespace dvd

extern "C" {

size_t dvd_xwgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int KS) {
  dvd::WgPlan p;
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !dvd::wg_plan(N, Cin, Cout, H, W, KS, p)) return 0;
  return (size_t)p.S * KS * KS * Cout * Cin * sizeof(float);
}

int dvd_xwgrad(const float* x, const float* gy, float* gw, void* workspace, size_t workspace_bytes, int N, int Cin,
               int Cout, int H, int W, int KS, int relu_in, dvd_stream_t stream) {
  DVD_REQUIRE(x && gy && gw && workspace, "xwgrad: null pointer");
  DVD_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "xwgrad: bad shape");
  DVD_REQUIRE((long long)H * W * (long long)(Cin > Cout ? Cin : Cout) < (1ll << 31), "xwgrad: image too large for 32-bit offsets");
  dvd::WgPlan p;
  DVD_REQUIRE(dvd::wg_plan(N, Cin, Cout, H, W, KS, p), "xwgrad: kernel size %d is not covered (1, 3, 5, 7, 11 are)", KS);
  const size_t need = (size_t)p.S * KS * KS * Cout * Cin * sizeof(float);
  if (workspace_bytes < need) {
    dvd::set_error("xwgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  dvd::WgArgs a;
  a.x = x;
  a.gy = gy;
  a.partial = static_cast<float*>(workspace);
  a.N = N; a.Cin = Cin; a.Cout = Cout;
  a.H = KS == 1 ? 1 : H;
  a.W = KS == 1 ? H * W : W;
  a.TC = p.TC; a.ntr = p.ntr; a.ntc = p.ntc; a.S = p.S;
  a.relu_in = relu_in ? 1 : 0;
  a.nkxb = p.nkxb;
  DVD_REQUIRE(p.nco <= 65535 && p.nci <= 65535, "xwgrad: too many channel blocks");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(p.S * p.nkxb, p.nci, p.nco);
  auto go = [&](auto kern, int waves) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * waves), p.lds, s, a);
    return DVD_OK;
  };
  if (KS == 5) {
    if (int e = go(dvd::xwgrad_kernel<5, 1, 1, 3>, 5)) return e;
  } else if (KS == 7) {
    if (int e = go(dvd::xwgrad_kernel<7, 1, 1, 3>, 7)) return e;
  } else if (KS == 11) {
    if (int e = go(dvd::xwgrad_kernel<11, 1, 1, 2, 6, 4>, 11)) return e;
  } else if (KS == 3) {
    auto kern = dvd::xwgrad_kernel<3, 2, 2, 3>;
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * 12), p.lds, s, a);
  } else {
    auto kern = dvd::xwgrad_kernel<1, 4, 4, 2>;
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * 16), p.lds, s, a);
  }
  DVD_LAUNCH_OK();
  const long long per = (long long)KS * KS * Cout * Cin;
  hipLaunchKernelGGL(dvd::xwgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                     static_cast<const float*>(workspace), gw, p.S, KS * KS, Cout, Cin);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
