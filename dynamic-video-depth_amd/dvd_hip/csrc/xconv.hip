// Dense stride-1 "same" convolution (1x1, 3x3, ... odd k x k), NCHW fp32 in / fp32 out, as an
// implicit GEMM on the 16-bit matrix cores of gfx950 with fp32-class accuracy: forward and
// backward-data (the same kernel with the weights packed transposed and the taps flipped).
//
// What it replaces (reference, /root/reference): every dense nn.Conv2d of the depth networks --
//   third_party/midas_blocks.py:102-168   ResidualConvUnit / FeatureFusionBlock 3x3, 256 -> 256
//   third_party/MiDaS.py:186-195          scratch.layerK_rn (3x3, {256..2048} -> 256), output_conv
//   third_party/midas_blocks.py:35-50     the 1x1 convolutions of the ResNeXt-101 32x8d bottlenecks
//   third_party/hourglass.py:21-57        the 1x1 and k x k branches of the inception blocks
// and their autograd backward w.r.t. the input (MIOpen: fp32 Winograd / Tensile GEMMs + layout
// transposes in profiles/r01_bench_kernel_trace_summary.txt).
//
// Arithmetic (csrc/dvd_split.h): every fp32 operand, scaled by a power of two taken from its tensor's max|.|, is split
// into two fp16 terms x * 2^e = h + l (22 significant bits) and a product is the three partial products l*h' + h*l' + h*h' on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, unscaled (exactly) in the epilogue.  Round 2 used three bf16 terms and six
// products; the matrix pipe is power limited, so halving the MFMAs per product is what moves the rate (183 -> ~300 TF/s on
// the decoder's 3x3 convolution in a timing-only build, profiles/r03_xconv_power_limit.txt) at the same measured error
// against float64 (<= 4e-6 of max|y|, tests/test_06_xconv_gpu.py).
//
// Mapping.  GEMM M = output channel, N = pixel, K = input channel (x taps).  D tile 32x32
// (row = channel, column = pixel; accumulator layout of the 32x32 MFMAs: row = (r&3) + 8 (r>>2) + 4 (lane>>5),
// column = lane & 31).  One K step = 16 input channels: lane l holds A[m = l&31][k = 8 (l>>5) .. +7] and
// B[k = 8 (l>>5) .. +7][n = l&31] as eight fp16 (16 bytes) per split term.
//   * weights: packed once per weight update (dvd_xconv_pack) into fragment order, already split; per K step
//     (16 channels x one tap) the block copies its A fragments L2 -> registers -> LDS two steps ahead of use
//     (first version: every wave streamed its own fragments from L2 -- 15 GB of L1 traffic per decoder
//     convolution, the kernel ran at the L1 rate, not the MFMA rate).
//   * activations: a block owns a TR x TC tile of one image and all input channels in chunks of 16.  The
//     chunk's haloed tile is read from HBM as fp32 (coalesced along x), split, and written to LDS as
//     [term][channel group of 8][position][8 fp16]; positions are linear in the PADDED tile,
//     q = r * (TC + 2 pad) + c, so the B fragment of tap (ky, kx) is the fragment of the centre tap at a
//     constant LDS offset ky * P + kx -- 32 consecutive 16-byte cells, conflict free -- and an N tile of 32
//     consecutive q may straddle rows (the 2 pad columns per row are computed and dropped).  The raw fp32
//     values of the next chunk are requested before the MFMAs of the current one.
//   * epilogue: + bias[co], + residual (optionally relu'd), * [mask_src > 0], ReLU; coalesced stores.
#include <type_traits>

#include "dvd_split.h"

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: arrays of it stay in registers

// Packed weights: a 256-byte header (float 0: max|A| over the whole tensor, written by xconv_pack_kernel from the partial maxima
// xconv_wamax*_kernel leaves in floats 4 .. 63) followed by
// frag[(((mt * nkc + kc) * T + tap) * 2 + term) * 64 + lane] : 8 fp16 of A * pow2_scale(max|A|)
//   forward:    A[m][k] = w[co = m][ci = k][tap]
//   transposed: A[m][k] = w[co = k][ci = m][T - 1 - tap]      (backward-data: roles swapped, taps flipped)
// rows m >= M and columns k >= K are zero (M, K are padded to the block / K-step granularity).
// Grouped convolutions: Cout / Cin are per group, w is [G * Cout][Cin][T]; group g's fragments follow group g - 1's.
// sc_gamma / sc_var (optional): every weight of output channel co is multiplied by gamma[co] / sqrt(var[co] + eps) (gamma
// null = 1) -- the backward-data pass of a convolution whose eval-mode BatchNorm is fused into its epilogue.
#ifndef DVD_XCONV_ROLL
#define DVD_XCONV_ROLL 1
#endif
#ifndef DVD_XCONV_SPARE
#define DVD_XCONV_SPARE 0     // 1: the per-tap loop stages past the last chunk unconditionally (rounds 2-5)
#endif
#ifndef DVD_XCONV_DIRECT_BATCH
#define DVD_XCONV_DIRECT_BATCH 4   // direct staging (k >= 5, stride 2): items whose loads are in flight together (1: rounds 2-5)
#endif
constexpr int kXHeader = 16;       // uint4 cells in front of the fragments

// value of A's element as the pack kernel sees it (BatchNorm scale folded in): shared by the amax and the pack kernels
__device__ __forceinline__ float xconv_weight(const float* __restrict__ w, int co_g, int idx, const float* __restrict__ sc_gamma,
                                              const float* __restrict__ sc_var, float sc_eps) {
  float val = w[idx];
  if (sc_var) val *= (sc_gamma ? sc_gamma[co_g] : 1.0f) / sqrtf(sc_var[co_g] + sc_eps);
  return val;
}

// Block maximum -> header float [kXPartial0 + blockIdx.x].  No atomics and nothing to clear: every block of the launch writes
// its slot, the pack kernel that follows takes the maximum of the slots (round 5; rounds 2-4 cleared header[0] with a launch of
// its own and let every wave race an atomic max into it: 503 clearing launches per step and, with many waves, ~88 serialised
// atomics per microsecond).
constexpr int kXPartial0 = 4, kXPartials = 60;      // floats 4 .. 63 of the 256-byte header
template <int NWAVES>
__device__ __forceinline__ void block_partial_max(float m, float* header) {
  __shared__ float wm[NWAVES];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = wm[0];
#pragma unroll
    for (int i = 1; i < NWAVES; ++i) r = fmaxf(r, wm[i]);
    header[kXPartial0 + blockIdx.x] = r;
  }
}

// max |w[co][...]| * |bn scale[co]| over all weights -> the header's partial slots.  grid.y = output channel (row), so the
// BatchNorm scale is a per-block constant and the loop is a plain strided read (a flat index with a 64-bit division per
// element ran 45 us per call, 20 ms per step).
__global__ __launch_bounds__(256) void xconv_wamax_kernel(const float* __restrict__ w, float* __restrict__ header, int rows,
                                                          int row_len, const float* __restrict__ sc_gamma,
                                                          const float* __restrict__ sc_var, float sc_eps) {
  float mall = 0.0f;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* wr = w + (size_t)row * row_len;
    float m = 0.0f;
    for (int i = threadIdx.x; i < row_len; i += 256) m = fmaxf(m, fabsf(wr[i]));
    if (sc_var) m *= fabsf((sc_gamma ? sc_gamma[row] : 1.0f) / sqrtf(sc_var[row] + sc_eps));
    mall = fmaxf(mall, m);
  }
  block_partial_max<4>(mall, header);
}

// The same for rows of whole float4s at 16-byte aligned addresses (every layer but the stem's 27-element rows, and weights
// that sit at an odd offset of a flat parameter buffer): a WAVE per (row, chunk of 1024 floats) unit, four 16-byte loads per lane
// in flight, at most 60 blocks of 16 waves walking the units.  (The row-per-block loop above reads 4 bytes per lane and instruction: 14.7 us
// per call on average over the 486 weight tensors a step packs = 98 GB/s.  A first version of this kernel with one wave per unit
// and an atomic per wave was SLOWER, 17.7 us: 2 048 waves finishing together on one header word; with one atomic per block 7.4.)
constexpr int kWamaxWaves = 16;      // 1024-thread blocks: the number of blocks is bounded by the header's partial slots
__global__ __launch_bounds__(64 * kWamaxWaves) void xconv_wamax4_kernel(const float* __restrict__ w, float* __restrict__ header, int rows,
                                                           int row_len, int cpr, const float* __restrict__ sc_gamma,
                                                           const float* __restrict__ sc_var, float sc_eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, units = rows * cpr, n4 = row_len >> 2;
  float mall = 0.0f;
  for (int unit = blockIdx.x * kWamaxWaves + wave; unit < units; unit += gridDim.x * kWamaxWaves) {
    const int row = unit / cpr, ch = unit - row * cpr;
    const float4* p = reinterpret_cast<const float4*>(w + (size_t)row * row_len);
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = ch * 256 + k * 64 + lane;
      v[k] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
    if (sc_var) m *= fabsf((sc_gamma ? sc_gamma[row] : 1.0f) / sqrtf(sc_var[row] + sc_eps));
    mall = fmaxf(mall, m);
  }
  block_partial_max<kWamaxWaves>(mall, header);
}

__device__ __forceinline__ void xconv_pack_block(const float* __restrict__ w, uint4* __restrict__ packed, int Cout,
                                                 int Cin, int T, int transposed, int mtiles, int nkc, int G,
                                                 const float* __restrict__ sc_gamma, const float* __restrict__ sc_var,
                                                 float sc_eps, int npart, unsigned block) {
  // max|A| from the partial maxima the wamax launch left in the header (one load per lane, wave maximum); block 0 publishes it
  // in header[0] for the convolution kernels
  float amax = (int)(threadIdx.x & 63) < npart ? reinterpret_cast<const float*>(packed)[kXPartial0 + (threadIdx.x & 63)] : 0.0f;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, kWave));
  if (block == 0 && threadIdx.x == 0) reinterpret_cast<float*>(packed)[0] = amax;
  const long long idx = (long long)block * 256 + threadIdx.x;
  const long long total = (long long)G * mtiles * nkc * T * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long long f = idx >> 6;
  const int tap = (int)(f % T);
  const int kc = (int)((f / T) % nkc);
  const int mtg = (int)(f / T / nkc);
  const int g = mtg / mtiles, mt = mtg - g * mtiles;
  w += (size_t)g * Cout * Cin * T;
  const int m = mt * 32 + (lane & 31);
  const int k0 = kc * 16 + 8 * (lane >> 5);
  const int M = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float val = 0.0f;
    if (m < M && k < K) {
      const int co = transposed ? k : m, ci = transposed ? m : k, tp = transposed ? T - 1 - tap : tap;
      val = xconv_weight(w, g * Cout + co, (int)(((size_t)co * Cin + ci) * T + tp), sc_gamma, sc_var, sc_eps);
    }
    v[e] = val;
  }
  const float sw = pow2_scale(amax);
  uint4 h, l;
  split8_f16(v, sw, h, l);
  packed[kXHeader + (f * 2 + 0) * 64 + lane] = h;
  packed[kXHeader + (f * 2 + 1) * 64 + lane] = l;
}

__global__ __launch_bounds__(256) void xconv_pack_kernel(const float* __restrict__ w, uint4* __restrict__ packed, int Cout,
                                                         int Cin, int T, int transposed, int mtiles, int nkc, int G,
                                                         const float* __restrict__ sc_gamma, const float* __restrict__ sc_var,
                                                         float sc_eps, int npart) {
  xconv_pack_block(w, packed, Cout, Cin, T, transposed, mtiles, nkc, G, sc_gamma, sc_var, sc_eps, npart, blockIdx.x);
}

// ---- all packings of a network in two launches (round 6: dvd_xconv_pack_many) ----------------------------------------------
// A training step of MiDaS packs ~240 weight tensors (forward and transposed orders); one wamax + one pack launch per tensor
// and per captured depth-net chunk were 964 launches of ~6.5 us each for 4 GB of traffic (0.67 TB/s).  The table below lists
// every packing once; block (p, i) of the first launch takes item i's partial maximum p, block b of the second finds its item
// by bisection of the items' first-block numbers.  Same arithmetic per element as the single-tensor kernels: the packed bytes
// are identical.
struct XPackDev {
  const float* w;
  uint4* packed;
  const float* gamma;
  const float* var;
  float eps;
  int co, ci, T, transposed, mtiles, nkc, G;      // per-group channel counts, like xconv_pack_kernel's arguments
  int rows, row_len;                              // rows = total output channels (the BatchNorm scale's index)
  unsigned block0;                                // first block of this item in the pack launch
  int pad_;
};
constexpr int kManyPartials = 16;

__global__ __launch_bounds__(64 * kWamaxWaves) void xconv_wamax_many_kernel(const XPackDev* __restrict__ table) {
  const XPackDev it = table[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float mall = 0.0f;
  if (it.row_len % 4 == 0 && reinterpret_cast<uintptr_t>(it.w) % 16 == 0) {
    const int cpr = (it.row_len + 1023) / 1024, units = it.rows * cpr, n4 = it.row_len >> 2;
    for (int unit = blockIdx.x * kWamaxWaves + wave; unit < units; unit += gridDim.x * kWamaxWaves) {
      const int row = unit / cpr, ch = unit - row * cpr;
      const float4* p = reinterpret_cast<const float4*>(it.w + (size_t)row * it.row_len);
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = ch * 256 + k * 64 + lane;
        v[k] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float m = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
      if (it.var) m *= fabsf((it.gamma ? it.gamma[row] : 1.0f) / sqrtf(it.var[row] + it.eps));
      mall = fmaxf(mall, m);
    }
  } else {
    for (int row = blockIdx.x * kWamaxWaves + wave; row < it.rows; row += gridDim.x * kWamaxWaves) {
      const float* wr = it.w + (size_t)row * it.row_len;
      float m = 0.0f;
      for (int i = lane; i < it.row_len; i += 64) m = fmaxf(m, fabsf(wr[i]));
      if (it.var) m *= fabsf((it.gamma ? it.gamma[row] : 1.0f) / sqrtf(it.var[row] + it.eps));
      mall = fmaxf(mall, m);
    }
  }
  block_partial_max<kWamaxWaves>(mall, reinterpret_cast<float*>(it.packed));
}

__global__ __launch_bounds__(256) void xconv_pack_many_kernel(const XPackDev* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;                         // the last item whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const XPackDev it = table[lo];
  xconv_pack_block(it.w, it.packed, it.co, it.ci, it.T, it.transposed, it.mtiles, it.nkc, it.G, it.gamma, it.var, it.eps,
                   kManyPartials, blockIdx.x - it.block0);
}

struct XArgs {
  const void* __restrict__ x;           // float, or _Float16 in the IN16 kernels (fp16 activation storage, BASELINE configs[4])
  const uint4* __restrict__ wp;
  const float* __restrict__ bias;
  const void* __restrict__ res;         // residual / mask source / y: float, or _Float16 in the OUT16 kernels
  const void* __restrict__ mask_src;
  const float* __restrict__ bn_gamma;   // fused eval-mode BatchNorm of the output (bn_var != null): gamma / beta may be null
  const float* __restrict__ bn_beta;
  const float* __restrict__ bn_mean;
  const float* __restrict__ bn_var;
  float bn_eps;
  const float* __restrict__ x_amax;     // device scalar: max|x| (or an upper bound) of the whole input tensor
  float* y_amax;                        // optional device scalar: max|y| is folded into it (atomic max)
  void* __restrict__ y;
  int N, Cin, Cout, H, W;   // Cin = real K, Cout = real M of this launch, PER GROUP
  int G, mbpg, mtiles;      // groups, channel blocks per group, packed 32-row tiles per group
  int KS, pad, T;
  int TR, TC, P, ntr, ntc, NV;
  int nkc, npos, nfi;     // nfi: B staging items per thread (FI == 0 kernels loop over them at run time)
  int relu_in, relu_out, res_relu;
  int vec_out;            // 1x1 kernels: positions are pixels in runs of 8 and every pointer is 16-byte aligned (wide epilogue)
  // Strided 3x3 convolutions (round 6; stride 1: Hi = H, Wi = W, S2 = ZI = 0).  H x W are always the OUTPUT's dimensions.
  //   S2 > 0  stride-2 forward: the input is Hi x Wi = (about) 2H x 2W; the haloed (2 TR + 1) x (2 TC + 1) input tile is staged
  //           as FOUR phase planes of S2 cells, plane (row parity, column parity), pitch P = TC + 1, so that tap (ky, kx) of
  //           output q = r * P + c is the cell q + ((ky & 1) * 2 + (kx & 1)) * S2 + (ky >> 1) * P + (kx >> 1): a constant
  //           offset per tap, as at stride 1
  //   ZI      backward-data of a stride-2 convolution: the input IS the Hi x Wi gradient of the strided output, staged as if it
  //           were zero-interleaved to H x W (v[2i][2j] = g[i][j]) -- the interleaved tensor never exists in HBM
  int Hi, Wi, S2, ZI;
};

// One block = WM x WN waves, each wave TM x TN tiles of 32 x 32; A and B double buffered in LDS, one barrier per
// K step (16 channels x one tap); two blocks share a CU, so one block's staging / barrier / epilogue overlaps
// the other's MFMAs.
// FAST (host: Cin a multiple of 16, one image's input channels span < 2^31 bytes): all main-loop global loads are raw
// BUFFER loads whose per-thread offset is computed once before the loop -- the chunk / channel / K-step advance is a
// wave-uniform SGPR offset, halo and padding items carry an out-of-range offset and read 0 from the hardware's bounds check.
// The generic path (64-bit per-element addresses, per-element clamps and selects) spent ~150 VALU instructions per staged
// item and chunk, ~35 of them the operand split itself; FAST leaves the split (+ ReLU) and the LDS stores.
// B1: ONE activation stage in LDS instead of two (an extra barrier at the end of each chunk separates the last fragment reads
// from the next chunk's stores): 42 KB per 128 x 128 block, so three blocks share a CU (3 waves per SIMD).  The counters of
// round 2 show the matrix pipe 60 % busy with two waves per SIMD -- each wave needs it 45 % of its time, the rest is barrier
// skew, LDS latency and staging -- and independent blocks are what fills the gaps.
// IN16 / OUT16 (round 4, fp16 activation storage): the input is read as _Float16 and IS the matrix operand -- one term, no
// scale, no split (the staging is a transposing copy), so a product costs TWO MFMAs (two-term weight x one-term activation)
// instead of three and the activation planes in LDS halve; the output (and the residual / mask operands of its epilogue) is
// _Float16.  The weights stay fp32 in HBM and two-term in the packed buffer, accumulation stays fp32.
constexpr int kXFitOne = 11;     // FIT value of the 1x1 kernels (see kOne)
template <int TM, int TN, int WM, int WN, int FIT, bool FAST, bool B1 = false, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(64 * WM * WN, B1 ? 3 : 2) void xconv_kernel(const XArgs a) {
  static_assert(!IN16 || FAST, "fp16 activations exist on the buffer-addressed main loop only");
  using TX = std::conditional_t<IN16, _Float16, float>;
  using TY = std::conditional_t<OUT16, _Float16, float>;
  using RawT = std::conditional_t<IN16, unsigned short, float>;
  constexpr int BT = IN16 ? 1 : 2;               // split terms of an activation in LDS
  constexpr int XB = IN16 ? 2 : 4;               // bytes per activation element in HBM
  constexpr int NT = 64 * WM * WN;
  constexpr bool kDirect = FIT == 0;             // big halos (k >= 5): stage without the register prefetch
  constexpr bool kOne = FIT == kXFitOne;         // 1x1 convolutions, one staging item per thread: the two-chunk loop
  constexpr int FI = (kDirect || kOne) ? 1 : FIT;
  constexpr int MT = WM * TM;                    // 32-channel tiles per block
  constexpr int AU = MT * 2 * 64;                // uint4 per A stage (all tiles, two terms)
  constexpr int AI = (AU + NT - 1) / NT;         // staging loads per thread
  constexpr int AS = AI * NT;                    // LDS cells per A stage (>= AU)
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* sA = smem;                              // [2][AS] : [MT][2][64] + padding
  u32x4* sB = smem + 2 * AS;                     // [2 | 1][term 2][channel group 2][npos]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int tr = blockIdx.x / a.ntc, tc = blockIdx.x - tr * a.ntc;
  const int r0 = tr * a.TR, c0 = tc * a.TC;
  const int grp = blockIdx.y / a.mbpg;                 // group of a grouped convolution (0 for dense)
  const int mt0 = (blockIdx.y - grp * a.mbpg) * MT;
  const int n = blockIdx.z;
  const size_t plane = (size_t)a.H * a.W;                // output plane
  const size_t planeI = (size_t)a.Hi * a.Wi;             // input plane (differs for the strided forms only)
  const TX* xn = static_cast<const TX*>(a.x) + ((size_t)n * a.G + grp) * a.Cin * planeI;
  const int npos = a.npos, P = a.P, T = a.T, KS = a.KS;
  const int nkt = a.nkc * T;

  // ---- B staging bookkeeping: item = channel group * NV + position of the haloed tile.  Every thread runs
  //      every staging step unconditionally (no divergent branches around loads: hipcc would serialise them
  //      with vmcnt(0) waits and park the staging registers in scratch): items beyond the tile load a valid
  //      dummy address and store to the spare cell npos - 1 of the first plane.
  // staged item p of the haloed tile -> element offset in the input plane (go; ok = inside the image), LDS cell (lp)
  auto locate = [&](int p, bool live, int& go, bool& ok, int& lp) {
    if (a.S2) {                                        // block-uniform: stride-2 forward, phase planes
      const int PI = 2 * a.TC + 1;
      const int rr = p / PI, cc = p - rr * PI;
      const int row = 2 * r0 - 1 + rr, col = 2 * c0 - 1 + cc;
      ok = live && row >= 0 && row < a.Hi && col >= 0 && col < a.Wi;
      go = ok ? (row * a.Wi + col) : 0;
      lp = ((rr & 1) * 2 + (cc & 1)) * a.S2 + (rr >> 1) * P + (cc >> 1);
      return;
    }
    const int rr = p / P, cc = p - rr * P;
    const int row = r0 - a.pad + rr, col = c0 - a.pad + cc;
    ok = live && row >= 0 && row < a.H && col >= 0 && col < a.W;
    if (a.ZI) {                                        // virtual zero-interleaved input: only (even, even) cells exist
      ok = ok && !((row | col) & 1);
      go = ok ? ((row >> 1) * a.Wi + (col >> 1)) : 0;
    } else {
      go = ok ? (row * a.W + col) : 0;
    }
    lp = p;
  };
  // LDS offset of tap (ky, kx) relative to the centre-less origin of the tile
  auto tapoff = [&](int ky, int kx) {
    return a.S2 ? ((ky & 1) * 2 + (kx & 1)) * a.S2 + (ky >> 1) * P + (kx >> 1) : ky * P + kx;
  };
  int goff[FI], lidx[FI], cig8[FI];
  bool gok[FI];
#pragma unroll
  for (int it = 0; it < FI; ++it) {
    const int item = it * NT + tid;
    const int cig = item >= a.NV ? 1 : 0;
    const int p = item - cig * a.NV;
    const bool live = item < 2 * a.NV;
    int lp;
    locate(p, live, goff[it], gok[it], lp);
    lidx[it] = live ? (cig * npos + lp) : (npos - 1);
    cig8[it] = live ? cig * 8 : 0;
  }
  // FAST: byte offset of channel (0 | 8) of the item's position from the image's base, or an offset beyond the buffer
  const int planeB = (int)planeI * XB;
  const float sx = IN16 ? 1.0f : pow2_scale(a.x_amax[0]);         // power-of-two operand scales (csrc/dvd_split.h)
  const float sw = pow2_scale(reinterpret_cast<const float*>(a.wp)[0]);
  int voff[FI];
#pragma unroll
  for (int it = 0; it < FI; ++it) voff[it] = gok[it] ? (goff[it] + cig8[it] * (int)planeI) * XB : (int)0x80000000;
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<TX*>(xn)), 0, FAST ? a.Cin * planeB : 0, 0x00020000);
  typedef RawT RawSet[FI][8];
  RawSet raw, raw_b;        // raw_b: the second set of the 1x1 loop (two chunks in flight per block)
  // 8 fp16 channel values of one position -> one LDS cell (+ ReLU on packed halves): the whole "split" of the IN16 kernels
  auto cell16 = [&](const unsigned short (&r)[8]) {
    u32x4 c = {(unsigned)r[0] | ((unsigned)r[1] << 16), (unsigned)r[2] | ((unsigned)r[3] << 16),
               (unsigned)r[4] | ((unsigned)r[5] << 16), (unsigned)r[6] | ((unsigned)r[7] << 16)};
    if (a.relu_in) {                                 // uniform branch
      const f16x2 z = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        c[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, (unsigned)c[j]), z));
    }
    return c;
  };
  auto load_raw_to = [&](RawSet& raw, int kc) {
    if constexpr (IN16) {
      const int s0 = kc * 16 * planeB;
#pragma unroll
      for (int it = 0; it < FI; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[it][e] = __builtin_amdgcn_raw_buffer_load_b16(srdB, voff[it], s0 + e * planeB, 0);
    } else {
    if (FAST) {
      const int s0 = kc * 16 * planeB;               // wave-uniform: channel kc * 16 + e of this image
#pragma unroll
      for (int it = 0; it < FI; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          raw[it][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdB, voff[it], s0 + e * planeB, 0));
      return;
    }
#pragma unroll
    for (int it = 0; it < FI; ++it) {
      const int ch0 = kc * 16 + cig8[it];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = (ch0 + e) < a.Cin ? (ch0 + e) : (a.Cin - 1);
        raw[it][e] = xn[(size_t)ch * planeI + goff[it]];
      }
    }
    }
  };
  auto split_write_from = [&](const RawSet& raw, int buf, int kc) {
    u32x4* dst = sB + (B1 ? 0 : buf * 2 * BT * npos);
    if constexpr (IN16) {
#pragma unroll
      for (int it = 0; it < FI; ++it) dst[lidx[it]] = cell16(raw[it]);
    } else {
#pragma unroll
    for (int it = 0; it < FI; ++it) {
      const int ch0 = kc * 16 + cig8[it];
      float v[8];
      if (FAST) {                                    // invalid items were loaded as 0 by the bounds check
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = raw[it][e];
        if (a.relu_in) {                             // uniform branch
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = (gok[it] && (ch0 + e) < a.Cin) ? raw[it][e] : 0.0f;
          v[e] = a.relu_in ? fmaxf(t, 0.0f) : t;
        }
      }
      uint4 h, l;
      split8_f16(v, sx, h, l);
      dst[lidx[it]] = (u32x4){h.x, h.y, h.z, h.w};
      dst[2 * npos + lidx[it]] = (u32x4){l.x, l.y, l.z, l.w};
    }
    }
  };

  auto load_raw = [&](int kc) { load_raw_to(raw, kc); };
  auto split_write = [&](int buf, int kc) { split_write_from(raw, buf, kc); };

  // direct staging (kDirect): load, split and store item by item, nothing kept in registers across the MFMAs
  auto stage_direct = [&](int buf, int kc) {
    u32x4* dst = sB + (B1 ? 0 : buf * 2 * BT * npos);
    if constexpr (FAST && DVD_XCONV_DIRECT_BATCH > 1) {
      // items in batches of U: the 8 U loads of a batch are in flight together (one item at a time exposed one memory round
      // trip per item: 9 per chunk for a stride-2 tile, up to 16 for the hourglass's 11x11 branches).  Items past the tile
      // (the last batch's spare slots) are not live: an out-of-range offset, the spare LDS cell.
      constexpr int U = DVD_XCONV_DIRECT_BATCH;
      const int s0 = kc * 16 * planeB;
      for (int it0 = 0; it0 < a.nfi; it0 += U) {
        int vo[U], li[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int item = (it0 + u) * NT + tid;
          const int cig = item >= a.NV ? 1 : 0;
          const bool live = item < 2 * a.NV;
          int go, lp;
          bool ok;
          locate(item - cig * a.NV, live, go, ok, lp);
          vo[u] = ok ? (go + cig * 8 * (int)planeI) * XB : (int)0x80000000;
          li[u] = live ? (cig * npos + lp) : (npos - 1);
        }
        RawT r[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if constexpr (IN16) r[u][e] = __builtin_amdgcn_raw_buffer_load_b16(srdB, vo[u], s0 + e * planeB, 0);
            else r[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdB, vo[u], s0 + e * planeB, 0));
          }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (IN16) {
            dst[li[u]] = cell16(r[u]);
          } else {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = a.relu_in ? fmaxf(r[u][e], 0.0f) : r[u][e];
            uint4 h, l;
            split8_f16(v, sx, h, l);
            dst[li[u]] = (u32x4){h.x, h.y, h.z, h.w};
            dst[2 * npos + li[u]] = (u32x4){l.x, l.y, l.z, l.w};
          }
        }
      }
      return;
    }
    for (int it = 0; it < a.nfi; ++it) {
      const int item = it * NT + tid;
      const int cig = item >= a.NV ? 1 : 0;
      const int p = item - cig * a.NV;
      const bool live = item < 2 * a.NV;
      int go, lp;
      bool ok;
      locate(p, live, go, ok, lp);
      const int ch0 = kc * 16 + (live ? cig * 8 : 0);
      const int li = live ? (cig * npos + lp) : (npos - 1);
      if constexpr (IN16) {
        const int vo = ok ? (go + cig * 8 * (int)planeI) * XB : (int)0x80000000;
        const int s0 = kc * 16 * planeB;
        unsigned short r16[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r16[e] = __builtin_amdgcn_raw_buffer_load_b16(srdB, vo, s0 + e * planeB, 0);
        dst[li] = cell16(r16);
        continue;
      }
      float v[8];
      if (FAST) {
        const int vo = ok ? (go + cig * 8 * (int)planeI) * 4 : (int)0x80000000;
        const int s0 = kc * 16 * planeB;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdB, vo, s0 + e * planeB, 0));
        if (a.relu_in) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ch = (ch0 + e) < a.Cin ? (ch0 + e) : (a.Cin - 1);
          v[e] = (float)xn[(size_t)ch * planeI + go];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = (ok && (ch0 + e) < a.Cin) ? v[e] : 0.0f;
          v[e] = a.relu_in ? fmaxf(t, 0.0f) : t;
        }
      }
      uint4 h, l;
      split8_f16(v, sx, h, l);
      dst[li] = (u32x4){h.x, h.y, h.z, h.w};
      dst[2 * npos + li] = (u32x4){l.x, l.y, l.z, l.w};
    }
  };

  // ---- A staging: the block's AU uint4 of K step kt are MT pieces of 128 uint4 in the packed buffer; the
  //      LDS stage is padded to AI * NT cells so that every thread loads and stores unconditionally
  const u32x4* asrc[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int j = i * NT + tid;
    const int jj = j < AU ? j : 0;
    const int mtl = jj / 128, rem = jj - mtl * 128;
    asrc[i] = reinterpret_cast<const u32x4*>(a.wp) + kXHeader + ((size_t)(grp * a.mtiles + mt0 + mtl) * nkt) * 128 + rem;
  }
  // FAST: the block's fragments start at a wave-uniform base; thread-constant byte offset + K step * 2048 bytes (SGPR)
  const u32x4* abase = reinterpret_cast<const u32x4*>(a.wp) + kXHeader + ((size_t)(grp * a.mtiles + mt0) * nkt) * 128;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<u32x4*>(abase)), 0, FAST ? MT * nkt * 128 * 16 : 0, 0x00020000);
  int aoff[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int j = i * NT + tid;
    const int jj = j < AU ? j : 0;
    const int mtl = jj / 128, rem = jj - mtl * 128;
    aoff[i] = (mtl * nkt * 128 + rem) * 16;
  }
  struct ARegs {
    u32x4 v[AI];
  };
  auto load_a = [&](ARegs& r, int kt) {
    if (FAST) {
#pragma unroll
      for (int i = 0; i < AI; ++i)
        r.v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, aoff[i], kt * 2048, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) r.v[i] = asrc[i][(size_t)kt * 128];
  };
  auto write_a = [&](const ARegs& r, int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) sA[buf * AS + i * NT + tid] = r.v[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  int qb[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) qb[tn] = (wn * TN + tn) * 32 + (lane & 31);
  const int bl = (lane >> 5) * npos;
  const int al = (wm * TM) * 128 + lane;

  // Order inside a K step: barrier | fragments of step kt from LDS | 3 * TM * TN MFMAs | stage A(kt + 1)
  // (requested two steps ago) into the other A buffer, request A(kt + 3) | on the last tap of a chunk: split
  // the raw values of the next chunk into the other B buffer, request the chunk after it.  The waits for
  // global data sit BEHIND the wave's own MFMAs, so the matrix pipe works while they resolve (first version:
  // staging at the top of the step -- every step waited an L2 round trip before its first MFMA, pipe 60 % busy).
  struct Frag {
    f16x8 a[TM][2], b[TN][BT];
  };
  auto read_frags = [&](Frag& f, int abuf, int kc, int off) {
    const u32x4* Ac = sA + abuf * AS + al;
    const u32x4* Bc = sB + (B1 ? 0 : (kc & 1) * 2 * BT * npos) + bl + off;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int s = 0; s < 2; ++s) f.a[tm][s] = __builtin_bit_cast(f16x8, Ac[(tm * 2 + s) * 64]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int s = 0; s < BT; ++s) f.b[tn][s] = __builtin_bit_cast(f16x8, Bc[s * 2 * npos + qb[tn]]);
  };
  auto mfmas = [&](const Frag& f) {
    // three partial products l*h' + h*l' + h*h', small terms first; the TM * TN accumulators are interleaved so that
    // consecutive MFMAs are independent
#define DVD_XTERM(SA, SB)                                                                             \
  _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[tm][SA], f.b[tn][SB], acc[tm][tn], 0, 0, 0);
    DVD_XTERM(1, 0)
    if constexpr (!IN16) { DVD_XTERM(0, 1) }
    DVD_XTERM(0, 0)
#undef DVD_XTERM
  };

  // ---- prologue: A(0), B(0) staged; A(1), A(2) and raw(1) in flight
  // Wide wave tiles (TM >= 4: 48 MFMAs = 1 500+ matrix-pipe cycles per wave and K step) keep ONE register set for the weight
  // fragments: A(kt + 1) is requested right after the step's barrier and written to LDS behind the step's own MFMAs, which
  // cover the L2 round trip; the narrow tiles (24 MFMAs per step) need the request two to three steps ahead, in two sets.
  constexpr bool kSingleA = TM >= 4;
  // Rolling fragment prefetch (wide wave tiles, DVD_XCONV_ROLL): the fragments of step kt + 1 are read from LDS DURING the
  // MFMAs of step kt -- the activation fragments into a second register set, the weight fragments of tile row tm into the
  // registers that tile row's MFMAs have just released -- so that after a barrier every wave starts on the matrix pipe at
  // once instead of all eight waves first queueing on LDS (one 512-thread block per CU runs in lock step: counters showed
  // the pipe 45 % busy).  For that the stages run one step further ahead: A(kt + 2) is requested at the top of step kt and
  // stored at its end into the buffer of A(kt), whose fragments were read during step kt - 1; the activation chunk of step
  // kt + 2 is stored at the end of step kt whenever step kt + 1 is the last tap of its chunk.
  constexpr bool kRoll = kSingleA && !B1 && (DVD_XCONV_ROLL != 0);
  ARegs ra, rb;
  if (kRoll) {
    load_a(ra, 0);
    if (!kDirect) load_raw(0);
    write_a(ra, 0);
    load_a(ra, nkt > 1 ? 1 : 0);
    if (kDirect) {
      stage_direct(0, 0);
      if (T == 1 && a.nkc > 1) stage_direct(1, 1);
    } else {
      split_write(0, 0);
      load_raw(a.nkc > 1 ? 1 : 0);
      if (T == 1) {
        if (a.nkc > 1) split_write(1, 1);
        load_raw(a.nkc > 2 ? 2 : 0);
      }
    }
    write_a(ra, 1);
    __syncthreads();
    f16x8 fa[TM][2], fb[2][TN][BT];
    {
      const u32x4* Ac = sA + al;
      const u32x4* Bc = sB + bl;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[tm][t] = __builtin_bit_cast(f16x8, Ac[(tm * 2 + t) * 64]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int t = 0; t < BT; ++t) fb[0][tn][t] = __builtin_bit_cast(f16x8, Bc[t * 2 * npos + qb[tn]]);
    }
    // tap counters of the step whose fragments are being fetched (kt + 1)
    int nc = 0, ny = 0, nx = 0;
    auto advance = [&]() {
      if (++nx == KS) {
        nx = 0;
        if (++ny == KS) {
          ny = 0;
          ++nc;
        }
      }
    };
    advance();
    auto step_roll = [&](int kt, f16x8 (&bc)[TN][BT], f16x8 (&bn)[TN][BT]) {
      __syncthreads();
      load_a(ra, kt + 2 < nkt ? kt + 2 : kt);
      const u32x4* Ac = sA + ((kt + 1) & 1) * AS + al;
      const u32x4* Bc = sB + (nc & 1) * 2 * BT * npos + bl + tapoff(ny, nx);      // (past the last step: in range, never used)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int t = 0; t < BT; ++t) bn[tn][t] = __builtin_bit_cast(f16x8, Bc[t * 2 * npos + qb[tn]]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        // l*h' + h*l' + h*h' per accumulator, small terms first (the order of the other block shapes: identical results)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][1], bc[tn][0], acc[tm][tn], 0, 0, 0);
        if constexpr (!IN16) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][0], bc[tn][1], acc[tm][tn], 0, 0, 0);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][0], bc[tn][0], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);           // the reload below stays behind this tile row's MFMAs
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[tm][t] = __builtin_bit_cast(f16x8, Ac[(tm * 2 + t) * 64]);
        __builtin_amdgcn_sched_barrier(0);
      }
      write_a(ra, kt & 1);                           // A(kt + 2) over A(kt)
      // activations: is step kt + 1 the last tap of its chunk?  then step kt + 2 opens chunk nc + 1
      if (nx == KS - 1 && ny == KS - 1) {            // block-uniform
        const int c2 = nc + 1;
        if (kDirect) {
          if (c2 < a.nkc) stage_direct(c2 & 1, c2);
        } else {
          split_write(c2 & 1, c2);                   // (beyond the last chunk: zeros into the idle buffer)
          load_raw(c2 + 1 < a.nkc ? c2 + 1 : c2);
        }
      }
      advance();
    };
    // 1x1 convolutions (one tap per chunk): a K step is 16 channels of activations from HBM, and with ONE chunk requested per
    // step a block had 8 KB in flight for one step's MFMAs -- a third of the HBM latency: every step waited on its chunk,
    // 3.1 TB/s at any block shape (round 4 counters: matrix pipe 34 % busy, 37 % of the wave cycles in s_waitcnt).  Two
    // register sets keep two chunks in flight: chunk kt + 3 is requested at the TOP of step kt, BEHIND the step's weight
    // request (loads return in order: the wait for A(kt + 2) at the end of the step must not drain the activation request),
    // and split at the end of step kt + 1.
    auto step_one = [&](int kt, f16x8 (&bc)[TN][BT], const RawSet& rs, RawSet& ro) {
      __syncthreads();
      // (unconditional on purpose: with block-uniform branches around the tail's spare requests and splits -- 3 of 16 chunks,
      //  what paid in the per-tap loop below -- this loop ran 3-6 % SLOWER: the branches cost it its counted waits)
      load_a(ra, kt + 2 < nkt ? kt + 2 : kt);
      load_raw_to(ro, kt + 3 < a.nkc ? kt + 3 : a.nkc - 1);
      const u32x4* Ac = sA + ((kt + 1) & 1) * AS + al;
      const u32x4* Bc = sB + ((kt + 1) & 1) * 2 * BT * npos + bl;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][1], bc[tn][0], acc[tm][tn], 0, 0, 0);
        if constexpr (!IN16) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][0], bc[tn][1], acc[tm][tn], 0, 0, 0);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm][0], bc[tn][0], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[tm][t] = __builtin_bit_cast(f16x8, Ac[(tm * 2 + t) * 64]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the activation fragments of step kt + 1 go into the registers this step's last MFMAs have read (one set: the second
      // pays for the second raw set); their LDS round trip is covered by the staging below
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int t = 0; t < BT; ++t) bc[tn][t] = __builtin_bit_cast(f16x8, Bc[t * 2 * npos + qb[tn]]);
      __builtin_amdgcn_sched_barrier(0);
      write_a(ra, kt & 1);                           // A(kt + 2) over A(kt)
      split_write_from(rs, kt & 1, kt + 2);          // chunk kt + 2 (requested at the top of step kt - 1) over chunk kt
    };
    if constexpr (kOne) {
      for (int kt = 0; kt < nkt; kt += 2) {
        step_one(kt, fb[0], raw, raw_b);
        if (kt + 1 < nkt) step_one(kt + 1, fb[0], raw_b, raw);
      }
    } else {
      for (int kt = 0; kt < nkt; kt += 2) {
        step_roll(kt, fb[0], fb[1]);
        if (kt + 1 < nkt) step_roll(kt + 1, fb[1], fb[0]);
      }
    }
  } else {
  load_a(ra, 0);
  if (!kDirect) load_raw(0);
  write_a(ra, 0);
  if (!kSingleA) {
    load_a(rb, nkt > 1 ? 1 : 0);                     // set of odd steps
    load_a(ra, nkt > 2 ? 2 : 0);                     // set of even steps
  }
  if (kDirect) {
    stage_direct(0, 0);
  } else {
    split_write(0, 0);
    load_raw(a.nkc > 1 ? 1 : 0);
  }

  int kc = 0, ky = 0, kx = 0;
  auto step = [&](int kt, ARegs& rn) {             // rn holds A(kt + 1) (two-set scheme)
    __syncthreads();
    Frag f;
    read_frags(f, kt & 1, kc, tapoff(ky, kx));
    if (kSingleA) {
      load_a(rn, kt + 1 < nkt ? kt + 1 : kt);
      __builtin_amdgcn_sched_barrier(0);             // keep the request above the MFMAs (the scheduler sinks it to its use)
    }
    mfmas(f);
    write_a(rn, (kt + 1) & 1);                      // (a spare write after the last step is harmless)
    if (!kSingleA) load_a(rn, kt + 3 < nkt ? kt + 3 : kt);
    if (++kx == KS) {
      kx = 0;
      if (++ky == KS) {                             // last tap of the chunk (block-uniform)
        ky = 0;
        if (B1) __syncthreads();                    // every wave has read this chunk's last fragments
        if (kDirect) {
          if (kc + 1 < a.nkc) stage_direct((kc + 1) & 1, kc + 1);
        } else {
#if DVD_XCONV_SPARE
          split_write((kc + 1) & 1, kc + 1);
          load_raw(kc + 2 < a.nkc ? kc + 2 : kc);
#else
          // (block-uniform branches: with two chunks per tile -- the grouped layers -- the unconditional form of rounds 2-5
          //  split one chunk in three and requested two in four for nothing)
          if (kc + 1 < a.nkc) split_write((kc + 1) & 1, kc + 1);
          if (kc + 2 < a.nkc) load_raw(kc + 2);
#endif
        }
        ++kc;
      }
    }
  };
  if (kSingleA) {
    for (int kt = 0; kt < nkt; ++kt) step(kt, ra);
  } else {
    for (int kt = 0; kt < nkt; kt += 2) {
      step(kt, rb);
      if (kt + 1 < nkt) step(kt + 1, ra);
    }
  }
  }

  // ---- epilogue (uniform branches only; the optional operands are loaded in batches of 16; 32-bit offsets
  //      from the image's base: the host checks Cout * H * W < 2^31)
  const int TRv = (a.H - r0) < a.TR ? (a.H - r0) : a.TR;
  const int TCv = (a.W - c0) < a.TC ? (a.W - c0) : a.TC;
  const size_t ibase = ((size_t)n * a.G + grp) * a.Cout * plane;     // this group's channels of image n
  TY* __restrict__ yb = static_cast<TY*>(a.y) + ibase;
  const int iplane = (int)plane;
  // fused eval-mode BatchNorm: y = (z - mean) / sqrt(var + eps) * gamma + beta as z * s + (beta - mean * s), the formula of
  // csrc/bnrelu.hip.  Lane j computes s and the shift of channel 32 * tile + j ONCE (IEEE sqrt and division: 16 channels
  // per lane and tile evaluated in place cost more than the MFMAs of a 1x1 convolution); the 16 channels a lane's
  // accumulators belong to are fetched from those lanes where they are applied (keeping them in registers tripled the
  // kernel's VGPR count and cost the 1x1 kernels their third block per CU).
  const float unscale = 1.0f / (sx * sw);
  float ymax = 0.0f;
  if constexpr (kOne) {
    // Wide epilogue of the 1x1 kernels.  In the accumulator layout a lane holds ONE pixel of 16 channels per tile: 128 dword
    // stores per wave (and as many dword loads per residual / mask operand) -- the epilogue was a quarter of a 1x1 kernel's
    // time, store-issue bound.  Here a tile row (32 channels x 64 pixels of the wave) goes through LDS once and comes back with
    // a lane holding 4 (fp32) / 8 (fp16 output) consecutive pixels of one channel: 16-byte loads and stores, one bias /
    // BatchNorm pair per run.  Same arithmetic per element in the same order as the narrow epilogue below.
    if (a.vec_out) {                               // block-uniform
      constexpr int PXL = OUT16 ? 8 : 4;           // pixels per lane and run
      constexpr int LPR = 64 / PXL;                // lanes per channel row of the wave's 64 positions
      constexpr int RPI = 64 / LPR;                // channel rows per pass
      constexpr int kTP = 68;                      // floats per LDS row (64 + 4: 16-byte aligned rows, conflict-free both ways)
      __syncthreads();                             // every wave has read its last fragments: the stages are free
      float* Tw = reinterpret_cast<float*>(smem) + wave * (32 * kTP);
      const int j = lane & 31, hh = lane >> 5;
      const int px = (lane % LPR) * PXL, rl = lane / LPR;
      const int q0 = wn * TN * 32;                 // the wave's first position in the block's tile (TN = 2: 64 positions)
      const bool pok = q0 + px < TCv;              // whole runs: TC and H * W are multiples of 8
      float my_sc[TM], my_sh[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        my_sc[tm] = 1.0f;
        my_sh[tm] = 0.0f;
        if (a.bn_var) {
          const int co = (mt0 + wm * TM + tm) * 32 + (lane & 31);
          const int cg = grp * a.Cout + (co < a.Cout ? co : a.Cout - 1);
          my_sc[tm] = (a.bn_gamma ? a.bn_gamma[cg] : 1.0f) / sqrtf(a.bn_var[cg] + a.bn_eps);
          my_sh[tm] = (a.bn_beta ? a.bn_beta[cg] : 0.0f) - a.bn_mean[cg] * my_sc[tm];
        }
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Tw[((r & 3) + 8 * (r >> 2) + 4 * hh) * kTP + tn * 32 + j] = acc[tm][tn][r] * unscale;      // exact: 2^-(e_x + e_w)
        __builtin_amdgcn_wave_barrier();           // LDS operations of one wave complete in order
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
          const int cl = rl + RPI * i;             // channel within the tile row
          const int co = (mt0 + wm * TM + tm) * 32 + cl;
          const bool ok = pok && co < a.Cout;
          float v[PXL];
#pragma unroll
          for (int e4 = 0; e4 < PXL; e4 += 4) {
            const float4 t = *reinterpret_cast<const float4*>(Tw + cl * kTP + px + e4);
            v[e4] = t.x; v[e4 + 1] = t.y; v[e4 + 2] = t.z; v[e4 + 3] = t.w;
          }
          const int cg = grp * a.Cout + (co < a.Cout ? co : a.Cout - 1);
          if (a.bias) {
            const float b = a.bias[cg];
#pragma unroll
            for (int e = 0; e < PXL; ++e) v[e] += b;
          }
          if (a.bn_var) {
            const float sc = __shfl(my_sc[tm], cl, 64), sh = __shfl(my_sh[tm], cl, 64);
#pragma unroll
            for (int e = 0; e < PXL; ++e) v[e] = __builtin_fmaf(v[e], sc, sh);
          }
          const int o = ok ? co * iplane + c0 + q0 + px : 0;        // idle lanes read the image's first run
          auto load_run = [&](const void* base, float (&d)[PXL]) {
            const TY* __restrict__ p = static_cast<const TY*>(base) + ibase + o;
            if constexpr (OUT16) {
              const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f16x2 h2 = __builtin_bit_cast(f16x2, (unsigned)t[e]);
                d[2 * e] = (float)h2[0];
                d[2 * e + 1] = (float)h2[1];
              }
            } else {
              const float4 t = *reinterpret_cast<const float4*>(p);
              d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
            }
          };
          if (a.res) {
            float rv[PXL];
            load_run(a.res, rv);
#pragma unroll
            for (int e = 0; e < PXL; ++e) v[e] += a.res_relu ? fmaxf(rv[e], 0.0f) : rv[e];
          }
          if (a.mask_src) {
            float mv[PXL];
            load_run(a.mask_src, mv);
#pragma unroll
            for (int e = 0; e < PXL; ++e) v[e] = mv[e] > 0.0f ? v[e] : 0.0f;
          }
          if (a.relu_out) {
#pragma unroll
            for (int e = 0; e < PXL; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          if (ok) {
            if constexpr (OUT16) {
              u32x4 t;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f16x2 h2 = {(_Float16)v[2 * e], (_Float16)v[2 * e + 1]};
                t[e] = __builtin_bit_cast(unsigned, h2);
              }
              *reinterpret_cast<u32x4*>(yb + o) = t;
            } else {
              *reinterpret_cast<float4*>(yb + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int e = 0; e < PXL; ++e) ymax = amax_acc(ymax, v[e]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (a.y_amax) wave_amax_to(ymax, a.y_amax);
      return;
    }
  }
  float my_sc[TM], my_sh[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    my_sc[tm] = 1.0f;
    my_sh[tm] = 0.0f;
    if (a.bn_var) {
      const int co = (mt0 + wm * TM + tm) * 32 + (lane & 31);
      const int cg = grp * a.Cout + (co < a.Cout ? co : a.Cout - 1);
      my_sc[tm] = (a.bn_gamma ? a.bn_gamma[cg] : 1.0f) / sqrtf(a.bn_var[cg] + a.bn_eps);
      my_sh[tm] = (a.bn_beta ? a.bn_beta[cg] : 0.0f) - a.bn_mean[cg] * my_sc[tm];
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int q = qb[tn];
    const int rr = q / P, cc = q - rr * P;
    const bool pok = rr < TRv && cc < TCv;
    const int pix = pok ? (r0 + rr) * a.W + (c0 + cc) : 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int cob = (mt0 + wm * TM + tm) * 32 + 4 * (lane >> 5);
      const int o0 = cob * iplane + pix;            // offset of accumulator register 0; register r adds
                                                    // ((r & 3) + 8 (r >> 2)) planes
      const bool full = cob + 28 <= a.Cout - 4;     // all 16 channels of this lane exist (block-uniform in practice)
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[tm][tn][r] * unscale;        // exact: 2^-(e_x + e_w)
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cob + (r & 3) + 8 * (r >> 2);
          v[r] += a.bias[grp * a.Cout + (co < a.Cout ? co : a.Cout - 1)];
        }
      }
      if (a.bn_var) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int src = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          v[r] = __builtin_fmaf(v[r], __shfl(my_sc[tm], src, 64), __shfl(my_sh[tm], src, 64));
        }
      }
      if (a.res) {
        const TY* __restrict__ rb = static_cast<const TY*>(a.res) + ibase;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cob + (r & 3) + 8 * (r >> 2);
          rv[r] = (float)rb[(full || co < a.Cout) ? o0 + ((r & 3) + 8 * (r >> 2)) * iplane : pix];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += a.res_relu ? fmaxf(rv[r], 0.0f) : rv[r];
      }
      if (a.mask_src) {
        const TY* __restrict__ mb = static_cast<const TY*>(a.mask_src) + ibase;
        float mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cob + (r & 3) + 8 * (r >> 2);
          mv[r] = (float)mb[(full || co < a.Cout) ? o0 + ((r & 3) + 8 * (r >> 2)) * iplane : pix];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = mv[r] > 0.0f ? v[r] : 0.0f;
      }
      if (a.relu_out) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cob + (r & 3) + 8 * (r >> 2);
        if (pok && (full || co < a.Cout)) {
          yb[o0 + ((r & 3) + 8 * (r >> 2)) * iplane] = (TY)v[r];
          ymax = amax_acc(ymax, v[r]);
        }
      }
    }
  }
  if (a.y_amax) wave_amax_to(ymax, a.y_amax);      // the consumer's operand scale comes from this (uniform branch)
}

// ---- host side ------------------------------------------------------------------------------------------------
struct XCfg {
  int TM, TN, WM, WN;
  bool b1 = false;        // single activation stage, three blocks per CU
  bool in16 = false;      // fp16 activations: one split term in LDS
  int blockM() const { return TM * WM * 32; }
  int NQ() const { return TN * WN * 32; }
  int NT() const { return 64 * WM * WN; }
};
// Block shapes.  Every activation element a block stages is split into its two fp16 terms ONCE per block (VALU work) and
// read from LDS once per 32-channel tile row, so the split / LDS cost per MFMA falls with the number of output channels a
// block owns: with >= 256 of them a block takes 256 channels (wave tile 128 channels x 64 positions: 48 MFMAs against 18
// fragment reads per K step, round 2: 24 against 12) -- 256 x 128 positions at two blocks per CU for 1x1 kernels, 256 x 256
// positions in ONE 512-thread block per CU for k >= 3 (the haloed tile of 256 positions needs ~120 KB of LDS).
static int g_xcfg = 0;    // test / A-B hook (dvd_xconv_select): 0 auto, 1 round-2 shapes only, 2 force 256x128, 3 force 256x256,
                          // 4 round-2 shapes on the generic (pointer-addressed) main loop, 5 128 x 128 blocks with one
                          // activation stage at three blocks per CU, 6 1x1 kernels without the two-chunk loop (kOne), 7 kOne with the narrow epilogue
static XCfg pick_cfg(int M, int KS) {
  if (M <= 32) return {1, 2, 1, 4};     // 32 channels x 256 positions
  if (M <= 64) return {2, 2, 1, 4};     // 64 x 256
  if (M > 64 && g_xcfg == 5) return {2, 2, 2, 2, true};
  if (M >= 256 && g_xcfg != 1 && g_xcfg != 4) {
    if (g_xcfg == 2) return {4, 2, 2, 2};
    if (g_xcfg == 3) return {4, 2, 2, 4};
    return KS == 1 ? XCfg{4, 2, 2, 2} : XCfg{4, 2, 2, 4};
  }
  return {2, 2, 2, 2};                  // 128 x 128
}
constexpr int kXLdsBudget = 78 * 1024;    // two blocks per CU
constexpr int kXMaxFI = 3;       // register-prefetched staging; larger halos use the direct-staging kernels
constexpr int kXMaxFIDirect = 16;

struct XTile {
  int TR, TC, P, ntr, ntc, NV, npos, FI;
  size_t lds;
};
static size_t xconv_lds(const XCfg& c, int npos) {
  const int AU = c.WM * c.TM * 128, NT = c.NT();
  const int AS = (AU + NT - 1) / NT * NT;
  return ((size_t)2 * AS + (size_t)(c.b1 ? 4 : 8) * npos / (c.in16 ? 2 : 1)) * sizeof(uint4);
}
// Tile of the image per block: TR x TC outputs, TR * (TC + 2 pad) <= NQ positions; choose the split of the
// width that wastes the fewest positions, subject to the LDS budget and the staging-iteration bound.
static bool pick_tile_budget(int H, int W, int KS, const XCfg& c, XTile& best, int budget) {
  const int pad = KS / 2, NQ = c.NQ(), NT = c.NT();
  double best_eff = -1.0;
  for (int nct = 1; nct <= W; ++nct) {
    int TC = (W + nct - 1) / nct;
    if (KS == 1) TC = (TC + 7) & ~7;                     // whole 16-byte runs per lane in the wide epilogue of the 1x1 kernels
    const int P = TC + 2 * pad;
    if (P > NQ) continue;
    const int ntc = (W + TC - 1) / TC;
    const int npos = NQ + (KS - 1) * (P + 1) + 1;        // + the spare cell of the staging code
    if (xconv_lds(c, npos) > (size_t)budget) continue;
    int trmax = NQ / P;
    if (trmax > H) trmax = H;
    for (int tr0 = trmax; tr0 >= 1; --tr0) {
      const int ntr = (H + tr0 - 1) / tr0;
      const int TR = (H + ntr - 1) / ntr;               // even out the rows
      const int NV = (TR + 2 * pad) * P;
      const int FI = (2 * NV + NT - 1) / NT;
      if (FI > kXMaxFIDirect) continue;
      // useful positions per computed position, minus what the halo costs in staging work
      const double eff = (double)H * W / ((double)ntr * ntc * NQ) - 0.02 * (double)NV / (TR * TC) - 1e-6 * nct;
      if (eff > best_eff) {
        best_eff = eff;
        best = {TR, TC, P, ntr, ntc, NV, npos, FI, xconv_lds(c, npos)};
      }
      break;                                             // smaller TR only lowers the efficiency
    }
    if (TC <= 8) break;
  }
  return best_eff > -1.0;
}

static bool pick_tile(int H, int W, int KS, const XCfg& c, XTile& best) {
  // two blocks per CU where the haloed tile allows it, one block (big kernels: k >= 7; the 512-thread shape) otherwise
  if (c.NT() >= 512) return pick_tile_budget(H, W, KS, c, best, 156 * 1024);
  if (c.b1) return pick_tile_budget(H, W, KS, c, best, 52 * 1024);        // three blocks per CU, or not this shape
  return pick_tile_budget(H, W, KS, c, best, kXLdsBudget) || pick_tile_budget(H, W, KS, c, best, 156 * 1024);
}

// Stride-2 forward (XArgs::S2): TR x TC OUTPUTS per block in a pitch of P = TC + 1, four phase planes of (TR + 1) * P cells.
static bool pick_tile_s2_budget(int Ho, int Wo, const XCfg& c, XTile& best, int& S2, int budget) {
  const int NQ = c.NQ(), NT = c.NT();
  double best_eff = -1.0;
  for (int nct = 1; nct <= Wo; ++nct) {
    const int TC = (Wo + nct - 1) / nct, P = TC + 1;
    if (P > NQ) continue;
    const int ntc = (Wo + TC - 1) / TC;
    int trmax = NQ / P;
    if (trmax > Ho) trmax = Ho;
    const int ntr = (Ho + trmax - 1) / trmax;
    const int TR = (Ho + ntr - 1) / ntr;
    const int s2 = (TR + 1) * P;
    const int npos = 3 * s2 + NQ + 1;                     // + the spare cell of the staging code
    const int NV = (2 * TR + 1) * (2 * TC + 1);
    const int FI = (2 * NV + NT - 1) / NT;
    if (xconv_lds(c, npos) <= (size_t)budget && FI <= 4 * kXMaxFIDirect) {
      const double eff = (double)Ho * Wo / ((double)ntr * ntc * NQ) - 1e-6 * nct;
      if (eff > best_eff) {
        best_eff = eff;
        best = {TR, TC, P, ntr, ntc, NV, npos, FI, xconv_lds(c, npos)};
        S2 = s2;
      }
    }
    if (TC <= 4) break;
  }
  return best_eff > -1.0;
}
static bool pick_tile_s2(int Ho, int Wo, const XCfg& c, XTile& best, int& S2) {
  return pick_tile_s2_budget(Ho, Wo, c, best, S2, kXLdsBudget) || pick_tile_s2_budget(Ho, Wo, c, best, S2, 156 * 1024);
}

template <int TM, int TN, int WM, int WN, bool FAST, bool B1 = false, bool IN16 = false, bool OUT16 = false>
static int launch_fi(const XArgs& a, int FI, dim3 grid, size_t lds, hipStream_t s) {
  auto go = [&](auto kern) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
    DVD_LAUNCH_OK();
    return DVD_OK;
  };
  // (a.Cin / a.Cout: per group; ZI: three of four staged cells are structural zeros -- the NEEDED work is counted)
  const double flops = 2.0 * a.N * (double)a.G * a.Cout * a.Cin * a.T * (double)a.H * a.W * (a.ZI ? 0.25 : 1.0);
  if constexpr (TM >= 4 && !B1 && FAST && DVD_XCONV_ROLL != 0) {
    if (a.T == 1 && FI == 1 && g_xcfg != 6) {
      flops_add(DVD_FLOP_XCONV_1X1_WIDE, flops);
      return go(xconv_kernel<TM, TN, WM, WN, kXFitOne, FAST, B1, IN16, OUT16>);
    }
  }
  flops_add(TM >= 4 ? DVD_FLOP_XCONV_WIDE : (WM == 2 ? DVD_FLOP_XCONV_128 : DVD_FLOP_XCONV_SMALL), flops);
  switch (FI) {
    case 1: return go(xconv_kernel<TM, TN, WM, WN, 1, FAST, B1, IN16, OUT16>);
    case 2: return go(xconv_kernel<TM, TN, WM, WN, 2, FAST, B1, IN16, OUT16>);
    case 3: return go(xconv_kernel<TM, TN, WM, WN, 3, FAST, B1, IN16, OUT16>);
    default: return go(xconv_kernel<TM, TN, WM, WN, 0, FAST, B1, IN16, OUT16>);
  }
}

// 32-row tiles per group in the packed buffer: M padded to the largest block height any shape of pick_cfg may use for it
// (the packing must not depend on the kernel size or on the A/B hook)
static int xconv_mtiles(int M) {
  const int bm = M <= 32 ? 32 : (M <= 64 ? 64 : (M >= 256 ? 256 : 128));
  return (M + bm - 1) / bm * (bm / 32);
}

}  // namespace dvd

extern "C" {

// Channel counts in this ABI are TOTALS; a grouped convolution (groups > 1) has Cin / groups inputs and
// Cout / groups outputs per group and a weight [Cout][Cin / groups][k][k], like nn.Conv2d.
size_t dvd_xconv_packed_bytes(int Cout, int Cin, int KS, int groups, int transposed) {
  if (Cout <= 0 || Cin <= 0 || KS <= 0 || !(KS & 1) || groups <= 0 || Cout % groups || Cin % groups) return 0;
  const int co = Cout / groups, ci = Cin / groups;
  const int M = transposed ? ci : co, K = transposed ? co : ci;
  return ((size_t)dvd::kXHeader + (size_t)groups * dvd::xconv_mtiles(M) * ((K + 15) / 16) * KS * KS * 2 * 64) * sizeof(uint4);
}

static int xconv_pack_impl(const float* w, void* packed, int Cout, int Cin, int KS, int groups, int transposed,
                           const float* gamma, const float* var, float eps, dvd_stream_t stream) {
  DVD_REQUIRE(w && packed, "xconv_pack: null pointer");
  DVD_REQUIRE(Cout > 0 && Cin > 0 && KS > 0 && (KS & 1) && KS <= 11 && groups > 0 && Cout % groups == 0 && Cin % groups == 0,
              "xconv_pack: bad shape Cout=%d Cin=%d KS=%d groups=%d", Cout, Cin, KS, groups);
  const int co = Cout / groups, ci = Cin / groups;
  const int M = transposed ? ci : co, K = transposed ? co : ci;
  const int mtiles = dvd::xconv_mtiles(M), nkc = (K + 15) / 16, T = KS * KS;
  const long long total = (long long)groups * mtiles * nkc * T * 64;
  // (bytes: the fp32 weights read twice -- maximum, then packing -- and the packed fragments written: 16 bytes per lane)
  dvd::bytes_add(DVD_BYTES_PACK, 8.0 * (double)Cout * ci * T + 16.0 * (double)total);
  // max |A| (BatchNorm scale included) -> the power-of-two operand scale of this packing: per-block partial maxima into the
  // header, reduced by the pack kernel
  const int row_len = ci * T;
  int npart;
  if (row_len % 4 == 0 && reinterpret_cast<uintptr_t>(w) % 16 == 0) {
    const int cpr = (row_len + 1023) / 1024;
    const long long nb = ((long long)Cout * cpr + dvd::kWamaxWaves - 1) / dvd::kWamaxWaves;
    npart = (int)(nb < dvd::kXPartials ? nb : dvd::kXPartials);
    hipLaunchKernelGGL(dvd::xconv_wamax4_kernel, dim3((unsigned)npart), dim3(64 * dvd::kWamaxWaves), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<float*>(packed), Cout, row_len, cpr, gamma, var, eps);
    DVD_LAUNCH_OK();
  } else {
    npart = Cout < dvd::kXPartials ? Cout : dvd::kXPartials;
    hipLaunchKernelGGL(dvd::xconv_wamax_kernel, dim3((unsigned)npart), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<float*>(packed), Cout, row_len, gamma, var, eps);
    DVD_LAUNCH_OK();
  }
  hipLaunchKernelGGL(dvd::xconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, static_cast<uint4*>(packed), co, ci, T, transposed ? 1 : 0,
                     mtiles, nkc, groups, gamma, var, eps, npart);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_xconv_pack(const float* w, void* packed, int Cout, int Cin, int KS, int groups, int transposed, dvd_stream_t stream) {
  return xconv_pack_impl(w, packed, Cout, Cin, KS, groups, transposed, nullptr, nullptr, 0.0f, stream);
}

int dvd_xconv_pack_scaled(const float* w, void* packed, int Cout, int Cin, int KS, int groups, int transposed,
                          const float* bn_gamma, const float* bn_var, float bn_eps, dvd_stream_t stream) {
  DVD_REQUIRE(bn_var, "xconv_pack_scaled: null variance");
  return xconv_pack_impl(w, packed, Cout, Cin, KS, groups, transposed, bn_gamma, bn_var, bn_eps, stream);
}

size_t dvd_xconv_pack_table_bytes(int n) { return n > 0 ? (size_t)n * sizeof(dvd::XPackDev) : 0; }

int dvd_xconv_pack_many(const dvd_xpack_item* items, int n, void* table, size_t table_bytes, int upload, dvd_stream_t stream) {
  DVD_REQUIRE(items && table && n > 0, "xconv_pack_many: bad arguments");
  DVD_REQUIRE(table_bytes >= dvd_xconv_pack_table_bytes(n), "xconv_pack_many: table %zu < %zu bytes", table_bytes,
              dvd_xconv_pack_table_bytes(n));
  std::vector<dvd::XPackDev> host((size_t)n);
  unsigned long long blocks = 0;
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const dvd_xpack_item& q = items[i];
    DVD_REQUIRE(q.w && q.packed, "xconv_pack_many: item %d: null pointer", i);
    DVD_REQUIRE(q.Cout > 0 && q.Cin > 0 && q.KS > 0 && (q.KS & 1) && q.KS <= 11 && q.groups > 0 && q.Cout % q.groups == 0 &&
                    q.Cin % q.groups == 0,
                "xconv_pack_many: item %d: bad shape Cout=%d Cin=%d KS=%d groups=%d", i, q.Cout, q.Cin, q.KS, q.groups);
    DVD_REQUIRE(!q.gamma || q.var, "xconv_pack_many: item %d: BatchNorm scale without a variance", i);
    dvd::XPackDev& d = host[(size_t)i];
    d.w = q.w;
    d.packed = static_cast<uint4*>(q.packed);
    d.gamma = q.gamma;
    d.var = q.var;
    d.eps = q.eps;
    d.co = q.Cout / q.groups;
    d.ci = q.Cin / q.groups;
    d.T = q.KS * q.KS;
    d.transposed = q.transposed ? 1 : 0;
    const int M = d.transposed ? d.ci : d.co, K = d.transposed ? d.co : d.ci;
    d.mtiles = dvd::xconv_mtiles(M);
    d.nkc = (K + 15) / 16;
    d.G = q.groups;
    d.rows = q.Cout;
    d.row_len = d.ci * d.T;
    d.block0 = (unsigned)blocks;
    d.pad_ = 0;
    const long long total = (long long)d.G * d.mtiles * d.nkc * d.T * 64;
    blocks += (unsigned long long)((total + 255) / 256);
    bytes += 8.0 * (double)q.Cout * d.ci * d.T + 16.0 * (double)total;
  }
  DVD_REQUIRE(blocks < (1ull << 31), "xconv_pack_many: too many blocks");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (upload) {      // (not capturable: call it outside a graph capture; replays and later calls pass upload = 0)
    DVD_HIP_OK(hipMemcpyAsync(table, host.data(), (size_t)n * sizeof(dvd::XPackDev), hipMemcpyHostToDevice, st));
    DVD_HIP_OK(hipStreamSynchronize(st));
  }
  dvd::bytes_add(DVD_BYTES_PACK, bytes);
  const dvd::XPackDev* t = static_cast<const dvd::XPackDev*>(table);
  hipLaunchKernelGGL(dvd::xconv_wamax_many_kernel, dim3(dvd::kManyPartials, (unsigned)n), dim3(64 * dvd::kWamaxWaves), 0, st, t);
  DVD_LAUNCH_OK();
  hipLaunchKernelGGL(dvd::xconv_pack_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t, n);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_xconv_select(int cfg) {
  DVD_REQUIRE(cfg >= 0 && cfg <= 7, "xconv_select: cfg %d", cfg);
  dvd::g_xcfg = cfg;
  return DVD_OK;
}

static int xconv_fwd_impl(const void* x, const float* x_amax, const void* packed, const float* bias, const void* residual,
                          const void* mask_src, const dvd_bn_params* bn, void* y, float* y_amax, int N, int Cin_total,
                          int Cout_total, int H, int W, int KS, int groups, int flags, int in16, int out16, dvd_stream_t stream) {
  DVD_REQUIRE(x && packed && y, "xconv: null pointer");
  DVD_REQUIRE(in16 || x_amax, "xconv: the input's max|x| scalar is missing (dvd_amax, or the producer's y_amax)");
  DVD_REQUIRE(N > 0 && Cin_total > 0 && Cout_total > 0 && H > 0 && W > 0, "xconv: bad shape N=%d Cin=%d Cout=%d H=%d W=%d", N,
              Cin_total, Cout_total, H, W);
  DVD_REQUIRE(groups > 0 && Cin_total % groups == 0 && Cout_total % groups == 0, "xconv: %d groups do not divide the channels",
              groups);
  DVD_REQUIRE(KS > 0 && (KS & 1) && KS <= 11, "xconv: kernel size %d (odd sizes up to 11)", KS);
  DVD_REQUIRE(N <= 65535, "xconv: too many images for the grid");
  DVD_REQUIRE((long long)H * W * (long long)(Cin_total > Cout_total ? Cin_total : Cout_total) < (1ll << 31),
              "xconv: image too large for 32-bit offsets");
  DVD_REQUIRE(in16 || !out16, "xconv: fp32 input with fp16 output is not a configuration of this kernel");
  // flags bit 3: stride-2 forward (x is [N, Cin, H, W], y [N, Cout, (H + 1) / 2, (W + 1) / 2]); bit 4: backward-data of a
  // stride-2 convolution (x is the [N, Cin, (H + 1) / 2, (W + 1) / 2] gradient of the strided output, y [N, Cout, H, W])
  const int s2 = (flags >> 3) & 1, zi = (flags >> 4) & 1;
  DVD_REQUIRE(!(s2 && zi), "xconv: stride-2 forward and stride-2 backward-data are different launches");
  DVD_REQUIRE(!(s2 || zi) || KS == 3, "xconv: the strided forms exist for 3x3 kernels (got %d)", KS);
  const int Hq = (H + 1) / 2, Wq = (W + 1) / 2;
  const int Cin = Cin_total / groups, Cout = Cout_total / groups;
  // buffer-addressed main loop: whole 16-channel chunks, 31-bit byte offsets inside one image's input channels and
  // inside the packed weights of one block row
  const bool fast = (Cin % 16 == 0) && ((long long)Cin * H * W * 4 < (1ll << 31)) &&
                    ((long long)8 * ((Cin + 15) / 16) * KS * KS * 2048 < (1ll << 31)) && dvd::g_xcfg != 4;
  DVD_REQUIRE(fast || !in16, "xconv: fp16 activations need input channels in multiples of 16 (got %d per group)", Cin);
  DVD_REQUIRE(!(s2 || zi) || fast, "xconv: the strided forms need input channels in multiples of 16 (got %d per group)", Cin);
  // (stride 2: four input cells per output -- blocks of at most 128 channels, fp32 activations on ONE activation stage)
  dvd::XCfg c = dvd::pick_cfg((fast && !s2) ? Cout : (Cout < 128 ? Cout : 128), KS);
  if (!fast) c.b1 = false;   // the wide shapes exist as FAST kernels only
  if (s2) c.b1 = true;
  if (in16) c.b1 = false;
  c.in16 = in16 != 0;
  int Hh = H, Ww = W;
  if (KS == 1) {           // no spatial structure: one row of H * W positions
    Hh = 1;
    Ww = H * W;
  }
  dvd::XTile t;
  int S2 = 0;
  if (s2) {
    Hh = Hq;
    Ww = Wq;
    DVD_REQUIRE(dvd::pick_tile_s2(Hh, Ww, c, t, S2), "xconv: no stride-2 tile of a %dx%d image fits the LDS", H, W);
  } else {
    if (c.b1 && !dvd::pick_tile(Hh, Ww, KS, c, t)) c.b1 = false;     // (large kernels: the haloed tile needs more LDS)
    DVD_REQUIRE(dvd::pick_tile(Hh, Ww, KS, c, t), "xconv: no tile of a %dx%d image with a %dx%d kernel fits the LDS", H, W, KS, KS);
  }
  dvd::XArgs a;
  a.x = x;
  a.wp = static_cast<const uint4*>(packed);
  a.bias = bias;
  a.res = residual;
  a.mask_src = mask_src;
  a.bn_gamma = bn ? bn->gamma : nullptr;
  a.bn_beta = bn ? bn->beta : nullptr;
  a.bn_mean = bn ? bn->mean : nullptr;
  a.bn_var = bn ? bn->var : nullptr;
  a.bn_eps = bn ? bn->eps : 0.0f;
  DVD_REQUIRE(!bn || (bn->mean && bn->var), "xconv: BatchNorm statistics missing");
  a.y = y;
  a.x_amax = x_amax;
  a.y_amax = y_amax;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = Hh; a.W = Ww;
  a.G = groups; a.mtiles = dvd::xconv_mtiles(Cout);
  a.KS = KS; a.pad = KS / 2; a.T = KS * KS;
  a.TR = t.TR; a.TC = t.TC; a.P = t.P; a.ntr = t.ntr; a.ntc = t.ntc; a.NV = t.NV;
  a.nkc = (Cin + 15) / 16; a.npos = t.npos; a.nfi = t.FI;
  a.Hi = s2 ? H : (zi ? Hq : Hh); a.Wi = s2 ? W : (zi ? Wq : Ww); a.S2 = S2; a.ZI = zi;
  a.relu_in = flags & 1; a.relu_out = (flags >> 1) & 1; a.res_relu = (flags >> 2) & 1;
  const int mblocks = (Cout + c.blockM() - 1) / c.blockM();
  a.mbpg = mblocks;
  DVD_REQUIRE((long long)mblocks * groups <= 65535, "xconv: too many channel blocks");
  const dim3 grid(t.ntr * t.ntc, mblocks * groups, N);
  // wide epilogue of the 1x1 kernels: whole 16-byte runs of pixels behind 16-byte aligned pointers
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec_out = (KS == 1 && ((long long)H * W) % 8 == 0 && t.TC % 8 == 0 && al16(y) && al16(residual) && al16(mask_src) &&
               dvd::g_xcfg != 7) ? 1 : 0;
  size_t lds = t.lds;
  if (a.vec_out && lds < (size_t)c.NT() / 64 * 32 * 68 * 4) lds = (size_t)c.NT() / 64 * 32 * 68 * 4;   // one tile row per wave
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DVD_XGO(TM_, TN_, WM_, WN_)                                                                                      \
  do {                                                                                                                   \
    if (in16 && out16) return dvd::launch_fi<TM_, TN_, WM_, WN_, true, false, true, true>(a, t.FI, grid, lds, s);         \
    if (in16) return dvd::launch_fi<TM_, TN_, WM_, WN_, true, false, true, false>(a, t.FI, grid, lds, s);                 \
    return dvd::launch_fi<TM_, TN_, WM_, WN_, true>(a, t.FI, grid, lds, s);                                               \
  } while (0)
  if (fast) {
    if (c.TM == 4 && c.WN == 4) DVD_XGO(4, 2, 2, 4);
    if (c.TM == 4) DVD_XGO(4, 2, 2, 2);
    if (c.WM == 2 && c.b1) return dvd::launch_fi<2, 2, 2, 2, true, true>(a, t.FI, grid, lds, s);
    if (c.TM == 2 && c.WM == 1 && c.b1) return dvd::launch_fi<2, 2, 1, 4, true, true>(a, t.FI, grid, lds, s);     // (stride 2 only)
    if (c.TM == 1 && c.b1) return dvd::launch_fi<1, 2, 1, 4, true, true>(a, t.FI, grid, lds, s);
    if (c.WM == 2) DVD_XGO(2, 2, 2, 2);
    if (c.TM == 2) DVD_XGO(2, 2, 1, 4);
    DVD_XGO(1, 2, 1, 4);
  }
#undef DVD_XGO
  if (c.WM == 2) return dvd::launch_fi<2, 2, 2, 2, false>(a, t.FI, grid, lds, s);
  if (c.TM == 2) return dvd::launch_fi<2, 2, 1, 4, false>(a, t.FI, grid, lds, s);
  return dvd::launch_fi<1, 2, 1, 4, false>(a, t.FI, grid, lds, s);
}

int dvd_xconv_fwd(const float* x, const float* x_amax, const void* packed, const float* bias, const float* residual,
                  const float* mask_src, const dvd_bn_params* bn, float* y, float* y_amax, int N, int Cin_total, int Cout_total,
                  int H, int W, int KS, int groups, int flags, dvd_stream_t stream) {
  return xconv_fwd_impl(x, x_amax, packed, bias, residual, mask_src, bn, y, y_amax, N, Cin_total, Cout_total, H, W, KS, groups,
                        flags, 0, 0, stream);
}

int dvd_xconv_fwd_h(const void* x, const void* packed, const float* bias, const void* residual, const void* mask_src,
                    const dvd_bn_params* bn, void* y, float* y_amax, int N, int Cin_total, int Cout_total, int H, int W, int KS,
                    int groups, int flags, int out_f16, dvd_stream_t stream) {
  return xconv_fwd_impl(x, nullptr, packed, bias, residual, mask_src, bn, y, y_amax, N, Cin_total, Cout_total, H, W, KS, groups,
                        flags, 1, out_f16 ? 1 : 0, stream);
}

}  // extern "C"
