// Backward-weight of the dense 3x3 stride-1 "same" convolutions on the 16-bit matrix cores with fp32-class
// accuracy (the two-term fp16 split of csrc/dvd_split.h: both operands scaled by a power of two from their tensors'
// max|.| scalars, three partial products per fp32 product, fp32 accumulation), deterministic (no atomics):
//   dW[co][ci][ky][kx] = sum_{n, r, c} gy[n][co][r][c] * act(x)[n][ci][r + ky - 1][c + kx - 1]
//
// What it replaces (reference, /root/reference): the autograd weight gradient of the 3x3 nn.Conv2d of the MiDaS
// decoder (third_party/midas_blocks.py:102-168, MiDaS.py:186-195); MIOpen's igemm_wrw (atomics, 113 TF/s) and the
// exact-fp32 kernel of csrc/xwgrad.hip (86 TF/s: the fp32 MFMA rate is the ceiling there).
//
// Mapping.  GEMM M = output channel, N = input channel, K = pixels.  v_mfma_f32_32x32x16_f16: lane l holds
// A[co = l&31][8 consecutive pixels, run (l>>5)] = gy and B[the same 8 pixels shifted by the tap][ci = l&31] = x.
// Pixels are contiguous in NCHW, so both operands are read as 16-byte runs along x; the run of tap kx = 1 is a
// cell of the LDS row, the runs of kx = 0 / 2 start one pixel earlier / later and are assembled from the cell and
// one dword of its neighbour with v_alignbit_b32 (five of them give both shifted fragments of a term).
// A block owns 64 output x 64 input channels and walks DOWN a 64-pixel-wide column strip of an image, one row
// per step: the gy row (64 channels x 64 pixels) and ONE new x row (64 channels x 80 pixels: the strip, one
// 8-pixel cell left and right) are staged per step -- the x rows a step needs live in a rolling buffer of four --
// scaled and split into the two fp16 terms on the way in; the next step's rows are requested before the MFMAs of the
// current one.  Wave (pm, pn, ky) of the 12 keeps the three accumulators of kernel row ky for its 32 x 32 pair.
// Every block writes its partial sums; xwgrad3_reduce_kernel adds them in slice order.
// Round 4: the row step exists in two instantiations -- FW (buffer loads, no branch, the staging of item i
// between the MFMAs of K step i; taken whenever rows start dword-aligned) and the round-3 one (per-element guarded loads ahead
// of the MFMAs; what is left for it: fp16 rows of odd width); xwgrad3g_kernel is the same walk on 32 x 32 channel blocks of
// three waves for grouped layers with at most 32 channels per group; xwgrad1b_kernel (1x1, wide layers) has an FW form too.
#include <type_traits>

#include "dvd_split.h"

namespace dvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

static int g_w1_variant = 0;      // test / A-B hook (dvd_xwgrad_select): 0 auto, 1 always the 128 x 128 blocks
static int g_w3_variant = 0;      // 1 (hook value 2): the 3x3 kernel's round-3 row step (staging ahead of the MFMAs) on every shape

struct Wg3Args {
  const void* __restrict__ x;         // float, or _Float16 in the H16 kernels (fp16 activation storage, BASELINE configs[4])
  const void* __restrict__ gy;
  const float* __restrict__ x_amax;   // device scalars: max|x|, max|gy| (or upper bounds) of the whole tensors (fp32 kernels)
  const float* __restrict__ g_amax;
  const float* __restrict__ out_scale;  // H16: device scalar the result is multiplied by (1 / loss scale of the fp16 gradients), or null
  float* __restrict__ partial;   // [S][9][Cout][Cin]
  int N, Cin, Cout, H, W;        // Cin / Cout per group
  int G, nco;                    // groups, output-channel blocks per group
  int nstrips, RS, nrseg, S;     // column strips per image, rows per work item, row segments per image, slices
  int relu_in;
};


// ---- fp16 operands (H16 kernels): both tensors are stored as _Float16 and ARE the matrix operands -- one term each, no scale,
// no split: the staging is a copy and a product is ONE MFMA instead of three.  The fp16 gradients carry the step's loss scale;
// the kernels multiply their result by *out_scale (its inverse).
template <bool H16>
struct Quad {
  typedef float4 type;
};
template <>
struct Quad<true> {
  typedef uint2 type;
};
__device__ __forceinline__ unsigned relu_h2(unsigned v) {
  const f16x2 z = {(_Float16)0.0f, (_Float16)0.0f};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, v), z));
}
// elements p[px .. px + 3] of a row of n elements (zero outside [0, n)); vec: the quad is inside the row and 4-element aligned
template <bool H16>
__device__ __forceinline__ typename Quad<H16>::type load_quad(const void* row, int px, int n, bool vec, bool relu) {
  if constexpr (H16) {
    const unsigned short* p = static_cast<const unsigned short*>(row);
    uint2 v = make_uint2(0u, 0u);
    if (vec) {
      v = *reinterpret_cast<const uint2*>(p + px);
    } else {
      unsigned e0 = (px >= 0 && px < n) ? p[px] : 0u, e1 = (px + 1 >= 0 && px + 1 < n) ? p[px + 1] : 0u;
      unsigned e2 = (px + 2 >= 0 && px + 2 < n) ? p[px + 2] : 0u, e3 = (px + 3 >= 0 && px + 3 < n) ? p[px + 3] : 0u;
      v = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
    }
    if (relu) v = make_uint2(relu_h2(v.x), relu_h2(v.y));
    return v;
  } else {
    const float* p = static_cast<const float*>(row);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
      v = *reinterpret_cast<const float4*>(p + px);
    } else {
      if (px >= 0 && px < n) v.x = p[px];
      if (px + 1 >= 0 && px + 1 < n) v.y = p[px + 1];
      if (px + 2 >= 0 && px + 2 < n) v.z = p[px + 2];
      if (px + 3 >= 0 && px + 3 < n) v.w = p[px + 3];
    }
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    return v;
  }
}
template <bool H16>
__device__ __forceinline__ typename Quad<H16>::type zero_quad() {
  if constexpr (H16) return make_uint2(0u, 0u);
  else return make_float4(0.f, 0.f, 0.f, 0.f);
}
// split (fp32) or copy (fp16) a staged quad into the LDS row(s): 8 bytes per term
template <bool H16>
__device__ __forceinline__ void store_quad(unsigned char* dst, int tstride, const typename Quad<H16>::type& q, float sc) {
  if constexpr (H16) {
    *reinterpret_cast<uint2*>(dst) = q;
  } else {
    unsigned h0, l0, h1, l1;
    split_pair_f16(q.x * sc, q.y * sc, h0, l0);
    split_pair_f16(q.z * sc, q.w * sc, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + tstride) = make_uint2(l0, l1);
  }
}
// fp16 rows are staged 8 pixels (16 bytes) per item: with 4-pixel items the H16 kernels issued twice the loads of the fp32
// kernels per byte and sat on the load-instruction rate (3x3 256 -> 256: the matrix pipe 24 % busy at one MFMA per product).
// p[px .. px + 7] of a row of n elements (zero outside); a8: rows are 16-byte aligned (n % 8 == 0), a4: 8-byte aligned
__device__ __forceinline__ uint4 load_oct_h(const void* row, int px, int n, bool a8, bool a4, bool relu) {
  const unsigned short* p = static_cast<const unsigned short*>(row);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (a8 && px >= 0 && px + 7 < n) {
    v = *reinterpret_cast<const uint4*>(p + px);
  } else if (a4) {
    if (px >= 0 && px + 3 < n) {
      const uint2 t = *reinterpret_cast<const uint2*>(p + px);
      v.x = t.x;
      v.y = t.y;
    }
    if (px + 4 >= 0 && px + 7 < n) {
      const uint2 t = *reinterpret_cast<const uint2*>(p + px + 4);
      v.z = t.x;
      v.w = t.y;
    }
  } else {
    unsigned e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = (px + i >= 0 && px + i < n) ? p[px + i] : 0u;
    v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  }
  if (relu) v = make_uint4(relu_h2(v.x), relu_h2(v.y), relu_h2(v.z), relu_h2(v.w));
  return v;
}

constexpr int kW3Strip = 64;                  // pixels per row step
constexpr int kW3GPitch = 128 + 16;           // bytes per gy row in LDS (64 fp16 + pad: conflict-free 16-byte reads across rows)
constexpr int kW3XPitch = 160 + 16;           // bytes per x row in LDS (80 fp16 + pad)
constexpr int kW3CB = 64;                     // channels per block, both operands
constexpr int kW3NT = 768;                    // 12 waves: 2 x 2 tile pairs x 3 kernel rows
constexpr int kW3LdsBytes = 2 * 2 * kW3CB * kW3GPitch + 2 * kW3CB * 4 * kW3XPitch;   // + 16 spare bytes (idle staging items)

// FW (host: rows of fp16 elements start dword-aligned, i.e. W even; any W for fp32): every staging item is a 16-byte run of one
// row (a run crossing the row's end is loaded whole and its tail zeroed on the way to LDS, as in xwgrad3g_kernel) or nothing,
// so the loads are raw BUFFER loads with a per-thread offset computed once per work item (channels past the tensor
// and columns outside the image fall out of the resource's range and read 0) -- no branch in the row step.  With the body one
// basic block, the staging work of item i (split + LDS store of the row loaded a step ago, request of the row two steps ahead)
// sits between the MFMAs of K step i: round 3 ran `barrier | all staging | all MFMAs` in every wave, the 12 waves in lock step
// behind the barrier, and the counters showed matrix pipe and VALU taking turns (55 % busy at 5.5 VALU per MFMA).
template <bool H16, bool FW>
__global__ __launch_bounds__(kW3NT) void xwgrad3_kernel(const Wg3Args a) {
  constexpr int EB = H16 ? 2 : 4;                // bytes per element in HBM
  constexpr int NTERM = H16 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  // sG [buffer 2][term 2][co 64][kW3GPitch], sX [term 2][slot 4][ci 64][kW3XPitch] (slot-major: a channel stride of
  // 44 dwords keeps the 16-byte fragment reads of 32 channels conflict free; 4 x 44 did not)
  unsigned char* sG = smem3;
  unsigned char* sX = smem3 + 2 * 2 * kW3CB * kW3GPitch;
  const float sx = H16 ? 1.0f : pow2_scale(a.x_amax[0]), sg = H16 ? 1.0f : pow2_scale(a.g_amax[0]);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ky = wave % 3, pn = (wave / 3) & 1, pm = wave / 6;
  const int grp = blockIdx.z / a.nco;                      // group of a grouped convolution (0 for dense)
  const int co0 = (blockIdx.z - grp * a.nco) * kW3CB, ci0 = blockIdx.y * kW3CB;
  const size_t plane = (size_t)a.H * a.W;
  const int items = a.N * a.nstrips * a.nrseg;

  // staging assignment: a thread stages QW consecutive pixels of one channel row (fp32: 4 = 16 bytes; fp16: 8 = 16 bytes).
  //   fp32: gy row 64 channels x 16 items, x row 64 x 20 -> 3 per thread;  fp16: 64 x 8 and 64 x 10 -> 2 per thread
  constexpr int QW = H16 ? 8 : 4, GPR = 64 / QW, XPR = 80 / QW;
  constexpr int GQ = kW3CB * GPR, XQ = kW3CB * XPR, NQ = (GQ + XQ + kW3NT - 1) / kW3NT;
  typedef std::conditional_t<H16, uint4, float4> StgT;
  StgT stg[NQ];
  const bool wvec = (a.W & 3) == 0, wvec8 = (a.W & 7) == 0;
  auto zero_item = [&]() -> StgT {
    if constexpr (H16) return make_uint4(0u, 0u, 0u, 0u);
    else return make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto load_item = [&](const unsigned char* row, int px, bool relu) -> StgT {
    if constexpr (H16) return load_oct_h(row, px, a.W, wvec8, wvec, relu);
    else return load_quad<false>(row, px, a.W, wvec && px >= 0 && px + 3 < a.W, relu);
  };
  auto stage_load = [&](int n, int c0, int r, bool with_g) {   // gy row r (if with_g) and x row r + 1
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = i * kW3NT + tid;
      StgT v = zero_item();
      if (q < GQ) {
        const int ch = q / GPR, px = c0 + (q - ch * GPR) * QW;
        if (with_g && (co0 + ch) < a.Cout && r < a.H) {
          const unsigned char* p = static_cast<const unsigned char*>(a.gy) +
                                   ((((size_t)n * a.G + grp) * a.Cout + co0 + ch) * plane + (size_t)r * a.W) * EB;
          v = load_item(p, px, false);
        }
      } else if (q < GQ + XQ) {
        const int qq = q - GQ;
        const int ch = qq / XPR, px = c0 - 8 + (qq - ch * XPR) * QW;
        const int row = r + 1;
        if ((ci0 + ch) < a.Cin && row >= 0 && row < a.H) {
          const unsigned char* p = static_cast<const unsigned char*>(a.x) +
                                   ((((size_t)n * a.G + grp) * a.Cin + ci0 + ch) * plane + (size_t)row * a.W) * EB;
          v = load_item(p, px, a.relu_in != 0);
        }
      }
      stg[i] = v;
    }
  };
  auto stage_store = [&](int xslot, int gbuf) {      // split (fp32) or copy (fp16) and write: 4 pixels = 8 bytes per term
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = i * kW3NT + tid;
      if (q >= GQ + XQ) continue;
      const float sc = q < GQ ? sg : sx;
      unsigned char* dst;
      int tstride;
      if (q < GQ) {
        const int ch = q / GPR;
        dst = sG + gbuf * (2 * kW3CB * kW3GPitch) + ch * kW3GPitch + (q - ch * GPR) * (QW * 2);
        tstride = kW3CB * kW3GPitch;
      } else {
        const int qq = q - GQ, ch = qq / XPR;
        dst = sX + (xslot * kW3CB + ch) * kW3XPitch + (qq - ch * XPR) * (QW * 2);
        tstride = kW3CB * 4 * kW3XPitch;
      }
      if constexpr (H16) *reinterpret_cast<uint4*>(dst) = stg[i];
      else store_quad<false>(dst, tstride, stg[i], sc);
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int half = lane >> 5;
  const unsigned char* ga = sG + (pm * 32 + (lane & 31)) * kW3GPitch + half * 16;
  const unsigned char* xa = sX + (pn * 32 + (lane & 31)) * kW3XPitch + 16 + half * 16;   // cell 1 + half of slot 0

  // ---- FW staging (see the kernel's header): per-thread constants of the NQ staging items
  bool f_isg[NQ];          // wave-uniform: the item is a gy run (else an x run)
  int f_ch[NQ], f_pq[NQ], f_lds[NQ];
  if constexpr (FW) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = i * kW3NT + tid;
      f_isg[i] = __builtin_amdgcn_readfirstlane(q < GQ ? 1 : 0) != 0;          // GQ is a multiple of 64
      const int qq = f_isg[i] ? q : q - GQ;
      const int pr = f_isg[i] ? GPR : XPR;
      f_ch[i] = qq / pr;
      const int cell = qq - f_ch[i] * pr;
      f_pq[i] = cell * QW - (f_isg[i] ? 0 : 8);
      // items past the two rows (fp16: the last six waves of item 1) load nothing and store into the spare 16 bytes
      f_lds[i] = q >= GQ + XQ ? kW3LdsBytes
                              : (f_isg[i] ? f_ch[i] * kW3GPitch : (int)(sX - smem3) + f_ch[i] * kW3XPitch) + cell * (QW * 2);
      if (q >= GQ + XQ) f_ch[i] = 1 << 20;                                       // beyond any resource's range
    }
  }
  u32x4 fstg[NQ];
  int f_voff[NQ];
  float f_sc[H16 ? 1 : NQ][4];          // fp32: the operand scale per element of the run, 0 past the row's end (xwgrad3g_kernel)
  unsigned f_msk[H16 ? NQ : 1][4];      // fp16: all-ones / zero halves
  __amdgpu_buffer_rsrc_t srdX, srdG;
  const float relu_lo = a.relu_in ? 0.0f : -__builtin_inff();
  const unsigned relu_lo_h = a.relu_in ? 0u : 0xfc00fc00u;                      // packed halves: 0 | -inf
  auto f_load = [&](int i, int rg, bool with_g) {                               // item i of (gy row rg, x row rg + 1)
    const int row = f_isg[i] ? rg : rg + 1;
    const bool ok = row >= 0 && row < a.H && (with_g || !f_isg[i]);              // wave-uniform
    const int vo = ok ? f_voff[i] : (int)0x80000000;
    const int so = ok ? row * a.W * EB : 0;
    fstg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(f_isg[i] ? srdG : srdX, vo, so, 0));
  };
  auto f_store = [&](int i, int xslot, int gbuf) {
    unsigned char* dst = smem3 + f_lds[i] + (f_isg[i] ? gbuf * (2 * kW3CB * kW3GPitch) : xslot * (kW3CB * kW3XPitch));
    if constexpr (H16) {
      u32x4 v = fstg[i];
      const f16x2 lo = __builtin_bit_cast(f16x2, f_isg[i] ? 0xfc00fc00u : relu_lo_h);   // ReLU of the activations (no-op against -inf)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, (unsigned)v[j]), lo)) & f_msk[i][j];
      *reinterpret_cast<u32x4*>(dst) = v;
    } else {
      const float lo = f_isg[i] ? -__builtin_inff() : relu_lo;
      const float v0 = fmaxf(__uint_as_float(fstg[i][0]), lo) * f_sc[i][0], v1 = fmaxf(__uint_as_float(fstg[i][1]), lo) * f_sc[i][1];
      const float v2 = fmaxf(__uint_as_float(fstg[i][2]), lo) * f_sc[i][2], v3 = fmaxf(__uint_as_float(fstg[i][3]), lo) * f_sc[i][3];
      unsigned h0, l0, h1, l1;
      split_pair_f16(v0, v1, h0, l0);
      split_pair_f16(v2, v3, h1, l1);
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + (f_isg[i] ? kW3CB * kW3GPitch : kW3CB * 4 * kW3XPitch)) = make_uint2(l0, l1);
    }
  };

  for (int item = blockIdx.x; item < items; item += a.S) {
    const int n = item / (a.nstrips * a.nrseg);
    const int rem = item - n * (a.nstrips * a.nrseg);
    const int strip = rem / a.nrseg, seg = rem - strip * a.nrseg;
    const int c0 = strip * kW3Strip, r0 = seg * a.RS;
    const int r1 = (r0 + a.RS) < a.H ? (r0 + a.RS) : a.H;
    if constexpr (FW) {
      // this image's 64 channels of either tensor as buffer resources: channels past the tensor are out of range
      const int nci = (a.Cin - ci0) < kW3CB ? (a.Cin - ci0) : kW3CB, nco = (a.Cout - co0) < kW3CB ? (a.Cout - co0) : kW3CB;
      const unsigned char* xb = static_cast<const unsigned char*>(a.x) + (((size_t)n * a.G + grp) * a.Cin + ci0) * plane * EB;
      const unsigned char* gb = static_cast<const unsigned char*>(a.gy) + (((size_t)n * a.G + grp) * a.Cout + co0) * plane * EB;
      srdX = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(xb)), 0, nci * (int)plane * EB, 0x00020000);
      srdG = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(gb)), 0, nco * (int)plane * EB, 0x00020000);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        // a run that crosses the end of its row is loaded anyway and the elements past the row are zeroed on the way to LDS
        const int px = c0 + f_pq[i];
        const int nv = px < 0 ? 0 : ((a.W - px) < QW ? (a.W - px) : QW);
        f_voff[i] = (nv > 0 && f_ch[i] < kW3CB) ? (f_ch[i] * (int)plane + px) * EB : (int)0x80000000;
        if constexpr (H16) {
#pragma unroll
          for (int j = 0; j < 4; ++j) f_msk[i][j] = (2 * j < nv ? 0xffffu : 0u) | (2 * j + 1 < nv ? 0xffff0000u : 0u);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) f_sc[i][e] = e < nv ? (f_isg[i] ? sg : sx) : 0.0f;
        }
      }
    }
    auto load_all = [&](int rg, bool with_g) {
      if constexpr (FW) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) f_load(i, rg, with_g);
      } else {
        stage_load(n, c0, rg, with_g);
      }
    };
    auto store_all = [&](int xslot, int gbuf) {
      if constexpr (FW) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) f_store(i, xslot, gbuf);
      } else {
        stage_store(xslot, gbuf);
      }
    };
    // Four x-row slots (row & 3) and two gy buffers (row & 1): step r reads x rows r - 1 .. r + 1 and gy row r while the
    // rows of step r + 1 (x row r + 2, gy row r + 1: loaded during step r - 1) are split and stored and the rows of step
    // r + 2 are requested -- ONE barrier per step, and no phase in which every wave of the block does staging work while
    // the matrix pipe idles (round 2 stored between two barriers).
    __syncthreads();                               // the previous item's MFMAs have read their operands
    load_all(r0 - 2, false);
    store_all((r0 - 1) & 3, (r0 + 1) & 1);         // x row r0 - 1 (the gy half of these two stores is zeros into the idle buffer)
    load_all(r0 - 1, false);
    store_all(r0 & 3, (r0 + 1) & 1);               // x row r0
    load_all(r0, true);
    store_all((r0 + 1) & 3, r0 & 1);               // gy row r0, x row r0 + 1
    load_all(r0 + 1, true);                        // gy row r0 + 1, x row r0 + 2 in registers
    for (int r = r0; r < r1; ++r) {
      __syncthreads();                             // the rows of step r are complete; step r - 1 has been read by every wave
      if constexpr (!FW) {
        stage_store((r + 2) & 3, (r + 1) & 1);     // for step r + 1 (x row r - 2 / gy row r - 1 are no longer needed)
        stage_load(n, c0, r + 2, true);            // for step r + 2: in flight under this step's MFMAs
      }
      const int slot = (r - 1 + ky) & 3;           // x row r - 1 + ky
      const unsigned char* xr = xa + slot * (kW3CB * kW3XPitch);
      const unsigned char* gr = ga + (r & 1) * (2 * kW3CB * kW3GPitch);
#pragma unroll
      for (int s = 0; s < 4; ++s) {                // four K steps of 16 pixels
        f16x8 A[NTERM], B[3][NTERM];
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {
          A[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(gr + t * (kW3CB * kW3GPitch) + s * 32));
          const unsigned char* xc = xr + t * (kW3CB * 4 * kW3XPitch) + s * 32;
          const u32x4 cur = *reinterpret_cast<const u32x4*>(xc);
          const unsigned prev3 = *reinterpret_cast<const unsigned*>(xc - 4);
          const unsigned next0 = *reinterpret_cast<const unsigned*>(xc + 16);
          const unsigned t0 = __builtin_amdgcn_alignbit(cur.x, prev3, 16), t1 = __builtin_amdgcn_alignbit(cur.y, cur.x, 16),
                         t2 = __builtin_amdgcn_alignbit(cur.z, cur.y, 16), t3 = __builtin_amdgcn_alignbit(cur.w, cur.z, 16),
                         t4 = __builtin_amdgcn_alignbit(next0, cur.w, 16);
          B[0][t] = __builtin_bit_cast(f16x8, (u32x4){t0, t1, t2, t3});   // kx = 0: pixels shifted by -1
          B[1][t] = __builtin_bit_cast(f16x8, cur);                        // kx = 1
          B[2][t] = __builtin_bit_cast(f16x8, (u32x4){t1, t2, t3, t4});   // kx = 2: shifted by +1
        }
#define DVD_W3TERM(SA, SB)                                                                                  \
  _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[kx] =                                                \
      __builtin_amdgcn_mfma_f32_32x32x16_f16(A[SA], B[kx][SB], acc[kx], 0, 0, 0);
        if constexpr (!H16) {
          DVD_W3TERM(NTERM - 1, 0)
          DVD_W3TERM(0, NTERM - 1)
        }
        DVD_W3TERM(0, 0)
#undef DVD_W3TERM
        if constexpr (FW) {
          // staging item s rides behind the MFMAs of K step s: store the run loaded a step ago (for step r + 1), then
          // request the same run two rows further down (for step r + 2)
          if (s < NQ) {
            f_store(s, (r + 2) & 3, (r + 1) & 1);
            f_load(s, r + 2, true);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // partial[s][tap][G * Cout][Cin]
  float* dst = a.partial + (size_t)blockIdx.x * 9 * a.G * a.Cout * a.Cin;
  const float unscale = H16 ? (a.out_scale ? a.out_scale[0] : 1.0f) : 1.0f / (sx * sg);          // fp32: exact power of two
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int tap = ky * 3 + kx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + pm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + pn * 32 + (lane & 31);
      if (co < a.Cout && ci < a.Cin) dst[((size_t)tap * a.G * a.Cout + grp * a.Cout + co) * a.Cin + ci] = acc[kx][r] * unscale;
    }
  }
}

// ---- 5x5 / 7x7 / 11x11 (round 4): the same walk with KS kernel rows.  Replaces the weight gradient of the hourglass's large
// inception branches (third_party/hourglass.py:21-57) -- MIOpen in round 3, the exact-fp32 MFMA kernel of csrc/xwgrad.hip at the
// start of round 4 (37 % of an hourglass step: fp32 MFMAs run at the vector rate).  A block owns 32 output x 32 input channels;
// wave ky keeps the KXN accumulators of kernel row ky for the columns [KX0, KX0 + KXN) (an 11-wide row takes two launches so
// that the accumulators fit the register file); KS + 1 x-row slots roll through LDS (rows r - pad .. r + pad are read while
// row r + pad + 1 is stored).  The fragment of column offset s is eight pixels starting s pixels off the lane's cell: the
// lane reads the cell and both neighbours as 16-byte cells (conflict free) and takes the run out of the 12 dwords -- a plain
// register selection for even s, v_alignbit for odd s (one shared chain of alignbits serves all odd offsets).
template <int KS, int KX0, int KXN>
__global__ __launch_bounds__(64 * KS) void xwgradk_kernel(const Wg3Args a) {
  constexpr int PAD = KS / 2, NS = KS + 1, CB = 32, NT = 64 * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* sG = smem3;                                   // [buffer 2][term 2][co 32][kW3GPitch]
  unsigned char* sX = smem3 + 2 * 2 * CB * kW3GPitch;          // [term 2][slot NS][ci 32][kW3XPitch]
  constexpr int kXTerm = NS * CB * kW3XPitch;
  const float sx = pow2_scale(a.x_amax[0]), sg = pow2_scale(a.g_amax[0]);
  const int tid = threadIdx.x, lane = tid & 63, ky = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.z * CB, ci0 = blockIdx.y * CB;
  const size_t plane = (size_t)a.H * a.W;
  const int items = a.N * a.nstrips * a.nrseg;
  constexpr int GQ = CB * 16, XQ = CB * 20, NQ = (GQ + XQ + NT - 1) / NT;
  float4 stg[NQ];
  const bool wvec = (a.W & 3) == 0;
  auto slot_of = [](int row) { return (row + 64) % NS; };      // rows >= -PAD
  auto stage_load = [&](int n, int c0, int grow, int xrow) {   // gy row `grow` (< 0: none) and x row `xrow`
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = i * NT + tid;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < GQ) {
        const int ch = q >> 4, px = c0 + ((q & 15) << 2);
        if (grow >= 0 && (co0 + ch) < a.Cout && grow < a.H)
          v = load_quad<false>(static_cast<const float*>(a.gy) + ((size_t)n * a.Cout + co0 + ch) * plane + (size_t)grow * a.W, px, a.W,
                               wvec && px + 3 < a.W, false);
      } else if (q < GQ + XQ) {
        const int qq = q - GQ;
        const int ch = qq / 20, px = c0 - 8 + ((qq - ch * 20) << 2);
        if ((ci0 + ch) < a.Cin && xrow >= 0 && xrow < a.H)
          v = load_quad<false>(static_cast<const float*>(a.x) + ((size_t)n * a.Cin + ci0 + ch) * plane + (size_t)xrow * a.W, px, a.W,
                               wvec && px >= 0 && px + 3 < a.W, a.relu_in != 0);
      }
      stg[i] = v;
    }
  };
  auto stage_store = [&](int xslot, int gbuf) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = i * NT + tid;
      if (q >= GQ + XQ) continue;
      if (q < GQ) {
        store_quad<false>(sG + gbuf * (2 * CB * kW3GPitch) + (q >> 4) * kW3GPitch + ((q & 15) << 3), CB * kW3GPitch, stg[i], sg);
      } else {
        const int qq = q - GQ, ch = qq / 20;
        store_quad<false>(sX + (xslot * CB + ch) * kW3XPitch + ((qq - ch * 20) << 3), kXTerm, stg[i], sx);
      }
    }
  };
  f32x16 acc[KXN];
#pragma unroll
  for (int k = 0; k < KXN; ++k) acc[k] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int half = lane >> 5;
  const unsigned char* ga = sG + (lane & 31) * kW3GPitch + half * 16;
  const unsigned char* xa = sX + (lane & 31) * kW3XPitch + 16 + half * 16;     // cell 1 + half of slot 0

  for (int item = blockIdx.x; item < items; item += a.S) {
    const int n = item / (a.nstrips * a.nrseg);
    const int rem = item - n * (a.nstrips * a.nrseg);
    const int strip = rem / a.nrseg, seg = rem - strip * a.nrseg;
    const int c0 = strip * kW3Strip, r0 = seg * a.RS;
    const int r1 = (r0 + a.RS) < a.H ? (r0 + a.RS) : a.H;
    __syncthreads();                               // the previous item's MFMAs have read their operands
    for (int j = -PAD; j < PAD; ++j) {             // x rows r0 - pad .. r0 + pad - 1 (the gy halves: zeros into the idle buffer)
      stage_load(n, c0, -1, r0 + j);
      stage_store(slot_of(r0 + j), (r0 + 1) & 1);
    }
    stage_load(n, c0, r0, r0 + PAD);
    stage_store(slot_of(r0 + PAD), r0 & 1);        // gy row r0, x row r0 + pad
    stage_load(n, c0, r0 + 1, r0 + 1 + PAD);       // in registers: the rows step r0 stores for step r0 + 1
    for (int r = r0; r < r1; ++r) {
      __syncthreads();                             // the rows of step r are complete; step r - 1 has been read by every wave
      stage_store(slot_of(r + 1 + PAD), (r + 1) & 1);
      stage_load(n, c0, r + 2, r + 2 + PAD);
      const unsigned char* xr = xa + slot_of(r - PAD + ky) * (CB * kW3XPitch);
      const unsigned char* gr = ga + (r & 1) * (2 * CB * kW3GPitch);
#pragma unroll
      for (int s = 0; s < 4; ++s) {                // four K steps of 16 pixels
        f16x8 A[2];
        unsigned d[2][12];                         // per term: previous cell, the lane's cell, next cell (24 pixels)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          A[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(gr + t * (CB * kW3GPitch) + s * 32));
          const unsigned char* xc = xr + t * kXTerm + s * 32;
          const u32x4 p = *reinterpret_cast<const u32x4*>(xc - 16), c = *reinterpret_cast<const u32x4*>(xc),
                      nn = *reinterpret_cast<const u32x4*>(xc + 16);
          d[t][0] = p.x; d[t][1] = p.y; d[t][2] = p.z; d[t][3] = p.w;
          d[t][4] = c.x; d[t][5] = c.y; d[t][6] = c.z; d[t][7] = c.w;
          d[t][8] = nn.x; d[t][9] = nn.y; d[t][10] = nn.z; d[t][11] = nn.w;
        }
#pragma unroll
        for (int kx = 0; kx < KXN; ++kx) {
          const int sh = KX0 + kx - PAD;           // column offset in pixels (compile-time after unrolling)
          const int st = 8 + sh;                   // first pixel of the run inside the 24-pixel window (3 .. 13)
          f16x8 B[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            u32x4 f;
            if ((st & 1) == 0) {
              const int j = st >> 1;
              f = (u32x4){d[t][j], d[t][j + 1], d[t][j + 2], d[t][j + 3]};
            } else {
              const int j = (st - 1) >> 1;
              f = (u32x4){__builtin_amdgcn_alignbit(d[t][j + 1], d[t][j], 16), __builtin_amdgcn_alignbit(d[t][j + 2], d[t][j + 1], 16),
                          __builtin_amdgcn_alignbit(d[t][j + 3], d[t][j + 2], 16), __builtin_amdgcn_alignbit(d[t][j + 4], d[t][j + 3], 16)};
            }
            B[t] = __builtin_bit_cast(f16x8, f);
          }
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], acc[kx], 0, 0, 0);     // small terms first
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], acc[kx], 0, 0, 0);
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], acc[kx], 0, 0, 0);
        }
      }
    }
  }
  float* dst = a.partial + (size_t)blockIdx.x * (KS * KS) * a.Cout * a.Cin;      // partial[s][tap][Cout][Cin]
  const float unscale = 1.0f / (sx * sg);
#pragma unroll
  for (int kx = 0; kx < KXN; ++kx) {
    const int tap = ky * KS + KX0 + kx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + (lane & 31);
      if (co < a.Cout && ci < a.Cin) dst[((size_t)tap * a.Cout + co) * a.Cin + ci] = acc[kx][r] * unscale;
    }
  }
}

// ---- grouped 3x3 with at most 32 channels per group on either side (ResNeXt stages 2 and 3: 16 / 32 per group) -----------
// On xwgrad3_kernel's 64 x 64 channel blocks only one of the four 32 x 32 tile pairs of a block holds data (16 per group: a
// quarter of that pair): three of twelve waves did useful MFMAs while all twelve staged zeros -- 0.42 ms per stage-3 layer
// at 48 x 1024 x 24 x 42 against 0.1 ms of HBM traffic, 4 % of a step for 1 % of its FLOPs.  Here a block is one 32 x 32
// pair and three waves (one per kernel row); two blocks share a CU.  Same walk, same row step as xwgrad3_kernel<., FW>
// (buffer loads, the staging of item i between the MFMAs), same per-element products in the same order: bit-identical sums.
// Rows need not be whole staging items: a run that crosses the end of its row is loaded anyway (dword-aligned 16-byte buffer
// loads) and the elements past the row are zeroed on the way to LDS -- for fp32 by the per-element operand scale the split
// multiplies by anyway (0 instead of 2^e), for fp16 by a mask -- so the 42- and 21-pixel rows of stages 3 and 4 take the
// same path as the rest.
constexpr int kWgCB = 32, kWgNT = 192;
constexpr int kWgLdsBytes = 2 * 2 * kWgCB * kW3GPitch + 2 * kWgCB * 4 * kW3XPitch;    // 63 488

template <bool H16>
__global__ __launch_bounds__(kWgNT) void xwgrad3g_kernel(const Wg3Args a) {
  constexpr int EB = H16 ? 2 : 4, NTERM = H16 ? 1 : 2, CB = kWgCB;
  constexpr int QW = H16 ? 8 : 4, GPR = 64 / QW, XPR = 80 / QW, GQ = CB * GPR, XQ = CB * XPR, NQ = (GQ + XQ) / kWgNT;
  static_assert((GQ + XQ) % kWgNT == 0 && GQ % 64 == 0, "staging items divide evenly over the threads, wave-uniform kind");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* sG = smem3;                                   // [buffer 2][term 2][co 32][kW3GPitch]
  unsigned char* sX = smem3 + 2 * 2 * CB * kW3GPitch;          // [term 2][slot 4][ci 32][kW3XPitch]
  const float sx = H16 ? 1.0f : pow2_scale(a.x_amax[0]), sg = H16 ? 1.0f : pow2_scale(a.g_amax[0]);
  const int tid = threadIdx.x, lane = tid & 63, ky = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.z / a.nco;
  const int co0 = (blockIdx.z - grp * a.nco) * CB, ci0 = blockIdx.y * CB;
  const size_t plane = (size_t)a.H * a.W;
  const int items = a.N * a.nstrips * a.nrseg;

  bool isg[NQ];
  int f_ch[NQ], f_pq[NQ], f_lds[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = i * kWgNT + tid;
    isg[i] = __builtin_amdgcn_readfirstlane(q < GQ ? 1 : 0) != 0;
    const int qq = isg[i] ? q : q - GQ, pr = isg[i] ? GPR : XPR;
    f_ch[i] = qq / pr;
    const int cell = qq - f_ch[i] * pr;
    f_pq[i] = cell * QW - (isg[i] ? 0 : 8);
    f_lds[i] = (isg[i] ? f_ch[i] * kW3GPitch : (int)(sX - smem3) + f_ch[i] * kW3XPitch) + cell * (QW * 2);
  }
  u32x4 fstg[NQ];
  int f_voff[NQ];
  float f_sc[H16 ? 1 : NQ][4];          // fp32: the operand scale per element of the run, 0 past the row's end
  unsigned f_msk[H16 ? NQ : 1][4];      // fp16: all-ones / zero halves
  __amdgpu_buffer_rsrc_t srdX, srdG;
  const float relu_lo = a.relu_in ? 0.0f : -__builtin_inff();
  const unsigned relu_lo_h = a.relu_in ? 0u : 0xfc00fc00u;
  auto f_load = [&](int i, int rg, bool with_g) {
    const int row = isg[i] ? rg : rg + 1;
    const bool ok = row >= 0 && row < a.H && (with_g || !isg[i]);
    const int vo = ok ? f_voff[i] : (int)0x80000000;
    const int so = ok ? row * a.W * EB : 0;
    fstg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(isg[i] ? srdG : srdX, vo, so, 0));
  };
  auto f_store = [&](int i, int xslot, int gbuf) {
    unsigned char* dst = smem3 + f_lds[i] + (isg[i] ? gbuf * (2 * CB * kW3GPitch) : xslot * (CB * kW3XPitch));
    if constexpr (H16) {
      u32x4 v = fstg[i];
      const f16x2 lo = __builtin_bit_cast(f16x2, isg[i] ? 0xfc00fc00u : relu_lo_h);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, (unsigned)v[j]), lo)) & f_msk[i][j];
      *reinterpret_cast<u32x4*>(dst) = v;
    } else {
      const float lo = isg[i] ? -__builtin_inff() : relu_lo;
      const float v0 = fmaxf(__uint_as_float(fstg[i][0]), lo) * f_sc[i][0], v1 = fmaxf(__uint_as_float(fstg[i][1]), lo) * f_sc[i][1];
      const float v2 = fmaxf(__uint_as_float(fstg[i][2]), lo) * f_sc[i][2], v3 = fmaxf(__uint_as_float(fstg[i][3]), lo) * f_sc[i][3];
      unsigned h0, l0, h1, l1;
      split_pair_f16(v0, v1, h0, l0);
      split_pair_f16(v2, v3, h1, l1);
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + (isg[i] ? CB * kW3GPitch : CB * 4 * kW3XPitch)) = make_uint2(l0, l1);
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int half = lane >> 5;
  const unsigned char* ga = sG + (lane & 31) * kW3GPitch + half * 16;
  const unsigned char* xa = sX + (lane & 31) * kW3XPitch + 16 + half * 16;     // cell 1 + half of slot 0

  for (int item = blockIdx.x; item < items; item += a.S) {
    const int n = item / (a.nstrips * a.nrseg);
    const int rem = item - n * (a.nstrips * a.nrseg);
    const int strip = rem / a.nrseg, seg = rem - strip * a.nrseg;
    const int c0 = strip * kW3Strip, r0 = seg * a.RS;
    const int r1 = (r0 + a.RS) < a.H ? (r0 + a.RS) : a.H;
    {
      const int nci = (a.Cin - ci0) < CB ? (a.Cin - ci0) : CB, nco = (a.Cout - co0) < CB ? (a.Cout - co0) : CB;
      const unsigned char* xb = static_cast<const unsigned char*>(a.x) + (((size_t)n * a.G + grp) * a.Cin + ci0) * plane * EB;
      const unsigned char* gb = static_cast<const unsigned char*>(a.gy) + (((size_t)n * a.G + grp) * a.Cout + co0) * plane * EB;
      srdX = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(xb)), 0, nci * (int)plane * EB, 0x00020000);
      srdG = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(gb)), 0, nco * (int)plane * EB, 0x00020000);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int px = c0 + f_pq[i];
        const int nv = px < 0 ? 0 : ((a.W - px) < QW ? (a.W - px) : QW);       // elements of the run inside the row (<= 0: none)
        f_voff[i] = nv > 0 ? (f_ch[i] * (int)plane + px) * EB : (int)0x80000000;
        if constexpr (H16) {
#pragma unroll
          for (int j = 0; j < 4; ++j) f_msk[i][j] = (2 * j < nv ? 0xffffu : 0u) | (2 * j + 1 < nv ? 0xffff0000u : 0u);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) f_sc[i][e] = e < nv ? (isg[i] ? sg : sx) : 0.0f;
        }
      }
    }
    auto load_all = [&](int rg, bool with_g) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) f_load(i, rg, with_g);
    };
    auto store_all = [&](int xslot, int gbuf) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) f_store(i, xslot, gbuf);
    };
    __syncthreads();                               // the previous item's MFMAs have read their operands
    load_all(r0 - 2, false);
    store_all((r0 - 1) & 3, (r0 + 1) & 1);         // x row r0 - 1 (the gy half: zeros into the idle buffer)
    load_all(r0 - 1, false);
    store_all(r0 & 3, (r0 + 1) & 1);               // x row r0
    load_all(r0, true);
    store_all((r0 + 1) & 3, r0 & 1);               // gy row r0, x row r0 + 1
    load_all(r0 + 1, true);                        // gy row r0 + 1, x row r0 + 2 in registers
    for (int r = r0; r < r1; ++r) {
      __syncthreads();                             // the rows of step r are complete; step r - 1 has been read by every wave
      const int slot = (r - 1 + ky) & 3;           // x row r - 1 + ky
      const unsigned char* xr = xa + slot * (CB * kW3XPitch);
      const unsigned char* gr = ga + (r & 1) * (2 * CB * kW3GPitch);
#pragma unroll
      for (int s = 0; s < 4; ++s) {                // four K steps of 16 pixels
        f16x8 A[NTERM], B[3][NTERM];
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {
          A[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(gr + t * (CB * kW3GPitch) + s * 32));
          const unsigned char* xc = xr + t * (CB * 4 * kW3XPitch) + s * 32;
          const u32x4 cur = *reinterpret_cast<const u32x4*>(xc);
          const unsigned prev3 = *reinterpret_cast<const unsigned*>(xc - 4);
          const unsigned next0 = *reinterpret_cast<const unsigned*>(xc + 16);
          const unsigned t0 = __builtin_amdgcn_alignbit(cur.x, prev3, 16), t1 = __builtin_amdgcn_alignbit(cur.y, cur.x, 16),
                         t2 = __builtin_amdgcn_alignbit(cur.z, cur.y, 16), t3 = __builtin_amdgcn_alignbit(cur.w, cur.z, 16),
                         t4 = __builtin_amdgcn_alignbit(next0, cur.w, 16);
          B[0][t] = __builtin_bit_cast(f16x8, (u32x4){t0, t1, t2, t3});
          B[1][t] = __builtin_bit_cast(f16x8, cur);
          B[2][t] = __builtin_bit_cast(f16x8, (u32x4){t1, t2, t3, t4});
        }
#define DVD_W3TERM(SA, SB)                                                                                  \
  _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[kx] =                                                \
      __builtin_amdgcn_mfma_f32_32x32x16_f16(A[SA], B[kx][SB], acc[kx], 0, 0, 0);
        if constexpr (!H16) {
          DVD_W3TERM(NTERM - 1, 0)
          DVD_W3TERM(0, NTERM - 1)
        }
        DVD_W3TERM(0, 0)
#undef DVD_W3TERM
        // the staging items i = s, s + 4 ride behind the MFMAs of K step s (for step r + 1 / r + 2, as in xwgrad3_kernel)
#pragma unroll
        for (int i = s; i < NQ; i += 4) {
          f_store(i, (r + 2) & 3, (r + 1) & 1);
          f_load(i, r + 2, true);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float* dst = a.partial + (size_t)blockIdx.x * 9 * a.G * a.Cout * a.Cin;        // partial[s][tap][G * Cout][Cin]
  const float unscale = H16 ? (a.out_scale ? a.out_scale[0] : 1.0f) : 1.0f / (sx * sg);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int tap = ky * 3 + kx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + (lane & 31);
      if (co < a.Cout && ci < a.Cin) dst[((size_t)tap * a.G * a.Cout + grp * a.Cout + co) * a.Cin + ci] = acc[kx][r] * unscale;
    }
  }
}

// gw[co][ci][tap] = sum_s partial[s][tap][co][ci], ascending s (two interleaved chains)
__global__ __launch_bounds__(256) void xwgrad3_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int S,
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

// ---- 1x1: dW[co][ci] = sum_{n, p} gy[n][co][p] * act(x)[n][ci][p] -- a plain "NT" GEMM with K = pixels contiguous in
// both operands.  A block owns 128 x 128 channels (8 waves: 4 x 2, one output-channel tile x two input-channel tiles
// each) and walks over chunks of 32 consecutive pixels of the flattened images; two blocks share a CU so that one
// block's staging (load, split, LDS write, barrier) overlaps the other's MFMAs.
constexpr int kW1Chunk = 32;
constexpr int kW1Pitch = 64 + 16;             // bytes per channel row in LDS (32 fp16 + pad)
constexpr int kW1CB = 128;
constexpr int kW1NT = 512;

template <bool H16>
__global__ __launch_bounds__(kW1NT, 2) void xwgrad1s_kernel(const Wg3Args a) {
  constexpr int EB = H16 ? 2 : 4;
  constexpr int NTERM = H16 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* sG = smem3;                               // [term 2][co 128][kW1Pitch]
  unsigned char* sX = smem3 + 2 * kW1CB * kW1Pitch;        // [term 2][ci 128][kW1Pitch]
  const float sx = H16 ? 1.0f : pow2_scale(a.x_amax[0]), sg = H16 ? 1.0f : pow2_scale(a.g_amax[0]);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pn = wave & 1, pm = wave >> 1;
  const int co0 = blockIdx.z * kW1CB, ci0 = blockIdx.y * kW1CB;
  const int HW = a.H * a.W;
  const size_t plane = (size_t)HW;
  const int cpi = (HW + kW1Chunk - 1) / kW1Chunk;          // chunks per image
  const int items = a.N * cpi;
  // staging: 128 channels x 8 quads per operand = 2048 quads -> 4 per thread (2 gy + 2 x)
  typename Quad<H16>::type stg[4];
  const bool hvec = (HW & 3) == 0;
  auto stage_load = [&](int item) {
    const int n = item / cpi, p0 = (item - n * cpi) * kW1Chunk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = (i & 1) * kW1NT + tid;                 // 0 .. 1023
      const int ch = q >> 3, px = p0 + ((q & 7) << 2);
      const bool isx = i >= 2;
      const int C = isx ? a.Cin : a.Cout, c = (isx ? ci0 : co0) + ch;
      typename Quad<H16>::type v = zero_quad<H16>();
      if (c < C) {
        const unsigned char* p = static_cast<const unsigned char*>(isx ? a.x : a.gy) + ((size_t)n * C + c) * plane * EB;
        v = load_quad<H16>(p, px, HW, hvec && px + 3 < HW, isx && a.relu_in);
      }
      stg[i] = v;
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = (i & 1) * kW1NT + tid;
      const float sc = i >= 2 ? sx : sg;
      unsigned char* dst = (i >= 2 ? sX : sG) + (q >> 3) * kW1Pitch + ((q & 7) << 3);
      constexpr int tstride = kW1CB * kW1Pitch;
      store_quad<H16>(dst, tstride, stg[i], sc);
    }
  };
  f32x16 acc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) acc[k] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int half = lane >> 5;
  const unsigned char* ga = sG + (pm * 32 + (lane & 31)) * kW1Pitch + half * 16;
  const unsigned char* xa = sX + (pn * 64 + (lane & 31)) * kW1Pitch + half * 16;

  int item = blockIdx.x;
  if (item < items) stage_load(item);
  for (; item < items; item += a.S) {
    __syncthreads();
    stage_store();
    __syncthreads();
    if (item + a.S < items) stage_load(item + a.S);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 A[NTERM], B[2][NTERM];
#pragma unroll
      for (int t = 0; t < NTERM; ++t) {
        A[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ga + t * (kW1CB * kW1Pitch) + s * 32));
        B[0][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(xa + t * (kW1CB * kW1Pitch) + s * 32));
        B[1][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(xa + 32 * kW1Pitch + t * (kW1CB * kW1Pitch) + s * 32));
      }
#define DVD_W1TERM(SA, SB)                                                                                \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] =                                                  \
      __builtin_amdgcn_mfma_f32_32x32x16_f16(A[SA], B[j][SB], acc[j], 0, 0, 0);
      if constexpr (!H16) {
        DVD_W1TERM(NTERM - 1, 0)
        DVD_W1TERM(0, NTERM - 1)
      }
      DVD_W1TERM(0, 0)
#undef DVD_W1TERM
    }
  }
  float* dst = a.partial + (size_t)blockIdx.x * a.Cout * a.Cin;       // partial[s][co][ci]
  const float unscale = H16 ? (a.out_scale ? a.out_scale[0] : 1.0f) : 1.0f / (sx * sg);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + pm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + pn * 64 + j * 32 + (lane & 31);
      if (co < a.Cout && ci < a.Cin) dst[(size_t)co * a.Cin + ci] = acc[j][r] * unscale;
    }
}

// ---- 1x1, wide layers: one 512-thread workgroup owns 256 x 256 channels (the whole weight matrix of a ResNeXt stage-1 /
// decoder 1x1; a 256 x 256 block of the wider ones) for a run of 16-pixel chunks.  With 128 x 128 blocks each operand row
// is read by two workgroups and the kernel sits on the HBM roofline of that blocking (32 FLOP/B -> ~160 TF/s, measured
// 156-179); 256 x 256 halves the traffic.  The structure is the scene-flow MLP's weight-gradient kernel (csrc/sf_mlp.hip
// dw_body): 512 channel rows x 16 pixels are loaded as fp32, split and written to a double-buffered LDS stage while the
// MFMAs of the previous chunk run, one barrier per chunk, wave (wr, wc) keeps rows [64 wr, +64) x columns [128 wc, +128) in
// 128 accumulators.  Each slice writes its partial matrix; xwgrad3_reduce_kernel sums them in slice order (deterministic).
constexpr int kWbPitch = 48;                       // bytes per channel row of a 16-pixel chunk in LDS (32 + 16 pad)
constexpr int kWbTerm = 256 * kWbPitch;            // one split term of one operand
constexpr int kWbBuf = 2 * 2 * kWbTerm;            // gy terms, then x terms
constexpr size_t kWbLds = 2 * (size_t)kWbBuf;      // double buffered: 98 304

// FW (host: H * W a multiple of 16, so every 16-pixel chunk is whole): the staging loads are raw buffer loads (channel rows past
// the tensor fall out of the resource's range and read 0), a column tile's MFMAs are followed by the split + LDS store of ONE
// staging item and the request of the same item two chunks ahead -- round 3 ran `barrier | requests | all MFMAs | all splits`
// with both waves of a SIMD in the same phase (matrix pipe 37 % busy) and one chunk of latency cover for the HBM request.
template <bool H16, bool FW>
__global__ __launch_bounds__(512) void xwgrad1b_kernel(const Wg3Args a) {
  constexpr int EB = H16 ? 2 : 4;
  constexpr int NTERM = H16 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int i32 = lane & 31, hh = lane >> 5;
  const int co0 = blockIdx.z * 256, ci0 = blockIdx.y * 256, s = blockIdx.x;
  const int HW = a.H * a.W;
  const size_t plane = (size_t)HW;
  const int cpi = (HW + 15) / 16;                              // 16-pixel chunks per image
  const long long items = (long long)a.N * cpi;
  const long long t0 = items * s / a.S, t1 = items * (s + 1) / a.S;
  const int n_it = (int)(t1 - t0);
  const float sx = H16 ? 1.0f : pow2_scale(a.x_amax[0]), sg = H16 ? 1.0f : pow2_scale(a.g_amax[0]);

  f32x16 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // staging: 512 rows x 4 quads of 4 pixels -> 4 float4 per thread; q = i * 512 + tid, row = q >> 2 (0..255 gy, 256..511 x)
  // (16-byte staging items for fp16 rows, as in xwgrad3_kernel, were tried here too: 3-10 % slower on the 1x1 shapes; not kept)
  typedef typename Quad<H16>::type QT;
  QT sg0[4], sg1[4];
  auto stage_load = [&](int it, QT (&st)[4]) {
    const long long item = t0 + it;
    const int n = (int)(item / cpi), p0 = (int)(item - (long long)n * cpi) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 128 + (tid >> 2), quad = tid & 3;
      const bool isx = i >= 2;
      const int r = row & 255;
      const int C = isx ? a.Cin : a.Cout, c = (isx ? ci0 : co0) + r;
      const int px = p0 + quad * 4;
      QT v = zero_quad<H16>();
      if (c < C && px < HW)                                      // HW % 4 == 0 (host): a quad is inside or outside as a whole
        v = load_quad<H16>(static_cast<const unsigned char*>(isx ? a.x : a.gy) + ((size_t)n * C + c) * plane * EB, px, HW, true,
                           isx && a.relu_in);
      st[i] = v;
    }
  };
  auto stage_store = [&](int buf, const QT (&st)[4]) {
    unsigned char* base = smem3 + buf * kWbBuf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 128 + (tid >> 2), quad = tid & 3;
      const float sc = i >= 2 ? sx : sg;
      unsigned char* dst = base + (i >= 2 ? 2 * kWbTerm : 0) + (row & 255) * kWbPitch + quad * 8;
      store_quad<H16>(dst, kWbTerm, st[i], sc);
    }
  };
  // Fragments of chunk it + 1 are read from LDS DURING the MFMAs of chunk it (the rolling prefetch of csrc/xconv.hip: one
  // 512-thread block per CU runs in lock step, so fragment reads placed right after the barrier leave the matrix pipe idle
  // while all eight waves queue on LDS): the gy fragments into a second register set, the x fragments of column tile c into
  // the registers that tile's MFMAs have just released.  Chunk it is therefore loaded during step it - 3, split and stored
  // during step it - 2 (into the buffer whose fragments were read during step it - 3), read during step it - 1.
  const unsigned char* gb0 = smem3 + (64 * wr + i32) * kWbPitch + hh * 16;
  const unsigned char* hb0 = smem3 + 2 * kWbTerm + (128 * wc + i32) * kWbPitch + hh * 16;
  u32x4 Bf[4][NTERM];
  auto read_a = [&](int buf, u32x4 (&A)[2][NTERM]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < NTERM; ++t) A[r][t] = *reinterpret_cast<const u32x4*>(gb0 + buf * kWbBuf + t * kWbTerm + r * 32 * kWbPitch);
  };
  auto read_b = [&](int buf, int c) {
#pragma unroll
    for (int t = 0; t < NTERM; ++t) Bf[c][t] = *reinterpret_cast<const u32x4*>(hb0 + buf * kWbBuf + t * kWbTerm + c * 32 * kWbPitch);
  };
  auto mfma_roll = [&](int nbuf, const u32x4 (&A)[2][NTERM], u32x4 (&An)[2][NTERM]) {    // MFMAs on (A, Bf); next fragments from nbuf
    read_a(nbuf, An);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#define DVD_WB_TERM(SA, SB)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(           \
      __builtin_bit_cast(f16x8, A[r][SA]), __builtin_bit_cast(f16x8, Bf[c][SB]), acc[r][c], 0, 0, 0);
      if constexpr (!H16) {
        DVD_WB_TERM(NTERM - 1, 0)
        DVD_WB_TERM(0, NTERM - 1)
      }
      DVD_WB_TERM(0, 0)
#undef DVD_WB_TERM
      __builtin_amdgcn_sched_barrier(0);           // the reload stays behind this column tile's MFMAs
      read_b(nbuf, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (FW) {
    if (n_it > 0) {
      const int nco = (a.Cout - co0) < 256 ? (a.Cout - co0) : 256, nci = (a.Cin - ci0) < 256 ? (a.Cin - ci0) : 256;
      // item i of a thread: channel row i * 128 + (tid >> 2) (0..255 gy, 256..511 x), quad tid & 3 of the chunk's 16 pixels
      int voff[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) voff[h] = ((h * 128 + (tid >> 2)) * (int)plane + (tid & 3) * 4) * EB;
      u32x4 f0[4], f1[4];                                              // fp16 rows: 8 bytes per item, in the low half
      const float relu_lo = a.relu_in ? 0.0f : -__builtin_inff();
      const unsigned relu_lo_h = a.relu_in ? 0u : 0xfc00fc00u;
      auto f_load = [&](int it, int i, u32x4 (&st)[4]) {
        // chunks past the slice's end read zeros (an offset beyond the range): the loop always runs whole pairs of chunks
        const bool live = it < n_it;
        const long long item = t0 + (live ? it : 0);
        const int n = (int)(item / cpi), p0 = (int)(item - (long long)n * cpi) * 16;     // wave-uniform
        const int vo = live ? voff[i & 1] : (int)0x80000000;
        const bool isx = i >= 2;
        const unsigned char* base = static_cast<const unsigned char*>(isx ? a.x : a.gy) +
                                    ((size_t)n * (isx ? a.Cin : a.Cout) + (isx ? ci0 : co0)) * plane * EB;
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(const_cast<unsigned char*>(base)), 0, (isx ? nci : nco) * (int)plane * EB, 0x00020000);
        if constexpr (H16) {
          const u32x2v v = __builtin_bit_cast(u32x2v, __builtin_amdgcn_raw_buffer_load_b64(srd, vo, p0 * EB, 0));
          st[i] = (u32x4){v[0], v[1], 0u, 0u};
        } else {
          st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, vo, p0 * EB, 0));
        }
      };
      auto f_store = [&](int buf, int i, const u32x4 (&st)[4]) {
        const bool isx = i >= 2;
        const int row = i * 128 + (tid >> 2), quad = tid & 3;
        unsigned char* dst = smem3 + buf * kWbBuf + (isx ? 2 * kWbTerm : 0) + (row & 255) * kWbPitch + quad * 8;
        if constexpr (H16) {
          unsigned v0 = st[i][0], v1 = st[i][1];
          if (isx) {
            const f16x2 lo = __builtin_bit_cast(f16x2, relu_lo_h);
            v0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, v0), lo));
            v1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2, v1), lo));
          }
          *reinterpret_cast<uint2*>(dst) = make_uint2(v0, v1);
        } else {
          const float sc = isx ? sx : sg;
          const float lo = isx ? relu_lo : -__builtin_inff();
          const float v0 = fmaxf(__uint_as_float(st[i][0]), lo) * sc, v1 = fmaxf(__uint_as_float(st[i][1]), lo) * sc;
          const float v2 = fmaxf(__uint_as_float(st[i][2]), lo) * sc, v3 = fmaxf(__uint_as_float(st[i][3]), lo) * sc;
          unsigned h0, l0, h1, l1;
          split_pair_f16(v0, v1, h0, l0);
          split_pair_f16(v2, v3, h1, l1);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + kWbTerm) = make_uint2(l0, l1);
        }
      };
      // MFMAs of one chunk on (A, Bf), next fragments from nbuf; behind column tile c: item c of `st` is stored into buffer
      // sbuf (chunk `it_st`, requested two chunks ago) and requested again for chunk it_ld
      auto mfma_fw = [&](int nbuf, const u32x4 (&A)[2][NTERM], u32x4 (&An)[2][NTERM], int sbuf, u32x4 (&st)[4], int it_ld) {
        read_a(nbuf, An);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#define DVD_WB_TERM(SA, SB)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(           \
      __builtin_bit_cast(f16x8, A[r][SA]), __builtin_bit_cast(f16x8, Bf[c][SB]), acc[r][c], 0, 0, 0);
          if constexpr (!H16) {
            DVD_WB_TERM(NTERM - 1, 0)
            DVD_WB_TERM(0, NTERM - 1)
          }
          DVD_WB_TERM(0, 0)
#undef DVD_WB_TERM
          read_b(nbuf, c);
          f_store(sbuf, c, st);
          f_load(it_ld, c, st);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) f_load(0, i, f0);
#pragma unroll
      for (int i = 0; i < 4; ++i) f_load(1, i, f1);
#pragma unroll
      for (int i = 0; i < 4; ++i) f_store(0, i, f0);
#pragma unroll
      for (int i = 0; i < 4; ++i) f_load(2, i, f0);
#pragma unroll
      for (int i = 0; i < 4; ++i) f_store(1, i, f1);
#pragma unroll
      for (int i = 0; i < 4; ++i) f_load(3, i, f1);
      __syncthreads();
      u32x4 A0[2][NTERM], A1[2][NTERM];
      read_a(0, A0);
#pragma unroll
      for (int c = 0; c < 4; ++c) read_b(0, c);
      for (int it = 0; it < n_it; it += 2) {
        __syncthreads();                   // chunk it + 1 is complete in buffer 1; buffer 0 has been read by every wave
        mfma_fw(1, A0, A1, 0, f0, it + 4);                  // chunk it; chunk it + 2 -> buffer 0; request chunk it + 4
        __syncthreads();
        mfma_fw(0, A1, A0, 1, f1, it + 5);                         // chunk it + 1 (zeros past the end); it + 3 -> buffer 1
      }
    }
  } else
  if (n_it > 0) {
    const int last = n_it - 1;
    auto clamp = [&](int i) { return i < last ? i : last; };
    stage_load(0, sg0);
    stage_store(0, sg0);
    stage_load(clamp(1), sg0);
    stage_store(1, sg0);
    stage_load(clamp(2), sg0);
    __syncthreads();
    u32x4 A0[2][NTERM], A1[2][NTERM];
    read_a(0, A0);
#pragma unroll
    for (int c = 0; c < 4; ++c) read_b(0, c);
    for (int it = 0; it < n_it; it += 2) {
      __syncthreads();                     // chunk it + 1 is complete in buffer 1; buffer 0 has been read by every wave
      stage_load(clamp(it + 3), sg1);
      __builtin_amdgcn_sched_barrier(0);   // the loads stay above the MFMAs
      mfma_roll(1, A0, A1);                // chunk it
      stage_store(0, sg0);                 // chunk it + 2 (past the end: a copy of the last chunk into an idle buffer)
      __syncthreads();
      stage_load(clamp(it + 4), sg0);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < n_it) mfma_roll(0, A1, A0);   // chunk it + 1
      stage_store(1, sg1);                 // chunk it + 3
    }
  }
  float* dst = a.partial + (size_t)s * a.Cout * a.Cin;       // partial[s][co][ci]
  const float unscale = H16 ? (a.out_scale ? a.out_scale[0] : 1.0f) : 1.0f / (sx * sg);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 64 * wr + 32 * rr + (r & 3) + 8 * (r >> 2) + 4 * hh, ci = ci0 + 128 * wc + 32 * c + i32;
        if (co < a.Cout && ci < a.Cin) dst[(size_t)co * a.Cin + ci] = acc[rr][c][r] * unscale;
      }
}

// out[c] = sum over images and pixels of g[n][c][:]: one block per channel, fixed order (per-thread strided partial sums, then
// a fixed tree).  (Summing the gy rows inside xwgrad1b_kernel, which stages them anyway, was tried in round 3: the eight extra
// adds per chunk cost the kernel 10 % -- more than this pass)
__global__ __launch_bounds__(256) void chansum_kernel(const float* __restrict__ g, int N, int C, int HW, float* __restrict__ out) {
  const int c = blockIdx.x;
  float v = 0.0f;
  for (int n = 0; n < N; ++n) {
    const float* p = g + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += 256) v += p[i];
  }
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// which 1x1 kernel: the 256 x 256 blocks when both channel counts fill them reasonably and rows are float4 aligned
static bool wg1_wide(int Cin, int Cout, int HW) { return Cin >= 192 && Cout >= 192 && (HW & 3) == 0 && g_w1_variant != 1; }
static int wg1_wide_slices(int N, int Cin, int Cout, int HW) {
  const int pairs = ((Cout + 255) / 256) * ((Cin + 255) / 256);
  int S = pairs >= 256 ? 1 : (256 + pairs - 1) / pairs;       // one workgroup per CU: a whole round over the 256 CUs
  const long long items = (long long)N * ((HW + 15) / 16);
  if (S > items) S = (int)items;
  return S;
}

struct Wg3Plan {
  int nstrips, RS, nrseg, S, nco, nci;
  size_t lds;
};
static void wg3_plan(int N, int Cin, int Cout, int H, int W, int G, Wg3Plan& p) {   // Cin / Cout per group
  p.nco = (Cout + kW3CB - 1) / kW3CB;
  p.nci = (Cin + kW3CB - 1) / kW3CB;
  p.nstrips = (W + kW3Strip - 1) / kW3Strip;
  const int pairs = p.nco * p.nci * G;
  // one block per CU is resident: a whole number of rounds over the 256 CUs, each block a few work items long
  int S = pairs >= 256 ? 1 : (512 + pairs - 1) / pairs;
  // rows per item: enough items to feed S slices evenly (>= 4 per slice), at least 8 rows (2 warm-up rows per item)
  int RS = H;
  while (RS > 8 && (long long)N * p.nstrips * ((H + RS - 1) / RS) < 4LL * S) RS = (RS + 1) / 2;
  p.RS = RS;
  p.nrseg = (H + RS - 1) / RS;
  const long long items = (long long)N * p.nstrips * p.nrseg;
  if (S > items) S = (int)items;
  p.S = S;
  p.lds = (size_t)kW3LdsBytes + 16;
}

// 32 x 32 channel blocks of three waves (xwgrad3g_kernel): grouped layers with at most 32 channels per group on both sides
static bool wg3_small(int Cin, int Cout, int G) { return G > 1 && Cin <= kWgCB && Cout <= kWgCB && g_w3_variant != 1; }
static void wg3g_plan(int N, int Cin, int Cout, int H, int W, int G, Wg3Plan& p) {
  p.nco = (Cout + kWgCB - 1) / kWgCB;
  p.nci = (Cin + kWgCB - 1) / kWgCB;
  p.nstrips = (W + kW3Strip - 1) / kW3Strip;
  const int pairs = p.nco * p.nci * G;
  int S = pairs >= 512 ? 1 : (512 + pairs - 1) / pairs;        // two blocks per CU are resident
  int RS = H;
  while (RS > 8 && (long long)N * p.nstrips * ((H + RS - 1) / RS) < 4LL * S) RS = (RS + 1) / 2;
  p.RS = RS;
  p.nrseg = (H + RS - 1) / RS;
  const long long items = (long long)N * p.nstrips * p.nrseg;
  if (S > items) S = (int)items;
  p.S = S;
  p.lds = (size_t)kWgLdsBytes;
}

}  // namespace dvd

extern "C" {

size_t dvd_xwgrad3_workspace_bytes(int N, int Cin, int Cout, int H, int W, int groups) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || groups <= 0 || Cin % groups || Cout % groups) return 0;
  dvd::Wg3Plan p;
  dvd::wg3_plan(N, Cin / groups, Cout / groups, H, W, groups, p);
  int S = p.S;
  if (dvd::wg3_small(Cin / groups, Cout / groups, groups)) {      // (the larger of the two plans: the launch may take either)
    dvd::wg3g_plan(N, Cin / groups, Cout / groups, H, W, groups, p);
    if (p.S > S) S = p.S;
  }
  return (size_t)S * 9 * Cout * (Cin / groups) * sizeof(float);
}

static int xwgrad3_impl(const void* x, const float* x_amax, const void* gy, const float* gy_amax, float* gw, void* workspace,
                        size_t workspace_bytes, int N, int Cin_total, int Cout_total, int H, int W, int groups, int relu_in,
                        bool h16, const float* out_scale, dvd_stream_t stream) {
  DVD_REQUIRE(x && gy && gw && workspace, "xwgrad3: null pointer");
  DVD_REQUIRE(h16 || (x_amax && gy_amax), "xwgrad3: the operands' max|.| scalars are missing (dvd_amax)");
  DVD_REQUIRE(N > 0 && Cin_total > 0 && Cout_total > 0 && H > 0 && W > 0, "xwgrad3: bad shape");
  DVD_REQUIRE(groups > 0 && Cin_total % groups == 0 && Cout_total % groups == 0, "xwgrad3: %d groups do not divide the channels", groups);
  DVD_REQUIRE((long long)H * W * (long long)(Cin_total > Cout_total ? Cin_total : Cout_total) < (1ll << 31),
              "xwgrad3: image too large for 32-bit offsets");
  const int Cin = Cin_total / groups, Cout = Cout_total / groups;
  dvd::Wg3Plan p;
  // rows of fp16 elements must start dword-aligned for the 32 x 32 kernel's buffer loads: even widths
  const bool small = dvd::wg3_small(Cin, Cout, groups) && (!h16 || W % 2 == 0) && (long long)32 * H * W * (h16 ? 2 : 4) < (1ll << 31);
  if (small) dvd::wg3g_plan(N, Cin, Cout, H, W, groups, p);
  else dvd::wg3_plan(N, Cin, Cout, H, W, groups, p);
  const size_t need = (size_t)p.S * 9 * Cout_total * Cin * sizeof(float);
  if (workspace_bytes < need) {
    dvd::set_error("xwgrad3: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  DVD_REQUIRE((long long)p.nco * groups <= 65535 && p.nci <= 65535, "xwgrad3: too many channel blocks");
  dvd::Wg3Args a;
  a.x = x;
  a.gy = gy;
  a.x_amax = x_amax;
  a.g_amax = gy_amax;
  a.partial = static_cast<float*>(workspace);
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.G = groups; a.nco = p.nco;
  a.nstrips = p.nstrips; a.RS = p.RS; a.nrseg = p.nrseg; a.S = p.S;
  a.relu_in = relu_in ? 1 : 0;
  a.out_scale = out_scale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go = [&](auto kern) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL(kern, dim3(p.S, p.nci, p.nco * groups), dim3(small ? dvd::kWgNT : dvd::kW3NT), p.lds, s, a);
    return DVD_OK;
  };
  // whole 16-byte staging items (and 64 channels of one image within a 31-bit buffer range): the branch-free row step
  const bool fw = (!h16 || W % 2 == 0) && (long long)64 * H * W * (h16 ? 2 : 4) < (1ll << 31) && dvd::g_w3_variant != 1;
  int e;
  dvd::flops_add(small ? DVD_FLOP_XWGRAD3G : DVD_FLOP_XWGRAD3, 2.0 * 9 * N * (double)Cout_total * Cin * (double)H * W);
  if (small) e = h16 ? go(dvd::xwgrad3g_kernel<true>) : go(dvd::xwgrad3g_kernel<false>);
  else if (h16) e = fw ? go(dvd::xwgrad3_kernel<true, true>) : go(dvd::xwgrad3_kernel<true, false>);
  else e = fw ? go(dvd::xwgrad3_kernel<false, true>) : go(dvd::xwgrad3_kernel<false, false>);
  if (e) return e;
  DVD_LAUNCH_OK();
  const long long per = (long long)9 * Cout_total * Cin;
  hipLaunchKernelGGL(dvd::xwgrad3_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                     static_cast<const float*>(workspace), gw, p.S, 9, Cout_total, Cin);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_xwgrad3(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                size_t workspace_bytes, int N, int Cin_total, int Cout_total, int H, int W, int groups, int relu_in,
                dvd_stream_t stream) {
  return xwgrad3_impl(x, x_amax, gy, gy_amax, gw, workspace, workspace_bytes, N, Cin_total, Cout_total, H, W, groups, relu_in, false,
                      nullptr, stream);
}

int dvd_xwgrad3_h(const void* x, const void* gy, const float* out_scale, float* gw, void* workspace, size_t workspace_bytes, int N,
                  int Cin_total, int Cout_total, int H, int W, int groups, int relu_in, dvd_stream_t stream) {
  return xwgrad3_impl(x, nullptr, gy, nullptr, gw, workspace, workspace_bytes, N, Cin_total, Cout_total, H, W, groups, relu_in, true,
                      out_scale, stream);
}

size_t dvd_xwgrad1s_workspace_bytes(int N, int Cin, int Cout, int H, int W) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const int pairs = ((Cout + dvd::kW1CB - 1) / dvd::kW1CB) * ((Cin + dvd::kW1CB - 1) / dvd::kW1CB);
  int S = pairs >= 512 ? 1 : (1024 + pairs - 1) / pairs;
  const int Sw = dvd::wg1_wide_slices(N, Cin, Cout, H * W);      // (either kernel may serve the call)
  if (Sw > S) S = Sw;
  return (size_t)S * Cout * Cin * sizeof(float);
}

// ---- 5x5 / 7x7 / 11x11
static bool wgk_plan(int N, int Cin, int Cout, int H, int W, int KS, dvd::Wg3Plan& p) {
  if (KS != 5 && KS != 7 && KS != 11) return false;
  p.nco = (Cout + 31) / 32;
  p.nci = (Cin + 31) / 32;
  p.nstrips = (W + dvd::kW3Strip - 1) / dvd::kW3Strip;
  const int pairs = p.nco * p.nci;
  int S = pairs >= 256 ? 1 : (256 + pairs - 1) / pairs;        // one block per CU is resident
  int RS = H;
  while (RS > 4 * KS && (long long)N * p.nstrips * ((H + RS - 1) / RS) < 4LL * S) RS = (RS + 1) / 2;   // (KS - 1 warm-up rows per item)
  p.RS = RS;
  p.nrseg = (H + RS - 1) / RS;
  const long long items = (long long)N * p.nstrips * p.nrseg;
  if (S > items) S = (int)items;
  p.S = S;
  p.lds = (size_t)2 * 2 * 32 * dvd::kW3GPitch + (size_t)2 * (KS + 1) * 32 * dvd::kW3XPitch;
  return true;
}

size_t dvd_xwgradk_workspace_bytes(int N, int Cin, int Cout, int H, int W, int KS) {
  dvd::Wg3Plan p;
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !wgk_plan(N, Cin, Cout, H, W, KS, p)) return 0;
  return (size_t)p.S * KS * KS * Cout * Cin * sizeof(float);
}

int dvd_xwgradk(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int KS, int relu_in, dvd_stream_t stream) {
  DVD_REQUIRE(x && gy && gw && workspace && x_amax && gy_amax, "xwgradk: null pointer");
  DVD_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "xwgradk: bad shape");
  DVD_REQUIRE((long long)H * W * (long long)(Cin > Cout ? Cin : Cout) < (1ll << 31), "xwgradk: image too large for 32-bit offsets");
  dvd::Wg3Plan p;
  DVD_REQUIRE(wgk_plan(N, Cin, Cout, H, W, KS, p), "xwgradk: kernel size %d (5, 7 and 11 are covered)", KS);
  const size_t need = (size_t)p.S * KS * KS * Cout * Cin * sizeof(float);
  if (workspace_bytes < need) {
    dvd::set_error("xwgradk: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  DVD_REQUIRE(p.nco <= 65535 && p.nci <= 65535, "xwgradk: too many channel blocks");
  dvd::Wg3Args a;
  a.x = x;
  a.gy = gy;
  a.x_amax = x_amax;
  a.g_amax = gy_amax;
  a.out_scale = nullptr;
  a.partial = static_cast<float*>(workspace);
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.G = 1; a.nco = p.nco;
  a.nstrips = p.nstrips; a.RS = p.RS; a.nrseg = p.nrseg; a.S = p.S;
  a.relu_in = relu_in ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(p.S, p.nci, p.nco);
  auto go = [&](auto kern, int waves) -> int {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL(kern, grid, dim3(64 * waves), p.lds, s, a);
    DVD_LAUNCH_OK();
    return DVD_OK;
  };
  dvd::flops_add(DVD_FLOP_XWGRADK, 2.0 * KS * KS * N * (double)Cout * Cin * (double)H * W);
  if (KS == 5) {
    if (int e = go(dvd::xwgradk_kernel<5, 0, 5>, 5)) return e;
  } else if (KS == 7) {
    if (int e = go(dvd::xwgradk_kernel<7, 0, 7>, 7)) return e;
  } else {
    if (int e = go(dvd::xwgradk_kernel<11, 0, 6>, 11)) return e;     // columns 0 .. 5
    if (int e = go(dvd::xwgradk_kernel<11, 6, 5>, 11)) return e;     // columns 6 .. 10 (disjoint taps of the same partials)
  }
  const long long per = (long long)KS * KS * Cout * Cin;
  hipLaunchKernelGGL(dvd::xwgrad3_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                     static_cast<const float*>(workspace), gw, p.S, KS * KS, Cout, Cin);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

int dvd_xwgrad_select(int variant) {
  DVD_REQUIRE(variant >= 0 && variant <= 2, "xwgrad_select: variant %d", variant);
  dvd::g_w1_variant = variant == 1 ? 1 : 0;
  dvd::g_w3_variant = variant == 2 ? 1 : 0;
  return DVD_OK;
}

static int xwgrad1s_impl(const void* x, const float* x_amax, const void* gy, const float* gy_amax, float* gw, void* workspace,
                         size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in, bool h16,
                         const float* out_scale, dvd_stream_t stream);

int dvd_xwgrad1s_rowsum(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, float* gy_rowsum,
                        void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in,
                        dvd_stream_t stream) {
  if (int e = dvd_xwgrad1s(x, x_amax, gy, gy_amax, gw, workspace, workspace_bytes, N, Cin, Cout, H, W, relu_in, stream)) return e;
  if (gy_rowsum) {
    hipLaunchKernelGGL(dvd::chansum_kernel, dim3(Cout), dim3(256), 0, static_cast<hipStream_t>(stream), gy, N, Cout, H * W, gy_rowsum);
    DVD_LAUNCH_OK();
  }
  return DVD_OK;
}

int dvd_xwgrad1s(const float* x, const float* x_amax, const float* gy, const float* gy_amax, float* gw, void* workspace,
                 size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in, dvd_stream_t stream) {
  return xwgrad1s_impl(x, x_amax, gy, gy_amax, gw, workspace, workspace_bytes, N, Cin, Cout, H, W, relu_in, false, nullptr, stream);
}

int dvd_xwgrad1s_h(const void* x, const void* gy, const float* out_scale, float* gw, void* workspace, size_t workspace_bytes, int N,
                   int Cin, int Cout, int H, int W, int relu_in, dvd_stream_t stream) {
  return xwgrad1s_impl(x, nullptr, gy, nullptr, gw, workspace, workspace_bytes, N, Cin, Cout, H, W, relu_in, true, out_scale, stream);
}

static int xwgrad1s_impl(const void* x, const float* x_amax, const void* gy, const float* gy_amax, float* gw, void* workspace,
                         size_t workspace_bytes, int N, int Cin, int Cout, int H, int W, int relu_in, bool h16,
                         const float* out_scale, dvd_stream_t stream) {
  DVD_REQUIRE(x && gy && gw && workspace, "xwgrad1s: null pointer");
  DVD_REQUIRE(h16 || (x_amax && gy_amax), "xwgrad1s: the operands' max|.| scalars are missing (dvd_amax)");
  DVD_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "xwgrad1s: bad shape");
  DVD_REQUIRE((long long)H * W * (long long)(Cin > Cout ? Cin : Cout) < (1ll << 31), "xwgrad1s: image too large for 32-bit offsets");
  if (dvd::wg1_wide(Cin, Cout, H * W)) {
    const int S = dvd::wg1_wide_slices(N, Cin, Cout, H * W);
    const size_t need = (size_t)S * Cout * Cin * sizeof(float);
    if (workspace_bytes < need) {
      dvd::set_error("xwgrad1s: workspace %zu < %zu bytes", workspace_bytes, need);
      return DVD_ENOSPC;
    }
    dvd::Wg3Args a;
    a.x = x;
    a.gy = gy;
    a.x_amax = x_amax;
    a.g_amax = gy_amax;
    a.partial = static_cast<float*>(workspace);
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.G = 1; a.nco = (Cout + 255) / 256;
    a.nstrips = 0; a.RS = 0; a.nrseg = 0; a.S = S;
    a.relu_in = relu_in ? 1 : 0;
    a.out_scale = out_scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(S, (Cin + 255) / 256, (Cout + 255) / 256);
    auto go = [&](auto kern) -> int {
      DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dvd::kWbLds));
      hipLaunchKernelGGL(kern, grid, dim3(512), dvd::kWbLds, s, a);
      return DVD_OK;
    };
    dvd::flops_add(DVD_FLOP_XWGRAD1B, 2.0 * N * (double)Cout * Cin * (double)H * W);
    // whole 16-pixel chunks, 256 channel rows of one image inside a 31-bit buffer range
    const bool fw = (H * W) % 16 == 0 && (long long)256 * H * W * (h16 ? 2 : 4) < (1ll << 31) && dvd::g_w3_variant != 1;
    int e;
    if (h16) e = fw ? go(dvd::xwgrad1b_kernel<true, true>) : go(dvd::xwgrad1b_kernel<true, false>);
    else e = fw ? go(dvd::xwgrad1b_kernel<false, true>) : go(dvd::xwgrad1b_kernel<false, false>);
    if (e) return e;
    DVD_LAUNCH_OK();
    const long long per = (long long)Cout * Cin;
    hipLaunchKernelGGL(dvd::xwgrad3_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), gw, S, 1, Cout, Cin);
    DVD_LAUNCH_OK();
    return DVD_OK;
  }
  const int nco = (Cout + dvd::kW1CB - 1) / dvd::kW1CB, nci = (Cin + dvd::kW1CB - 1) / dvd::kW1CB;
  const int pairs = nco * nci;
  // two blocks per CU are resident: whole rounds over 512 slots
  int S = pairs >= 512 ? 1 : (1024 + pairs - 1) / pairs;
  const long long items = (long long)N * ((H * W + dvd::kW1Chunk - 1) / dvd::kW1Chunk);
  if (S > items) S = (int)items;
  const size_t need = (size_t)S * Cout * Cin * sizeof(float);
  if (workspace_bytes < need) {
    dvd::set_error("xwgrad1s: workspace %zu < %zu bytes", workspace_bytes, need);
    return DVD_ENOSPC;
  }
  DVD_REQUIRE(nco <= 65535 && nci <= 65535, "xwgrad1s: too many channel blocks");
  dvd::Wg3Args a;
  a.x = x;
  a.gy = gy;
  a.x_amax = x_amax;
  a.g_amax = gy_amax;
  a.partial = static_cast<float*>(workspace);
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.G = 1; a.nco = nco;
  a.nstrips = 0; a.RS = 0; a.nrseg = 0; a.S = S;
  a.relu_in = relu_in ? 1 : 0;
  a.out_scale = out_scale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)2 * 2 * dvd::kW1CB * dvd::kW1Pitch;
  dvd::flops_add(DVD_FLOP_XWGRAD1S, 2.0 * N * (double)Cout * Cin * (double)H * W);
  if (h16) {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(dvd::xwgrad1s_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(dvd::xwgrad1s_kernel<true>, dim3(S, nci, nco), dim3(dvd::kW1NT), lds, s, a);
  } else {
    DVD_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(dvd::xwgrad1s_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(dvd::xwgrad1s_kernel<false>, dim3(S, nci, nco), dim3(dvd::kW1NT), lds, s, a);
  }
  DVD_LAUNCH_OK();
  const long long per = (long long)Cout * Cin;
  hipLaunchKernelGGL(dvd::xwgrad3_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                     static_cast<const float*>(workspace), gw, S, 1, Cout, Cin);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

}  // extern "C"
