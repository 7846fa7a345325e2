// 7x7 and 5x5 stride-1 "same" convolutions, fp32 in / fp32 out, the contraction on the bf16 matrix pipe with exact operand
// splits ("bf16x6"):
//   * 7x7: the five layers of SPyNet's basic module (8 -> 32 -> 64 -> 32 -> 16 -> 2 at six pyramid levels; reference
//     eavsrp_model.py:398-431, called from compute_flow eavsrp_model.py:433-488), which conv2d_mfma_kernel<7, ..> ran on the
//     fp32 MFMA (0.85 of its peak at the finest level, launch-bound below it: 28 ms of the 236 ms bench step);
//   * 5x5: the three heads of AdaptBlockOffset as one 64 -> 15 D launch (networks.py:289-315), which ran as F(2x2,5x5) on the
//     fp32 MFMA (conv_wino6_kernel<5>).
//
// Arithmetic (as dcnv2_il.hip, NPROD = 6): every fp32 operand is split EXACTLY into three bf16 terms (hi = trunc_bf16(x),
// mid = trunc_bf16(x - hi), lo = x - hi - mid; hi + mid + lo == x bit for bit), products of bf16 numbers are exact in fp32, the
// six partial products down to 2^-16 of the result are accumulated in fp32 and the three below 2^-23 (mid*lo, lo*mid, lo*lo)
// are dropped: at most 2 * 2^-24 relative per product -- one fp32 rounding.  No operand is rounded to 16 bits.
// v_mfma_f32_32x32x16_bf16 does 16 k in 32 cycles where v_mfma_f32_32x32x2_f32 does 2 k in 64: six of them are 2.67x the fp32 rate.
//
// Unlike conv_x9.hip (3x3, split in registers per use) the input is split ONCE per value, on its way into LDS: a 7x7 tap window
// reuses every input value 49 times.  Per workgroup (512 threads, 16 rows x 32 pixels, 32 MT output channels) and per chunk of
// 8 input channels (written for KS = 7; 5x5 in brackets):
//   * patch: (16+6) x 40 [(16+4) x 37] pixels x 8 channels, read from the NCHW image by plain coalesced loads one chunk ahead (16 values per
//     thread in flight), split in registers, stored channel-interleaved as three bf16 planes [plane][row][col][8 ch] (16 bytes
//     per pixel and plane): the MFMA B operand of lane (n = lane & 31, g = lane >> 5) for k-step s -- the 8 channels of pixel
//     n shifted by ITS tap 2 s + g -- is one ds_read_b128 per plane, consecutive lanes 16 bytes apart (no bank conflicts);
//   * weights: pre-split and pre-arranged by eavsr_pack_conv7_weight_x6 in MFMA A-operand order, streamed by 16-byte LDS-DMA
//     in slabs of 5 k-steps (10 taps; 25 k-steps = 49 taps + one zero tap per chunk: 2 % padding) [7 + 6 k-steps = 25 taps + one
//     zero tap: 4 %] into two stages; the slab barrier sits one k-step before the slab changes, so that the next slab's first A
//     operands are prefetched like any other.
// The main loop is LDS reads and MFMAs only: 6 + 3 MT reads of 16 bytes per 12 MT MFMAs and wave, operands read one k-step ahead.
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int S_NW = 8, S_TW = 32;
// KS: kernel size (7, 5); MT: 32-channel output tiles per workgroup; NT: image rows per wave (tile = 8 NT rows x 32 pixels).
// NT = 1 and MT = 1 exist for the coarse pyramid levels, where a launch is a handful of workgroups and its time is ONE
// workgroup's chain of k-steps.
// NP: operand planes -- 3 = the exact fp32 split (six partial products), 1 = the 16-bit modes' form (operands rounded ONCE to
// bf16 / fp16, one product; eavsr_conv_h16x1: SPyNet's 7x7 layers under BASELINE configs[2] / [4]).
template <int KS, int MT, int NT, int NP = 3> struct C7 {
  static constexpr int PAD = KS / 2, KK = KS * KS;
  static constexpr int KSTEPS = (KK + 1) / 2;                  // tap pairs per chunk: 25 (49 taps + a zero tap), 13 (25 + one)
  static constexpr int SLAB = KS == 5 ? 7 : 5;                 // k-steps per weight slab (the last slab of a 5x5 chunk has 6; 3x3: the chunk)
  static constexpr int NSLAB = (KSTEPS + SLAB - 1) / SLAB;     // 5, 2
  static constexpr int TH = S_NW * NT;
  static constexpr int IH = TH + KS - 1;                       // patch rows y0-PAD .. y0+TH+PAD-1
  static constexpr int IW = S_TW + KS;                         // columns x0-PAD .. x0+32+PAD-1, and one that stays zero (the zero tap)
  static constexpr int NPIX = IH * IW;
  static constexpr int PLANE_B = NPIX * 16;                    // bytes of one bf16 plane of the patch (8 channels per pixel)
  static constexpr int PATCH_B = NP * PLANE_B;                 // 7x7: 41,184 (NT = 2); 5x5: 35,520 (three planes)
  static constexpr int KS_U4 = NP * MT * 64;                   // 16-byte elements of one k-step's A operands
  static constexpr int SLAB_U4 = SLAB * KS_U4;                 // of a (full) weight slab
  static constexpr size_t LDS_BYTES = 2 * (size_t)PATCH_B + 2 * (size_t)SLAB_U4 * 16 + 64 * 4;   // 7x7: 144,064, 5x5: 157,312 at MT = NT = 2
  static_assert(NPIX <= 1024, "two patch pixels per thread");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct C7Args {
  const float* x;        // (n, cin, h, w)
  const u32x4* wsplit;   // [cot][chunk][k-step][plane][mt][lane] 16-byte elements
  const float* bias;
  float* out;            // (n, cout, h, w)
  int n, cin, cout, h, w, tiles_x, tiles_y;
  int act;
  float slope;
  int sig_from;      // output channels >= sig_from (a multiple of 8; -1: none) get the sigmoid instead of `act`
};

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void s_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 s_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef _Float16 s_h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 s_mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_h16x8, a), __builtin_bit_cast(s_h16x8, b), c, 0, 0, 0);
}
// two fp32 values rounded (nearest even) to one packed 16-bit pair: DT 1 = fp16, 2 = bf16
template <int DT> __device__ __forceinline__ unsigned s_round2(float a, float b) {
  if (DT == 1) return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
  typedef float f2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{a, b}, b2_));
}

#ifdef EAVSR_C7_STAMPS
// diagnostic build only (tools/build_c7_diag.sh): shader cycles per phase, summed over wave 0 of every workgroup
__device__ unsigned long long g_c7_stamps[8];
#define C7_STAMP(i)                                                   \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define C7_STAMP(i) do { } while (0)
#endif

template <int KS, int MT, int NT, int WMT, int NP = 3, int DT = 0>
__global__ __launch_bounds__(512) void conv_x6_kernel(C7Args a) {
  static_assert((NP == 3 && DT == 0) || (NP == 1 && (DT == 1 || DT == 2)), "three exact bf16 planes, or one rounded fp16 / bf16 plane");
  using K = C7<KS, MT, NT, NP>;
  constexpr int S_NT = NT, S_TH = K::TH, S_NPIX = K::NPIX, S_PLANE_B = K::PLANE_B, S_PATCH_B = K::PATCH_B;
  constexpr int S_PAD = K::PAD, S_IW = K::IW, S_KSTEPS = K::KSTEPS, S_SLAB = K::SLAB, S_NSLAB = K::NSLAB;
#ifdef EAVSR_C7_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
  unsigned char* s_patch = smem7;                                            // [2][3 planes][22][40][16 B]
  u32x4* s_w = reinterpret_cast<u32x4*>(smem7 + 2 * S_PATCH_B);              // [2][SLAB_U4]
  float* s_bias = reinterpret_cast<float*>(smem7 + 2 * S_PATCH_B + 2 * K::SLAB_U4 * 16);   // [32 MT]: the accumulators start from it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if defined(EAVSR_X6_PRIO) && EAVSR_X6_PRIO == 1      // A/B: static priority for one half of the workgroup (the two waves of a SIMD are w, w + 4)
  if (wave < 4) __builtin_amdgcn_s_setprio(1);
#elif defined(EAVSR_X6_PRIO) && EAVSR_X6_PRIO == 2
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int l31 = lane & 31, kg = lane >> 5;

  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * S_TH, x0 = tx * S_TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const int nch = a.cin >> 3;

  // ---- patch producer: thread t owns patch pixels t and t + 512 --------------------------------------------------
  bool pok[2];
  unsigned pgo[2], plo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + i * 512;
    const int r = p / S_IW, c = p - r * S_IW;
    const int gy = y0 - S_PAD + r, gx = x0 - S_PAD + c;
    // the last column is never a tap of anybody: kept zero (the zero tap of the last k-step reads it, below)
    pok[i] = p < S_NPIX && c < S_IW - 1 && gy >= 0 && gy < h && gx >= 0 && gx < w;
    pgo[i] = pok[i] ? (unsigned)(gy * w + gx) : 0u;
    plo[i] = (unsigned)p * 16u;
  }
  const bool second = tid + 512 < S_NPIX;    // 7x7: 346 (NT = 2) / 34 (NT = 1) threads own a second pixel
  float pv[2][8];
  auto load_patch = [&](int ch) __attribute__((always_inline)) {
    const float* sp = a.x + ((size_t)bn * a.cin + (size_t)ch * 8) * plane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[i][j] = pok[i] ? sp[(size_t)j * plane + pgo[i]] : 0.f;
  };
  auto store_patch = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 0 ? tid >= S_NPIX : !second) break;          // (5x5 on 8-row tiles: 444 patch pixels)
      u32x4 pl[NP];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (NP == 3) {
          unsigned h2, m2, l2;
          s_split2(pv[i][2 * c], pv[i][2 * c + 1], h2, m2, l2);
          pl[0][c] = h2; pl[1][c] = m2; pl[2][c] = l2;
        } else {
          pl[0][c] = s_round2<DT>(pv[i][2 * c], pv[i][2 * c + 1]);
        }
      }
#pragma unroll
      for (int p3 = 0; p3 < NP; ++p3)
        *reinterpret_cast<u32x4*>(s_patch + stage * S_PATCH_B + p3 * S_PLANE_B + plo[i]) = pl[p3];
    }
  };
  // ---- weight slabs: global slab index gs = chunk * NSLAB + si, LDS stage gs & 1 --------------------------------------
  // (the packed weight holds WMT >= MT 32-channel tiles per `cot`, one-KiB pieces in [k-step][plane][tile] order: a workgroup of
  //  fewer tiles picks its pieces out of the slab)
  const int nslabs = nch * S_NSLAB;
  constexpr int wsplit_n = WMT / MT;                        // workgroups per packed cot
  const int wcot = cot / wsplit_n, wsub = (cot - wcot * wsplit_n) * MT;
  auto issue_slab = [&](int gs, int ch, int si) __attribute__((always_inline)) {      // si (slab of its chunk) is a constant at every call
    const int nks = (si + 1) * S_SLAB <= S_KSTEPS ? S_SLAB : S_KSTEPS - si * S_SLAB;   // k-steps of this slab
    const int segs = nks * NP * MT;
    const char* wsrc = reinterpret_cast<const char*>(a.wsplit + (((size_t)wcot * nch + ch) * S_KSTEPS + si * S_SLAB) * (K::KS_U4 * wsplit_n));
    u32x4* dst = s_w + (gs & 1) * K::SLAB_U4;
#pragma unroll
    for (int i = 0; i < (S_SLAB * NP * MT + S_NW - 1) / S_NW; ++i) {
      const int seg = i * S_NW + wave;                      // piece (k-step, plane, tile m) = seg / MT, seg % MT of this workgroup
      if (seg < segs) {  // wave-uniform
        const int sp = MT == 1 ? seg * WMT + wsub : (seg >> 1) * WMT + wsub + (seg & 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(sp * 64 + lane) * 16u), (lptr_t)(dst + seg * 64), 16, 0, 0);
      }
    }
  };

  f32x16 acc[MT][S_NT];

  // B operand address of this lane: pixel (wave * 2 + row + ky, l31 + kx) of the patch, tap (ky, kx) = 2 s + kg.  The tap pair
  // of a k-step is (kx, kx + 1) of one kernel row, or (6 of row ky, 0 of row ky + 1): two lane bases, everything else immediate.
  // The last k-step pairs the last tap with a tap that does not exist: its weights are zero and its lanes read the always-zero
  // last column of patch row 0 / 1 (0 x 0, whatever the image holds next to the window).
  const int bbase = ((wave * S_NT) * S_IW + l31) * 16;
  const int b_same = bbase + (kg ? 16 : 0);
  const int b_wrap = bbase + (kg ? (S_IW - (KS - 1)) * 16 : 0);
  const int b_last = kg ? (S_IW - 1) * 16 - ((KS - 1) * S_IW + KS - 1) * 16 : bbase;
  auto read_b = [&](int stage, int s, u32x4 (&b)[S_NT][NP]) __attribute__((always_inline)) {
    const int tap0 = 2 * s, ky = tap0 / KS, kx = tap0 - KS * ky;
    const unsigned char* base = s_patch + stage * S_PATCH_B + ((s == S_KSTEPS - 1 ? b_last : kx == KS - 1 ? b_wrap : b_same) + (ky * S_IW + kx) * 16);
#pragma unroll
    for (int t = 0; t < S_NT; ++t)
#pragma unroll
      for (int p3 = 0; p3 < NP; ++p3) b[t][p3] = *reinterpret_cast<const u32x4*>(base + t * S_IW * 16 + p3 * S_PLANE_B);
  };
  auto read_a = [&](int gs, int sl, u32x4 (&av)[NP][MT]) __attribute__((always_inline)) {
    const u32x4* ws = s_w + (gs & 1) * K::SLAB_U4 + sl * K::KS_U4 + lane;
#pragma unroll
    for (int p3 = 0; p3 < NP; ++p3)
#pragma unroll
      for (int m = 0; m < MT; ++m) av[p3][m] = ws[(p3 * MT + m) * 64];
  };

  // ---- prologue --------------------------------------------------------------------------------------------------
  issue_slab(0, 0, 0);
  load_patch(0);
  if (tid < 32 * MT) {
    const int co = cot * 32 * MT + tid;
    s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
  }
  store_patch(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the bias is the accumulators' initial value (32 dependent global loads in the epilogue were 10 % of a workgroup's time)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = s_bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
#pragma unroll
      for (int t = 0; t < S_NT; ++t) acc[m][t][r] = bv;
    }
  if (nslabs > 1) issue_slab(1, S_NSLAB > 1 ? 0 : 1, S_NSLAB > 1 ? 1 : 0);
  u32x4 acur[NP][MT], bcur[S_NT][NP];
  read_a(0, 0, acur);
  read_b(0, 0, bcur);
  C7_STAMP(0);      // prologue

  for (int ch = 0; ch < nch; ++ch) {
    const int pst = ch & 1;
    const bool more = ch + 1 < nch;
    if (more) load_patch(ch + 1);
#pragma unroll
    for (int ks = 0; ks < S_KSTEPS; ++ks) {
      constexpr int STORE_KS = KS == 7 ? 11 : KS == 5 ? 7 : 3;   // (7x7, 5x5: behind a slab barrier, the chunk's loads have landed by its vmcnt(0))
      const int si = ks / S_SLAB, sl = ks % S_SLAB;
      const int gs = ch * S_NSLAB + si;
      const bool slab_end = sl == S_SLAB - 1 || ks == S_KSTEPS - 1;
      if (slab_end) {
        // the slab barrier, one k-step early: slab gs + 1 has landed (requested a slab ago), every wave has read the last A
        // operands of slab gs (they were prefetched in the previous k-step), so its stage takes slab gs + 2
        C7_STAMP(1);  // k-steps
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C7_STAMP(2);  // own DMA / loads outstanding
        __syncthreads();
        C7_STAMP(3);  // barrier
        if (gs + 2 < nslabs) issue_slab(gs + 2, ch + (si + 2) / S_NSLAB, (si + 2) % S_NSLAB);
        C7_STAMP(4);  // DMA issue
      }
      if (ks == STORE_KS && more) store_patch(pst ^ 1);     // published by the slab barriers behind it
      // operands of the next k-step
      u32x4 anext[NP][MT], bnext[S_NT][NP];
      const bool last = ks == S_KSTEPS - 1;
      if (!last) {
        read_a(slab_end ? gs + 1 : gs, slab_end ? 0 : sl + 1, anext);
        read_b(pst, ks + 1, bnext);
      } else if (more) {
        read_a(gs + 1, 0, anext);
        read_b(pst ^ 1, 0, bnext);
      }
      // the six partial products, smallest first: (A plane, B plane) = (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
#pragma unroll
      for (int t = 0; t < S_NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x16 c_ = acc[m][t];
          if constexpr (NP == 3) {
            c_ = s_mfma(acur[2][m], bcur[t][0], c_);
            c_ = s_mfma(acur[0][m], bcur[t][2], c_);
            c_ = s_mfma(acur[1][m], bcur[t][1], c_);
            c_ = s_mfma(acur[1][m], bcur[t][0], c_);
            c_ = s_mfma(acur[0][m], bcur[t][1], c_);
            c_ = s_mfma(acur[0][m], bcur[t][0], c_);
          } else if constexpr (DT == 1) {
            c_ = s_mfma_f16(acur[0][m], bcur[t][0], c_);
          } else {
            c_ = s_mfma(acur[0][m], bcur[t][0], c_);
          }
          acc[m][t] = c_;
        }
      // issue order: the next k-step's reads go out between the first MFMAs of this one (one MFMA, one read, ..), the rest of
      // the MFMAs cover their latency; nothing crosses the end of the k-step (the scheduler otherwise sinks every read to just
      // in front of its first use, behind an s_waitcnt lgkmcnt(0))
      {
        constexpr int NRD = NP * MT + NP * S_NT, NMF = (NP == 3 ? 6 : 1) * MT * S_NT;
        constexpr int NIL = NRD < NMF ? NRD : NMF;      // (one product per operand pair: fewer MFMAs than reads on the small tiles)
#pragma unroll
        for (int i = 0; i < NIL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // (the MFMA first: its wait for the operands read during the
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      //  previous k-step then does not cover a read issued just now)
        }
        if constexpr (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
        if constexpr (NRD > NMF) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMF, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!last || more) {
#pragma unroll
        for (int p3 = 0; p3 < NP; ++p3) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acur[p3][m] = anext[p3][m];
#pragma unroll
          for (int t = 0; t < S_NT; ++t) bcur[t][p3] = bnext[t][p3];
        }
      }
    }
  }

  C7_STAMP(1);
  // ---- epilogue: bias, activation, NCHW stores (lanes 0-31 / 32-63: 32 consecutive pixels of two channels 4 apart) ----
  int gx = x0 + l31;
  asm volatile("" : "+v"(gx));      // keeps the address arithmetic below here (hoisted above the loop its 64-bit results spill)
  const bool xok = gx < w;
  const float act_s = a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope;
  const int co0 = cot * 32 * MT + 4 * kg;
  // lane part of the address once; the channel advances by one plane per register, five across a group of four
  float* ob = a.out + ((size_t)bn * a.cout + co0) * plane + (size_t)(y0 + wave * S_NT) * w + gx;
  const bool full = cot * 32 * MT + 32 * MT <= a.cout;      // wave-uniform
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = m * 32 + (r & 3) + 8 * (r >> 2);
      const bool cok = full || co0 + cu < a.cout;
      // the mask logits of the predictor heads leave as masks (networks.py:314): wave-uniform per group of eight channels
      const bool sg = a.sig_from >= 0 && cot * 32 * MT + m * 32 + 8 * (r >> 2) >= a.sig_from;
#pragma unroll
      for (int t = 0; t < S_NT; ++t) {
        float v = acc[m][t][r];
        if (sg) v = eavsr_sigmoid_fast(v);
        else v = eavsr_act(v, act_s);      // branch-free: max(v, v s), 0 <= s <= 1
        if (cok && xok && y0 + wave * S_NT + t < h) ob[(size_t)cu * plane + (size_t)t * w] = v;
      }
    }
#ifdef EAVSR_C7_STAMPS
  C7_STAMP(5);      // epilogue issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C7_STAMP(6);      // stores acknowledged
  if (tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_c7_stamps[i], st_acc[i]);
#endif
}

// (cout, cin, KS, KS) fp32 -> [cot][chunk][k-step][plane][mt][lane] 16-byte elements: lane (r = lane & 31, g = lane >> 5) holds
// A[row r][k = 8 g + j] = W[cot * 32 MT + mt * 32 + r][chunk * 8 + j][tap 2 s + g], j = 0..7, plane 0 / 1 / 2 = hi / mid / lo
// tr: `wt` is the FORWARD weight (cin, cout, k, k) of the convolution whose input gradient this launch packs for -- transposed and
// flipped on the fly: element (co, ci, tap) = wt[ci][co][kk - 1 - tap]
__global__ void pack7_kernel(const float* __restrict__ wt, u32x4* __restrict__ p, int cout, int cin, int kk, int mt_n, long total,
                             int np = 3, int dt = 0, int tr = 0) {
  const int ksteps = (kk + 1) / 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  long q = e;
  const int lane = (int)(q % 64); q /= 64;
  const int mt = (int)(q % mt_n); q /= mt_n;
  const int pl = (int)(q % np); q /= np;
  const int s = (int)(q % ksteps); q /= ksteps;
  const int nch = cin / 8;
  const int ch = (int)(q % nch);
  const int cot = (int)(q / nch);
  const int co = cot * 32 * mt_n + mt * 32 + (lane & 31);
  const int tap = 2 * s + (lane >> 5);
  u32x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ci = ch * 8 + 2 * c + u;
      v[u] = (co < cout && tap < kk) ? (tr ? wt[((size_t)ci * cout + co) * kk + (kk - 1 - tap)] : wt[((size_t)co * cin + ci) * kk + tap]) : 0.f;
    }
    if (np == 3) {
      unsigned h2, m2, l2;
      s_split2(v[0], v[1], h2, m2, l2);
      o[c] = pl == 0 ? h2 : pl == 1 ? m2 : l2;
    } else {
      o[c] = dt == 1 ? s_round2<1>(v[0], v[1]) : s_round2<2>(v[0], v[1]);
    }
  }
  p[e] = o;
}

// pack7_kernel for up to PM_MAX weights of ONE shape in one launch (blockIdx.y = weight; pointers by value in the kernel arguments,
// so that the launch captures into a HIP graph): the training step re-packs every 3x3 64 -> 64 weight and its input-gradient form
// once per step -- 540 launches of 4.7 us on the step's one dependent chain as single-weight launches (round 6)
constexpr int PM_MAX = 48;
struct PackMultiArgs {
  const float* src[PM_MAX];
  u32x4* dst[PM_MAX];
  unsigned long long tr_mask;      // bit t: weight t is packed transposed and flipped (its input-gradient form)
};
__global__ void pack7_multi_kernel(PackMultiArgs a, int cout, int cin, int kk, int mt_n, long total) {
  const int t = blockIdx.y;
  const int tr = (int)((a.tr_mask >> t) & 1ull);
  const float* __restrict__ wt = a.src[t];
  u32x4* __restrict__ p = a.dst[t];
  const int ksteps = (kk + 1) / 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  long q = e;
  const int lane = (int)(q % 64); q /= 64;
  const int mt = (int)(q % mt_n); q /= mt_n;
  const int pl = (int)(q % 3); q /= 3;
  const int s = (int)(q % ksteps); q /= ksteps;
  const int nch = cin / 8;
  const int ch = (int)(q % nch);
  const int cot = (int)(q / nch);
  const int co = cot * 32 * mt_n + mt * 32 + (lane & 31);
  const int tap = 2 * s + (lane >> 5);
  u32x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ci = ch * 8 + 2 * c + u;
      v[u] = (co < cout && tap < kk) ? (tr ? wt[((size_t)ci * cout + co) * kk + (kk - 1 - tap)] : wt[((size_t)co * cin + ci) * kk + tap]) : 0.f;
    }
    unsigned h2, m2, l2;
    s_split2(v[0], v[1], h2, m2, l2);
    o[c] = pl == 0 ? h2 : pl == 1 ? m2 : l2;
  }
  p[e] = o;
}

int mt_of(int cout) { return cout > 32 ? 2 : 1; }

template <int KS, int MT, int NT, int WMT, int NP = 3, int DT = 0>
int launch7(const C7Args& a, void* stream) {
  using K = C7<KS, MT, NT, NP>;
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once_pd.flag[dev_], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6_kernel<KS, MT, NT, WMT, NP, DT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv_f32x6: hipFuncSetAttribute(%zu B of LDS): %s", K::LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  C7Args b = a;
  b.tiles_y = eavsr::cdiv(a.h, K::TH);
  const long blocks = (long)b.tiles_x * b.tiles_y * b.n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv_f32x6: too many tiles");
  dim3 grid((unsigned)blocks, eavsr::cdiv(a.cout, 32 * MT));
  hipLaunchKernelGGL((conv_x6_kernel<KS, MT, NT, WMT, NP, DT>), grid, dim3(64 * S_NW), K::LDS_BYTES, eavsr::as_stream(stream), b);
  return eavsr::launch_status("conv_f32x6");
}

// Tile height and channel tiles per workgroup: the full 16-row, all-channel workgroup where that gives the GPU enough of them;
// otherwise 8-row tiles, then one 32-channel tile per workgroup -- a coarse pyramid level is a few dozen workgroups and takes as
// long as ONE of them.  Every output is the same sum in the same order whichever shape computes it.
template <int KS, int NP = 3, int DT = 0>
int dispatch7(const C7Args& a, void* stream) {
  const long wg2 = (long)a.tiles_x * eavsr::cdiv(a.h, 16) * a.n;
  const long wg1 = (long)a.tiles_x * eavsr::cdiv(a.h, 8) * a.n;
  EAVSR_REQUIRE(wg1 * 2 < (1L << 31), -1, "conv_f32x6: too many tiles");
  if (mt_of(a.cout) == 2) {     // (the packed weight has two 32-channel tiles per `cot`)
    if (wg2 >= 128) return launch7<KS, 2, 2, 2, NP, DT>(a, stream);
    if (wg1 >= 128) return launch7<KS, 2, 1, 2, NP, DT>(a, stream);
    return launch7<KS, 1, 1, 2, NP, DT>(a, stream);
  }
  if (wg2 >= 128) return launch7<KS, 1, 2, 1, NP, DT>(a, stream);
  return launch7<KS, 1, 1, 1, NP, DT>(a, stream);
}

}  // namespace

#ifdef EAVSR_C7_STAMPS
extern "C" int eavsr_debug_c7_stamps(unsigned long long* host_out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c7_stamps), sizeof(unsigned long long) * 8);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_c7_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif

extern "C" size_t eavsr_conv_weight_x6_bytes(int32_t ksize, int32_t cout, int32_t cin) {
  if ((ksize != 3 && ksize != 5 && ksize != 7) || cout <= 0 || cin <= 0 || cin % 8) return 0;
  const int mt = mt_of(cout);
  return (size_t)eavsr::cdiv(cout, 32 * mt) * (cin / 8) * ((ksize * ksize + 1) / 2) * 3 * mt * 64 * 16;
}

extern "C" int eavsr_pack_conv_weight_x6(const float* weight, void* packed, int32_t ksize, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight_x6: NULL pointer");
  EAVSR_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, -2, "pack_conv_weight_x6: kernel size %d (3, 5 and 7 only)", ksize);
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % 8 == 0, -1, "pack_conv_weight_x6: cin %d must be a multiple of 8", cin);
  const long total = (long)(eavsr_conv_weight_x6_bytes(ksize, cout, cin) / 16);
  hipLaunchKernelGGL(pack7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight,
                     reinterpret_cast<u32x4*>(packed), cout, cin, ksize * ksize, mt_of(cout), total);
  return eavsr::launch_status("pack_conv_weight_x6");
}

extern "C" int eavsr_pack_conv_weight_x6_dgrad(const float* weight, void* packed, int32_t ksize, int32_t cout_w, int32_t cin_w, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight_x6_dgrad: NULL pointer");
  EAVSR_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, -2, "pack_conv_weight_x6_dgrad: kernel size %d (3, 5 and 7 only)", ksize);
  EAVSR_REQUIRE(cout_w > 0 && cin_w > 0 && cout_w % 8 == 0, -1, "pack_conv_weight_x6_dgrad: cout %d must be a multiple of 8", cout_w);
  // the input-gradient convolution has cin_w output and cout_w input channels
  const long total = (long)(eavsr_conv_weight_x6_bytes(ksize, cin_w, cout_w) / 16);
  hipLaunchKernelGGL(pack7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight,
                     reinterpret_cast<u32x4*>(packed), cin_w, cout_w, ksize * ksize, mt_of(cin_w), total, 3, 0, 1);
  return eavsr::launch_status("pack_conv_weight_x6_dgrad");
}

// `count` weights of one SQUARE shape (c, c, ksize, ksize) in ceil(count / 48) launches: weights[t] -> packed[t], transposed[t] != 0:
// the input-gradient form (eavsr_pack_conv_weight_x6_dgrad).  The three arrays are HOST arrays, read at the call.
extern "C" int eavsr_pack_conv_weight_x6_multi(const float* const* weights, void* const* packed, const int32_t* transposed, int32_t count,
                                               int32_t ksize, int32_t c, void* stream) {
  EAVSR_REQUIRE(count >= 0 && (count == 0 || (weights && packed && transposed)), -1, "pack_conv_weight_x6_multi: NULL pointer");
  EAVSR_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, -2, "pack_conv_weight_x6_multi: kernel size %d (3, 5 and 7 only)", ksize);
  EAVSR_REQUIRE(c > 0 && c % 8 == 0, -1, "pack_conv_weight_x6_multi: %d channels must be a multiple of 8", c);
  const long total = (long)(eavsr_conv_weight_x6_bytes(ksize, c, c) / 16);
  for (int t0 = 0; t0 < count; t0 += PM_MAX) {
    PackMultiArgs a;
    a.tr_mask = 0ull;
    const int m = count - t0 < PM_MAX ? count - t0 : PM_MAX;
    for (int t = 0; t < PM_MAX; ++t) {
      const int u = t < m ? t0 + t : t0;      // (unused slots repeat a valid pair; blockIdx.y never reaches them)
      EAVSR_REQUIRE(weights[u] && packed[u], -1, "pack_conv_weight_x6_multi: NULL pointer in entry %d", u);
      a.src[t] = weights[u];
      a.dst[t] = reinterpret_cast<u32x4*>(packed[u]);
      if (t < m && transposed[u]) a.tr_mask |= 1ull << t;
    }
    hipLaunchKernelGGL(pack7_multi_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)m), dim3(256), 0, eavsr::as_stream(stream), a, c, c,
                       ksize * ksize, mt_of(c), total);
  }
  return eavsr::launch_status("pack_conv_weight_x6_multi");
}

extern "C" int eavsr_conv_f32x6(const float* x, const void* weight_x6, const float* bias, float* out, int32_t n, int32_t cin,
                                int32_t cout, int32_t h, int32_t w, int32_t ksize, int32_t act, float slope, int32_t sigmoid_from,
                                void* stream) {
  // (KS = 3 instantiates and is correct, but at the bench's launch shape it takes 51 us against the fp32 Winograd kernel's 41:
  //  tools/gpu_conv3_x6_time.py, round 5; small 3x3 launches have their own kernel, conv3_x6s.hip)
#ifndef EAVSR_X6_KS3
  EAVSR_REQUIRE(ksize == 5 || ksize == 7, -2, "conv_f32x6: kernel size %d (5 and 7 only; 3x3 is eavsr_conv2d_f32 / eavsr_conv3x3_wino4_f32 / eavsr_conv3x3_f32x6s)", ksize);
#endif
  EAVSR_REQUIRE(n >= 0 && cin > 0 && cout > 0 && h > 0 && w > 0, -1, "conv_f32x6: bad dims");
  if (n == 0) return 0;      // (an empty batch has no buffers)
  EAVSR_REQUIRE(x && weight_x6 && out, -1, "conv_f32x6: NULL pointer");
  EAVSR_REQUIRE(cin % 8 == 0, -2, "conv_f32x6: cin %d must be a multiple of 8 (use eavsr_conv2d_f32)", cin);
  EAVSR_REQUIRE(act >= 0 && act <= 2, -1, "conv_f32x6: act %d", act);
  EAVSR_REQUIRE(act != EAVSR_ACT_LRELU || (slope >= 0.f && slope <= 1.f), -2,
                "conv_f32x6: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)slope);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "conv_f32x6: image plane too large for 32-bit pixel offsets");
  EAVSR_REQUIRE(sigmoid_from < 0 || (sigmoid_from % 8 == 0 && sigmoid_from < cout), -2,
                "conv_f32x6: sigmoid_from %d (a multiple of 8 below cout, or -1)", sigmoid_from);
  C7Args a;
  a.x = x; a.wsplit = reinterpret_cast<const u32x4*>(weight_x6); a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, S_TW);
  a.tiles_y = 0;   // per tile height, in launch7
  a.act = act; a.slope = slope; a.sig_from = sigmoid_from < 0 ? -1 : sigmoid_from;
#ifdef EAVSR_X6_KS3      // A/B build only (tools/gpu_conv3_x6_time.py)
  if (ksize == 3) return dispatch7<3>(a, stream);
#endif
  return ksize == 7 ? dispatch7<7>(a, stream) : dispatch7<5>(a, stream);
}

// ---- the 16-bit modes' form of the same kernel (NP = 1): operands rounded once to fp16 (dtype 1) / bf16 (dtype 2), one product ----
extern "C" size_t eavsr_conv_weight_h16x1_bytes(int32_t ksize, int32_t cout, int32_t cin) {
  return eavsr_conv_weight_x6_bytes(ksize, cout, cin) / 3;
}

extern "C" int eavsr_pack_conv_weight_h16x1(const float* weight, void* packed, int32_t ksize, int32_t cout, int32_t cin, int32_t dtype,
                                            void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight_h16x1: NULL pointer");
  EAVSR_REQUIRE(ksize == 7, -2, "pack_conv_weight_h16x1: kernel size %d (7 only)", ksize);
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "pack_conv_weight_h16x1: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % 8 == 0, -1, "pack_conv_weight_h16x1: cin %d must be a multiple of 8", cin);
  const long total = (long)(eavsr_conv_weight_h16x1_bytes(ksize, cout, cin) / 16);
  hipLaunchKernelGGL(pack7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight,
                     reinterpret_cast<u32x4*>(packed), cout, cin, ksize * ksize, mt_of(cout), total, 1, dtype);
  return eavsr::launch_status("pack_conv_weight_h16x1");
}

extern "C" int eavsr_conv_h16x1(const float* x, const void* weight_h16x1, const float* bias, float* out, int32_t n, int32_t cin,
                                int32_t cout, int32_t h, int32_t w, int32_t ksize, int32_t act, float slope, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(ksize == 7, -2, "conv_h16x1: kernel size %d (7 only)", ksize);
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv_h16x1: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && cin > 0 && cout > 0 && h > 0 && w > 0, -1, "conv_h16x1: bad dims");
  if (n == 0) return 0;
  EAVSR_REQUIRE(x && weight_h16x1 && out, -1, "conv_h16x1: NULL pointer");
  EAVSR_REQUIRE(cin % 8 == 0, -2, "conv_h16x1: cin %d must be a multiple of 8", cin);
  EAVSR_REQUIRE(act >= 0 && act <= 2, -1, "conv_h16x1: act %d", act);
  EAVSR_REQUIRE(act != EAVSR_ACT_LRELU || (slope >= 0.f && slope <= 1.f), -2, "conv_h16x1: leaky-ReLU slope %g outside [0, 1]", (double)slope);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "conv_h16x1: image plane too large for 32-bit pixel offsets");
  C7Args a;
  a.x = x; a.wsplit = reinterpret_cast<const u32x4*>(weight_h16x1); a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.cout = cout; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, S_TW);
  a.tiles_y = 0;
  a.act = act; a.slope = slope; a.sig_from = -1;
  return dtype == 1 ? dispatch7<7, 1, 1>(a, stream) : dispatch7<7, 1, 2>(a, stream);
}
