// DCNv2 forward, round-4 schedule of the hot-path kernel (a7; north star: >= 30 % of the HBM roofline).
//
// Reference semantics: mmcv.ops.modulated_deform_conv2d as called at models/networks.py:627-630, optionally with the
// affine -> 18 offsets expansion and the mask sigmoid of AdaptBlockOffset (networks.py:302-315) folded into the sampler
// ("heads" mode), exactly as csrc/dcnv2_il.hip (round 2), whose arithmetic per sample and per product this kernel keeps:
// IL8 input, both fp32 operands split exactly into three bf16 terms, NPROD = 6 or 9 partial products on
// v_mfma_f32_32x32x16_bf16, lane (n = lane & 31, kg = lane >> 5) samples one (pixel, tap) per k-step and feeds its eight
// blended, split values straight from registers.
//
// What changed against dcnv2_il.hip, and why (that kernel is bound by vector-instruction
// issue: ~880 vector instructions per wave and (tile, group) step where the sampler's own arithmetic is ~530):
//   * TAP PAIRING ACROSS TWO GROUPS.  A "pair step" is two 8-channel groups = 18 taps = NINE full k-steps: k-step u holds
//     taps 2u, 2u+1 of the 18-tap sequence (even group taps 0..8, then odd group taps 0..8); u = 4 straddles (tap 8 of the
//     even group on the kg = 0 lanes, tap 0 of the odd group on the kg = 1 lanes).  The padded tenth half k-step of every
//     group (10 % of the sampler's instructions and of the MFMAs) is gone.
//   * ONE UNIFORM SOFTWARE PIPELINE over all k-steps of a workgroup's run: during the MFMAs of k-step T the wave sets up
//     k-step T+2 and gathers / blends / splits k-step T+1, across group, pair and tile boundaries alike.  There is no
//     per-step prologue any more (it was 1.25 K of ~11 K cycles per step with no MFMA to hide behind); a run has one.
//   * LDS: three window slots (the even group's window is live k-steps 8' .. 3, the odd group's 3 .. 7: two live + one
//     landing), weights in k-step units: u = 0..2 and u = 4..7 single-buffered, u = 3 and u = 8 (read on both sides of a
//     barrier) double-buffered.  Two barriers per pair step (tops of k-steps 3 and 8) = one per group, as before.
//   * WINDOW AS TWO CHANNEL-HALF PLANES ([half][row][col] x 16 B): a sample's eight ds_read_b128 are one address register
//     + immediate offsets, conflict-free for neighbouring columns without the per-sample swizzle arithmetic.
//   * instruction diet: sigmoid by v_rcp_f32 (was a full IEEE division: 10 instructions per sample), out-of-window
//     bookkeeping on the scalar unit, parameters reloaded in place right after their use (no per-step copies), output
//     stores and parameter loads through scalar bases.
// Results differ from dcnv2_il.hip only by the re-association that the new k-step composition implies (and, in heads
// mode, by <= 1 ulp of the mask from v_rcp_f32); tests compare both against the oracle and an fp64 evaluation.
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int JT_ROWS = 8, JT_W = 32;                // pixel tile: one 32-pixel row per wave
constexpr int JG = 8;                                // channels per group = k of one tap
constexpr int JU = 9;                                // k-steps per pair step
constexpr int JPH = JT_ROWS + 12, JPW = 48;          // LDS window rows y0-6 .. y0+13, columns x0-8 .. x0+39
constexpr int JPY0 = 6, JPX0 = 8;
constexpr int JROW_B = JPW * 16;                     // 768: one window row of one channel half
constexpr int JPLANE_B = JPH * JROW_B;               // 15,360
constexpr int JWIN_B = 2 * JPLANE_B;                 // 30,720 = 30 one-KiB DMA pieces
constexpr int JWIN_SEGS = JWIN_B / 1024;
constexpr int JKS_B = 3 * 2 * 64 * 16;               // 6,144: the A operands of one k-step ([term][mt][lane] x 16 B)
constexpr int JPAIR_U4 = JU * 3 * 2 * 64;            // 16-byte elements of one pair's weight slab
// LDS map (bytes)
constexpr int JL_A = 0;                              // k-steps 0..2
constexpr int JL_U3 = 3 * JKS_B;                     // k-step 3, two parity slots
constexpr int JL_B2 = 5 * JKS_B;                     // k-steps 4..7
constexpr int JL_U8 = 9 * JKS_B;                     // k-step 8, two parity slots
constexpr int JL_WIN = 11 * JKS_B;                   // 67,584: three window slots
constexpr int JL_BIAS = JL_WIN + 3 * JWIN_B;         // 159,744: [kg][32] accumulator start values
constexpr size_t JLDS_BYTES = JL_BIAS + 2 * 32 * 4;  // 160,000

__device__ __attribute__((aligned(16))) float g_il2_zero[4] = {0.f, 0.f, 0.f, 0.f};   // window units outside the image

struct IL2Args {
  const float* xil;      // [n][cin/8][h][w][8]
  const float* offset;   // explicit mode: (n, dg*18, h, w);  heads mode: (n, 15*dg, h, w)
  const float* mask;     // explicit mode: (n, dg*9, h, w);   heads mode: unused
  const u32x4* wpair;    // [cot][pair][u][term][mt][lane] 16-byte elements (eavsr_pack_dcn_weight_il2)
  const float* bias;
  float* out;            // (n, cout, h, w)
  int n, cin, h, w, cout, dg, opg_shift, tiles_x, tiles_y, ntiles;
};

// One dword at (wave-uniform base) + (per-lane 32-bit byte offset): global_load / global_store v_off, s[base].  The base
// goes through an empty asm as an integer so that it IS a scalar register pair at the access: left alone, the compiler
// re-associates such addresses into (common base + lane offset) + uniform strides, i.e. one 64-bit VECTOR addition per
// access and a vector register pair per address.
typedef const __attribute__((address_space(1))) char* j_gcp;
typedef __attribute__((address_space(1))) char* j_gp;
__device__ __forceinline__ float ld_b(const char* base, unsigned byte_off) {
  unsigned long long b = reinterpret_cast<unsigned long long>(base);
  asm volatile("" : "+s"(b));
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<j_gcp>(b) + byte_off);
}
__device__ __forceinline__ void st_b(char* base, unsigned byte_off, float v) {
  unsigned long long b = reinterpret_cast<unsigned long long>(base);
  asm volatile("" : "+s"(b));
  *reinterpret_cast<__attribute__((address_space(1))) float*>(reinterpret_cast<j_gp>(b) + byte_off) = v;
}

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void j_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 j_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// sampling position of one (pixel, tap): the one place where it is computed (pipeline set-up and global fix-up must take
// the same in-window decision, so the products and sums are spelled out: no contraction choice is left to the compiler)
template <int HEADS>
__device__ __forceinline__ void j_position(const float (&t)[6], float dyx, float dxx, float ry, float rx, float fgy, float fgx,
                                           float& py, float& px) {
  float dy, dx;
  if (HEADS) {   // (T . R)[:,k] - R[:,k] + t   (matmul, subtract, add: networks.py:304-311)
    dy = (__builtin_fmaf(t[1], rx, __fmul_rn(t[0], ry)) - ry) + t[4];
    dx = (__builtin_fmaf(t[3], rx, __fmul_rn(t[2], ry)) - rx) + t[5];
  } else {
    dy = dyx;
    dx = dxx;
  }
  py = (fgy + ry) + dy;
  px = (fgx + rx) + dx;
}

__device__ __forceinline__ float j_sigmoid(float x) { return eavsr_sigmoid_fast(x); }

#ifdef EAVSR_IL2_STAMPS
// diagnostic build only (tools/build_il2_diag.sh): shader cycles per phase, summed over wave 0 and wave 4 of every workgroup
__device__ unsigned long long g_il2_stamps[32];
#define J_STAMP(i)                                                    \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define J_STAMP(i) do { } while (0)
#endif

// HEADS: 0 explicit offsets / masks;  1 predictor heads (mask logits);  2 predictor heads whose masks already went through the sigmoid
template <int NPROD, int HEADS>
__global__ __launch_bounds__(512, 2) void dcnv2_il2_kernel(IL2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const size_t pl4 = plane * 4;
  const unsigned upl4 = uplane * 4u;
  const int ngroups = a.cin / JG;
  const int npairs = ngroups >> 1;
  const int cot = blockIdx.y;
  const int D = a.dg;
  const bool same_dg = a.opg_shift > 0;      // both groups of a pair belong to one deformable group
  // timing ablations (tools/build_il2_diag.sh, -DEAVSR_IL2_EXP_*: results are wrong by construction): `never` is false at run
  // time and unknown at compile time
  const bool never = a.n < 0;
  (void)never;

  // persistent tile walk, XCD-aware (as dcnv2_il.hip): workgroups b, b + 8, .. share an XCD; XCD x owns a contiguous run
  const int nb = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd_wg = (nb + 7 - xcd) >> 3;
  const int tq = a.ntiles >> 3, tr = a.ntiles & 7;
  const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int t_count = tq + (xcd < tr ? 1 : 0);
  const int my_tiles = slot < t_count ? (t_count - slot + per_xcd_wg - 1) / per_xcd_wg : 0;
  if (my_tiles == 0) return;
  auto tile_of = [&](int i, int& bn, int& y0, int& x0) __attribute__((always_inline)) {
    int t = t_begin + slot + i * per_xcd_wg;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    bn = t / a.tiles_y;
    y0 = ty * JT_ROWS;
    x0 = tx * JT_W;
  };

  // ---- window DMA: piece p = i * 8 + wave (p < 30) is 64 sixteen-byte units e = p * 64 + lane of [half][row][col] -------
  constexpr int WIN_IT = (JWIN_SEGS + 7) / 8;   // 4
  int prc[WIN_IT];
  unsigned poff[WIN_IT];
#ifdef EAVSR_IL2_EXP_CONTIG
  unsigned poffc[WIN_IT];
#endif
#pragma unroll
  for (int i = 0; i < WIN_IT; ++i) {
    const int e = (i * 8 + wave) * 64 + lane;
    const int hh = e / (JPH * JPW);
    const int rem = e - hh * (JPH * JPW);
    const int rr = rem / JPW;
    const int cc = rem - rr * JPW;
    prc[i] = (rr << 8) | cc;
    poff[i] = (unsigned)((rr * w + cc) * 32 + hh * 16);
#ifdef EAVSR_IL2_EXP_CONTIG      // timing only (results wrong), interior tiles only (so that every address stays inside the image):
    poffc[i] = (unsigned)(rr * w * 32 + hh * 768 + cc * 16);      // the same bytes of the window row as two contiguous 768-byte runs
#endif
  }
  // the address of the zero unit once, kept in scalar registers: rematerialised at its uses it is a scalar load (and an
  // `lgkmcnt(0)` that also drains the LDS reads in flight) at every window request of a border tile
  const char* zero_src = reinterpret_cast<const char*>(g_il2_zero);
  asm volatile("" : "+s"(zero_src));
  auto issue_win = [&](int i, const char* xorg, int y0, int x0, bool interior, int wslot) __attribute__((always_inline)) {
    const int p = i * 8 + wave;  // wave-uniform
    if (p < JWIN_SEGS) {
      char* dst = smem + JL_WIN + wslot * JWIN_B + p * 1024;
#ifdef EAVSR_IL2_EXP_CONTIG
      const char* src = xorg + (interior ? poffc[i] : poff[i]);
#else
      const char* src = xorg + poff[i];
#endif
      if (!interior) {
        const int ylo = y0 - JPY0, xlo = x0 - JPX0;
        const bool ok = (unsigned)(ylo + (prc[i] >> 8)) < (unsigned)h && (unsigned)(xlo + (prc[i] & 255)) < (unsigned)w;
        src = ok ? src : zero_src;     // that IS the sampler's zero padding
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
  };
  // weights of pair slab `ws`: set "Ib" = k-steps 0,1,2,3 (24 pieces, i = 0..2) and k-step 8 (6 pieces, i = 3, waves 0..5)
  // -> regions A, U3[q], U8[q];  set "Ia" = k-steps 4..7 (24 pieces, i = 0..2) -> region B2
  auto issue_wgt_ib = [&](int i, const char* ws, int q) __attribute__((always_inline)) {
    if (i < 3) {
      const int wp = i * 8 + wave;
      const int dsto = JL_A + wp * 1024 + (wp >= 18 ? q * JKS_B : 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + (unsigned)(wp * 1024 + lane * 16)), (lptr_t)(smem + dsto), 16, 0, 0);
    } else if (wave < 6) {
      const int dsto = JL_U8 + q * JKS_B + wave * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + (unsigned)(8 * JKS_B + wave * 1024 + lane * 16)), (lptr_t)(smem + dsto), 16, 0,
                                       0);
    }
  };
  auto issue_wgt_ia = [&](int i, const char* ws) __attribute__((always_inline)) {
    const int wp = i * 8 + wave;
    __builtin_amdgcn_global_load_lds((gptr_t)(ws + (unsigned)(4 * JKS_B + wp * 1024 + lane * 16)),
                                     (lptr_t)(smem + JL_B2 + wp * 1024), 16, 0, 0);
  };

  // ---- contexts: the pair step being contracted ("cc") and the one after it ("nn": set up, gathered, loaded ahead) -------
  struct Ctx {
    int bn, y0, x0, P;       // image, tile origin, pair index inside the tile (groups 2P, 2P+1)
    bool ok;                 // this lane's pixel exists
    bool inter;              // the whole LDS window lies inside the image (wave-uniform)
    float fgy, fgx;          // its row (wave-uniform) and column as floats
    unsigned po;             // its byte offset inside a plane (0 when the pixel does not exist)
    const char* xw;          // window origin of group 0 in the IL8 image (may point before the image)
    const char* ws;          // this pair's weight slab
    const char* hb;          // heads mode: the image's head planes;  explicit mode: its offset planes
    const char* mb;          // explicit mode: its mask planes
  };
  auto tile_ctx = [&](Ctx& c) __attribute__((always_inline)) {      // after bn / y0 / x0 changed; P = 0
    const int gy = c.y0 + wave, gx = c.x0 + l31;
    c.ok = gy < h && gx < w;
    c.po = c.ok ? (unsigned)(gy * w + gx) * 4u : 0u;
    c.fgy = (float)gy;
    c.fgx = (float)gx;
    c.inter = c.y0 - JPY0 >= 0 && c.y0 - JPY0 + JPH <= h && c.x0 - JPX0 >= 0 && c.x0 - JPX0 + JPW <= w;
    c.xw = reinterpret_cast<const char*>(a.xil + (size_t)c.bn * ngroups * plane * JG) + ((long)(c.y0 - JPY0) * w + (c.x0 - JPX0)) * 32;
    c.ws = reinterpret_cast<const char*>(a.wpair + (size_t)cot * npairs * JPAIR_U4);
    c.hb = reinterpret_cast<const char*>(a.offset) + (size_t)c.bn * (HEADS ? 15 : 18) * D * pl4;
    c.mb = HEADS ? nullptr : reinterpret_cast<const char*>(a.mask) + (size_t)c.bn * 9 * D * pl4;
    c.P = 0;
  };
  // the plane stride as a value the optimizer cannot see through: every parameter address below is then computed where it is
  // used (a few scalar instructions) instead of being hoisted out of the loop as one more long-lived scalar register pair --
  // the kernel has ~100 of those and every spilled one costs VECTOR instructions (v_readlane / v_writelane)
  auto pl_ = [&]() __attribute__((always_inline)) {
    unsigned v = upl4;
    asm volatile("" : "+s"(v));
    return v;
  };
  Ctx cc, nn;
  int ti_ = 0;
  tile_of(0, nn.bn, nn.y0, nn.x0);
  tile_ctx(nn);
  const int total = my_tiles * npairs;
  const size_t grp_b = plane * (JG * 4);      // bytes of one group's IL8 image

  // sampling parameters, reloaded in place for the next pair step right after their last use
  float pa[JU], pb[JU], pm[JU];   // explicit: dy, dx, mask of the lane's tap of k-step u;  heads: pm = mask logit
  float tfA[6], tfB[6];           // heads: 2x2 transform + translation of the even / odd group
  // per-lane byte offsets of the parameter loads of the NEXT pair step: own tap (kg = 1: one tap further) ...
  unsigned n_pm, n_po2, n_pmX, n_poX;
  auto lane_voffs = [&]() __attribute__((always_inline)) {
    n_pm = nn.po + (kg ? upl4 : 0u);
    n_po2 = nn.po + (kg ? 2u * upl4 : 0u);
    // ... and of k-step 4 (kg = 0: tap 8 of the even group; kg = 1: tap 0 of the odd group): in one deformable group the
    // base is tap 0 and the kg = 0 lanes go 8 taps up, in two consecutive ones the base is tap 8 and kg = 1 goes one up
    n_pmX = same_dg ? nn.po + (kg ? 0u : 8u * upl4) : n_pm;
    n_poX = same_dg ? nn.po + (kg ? 0u : 16u * upl4) : n_po2;
  };
  lane_voffs();
  // wave-uniform bases of the next pair step's parameter planes (recomputed per use from few values: scalar work is cheap,
  // long-lived scalar registers are not)
  auto load_m = [&](int u) __attribute__((always_inline)) {
    const int seq0 = 2 * u, par0 = seq0 / 9, tap0 = seq0 % 9;         // the kg = 0 lanes' (group parity, tap)
    const int dgi = (2 * nn.P + par0) >> a.opg_shift;
    const int tapb = (u == 4 && same_dg) ? 0 : tap0;                    // see lane_voffs
    if (HEADS) {
      pm[u] = ld_b(nn.hb + (unsigned)(6 * D + dgi * 9 + tapb) * pl_(), u == 4 ? n_pmX : n_pm);
    } else {
      pa[u] = ld_b(nn.hb + (unsigned)(dgi * 18 + 2 * tapb) * pl_(), u == 4 ? n_poX : n_po2);
      pb[u] = ld_b(nn.hb + (unsigned)(dgi * 18 + 2 * tapb + 1) * pl_(), u == 4 ? n_poX : n_po2);
      pm[u] = ld_b(nn.mb + (unsigned)(dgi * 9 + tapb) * pl_(), u == 4 ? n_pmX : n_pm);
    }
  };
  auto load_tf = [&](float (&tf)[6], int par) __attribute__((always_inline)) {
    if (HEADS) {
      const int dgi = (2 * nn.P + par) >> a.opg_shift;
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[j] = ld_b(nn.hb + (unsigned)(dgi * 4 + j) * pl_(), nn.po);
#pragma unroll
      for (int j = 0; j < 2; ++j) tf[4 + j] = ld_b(nn.hb + (unsigned)(4 * D + dgi * 2 + j) * pl_(), nn.po);
    }
  };
  constexpr int M_LOADS = HEADS ? 1 : 3;      // vector-memory instructions of one load_m
  constexpr int TF_LOADS = HEADS ? 6 : 0;

  // Static priority for ONE half of the workgroup (the two waves of a SIMD are w and w + 4): the preferred half runs ahead
  // and the pair settles into a stagger, one wave's MFMAs beside the other's vector work (MI355X_MICROARCH.md, two waves per
  // SIMD, items 4 and 9).  Measured here (2 x 64 x 180 x 320, tools/visits/r4_d.sh): priority on waves 0-3 75-76 us, on waves
  // 4-7 (what dcnv2_il.hip does) 84-88 us, none 83-85 us; levels 1 and 3 alike.
#ifndef EAVSR_IL2_PRIO_LEVEL
#define EAVSR_IL2_PRIO_LEVEL 1
#endif
#if defined(EAVSR_IL2_PRIO_HIGH_HALF)
  if (wave >= 4) __builtin_amdgcn_s_setprio(EAVSR_IL2_PRIO_LEVEL);
#elif !defined(EAVSR_IL2_NO_PRIO)
  if (wave < 4) __builtin_amdgcn_s_setprio(EAVSR_IL2_PRIO_LEVEL);
#endif

  // ---- pipeline state ------------------------------------------------------------------------------------------------
  struct Pos {
    float w1, w2, w3, w4;   // bilinear corner weights x mask
    unsigned ad;            // byte address of the top-left corner's low channel half in LDS
    float py, px;           // the sampling position (read again only when the sample leaves the LDS window)
  };
  Pos pos[3];
  unsigned long long slowm[3] = {0, 0, 0};   // lanes whose sample of that k-step leaves the LDS window: served from global memory
  f32x4 gat[4];             // one channel half (4 channels) of the four corners: TL, TR, BL, BR
  u32x4 aop[6];             // [term * 2 + mt]
  u32x4 bop[3][3];
  f32x16 acc[2];

  const unsigned wA = (unsigned)lane * 16u;
  unsigned wpar = wA;                 // + (pair-step parity) * JKS_B
  int sE = 0, sO = 1, sEn = 0;        // window slots: even / odd group of the current pair step, even group of the next

  auto lds_f4 = [&](unsigned byte_addr) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(smem + byte_addr);
  };
  auto lds_u4 = [&](unsigned byte_addr) __attribute__((always_inline)) {
    return *reinterpret_cast<const u32x4*>(smem + byte_addr);
  };

  // Set-up of the lane's (pixel, tap) of k-step u (0..8) of context c (window slots wslE / wslO).  The window is zero outside
  // the image, which IS the sampler's corner-wise zero padding and its validity gate, so the fast path needs no image-bounds
  // test; lanes whose corners leave the WINDOW are recorded (one scalar mask per k-step) and served from global memory by
  // the gather (rare: the window has 6 rows / 8 columns of margin around the tile).
  auto setup = [&](int u, const Ctx& c, int wslE, int wslO) __attribute__((always_inline)) {
    const int t0 = (2 * u) % 9, t1 = (2 * u + 1) % 9;
    const float ry0 = (float)(t0 / 3 - 1), rx0 = (float)(t0 % 3 - 1), ry1 = (float)(t1 / 3 - 1), rx1 = (float)(t1 % 3 - 1);
    const float ryk = ry0 == ry1 ? ry0 : (kg ? ry1 : ry0);
    const float rxk = kg ? rx1 : rx0;
    float t[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) t[j] = !HEADS ? 0.f : (u < 4 ? tfA[j] : (u > 4 ? tfB[j] : (kg ? tfB[j] : tfA[j])));
    float py, px;
    j_position<HEADS>(t, pa[u], pb[u], ryk, rxk, c.fgy, c.fgx, py, px);
    const float m = HEADS == 1 ? j_sigmoid(pm[u]) : pm[u];
    const float fy0 = floorf(py), fx0 = floorf(px);
    const float lh = py - fy0, lw = px - fx0;
    const float hh = 1.f - lh, hw = 1.f - lw;
    // v_cvt_i32_f32 saturates (and maps NaN to 0): wild offsets stay defined and simply fail the window test
    const int ry = (int)fy0 - (c.y0 - JPY0), rx = (int)fx0 - (c.x0 - JPX0);
    const bool in_win = (unsigned)ry <= (unsigned)(JPH - 2) && (unsigned)rx <= (unsigned)(JPW - 2);
    const float mf = c.ok ? m : 0.f;
    const float hm = hh * mf, lm = lh * mf;
    Pos& ps = pos[u % 3];
    ps.w1 = hm * hw; ps.w2 = hm * lw; ps.w3 = lm * hw; ps.w4 = lm * lw;
    ps.py = py; ps.px = px;
    const int wsl = u < 4 ? wslE : (u > 4 ? wslO : (kg ? wslO : wslE));
    const unsigned base = (unsigned)(JL_WIN + wsl * JWIN_B);
    const unsigned ad = __umul24((unsigned)ry, (unsigned)JROW_B) + (((unsigned)rx << 4) + base);
    ps.ad = in_win ? ad : (unsigned)JL_WIN;
    slowm[u % 3] = __builtin_amdgcn_ballot_w64(c.ok && !in_win);
  };
  // the four corners (channel half `half`) of k-step u's sample: from the LDS window; lanes outside it read the image itself
  // with corner-wise zero padding (their corner weights are gated once, with the low half)
  auto gather = [&](int u, int half, const Ctx& c) __attribute__((always_inline)) {
#ifdef EAVSR_IL2_EXP_NO_GATHER
#pragma unroll
    for (int j = 0; j < 4; ++j) gat[j] = f32x4{pos[u % 3].w1, pos[u % 3].w2, (float)(pos[u % 3].ad + half), pos[u % 3].w4};
    return;
#endif
    Pos& ps = pos[u % 3];
    if (half == 0) {
      gat[0] = lds_f4(ps.ad);
      gat[1] = lds_f4(ps.ad + 16);
      gat[2] = lds_f4(ps.ad + JROW_B);
      gat[3] = lds_f4(ps.ad + JROW_B + 16);
    } else {
      gat[0] = lds_f4(ps.ad + JPLANE_B);
      gat[1] = lds_f4(ps.ad + JPLANE_B + 16);
      gat[2] = lds_f4(ps.ad + JPLANE_B + JROW_B);
      gat[3] = lds_f4(ps.ad + JPLANE_B + JROW_B + 16);
    }
#ifndef EAVSR_IL2_EXP_NO_FIXUP
    const unsigned long long sm = slowm[u % 3];
    if (__builtin_expect(sm != 0, 0)) {      // wave-uniform, rare: the block is laid out away from the pipeline
      const bool mine = (sm >> lane) & 1ull;
      f32x4 tq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) tq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (mine) {
        const float py = ps.py, px = ps.px;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
        const int hh_i = hl + 1, wh_i = wl + 1;
        const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
        if (half == 0) {      // a corner outside the image contributes zero: gate its weight (wild positions gate all four)
          const bool pos_ok = py > -1.f && px > -1.f && py < (float)h && px < (float)w;
          ps.w1 = (pos_ok && t_ok && l_ok) ? ps.w1 : 0.f;
          ps.w2 = (pos_ok && t_ok && r_ok) ? ps.w2 : 0.f;
          ps.w3 = (pos_ok && b_ok && l_ok) ? ps.w3 : 0.f;
          ps.w4 = (pos_ok && b_ok && r_ok) ? ps.w4 : 0.f;
        }
        const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
        const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
        const int par = u < 4 ? 0 : (u > 4 ? 1 : kg);       // the lane's group of the pair
        const char* xg = reinterpret_cast<const char*>(a.xil + ((size_t)c.bn * ngroups + 2 * c.P + par) * plane * JG) + half * 16;
        tq[0] = *reinterpret_cast<const f32x4*>(xg + (unsigned)(cy0 * w + cx0) * 32u);
        tq[1] = *reinterpret_cast<const f32x4*>(xg + (unsigned)(cy0 * w + cx1) * 32u);
        tq[2] = *reinterpret_cast<const f32x4*>(xg + (unsigned)(cy1 * w + cx0) * 32u);
        tq[3] = *reinterpret_cast<const f32x4*>(xg + (unsigned)(cy1 * w + cx1) * 32u);
      }
      // the loads complete HERE, inside the rare branch: past the join nothing waits on vector memory (window DMA, tile
      // stores and parameter loads stay in flight on the common path)
      asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) gat[j][e] = mine ? tq[j][e] : gat[j][e];
    }
#endif
  };
  // A operands of k-step u, output-channel tile mt
  auto load_a = [&](int u, int mt) __attribute__((always_inline)) {
    const unsigned base = u < 3 ? wA + (unsigned)(JL_A + u * JKS_B)
                        : u == 3 ? wpar + (unsigned)JL_U3
                        : u < 8 ? wA + (unsigned)(JL_B2 + (u - 4) * JKS_B)
                                : wpar + (unsigned)JL_U8;
#pragma unroll
    for (int term = 0; term < 3; ++term) aop[term * 2 + mt] = lds_u4(base + (unsigned)((term * 2 + mt) * 1024));
  };
  // blend + split of channels 2c, 2c+1 of k-step u
  auto blend_pair = [&](int u, int c) __attribute__((always_inline)) {
    const Pos& ps = pos[u % 3];
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = (2 * c + q) & 3;      // channel within the half that `gat` holds (c < 2: low half, else high half)
      float tv = ps.w1 * gat[0][e];
      tv = __builtin_fmaf(ps.w2, gat[1][e], tv);
      tv = __builtin_fmaf(ps.w3, gat[2][e], tv);
      tv = __builtin_fmaf(ps.w4, gat[3][e], tv);
      v[q] = tv;
    }
    unsigned h2, m2, l2;
#ifdef EAVSR_IL2_EXP_NO_SPLIT
    h2 = __float_as_uint(v[0]); m2 = __float_as_uint(v[1]); l2 = h2 ^ m2;
#else
    j_split2(v[0], v[1], h2, m2, l2);
#endif
    bop[u % 3][0][c] = h2; bop[u % 3][1][c] = m2; bop[u % 3][2][c] = l2;
  };
  // partial product i of a k-step (smallest first within each output-channel tile); a = [hi0 hi1 mid0 mid1 lo0 lo1]
  auto mfma_i = [&](int i, const u32x4 (&av)[6], const u32x4 (&b)[3], f32x16 (&ac)[2]) __attribute__((always_inline)) {
    constexpr int TA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, TB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
    constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
    const int mt = i / NPROD, j = i % NPROD;
    const int ta = NPROD == 9 ? TA9[j] : TA6[j], tb = NPROD == 9 ? TB9[j] : TB6[j];
#ifdef EAVSR_IL2_EXP_NO_MFMA
    ac[mt][i & 15] += __uint_as_float(b[tb][i & 3] ^ av[ta * 2 + mt][i & 3]);
#else
    ac[mt] = j_mfma(av[ta * 2 + mt], b[tb], ac[mt]);
#endif
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 b4 = lds_f4((unsigned)(JL_BIAS + kg * 128 + (m * 16 + e4 * 4) * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m][e4 * 4 + e] = b4[e];
      }
  };
  // finished tile: lane (n, kg) holds channels m * 32 + (e & 3) + 8 (e >> 2) + 4 kg of its pixel; the channel stride walks on
  // the scalar side (one wave-uniform base per store), the lane part of the address is one register
  bool st_pending = false;
  int st_bn = 0;
  unsigned st_po = 0;
  bool st_ok = false;
  auto store_tile = [&]() __attribute__((always_inline)) {
#ifdef EAVSR_IL2_EXP_NO_STORE
    if (st_ok && never) {
#else
    if (st_ok) {
#endif
      char* ob = reinterpret_cast<char*>(a.out + ((size_t)st_bn * a.cout + (size_t)cot * 64) * plane);
      const unsigned voff = st_po + (kg ? 4u * upl4 : 0u);
      if (__builtin_expect(cot * 64 + 64 <= a.cout, 1)) {      // wave-uniform
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
#ifdef EAVSR_IL2_NT_STORE
            __builtin_nontemporal_store(acc[m][e], reinterpret_cast<float*>(ob + voff));
#else
            *reinterpret_cast<float*>(ob + voff) = acc[m][e];
#endif
            ob += ((e & 3) == 3 ? 5u : 1u) * upl4;      // channel cu + 4 kg: 1 plane on, 5 planes across a group of four
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
            if (cot * 64 + cu + 4 * kg < a.cout) *reinterpret_cast<float*>(ob + voff) = acc[m][e];
            ob += ((e & 3) == 3 ? 5u : 1u) * upl4;
          }
      }
    }
    init_acc();
  };
  constexpr int ST_STORES = 32;

#define J_FENCE() __builtin_amdgcn_sched_barrier(0)
  // all DMA of the "Ia" set of context c: odd group's window (slot wsl) + k-steps 4..7 of its weights
  auto issue_ia = [&](const Ctx& c, int wsl, int part) __attribute__((always_inline)) {
    const char* xo = c.xw + grp_b * (size_t)(2 * c.P + 1);
    if (part == 0) { issue_win(0, xo, c.y0, c.x0, c.inter, wsl); issue_win(1, xo, c.y0, c.x0, c.inter, wsl); issue_wgt_ia(0, c.ws); }
    if (part == 1) { issue_win(2, xo, c.y0, c.x0, c.inter, wsl); issue_win(3, xo, c.y0, c.x0, c.inter, wsl); issue_wgt_ia(1, c.ws); }
    if (part == 2) { issue_wgt_ia(2, c.ws); }
  };

  // ---- run prologue: everything a previous pair step would have requested for the first one ------------------------------
  {
    const char* xe = nn.xw;
#pragma unroll
    for (int i = 0; i < WIN_IT; ++i) issue_win(i, xe, nn.y0, nn.x0, nn.inter, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_wgt_ib(i, nn.ws, 0);
    issue_ia(nn, 1, 0);
    issue_ia(nn, 1, 1);
    issue_ia(nn, 1, 2);
    load_tf(tfA, 0);
    load_tf(tfB, 1);
#pragma unroll
    for (int u = 0; u < JU; ++u) load_m(u);
    if (tid < 64) {
      // [kg][m * 16 + e] = bias of channel m * 32 + (e & 3) + 8 (e >> 2) + 4 kg
      const int kk = tid >> 5, idx = tid & 31, m = idx >> 4, e = idx & 15;
      const int co = cot * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
      reinterpret_cast<float*>(smem + JL_BIAS)[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();
    init_acc();
    setup(0, nn, 0, 1);
    gather(0, 0, nn);
    load_a(0, 0);
    setup(1, nn, 0, 1);
    blend_pair(0, 0);
    blend_pair(0, 1);
    J_FENCE();
    gather(0, 1, nn);
    J_FENCE();
    blend_pair(0, 2);
    blend_pair(0, 3);
    J_FENCE();
  }
  // advance: the next pair step becomes current; the one after it is next (pointers move by increments)
  auto advance = [&](int it_next) __attribute__((always_inline)) {
    cc = nn;
    if (it_next + 1 < total) {
      if (nn.P + 1 == npairs) {
        ++ti_;
        tile_of(ti_, nn.bn, nn.y0, nn.x0);
        tile_ctx(nn);
        lane_voffs();
      } else {
        ++nn.P;
        nn.ws += (size_t)JPAIR_U4 * 16;
      }
    }
  };
  advance(0);

#ifdef EAVSR_IL2_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  bool stored_now = false;      // this pair step began with a tile store: its stores are still in flight at the first barrier
  for (int it = 0; it < total; ++it) {
    J_STAMP(0);      // loop bookkeeping (advance, slot rotation)
    const int more_n = total - 1 - it;      // > 0: there is a next pair step (an integer: a bool carried across the body is copied through vector registers)
    const int q = it & 1;
    wpar = wA + (unsigned)(q * JKS_B);
    // window slots rotate by two per pair step: even group of the next pair step = slot after this pair's odd group
    sEn = sO == 2 ? 0 : sO + 1;
    const int sOn = sEn == 2 ? 0 : sEn + 1;
    J_FENCE();

#pragma unroll
    for (int u = 0; u < JU; ++u) {
      // ---- top of k-step u ---------------------------------------------------------------------------------------------
      if (u == 3 || u == 8) {
        // u = 3: the odd group's window + k-steps 4..7's weights have landed (requested in k-step 8 of the previous pair
        // step);  u = 8: the next even group's window + the next pair step's k-steps 0..3 and 8 (requested in k-steps 3..6).
        // Counted waits: whatever was issued after those requests (parameter loads; a finished tile's 32 stores) stays in
        // flight across the barrier.  The barrier publishes everybody's DMA share and retires what was read before.
        J_STAMP(u == 3 ? 2 : 5);                 // k-steps 0..2 / 3..7
#ifndef EAVSR_IL2_EXP_NO_BARRIER
        if (u == 3) {
          if (stored_now) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * M_LOADS + ST_STORES) & 15) | (((2 * M_LOADS + ST_STORES) >> 4) << 14));
          else __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * M_LOADS));
        } else {
          __builtin_amdgcn_s_waitcnt(0x0F70 | (M_LOADS + TF_LOADS));
        }
        J_STAMP(u == 3 ? 3 : 6);                 // wait for the own DMA share
        __builtin_amdgcn_s_barrier();
        J_STAMP(u == 3 ? 4 : 7);                 // wait for the other waves
#endif
      }
      if (u == 0) {
        stored_now = st_pending;
        if (st_pending) {      // wave-uniform: first k-step of a new tile
          store_tile();
          st_pending = false;
          J_STAMP(1);
        }
      }
      const int u1 = (u + 1) % JU, u2 = (u + 2) % JU;          // k-steps being blended / set up under this one's MFMAs
      const bool nx1 = u + 1 >= JU, nx2 = u + 2 >= JU;         // ... they belong to the next pair step
      if (nx1) gather(u1, 0, nn); else gather(u1, 0, cc);
      load_a(u, 1);
      J_FENCE();
      constexpr int NM = 2 * NPROD, CH = NM / 6;      // MFMAs per k-step; per chunk (2 for x6, 3 for x9)
#ifdef EAVSR_IL2_EXP_NO_DMA
      const bool dma_on = never;
#else
      const bool dma_on = true;
#endif
#ifdef EAVSR_IL2_EXP_NO_PARAMS
      const bool par_on = never;
#else
      const bool par_on = true;
#endif
#ifdef EAVSR_IL2_EXP_NO_BLEND
      const bool blend_on = never;
#else
      const bool blend_on = true;
#endif
#pragma unroll
      for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int i = 0; i < CH; ++i) mfma_i(k * CH + i, aop, bop[u % 3], acc);
#ifdef EAVSR_IL2_EXP_NO_SETUP
        if (k == 0 && never) {
#else
        if (k == 0) {
#endif
          if (nx2) setup(u2, nn, sEn, sOn);
          else setup(u2, cc, sE, sO);
        }
        if (k == 1) {
          // parameters of the next pair step, each reloaded after its last use in this one (before this k-step's DMA: the
          // counted waits above rely on the order)
          if (par_on) {
            if (u >= 1 && u <= 7) load_m(u - 1);
            if (u == 8) { load_m(7); load_m(8); }
            if (u == 3) load_tf(tfA, 0);
            if (u == 7) load_tf(tfB, 1);
          }
          // DMA: the next pair step's odd-group window + weights 4..7 all in k-step 8 (right behind the barrier that retired
          // their slots: 4 k-steps to land), its even-group window + weights 0..3, 8 over k-steps 3..6
          if (u == 8 && more_n > 0 && dma_on) issue_ia(nn, sOn, 0);
          if (u >= 3 && u <= 6 && more_n > 0 && dma_on) issue_win(u - 3, nn.xw + grp_b * (size_t)(2 * nn.P), nn.y0, nn.x0, nn.inter, sEn);
        }
        if (k == 2 && blend_on) blend_pair(u1, 0);
        if (k == 3) {
          if (blend_on) blend_pair(u1, 1);
          if (nx1) gather(u1, 1, nn); else gather(u1, 1, cc);
          load_a(u1, 0);
        }
        if (k == 4) {
          if (u == 8 && more_n > 0 && dma_on) issue_ia(nn, sOn, 1);
          if (u >= 3 && u <= 6 && more_n > 0 && dma_on) issue_wgt_ib(u - 3, nn.ws, q ^ 1);
        }
        if (k == 5) {
          if (blend_on) {
            blend_pair(u1, 2);
            blend_pair(u1, 3);
          }
          if (u == 8 && more_n > 0 && dma_on) issue_ia(nn, sOn, 2);
        }
        (void)nx1;
        J_FENCE();
      }
    }
    J_STAMP(8);      // k-step 8

    // ---- end of a pair step -----------------------------------------------------------------------------------------------
    if (cc.P + 1 == npairs) {      // the tile is complete: its accumulators leave at the top of the next k-step 0
      st_pending = true;
      st_bn = cc.bn; st_po = cc.po; st_ok = cc.ok;
    }
    sE = sEn;
    sO = sOn;
    advance(it + 1);
  }
  if (st_pending) store_tile();
#undef J_FENCE
#ifdef EAVSR_IL2_STAMPS
  J_STAMP(10);
  if (lane == 0 && (wave == 0 || wave == 4)) {
    for (int i = 0; i < 12; ++i) atomicAdd(&g_il2_stamps[(wave == 4 ? 16 : 0) + i], st_acc[i]);
  }
#endif
}

// weight (cout, cin, 3, 3) fp32 -> [cot][pair][u][term][mt][lane] 16-byte elements: lane (m = lane & 31, kgrp = lane >> 5)
// holds row co = 64 cot + 32 mt + m, k = the 8 channels of (group 2 pair + seq / 9, tap seq % 9), seq = 2 u + kgrp
__global__ void pack_il2_kernel(const float* __restrict__ wt, unsigned* __restrict__ out, int cout, int cin, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;   // one (16-byte element, pair j) per thread
  if (e >= total) return;
  const int j = (int)(e & 3);
  long u_ = e >> 2;
  const int lane = (int)(u_ % 64); u_ /= 64;
  const int mt = (int)(u_ % 2); u_ /= 2;
  const int term = (int)(u_ % 3); u_ /= 3;
  const int u = (int)(u_ % JU); u_ /= JU;
  const int npairs = cin / (2 * JG);
  const int pair = (int)(u_ % npairs);
  const int cot = (int)(u_ / npairs);
  const int co = cot * 64 + mt * 32 + (lane & 31);
  const int seq = 2 * u + (lane >> 5);
  const int grp = 2 * pair + seq / 9, tap = seq % 9;
  float v0 = 0.f, v1 = 0.f;
  if (co < cout) {
    v0 = wt[((size_t)co * cin + grp * JG + 2 * j) * 9 + tap];
    v1 = wt[((size_t)co * cin + grp * JG + 2 * j + 1) * 9 + tap];
  }
  unsigned hi, mid, lo;
  j_split2(v0, v1, hi, mid, lo);
  out[e] = term == 0 ? hi : (term == 1 ? mid : lo);
}

template <int NPROD, int HEADS>
int launch_il2(const IL2Args& a, dim3 grid, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_il2_kernel<NPROD, HEADS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)JLDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2_il2: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((dcnv2_il2_kernel<NPROD, HEADS>), grid, dim3(512), JLDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2_il2");
}

}  // namespace

#ifdef EAVSR_IL2_STAMPS
extern "C" int eavsr_debug_il2_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_il2_stamps), sizeof(g_il2_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_il2_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int64_t eavsr_dcn_weight_il2_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % (2 * JG) != 0) return 0;
  return (int64_t)eavsr::cdiv(cout, 64) * (cin / (2 * JG)) * JPAIR_U4 * 16;
}

extern "C" int eavsr_pack_dcn_weight_il2(const float* weight, void* packed, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_dcn_weight_il2: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % (2 * JG) == 0, -2, "pack_dcn_weight_il2: cin %d must be a multiple of 16", cin);
  const long total = eavsr_dcn_weight_il2_bytes(cout, cin) / 4;
  hipLaunchKernelGGL(pack_il2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight,
                     reinterpret_cast<unsigned*>(packed), cout, cin, total);
  return eavsr::launch_status("pack_dcn_weight_il2");
}

extern "C" int eavsr_dcnv2_il2_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_il2,
                                   const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                   int32_t deform_groups, int32_t nprod, int32_t heads, void* stream) {
  EAVSR_REQUIRE(x_il8 && offset_or_heads && weight_il2 && out && (heads || mask), -1, "dcnv2_il2: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2_il2: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2_il2: cin %d not divisible by deform_groups %d", cin, deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2, "dcnv2_il2: %d channels per deformable group unsupported (must be a multiple of 8)", cpg);
  EAVSR_REQUIRE(cin % 16 == 0, -2, "dcnv2_il2: cin %d must be a multiple of 16 (groups are contracted in pairs)", cin);
  EAVSR_REQUIRE(nprod == 6 || nprod == 9, -2, "dcnv2_il2: nprod %d (6 or 9)", nprod);
  EAVSR_REQUIRE((long)h * w * 64 < (1L << 32), -1, "dcnv2_il2: plane too large for 32-bit byte offsets");
  EAVSR_REQUIRE((long)h * w * 15 * deform_groups * 4 < (1L << 32) || !heads, -1, "dcnv2_il2: heads tensor too large");
  EAVSR_REQUIRE((long)h * w * 18 * 4 < (1L << 32), -1, "dcnv2_il2: offset planes too large");
  EAVSR_REQUIRE((((uintptr_t)x_il8) & 15) == 0, -2, "dcnv2_il2: x must be 16-byte aligned");
  if (n == 0) return 0;
  IL2Args a;
  a.xil = x_il8; a.offset = offset_or_heads; a.mask = mask; a.wpair = reinterpret_cast<const u32x4*>(weight_il2);
  a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups;
  {
    const int opg = cpg / 8;
    EAVSR_REQUIRE((opg & (opg - 1)) == 0, -2, "dcnv2_il2: %d channels per deformable group: cpg / 8 must be a power of two", cpg);
    a.opg_shift = 0;
    while ((1 << a.opg_shift) < opg) ++a.opg_shift;
  }
  a.tiles_x = eavsr::cdiv(w, JT_W);
  a.tiles_y = eavsr::cdiv(h, JT_ROWS);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "dcnv2_il2: too many tiles");
  a.ntiles = (int)tiles;
  int cus = 256;
  {
    static int cu_cache[eavsr::kMaxDevices] = {};
    const int dev = eavsr::current_device();
    if (cu_cache[dev] == 0) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_cache[dev] = v;
      else cu_cache[dev] = 256;
    }
    cus = cu_cache[dev];
  }
  dim3 grid((unsigned)(tiles < cus ? tiles : cus), eavsr::cdiv(cout, 64));
  hipStream_t st = eavsr::as_stream(stream);
  EAVSR_REQUIRE(heads >= 0 && heads <= 2, -1, "dcnv2_il2: heads %d (0, 1 or 2)", heads);
  if (nprod == 9) return heads == 2 ? launch_il2<9, 2>(a, grid, st) : heads ? launch_il2<9, 1>(a, grid, st) : launch_il2<9, 0>(a, grid, st);
  return heads == 2 ? launch_il2<6, 2>(a, grid, st) : heads ? launch_il2<6, 1>(a, grid, st) : launch_il2<6, 0>(a, grid, st);
}
