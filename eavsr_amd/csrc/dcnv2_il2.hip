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

// Every global access of the pipeline is a BUFFER instruction (round 5): a 128-bit resource (base + extent, four scalar
// registers, rebuilt once per tile) + one per-lane 32-bit offset register that lives as long as the tile + one scalar offset
// computed where it is used (one s_mul_i32 / s_add_i32).  The round-4 form (global_load / global_store / global_load_lds
// with 64-bit addresses) cost 4 scalar + 1-2 vector instructions per access for address arithmetic, a branch + two
// compares + two selects per window request of a border tile (zero padding) and an exec-mask region per store; here the
// hardware's range check does both jobs: a lane whose offset register holds J_OOB reads zeros (also into LDS) and its
// store is dropped.  The loop is bound by instruction ISSUE (one instruction per ~4 cycles and SIMD over both resident
// waves, scalar and branch instructions included: DESIGN.md 4b-r5), so every instruction removed is time.
typedef __amdgpu_buffer_rsrc_t j_rsrc;
constexpr unsigned J_OOB = 0x80000000u;      // beyond every extent the launcher accepts (< 2^31 bytes)
__device__ __forceinline__ j_rsrc j_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// scalar offset = c * s where c is a compile-time constant: computed AT the use by one s_mul_i32 the optimizer cannot hoist
// (hoisted, the ~60 distinct products of a pair step become long-lived scalar registers, spill, and a spilled scalar costs
// VECTOR instructions: v_readlane / v_writelane)
__device__ __forceinline__ unsigned j_smul(unsigned s, int c) {      // c: a constant once the caller is inlined / unrolled
  unsigned r;
  asm volatile("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(s), "i"(c));
  return r;
}
__device__ __forceinline__ float j_ld(j_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 j_ld4(j_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void j_st(j_rsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void j_dma16(j_rsrc r, unsigned voff, unsigned soff, char* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds_dst, 16, voff, soff, 0, 0);
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
  // Per lane and piece: (row, column) inside the window and the unit's byte offset from the window origin; per TILE (below)
  // the unit's byte offset inside a group's IL8 image, or J_OOB where the unit lies outside the image: the buffer range check
  // then writes zeros into LDS, which IS the sampler's corner-wise zero padding and its validity gate.
  constexpr int WIN_IT = (JWIN_SEGS + 7) / 8;   // 4
  int prc[WIN_IT];
  unsigned poff[WIN_IT];
  unsigned winvo[WIN_IT];
#pragma unroll
  for (int i = 0; i < WIN_IT; ++i) {
    const int e = (i * 8 + wave) * 64 + lane;
    const int hh = e / (JPH * JPW);
    const int rem = e - hh * (JPH * JPW);
    const int rr = rem / JPW;
    const int cc = rem - rr * JPW;
    prc[i] = (rr << 8) | cc;
    poff[i] = (unsigned)((rr * w + cc) * 32 + hh * 16);
  }
  const unsigned grp_b = uplane * (unsigned)(JG * 4);      // bytes of one group's IL8 image
  const unsigned wA = (unsigned)lane * 16u;
  const unsigned wv1k = (unsigned)wave * 1024u;
  const int jks_w2 = wave >= 2 ? JKS_B : 0;

  // buffer resources: the weight slab of this output-channel tile (fixed); of the NEXT pair step's image: its IL8 groups, its
  // heads (or offset) planes, its mask planes (explicit mode)
  const j_rsrc r_w = j_make_rsrc(a.wpair + (size_t)cot * npairs * JPAIR_U4, (unsigned)npairs * (unsigned)(JPAIR_U4 * 16));
  j_rsrc r_x, r_h, r_m;
  j_rsrc r_xc;      // the IL8 groups of the CURRENT pair step's image (the rare path of the gather reads it)

  // ---- contexts: the pair step being contracted ("cc") and the one after it ("nn": set up, gathered, loaded ahead) -------
  struct Ctx {
    int bn, y0, x0, P;       // image, tile origin, pair index inside the tile (groups 2P, 2P+1)
    int wy0, wx0;            // origin of the LDS window in the image: y0 - 6, x0 - 8
    bool ok;                 // this lane's pixel exists
    unsigned long long okm;  // ... as a lane mask
    float fgy, fgx;          // its row (wave-uniform) and column as floats
    unsigned po;             // its byte offset inside a plane (0 when the pixel does not exist)
  };
  Ctx cc, nn;
  unsigned n_wso = 0;        // byte offset of nn's pair slab inside r_w
  unsigned n_xso = 0;        // byte offset of nn's even group inside r_x
  // scalar byte offsets (inside r_h / r_m) of the parameter planes of nn's even / odd group
  unsigned n_sm[2] = {0, 0}, n_st[2] = {0, 0}, n_sr[2] = {0, 0};
  auto pair_ctx = [&]() __attribute__((always_inline)) {      // after nn.P changed
    n_wso = (unsigned)nn.P * (unsigned)(JPAIR_U4 * 16);
    n_xso = (unsigned)(2 * nn.P) * grp_b;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const unsigned dgi = (unsigned)((2 * nn.P + par) >> a.opg_shift);
      if (HEADS) {
        n_sm[par] = (6u * D + dgi * 9u) * upl4;      // mask logits / masks
        n_st[par] = (dgi * 4u) * upl4;               // 2x2 transform
        n_sr[par] = (4u * D + dgi * 2u) * upl4;      // translation
      } else {
        n_sm[par] = (dgi * 9u) * upl4;               // masks (r_m)
        n_st[par] = (dgi * 18u) * upl4;              // offsets (r_h)
      }
    }
  };
  auto tile_ctx = [&](Ctx& c) __attribute__((always_inline)) {      // after bn / y0 / x0 changed; P = 0
    const int gy = c.y0 + wave, gx = c.x0 + l31;
    c.ok = gy < h && gx < w;
    c.okm = __builtin_amdgcn_ballot_w64(c.ok);
    c.wy0 = c.y0 - JPY0;
    c.wx0 = c.x0 - JPX0;
    c.po = c.ok ? (unsigned)(gy * w + gx) * 4u : 0u;
    c.fgy = (float)gy;
    c.fgx = (float)gx;
    c.P = 0;
    r_x = j_make_rsrc(a.xil + (size_t)c.bn * ngroups * plane * JG, (unsigned)ngroups * grp_b);
    r_h = j_make_rsrc(a.offset + (size_t)c.bn * (HEADS ? 15 : 18) * D * plane, (unsigned)((HEADS ? 15 : 18) * D) * upl4);
    if (!HEADS) r_m = j_make_rsrc(a.mask + (size_t)c.bn * 9 * D * plane, (unsigned)(9 * D) * upl4);
    const int ylo = c.wy0, xlo = c.wx0;
    const unsigned torg = (unsigned)((ylo * w + xlo) * 32);      // wraps for windows that begin before the image: so does the sum
#pragma unroll
    for (int i = 0; i < WIN_IT; ++i) {
      const bool ok = (unsigned)(ylo + (prc[i] >> 8)) < (unsigned)h && (unsigned)(xlo + (prc[i] & 255)) < (unsigned)w;
      winvo[i] = ok ? poff[i] + torg : J_OOB;
    }
  };
  auto issue_win = [&](int i, unsigned xso, int wslot) __attribute__((always_inline)) {
    if (i < 3 || wave < JWIN_SEGS - 24)      // piece i * 8 + wave < 30 (wave-uniform)
      j_dma16(r_x, winvo[i], xso, smem + wslot + i * 8192 + wv1k);      // wslot: the slot's byte address in LDS
  };
  // weights of the pair slab at `wso`: set "Ib" = k-steps 0,1,2,3 (24 pieces, i = 0..2) and k-step 8 (6 pieces, i = 3, waves
  // 0..5) -> regions A, U3[q], U8[q];  set "Ia" = k-steps 4..7 (24 pieces, i = 0..2) -> region B2
  auto issue_wgt_ib = [&](int i, unsigned wso, int q) __attribute__((always_inline)) {
    if (i < 3) {
      // pieces 18.. (i = 2, waves 2..7) are k-step 3: parity slot q
      const int dsto = JL_A + i * 8192 + (i == 2 ? q * jks_w2 : 0);
      j_dma16(r_w, wA, wso + (unsigned)(i * 8192) + wv1k, smem + dsto + wv1k);
    } else if (wave < 6) {
      j_dma16(r_w, wA, wso + (unsigned)(8 * JKS_B) + wv1k, smem + JL_U8 + q * JKS_B + wv1k);
    }
  };
  auto issue_wgt_ia = [&](int i, unsigned wso) __attribute__((always_inline)) {
    j_dma16(r_w, wA, wso + (unsigned)(4 * JKS_B + i * 8192) + wv1k, smem + JL_B2 + i * 8192 + wv1k);
  };

  int ti_ = 0;
  tile_of(0, nn.bn, nn.y0, nn.x0);
  tile_ctx(nn);
  pair_ctx();
  const int total = my_tiles * npairs;

  // sampling parameters, reloaded in place for the next pair step right after their last use
  float pa[JU], pb[JU], pm[JU];   // explicit: dy, dx, mask of the lane's tap of k-step u;  heads: pm = mask logit
  float tfA[6], tfB[6];           // heads: 2x2 transform + translation of the even / odd group
  // per-lane byte offsets of the parameter loads of the NEXT pair step: own tap (kg = 1: one tap further) ...
  unsigned n_pm, n_po2, n_pmX, n_poX;
  unsigned n_poj[4];      // heads: the pixel's offset in plane j of a group's transform / translation block (no scalar arithmetic per load)
  auto lane_voffs = [&]() __attribute__((always_inline)) {
    // a lane without a pixel must contribute nothing: its MASK offsets hold J_OOB, the range check makes the value zero
    // (HEADS == 1 takes logits: there the set-up gates the activated value instead)
    const unsigned pom = (HEADS == 1 || nn.ok) ? nn.po : J_OOB;
    n_pm = pom + (kg ? upl4 : 0u);
    n_po2 = nn.po + (kg ? 2u * upl4 : 0u);
    // ... and of k-step 4 (kg = 0: tap 8 of the even group; kg = 1: tap 0 of the odd group): in one deformable group the
    // base is tap 0 and the kg = 0 lanes go 8 taps up, in two consecutive ones the base is tap 8 and kg = 1 goes one up
    n_pmX = same_dg ? pom + (kg ? 0u : 8u * upl4) : n_pm;
    if (HEADS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) n_poj[j] = nn.po + (unsigned)j * upl4;
    }
    n_poX = same_dg ? nn.po + (kg ? 0u : 16u * upl4) : n_po2;
  };
  lane_voffs();
  auto load_m = [&](int u) __attribute__((always_inline)) {
    const int seq0 = 2 * u, par0 = seq0 / 9, tap0 = seq0 % 9;         // the kg = 0 lanes' (group parity, tap)
    // u == 4: in one deformable group the base is tap 0 (see lane_voffs), else tap 8: a run-time choice of two scalars.
    // (Every operand below is read BY VALUE first: a ternary of two captured variables is a select of two addresses, which
    // keeps both in scratch memory.)
    const unsigned vm = u == 4 ? 0u + n_pmX : 0u + n_pm, vo = u == 4 ? 0u + n_poX : 0u + n_po2;
    const unsigned sm0 = n_sm[0], sm1 = n_sm[1], so0 = n_st[0], so1 = n_st[1];
    const unsigned x8 = u == 4 ? (same_dg ? 0u : j_smul(upl4, 8)) : 0u;
    if (HEADS) {
      const unsigned so = u == 4 ? sm0 + x8 : (par0 ? sm1 : sm0) + (tap0 ? j_smul(upl4, tap0) : 0u);
      pm[u] = j_ld(r_h, vm, so);
    } else {
      const unsigned so2 = u == 4 ? so0 + 2u * x8 : (par0 ? so1 : so0) + (tap0 ? j_smul(upl4, 2 * tap0) : 0u);
      const unsigned sm = u == 4 ? sm0 + x8 : (par0 ? sm1 : sm0) + (tap0 ? j_smul(upl4, tap0) : 0u);
      pa[u] = j_ld(r_h, vo, so2);
      pb[u] = j_ld(r_h, vo, so2 + upl4);
      pm[u] = j_ld(r_m, vm, sm);
    }
  };
  auto load_tf = [&](float (&tf)[6], int par) __attribute__((always_inline)) {
    if (HEADS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[j] = j_ld(r_h, n_poj[j], par ? 0u + n_st[1] : 0u + n_st[0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) tf[4 + j] = j_ld(r_h, n_poj[j], par ? 0u + n_sr[1] : 0u + n_sr[0]);
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

  unsigned wpar = wA;                 // + (pair-step parity) * JKS_B
  unsigned dummy_ad = (unsigned)JL_WIN;     // where lanes outside the window read (any valid address)
  asm volatile("" : "+v"(dummy_ad));
  // window slots (as LDS byte addresses: no multiplication where they are used): even / odd group of the current pair step,
  // even group of the next
  constexpr int S0 = JL_WIN, S1 = JL_WIN + JWIN_B, S2 = JL_WIN + 2 * JWIN_B;
  int sE = S0, sO = S1, sEn = S0;

  // LDS by absolute byte address: this kernel has no static LDS, its dynamic allocation begins at 0 (through `smem + x` the
  // address of a gather costs one more vector instruction, an addition of the zero base the optimizer no longer folds)
  auto lds_f4 = [&](unsigned byte_addr) __attribute__((always_inline)) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(byte_addr);
  };
  auto lds_u4 = [&](unsigned byte_addr) __attribute__((always_inline)) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(byte_addr);
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
    const int ry = (int)fy0 - c.wy0, rx = (int)fx0 - c.wx0;
    // lanes whose corners leave the window, as a lane mask straight from the two compares (a ballot of the bool costs two more
    // vector instructions)
    const unsigned long long outm = __builtin_amdgcn_uicmp((unsigned)ry, (unsigned)(JPH - 2), 34 /* ugt */) |
                                    __builtin_amdgcn_uicmp((unsigned)rx, (unsigned)(JPW - 2), 34);
    const float mf = HEADS == 1 ? (c.ok ? m : 0.f) : m;      // HEADS != 1: zero already (lane_voffs)
    const float hm = hh * mf, lm = lh * mf;
    Pos& ps = pos[u % 3];
    ps.w1 = hm * hw; ps.w2 = hm * lw; ps.w3 = lm * hw; ps.w4 = lm * lw;
    ps.py = py; ps.px = px;
    const int wsl = u < 4 ? wslE : (u > 4 ? wslO : (kg ? wslO : wslE));
    const unsigned base = (unsigned)wsl;
    const unsigned ad = __umul24((unsigned)ry, (unsigned)JROW_B) + (((unsigned)rx << 4) + base);
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(ps.ad) : "v"(ad), "v"(dummy_ad), "s"(outm));
    slowm[u % 3] = outm & c.okm;
  };
  // the four corners (channel half `half`) of k-step u's sample: from the LDS window; lanes outside it read the image itself
  // with corner-wise zero padding (their corner weights are gated once, with the low half)
  auto gather = [&](int u, int half, const Ctx& c, j_rsrc rxc) __attribute__((always_inline)) {
#ifdef EAVSR_IL2_EXP_NO_GATHER
#pragma unroll
    for (int j = 0; j < 4; ++j) gat[j] = f32x4{pos[u % 3].w1, pos[u % 3].w2, (float)(pos[u % 3].ad + half), pos[u % 3].w4};
    return;
#endif
    Pos& ps = pos[u % 3];
    // requested LAST corner first: the blend's first instruction needs the first corner, which is then the youngest read, and
    // the one wait in front of it covers all four (requested in blend order, each corner's first use gets a wait of its own:
    // six more s_waitcnt per k-step in a loop whose bound is instruction issue)
    const unsigned hb_ = half ? (unsigned)JPLANE_B : 0u;
#ifdef EAVSR_IL2_GATHER_FWD
    gat[0] = lds_f4(ps.ad + hb_);
    gat[1] = lds_f4(ps.ad + hb_ + 16);
    gat[2] = lds_f4(ps.ad + hb_ + JROW_B);
    gat[3] = lds_f4(ps.ad + hb_ + JROW_B + 16);
#else
    gat[3] = lds_f4(ps.ad + hb_ + JROW_B + 16);
    gat[2] = lds_f4(ps.ad + hb_ + JROW_B);
    gat[1] = lds_f4(ps.ad + hb_ + 16);
    gat[0] = lds_f4(ps.ad + hb_);
#endif
#ifndef EAVSR_IL2_EXP_NO_FIXUP
    unsigned long long sm = slowm[u % 3];
    // (tested afresh for either half: a boolean carried from the first test to the second costs three more scalar
    // instructions per k-step than a second s_cmp_lg_u64)
#ifndef EAVSR_IL2_SLOW_BOOL
    asm volatile("" : "+s"(sm));
#endif
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
        // the lane's group of the pair: 2 P (+ 1 for the odd group: the kg = 1 lanes of k-step 4, every lane past it)
        const unsigned so = (unsigned)(2 * c.P + (u > 4 ? 1 : 0)) * grp_b;
        const unsigned vb = (u == 4 && kg ? grp_b : 0u) + (unsigned)(half * 16);
        tq[0] = j_ld4(rxc, vb + (unsigned)(cy0 * w + cx0) * 32u, so);
        tq[1] = j_ld4(rxc, vb + (unsigned)(cy0 * w + cx1) * 32u, so);
        tq[2] = j_ld4(rxc, vb + (unsigned)(cy1 * w + cx0) * 32u, so);
        tq[3] = j_ld4(rxc, vb + (unsigned)(cy1 * w + cx1) * 32u, so);
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
  // finished tile: lane (n, kg) holds channels m * 32 + (e & 3) + 8 (e >> 2) + 4 kg of its pixel.  One buffer store per
  // value: the resource covers this tile's (up to) 64 output planes, the channel is a scalar offset (one s_mul_i32), the lane
  // part one register for the whole tile; a lane without a pixel holds J_OOB there and its stores are dropped by the range
  // check (no exec-mask region per store)
  bool st_pending = false;
  int st_bn = 0;
  unsigned st_vo = J_OOB;
  auto store_tile = [&]() __attribute__((always_inline)) {
    const int nco = min(64, a.cout - cot * 64);
    const j_rsrc r_o = j_make_rsrc(a.out + ((size_t)st_bn * a.cout + (size_t)cot * 64) * plane, (unsigned)nco * upl4);
#ifdef EAVSR_IL2_EXP_NO_STORE
    if (never) {
#else
    {
#endif
      if (__builtin_expect(nco == 64, 1)) {      // wave-uniform
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
            j_st(r_o, st_vo, cu ? j_smul(upl4, cu) : 0u, acc[m][e]);
          }
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
            j_st(r_o, cu + 4 * kg < nco ? st_vo : J_OOB, cu ? j_smul(upl4, cu) : 0u, acc[m][e]);
          }
      }
    }
    init_acc();
  };
  constexpr int ST_STORES = 32;

#define J_FENCE() __builtin_amdgcn_sched_barrier(0)
  // all DMA of the "Ia" set of the next pair step: odd group's window (slot wsl) + k-steps 4..7 of its weights
  auto issue_ia = [&](int wsl, int part) __attribute__((always_inline)) {
    const unsigned g = n_xso + grp_b;
    if (part == 0) { issue_win(0, g, wsl); issue_win(1, g, wsl); issue_wgt_ia(0, n_wso); }
    if (part == 1) { issue_win(2, g, wsl); issue_win(3, g, wsl); issue_wgt_ia(1, n_wso); }
    if (part == 2) { issue_wgt_ia(2, n_wso); }
  };

  // ---- run prologue: everything a previous pair step would have requested for the first one ------------------------------
  {
#pragma unroll
    for (int i = 0; i < WIN_IT; ++i) issue_win(i, n_xso, S0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_wgt_ib(i, n_wso, 0);
    load_tf(tfA, 0);
    load_tf(tfB, 1);
#pragma unroll
    for (int u = 0; u < JU; ++u) load_m(u);
    float bias_v = 0.f;
    if (tid < 64) {
      // [kg][m * 16 + e] = bias of channel m * 32 + (e & 3) + 8 (e >> 2) + 4 kg
      const int kk = tid >> 5, idx = tid & 31, m = idx >> 4, e = idx & 15;
      const int co = cot * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
      bias_v = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    // the "Ia" set (odd group's window, weights of k-steps 4..7) LAST: k-steps 0..2 do not read it, so it stays in flight across
    // the first barrier (6 or 7 requests per wave: vmcnt(6) covers both) and is waited for at k-step 3 like in every other pair
    // step -- the run starts ~1/3 of a prologue's DMA earlier
    J_FENCE();
    issue_ia(S1, 0);
    issue_ia(S1, 1);
    issue_ia(S1, 2);
    J_FENCE();
    __builtin_amdgcn_s_waitcnt(0x0F70 | 6);      // vmcnt(6)
    if (tid < 64) reinterpret_cast<float*>(smem + JL_BIAS)[tid] = bias_v;
    __syncthreads();
#ifdef EAVSR_IL2_STAGGER
    if (wave >= 4) __builtin_amdgcn_s_sleep(EAVSR_IL2_STAGGER);
#endif
    init_acc();
    setup(0, nn, S0, S1);
    gather(0, 0, nn, r_x);
    load_a(0, 0);
    setup(1, nn, S0, S1);
    blend_pair(0, 0);
    blend_pair(0, 1);
    J_FENCE();
    gather(0, 1, nn, r_x);
    J_FENCE();
    blend_pair(0, 2);
    blend_pair(0, 3);
    J_FENCE();
  }
  // advance: the next pair step becomes current; the one after it is next.  Past the end of the run `nn` stays where it is:
  // the last pair step then requests (and never reads) its own operands once more into the slots that are free by rotation --
  // the requests of a pair step are unconditional (no branch per request), and the run ends with a vmcnt(0).
  auto advance = [&](int it_next) __attribute__((always_inline)) {
    cc = nn;
    r_xc = r_x;
    if (it_next + 1 < total) {
      if (nn.P + 1 == npairs) {
        ++ti_;
        tile_of(ti_, nn.bn, nn.y0, nn.x0);
        tile_ctx(nn);
        lane_voffs();
      } else {
        ++nn.P;
      }
      pair_ctx();
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
    const int q = it & 1;
    wpar = wA + (unsigned)(q * JKS_B);
    // window slots rotate by two per pair step: even group of the next pair step = slot after this pair's odd group
    sEn = sO == S2 ? S0 : sO + JWIN_B;
    const int sOn = sEn == S2 ? S0 : sEn + JWIN_B;
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
#ifdef EAVSR_IL2_STAGGER
        // A/B: the two waves of a SIMD (w, w + 4) leave every meeting EAVSR_IL2_STAGGER x 64 cycles apart, so that one's MFMA
        // chunks fall beside the other's vector chunks instead of on top of them (MI355X_MICROARCH.md, two waves per SIMD, item 9)
        if (wave >= 4) __builtin_amdgcn_s_sleep(EAVSR_IL2_STAGGER);
#endif
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
      if (nx1) gather(u1, 0, nn, r_x); else gather(u1, 0, cc, r_xc);
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
          if (u == 8 && dma_on) issue_ia(sOn, 0);
          if (u >= 3 && u <= 6 && dma_on) issue_win(u - 3, n_xso, sEn);
        }
        if (k == 2 && blend_on) blend_pair(u1, 0);
        if (k == 3) {
          if (blend_on) blend_pair(u1, 1);
          if (nx1) gather(u1, 1, nn, r_x); else gather(u1, 1, cc, r_xc);
          load_a(u1, 0);
        }
        if (k == 4) {
          if (u == 8 && dma_on) issue_ia(sOn, 1);
          if (u >= 3 && u <= 6 && dma_on) issue_wgt_ib(u - 3, n_wso, q ^ 1);
        }
        if (k == 5) {
          if (blend_on) {
            blend_pair(u1, 2);
            blend_pair(u1, 3);
          }
          if (u == 8 && dma_on) issue_ia(sOn, 2);
        }
        (void)nx1;
        J_FENCE();
      }
    }
    J_STAMP(8);      // k-step 8

    // ---- end of a pair step -----------------------------------------------------------------------------------------------
    if (cc.P + 1 == npairs) {      // the tile is complete: its accumulators leave at the top of the next k-step 0
      st_pending = true;
      st_bn = cc.bn;
      st_vo = cc.ok ? cc.po + (kg ? 4u * upl4 : 0u) : J_OOB;
    }
    sE = sEn;
    sO = sOn;
    advance(it + 1);
  }
  if (st_pending) store_tile();
  // the last pair step's requests (see advance) must have landed before this workgroup's LDS is handed to another one
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
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
  // every per-image extent is a buffer resource addressed with 32-bit offsets, and J_OOB (2^31) must lie beyond each of them
  EAVSR_REQUIRE((long)h * w * 4 * cin < (1L << 31), -1, "dcnv2_il2: one image of x exceeds 2 GiB");
  EAVSR_REQUIRE((long)h * w * 4 * (heads ? 15 : 18) * deform_groups < (1L << 31), -1, "dcnv2_il2: one image of the heads / offsets exceeds 2 GiB");
  EAVSR_REQUIRE((long)h * w * 4 * 64 < (1L << 31), -1, "dcnv2_il2: 64 output planes exceed 2 GiB");
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
