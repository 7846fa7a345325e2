// DCNv2 forward with the fp32 contraction carried by the bf16 matrix pipe ("bf16x9": every fp32 operand is
// split EXACTLY into three bf16 terms and all nine partial products are accumulated in fp32).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the packed-fp32 vector rate and blocks the SIMD's vector issue while it
// runs (dcnv2.hip, "What bounds it"): in the native-fp32 kernel the 576-deep contraction alone costs 2304 SIMD
// cycles per 32 pixels x 64 channels x 36 k, on top of the sampler's vector work.  v_mfma_f32_32x32x16_bf16 does
// 16 k in 32 cycles and holds the vector issue for only 8 of them, so nine of them per 16 k cost 288 matrix-pipe
// cycles (vs 512) and leave the vector ALUs to the sampler.
//
// Arithmetic.  x = hi + mid + lo with hi = trunc_bf16(x), mid = trunc_bf16(x - hi), lo = x - hi - mid: both
// subtractions are exact in fp32 and lo has at most 8 significant bits, so the three bf16 terms reproduce the
// 24-bit fp32 value exactly (sub-normal tails of |x| < 2^-102 aside).  A product of two bf16 numbers is exact in
// fp32; the nine partial products of a * b therefore sum to the exact 48-bit product and the only rounding is
// the fp32 accumulation -- as in an fp32 FMA chain.  No operand is ever rounded to bf16.
// tests/test_hip_ops.py compares this kernel and the native one against an fp64 evaluation of the same sums.
//
// Structure.  With 16 k per MFMA the B operand of lane (n = lane & 31, kgrp = lane >> 5) is 8 consecutive k of
// pixel n: here k = 8 channels of ONE tap, tap = 2 s + kgrp in k-step s.  Each lane therefore samples its own
// pixel for its own tap and feeds the values to the MFMA straight from registers:
//   * no column tile in LDS, no barrier between sampling and contraction; a wave is self-contained
//   * the sampling position of a (pixel, tap) is computed by exactly one lane, once per deformable group
//   * one barrier per deformable group (8 per tile): it publishes the next 8-channel LDS window and the next
//     pre-split weight slab, both moved by 16-byte LDS-DMA while the current group computes
// Workgroup = 512 threads = 8 rows x 32 pixels, one per CU (123 KB LDS: 2 window stages + 2 weight stages).
//
// Reference semantics: mmcv.ops.modulated_deform_conv2d as called at models/networks.py:627-630 (see dcnv2.hip).
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ld_b(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __attribute__((aligned(16))) float g_x9_zero[4];

constexpr int XT_ROWS = 8, XT_W = 32;                 // pixel tile: one 32-pixel row per wave
constexpr int XG = 8;                                 // channels per k-step group (one tap x 8 channels = 8 k)
constexpr int XK = 9, XSTEPS = 5;                     // taps; k-steps per group (taps 2s, 2s+1; tap 9 is zero)
constexpr int XPH = XT_ROWS + 12, XPW = 48;           // LDS window rows y0-6 .. y0+13, columns x0-8 .. x0+39
constexpr int XPY0 = 6, XPX0 = 8;
constexpr int XPATCH_F = XG * XPH * XPW;              // 7680 floats = 30 one-KiB DMA pieces
constexpr int XPATCH_SEGS = XPATCH_F / 256;
constexpr int XW_U4 = XSTEPS * 3 * 2 * 64;            // 16-byte elements of one group's weight slab (CO = 64)
constexpr int XW_SEGS = XW_U4 / 64;                   // 30 one-KiB pieces
constexpr int XNPIECE = XPATCH_SEGS + XW_SEGS;        // 60
constexpr int XP_IT = (XNPIECE + 7) / 8;              // pieces per wave
constexpr size_t XLDS_BYTES = 2 * (size_t)XPATCH_F * 4 + 2 * (size_t)XW_U4 * 16;   // 122,880

struct X9Args {
  const float* x;
  const float* offset;
  const float* mask;
  const u32x4* wsplit;   // [cot][group][step][plane][mt][lane] 16-byte elements
  const float* bias;
  float* out;
  int n, cin, h, w, cout, dg, cpg, tiles_x, tiles_y;
};

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  // v_perm_b32: bytes 2,3 of the first value into the low half, bytes 2,3 of the second into the high half
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

#if EAVSR_LAB      // the round-1 NCHW kernel with all nine partial products: lab build only; the weight packing below is shared with eavsr_dcnv2_il_f32
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void dcnv2_x9_kernel(X9Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_patch = smem;                                                   // [2][XPATCH_F]
  u32x4* s_w = reinterpret_cast<u32x4*>(smem + 2 * XPATCH_F);              // [2][XW_U4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kgrp = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * XT_ROWS, x0 = tx * XT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const int ngroups = a.cin / XG;

  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // this lane's pixel and, per k-step, its tap
  const int gy = y0 + wave, gx = x0 + l31;
  const bool pix_ok = gy < h && gx < w;
  const unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;
  float fyb[XSTEPS], fxb[XSTEPS];   // un-displaced sampling position of the lane's tap in k-step s
#pragma unroll
  for (int s = 0; s < XSTEPS; ++s) {
    const int tap = min(2 * s + kgrp, XK - 1);
    const int ti = tap / 3, tj = tap - 3 * ti;
    fyb[s] = (float)(gy - 1 + ti);
    fxb[s] = (float)(gx - 1 + tj);
  }

  {
    f32x4* z = reinterpret_cast<f32x4*>(s_patch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * XPATCH_F / 4; e += 512) z[e] = zero;
  }
  // DMA piece q = i * 8 + wave of a group: q < XPATCH_SEGS is a window piece, the next XW_SEGS the weight slab
  unsigned voff[XP_IT];
#pragma unroll
  for (int i = 0; i < XP_IT; ++i) {
    const int q = i * 8 + wave;
    voff[i] = 0xFFFFFFFFu;
    if (q < XPATCH_SEGS) {
      const int e4 = q * 64 + lane;
      const int ci = e4 / (XPH * (XPW / 4));
      const int rem = e4 - ci * (XPH * (XPW / 4));
      const int r = rem / (XPW / 4), c4 = rem - r * (XPW / 4);
      const int yy = y0 - XPY0 + r, xx = x0 - XPX0 + 4 * c4;
      const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
      if (ok) voff[i] = (unsigned)(((size_t)ci * plane + (size_t)yy * w + xx) * 4);
    }
  }
  auto issue = [&](int g, int stage) __attribute__((always_inline)) {
    const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)bn * a.cin + (size_t)g * XG) * plane);
    const char* wsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)cot * ngroups + g) * XW_U4);
#pragma unroll
    for (int i = 0; i < XP_IT; ++i) {
      const int q = i * 8 + wave;  // wave-uniform
      if (q < XPATCH_SEGS) {
        const char* src = voff[i] != 0xFFFFFFFFu ? xb + voff[i] : reinterpret_cast<const char*>(g_x9_zero);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_patch + stage * XPATCH_F + q * 256), 16, 0, 0);
      } else if (q < XNPIECE) {
        const int seg = q - XPATCH_SEGS;
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(seg * 64 + lane) * 16u),
                                         (lptr_t)(s_w + stage * XW_U4 + seg * 64), 16, 0, 0);
      }
    }
  };
  // offsets / mask of the lane's tap in k-step s (dead once the step's position is set up: the next group's
  // values are loaded straight into them)
  float oy[XSTEPS], ox[XSTEPS], mk[XSTEPS];
  auto load_offset = [&](int g, int s) __attribute__((always_inline)) {
    const int dgi = g * XG / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + dgi) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + dgi) * 9 * plane;
    const unsigned tap = (unsigned)min(2 * s + kgrp, XK - 1);
#ifdef EAVSR_X9_EXP_NO_OFFSETS   // timing ablation only (tools/gpu_x9_ablate.py): results are wrong
    oy[s] = 0.25f; ox[s] = 0.25f; mk[s] = 0.5f;
    (void)offb; (void)mkb; (void)tap;
#else
    oy[s] = ld_b(offb, (2u * tap * uplane + pix) * 4u);
    ox[s] = ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
    mk[s] = ld_b(mkb, (tap * uplane + pix) * 4u);
#endif
  };

  // sampling state of one (pixel, tap): bilinear corner weights x mask and the LDS corner address
  struct Pos {
    f32x2 w12, w34;
    const float* q;
    bool slow;
  };
  auto setup = [&](int s, Pos& ps, const float* pst) __attribute__((always_inline)) {
    const bool tap_ok = (s < XSTEPS - 1) || kgrp == 0;   // tap 9 does not exist
    const float py = fyb[s] + oy[s];
    const float px = fxb[s] + ox[s];
    const bool in = pix_ok && tap_ok && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
    const float fy0 = floorf(py), fx0 = floorf(px);
    const float lh = py - fy0, lw = px - fx0;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
    const float m = in ? mk[s] : 0.f;
    const int ry = hl - (y0 - XPY0), rx = wl - (x0 - XPX0);
    const bool in_win = ry >= 0 && ry <= XPH - 2 && rx >= 0 && rx <= XPW - 2;
    const bool fast = in && in_win;
    const float mf = fast ? m : 0.f;        // lanes whose corners left the window contribute zero here
    const float hm = hh * mf, lm = lh * mf;
    ps.w12 = f32x2{hm * hw, hm * lw};
    ps.w34 = f32x2{lm * hw, lm * lw};
    ps.q = pst + (fast ? ry * XPW + rx : 0);
    ps.slow = in && !in_win;
  };
  auto gather = [&](const Pos& ps, float (&gt)[XG][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < XG; ++c) {
      gt[c][0] = ps.q[c * (XPH * XPW)];
      gt[c][1] = ps.q[c * (XPH * XPW) + 1];
      gt[c][2] = ps.q[c * (XPH * XPW) + XPW];
      gt[c][3] = ps.q[c * (XPH * XPW) + XPW + 1];
    }
  };

  constexpr int WAIT_VM0 = 0x0F70;
  __syncthreads();  // zero fill done before the first DMA may land
  issue(0, 0);
#pragma unroll
  for (int s = 0; s < XSTEPS; ++s) load_offset(0, s);

  for (int g = 0; g < ngroups; ++g) {
    const int stage = g & 1;
    // window(g), weights(g), offsets(g) have landed; every wave is done with the other stage
    __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    __syncthreads();
    const bool more = g + 1 < ngroups;
#ifndef EAVSR_X9_EXP_NO_DMA
    if (more) issue(g + 1, stage ^ 1);
#endif
    const int gnext = more ? g + 1 : g;
    const float* pst = s_patch + stage * XPATCH_F;
    const u32x4* wst = s_w + stage * XW_U4 + lane;
    const float* xg = a.x + ((size_t)bn * a.cin + (size_t)g * XG) * plane;

    // Three-deep software pipeline over the 5 k-steps, all in ONE basic block so that the scheduler directives can
    // place the vector work in the MFMA shadows (v_mfma_f32_32x32x16_bf16 holds the vector issue for 8 of its 32
    // cycles): iteration s issues   A: position + 32 LDS gathers of step s+2
    //                               B: blend + exact split of step s+1 (B operand in registers)
    //                               C: A-operand reads + the 18 MFMAs of step s
    Pos pos[2];
    float gat[2][XG][4];
    u32x4 bop[2][3];
    unsigned slow_steps = 0;
    auto stage_a = [&](int t) __attribute__((always_inline)) {
      setup(t, pos[t & 1], pst);
      gather(pos[t & 1], gat[t & 1]);
      slow_steps |= pos[t & 1].slow ? (1u << t) : 0u;
      load_offset(gnext, t);   // the last group reloads its own values: no branch inside the scheduled block
    };
    auto stage_b = [&](int t) __attribute__((always_inline)) {
      const Pos& ps = pos[t & 1];
      float v[XG];
#pragma unroll
      for (int c = 0; c < XG; ++c) {
        const f32x2 top = {gat[t & 1][c][0], gat[t & 1][c][1]};
        const f32x2 bot = {gat[t & 1][c][2], gat[t & 1][c][3]};
        const f32x2 r = ps.w12 * top + ps.w34 * bot;
        v[c] = r.x + r.y;
      }
#pragma unroll
      for (int c = 0; c < XG / 2; ++c) {
        unsigned h2, m2, l2;
        split2(v[2 * c], v[2 * c + 1], h2, m2, l2);
        bop[t & 1][0][c] = h2; bop[t & 1][1][c] = m2; bop[t & 1][2][c] = l2;
      }
    };
    auto stage_c = [&](int t, const u32x4 (&b)[3]) __attribute__((always_inline)) {
      const u32x4* ws = wst + t * (3 * 2 * 64);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const u32x4 ah = ws[(0 * 2 + mt) * 64], am = ws[(1 * 2 + mt) * 64], al = ws[(2 * 2 + mt) * 64];
        // the nine partial products, smallest terms first
        acc[mt] = mfma_bf16(al, b[2], acc[mt]);
        acc[mt] = mfma_bf16(al, b[1], acc[mt]);
        acc[mt] = mfma_bf16(am, b[2], acc[mt]);
        acc[mt] = mfma_bf16(al, b[0], acc[mt]);
        acc[mt] = mfma_bf16(ah, b[2], acc[mt]);
        acc[mt] = mfma_bf16(am, b[1], acc[mt]);
        acc[mt] = mfma_bf16(am, b[0], acc[mt]);
        acc[mt] = mfma_bf16(ah, b[1], acc[mt]);
        acc[mt] = mfma_bf16(ah, b[0], acc[mt]);
      }
    };
    stage_a(0);
    stage_a(1);
    stage_b(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < XSTEPS; ++t) {
      if (t + 2 < XSTEPS) stage_a(t + 2);
      if (t + 1 < XSTEPS) stage_b(t + 1);
#ifndef EAVSR_X9_EXP_NO_MFMA
      stage_c(t, bop[t & 1]);
#else
      acc[0][t] += __uint_as_float(bop[t & 1][0][0] ^ bop[t & 1][1][1] ^ bop[t & 1][2][2]);
#endif
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU in its shadow
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // rare: some corner left the LDS window.  Those lanes contributed exactly zero above (their weights were
    // zeroed); their samples are redone from global memory with corner-wise zero padding and multiplied in.
    if (__builtin_amdgcn_ballot_w64(slow_steps != 0) != 0) {
      const int dgi = g * XG / a.cpg;
      const float* offb = a.offset + ((size_t)bn * a.dg + dgi) * 18 * plane;
      const float* mkb = a.mask + ((size_t)bn * a.dg + dgi) * 9 * plane;
      for (int t = 0; t < XSTEPS; ++t) {
        const bool mine = (slow_steps >> t) & 1u;
        if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
        float v[XG];
#pragma unroll
        for (int c = 0; c < XG; ++c) v[c] = 0.f;
        if (mine) {
          const unsigned tap = (unsigned)min(2 * t + kgrp, XK - 1);
          const int ti = (int)tap / 3, tj = (int)tap - 3 * ti;
          const float py = (float)(gy - 1 + ti) + ld_b(offb, (2u * tap * uplane + pix) * 4u);
          const float px = (float)(gx - 1 + tj) + ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
          const float m = ld_b(mkb, (tap * uplane + pix) * 4u);
          const float fy0 = floorf(py), fx0 = floorf(px);
          const float lh = py - fy0, lw = px - fx0;
          const float hm = (1.f - lh) * m, lm = lh * m, hw = 1.f - lw;
          const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
          const int hh_i = hl + 1, wh_i = wl + 1;
          const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
          const float w1 = (t_ok & l_ok) ? hm * hw : 0.f;
          const float w2 = (t_ok & r_ok) ? hm * lw : 0.f;
          const float w3 = (b_ok & l_ok) ? lm * hw : 0.f;
          const float w4 = (b_ok & r_ok) ? lm * lw : 0.f;
          const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
          const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
          const unsigned i1 = (unsigned)(cy0 * w + cx0) * 4u, i2 = (unsigned)(cy0 * w + cx1) * 4u;
          const unsigned i3 = (unsigned)(cy1 * w + cx0) * 4u, i4 = (unsigned)(cy1 * w + cx1) * 4u;
#pragma unroll
          for (int c = 0; c < XG; ++c) {
            const float* qc = xg + (size_t)c * plane;
            float tv = w1 * ld_b(qc, i1);
            tv += w2 * ld_b(qc, i2);
            tv += w3 * ld_b(qc, i3);
            tv += w4 * ld_b(qc, i4);
            v[c] = tv;
          }
        }
        u32x4 b[3];
#pragma unroll
        for (int c = 0; c < XG / 2; ++c) {
          unsigned h2, m2, l2;
          split2(v[2 * c], v[2 * c + 1], h2, m2, l2);
          b[0][c] = h2; b[1][c] = m2; b[2][c] = l2;
        }
        const u32x4* ws = wst + t * (3 * 2 * 64);
        for (int mt = 0; mt < 2; ++mt) {
          const u32x4 ah = ws[(0 * 2 + mt) * 64], am = ws[(1 * 2 + mt) * 64], al = ws[(2 * 2 + mt) * 64];
          f32x16 c_ = acc[mt];
          c_ = mfma_bf16(al, b[2], c_);
          c_ = mfma_bf16(al, b[1], c_);
          c_ = mfma_bf16(am, b[2], c_);
          c_ = mfma_bf16(al, b[0], c_);
          c_ = mfma_bf16(ah, b[2], c_);
          c_ = mfma_bf16(am, b[1], c_);
          c_ = mfma_bf16(am, b[0], c_);
          c_ = mfma_bf16(ah, b[1], c_);
          c_ = mfma_bf16(ah, b[0], c_);
          acc[mt] = c_;
        }
      }
    }
  }

#ifdef EAVSR_X9_EXP_NO_STORE
  if (pix_ok && acc[0][0] == 12345.678f) {
#else
  if (pix_ok) {
#endif
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
        if (co < a.cout) {
          const float b = a.bias ? a.bias[co] : 0.f;
          a.out[((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx] = acc[m][r] + b;
        }
      }
  }
}

#endif  // EAVSR_LAB
// weight (cout, cin, 3, 3) fp32 -> [cot][group][step][plane][mt][lane] 16-byte elements: lane (m = lane & 31,
// kgrp = lane >> 5) holds row co = 64 cot + 32 mt + m, k = channels 8 g .. 8 g + 7 of tap 2 s + kgrp
__global__ void pack_x9_kernel(const float* __restrict__ wt, unsigned* __restrict__ out, int cout, int cin, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;   // one (16-byte element, pair j) per thread
  if (e >= total) return;
  const int j = (int)(e & 3);
  long u = e >> 2;
  const int lane = (int)(u % 64); u /= 64;
  const int mt = (int)(u % 2); u /= 2;
  const int pl = (int)(u % 3); u /= 3;
  const int s = (int)(u % XSTEPS); u /= XSTEPS;
  const int ngroups = cin / XG;
  const int g = (int)(u % ngroups);
  const int cot = (int)(u / ngroups);
  const int co = cot * 64 + mt * 32 + (lane & 31);
  const int tap = 2 * s + (lane >> 5);
  float v0 = 0.f, v1 = 0.f;
  if (co < cout && tap < XK) {
    v0 = wt[((size_t)co * cin + g * XG + 2 * j) * XK + tap];
    v1 = wt[((size_t)co * cin + g * XG + 2 * j + 1) * XK + tap];
  }
  unsigned hi, mid, lo;
  split2(v0, v1, hi, mid, lo);
  out[e] = pl == 0 ? hi : (pl == 1 ? mid : lo);
}

}  // namespace

extern "C" int64_t eavsr_dcn_weight_x9_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % XG != 0) return 0;
  return (int64_t)eavsr::cdiv(cout, 64) * (cin / XG) * XW_U4 * 16;
}

extern "C" int eavsr_pack_dcn_weight_x9(const float* weight, void* packed, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_dcn_weight_x9: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % XG == 0, -1, "pack_dcn_weight_x9: cin %d must be a multiple of 8", cin);
  const long total = eavsr_dcn_weight_x9_bytes(cout, cin) / 4;
  hipLaunchKernelGGL(pack_x9_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight,
                     reinterpret_cast<unsigned*>(packed), cout, cin, total);
  return eavsr::launch_status("pack_dcn_weight_x9");
}

#if EAVSR_LAB
extern "C" int eavsr_dcnv2_f32x9(const float* x, const float* offset, const float* mask, const void* weight_x9,
                                 const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w,
                                 int32_t cout, int32_t deform_groups, void* stream) {
  EAVSR_REQUIRE(x && offset && mask && weight_x9 && out, -1, "dcnv2_f32x9: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2_f32x9: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2_f32x9: cin %d not divisible by deform_groups %d", cin, deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2, "dcnv2_f32x9: %d channels per deformable group unsupported (must be a multiple of 8)", cpg);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "dcnv2_f32x9: plane too large");
  EAVSR_REQUIRE((w % 4) == 0 && (((uintptr_t)x) & 15) == 0, -2,
                "dcnv2_f32x9: needs w %% 4 == 0 and a 16-byte aligned input (use eavsr_dcnv2_f32 otherwise)");
  if (n == 0) return 0;
  X9Args a;
  a.x = x; a.offset = offset; a.mask = mask; a.wsplit = reinterpret_cast<const u32x4*>(weight_x9); a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups; a.cpg = cpg;
  a.tiles_x = eavsr::cdiv(w, XT_W);
  a.tiles_y = eavsr::cdiv(h, XT_ROWS);
  const long blocks = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "dcnv2_f32x9: too many tiles");
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_x9_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)XLDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2_f32x9: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  dim3 grid((unsigned)blocks, eavsr::cdiv(cout, 64));
  hipLaunchKernelGGL(dcnv2_x9_kernel, grid, dim3(512), XLDS_BYTES, eavsr::as_stream(stream), a);
  return eavsr::launch_status("dcnv2_f32x9");
}
#endif  // EAVSR_LAB
