// DCNv2 forward, wave-specialised form of dcnv2_il.hip (same arithmetic, same IL8 input, same weight packing).
//
// Why.  dcnv2_il_kernel is instruction-ISSUE bound: its two waves per SIMD run the same program -- sampler
// vector work, MFMAs, scalar address arithmetic, LDS reads -- and each wave issues at most one instruction per ~4 cycles, so
// the step costs (instructions of both waves) x 4 cycles while the matrix pipe idles 75 % and the vector ALU 60 % of the time.
// Here the two waves of a SIMD have DIFFERENT jobs that use different issue ports:
//   * waves 0-3 ("samplers", one per SIMD): positions, LDS gathers, bilinear blend, exact 3-way bf16 split of TWO pixel rows
//     each -- pure vector / LDS work, no accumulators, no MFMA; the split B operands go to an LDS staging buffer
//   * waves 4-7 ("contractors", the SIMD partners): A operands + staged B operands from LDS, the MFMAs of those two rows
//     (A shared by both rows), the LDS-DMA of the next group's window / weight slab, bias, output stores -- matrix and
//     memory work with a handful of vector instructions.
// v_mfma_f32_32x32x16_bf16 holds the SIMD's vector issue for 8 of its 32 cycles, so the sampler keeps ~75 % of the vector
// port while its partner keeps the matrix pipe fed; round 1 found the opposite for the fp32 MFMA (which blocks the port).
// One workgroup barrier per k-step hands a double-buffered B stage from samplers to contractors, which lag one k-step.
//
// Reference semantics: mmcv.ops.modulated_deform_conv2d as called at models/networks.py:627-630, optionally with the affine
// -> offsets expansion and mask sigmoid of AdaptBlockOffset (networks.py:302-315) folded in ("heads" mode); see dcnv2_il.hip.
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wld(const char* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(base + byte_off);
}

constexpr int WT_ROWS = 8, WT_W = 32;                 // pixel tile: two 32-pixel rows per sampler / contractor pair
constexpr int WG = 8;                                 // channels per group = k of one tap
constexpr int WK = 9, WSTEPS = 5;                     // taps; k-steps per group (taps 2s, 2s+1; tap 9 is zero)
constexpr int WPH = WT_ROWS + 10, WPW = 44;           // LDS window rows y0-5 .. y0+12, columns x0-6 .. x0+37
constexpr int WPY0 = 5, WPX0 = 6;
constexpr int WWIN_U = WPH * WPW * 2;                 // 1584 16-byte units (two per window position)
constexpr int WWIN_SEGS = (WWIN_U + 63) / 64;         // 25 one-KiB pieces (the last one partial)
constexpr int WWIN_BYTES = WWIN_SEGS * 1024;          // 25,600 per stage
constexpr int WW_U4 = WSTEPS * 3 * 2 * 64;            // 1920 units = 30 pieces: one group's weight slab (CO = 64)
constexpr int WW_SEGS = WW_U4 / 64;
constexpr int WB_BYTES = 2 * 4 * 2 * 3 * 1024;        // B stage: [buffer][pair][row][term][lane] x 16 B = 49,152
constexpr size_t WLDS_BYTES = 2 * (size_t)WWIN_BYTES + 2 * (size_t)WW_U4 * 16 + WB_BYTES + 64 * 4;   // 162,048

struct WSArgs {
  const float* xil;      // [n][cin/8][h][w][8]
  const float* offset;   // explicit: (n, dg*18, h, w);  heads: (n, 15*dg, h, w)
  const float* mask;
  const u32x4* wsplit;   // [cot][group][step][term][mt][lane] 16-byte elements (eavsr_pack_dcn_weight_x9)
  const float* bias;
  float* out;            // (n, cout, h, w)
  int n, cin, h, w, cout, dg, opg_shift, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ void ws_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 ws_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NPROD, bool HEADS>
__global__ __launch_bounds__(512, 2) void dcnv2_ws_kernel(WSArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  char* s_win = wsm;                                                        // [2][WWIN_BYTES]
  u32x4* s_w = reinterpret_cast<u32x4*>(wsm + 2 * WWIN_BYTES);              // [2][WW_U4]
  u32x4* s_b = reinterpret_cast<u32x4*>(wsm + 2 * WWIN_BYTES + 2 * WW_U4 * 16);   // [2][4][2][3][64]
  float* s_bias = reinterpret_cast<float*>(wsm + 2 * WWIN_BYTES + 2 * WW_U4 * 16 + WB_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool sampler = wave < 4;
  const int pr = wave & 3;                       // sampler / contractor pair = rows 2 pr, 2 pr + 1 of the tile
  const int l31 = lane & 31, kg = lane >> 5;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const size_t pl4 = plane * 4;
  const int ngroups = a.cin / WG;
  const int cot = blockIdx.y;

  // persistent, XCD-aware tile walk (as dcnv2_il_kernel)
  const int nb = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd_wg = (nb + 7 - xcd) >> 3;
  const int q_ = a.ntiles >> 3, r_ = a.ntiles & 7;
  const int t_begin = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
  const int t_count = q_ + (xcd < r_ ? 1 : 0);
  const int my_tiles = slot < t_count ? (t_count - slot + per_xcd_wg - 1) / per_xcd_wg : 0;
  if (my_tiles == 0) return;
  auto tile_of = [&](int i, int& bn, int& y0, int& x0) __attribute__((always_inline)) {
    int t = t_begin + slot + i * per_xcd_wg;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    bn = t / a.tiles_y;
    y0 = ty * WT_ROWS;
    x0 = tx * WT_W;
  };
  const int total = my_tiles * ngroups;          // steps (tile, group); phases = 5 per step + one for the contractors' lag

  if (tid < 64) {
    const int co = blockIdx.y * 64 + tid;
    s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
  }

  if (!sampler) {
    // =================================================================== contractors: DMA, MFMA, stores ==========
    // DMA pieces of a step, spread over the four contractor waves: window piece p = i * 4 + pr (p < 25), weight piece likewise
    constexpr int WIN_IT = (WWIN_SEGS + 3) / 4, W_IT = (WW_SEGS + 3) / 4;      // 7, 8
    auto issue = [&](int bn, int y0, int x0, int g, int stage, int i_lo, int i_hi) __attribute__((always_inline)) {
      const char* xorg = reinterpret_cast<const char*>(a.xil + ((size_t)bn * ngroups + g) * plane * WG) +
                         ((long)(y0 - WPY0) * w + (x0 - WPX0)) * 32;
      const char* wsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)cot * ngroups + g) * WW_U4);
      const int ylo = y0 - WPY0, xlo = x0 - WPX0;
#ifdef EAVSR_WS_EXP_NO_DMA
      if (stage >= 0) return;
#endif
      for (int i = i_lo; i < i_hi; ++i) {
        if (i < WIN_IT) {
          const int p = i * 4 + pr;
          if (p < WWIN_SEGS) {
            const int e = p * 64 + lane;                 // 16-byte unit of the window image
            const int rr = e / (2 * WPW), cc = e - rr * (2 * WPW);
            const int col = cc >> 1;
            const bool inside = e < WWIN_U;
            const bool ok = inside && (unsigned)(ylo + rr) < (unsigned)h && (unsigned)(xlo + col) < (unsigned)w;
            // bank swizzle (dcnv2_il.hip): the halves of window column x are stored swapped when bit 3 of x is set
            const unsigned off = (unsigned)((rr * w + col) * 32 + (((cc & 1) ^ ((col >> 3) & 1)) * 16));
            char* dst = s_win + stage * WWIN_BYTES + p * 1024;
            if (ok) __builtin_amdgcn_global_load_lds((gptr_t)(xorg + off), (lptr_t)dst, 16, 0, 0);
            else *reinterpret_cast<u32x4*>(dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
          }
        } else {
          const int seg = (i - WIN_IT) * 4 + pr;
          if (seg < WW_SEGS)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(seg * 64 + lane) * 16u),
                                             (lptr_t)(s_w + stage * WW_U4 + seg * 64), 16, 0, 0);
        }
      }
    };
    constexpr int N_IT = WIN_IT + W_IT;                 // 15 issue slots per contractor wave and step

    int bn, y0, x0;
    tile_of(0, bn, y0, x0);
    issue(bn, y0, x0, 0, 0, 0, N_IT);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                     // bias + window / weights of step 0 visible

    f32x16 acc[2][2];      // [row][mt]
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[r][m][e] = s_bias[m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg];
    };
    init_acc();
    auto store_tile = [&](int sbn, int sy0, int sx0) __attribute__((always_inline)) {
      unsigned pl4u = uplane * 4u;
      asm volatile("" : "+s"(pl4u));
      const char* ob = reinterpret_cast<const char*>(a.out + ((size_t)sbn * a.cout + (size_t)cot * 64) * plane);
      const bool full = cot * 64 + 64 <= a.cout;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gy = sy0 + 2 * pr + r, gx = sx0 + l31;
        if (gy < h && gx < w) {
          unsigned voff = ((unsigned)(gy * w + gx)) * 4u + (kg ? 4u * pl4u : 0u);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
              float* q = reinterpret_cast<float*>(const_cast<char*>(ob) + voff);
              if (full || cot * 64 + cu + 4 * kg < a.cout) *q = acc[r][m][e];
              voff += ((e & 3) == 3 ? 5u : 1u) * pl4u;
            }
        }
      }
      init_acc();
    };

    // phase ph = step * 5 + t: samplers produce B(ph) in buffer ph & 1, contractors consume B(ph - 1)
    int g = 0, ti_ = 0;            // the step being CONSUMED (lags the samplers by one phase)
    int nbn = bn, ny0 = y0, nx0 = x0;
    for (int st = 0; st < total; ++st) {
      const int stage = st & 1;
      const bool last_g = g + 1 == ngroups;
      const bool more = st + 1 < total;
      if (last_g && more) tile_of(ti_ + 1, nbn, ny0, nx0);
      const int ng = last_g ? 0 : g + 1;
      const u32x4* wst = s_w + stage * WW_U4 + lane;
#pragma unroll
      for (int t = 0; t < WSTEPS; ++t) {
        // ---- phase (st, t) for the samplers; the contractors first finish (st, t - 1) -> B of the previous phase
        // own DMA pieces landed (only needed at the step boundary), own LDS reads / zero fills done
        if (t == 0) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        // B(st, t) will be produced during this phase; consume the one produced in the previous phase:
        const int ct = t == 0 ? WSTEPS - 1 : t - 1;          // its k-step
#ifdef EAVSR_WS_EXP_NO_CONTRACT
        if (false) {
#else
        if (!(st == 0 && t == 0)) {
#endif
          const int cstage = t == 0 ? stage ^ 1 : stage;     // k-step 4 of the previous step used the other stage's weights
          const u32x4* cw = s_w + cstage * WW_U4 + lane + ct * (3 * 2 * 64);
          const int pbuf = ((st * WSTEPS + t - 1) & 1);      // parity of the phase that produced it
          const u32x4* bs = s_b + ((pbuf * 4 + pr) * 2) * 3 * 64 + lane;
          u32x4 av[6], b0[3], b1[3];
#pragma unroll
          for (int j = 0; j < 6; ++j) av[j] = cw[j * 64];
#pragma unroll
          for (int j = 0; j < 3; ++j) { b0[j] = bs[j * 64]; b1[j] = bs[(3 + j) * 64]; }
          constexpr int TA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, TB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
          constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < NPROD; ++j) {
              const int ta = NPROD == 9 ? TA9[j] : TA6[j], tb = NPROD == 9 ? TB9[j] : TB6[j];
#ifndef EAVSR_WS_EXP_NO_MFMA
              acc[0][mt] = ws_mfma(av[ta * 2 + mt], b0[tb], acc[0][mt]);
              acc[1][mt] = ws_mfma(av[ta * 2 + mt], b1[tb], acc[1][mt]);
#else
              acc[0][mt][j] += __uint_as_float(av[ta * 2 + mt][0] ^ b0[tb][1]);
              acc[1][mt][j] += __uint_as_float(av[ta * 2 + mt][2] ^ b1[tb][3]);
#endif
            }
          // the tile that just finished (its last k-step was consumed in phase t == 0 of the next tile's first step)
          if (t == 0 && g == 0) {
            int pbn, py0, px0;
            tile_of(ti_ - 1, pbn, py0, px0);
            store_tile(pbn, py0, px0);
          }
        }
        // next step's DMA: its stage was last read in phase (st, 0) by the contractors (weights) -> issue in phases 1 .. 3
        if (more && t >= 1 && t <= 3) issue(nbn, ny0, nx0, ng, stage ^ 1, (t - 1) * 5, min(t * 5, N_IT));
      }
      if (last_g) { bn = nbn; y0 = ny0; x0 = nx0; g = 0; ++ti_; } else ++g;
    }
    // the lagging last phase: B(total - 1, 4)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    {
      const int cstage = (total - 1) & 1;
      const u32x4* cw = s_w + cstage * WW_U4 + lane + (WSTEPS - 1) * (3 * 2 * 64);
      const int pbuf = (total * WSTEPS - 1) & 1;
      const u32x4* bs = s_b + ((pbuf * 4 + pr) * 2) * 3 * 64 + lane;
      u32x4 av[6], b0[3], b1[3];
#pragma unroll
      for (int j = 0; j < 6; ++j) av[j] = cw[j * 64];
#pragma unroll
      for (int j = 0; j < 3; ++j) { b0[j] = bs[j * 64]; b1[j] = bs[(3 + j) * 64]; }
      constexpr int TA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, TB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
      constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NPROD; ++j) {
          const int ta = NPROD == 9 ? TA9[j] : TA6[j], tb = NPROD == 9 ? TB9[j] : TB6[j];
          acc[0][mt] = ws_mfma(av[ta * 2 + mt], b0[tb], acc[0][mt]);
          acc[1][mt] = ws_mfma(av[ta * 2 + mt], b1[tb], acc[1][mt]);
        }
      int pbn, py0, px0;
      tile_of(my_tiles - 1, pbn, py0, px0);
      store_tile(pbn, py0, px0);
    }
    return;
  }

  // ===================================================================== samplers: positions, gathers, blend, split =======
  // per-lane addressing of the sampling parameters: wave-uniform base (SGPRs) + 32-bit byte offset (pixel, lane half's tap parity)
  struct POff { unsigned p4, po, pm; };
  auto make_poff = [&](unsigned pix) __attribute__((always_inline)) {
    POff o;
    o.p4 = pix * 4u;
    o.po = (pix + (kg ? 2u * uplane : 0u)) * 4u;
    o.pm = (pix + (kg ? uplane : 0u)) * 4u;
    return o;
  };
  float pa[2][WSTEPS], pb[2][WSTEPS], pm[2][WSTEPS], tf[2][6];
  auto load_params = [&](int bn, int g, const POff& o, int r) __attribute__((always_inline)) {
    const int dgi = g >> a.opg_shift;
    if (HEADS) {
      const char* hb = reinterpret_cast<const char*>(a.offset) + (size_t)bn * 15 * a.dg * pl4;
      const char* mb = hb + (size_t)(6 * a.dg + dgi * 9) * pl4;
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[r][j] = wld(hb + (size_t)(dgi * 4 + j) * pl4, o.p4);
#pragma unroll
      for (int j = 0; j < 2; ++j) tf[r][4 + j] = wld(hb + (size_t)(4 * a.dg + dgi * 2 + j) * pl4, o.p4);
#pragma unroll
      for (int s = 0; s < WSTEPS; ++s) pm[r][s] = wld(mb + (size_t)(2 * s) * pl4, s == WSTEPS - 1 ? o.p4 : o.pm);
    } else {
      const char* ob = reinterpret_cast<const char*>(a.offset) + ((size_t)bn * a.dg + dgi) * 18 * pl4;
      const char* mb = reinterpret_cast<const char*>(a.mask) + ((size_t)bn * a.dg + dgi) * 9 * pl4;
#pragma unroll
      for (int s = 0; s < WSTEPS; ++s) {
        const bool last = s == WSTEPS - 1;
        pa[r][s] = wld(ob + (size_t)(4 * s) * pl4, last ? o.p4 : o.po);
        pb[r][s] = wld(ob + (size_t)(4 * s + 1) * pl4, last ? o.p4 : o.po);
        pm[r][s] = wld(mb + (size_t)(2 * s) * pl4, last ? o.p4 : o.pm);
      }
    }
  };

  int bn, y0, x0;
  tile_of(0, bn, y0, x0);
  auto pix_of = [&](int ty0, int tx0, int r, bool& ok) __attribute__((always_inline)) {
    const int gy = ty0 + 2 * pr + r, gx = tx0 + l31;
    ok = gy < h && gx < w;
    return ok ? (unsigned)(gy * w + gx) : 0u;
  };
  bool ok0, ok1;
  unsigned pix0 = pix_of(y0, x0, 0, ok0), pix1 = pix_of(y0, x0, 1, ok1);
  load_params(bn, 0, make_poff(pix0), 0);
  load_params(bn, 0, make_poff(pix1), 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();                                     // matches the contractors' prologue barrier

  int g = 0, ti_ = 0;
  for (int st = 0; st < total; ++st) {
    const int stage = st & 1;
    const bool last_g = g + 1 == ngroups;
    const bool more = st + 1 < total;
    int nbn = bn, ny0 = y0, nx0 = x0;
    if (last_g && more) tile_of(ti_ + 1, nbn, ny0, nx0);
    const int ng = last_g ? 0 : g + 1;
    bool nok0, nok1;
    const unsigned npix0 = pix_of(ny0, nx0, 0, nok0), npix1 = pix_of(ny0, nx0, 1, nok1);
    // this step's parameters to working registers; the next step's requested at once (a whole step of latency cover)
    float ca[2][WSTEPS], cb[2][WSTEPS], cm[2][WSTEPS], ctf[2][6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int s = 0; s < WSTEPS; ++s) { ca[r][s] = HEADS ? 0.f : pa[r][s]; cb[r][s] = HEADS ? 0.f : pb[r][s]; cm[r][s] = pm[r][s]; }
#pragma unroll
      for (int j = 0; j < 6; ++j) ctf[r][j] = HEADS ? tf[r][j] : 0.f;
    }
    load_params(more ? nbn : bn, more ? ng : g, make_poff(more ? npix0 : pix0), 0);
    load_params(more ? nbn : bn, more ? ng : g, make_poff(more ? npix1 : pix1), 1);

    const float* win = reinterpret_cast<const float*>(s_win + stage * WWIN_BYTES);
    const float* xg = a.xil + ((size_t)bn * ngroups + g) * plane * WG;
#pragma unroll
    for (int t = 0; t < WSTEPS; ++t) {
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the previous phase's B writes are complete (vmcnt untouched)
      __builtin_amdgcn_s_barrier();
#ifdef EAVSR_WS_EXP_NO_SAMPLE
      continue;
#endif
      const int pbuf = (st * WSTEPS + t) & 1;
      u32x4* bdst = s_b + ((pbuf * 4 + pr) * 2) * 3 * 64 + lane;
      // regular-grid coordinates of this lane's tap
      const int tap = min(2 * t + kg, WK - 1);
      const int ti = (tap * 11) >> 5;
      const float ryk = (float)(ti - 1), rxk = (float)(tap - 3 * ti - 1);
      const bool tap_ok = (t < WSTEPS - 1) || kg == 0;
      float wgt[2][4];
      int qo[2];
      bool slow[2];
      f32x4 gat[2][8];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float dy, dx, m;
        if (HEADS) {
          dy = (ctf[r][0] * ryk + ctf[r][1] * rxk) - ryk + ctf[r][4];      // (T . R)[:,k] - R[:,k] + t   (networks.py:304-311)
          dx = (ctf[r][2] * ryk + ctf[r][3] * rxk) - rxk + ctf[r][5];
          m = 1.f / (1.f + __expf(-cm[r][t]));
        } else {
          dy = ca[r][t]; dx = cb[r][t]; m = cm[r][t];
        }
        const bool live = (r == 0 ? ok0 : ok1) && tap_ok;
        const float py = ((float)(y0 + 2 * pr + r) + ryk) + dy;
        const float px = ((float)(x0 + l31) + rxk) + dx;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const float lh = py - fy0, lw = px - fx0;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const int ry = (int)fy0 - (y0 - WPY0), rx = (int)fx0 - (x0 - WPX0);
        const bool in_win = (unsigned)ry <= (unsigned)(WPH - 2) && (unsigned)rx <= (unsigned)(WPW - 2);
        const bool fast = live && in_win;
        slow[r] = live && !in_win;
        const float mf = fast ? m : 0.f;
        const float hm = hh * mf, lm = lh * mf;
        wgt[r][0] = hm * hw; wgt[r][1] = hm * lw; wgt[r][2] = lm * hw; wgt[r][3] = lm * lw;
        qo[r] = fast ? (int)((__umul24((unsigned)ry, (unsigned)WPW) + (unsigned)rx) * WG) | ((rx >> 3) & 1) | ((((rx + 1) >> 3) & 1) << 1) : 0;
        const f32x4* qq = reinterpret_cast<const f32x4*>(win + (qo[r] & ~3));
        const int sl = qo[r] & 1, sr = (qo[r] >> 1) & 1;
#ifdef EAVSR_WS_EXP_NO_GATHER
        (void)qq;
#pragma unroll
        for (int j = 0; j < 8; ++j) gat[r][j] = f32x4{wgt[r][0], wgt[r][1], (float)sl, (float)sr};
        continue;
#endif
        gat[r][0] = qq[sl];                          // top-left, channels 0-3 (the halves are swizzled by column)
        gat[r][1] = qq[sl ^ 1];                      //           channels 4-7
        gat[r][2] = qq[2 + sr];                      // top-right
        gat[r][3] = qq[2 + (sr ^ 1)];
        gat[r][4] = qq[WPW * WG / 4 + sl];           // bottom-left
        gat[r][5] = qq[WPW * WG / 4 + (sl ^ 1)];
        gat[r][6] = qq[WPW * WG / 4 + 2 + sr];       // bottom-right
        gat[r][7] = qq[WPW * WG / 4 + 2 + (sr ^ 1)];
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float v[WG];
#pragma unroll
        for (int c = 0; c < WG; ++c) {
          const int j = c >> 2, e = c & 3;
          float tv = wgt[r][0] * gat[r][j][e];
          tv = __builtin_fmaf(wgt[r][1], gat[r][2 + j][e], tv);
          tv = __builtin_fmaf(wgt[r][2], gat[r][4 + j][e], tv);
          tv = __builtin_fmaf(wgt[r][3], gat[r][6 + j][e], tv);
          v[c] = tv;
        }
        // rare: the corners left the LDS window -> sampled from global memory (IL8: one corner = 32 contiguous bytes), with the
        // validity gate -1 < p < size and corner-wise zero padding applied explicitly; the LDS part contributed zero
        if (__builtin_amdgcn_ballot_w64(slow[r]) != 0) {
          if (slow[r]) {
            float dy, dx, m;
            if (HEADS) {
              dy = (ctf[r][0] * ryk + ctf[r][1] * rxk) - ryk + ctf[r][4];
              dx = (ctf[r][2] * ryk + ctf[r][3] * rxk) - rxk + ctf[r][5];
              m = 1.f / (1.f + __expf(-cm[r][t]));
            } else {
              dy = ca[r][t]; dx = cb[r][t]; m = cm[r][t];
            }
            const float py = ((float)(y0 + 2 * pr + r) + ryk) + dy;
            const float px = ((float)(x0 + l31) + rxk) + dx;
            const bool gate = py > -1.f && px > -1.f && py < (float)h && px < (float)w;   // false for NaN
            const float fy0 = floorf(py), fx0 = floorf(px);
            const float lh = py - fy0, lw = px - fx0;
            const float hm = (1.f - lh) * m, lm = lh * m, hw = 1.f - lw;
            const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
            const int hh_i = hl + 1, wh_i = wl + 1;
            const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
            const float cw[4] = {(gate & t_ok & l_ok) ? hm * hw : 0.f, (gate & t_ok & r_ok) ? hm * lw : 0.f,
                                 (gate & b_ok & l_ok) ? lm * hw : 0.f, (gate & b_ok & r_ok) ? lm * lw : 0.f};
            const int cy[2] = {min(max(hl, 0), h - 1), min(max(hh_i, 0), h - 1)};
            const int cx[2] = {min(max(wl, 0), w - 1), min(max(wh_i, 0), w - 1)};
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const f32x4* cp = reinterpret_cast<const f32x4*>(xg + ((size_t)cy[k4 >> 1] * w + cx[k4 & 1]) * WG);
              const f32x4 lo4 = cp[0], hi4 = cp[1];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] += cw[k4] * lo4[e];
                v[4 + e] += cw[k4] * hi4[e];
              }
            }
          }
        }
        u32x4 bt[3];
#pragma unroll
        for (int c = 0; c < WG / 2; ++c) {
          unsigned h2, m2, l2;
#ifdef EAVSR_WS_EXP_NO_SPLIT
          h2 = __float_as_uint(v[2 * c]); m2 = __float_as_uint(v[2 * c + 1]); l2 = h2 ^ m2;
#else
          ws_split2(v[2 * c], v[2 * c + 1], h2, m2, l2);
#endif
          bt[0][c] = h2; bt[1][c] = m2; bt[2][c] = l2;
        }
#pragma unroll
#ifdef EAVSR_WS_EXP_NO_BWRITE
        for (int j = 0; j < 3; ++j) if (bt[j][0] == 0x12345678u) bdst[(r * 3 + j) * 64] = bt[j];
#else
        for (int j = 0; j < 3; ++j) bdst[(r * 3 + j) * 64] = bt[j];
#endif
      }
    }
    if (last_g) {
      bn = nbn; y0 = ny0; x0 = nx0;
      pix0 = npix0; pix1 = npix1; ok0 = nok0; ok1 = nok1;
      g = 0;
      ++ti_;
    } else {
      ++g;
    }
  }
  // the contractors' lagging last phase
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
}

template <int NPROD, bool HEADS>
int launch_ws(const WSArgs& a, dim3 grid, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once_pd.flag[dev_], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_ws_kernel<NPROD, HEADS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)WLDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2_ws: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((dcnv2_ws_kernel<NPROD, HEADS>), grid, dim3(512), WLDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2_ws");
}

}  // namespace

extern "C" int eavsr_dcnv2_ws_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_x9,
                                  const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                  int32_t deform_groups, int32_t nprod, int32_t heads, void* stream) {
  EAVSR_REQUIRE(x_il8 && offset_or_heads && weight_x9 && out && (heads || mask), -1, "dcnv2_ws: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2_ws: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2_ws: cin %d not divisible by deform_groups %d", cin, deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2, "dcnv2_ws: %d channels per deformable group unsupported (must be a multiple of 8)", cpg);
  const int opg = cpg / 8;
  EAVSR_REQUIRE((opg & (opg - 1)) == 0, -2, "dcnv2_ws: cpg / 8 must be a power of two");
  EAVSR_REQUIRE(nprod == 6 || nprod == 9, -2, "dcnv2_ws: nprod %d (6 or 9)", nprod);
  EAVSR_REQUIRE((long)h * w * 32 < (1L << 31), -1, "dcnv2_ws: plane too large for 32-bit byte offsets");
  EAVSR_REQUIRE((long)h * w * 27 * deform_groups * 4 < (1L << 32), -1, "dcnv2_ws: offset tensor too large");
  EAVSR_REQUIRE((((uintptr_t)x_il8) & 15) == 0, -2, "dcnv2_ws: x must be 16-byte aligned");
  if (n == 0) return 0;
  WSArgs a;
  a.xil = x_il8; a.offset = offset_or_heads; a.mask = mask; a.wsplit = reinterpret_cast<const u32x4*>(weight_x9);
  a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups;
  a.opg_shift = 0;
  while ((1 << a.opg_shift) < opg) ++a.opg_shift;
  a.tiles_x = eavsr::cdiv(w, WT_W);
  a.tiles_y = eavsr::cdiv(h, WT_ROWS);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "dcnv2_ws: too many tiles");
  a.ntiles = (int)tiles;
  int cus = 256;
  {
    static int cu_cache[eavsr::kMaxDevices] = {};
    const int dev = eavsr::current_device();
    if (cu_cache[dev] == 0) {
      int v = 0;
      cu_cache[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    cus = cu_cache[dev];
  }
  dim3 grid((unsigned)(tiles < cus ? tiles : cus), eavsr::cdiv(cout, 64));
  hipStream_t st = eavsr::as_stream(stream);
  if (nprod == 9) return heads ? launch_ws<9, true>(a, grid, st) : launch_ws<9, false>(a, grid, st);
  return heads ? launch_ws<6, true>(a, grid, st) : launch_ws<6, false>(a, grid, st);
}
