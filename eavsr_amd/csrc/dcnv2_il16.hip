// DCNv2 forward in 16-bit (bf16 / fp16 samples and weights, fp32 bilinear blend and accumulation): the alignment kernel of
// the 16-bit mode (BASELINE.json configs[2] bf16, configs[4] fp16; SURVEY.md 8a: 688 B/px algorithmic in 16-bit).
//
// Same structure as dcnv2_il.hip (reference semantics: mmcv.ops.modulated_deform_conv2d as called at networks.py:627-630,
// optionally with AdaptBlockOffset's affine -> offsets expansion and mask sigmoid, networks.py:302-315, folded in):
//   * the sampled feature map arrives as "IL8" 16-bit: [n][c/8][h][w][8] of bf16 / fp16 -- one (pixel, group) is 16 bytes;
//     eavsr_flow_warp_pair_f32 writes it (the warp of networks.py:623 feeds nothing else), eavsr_nchw_to_il8_h16 converts
//   * persistent workgroups over a flattened (tile, group) sequence, 8 x 32-pixel tiles, the group's 20 x 48 window (15 KB)
//     and its 10 KB weight slab by LDS-DMA into two stages; a bilinear corner is ONE 16-byte read, a sample four
//   * the blend runs in fp32 on the unpacked corners (mask folded into the weights), is rounded ONCE to 16 bits and feeds
//     v_mfma_f32_32x32x16_{bf16,f16} straight from registers: 2 MFMAs per k-step (16 k = 2 taps x 8 channels) instead of
//     the 12 / 18 of the fp32-faithful kernel, and no operand split -- the kernel is bound by the sampler's vector work,
//     not by the matrix pipe
//   * output fp32 NCHW (the 1x1 fusion conv that follows takes fp32 sources), bias added in fp32.
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// parameter load as (wave-uniform base in a scalar register pair) + (32-bit lane offset): `global_load_dword v, v_off, s[base]` -- left
// alone the compiler forms a 64-bit vector address per load with a quarter-rate v_mad_u64_u32 (20 per (tile, group) step); the base
// goes through an empty asm as an integer (csrc/dcnv2_il2.hip's ld_b)
typedef const __attribute__((address_space(1))) char* h_gcp;
__device__ __forceinline__ float ld_b16v(const float* base, unsigned byte_off) {      // per-lane base (the rare fix-up path)
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld_b16(const float* base, unsigned byte_off) {
  unsigned long long b = reinterpret_cast<unsigned long long>(base);
  asm volatile("" : "+s"(b));
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<h_gcp>(b) + byte_off);
}

constexpr int HT_ROWS = 8, HT_W = 32;
constexpr int HK = 9, HSTEPS = 5;
constexpr int HPH = HT_ROWS + 12, HPW = 48, HPY0 = 6, HPX0 = 8;
constexpr int HWIN_U4 = HPH * HPW;                 // 960 16-byte units = 15 one-KiB pieces
constexpr int HWIN_SEGS = HWIN_U4 / 64;            // 15
constexpr int HW_U4 = HSTEPS * 2 * 64;             // 640 units = 10 pieces
constexpr int HW_SEGS = HW_U4 / 64;                // 10
constexpr int HNPIECE = HWIN_SEGS + HW_SEGS;       // 25
constexpr int HP_IT = (HNPIECE + 7) / 8;           // 4
constexpr size_t HLDS_BYTES = (2 * (size_t)HWIN_U4 + 2 * (size_t)HW_U4) * 16 + 64 * 4;   // 51,456

struct IL16Args {
  const void* xil;       // [n][cin/8][h][w][8] 16-bit
  const float* offset;   // explicit: (n, dg*18, h, w) fp32;  heads: (n, 15*dg, h, w) fp32
  const float* mask;
  const u32x4* wpack;    // [cot][group][step][mt][lane] 16-byte units (8 x 16-bit)
  const float* bias;
  float* out;            // (n, cout, h, w) fp32
  int n, cin, h, w, cout, dg, opg_shift, tiles_x, tiles_y, ntiles;
};

template <bool BF16>
__device__ __forceinline__ void unpack2(unsigned u, float& a, float& b) {
  if (BF16) {
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xFFFF0000u);
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, u);
    a = (float)v[0];
    b = (float)v[1];
  }
}

template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
  if (BF16) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, b2));
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2));
  }
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma_h(const u32x4& a, const u32x4& b, const f32x16& c) {
  if (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

template <bool BF16, bool HEADS>
__global__ __launch_bounds__(512, 2) void dcnv2_il16_kernel(IL16Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem16[];
  u32x4* s_win = reinterpret_cast<u32x4*>(smem16);                 // [2][HWIN_U4]
  u32x4* s_w = s_win + 2 * HWIN_U4;                                // [2][HW_U4]
  float* s_bias = reinterpret_cast<float*>(s_w + 2 * HW_U4);       // [64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const size_t pl4 = plane * 4;
  const int ngroups = a.cin / 8;
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
    y0 = ty * HT_ROWS;
    x0 = tx * HT_W;
  };

  // DMA piece p = i * 8 + wave: p < 15 window piece (64 units of one 16-byte (pixel, group) each), then 10 weight pieces
  int prc[2];
  unsigned poff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = (i * 8 + wave) * 64 + lane;
    const int rr = e / HPW, cc = e - rr * HPW;
    prc[i] = (rr << 8) | cc;
    poff[i] = (unsigned)((rr * w + cc) * 16);
  }
  auto issue = [&](int bn, int y0, int x0, int g, int stage) __attribute__((always_inline)) {
    const char* xorg = reinterpret_cast<const char*>(a.xil) + (((size_t)bn * ngroups + g) * plane) * 16 +
                       ((long)(y0 - HPY0) * w + (x0 - HPX0)) * 16;
    const char* wsrc = reinterpret_cast<const char*>(a.wpack + ((size_t)cot * ngroups + g) * HW_U4);
    const int ylo = y0 - HPY0, xlo = x0 - HPX0;
#pragma unroll
    for (int i = 0; i < HP_IT; ++i) {
      const int p = i * 8 + wave;      // wave-uniform
      if (p < HWIN_SEGS) {
        const bool ok = (unsigned)(ylo + (prc[i & 1] >> 8)) < (unsigned)h && (unsigned)(xlo + (prc[i & 1] & 255)) < (unsigned)w;
        u32x4* dst = s_win + stage * HWIN_U4 + p * 64;
        if (ok) __builtin_amdgcn_global_load_lds((gptr_t)(xorg + poff[i & 1]), (lptr_t)dst, 16, 0, 0);
        else dst[lane] = u32x4{0u, 0u, 0u, 0u};
      } else if (p < HNPIECE) {
        const int seg = p - HWIN_SEGS;
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(seg * 64 + lane) * 16u), (lptr_t)(s_w + stage * HW_U4 + seg * 64), 16, 0, 0);
      }
    }
  };

  auto grid_of = [&](int s, float& ry, float& rx) __attribute__((always_inline)) {
    const int tap = min(2 * s + kg, HK - 1);
    const int ti = (tap * 11) >> 5;
    ry = (float)(ti - 1);
    rx = (float)(tap - 3 * ti - 1);
  };
  float pa[HSTEPS], pb[HSTEPS], pm[HSTEPS], tf[6];
  struct POff { unsigned p4, po, pm; };
  auto make_poff = [&](unsigned pix) __attribute__((always_inline)) {
    POff o;
    o.p4 = pix * 4u;
    o.po = (pix + (kg ? 2u * uplane : 0u)) * 4u;
    o.pm = (pix + (kg ? uplane : 0u)) * 4u;
    return o;
  };
  auto load_params = [&](int bn, int g, const POff& o) __attribute__((always_inline)) {
    const int dgi = g >> a.opg_shift;
    if (HEADS) {
      const char* hb = reinterpret_cast<const char*>(a.offset) + (size_t)bn * 15 * a.dg * pl4;
      const char* mb = hb + (size_t)(6 * a.dg + dgi * 9) * pl4;
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[j] = ld_b16(reinterpret_cast<const float*>(hb + (size_t)(dgi * 4 + j) * pl4), o.p4);
#pragma unroll
      for (int j = 0; j < 2; ++j) tf[4 + j] = ld_b16(reinterpret_cast<const float*>(hb + (size_t)(4 * a.dg + dgi * 2 + j) * pl4), o.p4);
#pragma unroll
      for (int s = 0; s < HSTEPS; ++s)
        pm[s] = ld_b16(reinterpret_cast<const float*>(mb + (size_t)(2 * s) * pl4), s == HSTEPS - 1 ? o.p4 : o.pm);
    } else {
      const char* ob = reinterpret_cast<const char*>(a.offset) + ((size_t)bn * a.dg + dgi) * 18 * pl4;
      const char* mb = reinterpret_cast<const char*>(a.mask) + ((size_t)bn * a.dg + dgi) * 9 * pl4;
#pragma unroll
      for (int s = 0; s < HSTEPS; ++s) {
        const bool last = s == HSTEPS - 1;
        pa[s] = ld_b16(reinterpret_cast<const float*>(ob + (size_t)(4 * s) * pl4), last ? o.p4 : o.po);
        pb[s] = ld_b16(reinterpret_cast<const float*>(ob + (size_t)(4 * s + 1) * pl4), last ? o.p4 : o.po);
        pm[s] = ld_b16(reinterpret_cast<const float*>(mb + (size_t)(2 * s) * pl4), last ? o.p4 : o.pm);
      }
    }
  };

  int bn, y0, x0;
  tile_of(0, bn, y0, x0);
  int gy = y0 + wave, gx = x0 + l31;
  bool pix_ok = gy < h && gx < w;
  unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;

  issue(bn, y0, x0, 0, 0);
  load_params(bn, 0, make_poff(pix));
  if (tid < 64) {
    const int co = blockIdx.y * 64 + tid;
    s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
  }
  __syncthreads();
  f32x16 acc[2];
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = s_bias[m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg];
  };
  init_acc();

  const int total = my_tiles * ngroups;
  int g = 0, ti_ = 0;
  for (int it = 0; it < total; ++it) {
    const int stage = it & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) lgkmcnt(0): this step's window, weights and parameters have landed
    __syncthreads();
    const bool last_g = g + 1 == ngroups;
    const bool more = it + 1 < total;
    int nbn = bn, ny0 = y0, nx0 = x0;
    if (last_g && more) tile_of(ti_ + 1, nbn, ny0, nx0);
    const int ng = last_g ? 0 : g + 1;
    const int ngy = ny0 + wave, ngx = nx0 + l31;
    const bool npix_ok = ngy < h && ngx < w;
    const unsigned npix = npix_ok ? (unsigned)(ngy * w + ngx) : 0u;
    // this step's parameters move to working registers; the next step's DMA and parameters are requested at once
    float ca[HSTEPS], cb[HSTEPS], cm[HSTEPS], ctf[6];
#pragma unroll
    for (int s = 0; s < HSTEPS; ++s) { ca[s] = HEADS ? 0.f : pa[s]; cb[s] = HEADS ? 0.f : pb[s]; cm[s] = pm[s]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) ctf[j] = HEADS ? tf[j] : 0.f;
    if (more) issue(nbn, ny0, nx0, ng, stage ^ 1);
    load_params(more ? nbn : bn, more ? ng : g, make_poff(more ? npix : pix));

    const u32x4* win = s_win + stage * HWIN_U4;
    const u32x4* wst = s_w + stage * HW_U4 + lane;
    const float fgy = (float)gy, fgx = (float)gx;
    unsigned slow_steps = 0;
    float w1, w2, w3, w4;
    u32x4 cr[4];       // TL, TR, BL, BR: 8 x 16-bit each
    auto setup_gather = [&](int s) __attribute__((always_inline)) {
      float ryk, rxk;
      grid_of(s, ryk, rxk);
      float dy, dx, m;
      if (HEADS) {
        dy = (ctf[0] * ryk + ctf[1] * rxk) - ryk + ctf[4];      // (T . R)[:,k] - R[:,k] + t   (networks.py:304-311)
        dx = (ctf[2] * ryk + ctf[3] * rxk) - rxk + ctf[5];
        m = eavsr_sigmoid_fast(cm[s]);      // v_rcp_f32 (<= 1 ulp) instead of the IEEE division (10 instructions per sample), as dcnv2_il2.hip
      } else {
        dy = ca[s]; dx = cb[s]; m = cm[s];
      }
      const bool live = pix_ok && ((s < HSTEPS - 1) || kg == 0);
      const float py = (fgy + ryk) + dy;
      const float px = (fgx + rxk) + dx;
      const float fy0 = floorf(py), fx0 = floorf(px);
      const float lh = py - fy0, lw = px - fx0;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const int ry = (int)fy0 - (y0 - HPY0), rx = (int)fx0 - (x0 - HPX0);
      const bool in_win = (unsigned)ry <= (unsigned)(HPH - 2) && (unsigned)rx <= (unsigned)(HPW - 2);
      const bool fast = live && in_win;
      const float mf = fast ? m : 0.f;
      const float hm = hh * mf, lm = lh * mf;
      w1 = hm * hw; w2 = hm * lw; w3 = lm * hw; w4 = lm * lw;
      slow_steps |= (live && !in_win) ? (1u << s) : 0u;
      const u32x4* q = win + (fast ? (int)(__umul24((unsigned)ry, (unsigned)HPW) + (unsigned)rx) : 0);
      cr[0] = q[0]; cr[1] = q[1]; cr[2] = q[HPW]; cr[3] = q[HPW + 1];
    };
    auto blend = [&](u32x4& bop) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a0, a1, b0, b1, c0, c1, d0, d1;
        unpack2<BF16>(cr[0][c], a0, a1);
        unpack2<BF16>(cr[1][c], b0, b1);
        unpack2<BF16>(cr[2][c], c0, c1);
        unpack2<BF16>(cr[3][c], d0, d1);
        float v0 = w1 * a0, v1 = w1 * a1;
        v0 = __builtin_fmaf(w2, b0, v0); v1 = __builtin_fmaf(w2, b1, v1);
        v0 = __builtin_fmaf(w3, c0, v0); v1 = __builtin_fmaf(w3, c1, v1);
        v0 = __builtin_fmaf(w4, d0, v0); v1 = __builtin_fmaf(w4, d1, v1);
        bop[c] = pack2<BF16>(v0, v1);
      }
    };
    u32x4 bcur, bnext;
    setup_gather(0);
    blend(bcur);
#pragma unroll
    for (int t = 0; t < HSTEPS; ++t) {
      if (t + 1 < HSTEPS) setup_gather(t + 1);
      const u32x4 a0 = wst[(t * 2 + 0) * 64], a1 = wst[(t * 2 + 1) * 64];
      acc[0] = mfma_h<BF16>(a0, bcur, acc[0]);
      acc[1] = mfma_h<BF16>(a1, bcur, acc[1]);
      if (t + 1 < HSTEPS) {
        blend(bnext);
        bcur = bnext;
      }
    }

    // rare: a corner left the LDS window: redone from global memory with the validity gate and corner-wise zero padding
    if (__builtin_amdgcn_ballot_w64(slow_steps != 0) != 0) {
      const u32x4* xg = reinterpret_cast<const u32x4*>(a.xil) + ((size_t)bn * ngroups + g) * plane;
      for (int t = 0; t < HSTEPS; ++t) {
        const bool mine = (slow_steps >> t) & 1u;
        if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
        u32x4 b = u32x4{0u, 0u, 0u, 0u};
        if (mine) {
          const int tap = min(2 * t + kg, HK - 1);
          const float ryt = (float)(tap / 3 - 1), rxt = (float)(tap % 3 - 1);
          // rare path: the parameters are read again (their registers were recycled)
          const int dgi = g >> a.opg_shift;
          float dy, dx, m;
          if (HEADS) {
            const char* hb = reinterpret_cast<const char*>(a.offset) + (size_t)bn * 15 * a.dg * pl4;
            float tt[6];
#pragma unroll
            for (int j = 0; j < 4; ++j) tt[j] = ld_b16v(reinterpret_cast<const float*>(hb + (size_t)(dgi * 4 + j) * pl4), pix * 4u);
#pragma unroll
            for (int j = 0; j < 2; ++j) tt[4 + j] = ld_b16v(reinterpret_cast<const float*>(hb + (size_t)(4 * a.dg + dgi * 2 + j) * pl4), pix * 4u);
            dy = (tt[0] * ryt + tt[1] * rxt) - ryt + tt[4];
            dx = (tt[2] * ryt + tt[3] * rxt) - rxt + tt[5];
            m = eavsr_sigmoid_fast(ld_b16v(reinterpret_cast<const float*>(hb + (size_t)(6 * a.dg + dgi * 9 + tap) * pl4), pix * 4u));
          } else {
            const char* ob = reinterpret_cast<const char*>(a.offset) + ((size_t)bn * a.dg + dgi) * 18 * pl4;
            const char* mb = reinterpret_cast<const char*>(a.mask) + ((size_t)bn * a.dg + dgi) * 9 * pl4;
            dy = ld_b16v(reinterpret_cast<const float*>(ob + (size_t)(2 * tap) * pl4), pix * 4u);
            dx = ld_b16v(reinterpret_cast<const float*>(ob + (size_t)(2 * tap + 1) * pl4), pix * 4u);
            m = ld_b16v(reinterpret_cast<const float*>(mb + (size_t)tap * pl4), pix * 4u);
          }
          const float py = (fgy + ryt) + dy, px = (fgx + rxt) + dx;
          if (!(py > -1.f && px > -1.f && py < (float)h && px < (float)w)) m = 0.f;
          const float fy0 = floorf(py), fx0 = floorf(px);
          const float lh = py - fy0, lw = px - fx0;
          const float hm = (1.f - lh) * m, lm = lh * m, hw = 1.f - lw;
          const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
          const int hh_i = hl + 1, wh_i = wl + 1;
          const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
          const float cw[4] = {(t_ok & l_ok) ? hm * hw : 0.f, (t_ok & r_ok) ? hm * lw : 0.f,
                               (b_ok & l_ok) ? lm * hw : 0.f, (b_ok & r_ok) ? lm * lw : 0.f};
          const int cy[2] = {min(max(hl, 0), h - 1), min(max(hh_i, 0), h - 1)};
          const int cx[2] = {min(max(wl, 0), w - 1), min(max(wh_i, 0), w - 1)};
          float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const u32x4 cv = xg[(size_t)cy[k4 >> 1] * w + cx[k4 & 1]];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float e0, e1;
              unpack2<BF16>(cv[c], e0, e1);
              v[2 * c] += cw[k4] * e0;
              v[2 * c + 1] += cw[k4] * e1;
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) b[c] = pack2<BF16>(v[2 * c], v[2 * c + 1]);
        }
        const u32x4 a0 = wst[(t * 2 + 0) * 64], a1 = wst[(t * 2 + 1) * 64];
        acc[0] = mfma_h<BF16>(a0, b, acc[0]);
        acc[1] = mfma_h<BF16>(a1, b, acc[1]);
      }
    }

    if (last_g) {
      if (pix_ok) {
        unsigned pl4u = uplane * 4u;
        asm volatile("" : "+s"(pl4u));      // keeps the 32 channel offsets from being hoisted into long-lived scalars
        const char* ob = reinterpret_cast<const char*>(a.out + ((size_t)bn * a.cout + (size_t)cot * 64) * plane);
        unsigned voff = ((unsigned)(gy * w + gx)) * 4u + (kg ? 4u * pl4u : 0u);
        const bool full = cot * 64 + 64 <= a.cout;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
            float* q = reinterpret_cast<float*>(const_cast<char*>(ob) + voff);
            if (full || cot * 64 + cu + 4 * kg < a.cout) *q = acc[m][e];
            voff += ((e & 3) == 3 ? 5u : 1u) * pl4u;
          }
      }
      init_acc();
      bn = nbn; y0 = ny0; x0 = nx0;
      gy = ngy; gx = ngx; pix_ok = npix_ok; pix = npix;
      g = 0;
      ++ti_;
    } else {
      ++g;
    }
  }
}

// (n, c, h, w) fp32 -> IL8 16-bit [n][c/8][h][w][8]
template <bool BF16>
__global__ __launch_bounds__(256) void nchw_to_il8_h16_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int oct, int hw) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int o = blockIdx.y, bn = blockIdx.z;
  const float* xp = x + ((size_t)bn * oct + o) * 8 * hw + p;
  u32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = pack2<BF16>(xp[(size_t)(2 * e) * hw], xp[(size_t)(2 * e + 1) * hw]);
  out[((size_t)bn * oct + o) * hw + p] = v;
}

// weight (cout, cin, 3, 3) fp32 -> [cot][group][step][mt][lane] 16-byte units: lane (m = lane & 31, kg = lane >> 5) holds
// row co = 64 cot + 32 mt + m, k = channels 8 g .. 8 g + 7 of tap 2 s + kg (zero for tap 9 / co >= cout), rounded once
template <bool BF16>
__global__ void pack_il16_kernel(const float* __restrict__ wt, unsigned* __restrict__ out, int cout, int cin, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;      // one packed pair per thread
  if (e >= total) return;
  const int j = (int)(e & 3);
  long u = e >> 2;
  const int lane = (int)(u % 64); u /= 64;
  const int mt = (int)(u % 2); u /= 2;
  const int s = (int)(u % HSTEPS); u /= HSTEPS;
  const int ngroups = cin / 8;
  const int g = (int)(u % ngroups);
  const int cot = (int)(u / ngroups);
  const int co = cot * 64 + mt * 32 + (lane & 31);
  const int tap = 2 * s + (lane >> 5);
  float v0 = 0.f, v1 = 0.f;
  if (co < cout && tap < HK) {
    v0 = wt[((size_t)co * cin + g * 8 + 2 * j) * HK + tap];
    v1 = wt[((size_t)co * cin + g * 8 + 2 * j + 1) * HK + tap];
  }
  out[e] = pack2<BF16>(v0, v1);
}

template <bool BF16, bool HEADS>
int launch_il16(const IL16Args& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((dcnv2_il16_kernel<BF16, HEADS>), grid, dim3(512), HLDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2_il16");
}

}  // namespace

extern "C" int64_t eavsr_dcn_il16_weight_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % 8 != 0) return 0;
  return (int64_t)eavsr::cdiv(cout, 64) * (cin / 8) * HW_U4 * 16;
}

extern "C" int eavsr_pack_dcn_il16_weight(const float* weight, void* packed, int32_t cout, int32_t cin, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_dcn_il16_weight: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % 8 == 0 && (dtype == 1 || dtype == 2), -1, "pack_dcn_il16_weight: bad args");
  const long total = eavsr_dcn_il16_weight_bytes(cout, cin) / 4;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == 2) hipLaunchKernelGGL(pack_il16_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), weight, (unsigned*)packed, cout, cin, total);
  else hipLaunchKernelGGL(pack_il16_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), weight, (unsigned*)packed, cout, cin, total);
  return eavsr::launch_status("pack_dcn_il16_weight");
}

extern "C" int eavsr_nchw_to_il8_h16(const float* x, void* out, int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(x && out, -1, "nchw_to_il8_h16: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && c % 8 == 0 && (dtype == 1 || dtype == 2), -1, "nchw_to_il8_h16: bad args");
  EAVSR_REQUIRE((long)h * w < (1L << 28) && c / 8 <= 65535 && n <= 65535, -1, "nchw_to_il8_h16: too large");
  if (n == 0) return 0;
  const int hw = h * w;
  dim3 grid(eavsr::cdiv(hw, 256), c / 8, n);
  if (dtype == 2) hipLaunchKernelGGL(nchw_to_il8_h16_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), x, (u32x4*)out, c / 8, hw);
  else hipLaunchKernelGGL(nchw_to_il8_h16_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), x, (u32x4*)out, c / 8, hw);
  return eavsr::launch_status("nchw_to_il8_h16");
}

extern "C" int eavsr_dcnv2_il16(const void* x_il8_h16, const float* offset_or_heads, const float* mask, const void* weight_il16,
                                const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                int32_t deform_groups, int32_t dtype, int32_t heads, void* stream) {
  EAVSR_REQUIRE(x_il8_h16 && offset_or_heads && weight_il16 && out && (heads || mask), -1, "dcnv2_il16: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2_il16: bad dims");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "dcnv2_il16: dtype %d (1 = fp16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2_il16: cin %d not divisible by deform_groups %d", cin, deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2, "dcnv2_il16: %d channels per deformable group unsupported (must be a multiple of 8)", cpg);
  const int opg = cpg / 8;
  EAVSR_REQUIRE((opg & (opg - 1)) == 0, -2, "dcnv2_il16: cpg / 8 must be a power of two");
  EAVSR_REQUIRE((long)h * w * 16 < (1L << 31), -1, "dcnv2_il16: plane too large for 32-bit byte offsets");
  EAVSR_REQUIRE((long)h * w * 27 * deform_groups * 4 < (1L << 32), -1, "dcnv2_il16: offset tensor too large");
  EAVSR_REQUIRE((((uintptr_t)x_il8_h16) & 15) == 0, -2, "dcnv2_il16: x must be 16-byte aligned");
  if (n == 0) return 0;
  IL16Args a;
  a.xil = x_il8_h16; a.offset = offset_or_heads; a.mask = mask; a.wpack = reinterpret_cast<const u32x4*>(weight_il16);
  a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups;
  a.opg_shift = 0;
  while ((1 << a.opg_shift) < opg) ++a.opg_shift;
  a.tiles_x = eavsr::cdiv(w, HT_W);
  a.tiles_y = eavsr::cdiv(h, HT_ROWS);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "dcnv2_il16: too many tiles");
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
  const long slots = cus;           // one persistent workgroup per CU (at 128 registers for two per CU the kernel spills)
  dim3 grid((unsigned)(tiles < slots ? tiles : slots), eavsr::cdiv(cout, 64));
  hipStream_t st = eavsr::as_stream(stream);
  if (dtype == 2) return heads ? launch_il16<true, true>(a, grid, st) : launch_il16<true, false>(a, grid, st);
  return heads ? launch_il16<false, true>(a, grid, st) : launch_il16<false, false>(a, grid, st);
}
