// DCNv2 forward, the hot-path kernel of round 2 (a7; north star: >= 30 % of the HBM roofline).
//
// Reference semantics: mmcv.ops.modulated_deform_conv2d as called at models/networks.py:627-630 (see dcnv2.hip for
// the restated algorithm), optionally with the affine -> 18 offsets expansion and the mask sigmoid of
// AdaptBlockOffset (networks.py:302-315) folded into the sampler's per-group set-up ("heads" mode: the kernel reads
// the 15 D head channels of the predictor instead of 27 D offset / mask channels that another kernel wrote).
//
// What changed against dcnv2_x9.hip (round 1), and why -- that kernel was bound three ways at once (matrix pipe,
// LDS and vector issue all near 100 % of the same interval, so nothing overlapped):
//   * INPUT LAYOUT "IL8": the sampled feature map arrives as [n][c/8][h][w][8] (the 8 channels of a deformable group
//     interleaved per pixel, 32 bytes).  The producer is the flow_warp just before it (networks.py:623), which writes
//     that layout for free; eavsr_nchw_to_il8_f32 converts for every other caller.  A bilinear corner of a (pixel, tap)
//     is then ONE contiguous 32-byte read: a sample is 8 ds_read_b128 (256 B/clk) instead of 16 ds_read2_b32
//     (128 B/clk) -- half the LDS time of the sampler -- and the 8 values a lane needs for its MFMA B operand
//     (k = 8 channels of one tap) arrive already in k order.  The window DMA moves whole 1536-byte window rows.
//   * PRODUCTS: NPROD = 9 is the exact bf16x9 contraction of round 1 (every fp32 operand split exactly into three
//     bf16 terms, all nine partial products); NPROD = 6 drops the three products below 2^-23 of the result
//     (mid*lo, lo*mid, lo*lo): per product that is at most 2 * 2^-24 relative, the size of ONE fp32 rounding, and a
//     third less matrix work.  Both are fp32-faithful; tests compare both with an fp64 evaluation.
//   * persistent workgroups over a flattened (tile, group) sequence: the window / weight DMA of the next tile's first
//     group runs under the last group of the current one.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16): M = 32 output channels (2 tiles), N = 32 pixels of one image row,
// k-step = 16 k = 2 taps x 8 channels of one group; lane (n = lane & 31, kg = lane >> 5) samples pixel n for tap
// 2 s + kg of k-step s and feeds its 8 blended, split values straight from registers (no column tile in LDS).
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ld_b(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

constexpr int IT_ROWS = 8, IT_W = 32;                 // pixel tile: one 32-pixel row per wave
constexpr int IG = 8;                                 // channels per group = k of one tap
constexpr int IK = 9, ISTEPS = 5;                     // taps; k-steps per group (taps 2s, 2s+1; tap 9 is zero)
constexpr int IPH = IT_ROWS + 12, IPW = 48;           // LDS window rows y0-6 .. y0+13, columns x0-8 .. x0+39
constexpr int IPY0 = 6, IPX0 = 8;
constexpr int IWIN_F = IPH * IPW * IG;                // 7680 floats = 30 one-KiB DMA pieces
constexpr int IWIN_SEGS = IWIN_F / 256;               // 30
constexpr int IW_U4 = ISTEPS * 3 * 2 * 64;            // 16-byte elements of one group's weight slab (CO = 64)
constexpr int IW_SEGS = IW_U4 / 64;                   // 30
constexpr int INPIECE = IWIN_SEGS + IW_SEGS;          // 60
constexpr int IP_IT = (INPIECE + 7) / 8;              // pieces per wave: 8
constexpr size_t ILDS_BYTES = 2 * (size_t)IWIN_F * 4 + 2 * (size_t)IW_U4 * 16 + 64 * 4;   // 123,136 (+ the bias)

struct ILArgs {
  const float* xil;      // [n][cin/8][h][w][8]
  const float* offset;   // explicit mode: (n, dg*18, h, w);  heads mode: (n, 15*dg, h, w)
  const float* mask;     // explicit mode: (n, dg*9, h, w);   heads mode: unused
  const u32x4* wsplit;   // [cot][group][step][term][mt][lane] 16-byte elements (eavsr_pack_dcn_weight_x9)
  const float* bias;
  float* out;            // (n, cout, h, w)
  int n, cin, h, w, cout, dg, cpg, opg_shift, tiles_x, tiles_y, ntiles;   // opg_shift: log2(octets per deformable group)
};

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void il_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 il_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef EAVSR_IL_STAMPS
// diagnostic build only (tools/build_il_diag.sh): shader cycles per phase, summed over wave 0 and wave 4 of every workgroup
__device__ unsigned long long g_il_stamps[16];
#define IL_STAMP(i)                                                   \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define IL_STAMP(i) do { } while (0)
#endif

template <int NPROD, bool HEADS>
__global__ __launch_bounds__(512, 2) void dcnv2_il_kernel(ILArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_win = smem;                                                   // [2][IWIN_F]
  u32x4* s_w = reinterpret_cast<u32x4*>(smem + 2 * IWIN_F);              // [2][IW_U4]
  float* s_bias = smem + 2 * IWIN_F + 2 * IW_U4 * 4;                      // [64]: the accumulators start from it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const int ngroups = a.cin / IG;
  const int cot = blockIdx.y;

  // persistent tile walk, XCD-aware: workgroups b, b + 8, .. share an XCD (and its L2); XCD x owns a contiguous run of
  // tiles, so that tiles sharing halo rows / columns are fetched through the same L2
  const int nb = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd_wg = (nb + 7 - xcd) >> 3;                      // workgroups on this XCD
  const int q = a.ntiles >> 3, r = a.ntiles & 7;
  const int t_begin = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int t_count = q + (xcd < r ? 1 : 0);
  const int my_tiles = slot < t_count ? (t_count - slot + per_xcd_wg - 1) / per_xcd_wg : 0;
  if (my_tiles == 0) return;
  auto tile_of = [&](int i, int& bn, int& y0, int& x0) __attribute__((always_inline)) {
    int t = t_begin + slot + i * per_xcd_wg;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    bn = t / a.tiles_y;
    y0 = ty * IT_ROWS;
    x0 = tx * IT_W;
  };

  // DMA pieces of a (tile, group): window piece p = i * 8 + wave (p < IWIN_SEGS; 64 lanes x 16 bytes = 32 window
  // positions, the window row is 96 such 16-byte units) and weight-slab piece seg = i * 8 + wave (seg < IW_SEGS)
  constexpr int WIN_IT = (IWIN_SEGS + 7) / 8, W_IT = (IW_SEGS + 7) / 8;
  int prc[WIN_IT];                   // (window row << 8) | window column of this lane's 16-byte unit in piece i
  unsigned poff[WIN_IT];             // its byte offset from the window's origin pixel in the IL8 image: fixed per kernel
#pragma unroll
  for (int i = 0; i < WIN_IT; ++i) {
    const int e4 = (i * 8 + wave) * 64 + lane;
    const int rr = e4 / (2 * IPW);
    const int cc = e4 - rr * (2 * IPW);
    prc[i] = (rr << 8) | (cc >> 1);
    // LDS bank swizzle: the two 16-byte halves (channels 0-3 / 4-7) of window column x are stored swapped when bit 3 of x
    // is set.  A wave's 16-lane ds_read_b128 groups hold pixels 8 columns apart (stride 32 B: x and x + 8 are 256 bytes
    // apart = the same banks); reading "the low half" they collided two by two, swapped they cover all 64 banks.
    poff[i] = (unsigned)((rr * w + (cc >> 1)) * 32 + (((cc & 1) ^ ((cc >> 4) & 1)) * 16));
  }
  // Window units inside the image are fetched by LDS-DMA from (wave-uniform window origin) + (per-lane constant offset);
  // units outside the image are written as zeros (that IS the sampler's zero padding).  Every unit of a stage is
  // rewritten by every step, one way or the other.
  auto win_origin = [&](int bn, int y0, int x0, int g) __attribute__((always_inline)) {
    return reinterpret_cast<const char*>(a.xil + ((size_t)bn * ngroups + g) * plane * IG) +
           ((long)(y0 - IPY0) * w + (x0 - IPX0)) * 32;          // may point before the image: only `ok` lanes use it
  };
  auto issue_win = [&](int i, const char* xorg, int y0, int x0, int stage) __attribute__((always_inline)) {
    const int ylo = y0 - IPY0, xlo = x0 - IPX0;
    const int p = i * 8 + wave;  // wave-uniform
    if (p < IWIN_SEGS) {
      const bool ok = (unsigned)(ylo + (prc[i] >> 8)) < (unsigned)h && (unsigned)(xlo + (prc[i] & 255)) < (unsigned)w;
      float* dst = s_win + stage * IWIN_F + p * 256;
      if (ok) __builtin_amdgcn_global_load_lds((gptr_t)(xorg + poff[i]), (lptr_t)dst, 16, 0, 0);
      else *reinterpret_cast<f32x4*>(dst + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto issue_wgt = [&](int i, const char* wsrc, int stage) __attribute__((always_inline)) {
    const int seg = i * 8 + wave;
    if (seg < IW_SEGS)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(seg * 64 + lane) * 16u),
                                       (lptr_t)(s_w + stage * IW_U4 + seg * 64), 16, 0, 0);
  };
  auto issue = [&](int bn, int y0, int x0, int g, int stage) __attribute__((always_inline)) {
    const char* xorg = win_origin(bn, y0, x0, g);
    const char* wsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)cot * ngroups + g) * IW_U4);
#pragma unroll
    for (int i = 0; i < WIN_IT; ++i) issue_win(i, xorg, y0, x0, stage);
#pragma unroll
    for (int i = 0; i < W_IT; ++i) issue_wgt(i, wsrc, stage);
  };

  // per-lane tap of k-step s: 2 s + kg (tap 9 does not exist: kg = 1 lanes idle in the last step); its regular-grid
  // coordinates (-1, 0, 1) are recomputed where needed (s is a compile-time constant, kg one bit)
  auto grid_of = [&](int s, float& ry, float& rx) __attribute__((always_inline)) {
    const int tap = min(2 * s + kg, IK - 1);
    const int ti = (tap * 11) >> 5;          // tap / 3 for tap < 9
    ry = (float)(ti - 1);
    rx = (float)(tap - 3 * ti - 1);
  };

  // raw per-group sampling parameters (dead once set-up has turned them into weights and addresses; the next
  // group's values are loaded straight into them).  explicit: (dy, dx, mask) per k-step; heads: the 2x2 transform,
  // the translation and the mask logits of the lane's taps
  float pa[ISTEPS], pb[ISTEPS], pm[ISTEPS];
  float tf[6];
  // Addressing: wave-uniform base (SGPRs: image, group, 2 s) + per-lane 32-bit byte offset (pixel, and the lane half's
  // tap parity) -- three VGPRs for all 15 loads of a step instead of one hoisted offset register per load.
  struct POff { unsigned p4, po, pm; };     // pix * 4;  + kg * 2 planes (offset channels);  + kg * 1 plane (mask channels)
  auto make_poff = [&](unsigned pix) __attribute__((always_inline)) {
    POff o;
    o.p4 = pix * 4u;
    o.po = (pix + (kg ? 2u * uplane : 0u)) * 4u;
    o.pm = (pix + (kg ? uplane : 0u)) * 4u;
    return o;
  };
  // wave-uniform byte strides, computed once: the per-slot bases below are then additions, not 64-bit multiplications
  const size_t pl4 = plane * 4;
  struct PBase { const char* off; const char* msk; const char* aff; const char* trn; };   // explicit: offset, mask;  heads: mask logits, 2x2 transform, translation
  auto make_pbase = [&](int bn, int g) __attribute__((always_inline)) {
    const int dgi = g >> a.opg_shift;
    PBase b;
    if (HEADS) {
      const char* hb = reinterpret_cast<const char*>(a.offset) + (size_t)bn * 15 * a.dg * pl4;
      b.off = nullptr;
      b.msk = hb + (size_t)(6 * a.dg + dgi * 9) * pl4;
      b.aff = hb + (size_t)(dgi * 4) * pl4;
      b.trn = hb + (size_t)(4 * a.dg + dgi * 2) * pl4;
    } else {
      b.off = reinterpret_cast<const char*>(a.offset) + ((size_t)bn * a.dg + dgi) * 18 * pl4;
      b.msk = reinterpret_cast<const char*>(a.mask) + ((size_t)bn * a.dg + dgi) * 9 * pl4;
      b.aff = nullptr;
      b.trn = nullptr;
    }
    return b;
  };
  auto load_slot = [&](const PBase& pb_, const POff& o, int s) __attribute__((always_inline)) {
    const bool lastslot = s == ISTEPS - 1;          // taps 8 / (9): both lane halves read tap 8
#ifdef EAVSR_IL_EXP_NO_PARAMS
    pa[s] = 0.25f; pb[s] = 0.25f; pm[s] = 0.5f;
    (void)pb_; (void)o; (void)lastslot;
    return;
#endif
    pm[s] = ld_b(reinterpret_cast<const float*>(pb_.msk + (size_t)(2 * s) * pl4), lastslot ? o.p4 : o.pm);
    if (!HEADS) {
      pa[s] = ld_b(reinterpret_cast<const float*>(pb_.off + (size_t)(4 * s) * pl4), lastslot ? o.p4 : o.po);
      pb[s] = ld_b(reinterpret_cast<const float*>(pb_.off + (size_t)(4 * s + 1) * pl4), lastslot ? o.p4 : o.po);
    }
  };
  auto load_affine = [&](const PBase& pb_, const POff& o) __attribute__((always_inline)) {
#ifdef EAVSR_IL_EXP_NO_PARAMS
    tf[0] = 1.1f; tf[1] = 0.1f; tf[2] = -0.1f; tf[3] = 0.9f; tf[4] = 0.3f; tf[5] = -0.3f;
    (void)pb_; (void)o;
    return;
#endif
    if (HEADS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[j] = ld_b(reinterpret_cast<const float*>(pb_.aff + (size_t)j * pl4), o.p4);
#pragma unroll
      for (int j = 0; j < 2; ++j) tf[4 + j] = ld_b(reinterpret_cast<const float*>(pb_.trn + (size_t)j * pl4), o.p4);
    }
  };

  // waves 4-7 are dispatched second and lose every age-based arbitration on their SIMD (in-kernel stamps: waves 0-3 wait
  // 20 % of the step at the barrier for them): static priority for that half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

  // ---- current tile -------------------------------------------------------------------------------------------
  int bn, y0, x0;
  tile_of(0, bn, y0, x0);
  int gy = y0 + wave, gx = x0 + l31;
  bool pix_ok = gy < h && gx < w;
  unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;

  // every window position is (re)written by each step's DMA, from the image or from the zero line: no LDS pre-fill
  issue(bn, y0, x0, 0, 0);
  {
    const POff o0 = make_poff(pix);
    const PBase b0 = make_pbase(bn, 0);
    load_affine(b0, o0);
#pragma unroll
    for (int s = 0; s < ISTEPS; ++s) load_slot(b0, o0, s);
  }

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

  // finished tile whose accumulators still have to leave: stored at the TOP of the next step, in front of that step's
  // DMA / parameter loads, so that the stores drain under a whole step instead of in front of a barrier
  bool st_pending = false;
  int st_bn = 0, st_gy = 0, st_gx = 0;
  bool st_ok = false;
  auto store_tile = [&]() __attribute__((always_inline)) {
#ifdef EAVSR_IL_EXP_NO_STORE
    if (st_ok && acc[0][0] == 12345.678f) {
#else
    if (st_ok) {
#endif
      // lane part of the address: pixel + this lane half's 4 channels; the channel stride is added as the loop goes.
      // `pl4` is laundered through an empty asm so that the 32 channel offsets are computed here, once per tile, instead
      // of being hoisted out of the step loop into 32 long-lived scalar registers (which then spill).
      unsigned pl4 = uplane * 4u;
      asm volatile("" : "+s"(pl4));
      const char* ob = reinterpret_cast<const char*>(a.out + ((size_t)st_bn * a.cout + (size_t)cot * 64) * plane);
      unsigned voff = ((unsigned)(st_gy * w + st_gx)) * 4u + (kg ? 4u * pl4 : 0u);
      const bool full = cot * 64 + 64 <= a.cout;      // wave-uniform
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int cu = m * 32 + (e & 3) + 8 * (e >> 2);
          // channel cu + 4 kg: offsets advance by 1 plane, and by 5 planes across a group of four
          float* q = reinterpret_cast<float*>(const_cast<char*>(ob) + voff);
          if (full || cot * 64 + cu + 4 * kg < a.cout) *q = acc[m][e];
          voff += ((e & 3) == 3 ? 5u : 1u) * pl4;
        }
    }
    init_acc();
  };

  const int total = my_tiles * ngroups;
  int g = 0, ti_ = 0;
#ifdef EAVSR_IL_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  for (int it = 0; it < total; ++it) {
    const int stage = it & 1;
    IL_STAMP(0);
    // window, weights and sampling parameters of this step have landed (all were requested at the top of the previous
    // step: a whole step of latency cover); the barrier publishes everybody's DMA share and retires the other stage
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) lgkmcnt(0)
    IL_STAMP(1);
    __syncthreads();
    IL_STAMP(2);
    if (st_pending) {      // wave-uniform
      store_tile();
      st_pending = false;
    }
    IL_STAMP(3);
    // the step after this one: next group of this tile, or group 0 of the next tile
    const bool last_g = g + 1 == ngroups;
    const bool more = it + 1 < total;
    int nbn = bn, ny0 = y0, nx0 = x0;
    if (last_g && more) tile_of(ti_ + 1, nbn, ny0, nx0);
    const int ng = last_g ? 0 : g + 1;
    // (the DMA and the parameter loads of the next step are issued piecewise inside the pipeline below: all eight waves
    // issuing 8 LDS-DMAs + 15 loads each right behind the barrier cost 20 % of the step in in-kernel stamps)
    const int ngy = ny0 + wave, ngx = nx0 + l31;
    const bool npix_ok = ngy < h && ngx < w;
    const unsigned npix = npix_ok ? (unsigned)(ngy * w + ngx) : 0u;

    // Sampling parameters: this step's move to working registers; the NEXT step's are requested inside the pipeline (the
    // last step reloads its own values so that the code has no branch there).
    float ca[ISTEPS], cb[ISTEPS], cm[ISTEPS], ctf[6];
#pragma unroll
    for (int s = 0; s < ISTEPS; ++s) { ca[s] = pa[s]; cb[s] = pb[s]; cm[s] = pm[s]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) ctf[j] = HEADS ? tf[j] : 0.f;
    const POff lo = make_poff(more ? npix : pix);
    const PBase lb = make_pbase(more ? nbn : bn, more ? ng : g);
    const char* nxorg = win_origin(nbn, ny0, nx0, ng);
    const char* nwsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)cot * ngroups + ng) * IW_U4);
    __builtin_amdgcn_sched_barrier(0);
    IL_STAMP(4);

    const float* wst_win = s_win + stage * IWIN_F;
    const u32x4* wst = s_w + stage * IW_U4 + lane;
    const float fgy = (float)gy, fgx = (float)gx;

    struct Pos {
      float w1, w2, w3, w4;   // bilinear corner weights x mask (0 when the tap is idle or the sample is not served from LDS)
      int q;                  // float offset of the top-left corner in the LDS window stage, | swizzle bits of the two
                              // corner columns in bits 0 (left) and 1 (right): the corner itself is 32-byte aligned
    };
    Pos pos[2];
    unsigned slow_steps = 0;
    // Set-up of one (pixel, tap).  The window is zero outside the image, which IS the sampler's corner-wise zero padding
    // and also its validity gate (-1 < p < size: outside it both corners of that axis lie outside the image), so the fast
    // path needs no image-bounds test at all; samples whose corners leave the WINDOW are flagged and redone from global
    // memory, where the gate and the padding are applied explicitly.
    auto setup = [&](int s) __attribute__((always_inline)) {
      float ryk, rxk;
      grid_of(s, ryk, rxk);
      float dy, dx, m;
      if (HEADS) {
        // (T . R)[:,k] - R[:,k] + t   (matmul, subtract, add: networks.py:304-311)
        dy = (ctf[0] * ryk + ctf[1] * rxk) - ryk + ctf[4];
        dx = (ctf[2] * ryk + ctf[3] * rxk) - rxk + ctf[5];
        m = 1.f / (1.f + __expf(-cm[s]));
      } else {
        dy = ca[s]; dx = cb[s]; m = cm[s];
      }
      const bool live = pix_ok && ((s < ISTEPS - 1) || kg == 0);
      const float py = (fgy + ryk) + dy;
      const float px = (fgx + rxk) + dx;
      const float fy0 = floorf(py), fx0 = floorf(px);
      const float lh = py - fy0, lw = px - fx0;
      const float hh = 1.f - lh, hw = 1.f - lw;
      // v_cvt_i32_f32 saturates (and maps NaN to 0): wild offsets stay defined and simply fail the window test
      const int ry = (int)fy0 - (y0 - IPY0), rx = (int)fx0 - (x0 - IPX0);
      const bool in_win = (unsigned)ry <= (unsigned)(IPH - 2) && (unsigned)rx <= (unsigned)(IPW - 2);
      const bool fast = live && in_win;
      const float mf = fast ? m : 0.f;
      const float hm = hh * mf, lm = lh * mf;
      Pos& ps = pos[s & 1];
      ps.w1 = hm * hw; ps.w2 = hm * lw; ps.w3 = lm * hw; ps.w4 = lm * lw;
      ps.q = fast ? (int)((__umul24((unsigned)ry, (unsigned)IPW) + (unsigned)rx) * IG) | ((rx >> 3) & 1) | ((((rx + 1) >> 3) & 1) << 1) : 0;
      // outside the window: redone on the global path (which applies the validity gate -1 < p < size itself)
      slow_steps |= (live && !in_win) ? (1u << s) : 0u;
    };

    // Software pipeline over the k-steps, written out as fenced chunks so that the order below IS the issue order:
    // iteration t (MFMAs of k-step t) = [8 gathers + 6 A-operand reads of step t+1]  then per chunk a few MFMAs of step t
    // followed by vector work that runs in their shadow: the set-up of step t+2, an empty chunk (LDS latency cover), then
    // blend + exact split of step t+1 one channel pair at a time.  Every MFMA operand was in registers one iteration
    // before its use.
    f32x4 gat[4];         // one channel half (4 channels) of the four corners: TL, TR, BL, BR
    u32x4 aop[6];         // [term * 2 + mt]; the two output-channel tiles are refilled at different times (below)
    u32x4 bop[2][3];
    auto gather = [&](int t, int half) __attribute__((always_inline)) {
      const int pq = pos[t & 1].q;
      const f32x4* qq = reinterpret_cast<const f32x4*>(wst_win + (pq & ~3));
#ifdef EAVSR_IL_EXP_NO_GATHER
#pragma unroll
      for (int j = 0; j < 4; ++j) gat[j] = f32x4{pos[t & 1].w1, pos[t & 1].w2, (float)pos[t & 1].q, pos[t & 1].w4};
      (void)qq; (void)half;
#else
      // a corner is 8 channels = 2 x 16 bytes; which of the two holds channel half `half` depends on the column (swizzle)
      const int hl = half ^ (pq & 1), hr = half ^ ((pq >> 1) & 1);
      gat[0] = qq[hl];                          // top-left
      gat[1] = qq[2 + hr];                      // top-right
      gat[2] = qq[IPW * IG / 4 + hl];           // bottom-left
      gat[3] = qq[IPW * IG / 4 + 2 + hr];       // bottom-right
#endif
    };
    auto load_a = [&](int t, int mt) __attribute__((always_inline)) {
      const u32x4* ws = wst + t * (3 * 2 * 64);
#pragma unroll
      for (int term = 0; term < 3; ++term) aop[term * 2 + mt] = ws[(term * 2 + mt) * 64];
    };
    // blend + split of channels 2c, 2c+1 of step t
    auto blend_pair = [&](int t, int c) __attribute__((always_inline)) {
      const Pos& ps = pos[t & 1];
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = (2 * c + u) & 3;      // channel within the half that `gat` holds (c < 2: low half, else high half)
        float tv = ps.w1 * gat[0][e];
        tv = __builtin_fmaf(ps.w2, gat[1][e], tv);
        tv = __builtin_fmaf(ps.w3, gat[2][e], tv);
        tv = __builtin_fmaf(ps.w4, gat[3][e], tv);
        v[u] = tv;
      }
      unsigned h2, m2, l2;
#ifdef EAVSR_IL_EXP_NO_SPLIT
      h2 = __float_as_uint(v[0]); m2 = __float_as_uint(v[1]); l2 = h2 ^ m2;
#else
      il_split2(v[0], v[1], h2, m2, l2);
#endif
      bop[t & 1][0][c] = h2; bop[t & 1][1][c] = m2; bop[t & 1][2][c] = l2;
    };
    // partial product i of a k-step (smallest first within each output-channel tile); a = [hi0 hi1 mid0 mid1 lo0 lo1]
    auto mfma_i = [&](int i, const u32x4 (&av)[6], const u32x4 (&b)[3], f32x16 (&ac)[2]) __attribute__((always_inline)) {
#ifdef EAVSR_IL_EXP_NO_MFMA
      ac[0][i & 15] += __uint_as_float(b[i % 3][i & 3] ^ av[i % 6][i & 3]);
#else
      constexpr int TA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, TB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
      constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TB6[6] = {0, 2, 1, 0, 1, 0};
      const int mt = i / NPROD, j = i % NPROD;
      const int ta = NPROD == 9 ? TA9[j] : TA6[j], tb = NPROD == 9 ? TB9[j] : TB6[j];
      ac[mt] = il_mfma(av[ta * 2 + mt], b[tb], ac[mt]);
#endif
    };
    constexpr int NM = 2 * NPROD, CH = NM / 6;      // MFMAs per k-step; per chunk (2 for x6, 3 for x9)
#define IL_FENCE() __builtin_amdgcn_sched_barrier(0)
    setup(0);
    gather(0, 0);
    load_a(0, 0);
    setup(1);
    blend_pair(0, 0);
    blend_pair(0, 1);
    IL_FENCE();
    gather(0, 1);
    IL_FENCE();
    blend_pair(0, 2);
    blend_pair(0, 3);
    IL_FENCE();
    IL_STAMP(5);
#pragma unroll
    for (int t = 0; t < ISTEPS; ++t) {
      // the MFMAs run tile 0 first (chunks 0-2), tile 1 second (chunks 3-5): tile 1's A operands of THIS step are
      // requested now (three chunks ahead), tile 0's of the NEXT step as soon as chunk 2 has issued its last MFMA.
      // chunk:   0: set-up(t+2)   1: next step's window DMA piece + parameter loads   2: pair 0   3: pair 1, then the
      //          gathers of the high channel half + tile 0's next A operands   4: weight DMA piece   5: pairs 2 and 3
      // (every LDS read has at least one whole chunk between its issue and its first use)
      if (t + 1 < ISTEPS) gather(t + 1, 0);
      load_a(t, 1);
      IL_FENCE();
#pragma unroll
      for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int i = 0; i < CH; ++i) mfma_i(k * CH + i, aop, bop[t & 1], acc);
        if (k == 0 && t + 2 < ISTEPS) setup(t + 2);
        // next step's DMA and parameter loads, spread over the iterations (every piece has >= 1 k-step to land)
#ifndef EAVSR_IL_EXP_NO_DMA      // timing ablations (tools/gpu_il_ablate.py): results are wrong by construction
        if (k == 1 && t < WIN_IT && more) issue_win(t, nxorg, ny0, nx0, stage ^ 1);
        if (k == 4 && t < W_IT && more) issue_wgt(t, nwsrc, stage ^ 1);
#endif
        if (k == 1) {
          if (t == 0) { load_slot(lb, lo, 0); load_slot(lb, lo, 1); }
          if (t == 1) { load_slot(lb, lo, 2); load_slot(lb, lo, 3); }
          if (t == 2) { load_slot(lb, lo, 4); load_affine(lb, lo); }
        }
        if (t + 1 < ISTEPS) {
          if (k == 2) blend_pair(t + 1, 0);
          if (k == 3) {
            blend_pair(t + 1, 1);
            gather(t + 1, 1);
            load_a(t + 1, 0);
          }
          if (k == 5) {
            blend_pair(t + 1, 2);
            blend_pair(t + 1, 3);
          }
        }
        IL_FENCE();
      }
    }
#undef IL_FENCE
    IL_STAMP(6);

    // rare: a corner left the LDS window.  Those lanes contributed exactly zero above; their samples are redone from
    // global memory (IL8: one corner = 32 contiguous bytes) with corner-wise zero padding and multiplied in.
#ifdef EAVSR_IL_EXP_NO_FIXUP
    slow_steps = 0;
#endif
    if (__builtin_amdgcn_ballot_w64(slow_steps != 0) != 0) {
      const float* xg = a.xil + ((size_t)bn * ngroups + g) * plane * IG;
      for (int t = 0; t < ISTEPS; ++t) {
        const bool mine = (slow_steps >> t) & 1u;
        if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
        float v[IG];
#pragma unroll
        for (int c = 0; c < IG; ++c) v[c] = 0.f;
        if (mine) {
          const int tap = min(2 * t + kg, IK - 1);
          const float ryt = (float)(tap / 3 - 1), rxt = (float)(tap % 3 - 1);
          // rare path: the parameters are read again (their registers were recycled as the set-ups consumed them)
          const int dgi = g >> a.opg_shift;
          float dy, dx, m;
          if (HEADS) {
            const float* hb = a.offset + (size_t)bn * 15 * a.dg * plane;
            float tt[6];
#pragma unroll
            for (int j = 0; j < 4; ++j) tt[j] = ld_b(hb, ((unsigned)(dgi * 4 + j) * uplane + pix) * 4u);
#pragma unroll
            for (int j = 0; j < 2; ++j) tt[4 + j] = ld_b(hb, ((unsigned)(4 * a.dg + dgi * 2 + j) * uplane + pix) * 4u);
            dy = (tt[0] * ryt + tt[1] * rxt) - ryt + tt[4];
            dx = (tt[2] * ryt + tt[3] * rxt) - rxt + tt[5];
            m = 1.f / (1.f + __expf(-ld_b(hb, ((unsigned)(6 * a.dg + dgi * 9) * uplane + (unsigned)tap * uplane + pix) * 4u)));
          } else {
            const float* offb = a.offset + ((size_t)bn * a.dg + dgi) * 18 * plane;
            const float* mkb = a.mask + ((size_t)bn * a.dg + dgi) * 9 * plane;
            dy = ld_b(offb, (2u * (unsigned)tap * uplane + pix) * 4u);
            dx = ld_b(offb, ((2u * (unsigned)tap + 1u) * uplane + pix) * 4u);
            m = ld_b(mkb, ((unsigned)tap * uplane + pix) * 4u);
          }
          const float py = (fgy + ryt) + dy;
          const float px = (fgx + rxt) + dx;
          if (!(py > -1.f && px > -1.f && py < (float)h && px < (float)w)) m = 0.f;     // validity gate (flagged lanes pass it)
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
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const f32x4* cp = reinterpret_cast<const f32x4*>(xg + ((size_t)cy[k4 >> 1] * w + cx[k4 & 1]) * IG);
            const f32x4 lo4 = cp[0], hi4 = cp[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += cw[k4] * lo4[e];
              v[4 + e] += cw[k4] * hi4[e];
            }
          }
        }
        u32x4 b[3], av[6];
#pragma unroll
        for (int c = 0; c < IG / 2; ++c) {
          unsigned h2, m2, l2;
          il_split2(v[2 * c], v[2 * c + 1], h2, m2, l2);
          b[0][c] = h2; b[1][c] = m2; b[2][c] = l2;
        }
        const u32x4* ws = wst + t * (3 * 2 * 64);
#pragma unroll
        for (int j = 0; j < 6; ++j) av[j] = ws[j * 64];
#pragma unroll
        for (int i = 0; i < 2 * NPROD; ++i) mfma_i(i, av, b, acc);
      }
    }

    // ---- end of a tile: the accumulators leave at the top of the next step (or after the loop) --------------------------
    if (last_g) {
      st_pending = true;
      st_bn = bn; st_gy = gy; st_gx = gx; st_ok = pix_ok;
      bn = nbn; y0 = ny0; x0 = nx0;
      gy = ngy; gx = ngx; pix_ok = npix_ok; pix = npix;
      g = 0;
      ++ti_;
    } else {
      ++g;
    }
  }
  if (st_pending) store_tile();
#ifdef EAVSR_IL_STAMPS
  IL_STAMP(7);
  if (lane == 0 && (wave == 0 || wave == 4)) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_il_stamps[(wave == 4 ? 8 : 0) + i], st_acc[i]);
  }
#endif
}

// (n, c, h, w) fp32 -> IL8 [n][c/8][h][w][8]: one thread per (pixel, octet), 8 coalesced plane reads, two float4 writes
__global__ __launch_bounds__(256) void nchw_to_il8_kernel(const float* __restrict__ x, float* __restrict__ out, int oct,
                                                          int hw) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int o = blockIdx.y, bn = blockIdx.z;
  const float* xp = x + ((size_t)bn * oct + o) * 8 * hw + p;
  f32x4 lo, hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo[e] = xp[(size_t)e * hw];
    hi[e] = xp[(size_t)(4 + e) * hw];
  }
  f32x4* op = reinterpret_cast<f32x4*>(out + (((size_t)bn * oct + o) * hw + p) * 8);
  op[0] = lo;
  op[1] = hi;
}

template <int NPROD, bool HEADS>
int launch_il(const ILArgs& a, dim3 grid, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_il_kernel<NPROD, HEADS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)ILDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2_il: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((dcnv2_il_kernel<NPROD, HEADS>), grid, dim3(512), ILDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2_il");
}

}  // namespace

#ifdef EAVSR_IL_STAMPS
extern "C" int eavsr_debug_il_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_il_stamps), sizeof(g_il_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_il_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int eavsr_nchw_to_il8_f32(const float* x, float* out, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(x && out, -1, "nchw_to_il8: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && c % 8 == 0, -1, "nchw_to_il8: bad dims (c %% 8 == 0 required)");
  EAVSR_REQUIRE((long)h * w < (1L << 28) && c / 8 <= 65535 && n <= 65535, -1, "nchw_to_il8: too large");
  if (n == 0) return 0;
  const int hw = h * w;
  hipLaunchKernelGGL(nchw_to_il8_kernel, dim3(eavsr::cdiv(hw, 256), c / 8, n), dim3(256), 0, eavsr::as_stream(stream), x, out,
                     c / 8, hw);
  return eavsr::launch_status("nchw_to_il8");
}

extern "C" int eavsr_dcnv2_il_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_x9,
                                  const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                  int32_t deform_groups, int32_t nprod, int32_t heads, void* stream) {
  EAVSR_REQUIRE(x_il8 && offset_or_heads && weight_x9 && out && (heads || mask), -1, "dcnv2_il: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2_il: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2_il: cin %d not divisible by deform_groups %d", cin, deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2, "dcnv2_il: %d channels per deformable group unsupported (must be a multiple of 8)", cpg);
  EAVSR_REQUIRE(nprod == 6 || nprod == 9, -2, "dcnv2_il: nprod %d (6 or 9)", nprod);
  EAVSR_REQUIRE((long)h * w * 32 < (1L << 31), -1, "dcnv2_il: plane too large for 32-bit byte offsets");
  EAVSR_REQUIRE((long)h * w * 15 * deform_groups * 4 < (1L << 32) || !heads, -1, "dcnv2_il: heads tensor too large");
  EAVSR_REQUIRE((((uintptr_t)x_il8) & 15) == 0, -2, "dcnv2_il: x must be 16-byte aligned");
  if (n == 0) return 0;
  ILArgs a;
  a.xil = x_il8; a.offset = offset_or_heads; a.mask = mask; a.wsplit = reinterpret_cast<const u32x4*>(weight_x9);
  a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups; a.cpg = cpg;
  {
    const int opg = cpg / 8;
    EAVSR_REQUIRE((opg & (opg - 1)) == 0, -2, "dcnv2_il: %d channels per deformable group: cpg / 8 must be a power of two", cpg);
    a.opg_shift = 0;
    while ((1 << a.opg_shift) < opg) ++a.opg_shift;
  }
  a.tiles_x = eavsr::cdiv(w, IT_W);
  a.tiles_y = eavsr::cdiv(h, IT_ROWS);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "dcnv2_il: too many tiles");
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
  if (nprod == 9) return heads ? launch_il<9, true>(a, grid, st) : launch_il<9, false>(a, grid, st);
  return heads ? launch_il<6, true>(a, grid, st) : launch_il<6, false>(a, grid, st);
}
