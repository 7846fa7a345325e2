// Backward of DCNv2 on the sampler's side (round 6): dx, doffset, dmask, dweight of
//   out = W (64 x 576) . col + b,   col[c*9 + k][p] = bilinear_zero(x[c], p + tap_k + offset_{g(c),k}) * mask_{g(c),k}
// (mmcv.ops.modulated_deform_conv2d as models/networks.py:627-630 calls it, inside loss.backward() of the training step,
// models/eavsrp_model.py:109-119) WITHOUT a column tensor: rounds 1-5 followed the published op's own structure (im2col -> 576-channel
// column tensor in HBM -> two GEMM launches -> col2im with one thread per (n, g, k, pixel) and 12 global float atomics per thread;
// csrc/backward_dcn.hip, kept for the shapes this kernel does not take).  Here ONE kernel does the whole backward of a
// (deformable group, 4 x 16-pixel tile) unit:
//
//   1. dcol(80 x 16 px) = W_g^T (80 x 64) . dY (64 x 16 px) on v_mfma_f32_16x16x4_f32 (exact fp32; 16-row blocks of two taps x 8 channels, 72 rows
//      padded to 80): the column gradient of the group never leaves the accumulators.  Row order and lane layout are chosen so that
//      lane (kq, l15) ends up with dcol of pixel l15, tap 2 mt + (kq >> 1), channels 4 (kq & 1) .. + 3 -- exactly the 16 bytes one
//      bilinear corner of the IL8 input ([n][c/8][h][w][8]) holds for it.
//   2. The sampler: position, validity (-1 < p < size), the four corner weights as in the forward; four 16-byte corner reads; per
//      channel the sample (-> dmask), its position derivative (-> doffset), dcol * mask * corner weight (-> dx).  The two lanes
//      of a (pixel, tap) trade their 4-channel partial sums by one cross-lane add: d_offset / d_mask are reduced over the 8 channels
//      of the group IN THE WAVE.  dx: every wave owns an LDS window (13 rows x 32 columns x 8 channels) around its pixel row and
//      adds its lanes' corner contributions by plain read-add-write (a tag round per block finds lanes whose corners coincide;
//      ds_add_f32 costs ~160 cycles per wave instruction on gfx950 and was 215 of the first version's 265 us).  At the end of the
//      unit the four windows are added where they overlap and stored as the unit's 16 x 32-cell slab; dcn_bwd_dx_gather adds, per
//      dx cell, the (up to eight) slabs that cover it -- no global atomics either (they cost 100 us per call at the rate of the L2's
//      float atomic unit), except for corners outside the window (rare at the alignment's offsets).
//   3. dW_g (64 x 80) += dY (64 x 16 px) . col^T (16 px x 80) on the same matrix instructions, col = sample * mask handed over through a
//      per-wave LDS tile, accumulated in registers over the consecutive units of a workgroup (units are ordered group-major) and
//      written as one partial slab per (workgroup, group); dcn_bwd_dw_reduce adds the slabs in a fixed order (no atomics).
//
// Specialised to the model's configuration (64 -> 64 channels, 8 deformable groups of 8 channels, 3 x 3, stride / pad / dilation 1).
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int B_C = 64, B_CO = 64, B_DG = 8;
constexpr int B_TH = 4, B_TW = 16;              // tile: one 16-pixel row per wave, 4 waves
constexpr int B_MT = 5;                          // 16-row blocks of a group's (tap, channel) rows: 72 -> 80
constexpr int B_WY = 6, B_WX = 8;                // margins of a wave's dx window around its 16-pixel row
constexpr int B_WH = 1 + 2 * B_WY, B_WW = B_TW + 2 * B_WX;      // 13 x 32 cells x 8 channels, one window PER WAVE
constexpr int B_WIN = B_WH * B_WW * 8;           // 3328 floats
constexpr int B_UH = B_TH + 2 * B_WY;            // rows of a unit's window (the four waves' windows, one row apart): 16
constexpr int B_TAG = B_WH * B_WW * 2;           // one byte per (cell, channel half): who adds to it in this round
constexpr int B_LD = 17;                         // row pitch of the per-wave 16-pixel tiles (bank spread)
constexpr int B_COLB = 80 * B_LD, B_DYT = 64 * B_LD;
constexpr int B_SLAB = 64 * 80;                  // one partial dW slab
// The two taps that share a 16-row block (one instruction of the sampler works on both: lanes kq < 2 on the first, kq >= 2 on the
// second): t and t + 5, i.e. (0,5) (1,6) (2,7) (3,8) (4, padding) -- at least one filter row apart, so that their dx contributions of
// neighbouring pixels do not land on the same window cell in the same LDS atomic instruction (taps 2 mt, 2 mt + 1 are horizontal
// neighbours: pixel p's second tap and pixel p + 1's first one hit the same cell, a same-address conflict in every instruction).
#ifdef EAVSR_DCNB_ADJ_TAPS
__host__ __device__ constexpr int b_tap(int mt, int hi) { return 2 * mt + hi; }
#else
__host__ __device__ constexpr int b_tap(int mt, int hi) { return hi ? mt + 5 : mt; }
#endif

struct BwdArgs {
  const float* xil;      // [n][8][h][w][8]
  const float* offset;   // (n, 144, h, w)
  const float* mask;     // (n, 72, h, w)
  const float* dy;       // (n, 64, h, w)
  const float* wt;       // packed W^T: [g][mt][s][lane]
  float* dxil;           // [n][8][h][w][8], pre-zeroed; NULL: no input gradient
  float* doff;           // (n, 144, h, w)
  float* dmask;          // (n, 72, h, w)
  float* slabs;          // [gridDim.x][2][64][80]
  float* dxs;            // [total][16][32][8]: every unit's dx window (dcn_bwd_dx_gather adds them into dxil)
  int n, h, w, tiles_x, tiles_y, ntiles, per, total;
};

// the forward's sampling rule (csrc/backward_dcn.hip make_samp; mmcv 1.x modulated_deform_conv as networks.py:627-630 uses it): validity -1 < p < size, corner-wise
// zero padding, clamped corner indices (every read is in range; validity is applied to the values)
struct BSamp {
  bool in, v1, v2, v3, v4;
  float w1, w2, w3, w4, hh, hw, lh, lw;
  int cy0, cy1, cx0, cx1;
  int hl, wl;      // the top-left corner before clamping (-1 .. size - 1 when `in`)
};
__device__ __forceinline__ BSamp b_samp(float py, float px, int h, int w) {
  BSamp s;
  s.in = py > -1.f && px > -1.f && py < (float)h && px < (float)w;
  const float fy0 = floorf(py), fx0 = floorf(px);
  s.lh = py - fy0; s.lw = px - fx0; s.hh = 1.f - s.lh; s.hw = 1.f - s.lw;
  const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
  const int hh_i = hl + 1, wh_i = wl + 1;
  const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
  s.v1 = s.in && t_ok && l_ok; s.v2 = s.in && t_ok && r_ok; s.v3 = s.in && b_ok && l_ok; s.v4 = s.in && b_ok && r_ok;
  s.w1 = s.v1 ? s.hh * s.hw : 0.f; s.w2 = s.v2 ? s.hh * s.lw : 0.f;
  s.w3 = s.v3 ? s.lh * s.hw : 0.f; s.w4 = s.v4 ? s.lh * s.lw : 0.f;
  s.cy0 = min(max(hl, 0), h - 1); s.cy1 = min(max(hh_i, 0), h - 1);
  s.cx0 = min(max(wl, 0), w - 1); s.cx1 = min(max(wh_i, 0), w - 1);
  s.hl = hl; s.wl = wl;
  return s;
}

// DATA: dx / doffset / dmask (steps 1 and 2); WGT: dweight (step 3, with the samples taken again).  One kernel doing both needs 252
// registers (80 of them the dW accumulators) and leaves no room to keep a sampler's requests in flight: measured 278 us per 2 x 64 x
// 96 x 96 launch, latency-bound at two waves per SIMD.  As two instantiations each keeps its requests one block ahead.
template <bool DATA, bool WGT>
__global__ __launch_bounds__(256, 2) void dcn_bwd_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_win[DATA ? 4 * B_WIN : 8];
  __shared__ unsigned char s_tag[DATA ? 4 * B_TAG : 4];
  __shared__ float s_colb[4][WGT ? B_COLB : 4];      // per wave: col[(tap, channel) row][pixel]; at a slab flush: the 64 x 80 combine buffer
  __shared__ float s_dyt[4][WGT ? B_DYT : 4];        // per wave: dY[co][pixel]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int h = a.h, w = a.w, hw = h * w;
  if constexpr (DATA) {
    for (int i = tid; i < 4 * B_WIN; i += 256) s_win[i] = 0.f;
    __syncthreads();
  }

  const int u0 = (int)blockIdx.x * a.per;
  const int u1 = u0 + a.per < a.total ? u0 + a.per : a.total;
  f32x4 acc2[WGT ? 4 : 1][B_MT];      // dW_g partial of this wave: [co block][(tap, channel) block]
#pragma unroll
  for (int ct = 0; ct < (WGT ? 4 : 1); ++ct)
#pragma unroll
    for (int nt = 0; nt < B_MT; ++nt) acc2[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur_g = -1, slot = 0;

  // the workgroup's dW_g: the four waves' partials added in LDS in wave order, then one slab
  auto flush_slab = [&]() __attribute__((always_inline)) {
    if constexpr (!WGT) return;
    float* R = &s_colb[0][0];      // 4 x 80 x 17 = 5440 >= 64 x 80 floats
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int ct = 0; ct < (WGT ? 4 : 1); ++ct)
#pragma unroll
          for (int nt = 0; nt < B_MT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* q = R + (16 * ct + 4 * kq + r) * 80 + 16 * nt + l15;
              *q = (wv == 0 ? 0.f : *q) + acc2[ct][nt][r];
            }
      }
      __syncthreads();
    }
    float* dst = a.slabs + ((size_t)blockIdx.x * 2 + slot) * B_SLAB;
    for (int i = tid; i < B_SLAB; i += 256) dst[i] = R[i];
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < (WGT ? 4 : 1); ++ct)
#pragma unroll
      for (int nt = 0; nt < B_MT; ++nt) acc2[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  for (int u = u0; u < u1; ++u) {
    const int g = u / a.ntiles;
    if (g != cur_g) {
      if (cur_g >= 0) {
        flush_slab();
        slot = 1;
      }
      cur_g = g;
    }
    int t = u - g * a.ntiles;
    const int per_img = a.tiles_x * a.tiles_y;
    const int bn = t / per_img;
    t -= bn * per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * B_TH, x0 = tx * B_TW;
    const int gy = y0 + wave, gx = x0 + l15;
    const bool pvalid = gy < h && gx < w;
    const int p = (pvalid ? gy : 0) * w + (pvalid ? gx : 0);

    // ---- dY of this wave's 16 pixels: B operand of the dcol GEMM (k = co = 4 s + kq, column = pixel l15); also [co][pixel] in LDS
    float b1[16];
    {
      const float* dyp = a.dy + (size_t)bn * B_CO * hw + p;
#pragma unroll
      for (int s = 0; s < 16; ++s) b1[s] = pvalid ? dyp[(size_t)(4 * s + kq) * hw] : 0.f;
      if constexpr (WGT) {
#pragma unroll
        for (int s = 0; s < 16; ++s) s_dyt[wave][(4 * s + kq) * B_LD + l15] = b1[s];
      }
    }
    // ---- dcol = W_g^T . dY: five independent accumulation chains over the 16 k-steps
    f32x4 dcol[B_MT];
#pragma unroll
    for (int mt = 0; mt < B_MT; ++mt) dcol[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef EAVSR_DCNB_EXP_NO_DCOL
    if constexpr (DATA) {
      const float* wp = a.wt + (size_t)g * B_MT * 16 * 64 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float a1[B_MT];
#pragma unroll
        for (int mt = 0; mt < B_MT; ++mt) a1[mt] = wp[(mt * 16 + s) * 64];
#pragma unroll
        for (int mt = 0; mt < B_MT; ++mt) dcol[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b1[s], dcol[mt], 0, 0, 0);
      }
    }
#else
#pragma unroll
    for (int mt = 0; mt < B_MT; ++mt) dcol[mt] = f32x4{b1[mt], b1[mt + 5], b1[mt + 10], b1[3]};
#endif
    // ---- the sampler: lane = (pixel l15, tap 2 mt + (kq >> 1), channels 4 (kq & 1) .. + 3).  Every request is made as early as its
    // address is known: the five blocks' offsets / masks at the top (one round trip instead of five), the four corners of block
    // mt + 1 before block mt is worked on (the dx atomics in between may alias as far as the compiler knows: it keeps program order)
    const int wy0 = y0 - B_WY, wx0 = x0 - B_WX;
    const int ch0 = 4 * (kq & 1);
    float p_oy[B_MT], p_ox[B_MT], p_m[B_MT];
#pragma unroll
    for (int mt = 0; mt < B_MT; ++mt) {
      const int tap = b_tap(mt, kq >> 1), tapc = tap < 9 ? tap : 0;
      const size_t oi = ((size_t)(bn * B_DG + g) * 18 + 2 * tapc) * hw + p;
      p_oy[mt] = a.offset[oi];
      p_ox[mt] = a.offset[oi + hw];
      p_m[mt] = a.mask[((size_t)(bn * B_DG + g) * 9 + tapc) * hw + p];
    }
    const float* xg = a.xil + ((size_t)(bn * B_DG + g) * hw) * 8 + ch0;
    // (the sampling rule is evaluated twice per block -- for the request and again at the use -- instead of kept: its 17 registers
    // in flight beside the 16 of the corners spilled the accumulators)
    f32x4 cn[4];
    auto samp_of = [&](int mt) __attribute__((always_inline)) {
      const int tap = b_tap(mt, kq >> 1), tapc = tap < 9 ? tap : 0;
      float oy = p_oy[mt], ox = p_ox[mt];
      asm volatile("" : "+v"(oy), "+v"(ox));      // (opaque copies: the two evaluations must not be merged into one kept result)
      return b_samp((float)(gy - 1 + tapc / 3) + oy, (float)(gx - 1 + tapc % 3) + ox, h, w);
    };
    auto request = [&](int mt) __attribute__((always_inline)) {
      const BSamp sq = samp_of(mt);
      cn[0] = *reinterpret_cast<const f32x4*>(xg + (size_t)(sq.cy0 * w + sq.cx0) * 8);
      cn[1] = *reinterpret_cast<const f32x4*>(xg + (size_t)(sq.cy0 * w + sq.cx1) * 8);
      cn[2] = *reinterpret_cast<const f32x4*>(xg + (size_t)(sq.cy1 * w + sq.cx0) * 8);
      cn[3] = *reinterpret_cast<const f32x4*>(xg + (size_t)(sq.cy1 * w + sq.cx1) * 8);
    };
    request(0);
#pragma unroll
    for (int mt = 0; mt < B_MT; ++mt) {
      const int tap = b_tap(mt, kq >> 1);
      const bool tv = tap < 9 && pvalid;
      const f32x4 a1 = cn[0], a2 = cn[1], a3 = cn[2], a4 = cn[3];
      if (mt + 1 < B_MT) request(mt + 1);
      const BSamp sp = samp_of(mt);
      float col4[4] = {0.f, 0.f, 0.f, 0.f};
      float gm = 0.f, gpy = 0.f, gpx = 0.f;
      const bool in = tv && sp.in;
      const size_t oi = ((size_t)(bn * B_DG + g) * 18 + 2 * (tap < 9 ? tap : 0)) * hw + p;
      const size_t mi = ((size_t)(bn * B_DG + g) * 9 + (tap < 9 ? tap : 0)) * hw + p;
      bool part = false;      // this lane has something to add to dx
      f32x4 dxv[4];           // [corner]: the four channels' contributions
#pragma unroll
      for (int k = 0; k < 4; ++k) dxv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tv) {
        const float m = p_m[mt];
        float dxc[4][4];      // [corner][channel]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float q1 = sp.v1 ? a1[c] : 0.f, q2 = sp.v2 ? a2[c] : 0.f, q3 = sp.v3 ? a3[c] : 0.f, q4 = sp.v4 ? a4[c] : 0.f;
          const float val = sp.hh * sp.hw * q1 + sp.hh * sp.lw * q2 + sp.lh * sp.hw * q3 + sp.lh * sp.lw * q4;
          col4[c] = val * m;
          if constexpr (DATA) {
            const float dc = dcol[mt][c];
            gm += dc * val;
            const float dv = dc * m;
            gpy += dv * ((q3 - q1) * sp.hw + (q4 - q2) * sp.lw);
            gpx += dv * ((q2 - q1) * sp.hh + (q4 - q3) * sp.lh);
            dxc[0][c] = dv * sp.w1; dxc[1][c] = dv * sp.w2; dxc[2][c] = dv * sp.w3; dxc[3][c] = dv * sp.w4;
          }
        }
        if constexpr (DATA) {
          part = a.dxil != nullptr && sp.in;
#pragma unroll
          for (int k = 0; k < 4; ++k) dxv[k] = f32x4{dxc[k][0], dxc[k][1], dxc[k][2], dxc[k][3]};
        }
      }
      // ---- dx: the lane's four corners (by, bx) .. (by + 1, bx + 1) are added to the wave's own window without atomics (an LDS float
      // atomic costs ~160 cycles per wave instruction on gfx950, sixteen of them per block: 215 of the kernel's 265 us).  A wave's LDS
      // instructions execute in order, so a read-add-write is safe against every OTHER instruction of the wave; inside one
      // instruction two lanes must not add to the same cell.  Valid corners are never clamped, so two lanes' k-th corners coincide
      // exactly when their top-left corners do: one tag round per block finds those lanes (each writes its number at its top-left
      // cell and reads it back; who reads another number waits for the next round).  An invalid corner carries a zero (weight 0).
      if constexpr (DATA) {
        const int by = sp.hl - (gy - B_WY), bx = sp.wl - (x0 - B_WX);
        const bool fast = part && by >= 0 && by + 1 < B_WH && bx >= 0 && bx + 1 < B_WW;
        if (part && !fast) {      // outside the window: straight to memory
          const bool cv[4] = {sp.v1, sp.v2, sp.v3, sp.v4};
          const int cy[4] = {sp.cy0, sp.cy0, sp.cy1, sp.cy1}, cx[4] = {sp.cx0, sp.cx1, sp.cx0, sp.cx1};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (cv[k]) {
              float* q = a.dxil + ((size_t)(bn * B_DG + g) * hw + (size_t)cy[k] * w + cx[k]) * 8 + ch0;
#pragma unroll
              for (int c = 0; c < 4; ++c) atomicAdd(q + c, dxv[k][c]);
            }
        }
        const int cell = fast ? by * B_WW + bx : 0;
        volatile unsigned char* tg = s_tag + wave * B_TAG + cell * 2 + (kq & 1);
        float* q0 = s_win + wave * B_WIN + cell * 8 + ch0;
        bool pend = fast;
        while (__builtin_amdgcn_ballot_w64(pend) != 0) {
          if (pend) *tg = (unsigned char)lane;
          const bool win = pend && *tg == (unsigned char)lane;
          if (win) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              f32x4* q = reinterpret_cast<f32x4*>(q0 + ((k >> 1) * B_WW + (k & 1)) * 8);
              const f32x4 o = *q;
              *q = o + dxv[k];
              asm volatile("" ::: "memory");      // corner k + 1 of one lane is corner k of its neighbour: program order
            }
          }
          pend = pend && !win;
        }
      }
      // the two lanes of a (pixel, tap) hold four channels each: one cross-lane add gives both the sums over the group's 8 channels
      if constexpr (DATA) {
        gm += __shfl_xor(gm, 16);
        gpy += __shfl_xor(gpy, 16);
        gpx += __shfl_xor(gpx, 16);
        if (tv && (kq & 1) == 0) {
          a.doff[oi] = in ? gpy : 0.f;
          a.doff[oi + hw] = in ? gpx : 0.f;
          a.dmask[mi] = gm;
        }
      }
      if constexpr (WGT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_colb[wave][(16 * mt + 4 * kq + c) * B_LD + l15] = col4[c];
      }
    }
    if constexpr (WGT) __syncthreads();      // the col / dY tiles of every wave are in LDS
    // ---- dW_g += dY . col^T over this wave's 16 pixels (k = pixel = 4 s + kq)
    if constexpr (WGT) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float a2[4], b2[B_MT];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a2[ct] = s_dyt[wave][(16 * ct + l15) * B_LD + 4 * s + kq];
#pragma unroll
        for (int nt = 0; nt < B_MT; ++nt) b2[nt] = s_colb[wave][(16 * nt + l15) * B_LD + 4 * s + kq];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int nt = 0; nt < B_MT; ++nt) acc2[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[ct], b2[nt], acc2[ct][nt], 0, 0, 0);
      }
    }
    // ---- the unit's dx window (rows y0 - 6 .. y0 + 9, columns x0 - 8 .. x0 + 23): the four waves' windows added where they overlap
    // and stored -- plain, coalesced -- as the unit's slab; the windows are zero again for the next unit.  (Adding the windows to dx
    // with global atomics instead, one per touched cell and channel: 100 us of a 217 us call at 2 x 64 x 96 x 96, the rate of the
    // L2's float atomics.)  dcn_bwd_dx_gather adds, for every dx cell, the eight slabs that cover it.
    if (DATA && a.dxil != nullptr) {
      __syncthreads();
      float* dst = a.dxs + (size_t)u * (B_UH * B_WW * 8);
#pragma unroll
      for (int j = 0; j < B_UH / 4; ++j) {
        const int r = wave + 4 * j;      // window row of this wave's 64 (column, channel half) entries
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          const int pr = r - wv;
          if (pr >= 0 && pr < B_WH) {
            f32x4* q = reinterpret_cast<f32x4*>(s_win + wv * B_WIN + pr * (B_WW * 8) + lane * 4);
            v += *q;
            *q = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        *reinterpret_cast<f32x4*>(dst + (r * (B_WW * 2) + lane) * 4) = v;
      }
      __syncthreads();
    }
    if constexpr (WGT) __syncthreads();
  }
  if (cur_g >= 0) flush_slab();
}

// W (64, 64, 3, 3) -> the A operands of the dcol GEMM: wt[((g * 5 + mt) * 16 + s) * 64 + lane] = W[co = 4 s + kq][g * 8 + m % 8][m / 8],
// m = 16 mt + l15 (zero for the padding rows m >= 72)
__global__ void dcn_bwd_pack_kernel(const float* __restrict__ wgt, float* __restrict__ wt) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B_DG * B_MT * 16 * 64) return;
  const int lane = e & 63, s = (e >> 6) & 15, mt = (e >> 10) % B_MT, g = e / (B_MT * 16 * 64);
  const int l15 = lane & 15, kq = lane >> 4;
  const int tap = b_tap(mt, l15 >> 3), c = l15 & 7, co = 4 * s + kq;      // row l15 of block mt: (tap of the half, channel)
  wt[e] = tap < 9 ? wgt[((size_t)co * B_C + g * 8 + c) * 9 + tap] : 0.f;
}

// dweight[co][g * 8 + c][tap] (+)= sum over the workgroups that worked on group g, in increasing workgroup order, of their slab
__global__ __launch_bounds__(256) void dcn_bwd_dw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int grid, int per,
                                                               int ntiles, int total, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;      // (g, co, col < 80)
  if (e >= B_DG * 64 * 80) return;
  const int col = e % 80, co = (e / 80) & 63, g = e / (80 * 64);
  const int tap = b_tap(col >> 4, (col >> 3) & 1);
  if (tap >= 9) return;      // (padding rows)
  // units of group g: [g ntiles, (g + 1) ntiles) -> workgroups b_lo .. b_hi (units are dealt in runs of `per`); a workgroup's FIRST
  // group is in slot 0, its second in slot 1
  const int b_lo = (g * ntiles) / per, b_hi = ((g + 1) * ntiles - 1) / per;
  const float* base = slabs + co * 80 + col;
  float v0 = 0.f, v1 = 0.f;
  for (int b = b_lo; b <= b_hi; b += 8) {      // eight independent requests at a time, added in workgroup order
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bb = b + j;
      t[j] = 0.f;
      if (bb <= b_hi) t[j] = base[((size_t)bb * 2 + ((bb * per) / ntiles == g ? 0 : 1)) * B_SLAB];
    }
    v0 += (t[0] + t[1]) + (t[2] + t[3]);
    v1 += (t[4] + t[5]) + (t[6] + t[7]);
  }
  const float v = v0 + v1;
  float* q = dw + ((size_t)co * B_C + g * 8 + (col & 7)) * 9 + tap;
  *q = accumulate ? *q + v : v;
}

// dx[bn][g][y][x][8] += the slabs of the (up to 4 x 2) units of group g whose window holds (y, x), in increasing unit order
__global__ __launch_bounds__(256) void dcn_bwd_dx_gather_kernel(const float* __restrict__ dxs, float* __restrict__ dxil, int n, int h, int w,
                                                                int tiles_x, int tiles_y, int ntiles) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (bn, g, y, x, channel half)
  if (idx >= (long)n * B_DG * h * w * 2) return;
  const int half = (int)(idx & 1);
  const long cell = idx >> 1;
  const int x = (int)(cell % w), y = (int)((cell / w) % h), bg = (int)(cell / ((long)h * w));
  const int bn = bg / B_DG, g = bg - bn * B_DG;
  const int ty_lo = max((y - B_WY) >> 2, 0), ty_hi = min((y + B_WY) >> 2, tiles_y - 1);
  const int tx_lo = max((x - B_WX) >> 4, 0), tx_hi = min((x + B_WX) >> 4, tiles_x - 1);
  f32x4 t[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ty = ty_lo + i, tx = tx_lo + j;
      t[2 * i + j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ty <= ty_hi && tx <= tx_hi) {
        const long u = (long)g * ntiles + ((long)bn * tiles_y + ty) * tiles_x + tx;
        const int r = y - B_TH * ty + B_WY, c = x - B_TW * tx + B_WX;
        t[2 * i + j] = *reinterpret_cast<const f32x4*>(dxs + u * (B_UH * B_WW * 8) + ((r * B_WW + c) * 2 + half) * 4);
      }
    }
  f32x4* q = reinterpret_cast<f32x4*>(dxil + idx * 4);
  f32x4 v = *q;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += t[k];
  *q = v;
}

// dx in the IL8 layout the kernel accumulates in -> NCHW
__global__ __launch_bounds__(256) void il8_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int c8, int hw) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int g = blockIdx.y, bn = blockIdx.z;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + (((size_t)bn * c8 + g) * hw + p) * 8);
  const f32x4 v1 = *reinterpret_cast<const f32x4*>(in + (((size_t)bn * c8 + g) * hw + p) * 8 + 4);
  float* o = out + ((size_t)bn * c8 * 8 + g * 8) * hw + p;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o[(size_t)c * hw] = v0[c];
    o[(size_t)(4 + c) * hw] = v1[c];
  }
}

}  // namespace

extern "C" int32_t eavsr_dcnv2_bwd_grid(int32_t n, int32_t h, int32_t w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  const long units = (long)B_DG * n * eavsr::cdiv(h, B_TH) * eavsr::cdiv(w, B_TW);
  return (int32_t)(units < 512 ? units : 512);
}
extern "C" int64_t eavsr_dcnv2_bwd_workspace_floats(int32_t n, int32_t h, int32_t w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  const int64_t units = (int64_t)B_DG * n * eavsr::cdiv(h, B_TH) * eavsr::cdiv(w, B_TW);
  return (int64_t)eavsr_dcnv2_bwd_grid(n, h, w) * 2 * B_SLAB + (int64_t)B_DG * B_MT * 16 * 64 + units * (B_UH * B_WW * 8);
}

extern "C" int eavsr_dcnv2_bwd_f32(const float* x_il8, const float* offset, const float* mask, const float* weight, const float* dy,
                                   float* dx_il8, float* doffset, float* dmask, float* dweight, float* workspace, int32_t n, int32_t cin,
                                   int32_t h, int32_t w, int32_t cout, int32_t deform_groups, int32_t accumulate_dw, void* stream) {
  EAVSR_REQUIRE(x_il8 && offset && mask && weight && dy && doffset && dmask && dweight && workspace, -1, "dcnv2_bwd: NULL pointer");
  EAVSR_REQUIRE(cin == B_C && cout == B_CO && deform_groups == B_DG, -2,
                "dcnv2_bwd: 64 -> 64 channels in 8 deformable groups (got %d -> %d, %d groups): use eavsr_dcnv2_im2col / col2im", cin, cout,
                deform_groups);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0 && (long)n * h * w * 64 < (1L << 31), -1, "dcnv2_bwd: bad dims");
  EAVSR_REQUIRE(((((uintptr_t)x_il8) | ((uintptr_t)dx_il8) | ((uintptr_t)workspace)) & 15) == 0, -1,
                "dcnv2_bwd: x_il8, dx_il8 and workspace must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t st = eavsr::as_stream(stream);
  BwdArgs a;
  const int grid = eavsr_dcnv2_bwd_grid(n, h, w);
  a.slabs = workspace;
  float* wt = workspace + (size_t)grid * 2 * B_SLAB;
  a.xil = x_il8; a.offset = offset; a.mask = mask; a.dy = dy; a.wt = wt;
  a.dxil = dx_il8; a.doff = doffset; a.dmask = dmask;
  a.dxs = wt + B_DG * B_MT * 16 * 64;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, B_TW);
  a.tiles_y = eavsr::cdiv(h, B_TH);
  a.ntiles = n * a.tiles_x * a.tiles_y;
  a.total = B_DG * a.ntiles;
  a.per = eavsr::cdiv(a.total, grid);
  EAVSR_REQUIRE(a.per <= a.ntiles, -1, "dcnv2_bwd: internal: a workgroup would span more than two groups");
  hipLaunchKernelGGL(dcn_bwd_pack_kernel, dim3(eavsr::cdiv(B_DG * B_MT * 16 * 64, 256)), dim3(256), 0, st, weight, wt);
  hipLaunchKernelGGL((dcn_bwd_kernel<true, false>), dim3(grid), dim3(256), 0, st, a);
  if (dx_il8 != nullptr)
    hipLaunchKernelGGL(dcn_bwd_dx_gather_kernel, dim3((unsigned)(((long)n * B_DG * h * w * 2 + 255) / 256)), dim3(256), 0, st, a.dxs, dx_il8,
                       n, h, w, a.tiles_x, a.tiles_y, a.ntiles);
  hipLaunchKernelGGL((dcn_bwd_kernel<false, true>), dim3(grid), dim3(256), 0, st, a);
  hipLaunchKernelGGL(dcn_bwd_dw_reduce_kernel, dim3(eavsr::cdiv(B_DG * 64 * 80, 256)), dim3(256), 0, st, a.slabs, dweight, grid, a.per,
                     a.ntiles, a.total, accumulate_dw);
  return eavsr::launch_status("dcnv2_bwd");
}

extern "C" int eavsr_il8_to_nchw_f32(const float* x_il8, float* out, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(x_il8 && out, -1, "il8_to_nchw: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0 && n <= 65535 && c / 8 <= 65535, -1, "il8_to_nchw: bad dims (c %% 8 == 0)");
  EAVSR_REQUIRE((((uintptr_t)x_il8) & 15) == 0, -1, "il8_to_nchw: x_il8 must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(il8_to_nchw_kernel, dim3(eavsr::cdiv(h * w, 256), c / 8, n), dim3(256), 0, eavsr::as_stream(stream), x_il8, out, c / 8,
                     h * w);
  return eavsr::launch_status("il8_to_nchw");
}
