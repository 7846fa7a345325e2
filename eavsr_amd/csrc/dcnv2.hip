// Fused modulated deformable convolution (DCNv2) forward (a7).
//
// Reference: mmcv.ops.modulated_deform_conv2d called at models/networks.py:627-630 with the
// ModulatedDeformConv2d parameters of networks.py:575-583 (64 -> 64, 3x3, stride 1, pad 1,
// dilation 1, groups 1, deform_groups 8).  mmcv runs a per-sample im2col into a (cin*9, h*w) column
// buffer in HBM (132.7 MB at 180x320), an SGEMM and a bias pass.  Here the column tile never leaves
// the CU: per 4x32-pixel tile and per 4-channel chunk the sampler writes col[c][tap][px] into LDS and
// the 64 x (4*9) x 128 contraction runs on v_mfma_f32_32x32x2_f32 straight from LDS.
//
// Semantics restated from the published mmcv 1.x algorithm (modulated_deformable_im2col):
//   tap k = 3 i + j samples at p = (y - 1 + i + dy, x - 1 + j + dx), dy = offset[g*18 + 2k],
//   dx = offset[g*18 + 2k + 1], g = c / (cin / dg); the sample is taken iff -1 < p_y < h and
//   -1 < p_x < w and is a corner-wise zero-padded bilinear interpolation
//   (v = hh*hw*v1 + hh*lw*v2 + lh*hw*v3 + lh*lw*v4), col = v * mask[g*9 + k], out = W . col + b.
//
// HBM traffic per pixel (fp32, cin = cout = 64, dg = 8): 64 + 144 + 72 + 64 floats = 1376 B; the input
// patch is re-read through L1/L2 by the 9 taps (data-dependent gathers, neighbouring lanes hit the
// same lines).  The per-(pixel, tap) corner indices / weights are computed once per 4-channel chunk.
#include "common.h"

#include <mutex>
#include <stdlib.h>

namespace {

struct DcnArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* wp;
  const float* bias;
  float* out;
  int n, cin, h, w, cout, dg, cpg, cin_pad, tiles_x, tiles_y;
};

// load through a wave-uniform base + 32-bit BYTE offset: lowers to `global_load_dword v, v_off, s[base]`
// (an element index would have to be shifted in 64 bits, which forces per-lane 64-bit addresses)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float ld_b(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

constexpr int DT_H = 4, DT_W = 32, DT_PX = DT_H * DT_W;  // 128 pixels per workgroup, one row per wave
constexpr int DCK = 4;                                   // channels per chunk (half a deformable group at cpg = 8)
constexpr int DKK = 9;
constexpr int TAPS_PER_THREAD = 5;                       // 9 taps x 128 px over 256 threads: taps h, h+2, .. (h = tid >> 7)

// Small LDS footprint (27.6 KB) and few VGPRs on purpose: 4-5 workgroups per CU, so that the sampler
// (latency-bound gathers) of some workgroups runs under the MFMA contraction of the others.
template <int MT>
__global__ __launch_bounds__(256, 4) void dcnv2_kernel(DcnArgs a) {
  constexpr int CO = 32 * MT;
  constexpr int W4 = DCK * DKK * CO / 4;
  constexpr int W_SEGS = (DCK * DKK * CO + 255) / 256;  // 1 KiB (one wave-level dwordx4 DMA) each
  __shared__ float s_col[DCK * DKK * DT_PX];                            // 18,432 B
  __shared__ __attribute__((aligned(16))) float s_w[W_SEGS * 256];      //  9,216 B (MT = 2)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * DT_H, x0 = tx * DT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const float* bcol = s_col + half * (DKK * DT_PX) + wave * 32 + l31;
  const float* acol = s_w + half * (DKK * CO) + l31;

  // this thread samples pixel p for the taps tap0, tap0 + 2, ...
  const int p = tid & (DT_PX - 1);
  const int tap0 = tid >> 7;
  const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
  const bool pix_ok = gy < h && gx < w;
  const unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;
  const unsigned uplane = (unsigned)plane;

  for (int c0 = 0; c0 < a.cin; c0 += DCK) {
    const int g = c0 / a.cpg;
    // ---- sampler: issue every offset / mask load of this thread first, then the gathers ----------
    float oy[TAPS_PER_THREAD], ox[TAPS_PER_THREAD], mk[TAPS_PER_THREAD];
    // wave-uniform bases + 32-bit lane offsets: `global_load v, v_off, s[base]` (no 64-bit VALU address math)
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const unsigned tap = (unsigned)min(tap0 + 2 * j, DKK - 1);
      oy[j] = ld_b(offb, (2u * tap * uplane + pix) * 4u);
      ox[j] = ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
      mk[j] = ld_b(mkb, (tap * uplane + pix) * 4u);
    }
    const float* xp = a.x + ((size_t)bn * a.cin + c0) * plane;
    __syncthreads();  // the previous chunk's contraction is done with s_col / s_w
    // weight slab of this chunk: LDS-DMA (no VGPRs, no ds_write); lands before the barrier below
    {
      const float* wsrc = a.wp + ((size_t)cot * a.cin_pad + (size_t)c0) * (DKK * CO);
#pragma unroll
      for (int i = 0; i < (W_SEGS + 3) / 4; ++i) {
        const int seg = i * 4 + wave;
        if (seg < W_SEGS) {  // wave-uniform
          const int e4 = min(seg * 64 + lane, W4 - 1);
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)e4 * 4), (lptr_t)(s_w + seg * 256), 16, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const int tap = tap0 + 2 * j;
      const int ti = tap / 3, tj = tap - 3 * ti;
      const float py = (float)(gy - 1 + ti) + oy[j];
      const float px = (float)(gx - 1 + tj) + ox[j];
      const bool in = pix_ok && tap < DKK && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
      const float fy0 = floorf(py), fx0 = floorf(px);
      const float lh = py - fy0, lw = px - fx0;
      const float hh = 1.f - lh, hw = 1.f - lw;
      // clamp before the int conversion so that wild offsets stay defined (they are masked by `in`)
      const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
      const int hh_i = hl + 1, wh_i = wl + 1;
      const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
      const float w1 = (in & t_ok & l_ok) ? hh * hw : 0.f;
      const float w2 = (in & t_ok & r_ok) ? hh * lw : 0.f;
      const float w3 = (in & b_ok & l_ok) ? lh * hw : 0.f;
      const float w4 = (in & b_ok & r_ok) ? lh * lw : 0.f;
      const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
      const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
      const unsigned i1 = (unsigned)(cy0 * w + cx0) * 4u, i2 = (unsigned)(cy0 * w + cx1) * 4u;  // byte offsets
      const unsigned i3 = (unsigned)(cy1 * w + cx0) * 4u, i4 = (unsigned)(cy1 * w + cx1) * 4u;
      float vals[DCK];
#pragma unroll
      for (int c = 0; c < DCK; ++c) {
        const float* q = xp + (size_t)c * plane;  // uniform
        float v = w1 * ld_b(q, i1);
        v += w2 * ld_b(q, i2);
        v += w3 * ld_b(q, i3);
        v += w4 * ld_b(q, i4);
        vals[c] = v * mk[j];
      }
      if (tap < DKK) {
#pragma unroll
        for (int c = 0; c < DCK; ++c) s_col[(c * DKK + tap) * DT_PX + p] = vals[c];
      }
      // bound the gathers in flight (2 taps x 4 channels x 4 corners = 32 per lane): without this the
      // scheduler hoists all 80 gathers of the chunk and the kernel needs > 200 VGPRs
      if (j & 1) __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's weight DMA has landed
    __syncthreads();
    // ---- contraction: 18 k-steps of (channel pair, tap) ------------------------------------------
#pragma unroll
    for (int tap = 0; tap < DKK; ++tap) {
#pragma unroll
      for (int cp = 0; cp < DCK / 2; ++cp) {
        const float b = bcol[(cp * 2 * DKK + tap) * DT_PX];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float av = acol[(cp * 2 * DKK + tap) * CO + m * 32];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[m], 0, 0, 0);
        }
      }
    }
  }

  const int oy_ = y0 + wave, ox_ = x0 + l31;
  if (oy_ < h && ox_ < w) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.cout) {
          const float b = a.bias ? a.bias[co] : 0.f;
          a.out[((size_t)bn * a.cout + co) * plane + (size_t)oy_ * w + ox_] = acc[m][r] + b;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------
// LDS-patch variant, the one the hot path runs (w % 4 == 0, 16-byte aligned input): the (4 + 12) x (32 + 16) input window of the
// 4 channels of a chunk is brought into LDS once by 16-byte LDS-DMA (two stages: the window of chunk
// c+1 lands behind the sampling + contraction of chunk c) and the 9 taps gather from LDS (ds_read2_b32,
// 128 B/clk) instead of from L1 (the gathers through the texture path were the bottleneck of the
// global-gather kernel: ~27 clk per wave-instruction).  Taps that leave the window -- offsets beyond
// ~+-6 px, rare after the flow pre-warp of MultiAdSTN -- fall back to a global gather, lane by lane.
// The window is zero outside the image, which is exactly the corner-wise zero padding of the sampler.
// Offsets / masks of chunk c+1 are prefetched into registers during chunk c.
// ---------------------------------------------------------------------------------------------------
// window pieces outside the image read these zeros: every lane of every DMA is then always issued, which
// keeps the number of outstanding vector-memory operations per wave a compile-time constant (counted vmcnt)
__device__ __attribute__((aligned(16))) float g_dcn_zero[4];

constexpr int PH = 16, PW = 48;           // window rows y0-6 .. y0+9, columns x0-8 .. x0+39
constexpr int PY0 = 6, PX0 = 8;
constexpr int PATCH_F = DCK * PH * PW;    // 3072 floats = 12 KiB = 12 one-KiB DMA pieces
constexpr int PATCH_SEGS = PATCH_F / 256;
constexpr int PATCH_IT = PATCH_SEGS / 4;  // pieces per wave

template <int MT>
__global__ __launch_bounds__(256, 3) void dcnv2_patch_kernel(DcnArgs a) {
  constexpr int CO = 32 * MT;
  constexpr int W4 = DCK * DKK * CO / 4;
  constexpr int W_SEGS = (DCK * DKK * CO + 255) / 256;
  // one LDS array: [col tile][weight slab][window stage 0][window stage 1]
  __shared__ __attribute__((aligned(16))) float smem[DCK * DKK * DT_PX + W_SEGS * 256 + 2 * PATCH_F];
  float* s_col = smem;
  float* s_w = smem + DCK * DKK * DT_PX;
  float* s_patch = s_w + W_SEGS * 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * DT_H, x0 = tx * DT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const float* bcol = s_col + half * (DKK * DT_PX) + wave * 32 + l31;
  const float* acol = s_w + half * (DKK * CO) + l31;

  const int p = tid & (DT_PX - 1);
  const int tap0 = tid >> 7;
  const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
  const bool pix_ok = gy < h && gx < w;
  const unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;

  // zero both window stages once; pieces that lie outside the image are never moved and stay zero
  {
    f32x4* z = reinterpret_cast<f32x4*>(s_patch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * PATCH_F / 4; e += 256) z[e] = zero;
  }
  // per-lane byte offsets of this wave's window pieces within a 4-channel slab (0xFFFFFFFF: outside)
  unsigned voff[PATCH_IT];
#pragma unroll
  for (int i = 0; i < PATCH_IT; ++i) {
    const int e4 = (i * 4 + wave) * 64 + lane;
    const int ci = e4 / (PH * (PW / 4));
    const int rem = e4 - ci * (PH * (PW / 4));
    const int r = rem / (PW / 4), c4 = rem - r * (PW / 4);
    const int yy = y0 - PY0 + r, xx = x0 - PX0 + 4 * c4;
    const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
    voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)yy * w + xx) * 4) : 0xFFFFFFFFu;
  }
  auto issue_patch = [&](int c0, int stage) {
    const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)bn * a.cin + c0) * plane);
#pragma unroll
    for (int i = 0; i < PATCH_IT; ++i) {
      const char* src = voff[i] != 0xFFFFFFFFu ? xb + voff[i] : reinterpret_cast<const char*>(g_dcn_zero);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_patch + stage * PATCH_F + (i * 4 + wave) * 256), 16, 0, 0);
    }
  };
  float oy[TAPS_PER_THREAD], ox[TAPS_PER_THREAD], mk[TAPS_PER_THREAD];
  float oyn[TAPS_PER_THREAD], oxn[TAPS_PER_THREAD], mkn[TAPS_PER_THREAD];
  auto load_offsets = [&](int c0, float* fy, float* fx, float* fm) {
    const int g = c0 / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const unsigned tap = (unsigned)min(tap0 + 2 * j, DKK - 1);
      fy[j] = ld_b(offb, (2u * tap * uplane + pix) * 4u);
      fx[j] = ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
      fm[j] = ld_b(mkb, (tap * uplane + pix) * 4u);
    }
  };

  __syncthreads();  // zero fill done before the first DMA may land
  issue_patch(0, 0);
  load_offsets(0, oy, ox, mk);

  int stage = 0;
  for (int c0 = 0; c0 < a.cin; c0 += DCK, stage ^= 1) {
    // window(c0) has landed (vmcnt + barrier of the previous iteration, or here for the first chunk) and
    // the previous contraction is done with s_col / s_w
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      const char* wsrc = reinterpret_cast<const char*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)c0) * (DKK * CO));
      // exactly W_IT pieces per wave (a surplus piece rewrites the last slab piece with the same bytes)
#pragma unroll
      for (int i = 0; i < (W_SEGS + 3) / 4; ++i) {
        const int seg = min(i * 4 + wave, W_SEGS - 1);
        const unsigned e4 = (unsigned)min(seg * 64 + lane, W4 - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + e4 * 16u), (lptr_t)(s_w + seg * 256), 16, 0, 0);
      }
    }
    const bool more = c0 + DCK < a.cin;
    if (more) {
      // issued AFTER the weights: the wait before the contraction leaves exactly these in flight
      issue_patch(c0 + DCK, stage ^ 1);
      load_offsets(c0 + DCK, oyn, oxn, mkn);
    }
    const float* pst = s_patch + stage * PATCH_F;
    const float* xp = a.x + ((size_t)bn * a.cin + c0) * plane;
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const int tap = tap0 + 2 * j;
      const int ti = tap / 3, tj = tap - 3 * ti;
      const float py = (float)(gy - 1 + ti) + oy[j];
      const float px = (float)(gx - 1 + tj) + ox[j];
      const bool in = pix_ok && tap < DKK && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
      const float fy0 = floorf(py), fx0 = floorf(px);
      const float lh = py - fy0, lw = px - fx0;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
      const float m = in ? mk[j] : 0.f;
      float vals[DCK];
      // window coordinates of the top-left corner
      const int ry = hl - (y0 - PY0), rx = wl - (x0 - PX0);
      const bool in_win = ry >= 0 && ry <= PH - 2 && rx >= 0 && rx <= PW - 2;
      if (in_win || !in) {
        // fast path: the window holds zeros outside the image == corner-wise zero padding
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const float* q = pst + (in ? ry * PW + rx : 0);
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
          float v = w1 * q[c * (PH * PW)];
          v += w2 * q[c * (PH * PW) + 1];
          v += w3 * q[c * (PH * PW) + PW];
          v += w4 * q[c * (PH * PW) + PW + 1];
          vals[c] = v * m;
        }
      } else {
        const int hh_i = hl + 1, wh_i = wl + 1;
        const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
        const float w1 = (t_ok & l_ok) ? hh * hw : 0.f;
        const float w2 = (t_ok & r_ok) ? hh * lw : 0.f;
        const float w3 = (b_ok & l_ok) ? lh * hw : 0.f;
        const float w4 = (b_ok & r_ok) ? lh * lw : 0.f;
        const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
        const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
        const unsigned i1 = (unsigned)(cy0 * w + cx0) * 4u, i2 = (unsigned)(cy0 * w + cx1) * 4u;
        const unsigned i3 = (unsigned)(cy1 * w + cx0) * 4u, i4 = (unsigned)(cy1 * w + cx1) * 4u;
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
          const float* q = xp + (size_t)c * plane;
          float v = w1 * ld_b(q, i1);
          v += w2 * ld_b(q, i2);
          v += w3 * ld_b(q, i3);
          v += w4 * ld_b(q, i4);
          vals[c] = v * m;
        }
      }
      if (tap < DKK) {
#pragma unroll
        for (int c = 0; c < DCK; ++c) s_col[(c * DKK + tap) * DT_PX + p] = vals[c];
      }
    }
    // Only the weight slab is needed now.  vmcnt counts in issue order: leaving the PATCH_IT window pieces and the
    // 3 * TAPS_PER_THREAD offset / mask loads of the NEXT chunk in flight lets their HBM latency run under the
    // contraction instead of stalling every chunk; they are drained at the top of the next iteration.
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PATCH_IT + 3 * TAPS_PER_THREAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int tap = 0; tap < DKK; ++tap) {
#pragma unroll
      for (int cp = 0; cp < DCK / 2; ++cp) {
        const float b = bcol[(cp * 2 * DKK + tap) * DT_PX];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float av = acol[(cp * 2 * DKK + tap) * CO + m * 32];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[m], 0, 0, 0);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < TAPS_PER_THREAD; ++j) {
        oy[j] = oyn[j];
        ox[j] = oxn[j];
        mk[j] = mkn[j];
      }
    }
  }

  const int oy_ = y0 + wave, ox_ = x0 + l31;
  if (oy_ < h && ox_ < w) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.cout) {
          const float b = a.bias ? a.bias[co] : 0.f;
          a.out[((size_t)bn * a.cout + co) * plane + (size_t)oy_ * w + ox_] = acc[m][r] + b;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------
// Wave-specialised variant (EAVSR_DCN_VARIANT=w; an experiment kept for A-B runs, 410 us vs 270 us for the
// LDS-window kernel, so NOT the default): measured on the LDS-window kernel above, the matrix
// pipe is busy 38 % of the time because every wave alternates sample -> barrier -> contract and the phases of
// co-resident workgroups do not interleave (contraction alone 143 us, sampler 77 us, DMA + barriers 110 us
// of a 289 us launch).  Here the two jobs run on DIFFERENT waves of one 512-thread workgroup:
//   waves 4-7 (producers): issue the LDS-DMA of weight slabs and input windows, prefetch offsets / masks,
//                          sample chunk c+1 into col stage (c+1)&1
//   waves 0-3 (consumers): contract chunk c from col stage c&1 on the MFMA pipe, nothing else
// with ONE workgroup barrier per 4-channel chunk.  Producer VALU / LDS work and consumer MFMAs issue from
// different waves of the same SIMD, so they overlap by construction; the chunk period is
// max(sampling, 2 consumer waves x 36 MFMAs) on each SIMD (two workgroups per CU).
// ---------------------------------------------------------------------------------------------------
template <int MT>
struct WsCfg {
  static constexpr int CO = 32 * MT;
  static constexpr int COL_F = DCK * DKK * DT_PX;                 // 4608 floats per col stage
  static constexpr int W_SEGS = (DCK * DKK * CO + 255) / 256;     // weight slab pieces (1 KiB each)
  static constexpr int W_F = W_SEGS * 256;
  static constexpr int LDS_FLOATS = 2 * COL_F + 2 * W_F + 2 * PATCH_F;
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

template <int MT>
__global__ __launch_bounds__(512, 2) void dcnv2_ws_kernel(DcnArgs a) {
  using Cfg = WsCfg<MT>;
  constexpr int CO = Cfg::CO, COL_F = Cfg::COL_F, W_SEGS = Cfg::W_SEGS, W_F = Cfg::W_F;
  constexpr int W4 = DCK * DKK * CO / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_col = smem;                    // [2][COL_F]
  float* s_w = smem + 2 * COL_F;          // [2][W_F]
  float* s_patch = s_w + 2 * W_F;         // [2][PATCH_F]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int pw = wave & 3;                // index within the role
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * DT_H, x0 = tx * DT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;
  const int nchunks = a.cin / DCK;

  {  // zero both window stages once: pieces outside the image are never moved and stay zero
    f32x4* z = reinterpret_cast<f32x4*>(s_patch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * PATCH_F / 4; e += 512) z[e] = zero;
  }

  // ---- producer state ---------------------------------------------------------------------------------
  const int ptid = tid & 255;
  const int p = ptid & (DT_PX - 1);
  const int tap0 = ptid >> 7;
  const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
  const bool pix_ok = gy < h && gx < w;
  const unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;
  unsigned voff[PATCH_IT];
#pragma unroll
  for (int i = 0; i < PATCH_IT; ++i) {
    const int e4 = (i * 4 + pw) * 64 + lane;
    const int ci = e4 / (PH * (PW / 4));
    const int rem = e4 - ci * (PH * (PW / 4));
    const int r = rem / (PW / 4), c4 = rem - r * (PW / 4);
    const int yy = y0 - PY0 + r, xx = x0 - PX0 + 4 * c4;
    const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
    voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)yy * w + xx) * 4) : 0xFFFFFFFFu;
  }
  auto issue_window = [&](int chunk, int stage) {
    const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)bn * a.cin + chunk * DCK) * plane);
#pragma unroll
    for (int i = 0; i < PATCH_IT; ++i)
      if (voff[i] != 0xFFFFFFFFu)
        __builtin_amdgcn_global_load_lds((gptr_t)(xb + voff[i]),
                                         (lptr_t)(s_patch + stage * PATCH_F + (i * 4 + pw) * 256), 16, 0, 0);
  };
  auto issue_weights = [&](int chunk, int stage) {
    const char* wsrc = reinterpret_cast<const char*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)chunk * DCK) * (DKK * CO));
#pragma unroll
    for (int i = 0; i < (W_SEGS + 3) / 4; ++i) {
      const int seg = i * 4 + pw;
      if (seg < W_SEGS) {
        const unsigned e4 = (unsigned)min(seg * 64 + lane, W4 - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + e4 * 16u), (lptr_t)(s_w + stage * W_F + seg * 256), 16, 0, 0);
      }
    }
  };
  float oy[TAPS_PER_THREAD], ox[TAPS_PER_THREAD], mk[TAPS_PER_THREAD];
  float oyn[TAPS_PER_THREAD], oxn[TAPS_PER_THREAD], mkn[TAPS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < TAPS_PER_THREAD; ++j) oyn[j] = oxn[j] = mkn[j] = 0.f;
  auto load_offsets = [&](int chunk, float* fy, float* fx, float* fm) {
    const int g = chunk * DCK / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const unsigned tap = (unsigned)min(tap0 + 2 * j, DKK - 1);
      fy[j] = ld_b(offb, (2u * tap * uplane + pix) * 4u);
      fx[j] = ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
      fm[j] = ld_b(mkb, (tap * uplane + pix) * 4u);
    }
  };
  // sample chunk `chunk` (offsets in fy/fx/fm) from window stage `wst` into col stage `cst`
  auto sample = [&](int chunk, int wst, int cst, const float* fy, const float* fx, const float* fm) {
    const float* pst = s_patch + wst * PATCH_F;
    float* col = s_col + cst * COL_F;
    const float* xp = a.x + ((size_t)bn * a.cin + chunk * DCK) * plane;
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) {
      const int tap = tap0 + 2 * j;
      const int ti = tap / 3, tj = tap - 3 * ti;
      const float py = (float)(gy - 1 + ti) + fy[j];
      const float px = (float)(gx - 1 + tj) + fx[j];
      const bool in = pix_ok && tap < DKK && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
      const float fy0 = floorf(py), fx0 = floorf(px);
      const float lh = py - fy0, lw = px - fx0;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
      const float m = in ? fm[j] : 0.f;
      float vals[DCK];
      const int ry = hl - (y0 - PY0), rx = wl - (x0 - PX0);
      const bool in_win = ry >= 0 && ry <= PH - 2 && rx >= 0 && rx <= PW - 2;
      if (in_win || !in) {
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const float* q = pst + (in ? ry * PW + rx : 0);
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
          float v = w1 * q[c * (PH * PW)];
          v += w2 * q[c * (PH * PW) + 1];
          v += w3 * q[c * (PH * PW) + PW];
          v += w4 * q[c * (PH * PW) + PW + 1];
          vals[c] = v * m;
        }
      } else {
        const int hh_i = hl + 1, wh_i = wl + 1;
        const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
        const float w1 = (t_ok & l_ok) ? hh * hw : 0.f;
        const float w2 = (t_ok & r_ok) ? hh * lw : 0.f;
        const float w3 = (b_ok & l_ok) ? lh * hw : 0.f;
        const float w4 = (b_ok & r_ok) ? lh * lw : 0.f;
        const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
        const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
        const unsigned i1 = (unsigned)(cy0 * w + cx0) * 4u, i2 = (unsigned)(cy0 * w + cx1) * 4u;
        const unsigned i3 = (unsigned)(cy1 * w + cx0) * 4u, i4 = (unsigned)(cy1 * w + cx1) * 4u;
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
          const float* q = xp + (size_t)c * plane;
          float v = w1 * ld_b(q, i1);
          v += w2 * ld_b(q, i2);
          v += w3 * ld_b(q, i3);
          v += w4 * ld_b(q, i4);
          vals[c] = v * m;
        }
      }
      if (tap < DKK) {
#pragma unroll
        for (int c = 0; c < DCK; ++c) col[(c * DKK + tap) * DT_PX + p] = vals[c];
      }
    }
  };

  // After an explicit vmcnt(0) the prefetched offsets ARE in their registers, but hipcc still attributes them to
  // the loads and, with an LDS-DMA in flight, waits vmcnt(0) again at their first use -- which then also waits
  // for everything issued since (a full HBM latency per chunk).  Passing the registers through an empty asm
  // makes the asm their definition.
  auto launder = [&](float* fy, float* fx, float* fm) {
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) asm volatile("" : "+v"(fy[j]), "+v"(fx[j]), "+v"(fm[j]));
  };

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  __syncthreads();  // zero fill complete before any DMA may land
  if (producer) {
    issue_window(0, 0);
    issue_weights(0, 0);
    load_offsets(0, oy, ox, mk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    launder(oy, ox, mk);
  }
  __syncthreads();  // window(0) visible to every producer wave
  if (producer) {
    if (nchunks > 1) {
      issue_window(1, 1);
      load_offsets(1, oyn, oxn, mkn);
    }
    sample(0, 0, 0, oy, ox, mk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    launder(oyn, oxn, mkn);
  }
  __syncthreads();  // col(0), weights(0), window(1) ready

  for (int c = 0; c < nchunks; ++c) {
    if (producer) {
      if (c + 1 < nchunks) {
        // offsets(c+1) are in (oyn, oxn, mkn): rotate them BEFORE anything is issued in this iteration -- hipcc
        // waits vmcnt(0) at the first use of an ordinary load's result while an LDS-DMA is in flight, and right
        // here nothing is (the previous iteration drained the queue), so the wait is free
#pragma unroll
        for (int j = 0; j < TAPS_PER_THREAD; ++j) { oy[j] = oyn[j]; ox[j] = oxn[j]; mk[j] = mkn[j]; }
        asm volatile("" ::: "memory");
        issue_weights(c + 1, (c + 1) & 1);
        if (c + 2 < nchunks) {
          issue_window(c + 2, c & 1);               // stage c&1 was sampled during iteration c-1: free
          load_offsets(c + 2, oyn, oxn, mkn);
        }
        sample(c + 1, (c + 1) & 1, (c + 1) & 1, oy, ox, mk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        launder(oyn, oxn, mkn);
      }
    } else {
      const float* bcol = s_col + (c & 1) * COL_F + half * (DKK * DT_PX) + pw * 32 + l31;
      const float* acol = s_w + (c & 1) * W_F + half * (DKK * CO) + l31;
#pragma unroll
      for (int tap = 0; tap < DKK; ++tap) {
#pragma unroll
        for (int cp = 0; cp < DCK / 2; ++cp) {
          const float b = bcol[(cp * 2 * DKK + tap) * DT_PX];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = acol[(cp * 2 * DKK + tap) * CO + m * 32];
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[m], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  if (!producer) {
    const int oy_ = y0 + pw, ox_ = x0 + l31;
    if (oy_ < h && ox_ < w) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co < a.cout) {
            const float b = a.bias ? a.bias[co] : 0.f;
            a.out[((size_t)bn * a.cout + co) * plane + (size_t)oy_ * w + ox_] = acc[m][r] + b;
          }
        }
    }
  }
}

template <int MT>
int launch_ws(const DcnArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = WsCfg<MT>;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_ws_kernel<MT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(dcnv2_ws_kernel<MT>, grid, dim3(512), Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2");
}

// ---------------------------------------------------------------------------------------------------
// Register-fed variant (EAVSR_DCN_VARIANT=r, 308 us; A-B only): no column tile at all.
// v_mfma_f32_32x32x2_f32 wants B[k = lane >> 5][j = lane & 31]: lane (half, j) therefore samples ITS OWN
// pixel j of channel 2 cp + half and hands the value to the MFMA straight from a VGPR -- the sampler's
// output never touches LDS, and the only barrier is the one per 8-channel group that publishes the next
// LDS window / weight slab (both moved by 16-byte LDS-DMA while the current group computes).
// Workgroup = 512 threads = 8 waves = 8 rows x 32 pixels; per tap a lane computes the sampling position
// once (shared by the 4 channels it owns in the group), gathers 4 x 4 corners from the LDS window
// (ds_read2_b32) and issues 4 x MT MFMAs whose 64-cycle shadows hide the next tap's arithmetic.
// ---------------------------------------------------------------------------------------------------
constexpr int RCK = 8;                        // channels per chunk
constexpr int WW = 48;                        // window columns x0-8 .. x0+39

// R = rows (= waves) per workgroup: R/4 waves per SIMD.  More waves per SIMD hide the sampler's LDS /
// VALU latency under the other waves' MFMAs; the VGPR budget shrinks accordingly (512 / (R/4)).
template <int MT, int R>
struct RegCfg {
  static constexpr int CO = 32 * MT;
  static constexpr int WH = R + 12;                         // window rows y0-6 .. y0+R+5
  static constexpr int WIN_F = RCK * WH * WW;
  static constexpr int WIN_SEGS = (WIN_F + 255) / 256;
  static constexpr int WIN_IT = (WIN_SEGS + R - 1) / R;
  static constexpr int W_F = RCK * DKK * CO;               // weight slab floats
  static constexpr int W_SEGS = (W_F + 255) / 256;
  static constexpr int W_IT = (W_SEGS + R - 1) / R;
  static constexpr int STAGE = WIN_SEGS * 256 + W_SEGS * 256;
  static constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(float);
};

template <int MT, int R, bool PREFETCH>
__global__ __launch_bounds__(64 * R, R / 4) void dcnv2_reg_kernel(DcnArgs a) {
  using Cfg = RegCfg<MT, R>;
  constexpr int CO = Cfg::CO, W_F = Cfg::W_F, W_SEGS = Cfg::W_SEGS, W_IT = Cfg::W_IT, STAGE = Cfg::STAGE;
  constexpr int WH = Cfg::WH, WIN_F = Cfg::WIN_SEGS * 256, WIN_SEGS = Cfg::WIN_SEGS, WIN_IT = Cfg::WIN_IT;
  constexpr int RT_H = R, NTHR = 64 * R;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [stage][window | weights]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * RT_H, x0 = tx * DT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const unsigned uplane = (unsigned)plane;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const int gy = y0 + wave, gx = x0 + l31;
  const bool pix_ok = gy < h && gx < w;
  const unsigned pix = pix_ok ? (unsigned)(gy * w + gx) : 0u;

  {  // zero both stages once: window pieces outside the image are never moved and stay zero
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * STAGE / 4; e += NTHR) z[e] = zero;
  }
  unsigned voff[WIN_IT];
#pragma unroll
  for (int i = 0; i < WIN_IT; ++i) {
    const int seg = i * R + wave;
    const int e4 = seg * 64 + lane;
    const int ci = e4 / (WH * (WW / 4));
    const int rem = e4 - ci * (WH * (WW / 4));
    const int r = rem / (WW / 4), c4 = rem - r * (WW / 4);
    const int yy = y0 - 6 + r, xx = x0 - 8 + 4 * c4;
    const bool ok = seg < WIN_SEGS && e4 < RCK * WH * WW / 4 && yy >= 0 && yy < h && xx >= 0 && xx < w;
    voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)yy * w + xx) * 4) : 0xFFFFFFFFu;
  }
  auto issue_stage = [&](int c0, int stage) {
    float* win = smem + stage * STAGE;
    float* sw = win + WIN_F;
    const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)bn * a.cin + c0) * plane);
#pragma unroll
    for (int i = 0; i < WIN_IT; ++i)
      if (voff[i] != 0xFFFFFFFFu)
        __builtin_amdgcn_global_load_lds((gptr_t)(xb + voff[i]), (lptr_t)(win + (i * R + wave) * 256), 16, 0, 0);
    const char* wsrc = reinterpret_cast<const char*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)c0) * (DKK * CO));
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = i * R + wave;
      if (seg < W_SEGS) {
        const unsigned e4 = (unsigned)min(seg * 64 + lane, W_F / 4 - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + e4 * 16u), (lptr_t)(sw + seg * 256), 16, 0, 0);
      }
    }
  };
  float oy[DKK], ox[DKK], mk[DKK], oyn[PREFETCH ? DKK : 1], oxn[PREFETCH ? DKK : 1], mkn[PREFETCH ? DKK : 1];
  auto load_offsets = [&](int c0, float* fy, float* fx, float* fm) {
    const int g = c0 / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int t = 0; t < DKK; ++t) {
      fy[t] = ld_b(offb, (2u * t * uplane + pix) * 4u);
      fx[t] = ld_b(offb, ((2u * t + 1u) * uplane + pix) * 4u);
      fm[t] = ld_b(mkb, ((unsigned)t * uplane + pix) * 4u);
    }
  };

  __syncthreads();  // zero fill complete before the first DMA may land
  issue_stage(0, 0);
  if (PREFETCH) load_offsets(0, oy, ox, mk);

  int stage = 0;
  for (int c0 = 0; c0 < a.cin; c0 += RCK, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage data + this chunk's offsets have landed
    __syncthreads();                                   // ... for every wave; the other stage is free
    const bool more = c0 + RCK < a.cin;
    if (!PREFETCH) load_offsets(c0, oy, ox, mk);  // issued BEFORE the DMAs: waited for by a counted vmcnt
    if (more) {
      issue_stage(c0 + RCK, stage ^ 1);
      if (PREFETCH) load_offsets(c0 + RCK, oyn, oxn, mkn);  // first used after the next vmcnt(0)
    }
    const float* win = smem + stage * STAGE;
    const float* awt = win + WIN_F + half * (DKK * CO) + l31;
    const float* xg = a.x + ((size_t)bn * a.cin + c0) * plane;
    // Does any lane of this wave have a valid tap whose 2x2 footprint leaves the LDS window?  (rare)
    bool need_fb = false;
#pragma unroll
    for (int tap = 0; tap < DKK; ++tap) {
      const float py = (float)(gy - 1 + tap / 3) + oy[tap];
      const float px = (float)(gx - 1 + tap % 3) + ox[tap];
      const bool in = pix_ok && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
      // window rows y0-6 .. y0+WH-7: floor(py) in [y0-6, y0+WH-8]  <=>  py in [y0-6, y0+WH-7)
      const bool in_win = py >= (float)(y0 - 6) && py < (float)(y0 - 6 + WH - 1) &&
                          px >= (float)(x0 - 8) && px < (float)(x0 - 8 + WW - 1);
      need_fb |= in && !in_win;
    }
    if (!__any(need_fb)) {
      // ---- straight-line fast path: no branch inside, so the scheduler can slide the next tap's address
      // arithmetic and LDS gathers under the current tap's MFMAs
#pragma unroll
      for (int tap = 0; tap < DKK; ++tap) {
        const int ti = tap / 3, tj = tap - 3 * ti;
        const float py = (float)(gy - 1 + ti) + oy[tap];
        const float px = (float)(gx - 1 + tj) + ox[tap];
        const bool in = pix_ok && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const float lh = py - fy0, lw = px - fx0;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float m = in ? mk[tap] : 0.f;
        // clamped window coordinates (every lane that is `in` is inside the window here)
        const int ry = (int)fminf(fmaxf(fy0 - (float)(y0 - 6), 0.f), (float)(WH - 2));
        const int rx = (int)fminf(fmaxf(fx0 - (float)(x0 - 8), 0.f), (float)(WW - 2));
        const float w1 = hh * hw * m, w2 = hh * lw * m, w3 = lh * hw * m, w4 = lh * lw * m;
        const float* q = win + half * (WH * WW) + ry * WW + rx;
        float bval[RCK / 2];
#pragma unroll
        for (int cp = 0; cp < RCK / 2; ++cp) {
          const float* qc = q + cp * 2 * (WH * WW);
          float v = w1 * qc[0];
          v += w2 * qc[1];
          v += w3 * qc[WW];
          v += w4 * qc[WW + 1];
          bval[cp] = v;
        }
#pragma unroll
        for (int cp = 0; cp < RCK / 2; ++cp) {
#pragma unroll
          for (int mm = 0; mm < MT; ++mm) {
            const float av = awt[(cp * 2 * DKK + tap) * CO + mm * 32];
            acc[mm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bval[cp], acc[mm], 0, 0, 0);
          }
        }
      }
    } else {
  #pragma unroll
      for (int tap = 0; tap < DKK; ++tap) {
        const int ti = tap / 3, tj = tap - 3 * ti;
        const float py = (float)(gy - 1 + ti) + oy[tap];
        const float px = (float)(gx - 1 + tj) + ox[tap];
        const bool in = pix_ok && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const float lh = py - fy0, lw = px - fx0;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
        const float m = in ? mk[tap] : 0.f;
        const int ry = hl - (y0 - 6), rx = wl - (x0 - 8);
        const bool in_win = ry >= 0 && ry <= WH - 2 && rx >= 0 && rx <= WW - 2;
        float bval[RCK / 2];
        if (in_win || !in) {
          // the window holds zeros outside the image == the sampler's corner-wise zero padding
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          const float* q = win + half * (WH * WW) + (in ? ry * WW + rx : 0);
  #pragma unroll
          for (int cp = 0; cp < RCK / 2; ++cp) {
            const float* qc = q + cp * 2 * (WH * WW);
            float v = w1 * qc[0];
            v += w2 * qc[1];
            v += w3 * qc[WW];
            v += w4 * qc[WW + 1];
            bval[cp] = v * m;
          }
        } else {
          const int hh_i = hl + 1, wh_i = wl + 1;
          const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
          const float w1 = (t_ok & l_ok) ? hh * hw : 0.f;
          const float w2 = (t_ok & r_ok) ? hh * lw : 0.f;
          const float w3 = (b_ok & l_ok) ? lh * hw : 0.f;
          const float w4 = (b_ok & r_ok) ? lh * lw : 0.f;
          const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
          const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
          const unsigned i1 = (unsigned)(cy0 * w + cx0) * 4u, i2 = (unsigned)(cy0 * w + cx1) * 4u;
          const unsigned i3 = (unsigned)(cy1 * w + cx0) * 4u, i4 = (unsigned)(cy1 * w + cx1) * 4u;
          const unsigned hoff = (unsigned)half * uplane * 4u;
  #pragma unroll
          for (int cp = 0; cp < RCK / 2; ++cp) {
            const float* qc = xg + (size_t)(2 * cp) * plane;  // uniform; + half * plane via the lane offset
            float v = w1 * ld_b(qc, i1 + hoff);
            v += w2 * ld_b(qc, i2 + hoff);
            v += w3 * ld_b(qc, i3 + hoff);
            v += w4 * ld_b(qc, i4 + hoff);
            bval[cp] = v * m;
          }
        }
  #pragma unroll
        for (int cp = 0; cp < RCK / 2; ++cp) {
  #pragma unroll
          for (int mm = 0; mm < MT; ++mm) {
            const float av = awt[(cp * 2 * DKK + tap) * CO + mm * 32];
            acc[mm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bval[cp], acc[mm], 0, 0, 0);
          }
        }
      }
    }
    if (PREFETCH && more) {
#pragma unroll
      for (int t = 0; t < DKK; ++t) {
        oy[t] = oyn[t];
        ox[t] = oxn[t];
        mk[t] = mkn[t];
      }
    }
  }

  if (pix_ok) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.cout) {
          const float b = a.bias ? a.bias[co] : 0.f;
          a.out[((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx] = acc[m][r] + b;
        }
      }
  }
}

template <int MT, int R, bool PREFETCH>
int launch_reg(const DcnArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = RegCfg<MT, R>;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcnv2_reg_kernel<MT, R, PREFETCH>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("dcnv2: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((dcnv2_reg_kernel<MT, R, PREFETCH>), grid, dim3(64 * R), Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("dcnv2");
}

}  // namespace

extern "C" int eavsr_dcnv2_f32(const float* x, const float* offset, const float* mask,
                               const float* weight_packed, const float* bias, float* out, int32_t n,
                               int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t deform_groups,
                               void* stream) {
  EAVSR_REQUIRE(x && offset && mask && weight_packed && out, -1, "dcnv2: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2: cin %d not divisible by deform_groups %d", cin,
                deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % 8 == 0, -2,
                "dcnv2: %d channels per deformable group unsupported (must be a multiple of 8; the reference "
                "uses 64 channels / 8 groups)", cpg);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "dcnv2: plane too large");
  if (n == 0) return 0;
  DcnArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wp = weight_packed; a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups; a.cpg = cpg;
  a.cin_pad = cin;  // multiple of 8 == the 3x3 packing chunk
  a.tiles_x = eavsr::cdiv(w, DT_W);
  a.tiles_y = eavsr::cdiv(h, DT_H);
  const long blocks = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "dcnv2: too many tiles");
  const int CO = cout <= 32 ? 32 : 64;
  dim3 grid((unsigned)blocks, eavsr::cdiv(cout, CO));
  // LDS-window path: whole 16-byte pieces are inside or outside the image, and aligned
  const bool patch = (w % 4) == 0 && (((uintptr_t)x) & 15) == 0;
  hipStream_t st = eavsr::as_stream(stream);
  static const int variant = [] {
    // A-B switch: default = LDS-window kernel; "w" = wave-specialised producers / consumers, "r" = register-fed
    // B operand, "g" = global gathers without an LDS window (the measured ranking is in DESIGN.md)
    const char* e = getenv("EAVSR_DCN_VARIANT");
    return e == nullptr ? 1 : (e[0] == 'r' ? 0 : (e[0] == 'g' ? 2 : (e[0] == 'w' ? 3 : 1)));
  }();
  static const int rows = [] {
    const char* e = getenv("EAVSR_DCN_ROWS");  // A-B: 8, 12 or 16 rows (waves) per workgroup
    return e == nullptr ? 8 : atoi(e);
  }();
  if (patch && variant == 0) {
    const int R = rows == 8 ? 8 : rows == 16 ? 16 : 12;
    a.tiles_y = eavsr::cdiv(h, R);
    const long rblocks = (long)a.tiles_x * a.tiles_y * n;
    dim3 rgrid((unsigned)rblocks, eavsr::cdiv(cout, CO));
    if (R == 8) return CO == 32 ? launch_reg<1, 8, true>(a, rgrid, st) : launch_reg<2, 8, true>(a, rgrid, st);
    if (R == 16) return CO == 32 ? launch_reg<1, 16, false>(a, rgrid, st) : launch_reg<2, 16, false>(a, rgrid, st);
    return CO == 32 ? launch_reg<1, 12, true>(a, rgrid, st) : launch_reg<2, 12, true>(a, rgrid, st);
  }
  if (patch && variant == 3) return CO == 32 ? launch_ws<1>(a, grid, st) : launch_ws<2>(a, grid, st);
  if (patch && variant == 1) {
    if (CO == 32) hipLaunchKernelGGL(dcnv2_patch_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dcnv2_patch_kernel<2>, grid, dim3(256), 0, st, a);
  } else {
    if (CO == 32) hipLaunchKernelGGL(dcnv2_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dcnv2_kernel<2>, grid, dim3(256), 0, st, a);
  }
  return eavsr::launch_status("dcnv2");
}
