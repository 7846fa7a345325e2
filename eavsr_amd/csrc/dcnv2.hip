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
// LDS-window kernel, the one the hot path runs (w % 4 == 0, 16-byte aligned input): the (4 + 12) x (32 + 16)
// input window of the 4 channels of a chunk is brought into LDS once by 16-byte LDS-DMA (two stages: the window
// of chunk c+1 lands behind the sampling + contraction of chunk c) and the 9 taps gather from LDS (ds_read2_b32)
// instead of from L1 (the gathers through the texture path were the bottleneck of the global-gather kernel:
// ~27 clk per wave-instruction).  Taps that leave the window -- offsets beyond ~+-6 px, rare after the flow
// pre-warp of MultiAdSTN -- are redone from global memory.  The window is zero outside the image, which is
// exactly the corner-wise zero padding of the sampler.
// ---------------------------------------------------------------------------------------------------
// window pieces outside the image read these zeros: every lane of every DMA is then always issued, which
// keeps the number of outstanding vector-memory operations per wave a compile-time constant (counted vmcnt)
__device__ __attribute__((aligned(16))) float g_dcn_zero[4];

constexpr int PH = 16, PW = 48;           // window rows y0-6 .. y0+9, columns x0-8 .. x0+39
constexpr int PY0 = 6, PX0 = 8;
constexpr int PATCH_F = DCK * PH * PW;    // 3072 floats = 12 KiB = 12 one-KiB DMA pieces
constexpr int PATCH_SEGS = PATCH_F / 256;
constexpr int PATCH_IT = PATCH_SEGS / 4;  // pieces per wave

// What bounds it.  v_mfma_f32_32x32x2_f32 runs at exactly the packed-fp32 vector rate and, measured with
// in-kernel cycle stamps on four differently structured kernels (tools/gpu_dcn_stamps.py; the software-
// pipelined, wave-specialised and ping-pong variants are described in DESIGN.md), a wave's vector instructions
// do not issue while another wave of its SIMD streams fp32 MFMAs: matrix time and vector time ADD per SIMD
// instead of overlapping.  The lever is therefore the vector instruction count of the sampler, not the overlap
// structure (409 -> 172 vector instructions per wave and chunk against 36 MFMAs):
//   * sampling positions, bilinear weights (mask folded in) and the LDS corner address are computed once per
//     deformable group (8 channels = 2 chunks) and kept in registers for the second chunk; offsets / masks are
//     loaded once per group, straight into the registers the previous group no longer needs
//   * the main sampling path is branch-free (taps that leave the LDS window are flagged and redone from global
//     memory in a wave-uniform fix-up branch); waves 0-1 own the 5 even taps, waves 2-3 the 4 odd taps (no
//     dummy slot)
//   * the blend uses packed fp32: (w1, w2) * (a, b) + (w3, w4) * (c, d) then one add -- 3 instructions per
//     column value
//   * column tile, weight slab and window are three distinct LDS objects, so the compiler may move gathers
//     across column stores; 52 KB LDS and <= 168 VGPRs keep three workgroups per CU
// ---------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef EAVSR_DCN_STAMPS
// diagnostic build only (tools/build_dcn_diag.sh): cycles of wave 0 / wave 2 of every workgroup per phase
__device__ unsigned long long g_dcn_stamps[16];
#define DCN_STAMP(i)                                                  \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define DCN_STAMP(i) do { } while (0)
#endif
template <bool V> struct BoolTag { static constexpr bool value = V; };
template <int V> struct IntTag { static constexpr int value = V; };

template <int MT>
__global__ __launch_bounds__(256, 3) void dcnv2_grp_kernel(DcnArgs a) {
  constexpr int CO = 32 * MT;
  constexpr int W4 = DCK * DKK * CO / 4;
  constexpr int W_SEGS = (DCK * DKK * CO + 255) / 256;
  constexpr int W_IT = (W_SEGS + 3) / 4;
  // three distinct LDS objects: the compiler may then move window gathers across column-tile stores
  __shared__ float s_col[DCK * DKK * DT_PX];
  __shared__ __attribute__((aligned(16))) float s_w[W_SEGS * 256];
  __shared__ __attribute__((aligned(16))) float s_patch[2 * PATCH_F];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool even_taps = wave < 2;
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

  {
    f32x4* z = reinterpret_cast<f32x4*>(s_patch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * PATCH_F / 4; e += 256) z[e] = zero;
  }
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
  auto issue_window = [&](int chunk, int stage) __attribute__((always_inline)) {
    const char* xb = reinterpret_cast<const char*>(a.x + ((size_t)bn * a.cin + chunk * DCK) * plane);
#pragma unroll
    for (int i = 0; i < PATCH_IT; ++i) {
      const char* src = voff[i] != 0xFFFFFFFFu ? xb + voff[i] : reinterpret_cast<const char*>(g_dcn_zero);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_patch + stage * PATCH_F + (i * 4 + wave) * 256), 16, 0, 0);
    }
  };
  auto issue_weights = [&](int chunk) __attribute__((always_inline)) {
    const char* wsrc = reinterpret_cast<const char*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)chunk * DCK) * (DKK * CO));
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = min(i * 4 + wave, W_SEGS - 1);
      const unsigned e4 = (unsigned)min(seg * 64 + lane, W4 - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + e4 * 16u), (lptr_t)(s_w + seg * 256), 16, 0, 0);
    }
  };
  constexpr int MAXS = 5;
  // offsets / masks of the NEXT deformable group (dead once setup() has turned them into weights and addresses,
  // so the following group's values are loaded straight into them during the group's second chunk)
  float oy[MAXS], ox[MAXS], mk[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; ++j) oy[j] = ox[j] = mk[j] = 0.f;
  auto load_offsets = [&](int chunk, float* fy, float* fx, float* fm, auto ns_tag) __attribute__((always_inline)) {
    constexpr int NS = decltype(ns_tag)::value;
    const int g = chunk * DCK / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const unsigned tap = (unsigned)(tap0 + 2 * j);
      fy[j] = ld_b(offb, (2u * tap * uplane + pix) * 4u);
      fx[j] = ld_b(offb, ((2u * tap + 1u) * uplane + pix) * 4u);
      fm[j] = ld_b(mkb, (tap * uplane + pix) * 4u);
    }
  };
  // per-slot sampling state of the current deformable group (positions are shared by its 8 channels)
  f32x2 sw[MAXS][2];      // (w1, w2), (w3, w4): bilinear corner weights x mask (0 when outside the image)
  int sq[MAXS];           // float offset of the top-left corner in a window stage
  unsigned slow = 0;      // slots whose corners left the LDS window: redone by fixup()
  auto setup = [&](int j) __attribute__((always_inline)) {
    const int tap = tap0 + 2 * j;
    const int ti = tap / 3, tj = tap - 3 * ti;
    const float py = (float)(gy - 1 + ti) + oy[j];
    const float px = (float)(gx - 1 + tj) + ox[j];
    const bool in = pix_ok && py > -1.f && px > -1.f && py < (float)h && px < (float)w;
    const float fy0 = floorf(py), fx0 = floorf(px);
    const float lh = py - fy0, lw = px - fx0;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
    const float m = in ? mk[j] : 0.f;
    const int ry = hl - (y0 - PY0), rx = wl - (x0 - PX0);
    const bool in_win = ry >= 0 && ry <= PH - 2 && rx >= 0 && rx <= PW - 2;
    const float hm = hh * m, lm = lh * m;
    sw[j][0] = f32x2{hm * hw, hm * lw};
    sw[j][1] = f32x2{lm * hw, lm * lw};
    sq[j] = (in && in_win) ? ry * PW + rx : 0;
    slow = (in && !in_win) ? (slow | (1u << j)) : (slow & ~(1u << j));
  };
  auto sample_slot = [&](int j, const float* pst) __attribute__((always_inline)) {
    const float* q = pst + sq[j];
    float vals[DCK];
#pragma unroll
    for (int c = 0; c < DCK; ++c) {
      const f32x2 top = {q[c * (PH * PW)], q[c * (PH * PW) + 1]};
      const f32x2 bot = {q[c * (PH * PW) + PW], q[c * (PH * PW) + PW + 1]};
      const f32x2 r = sw[j][0] * top + sw[j][1] * bot;
      vals[c] = r.x + r.y;
    }
#pragma unroll
    for (int c = 0; c < DCK; ++c) s_col[(c * DKK + tap0 + 2 * j) * DT_PX + p] = vals[c];
  };
  auto fixup = [&](int chunk, auto ns_tag) __attribute__((always_inline)) {
    constexpr int NS = decltype(ns_tag)::value;
    const float* xp = a.x + ((size_t)bn * a.cin + chunk * DCK) * plane;
    const int g = chunk * DCK / a.cpg;
    const float* offb = a.offset + ((size_t)bn * a.dg + g) * 18 * plane;
    const float* mkb = a.mask + ((size_t)bn * a.dg + g) * 9 * plane;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (slow & (1u << j)) {
        const int tap = tap0 + 2 * j;
        const int ti = tap / 3, tj = tap - 3 * ti;
        // rare path: the offsets are read again (the prefetch registers already hold the next group's)
        const float fy = ld_b(offb, (2u * (unsigned)tap * uplane + pix) * 4u);
        const float fx = ld_b(offb, ((2u * (unsigned)tap + 1u) * uplane + pix) * 4u);
        const float fm = ld_b(mkb, ((unsigned)tap * uplane + pix) * 4u);
        const float py = (float)(gy - 1 + ti) + fy;
        const float px = (float)(gx - 1 + tj) + fx;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const float lh = py - fy0, lw = px - fx0;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
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
          s_col[(c * DKK + tap) * DT_PX + p] = v * fm;
        }
      }
    }
  };
#ifdef EAVSR_DCN_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  constexpr int WAIT_VM0 = 0x0F70;
  auto wait_vm_lgkm0 = [](auto n_tag) __attribute__((always_inline)) {
    constexpr int N = decltype(n_tag)::value;
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
  };

  // one chunk: FIRST opens a deformable group (positions set up); otherwise the next group's offsets are loaded
  auto chunk_step = [&](int c, auto first_tag, auto ns_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int NS = decltype(ns_tag)::value;
    const int stage = c & 1;
    DCN_STAMP(0);
    // window(c) has landed and is visible; the previous contraction is done with s_col / s_w
    __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    __syncthreads();
    DCN_STAMP(1);
    issue_weights(c);
    const bool more = c + 1 < nchunks;
    if (more) {
      issue_window(c + 1, stage ^ 1);
      if (!FIRST) load_offsets(c + 1, oy, ox, mk, ns_tag);
    }
    DCN_STAMP(2);
    const float* pst = s_patch + stage * PATCH_F;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (FIRST) setup(j);
      sample_slot(j, pst);
      // bound the gathers in flight (2 slots x 16 dwords): the scheduler would hoist all of them and spill
      if (j & 1) __builtin_amdgcn_sched_barrier(0);
    }
    DCN_STAMP(3);
    if (__builtin_amdgcn_ballot_w64(slow != 0) != 0) fixup(c, ns_tag);
    DCN_STAMP(4);
    // only the weight slab is needed now: the window pieces (and offset loads) of the next chunk stay in flight
    if (more) wait_vm_lgkm0(IntTag<PATCH_IT + (FIRST ? 0 : 3 * NS)>{});
    else wait_vm_lgkm0(IntTag<0>{});
    __builtin_amdgcn_s_barrier();
    DCN_STAMP(5);
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
  };

  __syncthreads();  // zero fill done before the first DMA may land
  issue_window(0, 0);
  if (even_taps) {
    load_offsets(0, oy, ox, mk, IntTag<5>{});
    for (int c = 0; c < nchunks; c += 2) {
      chunk_step(c, BoolTag<true>{}, IntTag<5>{});
      if (c + 1 < nchunks) chunk_step(c + 1, BoolTag<false>{}, IntTag<5>{});
    }
  } else {
    load_offsets(0, oy, ox, mk, IntTag<4>{});
    for (int c = 0; c < nchunks; c += 2) {
      chunk_step(c, BoolTag<true>{}, IntTag<4>{});
      if (c + 1 < nchunks) chunk_step(c + 1, BoolTag<false>{}, IntTag<4>{});
    }
  }

#ifdef EAVSR_DCN_STAMPS
  DCN_STAMP(0);
  if (lane == 0 && (wave == 0 || wave == 2)) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_dcn_stamps[(wave == 2 ? 8 : 0) + i], st_acc[i]);
  }
#endif
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

}  // namespace

#ifdef EAVSR_DCN_STAMPS
extern "C" int eavsr_debug_dcn_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dcn_stamps), sizeof(g_dcn_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_dcn_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

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
  static const bool force_global = [] {
    const char* e = getenv("EAVSR_DCN_VARIANT");  // A-B switch: "g" = global gathers without an LDS window
    return e != nullptr && e[0] == 'g';
  }();
  if (patch && !force_global) {
    if (CO == 32) hipLaunchKernelGGL(dcnv2_grp_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dcnv2_grp_kernel<2>, grid, dim3(256), 0, st, a);
  } else {
    if (CO == 32) hipLaunchKernelGGL(dcnv2_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dcnv2_kernel<2>, grid, dim3(256), 0, st, a);
  }
  return eavsr::launch_status("dcnv2");
}
