// Weight gradient of the dense stride-1 convolution (config 4, training):
//   dW[co][ci][tap] = sum_{n,y,x} dY[n,co,y,x] * X[n,ci,y+ky-P,x+kx-P]
// Reference: the cuDNN / MKLDNN conv backward-weights that autograd runs for every nn.Conv2d on the path
// (loss.backward(), models/eavsrp_model.py:109-113).
//
// GEMM view: M = cout (64 per launch), N = (ci, tap) (64 input channels x k*k taps per launch),
// K = pixels.  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] = dY of output channel i at
// pixel 2 kk + k and B[k][j = l & 31] = X of input channel j at that pixel shifted by the tap.  Each of the
// 8 waves owns up to 5 of the 4 * k*k (M-tile, ci-tile, tap) output tiles of a pass and keeps them in
// registers while the persistent workgroup walks its pixel tiles; partial sums go to a per-workgroup slab
// that a second kernel adds up in a fixed order (deterministic, no atomics).
// LDS operands are stored with ODD pitches so that the 32 lanes of a half-wave, which read 32 different
// channels at the same pixel, hit 32 different banks.
#include "common.h"

#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int WG_MAX_SEG = 8;
struct WgradArgs {
  // Up to WG_MAX_SEG SEGMENTS per launch (round 5): segment s is one (dY, X) pair of the same shapes -- one use of the weight.  A
  // weight of the recurrent path is used once per frame (7 uses per branch); its per-use gradients used to be 7 launches that
  // each fill 72 tiles of a 2 x 96 x 96 crop into 256 CUs and accumulate into dW one after the other.  As segments of ONE launch
  // the K dimension is pixels x frames: 504 tiles, one slab reduction instead of seven.  The pointers travel by value in the
  // kernel arguments (no table in device memory: the launch is captured in a HIP graph with them).
  const float* dyv[WG_MAX_SEG];   // (n, cout_total, h, w) each, this launch uses channels co0 .. co0+63
  const float* xv[WG_MAX_SEG];    // (n, cin_src, h, w) each,   this launch uses channels ci0 .. ci0+63
  float* ws;         // [blocks][64][64][KK]
  float* ws_bias;    // [blocks][64] per-workgroup sums of dY over its tiles (the bias gradient), or NULL; conv_wgrad3_x6_kernel only
  int n, h, w, cout_total, co0, co_valid, cin_src, ci0, ci_valid, tiles_x, tiles_y, num_tiles;   // n: images per segment
};

template <int KS>
struct WgCfg {
  static constexpr int KK = KS * KS, PAD = KS / 2;
  static constexpr int TH = (KS <= 3) ? 8 : 4, TW = 32, PX = TH * TW;
  static constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
  static constexpr int PA = PX + 1;                       // odd pitch of the dY tile rows
  static constexpr int PB = (IH * IW) | 1;                // odd pitch of the X patch planes
  static constexpr int T = 4 * KK;                        // output tiles: 2 (co) x 2 (ci) x taps
  static constexpr int TILES_PER_PASS = 40;               // 8 waves x 5
  static constexpr int PASSES = (T + TILES_PER_PASS - 1) / TILES_PER_PASS;
  static constexpr size_t LDS_BYTES = (size_t)(64 * PA + 64 * PB) * sizeof(float);
};

template <int KS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(WgradArgs a) {
  using Cfg = WgCfg<KS>;
  constexpr int KK = Cfg::KK, PAD = Cfg::PAD, TH = Cfg::TH, TW = Cfg::TW, PX = Cfg::PX, IH = Cfg::IH, IW = Cfg::IW;
  constexpr int PA = Cfg::PA, PB = Cfg::PB, T = Cfg::T, MAXT = 5;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;             // [64 co][PA]
  float* sB = smem + 64 * PA;   // [64 ci][PB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  // blockIdx.y: the 64-channel block of the source this workgroup takes (eavsr_conv_wgrad_span_f32: the nine blocks of DCNv2's
  // 576-channel column tensor as ONE launch instead of nine); 0 for every other caller
  const int a_ci0 = a.ci0 + 64 * (int)blockIdx.y;
  const int a_ci_valid = min(a.cin_src - a_ci0, 64);
  const size_t slab = (size_t)blockIdx.y * gridDim.x + blockIdx.x;

  for (int pass = 0; pass < Cfg::PASSES; ++pass) {
    // this wave's output tiles of the pass: t = pass*40 + wave + 8*i ; (mt, ct, tap) = (t&1, (t>>1)&1, t>>2)
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int aoff[MAXT], boff[MAXT];
    bool tv[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = pass * Cfg::TILES_PER_PASS + wave + 8 * i;
      tv[i] = t < T;
      const int tt = tv[i] ? t : 0;
      const int mt = tt & 1, ct = (tt >> 1) & 1, tap = tt >> 2;
      const int ky = tap / KS, kx = tap - ky * KS;
      aoff[i] = (mt * 32 + l31) * PA + half;
      boff[i] = (ct * 32 + l31) * PB + ky * IW + kx;
    }

    for (int tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
      int t = tile;
      const int tx = t % a.tiles_x;
      t /= a.tiles_x;
      const int ty = t % a.tiles_y;
      const int bn = t / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
      // Stage the tiles in batches of 16 loads per thread, every load of a batch issued before its first LDS
      // write: a load / wait / write loop serialises ~75 HBM latencies per tile (this kernel is latency-bound at
      // training-crop sizes).  Loads go through a wave-uniform base + 32-bit byte offset.
      constexpr int A_N = 64 * PX, B_N = 64 * IH * IW;
      constexpr int BATCH = 16;
      const int seg = bn / a.n, bl = bn - seg * a.n;      // wave-uniform
      const char* dyb = reinterpret_cast<const char*>(a.dyv[seg] + ((size_t)bl * a.cout_total + a.co0) * plane);
      const char* xb = reinterpret_cast<const char*>(a.xv[seg] + ((size_t)bl * a.cin_src + a_ci0) * plane);
      __syncthreads();  // previous tile's MFMAs are done with the LDS tiles
      // dY tile: 64 channels x PX pixels (zero beyond the image / beyond co_valid)
#pragma unroll 1
      for (int e0 = 0; e0 < A_N; e0 += BATCH * 512) {
        float tv_[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const int e = min(e0 + tid + i * 512, A_N - 1);
          const int co = e / PX, p = e - co * PX;
          const int gy = y0 + p / TW, gx = x0 + (p % TW);
          const bool ok = co < a.co_valid && gy < h && gx < w;
          const unsigned off = ((unsigned)min(co, a.co_valid - 1) * (unsigned)plane + (unsigned)(min(gy, h - 1) * w + min(gx, w - 1))) * 4u;
          const float v = *reinterpret_cast<const float*>(dyb + off);
          tv_[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const int e = e0 + tid + i * 512;
          if (e < A_N) {
            const int co = e / PX, p = e - co * PX;
            sA[co * PA + p] = tv_[i];
          }
        }
      }
      // X patch: 64 channels x IH x IW (zero padding)
#pragma unroll 1
      for (int e0 = 0; e0 < B_N; e0 += BATCH * 512) {
        float tv_[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const int e = min(e0 + tid + i * 512, B_N - 1);
          const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
          const int r = rem / IW, c = rem - r * IW;
          const int gy = y0 - PAD + r, gx = x0 - PAD + c;
          const bool ok = ci < a_ci_valid && gy >= 0 && gy < h && gx >= 0 && gx < w;
          const unsigned off = ((unsigned)min(ci, a_ci_valid - 1) * (unsigned)plane +
                                (unsigned)(min(max(gy, 0), h - 1) * w + min(max(gx, 0), w - 1))) * 4u;
          const float v = *reinterpret_cast<const float*>(xb + off);
          tv_[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const int e = e0 + tid + i * 512;
          if (e < B_N) {
            const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
            sB[ci * PB + rem] = tv_[i];
          }
        }
      }
      __syncthreads();
      // K loop over pixel pairs: pixel p = 2 kk + half -> (row p / 32, col p % 32)
#pragma unroll 2
      for (int kk = 0; kk < PX / 2; ++kk) {
        const int p = 2 * kk;
        const int prow = p / TW, pcol = (p % TW) + half;
        const int bpix = prow * IW + pcol;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          if (tv[i]) {  // wave-uniform
            const float av = sA[aoff[i] + p];
            const float bv = sB[boff[i] + bpix];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
          }
        }
      }
    }
    // partial slab of this workgroup: ws[blk][co][ci][tap]
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      if (tv[i]) {
        const int t = pass * Cfg::TILES_PER_PASS + wave + 8 * i;
        const int mt = t & 1, ct = (t >> 1) & 1, tap = t >> 2;
        const int ci = ct * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          a.ws[((slab * 64 + co) * 64 + ci) * KK + tap] = acc[i][r];
        }
      }
    }
  }
}


// 3x3 variant.  One workgroup owns ONE (co half, ci half) quadrant of the 64 x 64 x 9 output and all nine taps of it;
// its eight waves split K - wave r takes pixel row r of the 8 x 32 tile - so one dY operand feeds nine MFMAs (the
// generic kernel reads one per MFMA) and a 72-tile training crop runs 288 workgroups instead of 72 on the 256 CUs.
// The eight per-wave partial sums are added through LDS in a fixed order before the slab is written.
struct Wg3Cfg {
  static constexpr int KK = 9, TH = 8, TW = 32, PX = TH * TW, IH = TH + 2, IW = TW + 2;
  static constexpr int PA = PX + 1, PB = (IH * IW) | 1;
  static constexpr size_t LDS_BYTES = (size_t)(32 * PA + 32 * PB) * sizeof(float);
};

__global__ __launch_bounds__(512, 2) void conv_wgrad3_kernel(WgradArgs a) {
  using Cfg = Wg3Cfg;
  constexpr int KK = Cfg::KK, TW = Cfg::TW, PX = Cfg::PX, IH = Cfg::IH, IW = Cfg::IW, PA = Cfg::PA, PB = Cfg::PB;
  static_assert(8 * 1024 <= 32 * PA + 32 * PB, "the cross-wave sum reuses the operand tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;             // [32 co][PA]
  float* sB = smem + 32 * PA;   // [32 ci][PB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = blockIdx.y & 1, ct = blockIdx.y >> 1;
  const int co_valid = min(a.co_valid - 32 * mt, 32), ci_valid = min(a.ci_valid - 32 * ct, 32);
  if (co_valid <= 0 || ci_valid <= 0) return;   // the reduction never reads this quadrant
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int bn = t / a.tiles_y;
    const int y0 = ty * Cfg::TH, x0 = tx * TW;
    constexpr int A_N = 32 * PX, B_N = 32 * IH * IW;
    const int seg = bn / a.n, bl = bn - seg * a.n;      // wave-uniform
    const char* dyb = reinterpret_cast<const char*>(a.dyv[seg] + ((size_t)bl * a.cout_total + a.co0 + 32 * mt) * plane);
    const char* xb = reinterpret_cast<const char*>(a.xv[seg] + ((size_t)bl * a.cin_src + a.ci0 + 32 * ct) * plane);
    __syncthreads();  // the previous tile's MFMAs are done with the LDS tiles
    {
      // one batch of loads for dY and two for the X patch, every load of a batch in flight before its first LDS
      // write (all three at once would spill next to the 144 accumulator registers)
      constexpr int NA = A_N / 512, NB = (B_N + 1023) / 1024;
      float va[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = tid + i * 512;
        const int co = e / PX, p = e - co * PX;
        const int gy = y0 + p / TW, gx = x0 + (p % TW);
        const bool ok = co < co_valid && gy < h && gx < w;
        const unsigned off = ((unsigned)min(co, co_valid - 1) * (unsigned)plane + (unsigned)(min(gy, h - 1) * w + min(gx, w - 1))) * 4u;
        const float v = *reinterpret_cast<const float*>(dyb + off);
        va[i] = ok ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int e = tid + i * 512;
        const int co = e / PX, p = e - co * PX;
        sA[co * PA + p] = va[i];
      }
#pragma unroll 1
      for (int e0 = 0; e0 < B_N; e0 += NB * 512) {
        float vb[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int e = min(e0 + tid + i * 512, B_N - 1);
          const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
          const int r = rem / IW, c = rem - r * IW;
          const int gy = y0 - 1 + r, gx = x0 - 1 + c;
          const bool ok = ci < ci_valid && gy >= 0 && gy < h && gx >= 0 && gx < w;
          const unsigned off = ((unsigned)min(ci, ci_valid - 1) * (unsigned)plane +
                                (unsigned)(min(max(gy, 0), h - 1) * w + min(max(gx, 0), w - 1))) * 4u;
          const float v = *reinterpret_cast<const float*>(xb + off);
          vb[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int e = e0 + tid + i * 512;
          if (e < B_N) {
            const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
            sB[ci * PB + rem] = vb[i];
          }
        }
      }
    }
    __syncthreads();
    // K loop of this wave: the 16 pixel pairs of tile row `wave`
    const float* ap = sA + l31 * PA + wave * TW + half;
    const float* bp = sB + l31 * PB + wave * IW + half;
#pragma unroll 4
    for (int kk = 0; kk < TW / 2; ++kk) {
      const float av = ap[2 * kk];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[ky * IW + kx + 2 * kk], acc[ky * 3 + kx], 0, 0, 0);
    }
  }

  // partial slab of this workgroup, ws[blk][co][ci][tap]: sum of the eight waves, wave 0 first
#pragma unroll
  for (int tap = 0; tap < KK; ++tap) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) smem[wave * 1024 + r * 64 + lane] = acc[tap][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * 512;
      float v = smem[e];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += smem[k * 1024 + e];
      const int r = e >> 6, ln = e & 63;
      const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      const int ci = ct * 32 + (ln & 31);
      a.ws[(((size_t)blockIdx.x * 64 + co) * 64 + ci) * KK + tap] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 on the bf16 matrix pipe (round 5): both operands split exactly into three bf16 terms on their way into LDS, six partial
// products per 16 pixels of K (v_mfma_f32_32x32x16_bf16: 0.375 x the fp32 MFMA's cycles for the same sum; no operand rounded,
// the dropped products are below 2^-24 of the result -- as conv_x6.hip / dcnv2_il2.hip).  Weight gradients of RCABlock's
// convolutions (models/networks.py:456-464) at a training crop: 352 launches of 7 segments x 2 x 96 x 96 per step, 54 ms of the
// 241 ms step on the fp32 kernel above (0.37 of the fp32 MFMA peak, staging and multiplying in turns).
//
// One workgroup = FOUR waves on one (co half, ci half) quadrant and a 4 x 32-pixel tile; wave r takes pixel row r (two k-steps of
// 16 pixels), all nine taps: 144 accumulator registers.  73,728 bytes of LDS, so that TWO workgroups share a CU and one stages
// while the other multiplies; a workgroup's own next tile is requested (float4 row segments, 42 registers) before it multiplies the
// current one.
// LDS, per plane (hi / mid / lo):  dY [32 co][4 rows][32 px] bf16, 272 bytes per channel (16 B of padding: the 16 lanes of a
// ds_read_b128 group hit 16 different bank quads);  X [32 ci][6 rows][40 el] bf16, patch column c (gx = x0 - 1 + c) stored at
// element c + 1, 496 bytes per channel.  K runs over pixels, so a lane's B operand is EIGHT CONSECUTIVE PIXELS of one input
// channel shifted by the tap: elements c0 + kx + 1 .. c0 + kx + 8 of a row, c0 = 16 s + 8 g.  One aligned ds_read_b128 at c0 plus
// a ds_read_b64 behind it give dwords D0..D5; kx = 1 is D1..D4 as they are, kx = 0 / 2 are funnel shifts by 16 bits
// (v_alignbyte_b32), five per row and plane for both.
struct Wx6Cfg {
  static constexpr int TH = 4, TW = 32, IH = TH + 2;
  static constexpr int A_PITCH = TH * 64 + 16, A_PLANE = 32 * A_PITCH;          // 272, 8,704
  static constexpr int B_ROW = 80, B_PITCH = IH * B_ROW + 16, B_PLANE = 32 * B_PITCH;   // 496, 15,872
  static constexpr size_t LDS_BYTES = 3 * (size_t)A_PLANE + 3 * (size_t)B_PLANE;        // 73,728
};

typedef unsigned wx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wx_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void wx_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

typedef __amdgpu_buffer_rsrc_t wx_rsrc;
constexpr unsigned WX_OOB = 0x80000000u;      // beyond every extent the launcher accepts (< 2^31 bytes per 32-channel image block)
__device__ __forceinline__ wx_rsrc wx_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned wx_smul(unsigned s_, int c) {      // c: a constant once the caller is unrolled
  unsigned r;
  asm volatile("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(s_), "i"(c));
  return r;
}
__device__ __forceinline__ float wx_ld(wx_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 wx_ld4(wx_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ f32x16 wx_mfma(const wx_u32x4& a, const wx_u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wx_bf16x8, a), __builtin_bit_cast(wx_bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad3_x6_kernel(WgradArgs a) {
  using Cfg = Wx6Cfg;
  constexpr int KK = 9, TH = Cfg::TH, TW = Cfg::TW, IH = Cfg::IH;
  constexpr int A_PITCH = Cfg::A_PITCH, A_PLANE = Cfg::A_PLANE, B_ROW = Cfg::B_ROW, B_PITCH = Cfg::B_PITCH, B_PLANE = Cfg::B_PLANE;
  static_assert(4 * 1024 * 4 <= Cfg::LDS_BYTES, "the cross-wave sum reuses the operand tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* sA = reinterpret_cast<unsigned char*>(smem);
  unsigned char* sB = sA + 3 * A_PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  const int mt = blockIdx.y & 1, ct = blockIdx.y >> 1;
  const int co_valid = min(a.co_valid - 32 * mt, 32), ci_valid = min(a.ci_valid - 32 * ct, 32);
  if (co_valid <= 0 || ci_valid <= 0) return;   // the reduction never reads this quadrant
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging items of a tile: thread t = (channel t >> 3 of the quadrant, quad q = t & 7 of four pixels) takes that channel's quad
  // of every row -- 4 rows of dY, 6 rows of the X patch interior (patch columns 1..32) -- and, for (t & 7) < 6, the two halo
  // pixels (patch columns 0 and 33) of patch row t & 7: 12 requests per thread and tile.  All of them are buffer loads: one
  // per-thread offset for the life of the kernel, the tile's row as the scalar offset, channels beyond the valid ones beyond
  // the resource (the range check returns zeros); rows outside the image are wave-uniform (no request), columns per thread.
  const unsigned upl4 = (unsigned)plane * 4u, uw4 = (unsigned)w * 4u;
  const int chn = tid >> 3, q8 = tid & 7;
  const unsigned vo_q = (unsigned)chn * upl4 + (unsigned)q8 * 16u;
  const unsigned vo_c = (unsigned)chn * upl4;                                  // halo: + its row, per tile
  unsigned char* lds_a = sA + chn * A_PITCH + q8 * 8;
  unsigned char* lds_b = sB + chn * B_PITCH + 4 + q8 * 8;                       // elements 2 + 4 q .. 5 + 4 q of patch row 0
  unsigned char* lds_h = sB + chn * B_PITCH + q8 * B_ROW;                       // element 0 of patch row q8
  f32x4 va[4], vb[6];
  float vh[2];
  auto request = [&](int tile) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int bn = t / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int seg = bn / a.n, bl = bn - seg * a.n;      // wave-uniform
    const wx_rsrc r_dy = wx_make_rsrc(a.dyv[seg] + ((size_t)bl * a.cout_total + a.co0 + 32 * mt) * plane, (unsigned)co_valid * upl4);
    const wx_rsrc r_x = wx_make_rsrc(a.xv[seg] + ((size_t)bl * a.cin_src + a.ci0 + 32 * ct) * plane, (unsigned)ci_valid * upl4);
    const unsigned vq = x0 + 4 * q8 < w ? vo_q : WX_OOB;      // (w % 4 == 0: the whole quad is inside or outside)
    const unsigned so0 = (unsigned)(y0 * w + x0) * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      va[i] = y0 + i < h ? wx_ld4(r_dy, vq, so0 + wx_smul(uw4, i)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int gy = y0 - 1 + i;
      vb[i] = (gy >= 0 && gy < h) ? wx_ld4(r_x, vq, so0 + wx_smul(uw4, i) - uw4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (the scalar offset is not range-checked and must not go negative: the halo's row lives in the per-thread offset)
    const int hy = y0 - 1 + q8;
    const unsigned vh_ = (q8 < 6 && hy >= 0 && hy < h) ? vo_c + (unsigned)hy * uw4 : WX_OOB;
    vh[0] = x0 > 0 ? wx_ld(r_x, vh_, (unsigned)x0 * 4u - 4u) : 0.f;
    vh[1] = x0 + TW < w ? wx_ld(r_x, vh_, (unsigned)x0 * 4u + TW * 4u) : 0.f;
  };
  float bsum = 0.f;      // this thread's share of sum(dY) of channel `chn` (the bias gradient rides along: dY is staged here anyway)
  auto stage = [&]() __attribute__((always_inline)) {
    if (ct == 0 && a.ws_bias != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bsum += (va[i][0] + va[i][1]) + (va[i][2] + va[i][3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      wx_split2(va[i][0], va[i][1], h0, m0, l0);
      wx_split2(va[i][2], va[i][3], h1, m1, l1);
      unsigned char* d = lds_a + i * 64;
      *reinterpret_cast<wx_u32x2*>(d) = wx_u32x2{h0, h1};
      *reinterpret_cast<wx_u32x2*>(d + A_PLANE) = wx_u32x2{m0, m1};
      *reinterpret_cast<wx_u32x2*>(d + 2 * A_PLANE) = wx_u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      wx_split2(vb[i][0], vb[i][1], h0, m0, l0);
      wx_split2(vb[i][2], vb[i][3], h1, m1, l1);
      unsigned* d = reinterpret_cast<unsigned*>(lds_b + i * B_ROW);
      d[0] = h0; d[1] = h1;
      d[B_PLANE / 4] = m0; d[B_PLANE / 4 + 1] = m1;
      d[2 * (B_PLANE / 4)] = l0; d[2 * (B_PLANE / 4) + 1] = l1;
    }
    if (q8 < 6) {
      unsigned h0, m0, l0;
      wx_split2(vh[0], vh[1], h0, m0, l0);      // low halves: patch column 0 (element 1), high halves: column 33 (element 34)
      unsigned short* d = reinterpret_cast<unsigned short*>(lds_h);
      d[1] = (unsigned short)h0; d[34] = (unsigned short)(h0 >> 16);
      d[B_PLANE / 2 + 1] = (unsigned short)m0; d[B_PLANE / 2 + 34] = (unsigned short)(m0 >> 16);
      d[2 * (B_PLANE / 2) + 1] = (unsigned short)l0; d[2 * (B_PLANE / 2) + 34] = (unsigned short)(l0 >> 16);
    }
  };

#ifdef EAVSR_WX6_PREFETCH      // A/B: a workgroup's own next tile in flight under its MFMAs (42 more registers)
  int tile = blockIdx.x;
  if (tile < a.num_tiles) request(tile);
  for (; tile < a.num_tiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's MFMAs are done with the LDS tiles
    stage();
    __syncthreads();
    if (tile + (int)gridDim.x < a.num_tiles) request(tile + gridDim.x);
#else
  for (int tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
#ifdef EAVSR_WX6_EXP_NO_LOADS       // ablation: one request per workgroup
    if (tile == (int)blockIdx.x)
#endif
    request(tile);      // (the other workgroup of the CU multiplies meanwhile)
    __syncthreads();    // the previous tile's MFMAs are done with the LDS tiles
#ifdef EAVSR_WX6_EXP_NO_STAGE       // ablation: one split + store per workgroup
    if (tile == (int)blockIdx.x)
#endif
    stage();
    __syncthreads();
#endif
    // K loop of this wave: pixel row `wave`, two k-steps of 16 pixels x three filter rows = six groups of 18 MFMAs; the raw
    // operand dwords of group g + 1 are read before the MFMAs of group g, nothing else crosses a group boundary (left alone the
    // scheduler hoists every read of the tile to the top and spills the accumulators)
    const unsigned char* ap = sA + l31 * A_PITCH + wave * 64 + kg * 16;
    const unsigned char* bp = sB + l31 * B_PITCH + wave * B_ROW + kg * 16;
    wx_u32x4 av[3], d03[3], n03[3];
    wx_u32x2 d45[3], n45[3];
    auto read_raw = [&](int g, wx_u32x4 (&r03)[3], wx_u32x2 (&r45)[3]) __attribute__((always_inline)) {
      const int s_ = g / 3, ky = g - 3 * s_;
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) {
        r03[p3] = *reinterpret_cast<const wx_u32x4*>(bp + p3 * B_PLANE + ky * B_ROW + s_ * 32);
        r45[p3] = *reinterpret_cast<const wx_u32x2*>(bp + p3 * B_PLANE + ky * B_ROW + s_ * 32 + 16);
      }
    };
#ifndef EAVSR_WX6_NO_READAHEAD
    read_raw(0, d03, d45);
#endif
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const int s_ = g / 3, ky = g - 3 * s_;
      if (ky == 0) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) av[p3] = *reinterpret_cast<const wx_u32x4*>(ap + p3 * A_PLANE + s_ * 32);
      }
#ifdef EAVSR_WX6_NO_READAHEAD      // A/B (with EAVSR_WX6_PREFETCH: 18 registers less)
      read_raw(g, d03, d45);
#else
      if (g + 1 < 6) read_raw(g + 1, n03, n45);
#endif
      wx_u32x4 b0[3], b1[3], b2[3];
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) {
        const unsigned s10 = __builtin_amdgcn_alignbyte(d03[p3][1], d03[p3][0], 2), s21 = __builtin_amdgcn_alignbyte(d03[p3][2], d03[p3][1], 2);
        const unsigned s32 = __builtin_amdgcn_alignbyte(d03[p3][3], d03[p3][2], 2), s43 = __builtin_amdgcn_alignbyte(d45[p3][0], d03[p3][3], 2);
        const unsigned s54 = __builtin_amdgcn_alignbyte(d45[p3][1], d45[p3][0], 2);
        b0[p3] = wx_u32x4{s10, s21, s32, s43};                                  // elements c0 + 1 .. c0 + 8: tap kx = 0
        b1[p3] = wx_u32x4{d03[p3][1], d03[p3][2], d03[p3][3], d45[p3][0]};      // c0 + 2 .. c0 + 9:  kx = 1
        b2[p3] = wx_u32x4{s21, s32, s43, s54};                                  // c0 + 3 .. c0 + 10: kx = 2
      }
      // the six partial products, smallest first: (A plane, B plane) = (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
#define WX_TAP(T, B)                             \
      acc[T] = wx_mfma(av[2], B[0], acc[T]);     \
      acc[T] = wx_mfma(av[0], B[2], acc[T]);     \
      acc[T] = wx_mfma(av[1], B[1], acc[T]);     \
      acc[T] = wx_mfma(av[1], B[0], acc[T]);     \
      acc[T] = wx_mfma(av[0], B[1], acc[T]);     \
      acc[T] = wx_mfma(av[0], B[0], acc[T]);
#ifdef EAVSR_WX6_EXP_NO_MFMA      // ablation (tools/gpu_wgrad_diag.py): operands read and shifted, not multiplied
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) asm volatile("" ::"v"(b0[p3]), "v"(b1[p3]), "v"(b2[p3]), "v"(av[p3]));
#else
      WX_TAP(ky * 3 + 0, b0)
      WX_TAP(ky * 3 + 1, b1)
      WX_TAP(ky * 3 + 2, b2)
#endif
#undef WX_TAP
      __builtin_amdgcn_sched_barrier(0);
#ifndef EAVSR_WX6_NO_READAHEAD
      if (g + 1 < 6) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          d03[p3] = n03[p3];
          d45[p3] = n45[p3];
        }
      }
#endif
    }
  }

  // partial slab of this workgroup, ws[blk][co][ci][tap]: sum of the four waves, wave 0 first
#ifdef EAVSR_WX6_EXP_NO_EPI      // ablation: no cross-wave sum, no slab
#pragma unroll
  for (int tap = 0; tap < KK; ++tap) asm volatile("" ::"v"(acc[tap]));
  return;
#endif
  if (ct == 0 && a.ws_bias != nullptr) {      // eight lanes per channel (the quads of a row), then the slab's 32 channels of this quadrant
    float v = bsum;
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    if (q8 == 0) a.ws_bias[(size_t)blockIdx.x * 64 + 32 * mt + chn] = v;
  }
  // Three taps per round (48 KB of LDS, two barriers); the slab is TAP-MAJOR, ws[blk][tap][co][ci], so that a half-wave stores
  // 32 consecutive input channels (the [co][ci][tap] form of the fp32 kernels is a 36-byte stride per lane: nine partial passes
  // over every line; 10 us of a 27 us one-segment launch together with nine rounds of barriers).
  static_assert(3 * 4 * 1024 * 4 <= Cfg::LDS_BYTES, "three taps of four waves");
#pragma unroll
  for (int t3 = 0; t3 < KK; t3 += 3) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) smem[(u * 4 + wave) * 1024 + r * 64 + lane] = acc[t3 + u][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = tid + j * 256;
        float v = smem[u * 4096 + e];
#pragma unroll
        for (int k = 1; k < 4; ++k) v += smem[u * 4096 + k * 1024 + e];
        const int r = e >> 6, ln = e & 63;
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int ci = ct * 32 + (ln & 31);
        a.ws[(((size_t)blockIdx.x * KK + t3 + u) * 64 + co) * 64 + ci] = v;
      }
  }
}

// dW[co0+co][ci_dst0+ci][tap] (+)= sum_blk ws[blk][..].  64 outputs per workgroup; the eight waves take every eighth slab, four
// loads in flight each (a launch of 128-256 slabs is one or two latency round trips per wave instead of sixteen), and are added
// in wave order: deterministic.  tap_major: the slab is [tap][co][ci] (the bf16x6 3x3 kernel), otherwise [co][ci][tap].
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int blocks, int kk, int co0, int co_valid, int ci_dst0,
                                                           int ci_valid, int cin_total, int accumulate, int tap_major,
                                                           const float* __restrict__ ws_bias, float* __restrict__ dbias) {
  __shared__ float part[8][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if ((int)blockIdx.x == 64 * kk) {      // the extra workgroup: dbias[co0 + co] (+)= sum_blk ws_bias[blk][co], same order
    float s = 0.f;
    for (int b = wv; b < blocks; b += 8) s += ws_bias[(size_t)b * 64 + lane];
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && lane < co_valid) {
      const float t = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
                      ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
      dbias[co0 + lane] = accumulate ? dbias[co0 + lane] + t : t;
    }
    return;
  }
  const int i = blockIdx.x * 64 + lane;
  const int total = 64 * 64 * kk;   // a multiple of 64
  // blockIdx.y: the 64-channel source block (eavsr_conv_wgrad_span_f32); its slabs follow the previous block's
  ws += (size_t)blockIdx.y * blocks * total;
  ci_dst0 += 64 * (int)blockIdx.y;
  ci_valid = min(ci_valid - 64 * (int)blockIdx.y, 64);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = wv;
  for (; b + 24 < blocks; b += 32) {
    const float v0 = ws[(size_t)b * total + i], v1 = ws[(size_t)(b + 8) * total + i];
    const float v2 = ws[(size_t)(b + 16) * total + i], v3 = ws[(size_t)(b + 24) * total + i];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; b < blocks; b += 8) s0 += ws[(size_t)b * total + i];
  part[wv][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wv != 0) return;
  const int tap = tap_major ? i / 4096 : i % kk;
  const int ci = tap_major ? i % 64 : (i / kk) % 64;
  const int co = tap_major ? (i / 64) % 64 : i / (kk * 64);
  if (co >= co_valid || ci >= ci_valid) return;
  const float s = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
                  ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
  float* dst = dw + ((size_t)(co0 + co) * cin_total + ci_dst0 + ci) * kk + tap;
  *dst = accumulate ? *dst + s : s;
}

template <int KS>
int launch_wgrad(const WgradArgs& a, int blocks, hipStream_t st, int ci_blocks = 1) {
  using Cfg = WgCfg<KS>;
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<KS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv_wgrad_kernel<KS>, dim3(blocks, ci_blocks), dim3(512), Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("conv_wgrad");
}

int launch_wgrad3(const WgradArgs& a, int blocks, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg3Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv_wgrad3_kernel, dim3(blocks, 4), dim3(512), Wg3Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("conv_wgrad");
}

int launch_wgrad3_x6(const WgradArgs& a, int blocks, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3_x6_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wx6Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv_wgrad3_x6_kernel, dim3(blocks, 4), dim3(256), Wx6Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("conv_wgrad");
}

// EAVSR_WGRAD3=fp32 keeps the 3x3 weight gradient on the fp32 MFMA kernel (A/B switch; chosen once per process)
bool wgrad3_x6_enabled() {
  static const bool on = [] { const char* e = getenv("EAVSR_WGRAD3"); return e == nullptr || strcmp(e, "fp32") != 0; }();
  return on;
}

}  // namespace

extern "C" int eavsr_wgrad3_mode(void) { return wgrad3_x6_enabled() ? 1 : 0; }

extern "C" int32_t eavsr_conv_wgrad_blocks(int32_t n, int32_t h, int32_t w, int32_t ksize) {
  const int th = ksize <= 3 ? 8 : 4;
  const long tiles = (long)n * eavsr::cdiv(h, th) * eavsr::cdiv(w, 32);
  return (int32_t)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

extern "C" int eavsr_channel_sum_multi_f32(const void* const* a_list, int32_t nseg, float* out, int32_t n, int32_t c, int32_t hw,
                                           int32_t accumulate, void* stream);
extern "C" int eavsr_conv_wgrad_bias_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                                               float* dbias, float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total,
                                               int32_t co0, int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0,
                                               int32_t ksize, int32_t accumulate, void* stream);
extern "C" int eavsr_conv_wgrad_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                                          float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total, int32_t co0,
                                          int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0, int32_t ksize,
                                          int32_t accumulate, void* stream) {
  return eavsr_conv_wgrad_bias_multi_f32(dy_list, x_list, nseg, dweight, nullptr, workspace, n, h, w, cout_total, co0, cin_src, ci0,
                                         cin_total, ci_dst0, ksize, accumulate, stream);
}

extern "C" int eavsr_conv_wgrad_bias_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                                               float* dbias, float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total,
                                               int32_t co0, int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0,
                                               int32_t ksize, int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(dy_list && x_list && dweight && workspace, -1, "conv_wgrad: NULL pointer");
  EAVSR_REQUIRE(nseg >= 1 && nseg <= WG_MAX_SEG, -1, "conv_wgrad: %d segments (1..%d)", nseg, WG_MAX_SEG);
  EAVSR_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, -2, "conv_wgrad: kernel size %d unsupported (1, 3, 5)", ksize);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0 && cout_total > 0 && cin_src > 0 && cin_total > 0, -1, "conv_wgrad: bad dims");
  EAVSR_REQUIRE(co0 >= 0 && co0 < cout_total && ci0 >= 0 && ci0 < cin_src && ci_dst0 >= 0 && ci_dst0 < cin_total, -1,
                "conv_wgrad: channel offsets out of range");
  WgradArgs a;
  for (int s = 0; s < WG_MAX_SEG; ++s) {
    a.dyv[s] = reinterpret_cast<const float*>(dy_list[s < nseg ? s : 0]);
    a.xv[s] = reinterpret_cast<const float*>(x_list[s < nseg ? s : 0]);
    EAVSR_REQUIRE(a.dyv[s] && a.xv[s], -1, "conv_wgrad: NULL segment pointer");
  }
  a.ws = workspace;
  a.ws_bias = nullptr;
  a.n = n; a.h = h; a.w = w;
  a.cout_total = cout_total; a.co0 = co0; a.co_valid = cout_total - co0 < 64 ? cout_total - co0 : 64;
  a.cin_src = cin_src; a.ci0 = ci0; a.ci_valid = cin_src - ci0 < 64 ? cin_src - ci0 : 64;
  EAVSR_REQUIRE(ci_dst0 + a.ci_valid <= cin_total, -1, "conv_wgrad: destination channel range exceeds cin_total");
  const int th = ksize <= 3 ? 8 : 4;
  a.tiles_x = eavsr::cdiv(w, 32);
  a.tiles_y = eavsr::cdiv(h, th);
  EAVSR_REQUIRE((long)a.tiles_x * a.tiles_y * n * nseg < (1L << 31), -1, "conv_wgrad: too many tiles");
  a.num_tiles = a.tiles_x * a.tiles_y * n * nseg;
  const int blocks = eavsr_conv_wgrad_blocks(n * nseg, h, w, ksize);
  hipStream_t st = eavsr::as_stream(stream);
  int rc, slabs = blocks;
  // the bf16x6 kernel: float4 row segments, i.e. w % 4 == 0 and 16-byte aligned tensors (anything else: the fp32 kernel)
  bool x6 = ksize == 3 && n > 0 && w % 4 == 0 && (long)h * w * 128 < (1L << 31) && wgrad3_x6_enabled();
  for (int s = 0; x6 && s < nseg; ++s)
    x6 = ((reinterpret_cast<uintptr_t>(a.dyv[s]) | reinterpret_cast<uintptr_t>(a.xv[s])) & 15) == 0;
  const int kk = ksize * ksize;
  if (x6) {
    WgradArgs b = a;
    if (dbias != nullptr) b.ws_bias = workspace + (size_t)blocks * 64 * 64 * kk;      // behind the slabs (the caller sized it)
    b.tiles_y = eavsr::cdiv(h, Wx6Cfg::TH);
    EAVSR_REQUIRE((long)b.tiles_x * b.tiles_y * n * nseg < (1L << 31), -1, "conv_wgrad: too many tiles");
    b.num_tiles = b.tiles_x * b.tiles_y * n * nseg;
    slabs = b.num_tiles < 128 ? b.num_tiles : 128;      // two 4-wave workgroups per CU; never more slabs than `blocks`
    if (slabs > blocks) slabs = blocks;
    rc = launch_wgrad3_x6(b, slabs, st);
  } else {
    switch (ksize) {
      case 1: rc = launch_wgrad<1>(a, blocks, st); break;
      case 3: rc = launch_wgrad3(a, blocks, st); break;
      default: rc = launch_wgrad<5>(a, blocks, st); break;
    }
  }
  if (rc) return rc;
  const bool bias_in_kernel = x6 && dbias != nullptr;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64 * kk + (bias_in_kernel ? 1 : 0)), dim3(512), 0, st, workspace, dweight,
                     n == 0 ? 0 : slabs, kk, co0, a.co_valid, ci_dst0, a.ci_valid, cin_total, accumulate, x6 ? 1 : 0,
                     bias_in_kernel ? workspace + (size_t)blocks * 64 * 64 * kk : nullptr, dbias);
  rc = eavsr::launch_status("conv_wgrad_reduce");
  if (rc) return rc;
  // every other kernel: the bias gradient of ALL output channels as its own launch, once (with the first 64-channel block)
  if (dbias != nullptr && !bias_in_kernel && co0 == 0)
    return eavsr_channel_sum_multi_f32(dy_list, nseg, dbias, n, cout_total, h * w, accumulate, stream);
  return 0;
}

// ksize 1 over a span of 64-channel source blocks in ONE launch (+ one reduction): DCNv2's weight gradient is the 1x1 weight gradient
// of dY against the 576-channel column tensor (backward_dcn.hip) -- nine launches and nine reductions of ~20 us each per call before
extern "C" int eavsr_conv_wgrad_span_f32(const float* dy, const float* x, float* dweight, float* workspace, int32_t n, int32_t h,
                                         int32_t w, int32_t cout_total, int32_t co0, int32_t cin_src, int32_t cin_total,
                                         int32_t ci_dst0, int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(dy && x && dweight && workspace, -1, "conv_wgrad_span: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0 && cout_total > 0 && cin_src > 0 && cin_total > 0, -1, "conv_wgrad_span: bad dims");
  EAVSR_REQUIRE(co0 >= 0 && co0 < cout_total && ci_dst0 >= 0 && ci_dst0 + cin_src <= cin_total, -1,
                "conv_wgrad_span: channel offsets out of range");
  const int ci_blocks = eavsr::cdiv(cin_src, 64);
  EAVSR_REQUIRE(ci_blocks <= 1024, -1, "conv_wgrad_span: %d source channels", cin_src);
  WgradArgs a;
  for (int s = 0; s < WG_MAX_SEG; ++s) { a.dyv[s] = dy; a.xv[s] = x; }
  a.ws = workspace; a.ws_bias = nullptr;
  a.n = n; a.h = h; a.w = w;
  a.cout_total = cout_total; a.co0 = co0; a.co_valid = cout_total - co0 < 64 ? cout_total - co0 : 64;
  a.cin_src = cin_src; a.ci0 = 0; a.ci_valid = cin_src < 64 ? cin_src : 64;
  a.tiles_x = eavsr::cdiv(w, 32);
  a.tiles_y = eavsr::cdiv(h, 8);
  EAVSR_REQUIRE((long)a.tiles_x * a.tiles_y * n < (1L << 31), -1, "conv_wgrad_span: too many tiles");
  a.num_tiles = a.tiles_x * a.tiles_y * n;
  const int blocks = eavsr_conv_wgrad_blocks(n, h, w, 1);
  hipStream_t st = eavsr::as_stream(stream);
  const int rc = launch_wgrad<1>(a, blocks, st, ci_blocks);
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64, ci_blocks), dim3(512), 0, st, workspace, dweight, n == 0 ? 0 : blocks, 1, co0,
                     a.co_valid, ci_dst0, cin_src, cin_total, accumulate, 0, nullptr, nullptr);
  return eavsr::launch_status("conv_wgrad_reduce");
}

extern "C" int eavsr_conv_wgrad_f32(const float* dy, const float* x, float* dweight, float* workspace,
                                    int32_t n, int32_t h, int32_t w, int32_t cout_total, int32_t co0,
                                    int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0, int32_t ksize,
                                    int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(dy && x, -1, "conv_wgrad: NULL pointer");
  const void* dl[1] = {dy};
  const void* xl[1] = {x};
  return eavsr_conv_wgrad_multi_f32(dl, xl, 1, dweight, workspace, n, h, w, cout_total, co0, cin_src, ci0, cin_total, ci_dst0, ksize,
                                    accumulate, stream);
}
