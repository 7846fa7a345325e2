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
      const char* xb = reinterpret_cast<const char*>(a.xv[seg] + ((size_t)bl * a.cin_src + a.ci0) * plane);
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
          const bool ok = ci < a.ci_valid && gy >= 0 && gy < h && gx >= 0 && gx < w;
          const unsigned off = ((unsigned)min(ci, a.ci_valid - 1) * (unsigned)plane +
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
          a.ws[(((size_t)blockIdx.x * 64 + co) * 64 + ci) * KK + tap] = acc[i][r];
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

// dW[co0+co][ci_dst0+ci][tap] (+)= sum_blk ws[blk][co][ci][tap].  64 outputs per workgroup; the four waves take every
// fourth slab (four independent load streams per output instead of one serial chain) and are added in wave order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int blocks, int kk, int co0, int co_valid, int ci_dst0,
                                                           int ci_valid, int cin_total, int accumulate) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int total = 64 * 64 * kk;   // a multiple of 64
  float s0 = 0.f, s1 = 0.f;
  int b = wv;
  for (; b + 4 < blocks; b += 8) {
    s0 += ws[(size_t)b * total + i];
    s1 += ws[(size_t)(b + 4) * total + i];
  }
  if (b < blocks) s0 += ws[(size_t)b * total + i];
  part[wv][lane] = s0 + s1;
  __syncthreads();
  if (wv != 0) return;
  const int tap = i % kk;
  const int ci = (i / kk) % 64;
  const int co = i / (kk * 64);
  if (co >= co_valid || ci >= ci_valid) return;
  const float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  float* dst = dw + ((size_t)(co0 + co) * cin_total + ci_dst0 + ci) * kk + tap;
  *dst = accumulate ? *dst + s : s;
}

template <int KS>
int launch_wgrad(const WgradArgs& a, int blocks, hipStream_t st) {
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
  hipLaunchKernelGGL(conv_wgrad_kernel<KS>, dim3(blocks), dim3(512), Cfg::LDS_BYTES, st, a);
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

}  // namespace

extern "C" int32_t eavsr_conv_wgrad_blocks(int32_t n, int32_t h, int32_t w, int32_t ksize) {
  const int th = ksize <= 3 ? 8 : 4;
  const long tiles = (long)n * eavsr::cdiv(h, th) * eavsr::cdiv(w, 32);
  return (int32_t)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

extern "C" int eavsr_conv_wgrad_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                                          float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total, int32_t co0,
                                          int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0, int32_t ksize,
                                          int32_t accumulate, void* stream) {
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
  int rc;
  switch (ksize) {
    case 1: rc = launch_wgrad<1>(a, blocks, st); break;
    case 3: rc = launch_wgrad3(a, blocks, st); break;
    default: rc = launch_wgrad<5>(a, blocks, st); break;
  }
  if (rc) return rc;
  const int kk = ksize * ksize;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64 * kk), dim3(256), 0, st, workspace, dweight,
                     n == 0 ? 0 : blocks, kk, co0, a.co_valid, ci_dst0, a.ci_valid, cin_total, accumulate);
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
