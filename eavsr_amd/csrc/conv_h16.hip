// 16-bit (bf16 / fp16) residual-backbone kernels (BASELINE.json configs[2] "bf16 ... MFMA residual convs" and
// configs[4] "fp16 ... HBM-bound stress"; SURVEY.md 8a: a10, a11).
//
// Reference: the 3x3 64->64 convs of RCABlock / RCAGroup (models/networks.py:456-458,478) -- in 16-bit the
// arithmetic intensity of one conv (288 FLOP/B) is below the MI355X ridge (~400 FLOP/B at 2.5 PF / 6.3 TB/s),
// so the kernel is built to stream: activations live in HBM as NHWC 16-bit (one pixel = 128 contiguous
// bytes), the whole 64 x 576 weight matrix stays resident in LDS for the lifetime of a persistent
// workgroup, and the only HBM traffic is one read of the input (+ tile halo) and one write of the output.
//
//   v_mfma_f32_32x32x16_{bf16,f16}:  A[row = co][k = 8 h + j]  (weights),  B[k = 8 h + j][col = pixel]
//   (h = lane >> 5, j = 0..7): a lane needs 8 consecutive input channels of its pixel at one tap -> one
//   ds_read_b128 from the NHWC tile.  k-step s = (tap, 16-channel block): 36 steps, 2 M-tiles -> 72 MFMAs
//   per wave and 8 x 32-pixel tile.
//   LDS image of the input patch: [row][col][8 x 16-byte blocks], block index XOR-swizzled with
//   ((col >> 1) & 7) so that the 16 lanes of a ds_read_b128 group (consecutive pixels, 128-byte stride) hit
//   16 different bank quads.  The patch arrives by 16-byte LDS-DMA, whose LDS side is lane-linear: the
//   swizzle is applied to the per-lane SOURCE address (zero padding = never-written, zero-initialised LDS).
//   Two patch stages: the DMA of tile t+1 runs behind the MFMAs of tile t.
//   Epilogue: fp32 accumulators (+ bias, ReLU, optional per-tile channel sums in fp32) are rounded to
//   16-bit, staged through the consumed patch stage with the same swizzle and leave as whole 128-byte pixel
//   rows (16 bytes per lane).
#include "common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <mutex>

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int HT_H = 8, HT_W = 32;                 // output tile: one row of 32 pixels per wave, 8 waves
constexpr int HP_H = HT_H + 2, HP_W = HT_W + 2;    // patch with the 3x3 halo
constexpr int HP_PIX = HP_H * HP_W;                // 340 pixels x 128 B = 43,520 B
constexpr int HP_BYTES = HP_PIX * 128;
constexpr int HP_SEGS = (HP_BYTES + 1023) / 1024;  // 1 KiB DMA pieces (43 pieces, the last one partial)
constexpr int HP_IT = (HP_SEGS + 7) / 8;
constexpr int HW_BYTES = 64 * 576 * 2;             // 73,728 B of weights
constexpr int HW_SEGS = HW_BYTES / 1024;           // 72
constexpr int H_LDS_BYTES = HW_BYTES + 2 * HP_SEGS * 1024 + 8 * 64 * 4;

#ifdef EAVSR_H16_STAMPS
// diagnostic build only: shader cycles per phase, summed over wave 0 of every workgroup (tools/gpu_h16_ablate.py)
__device__ unsigned long long g_h16_stamps[8];
#define H16_STAMP(i)                                                  \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define H16_STAMP(i) do { } while (0)
#endif

struct H16Args {
  const void* x;      // (n, h, w, 64) 16-bit
  const void* wp;     // packed weights [36 k-steps][2 halves][64 co][8] 16-bit
  const float* bias;  // fp32 [64] or NULL
  void* out;          // (n, h, w, 64) 16-bit
  float* chan_partial;
  int n, h, w, tiles_x, tiles_y, num_tiles, relu;
};

template <bool BF16> __device__ __forceinline__ unsigned short to_h16(float v);
template <> __device__ __forceinline__ unsigned short to_h16<true>(float v) {
  return __builtin_bit_cast(unsigned short, __float2bfloat16(v));
}
template <> __device__ __forceinline__ unsigned short to_h16<false>(float v) {
  return __builtin_bit_cast(unsigned short, (_Float16)v);
}

template <bool BF16> __device__ __forceinline__ float from_h16(unsigned short v);
template <> __device__ __forceinline__ float from_h16<true>(unsigned short v) {
  return __builtin_bit_cast(float, (unsigned)v << 16);
}
template <> __device__ __forceinline__ float from_h16<false>(unsigned short v) {
  return (float)__builtin_bit_cast(_Float16, v);
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const f32x4& a, const f32x4& b, const f32x16& c) {
  if (BF16) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
}

// byte offset of 16-byte block b of patch pixel (r, c) in the swizzled LDS image
__device__ __forceinline__ int patch_off(int r, int c, int b) { return ((r * HP_W + c) << 7) + ((b ^ ((c >> 1) & 7)) << 4); }

template <bool BF16>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_h16_kernel(H16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                                  // resident weights
  unsigned char* s_p = smem + HW_BYTES;                       // two patch stages
  float* s_red = reinterpret_cast<float*>(smem + HW_BYTES + 2 * HP_SEGS * 1024);
  // the bias once per launch through LDS (the 512 never-written bytes behind patch stage 0: 43 one-KiB pieces hold 42.5 KiB):
  // 32 global loads per lane in every tile's epilogue (8 dependent load -> wait -> convert groups) were most of a tile's time
  float* s_bias = reinterpret_cast<float*>(s_p + HP_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  eavsr_stagger_priority(wave);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = a.h, w = a.w;

  // zero both patch stages once (pieces outside the image are never moved), fetch the weights once
  {
    f32x4* z = reinterpret_cast<f32x4*>(s_p);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < 2 * HP_SEGS * 64; e += 512) z[e] = zero;
  }
  __syncthreads();
  if (tid < 64) s_bias[tid] = a.bias ? a.bias[tid] : 0.f;
#ifndef EAVSR_H16_EXP_NO_WDMA
#pragma unroll 1
  for (int seg = wave; seg < HW_SEGS; seg += 8)
#else
  for (int seg = wave; seg < 0; seg += 8)
#endif
    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a.wp) + seg * 1024 + lane * 16),
                                     (lptr_t)(s_w + seg * 1024), 16, 0, 0);

  // patch DMA: piece = 64 consecutive 16-byte LDS slots; slot e -> pixel e >> 3, stored block e & 7 holds the
  // logical block (e & 7) ^ swz(col)
  auto issue_patch = [&](int tile, int stage) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int bn = t / a.tiles_y;
    const int y0 = ty * HT_H - 1, x0 = tx * HT_W - 1;
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)bn * h * w * 128;
#pragma unroll 1
    for (int i = 0; i < HP_IT; ++i) {
      const int seg = i * 8 + wave;
      const int e = seg * 64 + lane;
      const int p = e >> 3, sb = e & 7;
      const int r = p / HP_W, c = p - r * HP_W;
      const int gy = y0 + r, gx = x0 + c;
      if (seg < HP_SEGS && p < HP_PIX && gy >= 0 && gy < h && gx >= 0 && gx < w) {
        const int lb = sb ^ ((c >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(xb + ((size_t)gy * w + gx) * 128 + lb * 16),
                                         (lptr_t)(s_p + stage * (HP_SEGS * 1024) + seg * 1024), 16, 0, 0);
      }
    }
  };
  // out-of-image pieces of a stage may hold the previous tile's data: clear them (cheap, only border tiles)
  auto clear_border = [&](int tile, int stage) -> bool {   // true: something was cleared (the same answer in every wave)
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int y0 = ty * HT_H - 1, x0 = tx * HT_W - 1;
    if (y0 >= 0 && x0 >= 0 && y0 + HP_H <= h && x0 + HP_W <= w) return false;  // interior tile (wave-uniform)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < HP_PIX * 8; e += 512) {
      const int p = e >> 3;
      const int r = p / HP_W, c = p - r * HP_W;
      const int gy = y0 + r, gx = x0 + c;
      if (!(gy >= 0 && gy < h && gx >= 0 && gx < w))
        *reinterpret_cast<f32x4*>(s_p + stage * (HP_SEGS * 1024) + e * 16) = zero;
    }
    return true;
  };

  int tile = blockIdx.x;
  int stage = 0;
#ifdef EAVSR_H16_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
#ifndef EAVSR_H16_EXP_NO_DMA
  if (tile < a.num_tiles) issue_patch(tile, 0);
#endif
  for (; tile < a.num_tiles; tile += gridDim.x, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // weights + patch(tile) landed; the other stage is free (its epilogue stores are done)
    H16_STAMP(0);     // prologue / waiting for the DMA and the barrier
    const int next = tile + gridDim.x;
    const unsigned char* pst = s_p + stage * (HP_SEGS * 1024);
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // No LDS-DMA is in flight inside this loop (the next tile's patch is requested right after it, under the epilogue): with one
    // pending the compiler turns every operand wait into lgkmcnt(0) -- each MFMA pair then waits for the reads just issued for
    // the NEXT pair, an exposed LDS round trip per k-step (14 us per tile instead of 2).  The operands of tap t+1 (4 B + 8 A
    // fragments) are requested before the 8 MFMAs of tap t.
    f32x4 bq[2][4], aq[2][8];
    auto load_tap = [&](int tap, int set) __attribute__((always_inline)) {
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        bq[set][cb] = *reinterpret_cast<const f32x4*>(pst + patch_off(wave + ky, l31 + kx, cb * 2 + half));
#pragma unroll
        for (int m = 0; m < 2; ++m)
          aq[set][cb * 2 + m] = *reinterpret_cast<const f32x4*>(s_w + ((((tap * 4 + cb) * 2 + half) * 64 + m * 32 + l31) << 4));
      }
    };
    load_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      __builtin_amdgcn_sched_barrier(0);     // keep the requests of tap t+1 in front of the MFMAs of tap t
      if (tap + 1 < 9) load_tap(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#ifdef EAVSR_H16_EXP_NO_MFMA
          acc[m][cb] += aq[tap & 1][cb * 2 + m][0] + bq[tap & 1][cb][1];
#else
          acc[m] = mfma16<BF16>(aq[tap & 1][cb * 2 + m], bq[tap & 1][cb], acc[m]);
#endif
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    H16_STAMP(1);     // operand reads + MFMAs
    if (next < a.num_tiles) {
      clear_border(next, stage ^ 1);   // zero fills and DMA pieces touch disjoint slots: no barrier between them
#ifndef EAVSR_H16_EXP_NO_DMA
      issue_patch(next, stage ^ 1);
#endif
    }
    H16_STAMP(2);     // border clear + next patch DMA issue
    // ---- epilogue ------------------------------------------------------------------------------
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int bn = t / a.tiles_y;
    const int gy = ty * HT_H + wave, gx = tx * HT_W + l31;
    const bool ok = gy < h && gx < w;
    __syncthreads();  // every wave is done reading this patch stage: reuse it as the output staging tile
    unsigned char* ost = s_p + stage * (HP_SEGS * 1024);   // [8 rows][32 px][128 B], same block swizzle
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // registers 4q .. 4q+3 are 4 consecutive output channels co = m*32 + 8q + 4*half + (0..3)
        const int co = m * 32 + 8 * q + 4 * half;
        unsigned short pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[m][4 * q + j] + s_bias[co + j];
          if (a.relu) v = fmaxf(v, 0.f);
          pk[j] = to_h16<BF16>(ok ? v : 0.f);   // pixels outside the image stage as zeros (they are not stored)
        }
        const int blk = co >> 3;  // 16-byte block of the pixel row, this lane writes its 8-byte half
        const int off = ((wave * HT_W + l31) << 7) + ((blk ^ ((l31 >> 1) & 7)) << 4) + ((co & 4) << 1);
        *reinterpret_cast<unsigned long long*>(ost + off) =
            (unsigned long long)pk[0] | ((unsigned long long)pk[1] << 16) | ((unsigned long long)pk[2] << 32) |
            ((unsigned long long)pk[3] << 48);
      }
    }
    __syncthreads();
    H16_STAMP(3);     // bias / activation / rounding / staging writes
    if (a.chan_partial) {
      // per-tile channel sums from the staged tile (the 16-bit values the next layer will actually read):
      // thread -> 8 channels (one 16-byte block) of 4 pixels, then 3 shuffle steps over the 8 lanes that share
      // the block, then the 8 waves through LDS.  (A 160-shuffle reduction of the accumulators cost 22 us.)
      const int cb = tid & 7;
      float cs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = (tid >> 3) + 64 * i;
        const int c = p & (HT_W - 1);
        const s16x8 v = *reinterpret_cast<const s16x8*>(ost + (p << 7) + ((cb ^ ((c >> 1) & 7)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += from_h16<BF16>((unsigned short)v[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = cs[j];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        cs[j] = v;
      }
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_red[wave * 64 + lane * 8 + j] = cs[j];
      }
      __syncthreads();
      if (tid < 64) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += s_red[k * 64 + tid];
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * 64 + tid] = v;
      }
    }
    H16_STAMP(4);     // channel sums
    // whole pixel rows leave as 16 bytes per lane: 8 rows x 32 px x 8 blocks = 2048 pieces
    char* ob = reinterpret_cast<char*>(a.out) + (size_t)bn * h * w * 128;
    for (int e = tid; e < HT_H * HT_W * 8; e += 512) {
      const int p = e >> 3, sb = e & 7;
      const int r = p / HT_W, c = p - r * HT_W;
      const int oy = ty * HT_H + r, ox = tx * HT_W + c;
      if (oy < h && ox < w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ost + (p << 7) + (sb << 4));
        const int lb = sb ^ ((c >> 1) & 7);
#ifdef EAVSR_H16_EXP_NO_STORE
        if (v[0] == 1.2345e-33f)
#endif
        *reinterpret_cast<f32x4*>(ob + ((size_t)oy * w + ox) * 128 + lb * 16) = v;
      }
    }
    // the staging tile sits where the interior of the patch was: restore zeros where the NEXT use of this stage
    // expects "never written" (handled by clear_border before each DMA; interior pieces are always rewritten)
    H16_STAMP(5);     // output stores
  }
#ifdef EAVSR_H16_STAMPS
  if (tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_h16_stamps[i], st_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// layout / precision converters and the 16-bit RCAB tail
// ---------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_h16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                                   int c, int hw) {
  // one workgroup: 64 pixels x all channels through LDS (transpose)
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, bn = blockIdx.y;
  for (int c0 = 0; c0 < c; c0 += 64) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int ch = e >> 6, p = e & 63;
      tile[ch][p] = (c0 + ch < c && p0 + p < hw) ? in[((size_t)bn * c + c0 + ch) * hw + p0 + p] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int p = e >> 6, ch = e & 63;
      if (c0 + ch < c && p0 + p < hw) out[((size_t)bn * hw + p0 + p) * c + c0 + ch] = to_h16<BF16>(tile[ch][p]);
    }
    __syncthreads();
  }
}

// out_f32_nchw = in_h16_nhwc (+ residual_f32_nchw)
template <bool BF16>
__global__ __launch_bounds__(256) void nhwc_h16_to_nchw_f32_kernel(const unsigned short* __restrict__ in,
                                                                   const float* __restrict__ residual,
                                                                   float* __restrict__ out, int c, int hw) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, bn = blockIdx.y;
  for (int c0 = 0; c0 < c; c0 += 64) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int p = e >> 6, ch = e & 63;
      tile[ch][p] = (c0 + ch < c && p0 + p < hw) ? from_h16<BF16>(in[((size_t)bn * hw + p0 + p) * c + c0 + ch]) : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int ch = e >> 6, p = e & 63;
      if (c0 + ch < c && p0 + p < hw) {
        const size_t o = ((size_t)bn * c + c0 + ch) * hw + p0 + p;
        out[o] = tile[ch][p] + (residual ? residual[o] : 0.f);
      }
    }
    __syncthreads();
  }
}

// NHWC 16-bit: out = r * scale[n, c] + x     (8 channels = 16 bytes per thread)
template <bool BF16>
__global__ __launch_bounds__(256) void scale_residual_h16_kernel(const s16x8* __restrict__ r, const float* __restrict__ scale,
                                                                 const s16x8* __restrict__ x, s16x8* __restrict__ out,
                                                                 long pieces_per_sample, int c8) {
  const int bn = blockIdx.y;
  const float* sc = scale + (size_t)bn * c8 * 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pieces_per_sample; i += (long)gridDim.x * 256) {
    const int cb = (int)(i % c8);
    const s16x8 rv = r[(size_t)bn * pieces_per_sample + i], xv = x[(size_t)bn * pieces_per_sample + i];
    s16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = (short)to_h16<BF16>(from_h16<BF16>((unsigned short)rv[j]) * sc[cb * 8 + j] + from_h16<BF16>((unsigned short)xv[j]));
    out[(size_t)bn * pieces_per_sample + i] = o;
  }
}

// weight (64, 64, 3, 3) fp32 -> [36 k-steps][2 halves][64 co][8] 16-bit;  k-step s = tap * 4 + cb, element j of
// half hf = input channel cb*16 + hf*8 + j
template <bool BF16>
__global__ void pack_weight_h16_kernel(const float* __restrict__ w, unsigned short* __restrict__ p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 576) return;
  const int j = i & 7, co = (i >> 3) & 63, hf = (i >> 9) & 1, s = i >> 10;
  const int tap = s >> 2, cb = s & 3;
  const int ci = cb * 16 + hf * 8 + j;
  p[i] = to_h16<BF16>(w[((size_t)co * 64 + ci) * 9 + tap]);
}

template <bool BF16>
int launch_conv_h16(const H16Args& a, int blocks, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_h16_kernel<BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_c64_h16: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv3x3_c64_h16_kernel<BF16>, dim3(blocks), dim3(512), H_LDS_BYTES, st, a);
  return eavsr::launch_status("conv3x3_c64_h16");
}

}  // namespace

#ifdef EAVSR_H16_STAMPS
extern "C" int eavsr_debug_h16_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_h16_stamps), sizeof(g_h16_stamps));
  if (reset) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_h16_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int32_t eavsr_conv_h16_tiles(int32_t h, int32_t w) { return eavsr::cdiv(h, HT_H) * eavsr::cdiv(w, HT_W); }

extern "C" int eavsr_pack_conv3x3_c64_h16(const float* weight, void* packed, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv3x3_c64_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "pack_conv3x3_c64_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  hipStream_t st = eavsr::as_stream(stream);
  if (dtype == 2)
    hipLaunchKernelGGL(pack_weight_h16_kernel<true>, dim3(144), dim3(256), 0, st, weight, (unsigned short*)packed);
  else
    hipLaunchKernelGGL(pack_weight_h16_kernel<false>, dim3(144), dim3(256), 0, st, weight, (unsigned short*)packed);
  return eavsr::launch_status("pack_conv3x3_c64_h16");
}

extern "C" int eavsr_conv3x3_c64_h16(const void* x, const void* weight_packed, const float* bias, void* out,
                                     float* chan_partial, int32_t n, int32_t h, int32_t w, int32_t relu,
                                     int32_t dtype, void* stream) {
  EAVSR_REQUIRE(x && weight_packed && out, -1, "conv3x3_c64_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv3x3_c64_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0, -1, "conv3x3_c64_h16: bad dims");
  EAVSR_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight_packed) & 15) == 0, -1,
                "conv3x3_c64_h16: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  H16Args a;
  a.x = x; a.wp = weight_packed; a.bias = bias; a.out = out; a.chan_partial = chan_partial;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, HT_W);
  a.tiles_y = eavsr::cdiv(h, HT_H);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "conv3x3_c64_h16: too many tiles");
  a.num_tiles = (int)tiles;
  a.relu = relu;
  const int blocks = tiles < 256 ? (int)tiles : 256;  // persistent: one workgroup per CU
  return dtype == 2 ? launch_conv_h16<true>(a, blocks, eavsr::as_stream(stream))
                    : launch_conv_h16<false>(a, blocks, eavsr::as_stream(stream));
}

extern "C" int eavsr_nchw_f32_to_nhwc_h16(const float* in, void* out, int32_t n, int32_t c, int32_t hw, int32_t dtype,
                                          void* stream) {
  EAVSR_REQUIRE(in && out, -1, "nchw_f32_to_nhwc_h16: NULL pointer");
  EAVSR_REQUIRE((dtype == 1 || dtype == 2) && n >= 0 && c > 0 && hw > 0 && n <= 65535, -1, "nchw_f32_to_nhwc_h16: bad args");
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(hw, 64), n);
  if (dtype == 2)
    hipLaunchKernelGGL(nchw_f32_to_nhwc_h16_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), in, (unsigned short*)out, c, hw);
  else
    hipLaunchKernelGGL(nchw_f32_to_nhwc_h16_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), in, (unsigned short*)out, c, hw);
  return eavsr::launch_status("nchw_f32_to_nhwc_h16");
}

extern "C" int eavsr_nhwc_h16_to_nchw_f32(const void* in, const float* residual, float* out, int32_t n, int32_t c,
                                          int32_t hw, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(in && out, -1, "nhwc_h16_to_nchw_f32: NULL pointer");
  EAVSR_REQUIRE((dtype == 1 || dtype == 2) && n >= 0 && c > 0 && hw > 0 && n <= 65535, -1, "nhwc_h16_to_nchw_f32: bad args");
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(hw, 64), n);
  if (dtype == 2)
    hipLaunchKernelGGL(nhwc_h16_to_nchw_f32_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), (const unsigned short*)in, residual, out, c, hw);
  else
    hipLaunchKernelGGL(nhwc_h16_to_nchw_f32_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), (const unsigned short*)in, residual, out, c, hw);
  return eavsr::launch_status("nhwc_h16_to_nchw_f32");
}

extern "C" int eavsr_scale_residual_h16(const void* r, const float* scale, const void* x, void* out, int32_t n,
                                        int32_t c, int32_t hw, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(r && scale && x && out, -1, "scale_residual_h16: NULL pointer");
  EAVSR_REQUIRE((dtype == 1 || dtype == 2) && n >= 0 && c > 0 && c % 8 == 0 && hw > 0 && n <= 65535, -1,
                "scale_residual_h16: bad args (c must be a multiple of 8)");
  if (n == 0) return 0;
  const long pieces = (long)hw * (c / 8);
  long bx = (pieces + 255) / 256;
  if (bx > 1024) bx = 1024;
  dim3 grid((unsigned)bx, n);
  if (dtype == 2)
    hipLaunchKernelGGL(scale_residual_h16_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), (const s16x8*)r, scale, (const s16x8*)x, (s16x8*)out, pieces, c / 8);
  else
    hipLaunchKernelGGL(scale_residual_h16_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), (const s16x8*)r, scale, (const s16x8*)x, (s16x8*)out, pieces, c / 8);
  return eavsr::launch_status("scale_residual_h16");
}
