// 16-bit (bf16 / fp16) form of the predictor's three 5x5 heads (BASELINE.json configs[2] bf16 / configs[4] fp16; SURVEY.md 8a: a6).
//
// Reference: AdaptBlockOffset's transform_matrix_conv (64 -> 4 D), translation_conv (64 -> 2 D) and mask_conv (64 -> 9 D), all
// 5x5 / pad 2 on the same front-end feature (models/networks.py:283-285, 298-301), evaluated as ONE 64 -> 15 D convolution
// (D = 8: 120 output channels, padded to 128).  In fp32 this is the F(2x2,5x5) Winograd kernel (302 us per 2 x 180 x 320
// launch); in the 16-bit modes the operands are rounded once to bf16 / fp16 and the sum runs on the 16-bit matrix pipe with
// fp32 accumulation: 44 GFLOP per launch = 18 us at the dense peak.
//
//   v_mfma_f32_32x32x16_{bf16,f16}: A[row = co][k = 8 hf + j] (weights), B[k][col = pixel] -- as conv_h16.hip: a lane needs 8
//   consecutive input channels of its pixel at one tap = one ds_read_b128 from the NHWC patch (XOR-swizzled 16-byte blocks).
//   Workgroup = 8 waves, output tile 16 x 32 pixels x 128 channels; wave w owns rows 2w, 2w+1 (A operands shared by both rows):
//   per tap 4 channel blocks x 4 M-tiles x 2 rows = 32 MFMAs per wave for 16 + 8 ds_read_b128.
//   The 400 KB of weights do not fit the LDS: TWO taps (2 x 128 co x 64 ci x 2 B = 32 KB) per stage, two stages, LDS-DMA of the
//   next pair behind the MFMAs of this one, one barrier per pair of taps (13 per tile).  The 16-row tile halves the weight re-streaming per pixel (98 MB of
//   L2 -> LDS traffic per launch against 21 us of MFMA per tile).
//   Output: fp32 NCHW (n, cout, h, w) -- the "heads" operand of eavsr_dcnv2_il16 / eavsr_affine_offsets_f32.
#include "common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <mutex>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int FT_H = 16, FT_W = 32, FR = 5, FPAD = 2;
constexpr int FP_H = FT_H + 2 * FPAD, FP_W = FT_W + 2 * FPAD;   // 20 x 36 patch
constexpr int FP_PIX = FP_H * FP_W;                             // 720 pixels x 128 B = 92,160 B = 90 one-KiB pieces
constexpr int FP_SEGS = FP_PIX * 128 / 1024;
constexpr int FP_IT = (FP_SEGS + 7) / 8;                        // 12
constexpr int FCO = 128;                                        // output channels, padded
constexpr int FW_TAP = FCO * 64 * 2;                            // 16,384 B per tap
constexpr int FW_SEGS = FW_TAP / 1024;                          // 16 = 2 per wave
constexpr int FW_PAIR = 2 * FW_TAP;                             // a weight stage holds TWO taps: half as many barriers
constexpr int F_LDS_BYTES = FP_SEGS * 1024 + 2 * FW_PAIR;       // 157,696

struct F16Args {
  const void* x;      // (n, h, w, 64) 16-bit
  const void* wp;     // [25 taps][4 cb][2 halves][128 co][8] 16-bit
  const float* bias;  // fp32 [cout] or NULL
  float* out;         // (n, cout, h, w) fp32
  int n, h, w, cout, tiles_x, tiles_y, num_tiles;
};

template <bool BF16> __device__ __forceinline__ unsigned short f_to_h16(float v);
template <> __device__ __forceinline__ unsigned short f_to_h16<true>(float v) {
  return __builtin_bit_cast(unsigned short, __float2bfloat16(v));
}
template <> __device__ __forceinline__ unsigned short f_to_h16<false>(float v) {
  return __builtin_bit_cast(unsigned short, (_Float16)v);
}

template <bool BF16>
__device__ __forceinline__ f32x16 f_mfma(const f32x4& a, const f32x4& b, const f32x16& c) {
  if (BF16) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
}

template <bool BF16>
__global__ __launch_bounds__(512, 2) void conv5x5_c64_h16_kernel(F16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  unsigned char* s_p = fsm;                          // the patch, [pixel][8 swizzled 16-byte blocks]
  unsigned char* s_w = fsm + FP_SEGS * 1024;         // two weight stages (one tap each)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  eavsr_stagger_priority(wave);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = a.h, w = a.w;

  // taps 2 pr, 2 pr + 1 (the last pair is the 25th tap alone) into stage `stage`
  auto issue_pair = [&](int pr, int stage) {
    const char* src = reinterpret_cast<const char*>(a.wp) + (size_t)pr * FW_PAIR;
    const int segs = 2 * pr + 1 < FR * FR ? 2 * FW_SEGS : FW_SEGS;
#pragma unroll
    for (int i = 0; i < 2 * FW_SEGS / 8; ++i) {
      const int seg = i * 8 + wave;
      if (seg < segs)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + seg * 1024 + lane * 16), (lptr_t)(s_w + stage * FW_PAIR + seg * 1024), 16, 0, 0);
    }
  };

  for (int tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int bn = t / a.tiles_y;
    const int y0 = ty * FT_H - FPAD, x0 = tx * FT_W - FPAD;
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)bn * h * w * 128;

    __syncthreads();       // the previous tile's last tap has been read by every wave: patch and weight stages are free
    // patch: slot e -> pixel e >> 3, stored block e & 7 holds the logical block (e & 7) ^ swz(col); outside the image: zeros
#pragma unroll 1
    for (int i = 0; i < FP_IT; ++i) {
      const int seg = i * 8 + wave;
      if (seg < FP_SEGS) {
        const int e = seg * 64 + lane;
        const int p = e >> 3, sb = e & 7;
        const int r = p / FP_W, c = p - r * FP_W;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
          const int lb = sb ^ ((c >> 1) & 7);
          __builtin_amdgcn_global_load_lds((gptr_t)(xb + ((size_t)gy * w + gx) * 128 + lb * 16), (lptr_t)(s_p + seg * 1024), 16, 0, 0);
        } else {
          *reinterpret_cast<f32x4*>(s_p + e * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    issue_pair(0, 0);

    // the bias is the accumulators' initial value, through LDS (weight stage 1, free until the second pair of taps is requested
    // behind the first barrier): 128 global loads per lane in the epilogue, one in front of every store, were a third of a tile's time
    float* s_bias = reinterpret_cast<float*>(s_w + FW_PAIR);
    if (tid < FCO) s_bias[tid] = (a.bias && tid < a.cout) ? a.bias[tid] : 0.f;
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float bv = s_bias[m * 32 + (e & 3) + 8 * (e >> 2) + 4 * half];
        acc[0][m][e] = bv;
        acc[1][m][e] = bv;
      }

#pragma unroll 1
    for (int tap = 0; tap < FR * FR; ++tap) {
      const int pr = tap >> 1, stage = pr & 1;
      if ((tap & 1) == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();           // this pair's weights (and the patch) landed; the other weight stage has been read
        if (2 * pr + 2 < FR * FR) issue_pair(pr + 1, stage ^ 1);
      }
      const int ky = tap / FR, kx = tap - FR * ky;
      const int c = l31 + kx;
      const int swz = (c >> 1) & 7;
      const unsigned char* prow0 = s_p + (((2 * wave + ky) * FP_W + c) << 7);
      const unsigned char* prow1 = prow0 + (FP_W << 7);
      const unsigned char* wst = s_w + stage * FW_PAIR + (tap & 1) * FW_TAP;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int boff = ((cb * 2 + half) ^ swz) << 4;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(prow0 + boff);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(prow1 + boff);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(wst + ((((cb * 2 + half) * FCO) + m * 32 + l31) << 4));
          acc[0][m] = f_mfma<BF16>(af, b0, acc[0][m]);
          acc[1][m] = f_mfma<BF16>(af, b1, acc[1][m]);
        }
      }
    }

    // epilogue: fp32 NCHW stores (32 consecutive pixels of one channel per half wave).  The lane's column goes through an empty
    // asm so that the address arithmetic stays HERE: hoisted above the tap loop its 128 64-bit results spilled, and every store
    // sat behind a scratch load (a third of a tile's time).  One base per row, the channel advances by pointer increments.
    const size_t plane = (size_t)h * w;
    int gx = tx * FT_W + l31;
    asm volatile("" : "+v"(gx));
    const bool full = a.cout == FCO;      // wave-uniform
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int gy = ty * FT_H + 2 * wave + r;
      if (gy < h && gx < w) {
        float* ob = a.out + ((size_t)bn * a.cout + 4 * half) * plane + (size_t)gy * w + gx;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int cu = m * 32 + (e & 3) + 8 * (e >> 2);      // channel cu + 4 half
            if (full || cu + 4 * half < a.cout) *ob = acc[r][m][e];
            ob += ((e & 3) == 3 ? 5 : 1) * plane;                  // one plane on, five across a group of four
          }
      }
    }
  }
}

// weight (cout, 64, 5, 5) fp32 -> [25 taps][4 cb][2 halves][128 co][8] 16-bit (rows >= cout are zero);  element j of half hf of
// block cb = input channel cb * 16 + hf * 8 + j
template <bool BF16>
__global__ void pack_weight5_h16_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int cout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 25 * 4 * 2 * FCO * 8) return;
  const int j = i & 7, co = (i >> 3) & (FCO - 1), hf = (i >> 10) & 1, cb = (i >> 11) & 3, tap = i >> 13;
  const int ci = cb * 16 + hf * 8 + j;
  p[i] = co < cout ? f_to_h16<BF16>(w[((size_t)co * 64 + ci) * 25 + tap]) : (unsigned short)0;
}

template <bool BF16>
int launch_conv5_h16(const F16Args& a, int blocks, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once_pd.flag[dev_], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_c64_h16_kernel<BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv5x5_c64_h16: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv5x5_c64_h16_kernel<BF16>, dim3(blocks), dim3(512), F_LDS_BYTES, st, a);
  return eavsr::launch_status("conv5x5_c64_h16");
}

}  // namespace

extern "C" int64_t eavsr_conv5x5_c64_h16_weight_bytes(void) { return (int64_t)25 * FW_TAP; }

extern "C" int eavsr_pack_conv5x5_c64_h16(const float* weight, void* packed, int32_t cout, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv5x5_c64_h16: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cout <= FCO, -2, "pack_conv5x5_c64_h16: cout %d unsupported (1..%d)", cout, FCO);
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "pack_conv5x5_c64_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  hipStream_t st = eavsr::as_stream(stream);
  const int blocks = 25 * 4 * 2 * FCO * 8 / 256;
  if (dtype == 2)
    hipLaunchKernelGGL(pack_weight5_h16_kernel<true>, dim3(blocks), dim3(256), 0, st, weight, (unsigned short*)packed, cout);
  else
    hipLaunchKernelGGL(pack_weight5_h16_kernel<false>, dim3(blocks), dim3(256), 0, st, weight, (unsigned short*)packed, cout);
  return eavsr::launch_status("pack_conv5x5_c64_h16");
}

extern "C" int eavsr_conv5x5_c64_h16(const void* x, const void* weight_packed, const float* bias, float* out, int32_t n, int32_t h,
                                     int32_t w, int32_t cout, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(x && weight_packed && out, -1, "conv5x5_c64_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv5x5_c64_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0, -1, "conv5x5_c64_h16: bad dims");
  EAVSR_REQUIRE(cout > 0 && cout <= FCO, -2, "conv5x5_c64_h16: cout %d unsupported (1..%d)", cout, FCO);
  EAVSR_REQUIRE((((uintptr_t)x | (uintptr_t)weight_packed) & 15) == 0, -1, "conv5x5_c64_h16: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  F16Args a;
  a.x = x; a.wp = weight_packed; a.bias = bias; a.out = out;
  a.n = n; a.h = h; a.w = w; a.cout = cout;
  a.tiles_x = eavsr::cdiv(w, FT_W);
  a.tiles_y = eavsr::cdiv(h, FT_H);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "conv5x5_c64_h16: too many tiles");
  a.num_tiles = (int)tiles;
  const int blocks = tiles < 256 ? (int)tiles : 256;   // persistent: one workgroup per CU
  return dtype == 2 ? launch_conv5_h16<true>(a, blocks, eavsr::as_stream(stream))
                    : launch_conv5_h16<false>(a, blocks, eavsr::as_stream(stream));
}
