// Generic 3x3 stride-1 "same" convolution for the 16-bit modes (BASELINE.json configs[2] bf16 / configs[4] fp16): fp32 NCHW in
// (a virtual channel concatenation of up to five sources), operands rounded ONCE to bf16 / fp16 on their way into LDS,
// v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation, fp32 NCHW out.  It takes the convolutions of the path that are NOT the
// 64 -> 64 NHWC backbone kernel (csrc/conv_h16.hip) when the caller has opted into a 16-bit mode: the first convolution of every
// residual backbone (models/eavsrp_model.py:375-381: 128 / 192 / 256 / 320 -> 64 over the concatenated feature lists), the
// encoder's 64 -> 64 / 64 -> 128 / 128 -> 128 / 128 -> 256 / 256 -> 64 layers (models/networks.py:522-552) -- in fp32 they run as
// F(4x4,3x3) on the fp32 matrix pipe.
//
// Workgroup: 512 threads, 16 rows x 32 pixels x 64 output channels (wave = 2 rows, 2 x 2 MFMA tiles), grid.y = 64-channel
// output tiles.  K loop over chunks of 16 input channels; k-step = (tap, chunk): the B operand of lane (pixel n = lane & 31,
// half = lane >> 5) is the 8 channels half * 8 .. of pixel n shifted by the tap = one ds_read_b128 from the chunk's patch, kept in
// LDS as two channel-half planes [half][18 rows][34 columns] x 16 bytes (consecutive lanes 16 bytes apart: no bank conflicts);
// the A operand comes from the packed weights ([cot][chunk][tap][half][64 co][8 ch], eavsr_pack_conv3x3_h16g) streamed by
// LDS-DMA, 18 KB per chunk.  Two stages of (patch + weights) = 74.5 KB: two workgroups per CU.
// Per chunk: ONE barrier; behind it the next chunk's weights are requested (LDS-DMA) and its patch loaded into registers
// (thread = one aligned 4-pixel quad x 8 channels: eight 16-byte loads, rounded and stored as four ds_write_b128 after this
// chunk's MFMAs), so global latency hides under the 36 MFMAs per wave.
#include "common.h"

#include <hip/hip_bf16.h>
#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int G_TH = 16, G_TW = 32, G_CO = 64, G_CK = 16;
constexpr int G_IH = G_TH + 2;                                  // patch rows y0-1 .. y0+16
constexpr int G_QW = 10;                                        // loaded as aligned quads: columns x0-4 .. x0+35
constexpr int G_IW = G_TW + 2;                                  // kept in LDS: columns x0-1 .. x0+32
constexpr int G_PLANE_B = G_IH * G_IW * 16;                     // one channel-half plane: 9,792 B
constexpr int G_PATCH_B = 2 * G_PLANE_B;                        // 19,584 B
constexpr int G_W_B = 9 * 2 * G_CO * 16;                        // one chunk's weights: 18,432 B = 18 one-KiB pieces
constexpr int G_STAGE_B = G_PATCH_B + G_W_B;                    // 38,016 B
constexpr int G_LDS_BYTES = 2 * G_STAGE_B + G_CO * 4;           // 76,288 B: two workgroups per CU
constexpr int G_UNITS = 2 * G_IH * G_QW;                        // (half, row, quad) load units of a chunk: 360

struct G16Args {
  const float* src[5];
  int src_c[5];
  int n_src;
  const void* wp;
  const float* bias;
  float* out;
  int n, h, w, cin, cout, tiles_x, tiles_y;
  int act;
  float slope;
};

template <bool BF16> __device__ __forceinline__ unsigned pack2_h16(float a, float b);
template <> __device__ __forceinline__ unsigned pack2_h16<true>(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(a)) |
         ((unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(b)) << 16);
}
template <> __device__ __forceinline__ unsigned pack2_h16<false>(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
}

template <bool BF16>
__device__ __forceinline__ f32x16 g_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  if (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

#ifndef EAVSR_H16G_WAVES
#define EAVSR_H16G_WAVES 4      // waves per SIMD the register budget is cut for: 4 = two workgroups per CU (128 registers, 9 spilled outside the loop)
#endif
template <bool BF16>
__global__ __launch_bounds__(512, EAVSR_H16G_WAVES) void conv3x3_h16g_kernel(G16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  float* s_bias = reinterpret_cast<float*>(gsm + 2 * G_STAGE_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y, bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * G_TH, x0 = tx * G_TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const int nch = a.cin / G_CK;

  // ---- patch producer: unit u = tid (< 360) = (channel half hf, row r, quad q): eight float4 loads, four ds_write_b128 ----
  const bool unit = tid < G_UNITS;
  const int u_hf = tid / (G_IH * G_QW), u_rem = tid - u_hf * (G_IH * G_QW);
  const int u_r = u_rem / G_QW, u_q = u_rem - u_r * G_QW;
  const int u_gy = y0 - 1 + u_r, u_gx = x0 - 4 + 4 * u_q;
  const bool u_ok = unit && u_gy >= 0 && u_gy < h && u_gx >= 0 && u_gx + 3 < w;      // (w % 4 == 0: a quad is inside or outside)
  const unsigned u_go = u_ok ? (unsigned)(u_gy * w + u_gx) : 0u;
  const int u_c0 = 4 * u_q - 3;                                    // LDS column of the quad's first pixel (x0 - 1 is column 0)
  const unsigned u_lo = (unsigned)(((u_hf * G_IH + u_r) * G_IW + u_c0) * 16);
  f32x4 pv[8];
  auto load_patch = [&](int ch) __attribute__((always_inline)) {
    // the source holding channels ch * 16 .. + 15 of the concatenation (every source's channel count is a multiple of 16)
    int c0 = ch * G_CK, s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (s == i && i + 1 < a.n_src && c0 >= a.src_c[i]) { c0 -= a.src_c[i]; s = i + 1; }
    const float* sp = (s == 0 ? a.src[0] : s == 1 ? a.src[1] : s == 2 ? a.src[2] : s == 3 ? a.src[3] : a.src[4]);
    const int sc = (s == 0 ? a.src_c[0] : s == 1 ? a.src_c[1] : s == 2 ? a.src_c[2] : s == 3 ? a.src_c[3] : a.src_c[4]);
    sp += ((size_t)bn * sc + c0 + u_hf * 8) * plane + u_go;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = u_ok ? *reinterpret_cast<const f32x4*>(sp + (size_t)j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_patch = [&](int stage) __attribute__((always_inline)) {
    if (unit) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        u32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = pack2_h16<BF16>(pv[2 * c][p], pv[2 * c + 1][p]);
        if ((unsigned)(u_c0 + p) < (unsigned)G_IW) *reinterpret_cast<u32x4*>(gsm + stage * G_STAGE_B + u_lo + p * 16) = v;
      }
    }
  };
  auto issue_w = [&](int ch, int stage) __attribute__((always_inline)) {
    const char* wsrc = reinterpret_cast<const char*>(a.wp) + ((size_t)cot * nch + ch) * G_W_B;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int seg = i * 8 + wave;
      if (seg < G_W_B / 1024)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + seg * 1024 + lane * 16), (lptr_t)(gsm + stage * G_STAGE_B + G_PATCH_B + seg * 1024), 16, 0, 0);
    }
  };

  // ---- prologue -----------------------------------------------------------------------------------------------------------
  load_patch(0);
  issue_w(0, 0);
  if (tid < G_CO) {
    const int co = cot * G_CO + tid;
    s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
  }
  store_patch(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 acc[2][2];      // [m][row]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float bv = s_bias[m * 32 + (e & 3) + 8 * (e >> 2) + 4 * half];
      acc[m][0][e] = bv;
      acc[m][1][e] = bv;
    }

  const int b_lane = ((half * G_IH + 2 * wave) * G_IW + l31) * 16;      // pixel (row 2 wave + t + ky, column l31 + kx)
  const int a_lane = (half * G_CO + l31) * 16;

  for (int ch = 0; ch < nch; ++ch) {
    const int st = ch & 1;
    const bool more = ch + 1 < nch;
    if (more) {
      load_patch(ch + 1);      // (the loads first: the counted wait below covers them and leaves the weight requests in flight)
      issue_w(ch + 1, st ^ 1);
    }
    const unsigned char* pb = gsm + st * G_STAGE_B + b_lane;
    const unsigned char* wb = gsm + st * G_STAGE_B + G_PATCH_B + a_lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      u32x4 av[2], bv[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) av[m] = *reinterpret_cast<const u32x4*>(wb + (tap * 2 * G_CO + m * 32) * 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const u32x4*>(pb + ((t + ky) * G_IW + kx) * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = g_mfma<BF16>(av[m], bv[t], acc[m][t]);
    }
    if (more) {
      // the patch of the next chunk: its eight loads were issued before the weight requests (3 per wave, 2 for waves 2..7)
      if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      store_patch(st ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: activation, fp32 NCHW stores (lanes 0-31 / 32-63: 32 consecutive pixels of two channels 4 apart) ---------
  const float act_s = a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope;
  const int gx = x0 + l31;
  if (gx < w) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int gy = y0 + 2 * wave + t;
      if (gy < h) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int co = cot * G_CO + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (co < a.cout) {
              const float v = acc[m][t][e];
              a.out[((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx] = eavsr_act(v, act_s);
            }
          }
      }
    }
  }
}

// weight (cout, cin, 3, 3) fp32 -> [cot][chunk][tap][half][64 co][8 ch] 16-bit (rows >= cout are zero)
template <bool BF16>
__global__ void pack_weight_h16g_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int cout, int cin, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), co = (int)((i >> 3) & 63), hf = (int)((i >> 9) & 1);
  long r = i >> 10;
  const int tap = (int)(r % 9);
  r /= 9;
  const int nch = cin / G_CK;
  const int ch = (int)(r % nch), cot = (int)(r / nch);
  const int o = cot * G_CO + co, ci = ch * G_CK + hf * 8 + j;
  const float v = o < cout ? w[((size_t)o * cin + ci) * 9 + tap] : 0.f;
  p[i] = BF16 ? __builtin_bit_cast(unsigned short, __float2bfloat16(v)) : __builtin_bit_cast(unsigned short, (_Float16)v);
}

template <bool BF16>
int launch_h16g(const G16Args& a, dim3 grid, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once_pd.flag[dev_], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h16g_kernel<BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_h16g: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(conv3x3_h16g_kernel<BF16>, grid, dim3(512), G_LDS_BYTES, st, a);
  return eavsr::launch_status("conv3x3_h16g");
}

}  // namespace

extern "C" int64_t eavsr_conv3x3_h16g_weight_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % G_CK) return -1;
  return (int64_t)eavsr::cdiv(cout, G_CO) * (cin / G_CK) * G_W_B;
}

extern "C" int eavsr_pack_conv3x3_h16g(const float* weight, void* packed, int32_t cout, int32_t cin, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv3x3_h16g: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "pack_conv3x3_h16g: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % G_CK == 0, -2, "pack_conv3x3_h16g: cin %d must be a multiple of 16", cin);
  const long total = eavsr_conv3x3_h16g_weight_bytes(cout, cin) / 2;
  hipStream_t st = eavsr::as_stream(stream);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == 2) hipLaunchKernelGGL(pack_weight_h16g_kernel<true>, dim3(blocks), dim3(256), 0, st, weight, (unsigned short*)packed, cout, cin, total);
  else hipLaunchKernelGGL(pack_weight_h16g_kernel<false>, dim3(blocks), dim3(256), 0, st, weight, (unsigned short*)packed, cout, cin, total);
  return eavsr::launch_status("pack_conv3x3_h16g");
}

extern "C" int eavsr_conv3x3_h16g_f32(const eavsr_conv2d_desc* d, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(d && d->out && d->weight_packed, -1, "conv3x3_h16g: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv3x3_h16g: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(d->ksize == 3 && d->n_src >= 1 && d->n_src <= 5, -2, "conv3x3_h16g: 3x3, 1..5 sources");
  EAVSR_REQUIRE(!d->residual && !d->chan_partial && !d->ca_scale && !d->ca_x && !d->ca_out && d->out_shuffle == 0 && !d->res_scale && !d->border_pieces && !d->sum_mul, -2,
                "conv3x3_h16g: plain convolution only (no residual / channel sums / channel-attention prologue / pixel shuffle)");
  EAVSR_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cout > 0 && d->w % 4 == 0, -2, "conv3x3_h16g: bad dims (w %% 4 == 0)");
  int cin = 0;
  G16Args a;
  for (int i = 0; i < 5; ++i) {
    a.src[i] = i < d->n_src ? d->src[i] : nullptr;
    a.src_c[i] = i < d->n_src ? d->src_c[i] : 0;
    if (i < d->n_src) {
      EAVSR_REQUIRE(d->src[i] && d->src_c[i] > 0 && d->src_c[i] % G_CK == 0, -2, "conv3x3_h16g: source %d: %d channels (multiples of 16)", i, d->src_c[i]);
      EAVSR_REQUIRE(((uintptr_t)d->src[i] & 15) == 0, -2, "conv3x3_h16g: sources must be 16-byte aligned");
      cin += d->src_c[i];
    }
  }
  EAVSR_REQUIRE(cin == d->cin, -1, "conv3x3_h16g: cin %d != sum of the sources %d", d->cin, cin);
  if (d->n == 0) return 0;
  a.n_src = d->n_src; a.wp = d->weight_packed; a.bias = d->bias; a.out = d->out;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = cin; a.cout = d->cout;
  a.tiles_x = eavsr::cdiv(d->w, G_TW);
  a.tiles_y = eavsr::cdiv(d->h, G_TH);
  a.act = d->act; a.slope = d->slope;
  const long blocks = (long)a.tiles_x * a.tiles_y * d->n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_h16g: too many tiles");
  dim3 grid((unsigned)blocks, (unsigned)eavsr::cdiv(d->cout, G_CO));
  return dtype == 2 ? launch_h16g<true>(a, grid, eavsr::as_stream(stream)) : launch_h16g<false>(a, grid, eavsr::as_stream(stream));
}
