// 3x3 stride-1 convolution, fp32 in / fp32 out, with the contraction carried by the bf16 matrix pipe ("bf16x9").
// Opt-in counterpart of conv2d_mfma_kernel<3, 2, true> (conv_mfma.hip) for the residual backbone
// (networks.py:456-458,478; eavsrp_model.py:381): same operands, same epilogue, same tensors in HBM.
//
// Arithmetic (as in dcnv2_x9.hip): every fp32 operand is split EXACTLY into three bf16 terms (hi = trunc_bf16(x),
// mid = trunc_bf16(x - hi), lo = x - hi - mid; hi + mid + lo == x bit for bit) and all nine partial products are
// accumulated in fp32.  Products of bf16 numbers are exact in fp32, so the only rounding is the accumulation -- as
// in the fp32 fma chain of the native kernel.  No operand is rounded to bf16.
//
// Why: v_mfma_f32_32x32x2_f32 does 2 k in 64 cycles; nine v_mfma_f32_32x32x16_bf16 do 16 k in 288.  The 3x3 64->64
// conv is matrix-bound (72 % of the forward), so its floor drops from 134 us to 84 us per launch at this tile.
//
// Structure: the native kernel's, with 16-deep k-steps.  Per workgroup (512 threads, 32 x 32 pixels, 64 output
// channels) and per chunk of 8 input channels, two LDS stages filled by 16-byte LDS-DMA: the fp32 input patch
// (8 x 34 x 40) and the PRE-SPLIT weight slab (5 k-steps x 3 planes x 2 M-tiles x 64 lanes x 16 B, written by
// eavsr_pack_dcn_weight_x9).  A k-step is (tap 2s, tap 2s+1) x 8 channels: lane (n = lane & 31, g = lane >> 5)
// reads the 8 channels of pixel n shifted by ITS tap (2s + g), splits them in registers (44 vector instructions per
// 18 MFMAs, hidden in their shadows: the bf16 MFMA holds the vector issue for 8 of its 32 cycles) and multiplies.
// Tap 9 does not exist: its weights are zero (10 % of the MFMAs are padding).
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct CxArgs {
  const float* src[5];
  int src_c[5];
  int n_src;
  const u32x4* wsplit;   // [cot][chunk of 8 channels][step][plane][mt][lane]
  const float* bias;
  const float* residual;
  float* out;
  float* chan_partial;
  int n, h, w, cin, cout, tiles_x, tiles_y;
  int act;
  float slope;
};

constexpr int CK = 8, NT = 4, NW = 8, TH = 32, TW = 32, PAD = 1, MARG = 4;
constexpr int IH = TH + 2, IW = TW + 2 * MARG;           // 34 x 40
constexpr int IN_ELEMS = CK * IH * IW;                   // 10,880 floats
constexpr int IN_SEGS = (IN_ELEMS + 255) / 256;          // 43 one-KiB pieces
constexpr int IN_PAD = IN_SEGS * 256;
constexpr int IN_IT = (IN_SEGS + NW - 1) / NW;
constexpr int XSTEPS = 5;
constexpr int W_U4 = XSTEPS * 3 * 2 * 64;                // 1920 16-byte elements = 30 one-KiB pieces
constexpr int W_SEGS = W_U4 / 64;
constexpr int W_IT = (W_SEGS + NW - 1) / NW;
constexpr int BUF = IN_PAD + W_U4 * 4;                   // floats per pipeline stage
constexpr int LDS_FLOATS = 2 * BUF + NW * 64;
constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void conv3x3_x9_kernel(CxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_red = smem + 2 * BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kgrp = lane >> 5;

  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  int cs = 0, cc0 = 0, cbase = 0;
  int total_chunks = 0;
  for (int s = 0; s < a.n_src; ++s) total_chunks += a.src_c[s] / CK;

  // per-lane byte offsets of this wave's patch pieces inside one 8-channel slab (0xFFFFFFFF: zero padding, never
  // moved -- the LDS words keep the zeros written here)
  unsigned voff[IN_IT];
  {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < (2 * BUF) / 4; e += 64 * NW) z[e] = zero;
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      const int e4 = seg * 64 + lane;
      const int ci = e4 / (IH * (IW / 4));
      const int rem = e4 - ci * (IH * (IW / 4));
      const int r = rem / (IW / 4);
      const int c4 = rem - r * (IW / 4);
      const int gy = y0 - PAD + r, gx = x0 - MARG + 4 * c4;
      const bool ok = seg < IN_SEGS && e4 < IN_ELEMS / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;
      voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)gy * w + gx) * 4) : 0xFFFFFFFFu;
    }
    __syncthreads();
  }

  auto issue_chunk = [&](int stage) {
    float* s_in = smem + stage * BUF;
    u32x4* s_w = reinterpret_cast<u32x4*>(s_in + IN_PAD);
    const int sc = a.src_c[cs];
    const float* sp = a.src[cs] + ((size_t)bn * sc + cc0) * plane;
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      if (voff[i] != 0xFFFFFFFFu)
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(sp) + voff[i]), (lptr_t)(s_in + seg * 256), 16, 0, 0);
    }
    const char* wsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)cot * (a.cin / CK) + (size_t)(cbase + cc0) / CK) * W_U4);
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = i * NW + wave;
      if (seg < W_SEGS)  // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)(seg * 64 + lane) * 16u), (lptr_t)(s_w + seg * 64), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    cc0 += CK;
    if (cc0 >= a.src_c[cs]) {
      cbase += a.src_c[cs];
      ++cs;
      cc0 = 0;
    }
  };

  issue_chunk(0);
  for (int it = 0; it < total_chunks; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + 1 < total_chunks) {
      advance();
      issue_chunk((it + 1) & 1);
    }
    const float* bin = smem + (it & 1) * BUF + (wave * NT) * IW + (MARG - PAD) + l31;
    const u32x4* win = reinterpret_cast<const u32x4*>(smem + (it & 1) * BUF + IN_PAD) + lane;
    // Flat software pipeline over the 20 (k-step, row) pairs of the chunk: the 8 LDS reads and the exact split of
    // pair i+1 are issued between the 18 MFMAs of pair i (v_mfma_f32_32x32x16_bf16 holds the vector issue for 8 of
    // its 32 cycles: 2-3 vector instructions fit every gap), so the wave's MFMA stream never waits for operands.
    auto tap_off = [&](int s_) __attribute__((always_inline)) {
      const int tap = min(2 * s_ + kgrp, 8);   // tap 9 (s = 4, kgrp = 1) has zero weights: any valid address
      const int ky = (tap * 11) >> 5;          // tap / 3 for tap in 0..8
      return ky * IW + (tap - 3 * ky);
    };
    auto read_row = [&](int off, int t, float (&v)[CK]) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < CK; ++c) v[c] = bin[off + c * (IH * IW) + t * IW];
    };
    auto split_row = [&](const float (&v)[CK], u32x4 (&bo)[3]) __attribute__((always_inline)) {
#ifdef EAVSR_CX9_EXP_NOSPLIT   // timing ablation only: results are wrong
      for (int c = 0; c < CK / 2; ++c) {
        bo[0][c] = __float_as_uint(v[c]); bo[1][c] = __float_as_uint(v[c + 4]); bo[2][c] = __float_as_uint(v[c]) ^ 1u;
      }
      return;
#endif
#pragma unroll
      for (int c = 0; c < CK / 2; ++c) {
        unsigned h2, m2, l2;
        split2(v[2 * c], v[2 * c + 1], h2, m2, l2);
        bo[0][c] = h2; bo[1][c] = m2; bo[2][c] = l2;
      }
    };
    auto load_a = [&](int s_, u32x4 (&wa)[3][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) wa[pl][mt] = win[(s_ * 3 + pl) * 2 * 64 + mt * 64];
    };
    // three stages in flight: LDS reads of pair i+2, split of pair i+1, MFMAs of pair i
    u32x4 wa[3][2], bcur[3];
    float vnext[CK];
    {
      float v0[CK];
      load_a(0, wa);
      read_row(tap_off(0), 0, v0);
      read_row(tap_off(0), 1, vnext);
      split_row(v0, bcur);
    }
#pragma unroll 1   // rolled over the k-steps: 128 accumulator registers leave room for one step's operands
    for (int s = 0; s < XSTEPS; ++s) {
      const int off_s = tap_off(s);
      const int sn = min(s + 1, XSTEPS - 1);       // after the last step the look-ahead re-reads it (unused)
      const int off_n = tap_off(sn);
      u32x4 wan[3][2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // pair i+2 = (s, t+2) or (s+1, t-2)
        float vfar[CK];
        if (t + 2 < NT) read_row(off_s, t + 2, vfar);
        else read_row(off_n, t + 2 - NT, vfar);
        if (t == NT - 1) load_a(sn, wan);
        u32x4 bnext[3];
        split_row(vnext, bnext);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x16 c_ = acc[mt][t];
          // the nine partial products, smallest terms first
          c_ = mfma_bf16(wa[2][mt], bcur[2], c_);
          c_ = mfma_bf16(wa[2][mt], bcur[1], c_);
          c_ = mfma_bf16(wa[1][mt], bcur[2], c_);
          c_ = mfma_bf16(wa[2][mt], bcur[0], c_);
          c_ = mfma_bf16(wa[0][mt], bcur[2], c_);
          c_ = mfma_bf16(wa[1][mt], bcur[1], c_);
          c_ = mfma_bf16(wa[1][mt], bcur[0], c_);
          c_ = mfma_bf16(wa[0][mt], bcur[1], c_);
          c_ = mfma_bf16(wa[0][mt], bcur[0], c_);
          acc[mt][t] = c_;
        }
        // placement: the far reads first, then every MFMA followed by 3 of the split's vector instructions
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bcur[pl] = bnext[pl];
#pragma unroll
        for (int c = 0; c < CK; ++c) vnext[c] = vfar[c];
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) wa[pl][mt] = wan[pl][mt];
    }
  }

  // ---- epilogue (as conv2d_mfma_kernel) --------------------------------------------------------
  const int gx = x0 + l31;
  const bool xok = gx < w;
  float csum[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
      const bool cok = co < a.cout;
      const float b = (cok && a.bias) ? a.bias[co] : 0.f;
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gy = y0 + wave * NT + t;
        float v = acc[m][t][r] + b;
        v = eavsr_act(v, a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope);   // branch-free: max(v, v s), 0 <= s <= 1
        if (cok && xok && gy < h) {
          const size_t o = ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx;
          sum += v;
          if (a.residual) v += a.residual[o];
          a.out[o] = v;
        }
      }
      csum[m][r] = sum;
    }
  }
  if (a.chan_partial) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = csum[m][r];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (l31 == 0) s_red[wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp] = v;
      }
    __syncthreads();
    if (tid < 64) {
      const int co = cot * 64 + tid;
      if (co < a.cout) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < NW; ++k) v += s_red[k * 64 + tid];
        const int tile = ty * a.tiles_x + tx;
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + tile) * a.cout + co] = v;
      }
    }
  }
}

}  // namespace

extern "C" int eavsr_conv3x3_f32x9(const eavsr_conv2d_desc* d, const void* weight_x9, void* stream) {
  EAVSR_REQUIRE(d != nullptr && weight_x9 != nullptr, -1, "conv3x3_f32x9: NULL descriptor / weights");
  EAVSR_REQUIRE(d->n_src >= 1 && d->n_src <= 5, -1, "conv3x3_f32x9: n_src %d not in 1..5", d->n_src);
  EAVSR_REQUIRE(d->ksize == 3, -2, "conv3x3_f32x9: kernel size %d (3 only)", d->ksize);
  EAVSR_REQUIRE(d->out, -1, "conv3x3_f32x9: NULL out");
  EAVSR_REQUIRE(d->out_shuffle == 0 && d->res_scale == nullptr && d->border_pieces == nullptr && d->sum_mul == nullptr, -2, "conv3x3_f32x9: the pixel-shuffle and scaled-residual epilogues exist in eavsr_conv3x3_wino4_f32 only");
  EAVSR_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, -1, "conv3x3_f32x9: bad dims");
  EAVSR_REQUIRE(d->act >= 0 && d->act <= 2, -1, "conv3x3_f32x9: act %d", d->act);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_LRELU || (d->slope >= 0.f && d->slope <= 1.f), -2,
                "conv3x3_f32x9: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)d->slope);
  EAVSR_REQUIRE(d->ca_scale == nullptr && d->ca_x == nullptr && d->ca_out == nullptr, -2,
                "conv3x3_f32x9: no fused channel-attention prologue");
  EAVSR_REQUIRE(d->w % 4 == 0, -2, "conv3x3_f32x9: w %% 4 != 0 (use eavsr_conv2d_f32)");
  CxArgs a;
  int csum = 0;
  for (int s = 0; s < 5; ++s) {
    a.src[s] = s < d->n_src ? d->src[s] : nullptr;
    a.src_c[s] = s < d->n_src ? d->src_c[s] : 0;
    if (s < d->n_src) {
      EAVSR_REQUIRE(d->src[s] != nullptr && d->src_c[s] > 0 && d->src_c[s] % CK == 0 && (((uintptr_t)d->src[s]) & 15) == 0, -2,
                    "conv3x3_f32x9: source %d must be 16-byte aligned with a multiple of 8 channels", s);
      csum += d->src_c[s];
    }
  }
  EAVSR_REQUIRE(csum == d->cin, -1, "conv3x3_f32x9: sources sum to %d channels, cin = %d", csum, d->cin);
  EAVSR_REQUIRE(eavsr_conv2d_tile_rows(d->n, d->h, d->w, 3) == TH, -2,
                "conv3x3_f32x9: exists for the 32-row tile only (this problem size runs shorter tiles)");
  if (d->n == 0) return 0;
  a.n_src = d->n_src;
  a.wsplit = reinterpret_cast<const u32x4*>(weight_x9);
  a.bias = d->bias; a.residual = d->residual; a.out = d->out; a.chan_partial = d->chan_partial;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
  a.tiles_x = eavsr::cdiv(d->w, TW);
  a.tiles_y = eavsr::cdiv(d->h, TH);
  a.act = d->act; a.slope = d->slope;
  const long blocks = (long)a.tiles_x * a.tiles_y * d->n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_f32x9: too many tiles");
  EAVSR_REQUIRE((long)d->h * d->w * 16 < (1L << 31), -1, "conv3x3_f32x9: image plane too large for 32-bit tile offsets");
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x9_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_f32x9: hipFuncSetAttribute(%zu B of LDS): %s", LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  dim3 grid((unsigned)blocks, eavsr::cdiv(d->cout, 64));
  hipLaunchKernelGGL(conv3x3_x9_kernel, grid, dim3(64 * NW), LDS_BYTES, eavsr::as_stream(stream), a);
  return eavsr::launch_status("conv3x3_f32x9");
}
