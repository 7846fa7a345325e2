// 3x3 stride-1 convolution by Winograd F(4x4, 3x3) on the fp32 matrix cores: 4x fewer multiplications than the
// direct sum, 1.78x fewer than F(2x2, 3x3) (conv_wino.hip).  Same operands / epilogue / tensors as
// conv3x3_wino_kernel<false>; the residual backbone's convolutions (networks.py:456-458,478; eavsrp_model.py:381).
//
//   Y(4x4) = A^T [ (G g G^T) .* (B^T d B) ] A      per 6x6 input tile d, 3x3 filter g   (Lavin & Gray 2016, points 0, +-1, +-2, inf)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// 36 independent GEMMs (one per transform-domain position xi): M_xi[co, t] = sum_ci U_xi[co, ci] V_xi[ci, t], all fp32.
// The larger transform costs accuracy: ~1e-5 of the output scale against 4e-7 for F(2x2, 3x3) (tests/test_hip_ops.py
// measures both against fp64) - still fp32 arithmetic throughout and two orders inside the path's 1e-3 parity bound.
//
// Per workgroup (512 threads = 8 waves, one per CU): 8 x 64 output pixels = 2 x 16 Winograd tiles, 64 output channels.
//   wave w owns output channels 16 (w >> 1) .. +15 and tile row (w & 1): 36 accumulator quads = 144 registers
//   (v_mfma_f32_16x16x4_f32: lane (kq, l15) holds M_xi[16 cb + 4 kq + r][tile l15] for every xi), so the output
//   transform needs no exchange between lanes or waves.
//   per chunk of 4 input channels (= one k-step; two LDS stages each for the patch, U and V; one barrier per chunk):
//     LDS-DMA: the (4 x 10 x 72) fp32 input patch (lanes outside the image copy zeros) and the pre-transformed weight
//              slab U[36][4][64] (eavsr_pack_conv_weight_wino4)
//     input transform: the 128 (channel, tile) jobs of a chunk are the lanes of two waves; the pair on duty rotates with
//              the chunk; both passes (packed fp32 pairs) stay in registers
//     GEMM: 18 x (2 ds_read_b64 + 2 MFMA): U and V keep the positions in pairs, columns XOR-swizzled (^ 16 (c & 1)), so
//           that the operand reads run at the 256 B/clk of ds_read_b64 without bank conflicts
//   epilogue in registers: output transform, + bias, activation, + residual, float4 row stores (one 256-byte line per
//   16 lanes), optional per-tile channel sums (deterministic).
#include "common.h"

// -DEAVSR_W4_REG_WEIGHTS: the weight operands straight from global memory into registers (buffer loads, re-requested for the
// next chunk right after the MFMAs that consumed them) instead of the slab through LDS-DMA + ds_read_b64.  Bit-identical; 5 %
// faster back to back on hot weights (47.6 -> 45.2 us), 5.5 % SLOWER in the step (262.6 -> 277.2 ms on one box), where every
// launch meets its 590 KB of weights cold and each is requested by two waves per CU instead of once: off by default
// (tools/build_wino4_diag.sh builds both).
#ifdef EAVSR_W4_REG_WEIGHTS
#define EAVSR_W4_UREGS 1
#endif
// Round 3 (per-wave stamps, tools/gpu_wino4_itstamp.py): the pair on duty has the longest path of an iteration -- its DMA pieces
// (every wave that requests pieces at the same time as others is held ~1,100-1,250 cycles: the CU moves one 1-KiB piece per
// ~45 cycles whoever issues it), then the transform (1,300 cycles with the SIMD's priority, 2,950 without), then its 36 MFMAs.
// EAVSR_W4_DUTY_NO_U: it requests no weight pieces (the other six waves take six each) and its patch pieces behind the
// transform; EAVSR_W4_DUTYPRIO: it runs the transform at priority 3.  46.8 -> 40.6 us per 2 x 64 x 180 x 320 launch together
// with the re-paired positions (w6_slot) and the branch-free activation.  -DEAVSR_W4_ROUND2_DUTY builds the A/B reference.
#ifndef EAVSR_W4_ROUND2_DUTY
#define EAVSR_W4_DUTY_NO_U 1
#define EAVSR_W4_DUTYPRIO 1
#endif

#include <mutex>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace {

// the grouped schedule is the default; EAVSR_W4_GRP=0 keeps the duty-pair schedule (A/B switch), read once
static bool w6_grouped_schedule() {
#ifdef EAVSR_W4_NOGRP      // diagnostic builds (tools/build_wino4_diag.sh)
  return false;
#else
  static const bool on = [] { const char* e = getenv("EAVSR_W4_GRP"); return e == nullptr || atoi(e) != 0; }();
  return on;
#endif
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) float g_wino4_zero[4];   // source of the zero-padding DMA lanes

// Buffer addressing for the grouped kernel's epilogue (round 6; as csrc/dcnv2_il2.hip since round 5): a 128-bit resource per
// (image, 64-channel block) of `out` / `residual`, ONE per-lane 32-bit offset register per tile, a scalar offset per (channel,
// row) computed at the use.  The hardware's range check replaces the exec-mask regions: a lane right of the image holds W6_OOB
// (its loads return zeros, its stores are dropped), a lane whose channel is >= cout lies beyond the resource's extent.
typedef __amdgpu_buffer_rsrc_t w6_rsrc;
constexpr unsigned W6_OOB = 0x80000000u;      // beyond every extent the launcher accepts (64 channels x plane x 4 B < 2^31)
__device__ __forceinline__ w6_rsrc w6_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 w6_ld4(w6_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// The store's offset goes through the VECTOR offset (one v_add per store), never through the scalar one.  A 128-bit buffer store
// reads its data registers over the cycles after it issues, and a vector instruction that rewrites them too early corrupts the
// store.  The compiler pads that hazard -- except for stores with a REGISTER scalar offset, which older parts exempt; gfx950 does
// not honour the exemption: with `soffset` in a register the next channel's results landed in the previous channel's rows for the
// last lanes of each row segment, run-dependent (found by tools/dbg/rsc_diff.py against the round-5 library; one wait state,
// DESIGN.md 3.2).  With an immediate scalar offset the hazard recogniser does its job.
__device__ __forceinline__ void w6_st4(w6_rsrc r, unsigned voff, unsigned off, f32x4 v) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff + off, 0, 0);
}

struct W4Args {
  const float* src[5];
  int src_c[5];
  int n_src;
  const float* wu;        // [cot][cin / 4][18][4][64][2]
  const float* bias;
  const float* residual;
  float* out;
  float* chan_partial;
  const float* ca_scale;  // FUSE: effective input = src[0] * ca_scale[n, c] + ca_x (RCABlock tail, networks.py:447,463-464)
  const float* ca_x;
  float* ca_out;          // FUSE: optional copy of the effective input (the next block's residual stream)
  int n, h, w, cin, cout, tiles_x, tiles_y;
  int act;
  float slope;
  int out_shuffle;        // 2: out is F.pixel_shuffle(conv, 2), (n, cout / 4, 2h, 2w); 0: (n, cout, h, w)
  const float* res_scale; // (n, cout) or NULL: out = residual + res_scale[n][co] * act(conv + bias) (3x3; the RCAB tail as the epilogue)
  float* border;          // NULL, or [n][4][border_stride][64]: sums of the OUTPUT's four border lines per border tile (desc.border_pieces)
  int border_stride;
};

constexpr int CK = 4, NW = 8, NPOS = 36;
constexpr int MARG = 4;                               // patch starts 4 columns left of the tile: 16-byte DMA pieces
constexpr int U_ELEMS = NPOS * CK * 64;               // 9216 floats = 36 KB: one 1-KiB piece per position pair row
constexpr int U_SEGS = U_ELEMS / 256;                 // 36
constexpr int U_IT = (U_SEGS + NW - 1) / NW;          // 5
constexpr int V_ELEMS = NPOS * CK * 32;               // 4608 floats = 18 KB

// Transform-domain position (row i, column q) of the 6 x 6 block -> its slot in U, V and the accumulators.  Slots are consumed in
// PAIRS (one ds_read_b64 per operand and two positions); a row's pairs are the columns (1, 2), (3, 4), (0, 5) -- the pairs the
// row pass of the input transform produces as packed fp32 registers: (x3, x4) - 4 (x1, x2) = (b, a) and (x3, x4) - (x1, x2) =
// (e, c) are ONE packed instruction each and (a + b, a - b), (c + 2 e, c - 2 e) come out as adjacent registers, ready for their
// 8-byte LDS store (with the natural pairing (0, 1), (2, 3), (4, 5) the row pass was 14 scalar instructions + moves per row).
__host__ __device__ constexpr int w6_pair_of(int q) { return (q == 1 || q == 2) ? 0 : (q == 3 || q == 4) ? 1 : 2; }
__host__ __device__ constexpr int w6_half_of(int q) { return (q == 1 || q == 3 || q == 0) ? 0 : 1; }
__host__ __device__ constexpr int w6_slot(int i, int q) { return 2 * (3 * i + w6_pair_of(q)) + w6_half_of(q); }
// inverse: column of (pair within the row, half)
__host__ __device__ constexpr int w6_col_of(int pp, int half) { return pp == 0 ? 1 + half : pp == 1 ? 3 + half : (half ? 5 : 0); }

// R = 3: F(4x4, 3x3), output tile 8 x 64 px.  R = 5: F(2x2, 5x5) - the same 6 x 6 input tiles, points and B^T, 2 x 2
// outputs per tile (2.78x fewer multiplications than the direct 5x5 sum), output tile 4 x 32 px: the predictor's
// 5x5 offset / mask heads (networks.py:283-285).
template <int R>
struct WCfg {
  static_assert(R == 3 || R == 5, "F(4x4,3x3) or F(2x2,5x5)");
  static constexpr int M = 7 - R, PADR = R / 2;          // outputs per tile side, zero padding of the convolution
  static constexpr int TOH = 2 * M, TOW = 16 * M;         // output tile: 2 x 16 Winograd tiles
  static constexpr int IH = TOH + 2 * PADR, IW = TOW + 2 * MARG;   // 10 x 72 | 8 x 40
  static constexpr int IN_ELEMS = CK * IH * IW;           // 2880 | 1280 floats
  static constexpr int IN_SEGS = (IN_ELEMS / 4 + 63) / 64;   // 12 | 5 one-KiB pieces
  static constexpr int IN_PAD = IN_SEGS * 256;
  static constexpr int IN_IT = (IN_SEGS + NW - 1) / NW;   // 2 | 1
  static constexpr int OFF_U = 2 * IN_PAD;                // LDS map: [patch 0][patch 1][U 0][U 1][V 0][V 1][channel sums]
  static constexpr int OFF_V = OFF_U + 2 * U_ELEMS;
  static constexpr int LDS_MAIN = OFF_V + 2 * V_ELEMS;    // 132 KB | 118 KB
#ifdef EAVSR_W4_ITSTAMP
  static constexpr int LDS_FLOATS = LDS_MAIN + 128 + 512;   // + 256 shader-clock stamps (diagnostic build)
#else
  static constexpr int LDS_FLOATS = LDS_MAIN + 128;
#endif
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

// t = B^T d for one 6-vector (works on packed pairs: two columns / rows at once)
template <typename T>
__device__ __forceinline__ void in1d(const T (&d)[6], T (&t)[6]) {
  const T a = d[4] - 4.f * d[2];
  const T b = d[3] - 4.f * d[1];
  const T c = d[4] - d[2];
  const T e = d[3] - d[1];
  t[0] = 4.f * d[0] + (d[4] - 5.f * d[2]);
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + 2.f * e;
  t[4] = c - 2.f * e;
  t[5] = 4.f * d[1] + (d[5] - 5.f * d[3]);
}

// s = A^T m for one 6-vector
__device__ __forceinline__ void out1d(const float (&m)[6], float (&s)[4]) {
  const float p1 = m[1] + m[2], p2 = m[1] - m[2], p3 = m[3] + m[4], p4 = m[3] - m[4];
  s[0] = m[0] + p1 + p3;
  s[1] = p2 + 2.f * p4;
  s[2] = p1 + 4.f * p3;
  s[3] = (p2 + 8.f * p4) + m[5];
}

// FUSE (3x3 only): the channel-attention tail of the previous residual block is applied in the input transform - the
// patch of r and the patch of x are both staged ([r 0][r 1][x 0][x 1][U ..]), d = r * scale[n, c] + x, and the interior
// 4 x 4 of every tile's d goes to ca_out (the next block's residual stream).
#ifdef EAVSR_W4_STAMPS
// diagnostic build only: shader cycles per phase, summed over waves 0, 2, 4, 6 of every workgroup (tools/gpu_wino4_ablate.py)
__device__ unsigned long long g_w4_stamps[32];
#define W4_STAMP(i)                                                   \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#endif

#ifdef EAVSR_W4_TIMELINE
// diagnostic build only: wall-clock (s_memrealtime, 100 MHz) and shader-cycle (s_memtime) stamps of wave 0 of every workgroup
// at six points of the kernel (tools/gpu_wino4_timeline.py); six stamps per workgroup do not move the timing
__device__ unsigned long long g_w4_tl[512 * 16];
#define W4_TL(i)                                                                           \
  do {                                                                                     \
    if (tid == 0) {                                                                        \
      g_w4_tl[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 2 * (i)] = __builtin_amdgcn_s_memrealtime(); \
      g_w4_tl[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memtime(); \
    }                                                                                      \
  } while (0)
#else
#define W4_TL(i) do { } while (0)
#endif

#ifdef EAVSR_W4_ITSTAMP
// diagnostic build only: shader-clock stamps of EVERY wave at eight points of iterations 8..11 (one full rotation of the duty
// pair), kept in LDS and dumped at the end of the kernel: the anatomy of a steady-state iteration without instrumenting the rest
__device__ unsigned long long g_w4_it[256 * 256];
#define W4_IT(i)                                                                                       \
  do {                                                                                                 \
    if (it >= 8 && it < 12 && lane == 0) s_its[((it - 8) * 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define W4_IT(i) do { } while (0)
#endif

// GRP ("grouped", R = 3 without the fused prologue, an even number of 4-channel chunks): the input transform leaves the iterations.
// The chunks are taken two at a time: a transform phase in which waves 0-3 -- one per SIMD, nothing beside them -- transform both
// chunks (128 jobs each) while waves 4-7 request the second chunk's weight slab, then two pure GEMM iterations.  An fp32 MFMA
// holds its SIMD's vector ALU for all of its 32 cycles (tools/ubench/mfma_fill.hip), so a transform beside MFMAs only ever adds
// to ONE pair of SIMDs what the barrier then makes everybody wait for; as a phase of its own it costs every SIMD the same.
// RSC: the instantiation whose epilogue is out = residual + res_scale[n][co] * act(conv + bias) (the RCAB tail; the grouped 3x3
// kernel only) -- a kernel of its own so that the plain one's code and registers are exactly what they were
template <int R, bool FUSE = false, bool GRP = false, bool RSC = false>
__global__ __launch_bounds__(512, 2) void conv_wino6_kernel(W4Args a) {
  static_assert(!RSC || (GRP && R == 3 && !FUSE), "the scaled-residual epilogue exists for the grouped 3x3 kernel");
  static_assert(!FUSE || R == 3, "the fused channel-attention prologue exists for the 3x3 kernel");
  static_assert(!GRP || (R == 3 && !FUSE), "the grouped schedule exists for the plain 3x3 kernel");
  using C = WCfg<R>;
  constexpr int M = C::M, PADR = C::PADR, TOH = C::TOH, TOW = C::TOW, IH = C::IH, IW = C::IW, IN_ELEMS = C::IN_ELEMS;
  constexpr int IN_SEGS = C::IN_SEGS, IN_PAD = C::IN_PAD, IN_IT = C::IN_IT;
  constexpr int X_OFF = 2 * IN_PAD;                                   // FUSE: the x patch stages follow the r stages
  constexpr int OFF_U = C::OFF_U + (FUSE ? 2 * IN_PAD : 0), OFF_V = C::OFF_V + (FUSE ? 2 * IN_PAD : 0);
  constexpr int LDS_MAIN = C::LDS_MAIN + (FUSE ? 2 * IN_PAD : 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem + OFF_V;
  float* s_red = smem + LDS_MAIN;
#ifdef EAVSR_W4_ITSTAMP
  unsigned long long* s_its = reinterpret_cast<unsigned long long*>(smem + LDS_MAIN + 128);
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  W4_TL(0);   // kernel entry
#ifndef EAVSR_W4_UREGS
  // the first weight slab does not depend on the tile: requested before the set-up arithmetic (the set-up is 1.9 us of
  // dependent scalar loads and index arithmetic per workgroup: tools/gpu_wino4_timeline.py)
  {
    const char* usrc0 = reinterpret_cast<const char*>(a.wu + (size_t)blockIdx.y * (a.cin / CK) * U_ELEMS) + wave * 1024;
#pragma unroll
    for (int i = 0; i < U_IT; ++i) {
      const int seg = i * NW + wave;
      if (seg < U_SEGS)
        __builtin_amdgcn_global_load_lds((gptr_t)(usrc0 + i * (NW * 1024) + (tid & 63) * 16u), (lptr_t)(smem + OFF_U + seg * 256), 16, 0, 0);
    }
  }
#endif

  // Persistent workgroups: workgroup b walks the tiles b, b + gridDim.x, ... as ONE flattened sequence of (tile, chunk)
  // iterations (as conv3x3_wino_kernel): the LDS-DMA stream never drains at a tile boundary.
  const int cot = blockIdx.y;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const int total_tiles = a.tiles_x * a.tiles_y * a.n;
  const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  // (unsigned divisions: the set-up in front of the first request is ~500 executed scalar instructions, a third of them the
  // sign handling and fix-ups of signed divisions)
  const unsigned u_tx = (unsigned)a.tiles_x, u_ty = (unsigned)a.tiles_y;
  auto tile_coords = [&](int k, int& bn_, int& y0_, int& x0_, int& lin_) __attribute__((always_inline)) {
    unsigned t = (unsigned)eavsr_xcd_remap((int)blockIdx.x + k * (int)gridDim.x, total_tiles);
    const unsigned q1 = t / u_tx;
    const unsigned tx_ = t - q1 * u_tx;
    const unsigned q2 = q1 / u_ty;
    const unsigned ty_ = q1 - q2 * u_ty;
    bn_ = (int)q2;
    y0_ = (int)ty_ * TOH;
    x0_ = (int)tx_ * TOW;
    lin_ = (int)(ty_ * u_tx + tx_);
  };

  // acc[xi]: M_xi of output channels 16 cb + 4 kq + r, Winograd tile (row tg, column l15)
  const int l15 = lane & 15, kq = lane >> 4;
  const unsigned lane16 = lane * 16u;
  const int cb = wave >> 1, tg = wave & 1;
  eavsr_stagger_priority(wave);      // common.h: 47.8 -> 43.9 us per 2 x 64 x 180 x 320 convolution
#ifdef EAVSR_W4_DUTYPRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  float bias_r[4];   // this lane's four output channels are the same for every tile of the launch
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = cot * 64 + cb * 16 + 4 * kq + r;
    bias_r[r] = (co < a.cout && a.bias != nullptr) ? a.bias[co] : 0.f;
  }
  f32x4 acc[NPOS];
#pragma unroll
  for (int x = 0; x < NPOS; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  // activation without a branch per value: max(t, t s) with s = 1 (none), 0 (ReLU), slope (leaky ReLU, 0 <= slope <= 1; the
  // launcher rejects other slopes).  With `if (act == ..)` per value the epilogue carried 128 scalar compare-and-branch pairs:
  // 15.2 K -> 11.2 K cycles per tile, 46.8 -> 43.7 us per 2 x 64 x 180 x 320 launch (bit-identical for finite values)
  const float act_s = a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope;

  int total_chunks = 0;
  for (int s = 0; s < a.n_src; ++s) total_chunks += a.src_c[s] / CK;
  const int total_iters = my_tiles * total_chunks;

  // ---- prefetch stream (runs up to two chunks ahead of the compute stream, across tile boundaries) -------------
  int p_k = 0, p_cs = 0, p_cc0 = 0, p_bn = 0, p_y0 = 0, p_x0 = 0, p_lin = 0;
  unsigned voff[IN_IT];   // per-lane byte offsets of this wave's patch pieces; 0xFFFFFFFF: zero padding
  auto p_setup_lanes = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      const int e4 = seg * 64 + lane;
      const int ci = e4 / (IH * (IW / 4));
      const int rem = e4 - ci * (IH * (IW / 4));
      const int r = rem / (IW / 4);
      const int c4 = rem - r * (IW / 4);
      const int gy = p_y0 - PADR + r, gx = p_x0 - MARG + 4 * c4;
      const bool ok = e4 < IN_ELEMS / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;
      voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)gy * w + gx) * 4) : 0xFFFFFFFFu;
    }
  };
  auto p_setup_tile = [&]() __attribute__((always_inline)) {
    tile_coords(p_k, p_bn, p_y0, p_x0, p_lin);
    p_setup_lanes();
  };
  int bn = 0, y0 = 0, x0 = 0, tile_lin = 0;
  tile_coords(0, bn, y0, x0, tile_lin);
  p_bn = bn; p_y0 = y0; p_x0 = x0; p_lin = tile_lin;   // the prefetch stream starts on the same tile: its coordinates once
  p_setup_lanes();

  // The source table is read from the kernel arguments ONCE, into scalar registers, and indexed by select chains (no
  // scalar loads inside the loop: they share the lgkmcnt counter with the LDS reads and return out of order).
  const float* const sp0 = a.src[0]; const float* const sp1 = a.src[1]; const float* const sp2 = a.src[2];
  const float* const sp3 = a.src[3]; const float* const sp4 = a.src[4];
  const int sc0 = a.src_c[0], sc1 = a.src_c[1], sc2 = a.src_c[2], sc3 = a.src_c[3], sc4 = a.src_c[4];
  auto src_of = [&](int i) __attribute__((always_inline)) { return i == 0 ? sp0 : i == 1 ? sp1 : i == 2 ? sp2 : i == 3 ? sp3 : sp4; };
  auto src_c_of = [&](int i) __attribute__((always_inline)) { return i == 0 ? sc0 : i == 1 ? sc1 : i == 2 ? sc2 : i == 3 ? sc3 : sc4; };
  const char* zero_src = reinterpret_cast<const char*>(g_wino4_zero);   // its address comes from a scalar load too:
  asm volatile("" : "+s"(zero_src));                                     // taken once, not rematerialised in the loop
  const float* const wu_base = a.wu + (size_t)cot * (a.cin / CK) * U_ELEMS;
  const int n_src = a.n_src;
  float sc_use = 0.f, sc_next = 0.f;   // FUSE: channel scale for the transform of this / the next iteration (prefetched
                                       // with the patch, two chunks ahead)
  // the patch of the NEXT chunk of the prefetch stream -> patch stage; every piece is always issued (zeros outside)
  auto issue_patch = [&](int stage) {
    float* s_in = smem + stage * IN_PAD;
    const int sc = src_c_of(p_cs);
    const char* sp = reinterpret_cast<const char*>(src_of(p_cs) + ((size_t)p_bn * sc + p_cc0) * plane);
    const char* zp = zero_src;
    const char* xp = FUSE ? reinterpret_cast<const char*>(a.ca_x + ((size_t)p_bn * sc + p_cc0) * plane) : nullptr;
    if (FUSE) sc_next = a.ca_scale[(size_t)p_bn * sc + p_cc0 + kq];   // this lane's channel of that chunk (single source)
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      if (seg < IN_SEGS) {   // wave-uniform
        const bool ok = voff[i] != 0xFFFFFFFFu;
        __builtin_amdgcn_global_load_lds((gptr_t)(ok ? sp + voff[i] : zp), (lptr_t)(s_in + seg * 256), 16, 0, 0);
        if (FUSE)
          __builtin_amdgcn_global_load_lds((gptr_t)(ok ? xp + voff[i] : zp), (lptr_t)(s_in + X_OFF + seg * 256), 16, 0, 0);
      }
    }
    p_cc0 += CK;
    if (p_cc0 >= sc) {
      ++p_cs;
      p_cc0 = 0;
      if (p_cs >= n_src) {
        p_cs = 0;
        ++p_k;
        if (p_k < my_tiles) p_setup_tile();
      }
    }
  };
  auto issue_u = [&](int g, int stage) {
    float* s_u = smem + OFF_U + stage * U_ELEMS;
    // wave-uniform piece base + one per-lane offset (a per-piece vector offset costs two registers per piece)
    const char* usrc = reinterpret_cast<const char*>(wu_base + (size_t)g * U_ELEMS) + wave * 1024;
#pragma unroll
    for (int i = 0; i < U_IT; ++i) {
      const int seg = i * NW + wave;
#ifdef EAVSR_WINO_EXP_QUARTER_U      // timing only: a quarter of the weight pieces (what an in-kernel G g G^T expansion would request)
      if (seg < U_SEGS && (seg & 3) == 0)
#else
      if (seg < U_SEGS)
#endif
        __builtin_amdgcn_global_load_lds((gptr_t)(usrc + i * (NW * 1024) + lane16), (lptr_t)(s_u + seg * 256), 16, 0, 0);
    }
  };

  // the same two transfers one 1-KiB piece at a time (EAVSR_W4_SPREAD: issued between the GEMM steps instead of as a burst)
  auto issue_u_piece = [&](int i, int g, int stage) __attribute__((always_inline)) {
    float* s_u = smem + OFF_U + stage * U_ELEMS;
    const char* usrc = reinterpret_cast<const char*>(wu_base + (size_t)g * U_ELEMS) + wave * 1024;
    const int seg = i * NW + wave;
#ifdef EAVSR_WINO_EXP_QUARTER_U
    if (seg < U_SEGS && (seg & 3) == 0)
#else
    if (seg < U_SEGS)
#endif
      __builtin_amdgcn_global_load_lds((gptr_t)(usrc + i * (NW * 1024) + lane16), (lptr_t)(s_u + seg * 256), 16, 0, 0);
  };
  auto issue_patch_piece = [&](int i, int stage) __attribute__((always_inline)) {
    float* s_in = smem + stage * IN_PAD;
    const int sc = src_c_of(p_cs);
    const char* sp = reinterpret_cast<const char*>(src_of(p_cs) + ((size_t)p_bn * sc + p_cc0) * plane);
    const int seg = i * NW + wave;
    if (seg < IN_SEGS) {
      const bool ok = voff[i] != 0xFFFFFFFFu;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? sp + voff[i] : zero_src), (lptr_t)(s_in + seg * 256), 16, 0, 0);
    }
  };
  auto patch_advance = [&]() __attribute__((always_inline)) {
    const int sc = src_c_of(p_cs);
    p_cc0 += CK;
    if (p_cc0 >= sc) {
      ++p_cs;
      p_cc0 = 0;
      if (p_cs >= n_src) {
        p_cs = 0;
        ++p_k;
        if (p_k < my_tiles) p_setup_tile();
      }
    }
  };

  // Input transform of one chunk: 4 channels x 32 tiles = 128 jobs = the lanes of TWO waves (lane (kq, l15) of wave
  // 2d + u: channel kq, tile (row u, column l15)); the wave pair on duty rotates with the chunk.  Both passes stay in
  // registers - one LDS round trip - and the partner wave of each duty wave's SIMD has the matrix pipe to itself
  // meanwhile.  (A version that split every job over three lanes of all eight waves, second pass in place through
  // LDS, was 8 % slower: twice the LDS traffic, two dependent round trips in every wave.)
  auto transform = [&](int ps, int vs, int t_bn, int t_y0, int t_x0, int t_chunk) __attribute__((always_inline)) {
#ifndef EAVSR_WINO_EXP_NOTRANSFORM   // timing ablations only: results are wrong
    float* vd = s_v + vs * V_ELEMS + kq * 64 + 2 * ((tg * 16 + l15) ^ ((kq & 1) << 4));   // + (xi / 2) * 256 + (xi & 1)
    f32x2 t12[6], t34[6], t05[6];   // column pass B^T d, two columns per packed operation
    if (R == 3) {
      // the 6 x 6 patch of tile column l15 starts at column MARG - 1 + 4 l15 = 3 + 4 l15: one b32, one aligned b128, one b32
      const float* pp = smem + ps * IN_PAD + kq * (IH * IW) + (M * tg) * IW + 4 * l15;
#if !defined(EAVSR_W4_COMPILER_READS)
      // Grouped schedule: ONE wave per SIMD runs this transform while the others wait at a barrier, so every exposed LDS round
      // trip is the phase's time.  Left to the compiler the twelve single-dword reads of columns 0 / 5 were sunk behind the first
      // column passes and each batch waited for with `lgkmcnt(0)` right behind its issue (two exposed round trips).  All eighteen
      // reads up front as inline assembly, the six quads first: the column passes on them start behind `lgkmcnt(12)` (the counter
      // is in order) while the twelve dwords land.
      constexpr bool HAND_READS = GRP;
#else
      constexpr bool HAND_READS = false;
#endif
      f32x2 e05[6];
      {
        f32x4 q[6];
        if constexpr (HAND_READS) {
          const unsigned pp_l = (unsigned)(unsigned long long)(lptr_t)pp;
#define W4_RDQ(R) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[R]) : "v"(pp_l), "n"(((R) * IW + 4) * 4))
#define W4_RDE(R)                                                                                              \
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(e05[R].x) : "v"(pp_l), "n"(((R) * IW + 3) * 4));        \
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(e05[R].y) : "v"(pp_l), "n"(((R) * IW + 8) * 4))
          W4_RDQ(0); W4_RDQ(1); W4_RDQ(2); W4_RDQ(3); W4_RDQ(4); W4_RDQ(5);
          W4_RDE(0); W4_RDE(1); W4_RDE(2); W4_RDE(3); W4_RDE(4); W4_RDE(5);
#undef W4_RDQ
#undef W4_RDE
          asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]));
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r) q[r] = *reinterpret_cast<const f32x4*>(pp + r * IW + 4);
        }
        if (FUSE) {
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const f32x4 qx = *reinterpret_cast<const f32x4*>(pp + X_OFF + r * IW + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) q[r][j] = fmaf(q[r][j], sc_use, qx[j]);
          }
          if (a.ca_out != nullptr && cot == 0) {   // rows 1..4, columns 1..4 of the 6 x 6 patch = this tile's 4 x 4 pixels
            const int gx = t_x0 + 4 * l15;
#pragma unroll
            for (int r = 1; r < 5; ++r) {
              const int gy = t_y0 + 4 * tg + r - 1;
              if (gy < h && gx < w)
                *reinterpret_cast<f32x4*>(a.ca_out + ((size_t)t_bn * a.cin + t_chunk * CK + kq) * plane + (size_t)gy * w + gx) = q[r];
            }
          }
        }
        f32x2 d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) d[r] = f32x2{q[r][0], q[r][1]};
        in1d(d, t12);
#pragma unroll
        for (int r = 0; r < 6; ++r) d[r] = f32x2{q[r][2], q[r][3]};
        in1d(d, t34);
      }
      {
        f32x2 d[6];
        if constexpr (HAND_READS) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e05[0]), "+v"(e05[1]), "+v"(e05[2]), "+v"(e05[3]), "+v"(e05[4]), "+v"(e05[5]));
#pragma unroll
          for (int r = 0; r < 6; ++r) d[r] = e05[r];
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r) d[r] = f32x2{pp[r * IW + 3], pp[r * IW + 8]};
        }
        if (FUSE) {
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            d[r].x = fmaf(d[r].x, sc_use, pp[X_OFF + r * IW + 3]);
            d[r].y = fmaf(d[r].y, sc_use, pp[X_OFF + r * IW + 8]);
          }
        }
        in1d(d, t05);
      }
    } else {
      // 5x5: the patch of tile column l15 starts at column MARG - 2 + 2 l15 (even): three aligned b64 per row.  The
      // pairs are (c0, c1), (c2, c3), (c4, c5); they are renamed below so that both filter sizes share the row pass.
      const float* pp = smem + ps * IN_PAD + kq * (IH * IW) + (M * tg) * IW + (MARG - PADR) + 2 * l15;
      f32x2 d[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = *reinterpret_cast<const f32x2*>(pp + r * IW);
      in1d(d, t05);   // .x = column 0, .y = column 1
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = *reinterpret_cast<const f32x2*>(pp + r * IW + 2);
      in1d(d, t12);   // columns 2, 3
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = *reinterpret_cast<const f32x2*>(pp + r * IW + 4);
      in1d(d, t34);   // columns 4, 5
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {   // row pass (.) B; a row's positions are stored as the pairs (1, 2), (3, 4), (0, 5): w6_slot
      f32x2 o12, o34, o05;
      if (R == 3) {
        // the column pass left exactly these pairs in packed registers: x1,x2 = t12[i], x3,x4 = t34[i], x0,x5 = t05[i]
        const f32x2 ba = t34[i] - 4.f * t12[i];          // (b, a) = (x3 - 4 x1, x4 - 4 x2)
        const f32x2 ec = t34[i] - t12[i];                // (e, c) = (x3 - x1, x4 - x2)
        // (a + b, a - b) and (c + 2 e, c - 2 e): one packed instruction each with half selects / a negated half (the compiler
        // builds (b, -b) with a move and an xor instead)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o12) : "v"(ba));
        asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(o34) : "v"(ec));
        o05 = f32x2{4.f * t05[i].x + (t34[i].y - 5.f * t12[i].y), 4.f * t12[i].x + (t05[i].y - 5.f * t34[i].x)};
      } else {
        const float d5[6] = {t05[i].x, t05[i].y, t12[i].x, t12[i].y, t34[i].x, t34[i].y};
        float o[6];
        in1d(d5, o);
        o12 = f32x2{o[1], o[2]};
        o34 = f32x2{o[3], o[4]};
        o05 = f32x2{o[0], o[5]};
      }
      *reinterpret_cast<f32x2*>(vd + (i * 3 + 0) * (CK * 64)) = o12;
      *reinterpret_cast<f32x2*>(vd + (i * 3 + 1) * (CK * 64)) = o34;
      *reinterpret_cast<f32x2*>(vd + (i * 3 + 2) * (CK * 64)) = o05;
    }
#endif
  };

  // Pipeline (one barrier per chunk): iteration j multiplies chunk j (V[j&1], U[j&1]) right after transforming chunk
  // j+1 (patch[(j+1)&1] -> V[(j+1)&1]); the weight slab runs one chunk ahead of its GEMM, the input patch two.
#ifdef EAVSR_W4_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  issue_patch(0);
  const float sc_first = sc_next;
#ifdef EAVSR_W4_UREGS
  // The weight operands never touch LDS: lane (kq, l15) of the waves of channel block cb reads its float2 of position pair i
  // straight from the packed slab (the same bytes the LDS-DMA path moves: 36 of its 48 one-KiB pieces per iteration), keeps the
  // chunk's 18 pairs in 36 registers and re-requests pair i for the NEXT chunk right after the two MFMAs that consumed it, so
  // every request has a whole iteration to land.
  // Buffer loads: resource = this launch's packed weights, voffset = the lane's byte offset inside a slab (+ the pair's
  // immediate), soffset = the slab's byte offset in a scalar register -- one instruction per request, no address arithmetic.
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wu_base), 0, (int)((size_t)(a.cin / CK) * U_ELEMS * sizeof(float)), 0x00020000);
  const int lane_uoff = (kq * 128 + 2 * ((cb * 16 + l15) ^ ((kq & 1) << 4))) * 4;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  auto load_u = [&](int soff, int i) __attribute__((always_inline)) {
    // the pair's offset goes into the SCALAR offset: a per-pair vector offset costs a register (or an add) per request
    const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, lane_uoff, soff + i * (CK * 128 * 4), 0);
    return __builtin_bit_cast(f32x2, t);      // (bit-casting t.x / t.y one by one is narrowed to a single dword load by this compiler)
  };
  f32x2 ur[NPOS / 2];
#pragma unroll
  for (int i = 0; i < NPOS / 2; ++i) ur[i] = load_u(0, i);
#endif
  if (total_iters > 1) issue_patch(1);
  W4_TL(1);   // set-up done, first DMA issued
  if constexpr (!GRP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    W4_TL(2);   // first DMA landed for every wave
    if ((wave >> 1) == 3) { sc_use = sc_first; transform(0, 0, bn, y0, x0, 0); }   // the pair on duty "before iteration 0"
  }
  sc_use = sc_next;   // chunk 1's scale (loaded with patch 1), for the transform of iteration 0
  int chunk = 0;   // chunk of iteration `it` within its tile
  W4_STAMP(0);      // prologue: first DMA round trip, first transform
  for (int it = 0; it < total_iters; ++it) {
    // U(it) and the patch the next transform needs have landed; every wave is done with the GEMM of iteration it-1
    W4_IT(0);
#ifdef EAVSR_W4_UREGS
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) as an instruction the wait-count pass sees: the operand registers requested
                                          // during the last GEMM are known-complete, no conservative waits inside this iteration
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    W4_STAMP(1);    // waiting for this wave's DMA
    W4_IT(1);
    __syncthreads();
    W4_IT(2);
    W4_STAMP(2);    // waiting at the barrier
    const int chunk_n = chunk + 1 == total_chunks ? 0 : chunk + 1;
    const bool on_duty = !GRP && (wave >> 1) == (it & 3);
    if constexpr (GRP) {
      if (it == 0) W4_TL(2);
      if ((it & 1) == 0) {
        // ---- transform phase of chunks it, it + 1 (patch 0 / 1 -> V 0 / 1): waves 0-3, one per SIMD; waves 4-7 request the weight
        // slab of chunk it + 1 (stage 1: the GEMM of chunk it - 1 is behind the barrier above) and wait
        if (wave < 4) {
          transform(wave >> 1, wave >> 1, bn, y0, x0, chunk + (wave >> 1));
        } else {
#if !defined(EAVSR_WINO_EXP_NODMA) && !defined(EAVSR_WINO_EXP_NOUDMA)
          const int rank = wave - 4;
          float* s_u = smem + OFF_U + U_ELEMS;
          const char* usrc = reinterpret_cast<const char*>(wu_base + (size_t)chunk_n * U_ELEMS) + rank * 1024;
#pragma unroll
          for (int i = 0; i < U_SEGS / 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(usrc + i * (4 * 1024) + lane16), (lptr_t)(s_u + (i * 4 + rank) * 256), 16, 0, 0);
#endif
        }
        // V 0 / 1 published, the patches consumed.  A raw barrier behind the LDS stores only: __syncthreads() would also wait
        // for the weight requests just made
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    auto issue_dma = [&]() __attribute__((always_inline)) {
      if constexpr (GRP) {
#ifndef EAVSR_WINO_EXP_NODMA
        // first GEMM of the pair: the patches of the next two chunks (both patch buffers are free behind the transform phase);
        // second GEMM: the weight slab of the next pair's first chunk (stage 0: its last reader was the first GEMM)
        if ((it & 1) == 0) {
          if (it + 2 < total_iters) { issue_patch(0); issue_patch(1); }
        } else {
#ifndef EAVSR_WINO_EXP_NOUDMA
          if (it + 1 < total_iters) issue_u(chunk_n, 0);
#endif
        }
#endif
        return;
      }
#ifndef EAVSR_WINO_EXP_NODMA
#if !defined(EAVSR_WINO_EXP_NOUDMA) && !defined(EAVSR_W4_UREGS)
#ifdef EAVSR_W4_DUTY_NO_U
      // the pair on duty has the longest path of the iteration (its DMA pieces, the transform, then its MFMAs: per-wave stamps,
      // tools/gpu_wino4_itstamp.py): it requests no weight pieces, the other six waves take six each
      if (it + 1 < total_iters && !on_duty) {
        const int rank = wave < 2 * (it & 3) ? wave : wave - 2;
        float* s_u = smem + OFF_U + ((it + 1) & 1) * U_ELEMS;
        const char* usrc = reinterpret_cast<const char*>(wu_base + (size_t)chunk_n * U_ELEMS) + rank * 1024;
#pragma unroll
        for (int i = 0; i < U_SEGS / 6; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(usrc + i * (6 * 1024) + lane16), (lptr_t)(s_u + (i * 6 + rank) * 256), 16, 0, 0);
      }
#else
      if (it + 1 < total_iters) issue_u(chunk_n, (it + 1) & 1);
#endif
#endif
      if (it + 2 < total_iters) issue_patch(it & 1);
#endif
    };
    // the two waves of a SIMD (w and w + 4) issue their DMA pieces at different points of the iteration
    const bool dma_late = wave >= 4;
#ifndef EAVSR_W4_SPREAD
#ifdef EAVSR_W4_DUTY_NO_U
    if (!dma_late && !on_duty) issue_dma();    // the pair on duty requests its (patch) pieces behind its transform
#else
    if (!dma_late) issue_dma();
#endif
#endif
    // the pair on duty transforms the next chunk before its GEMM steps
    W4_IT(3);
    W4_STAMP(3);    // DMA issue (waves 0-3)
    auto duty_transform = [&]() __attribute__((always_inline)) {
      if (on_duty && it + 1 < total_iters) {
        int t_bn = bn, t_y0 = y0, t_x0 = x0, t_lin = 0;
        if (FUSE && chunk_n == 0) tile_coords((it + 1) / total_chunks, t_bn, t_y0, t_x0, t_lin);   // first chunk of the next tile
#ifdef EAVSR_W4_DUTYPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        transform((it + 1) & 1, (it + 1) & 1, t_bn, t_y0, t_x0, chunk_n);
#ifdef EAVSR_W4_DUTYPRIO
        __builtin_amdgcn_sched_barrier(0);
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
      }
    };
#ifndef EAVSR_W4_UREGS
    duty_transform();
#if defined(EAVSR_W4_DUTY_NO_U) && !defined(EAVSR_W4_SPREAD)
    if (!dma_late && on_duty) issue_dma();
#endif
#endif
    // ---- the 36 GEMM steps of this wave: M_xi[co, t] += sum over the chunk's 4 channels U_xi[co, c] V_xi[c, t]
    // U and V hold the positions in PAIRS ([xi / 2][c][column][xi & 1], column ^ 16 (c & 1)): one ds_read_b64 per operand
    // and two positions - ds_read_b64 moves 256 B/clk against 128 for ds_read_b32 (whose 32 banks would also put the two
    // k-rows of a half-wave on the same banks), and the swizzle keeps those two rows on disjoint banks
    W4_IT(4);
    W4_STAMP(4);    // transform (the pair on duty)
    const float* ua = smem + OFF_U + (it & 1) * U_ELEMS + kq * 128 + 2 * ((cb * 16 + l15) ^ ((kq & 1) << 4));
    const float* vb = s_v + (it & 1) * V_ELEMS + kq * 64 + 2 * ((tg * 16 + l15) ^ ((kq & 1) << 4));
#ifndef EAVSR_W4_AHEAD
#define EAVSR_W4_AHEAD 3
#endif
    constexpr int AHEAD = EAVSR_W4_AHEAD, NSTEP = NPOS / 2;
    f32x2 av[AHEAD + 1], bv[AHEAD + 1];
#if !defined(EAVSR_W4_COMPILER_READS) && !defined(EAVSR_W4_UREGS) && !defined(EAVSR_WINO_EXP_UREGS) && !defined(EAVSR_WINO_EXP_NOMFMA) && \
    !defined(EAVSR_W4_SPREAD)
    // The operand reads as inline assembly with hand-counted waits.  Left to the compiler every wait in this block is
    // `lgkmcnt(0)` (with LDS-DMA in the kernel its wait-count pass never counts LDS reads): five of them per 18 steps, each
    // right behind freshly issued reads of a LATER step, i.e. an LDS round trip exposed in front of the MFMAs it guards and the
    // AHEAD-deep prefetch defeated.  The counter is in order: before the MFMAs of step i at most the 2 * min(AHEAD, NSTEP-1-i)
    // reads of the later steps may be outstanding.
    {
      const unsigned ua_l = (unsigned)(unsigned long long)(lptr_t)ua, vb_l = (unsigned)(unsigned long long)(lptr_t)vb;
      // (macros, not generic lambdas: clang rejects captured variables as asm operands inside one)
#define W4_RD(I)                                                                                                     \
  do {                                                                                                               \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(av[(I) % (AHEAD + 1)]) : "v"(ua_l), "n"((I) * (CK * 128) * 4)); \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(bv[(I) % (AHEAD + 1)]) : "v"(vb_l), "n"((I) * (CK * 64) * 4));  \
  } while (0)
#define W4_S(I)                                                                                                      \
  do {                                                                                                               \
    constexpr int cur_ = (I) % (AHEAD + 1);                                                                          \
    if constexpr ((I) == NSTEP / 2) {                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      W4_IT(5);                                                                                                      \
      if (dma_late) issue_dma();                                                                                     \
      W4_IT(6);                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    if constexpr ((I) + AHEAD < NSTEP) W4_RD(((I) + AHEAD < NSTEP ? (I) + AHEAD : 0));                               \
    constexpr int later_ = NSTEP - 1 - (I) < AHEAD ? NSTEP - 1 - (I) : AHEAD;                                        \
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(av[cur_]), "+v"(bv[cur_]) : "n"(2 * later_));                        \
    acc[2 * (I)] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur_].x, bv[cur_].x, acc[2 * (I)], 0, 0, 0);              \
    acc[2 * (I) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur_].y, bv[cur_].y, acc[2 * (I) + 1], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);      /* the MFMAs stay between their wait and the next step's reads */        \
  } while (0)
      static_assert(AHEAD >= 1 && AHEAD <= 3 && NSTEP == 18, "the steps below are written out for 18 steps, up to 3 ahead");
      W4_RD(0);
      if constexpr (AHEAD > 1) W4_RD(1);
      if constexpr (AHEAD > 2) W4_RD(2);
      W4_S(0); W4_S(1); W4_S(2); W4_S(3); W4_S(4); W4_S(5); W4_S(6); W4_S(7); W4_S(8);
      W4_S(9); W4_S(10); W4_S(11); W4_S(12); W4_S(13); W4_S(14); W4_S(15); W4_S(16); W4_S(17);
#undef W4_S
#undef W4_RD
    }
#else
#ifdef EAVSR_W4_UREGS
    // the last iteration re-requests its own chunk (unconditional requests: a branch per step would cut the GEMM into 18 blocks)
    const int u_next = (it + 1 < total_iters ? chunk_n : chunk) * (U_ELEMS * 4);
    (void)ua; (void)av;
#endif
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
#if defined(EAVSR_W4_UREGS)
      // operands in registers
#elif defined(EAVSR_WINO_EXP_UREGS)
      av[i] = f32x2{bias_r[0] + (float)i, bias_r[1]};
#else
      av[i] = *reinterpret_cast<const f32x2*>(ua + i * (CK * 128));
#endif
      bv[i] = *reinterpret_cast<const f32x2*>(vb + i * (CK * 64));
    }
#ifndef EAVSR_WINO_EXP_NOMFMA
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
#ifdef EAVSR_W4_SPREAD
      // one DMA piece per GEMM step, in the steps' matrix-pipe shadow (FUSE keeps the burst: its patch has two parts)
#ifndef EAVSR_W4_SPREAD_STRIDE
#define EAVSR_W4_SPREAD_STRIDE 1
#endif
      constexpr int SS = EAVSR_W4_SPREAD_STRIDE;
      if (!FUSE && i % SS == SS - 1 && i / SS < U_IT + IN_IT) {
        constexpr int dummy_ = 0; (void)dummy_;
        const int k = i / SS;
        __builtin_amdgcn_sched_barrier(0);
        if (k < U_IT) { if (it + 1 < total_iters) issue_u_piece(k, chunk_n, (it + 1) & 1); }
        else if (it + 2 < total_iters) {
          issue_patch_piece(k - U_IT, it & 1);
          if (k == U_IT + IN_IT - 1) patch_advance();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FUSE && i == NSTEP / 2) {
        __builtin_amdgcn_sched_barrier(0);
        issue_dma();
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      if (i == NSTEP / 2) {
        __builtin_amdgcn_sched_barrier(0);
        W4_IT(5);
        if (dma_late) issue_dma();
        W4_IT(6);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      if (i + AHEAD < NSTEP) {
#if defined(EAVSR_W4_UREGS)
        // operands in registers
#elif defined(EAVSR_WINO_EXP_UREGS)
        av[(i + AHEAD) % (AHEAD + 1)] = f32x2{bias_r[(i + AHEAD) & 3], bias_r[i & 3]};
#else
        av[(i + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f32x2*>(ua + (i + AHEAD) * (CK * 128));
#endif
        bv[(i + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f32x2*>(vb + (i + AHEAD) * (CK * 64));
      }
      const int cur = i % (AHEAD + 1);
#ifdef EAVSR_W4_UREGS
      acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[i].x, bv[cur].x, acc[2 * i], 0, 0, 0);
      acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[i].y, bv[cur].y, acc[2 * i + 1], 0, 0, 0);
      ur[i] = load_u(u_next, i);   // for the next chunk
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the LDS read of step i + AHEAD
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // the 2 MFMAs of step i
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the operand request that reuses their registers
    }
#else
      acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur].x, bv[cur].x, acc[2 * i], 0, 0, 0);
      acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur].y, bv[cur].y, acc[2 * i + 1], 0, 0, 0);
#ifdef EAVSR_WINO_EXP_UREGS
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#else
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the 2 reads of step i + AHEAD
#endif
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // the 2 MFMAs of step i
    }
#endif   // EAVSR_W4_UREGS
#else
    if (dma_late) issue_dma();
    acc[0][0] += av[0].x + bv[0].x;
#endif
#endif   // hand-counted reads | the compiler's
#ifdef EAVSR_W4_UREGS
    // the pair on duty transforms the next chunk AFTER its GEMM steps: its last operand requests land under the transform
    __builtin_amdgcn_sched_barrier(0);
    duty_transform();
#endif
    if (FUSE) sc_use = sc_next;   // loaded by this iteration's issue_patch (chunk it + 2) for the transform of it + 1
    chunk = chunk_n;
    W4_IT(7);
    W4_STAMP(5);    // the 36 GEMM steps (+ the DMA issue of waves 4-7)
    if (chunk != 0) continue;   // the tile is not finished yet
    if (it + 1 == total_iters) W4_TL(3);   // last tile: GEMM loop done

    // ---- epilogue, all in registers: lane (kq, l15) holds M_xi[co = 16 cb + 4 kq + r][tile (tg, l15)] for every xi ----
    {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
      constexpr bool fast_done = GRP;
      if constexpr (GRP) {
       {
        // ---- grouped kernel (round 6): the same output transform, bias, activation and sums in ~half the instructions.  The
        // epilogue of a one-tile workgroup is exposed whole (230 tiles on 256 CUs) and it is bound by instruction ISSUE: ~1,250
        // vector instructions per wave and tile in the round-5 form (64-bit address arithmetic and an exec-mask region per
        // store / residual load, bias + activation + sums value by value, zero-filled residual registers) against the ~260
        // packed operations of the transform itself.  Here: buffer stores / loads (one lane offset per tile, scalar offsets,
        // range-checked edges), the bias as packed pairs, a three-instruction activation, packed channel sums, the residual's
        // rows requested one channel pair ahead.
        const int gx = x0 + 4 * l15;
        const int gy0 = y0 + 4 * tg;                                  // wave-uniform
        const int rows = h - gy0;                                     // rows of this wave's tile row inside the image (may be <= 0)
        const int cblk = a.cout - cot * 64 < 64 ? a.cout - cot * 64 : 64;
        const size_t img_off = ((size_t)bn * a.cout + (size_t)cot * 64) * plane;
        const unsigned blk_bytes = (unsigned)((size_t)cblk * plane * sizeof(float));
        const w6_rsrc r_out = w6_make_rsrc(a.out + img_off, blk_bytes);
        const bool has_res = a.residual != nullptr;
        const w6_rsrc r_res = w6_make_rsrc(has_res ? a.residual + img_off : a.out + img_off, blk_bytes);
        const unsigned lane_off = gx < w ? (unsigned)((((size_t)(cb * 16 + 4 * kq)) * plane + (size_t)gy0 * w + gx) * sizeof(float)) : W6_OOB;
        const unsigned plane_b = (unsigned)(plane * sizeof(float)), row_b = (unsigned)(w * sizeof(float));
        // F.pixel_shuffle(out, 2) written directly (eavsrp_model.py:343-347): channel co = 4 c' + 2 i + j goes to out'[c'][2 y + i][2 x + j].
        // This lane's four channels are ONE c' = co >> 2 with (i, j) = (rp, ch); its four pixels x .. x + 3 of the pair (ch = 0, 1)
        // become the eight adjacent floats 2 x .. 2 x + 7 of row 2 y + rp: two 16-byte stores.  Same resource (the image's 64-channel
        // block = 16 shuffled planes of 4 h w), another lane offset.
        const bool shuf = a.out_shuffle == 2;
        const unsigned lane_off_ps = gx < w ? (unsigned)((((size_t)(cb * 4 + kq)) * 4 * plane + (size_t)(2 * gy0) * (2 * w) + 2 * gx) * sizeof(float)) : W6_OOB;
        const bool has_act = a.act != EAVSR_ACT_NONE;
        // The plain instantiation is branch-free in what it computes: a residual that is not there is sixteen zero registers (never
        // loaded), a scale that is not there is 1, the channel sums are always accumulated (and dropped at the end), the
        // activation is the select form with s = 1 for "none" -- runtime flags here came out as per-VALUE branches and selects.
        // The scaled-residual instantiation (the RCAB's second convolution: no activation in the reference, networks.py:461-464)
        // skips the activation by ONE uniform branch per step.
        float sc4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (RSC) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = cot * 64 + cb * 16 + 4 * kq + r;
            sc4[r] = co < a.cout ? a.res_scale[(size_t)bn * a.cout + co] : 0.f;      // (L2-resident)
          }
        }
        // the residual rows are requested ONE (channel pair, row) step ahead of their use, two steps' worth of registers
        f32x4 rr[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};      // [step & 1][ch]
        auto load_rr = [&](int step) __attribute__((always_inline)) {      // step = 4 rp + dy
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
            rr[step & 1][ch] = w6_ld4(r_res, lane_off, (unsigned)(2 * (step >> 2) + ch) * plane_b + (unsigned)(step & 3) * row_b);
        };
        if (has_res) load_rr(0);
        f32x2 cs2[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
        // desc.border_pieces (round 6): the sums of the OUTPUT's border lines that fall into this tile -- what eavsr_ca_scale_pre needs
        // beside the plane sums to know the NEXT convolution's channel means before it runs -- as a by-product of the epilogue (was: a
        // launch of its own on every RCAB's dependent chain).  Row 0 / h - 1 of the image: this wave's row dy of the 4-pixel blocks,
        // summed over its 16 lanes; column 0 / w - 1: pixel 0 / 3 of one lane's block, summed over the wave's rows.
        const bool b_tile = !RSC && a.border != nullptr && (y0 == 0 || y0 + TOH >= h || x0 == 0 || x0 + TOW >= w);
        float btop[4] = {0.f, 0.f, 0.f, 0.f}, bbot[4] = {0.f, 0.f, 0.f, 0.f}, blft[4] = {0.f, 0.f, 0.f, 0.f}, brgt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const f32x2 b2 = f32x2{bias_r[2 * rp], bias_r[2 * rp + 1]};
#pragma unroll
          for (int dy = 0; dy < 4; ++dy) {
            __builtin_amdgcn_sched_barrier(0);
            const int step = 4 * rp + dy;
            if (has_res && step < 7) load_rr(step + 1);
            f32x2 sq[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {      // row dy of A^T m, column q of the 6 x 6 block (two channels per packed operation)
              f32x2 m[6];
#pragma unroll
              for (int i = 0; i < 6; ++i) m[i] = f32x2{acc[w6_slot(i, q)][2 * rp], acc[w6_slot(i, q)][2 * rp + 1]};
              if (dy == 0) sq[q] = m[0] + (m[1] + m[2]) + (m[3] + m[4]);
              else if (dy == 1) sq[q] = (m[1] - m[2]) + 2.f * (m[3] - m[4]);
              else if (dy == 2) sq[q] = (m[1] + m[2]) + 4.f * (m[3] + m[4]);
              else sq[q] = ((m[1] - m[2]) + 8.f * (m[3] - m[4])) + m[5];
            }
            const f32x2 p1 = sq[1] + sq[2], p2 = sq[1] - sq[2], p3 = sq[3] + sq[4], p4 = sq[3] - sq[4];
            f32x2 y[4];
            y[0] = (sq[0] + p1 + p3) + b2;
            y[1] = (p2 + 2.f * p4) + b2;
            y[2] = (p1 + 4.f * p3) + b2;
            y[3] = ((p2 + 8.f * p4) + sq[5]) + b2;
            f32x4 vc[2];
            if (!RSC || has_act) {
#pragma unroll
              for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int j = 0; j < 4; ++j) vc[ch][j] = eavsr_act(ch == 0 ? y[j].x : y[j].y, act_s);
            } else {
#pragma unroll
              for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int j = 0; j < 4; ++j) vc[ch][j] = ch == 0 ? y[j].x : y[j].y;
            }
            if (!RSC && shuf) {      // (no residual, no channel sums: the launcher checks)
              if (dy < rows) {
                const unsigned so = (unsigned)(2 * dy + rp) * (2 * row_b);
                w6_st4(r_out, lane_off_ps, so, f32x4{vc[0][0], vc[1][0], vc[0][1], vc[1][1]});
                w6_st4(r_out, lane_off_ps, so + 16u, f32x4{vc[0][2], vc[1][2], vc[0][3], vc[1][3]});
              }
            } else {
#pragma unroll
              for (int ch = 0; ch < 2; ++ch) {
                const int r = 2 * rp + ch;
                const f32x4 v = vc[ch];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(v[j], sc4[r], rr[step & 1][ch][j]);
                if (dy < rows) {      // wave-uniform: this row of the tile lies inside the image
                  if constexpr (!RSC) {
                    const f32x2 h2 = f32x2{v[0], v[1]} + f32x2{v[2], v[3]};
                    cs2[r] += h2;
                    if (b_tile) {      // (wave-uniform)
                      if (gy0 + dy == 0) btop[r] += h2.x + h2.y;
                      if (gy0 + dy == h - 1) bbot[r] += h2.x + h2.y;
                      blft[r] += v[0];
                      brgt[r] += v[3];
                    }
                  }
                  w6_st4(r_out, lane_off, (unsigned)r * plane_b + (unsigned)dy * row_b, o);
                }
              }
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cot * 64 + cb * 16 + 4 * kq + r;
          // (columns right of the image and channels >= cout are excluded here, rows below it above)
          csum[r] = (co < a.cout && gx < w) ? cs2[r].x + cs2[r].y : 0.f;
        }
        if constexpr (!RSC) {
          if (b_tile) {      // (cout == 64 and one output-channel tile: the launcher checks)
            const int txi = x0 / TOW, tyi = y0 / TOH;
            float* bp = a.border + (size_t)bn * 4 * a.border_stride * 64 + cb * 16 + 4 * kq;
            const bool has_top = gy0 == 0 && rows > 0, has_bot = gy0 <= h - 1 && h - 1 < gy0 + M;      // wave-uniform
            if (has_top || has_bot) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float vt = gx < w ? btop[r] : 0.f, vb = gx < w ? bbot[r] : 0.f;
                vt += __shfl_xor(vt, 8); vb += __shfl_xor(vb, 8);
                vt += __shfl_xor(vt, 4); vb += __shfl_xor(vb, 4);
                vt += __shfl_xor(vt, 2); vb += __shfl_xor(vb, 2);
                vt += __shfl_xor(vt, 1); vb += __shfl_xor(vb, 1);
                if (l15 == 0) {
                  if (has_top) bp[(size_t)(0 * a.border_stride + txi) * 64 + r] = vt;
                  if (has_bot) bp[(size_t)(1 * a.border_stride + txi) * 64 + r] = vb;
                }
              }
            }
            // column pieces: one per (tile row, wave row); a wave whose rows lie below the image contributes zeros
            if (x0 == 0 && l15 == 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r) bp[(size_t)(2 * a.border_stride + 2 * tyi + tg) * 64 + r] = blft[r];
            }
            if (x0 + TOW >= w && gx == w - 4) {
#pragma unroll
              for (int r = 0; r < 4; ++r) bp[(size_t)(3 * a.border_stride + 2 * tyi + tg) * 64 + r] = brgt[r];
            }
          }
        }
       }
      }
      if constexpr (fast_done) {
        // (nothing left: the grouped kernel's epilogue above)
      } else if constexpr (R == 3) {
        const int gx = x0 + 4 * l15;   // 16 lanes x float4 = one 256-byte row segment
        // Two output channels at a time as packed fp32 pairs (r = 2 rp, 2 rp + 1 are adjacent accumulator registers), one
        // output row at a time (registers): half the vector instructions of a channel-by-channel transform - the epilogue
        // is paid once per launch when a workgroup has a single tile.
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const int co0 = cot * 64 + cb * 16 + 4 * kq + 2 * rp;
#pragma unroll
          for (int dy = 0; dy < 4; ++dy) {
            __builtin_amdgcn_sched_barrier(0);
            const int gy = y0 + 4 * tg + dy;
            const bool pok = gy < h && gx < w;     // w % 4 == 0 and gx % 4 == 0: gx + 3 < w as well
            f32x4 rr[2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
              rr[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (a.residual != nullptr && co0 + ch < a.cout && pok)
                rr[ch] = *reinterpret_cast<const f32x4*>(a.residual + ((size_t)bn * a.cout + co0 + ch) * plane + (size_t)gy * w + gx);
            }
            f32x2 s[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {      // row dy of A^T m, column q of the 6 x 6 block
              f32x2 m[6];
#pragma unroll
              for (int i = 0; i < 6; ++i) m[i] = f32x2{acc[w6_slot(i, q)][2 * rp], acc[w6_slot(i, q)][2 * rp + 1]};
              if (dy == 0) s[q] = m[0] + (m[1] + m[2]) + (m[3] + m[4]);
              else if (dy == 1) s[q] = (m[1] - m[2]) + 2.f * (m[3] - m[4]);
              else if (dy == 2) s[q] = (m[1] + m[2]) + 4.f * (m[3] + m[4]);
              else s[q] = ((m[1] - m[2]) + 8.f * (m[3] - m[4])) + m[5];
            }
            const f32x2 p1 = s[1] + s[2], p2 = s[1] - s[2], p3 = s[3] + s[4], p4 = s[3] - s[4];
            f32x2 y[4];
            y[0] = s[0] + p1 + p3;
            y[1] = p2 + 2.f * p4;
            y[2] = p1 + 4.f * p3;
            y[3] = (p2 + 8.f * p4) + s[5];
            float vv[2][4];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
              const int r = 2 * rp + ch;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float t = (ch == 0 ? y[j].x : y[j].y) + bias_r[r];
                t = eavsr_act(t, act_s);
                vv[ch][j] = t;
              }
            }
            if (a.out_shuffle == 2) {
              // F.pixel_shuffle(out, 2) written directly (eavsrp_model.py:343-347): channel co = 4 c' + 2 i + j goes to
              // out'[c'][2 y + i][2 x + j].  This lane's pair (co0, co0 + 1) is (i = rp, j = 0 / 1) of one c', its four
              // pixels x .. x + 3 become the eight adjacent floats 2 x .. 2 x + 7 of row 2 y + rp: two float4 stores.
              if (co0 + 1 < a.cout && pok) {
                float* o = a.out + ((size_t)bn * (a.cout >> 2) + (co0 >> 2)) * (4 * plane) + (size_t)(2 * gy + rp) * (2 * w) + 2 * gx;
                *reinterpret_cast<f32x4*>(o) = f32x4{vv[0][0], vv[1][0], vv[0][1], vv[1][1]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{vv[0][2], vv[1][2], vv[0][3], vv[1][3]};
              }
            } else {
#pragma unroll
              for (int ch = 0; ch < 2; ++ch) {
                const int r = 2 * rp + ch, co = co0 + ch;
                if (co < a.cout && pok) {
                  csum[r] += (vv[ch][0] + vv[ch][1]) + (vv[ch][2] + vv[ch][3]);
                  if constexpr (RSC) {
                    const float sc = a.res_scale[(size_t)bn * a.cout + co];      // (L2-resident)
                    *reinterpret_cast<f32x4*>(a.out + ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx) =
                        f32x4{vv[ch][0] * sc + rr[ch][0], vv[ch][1] * sc + rr[ch][1], vv[ch][2] * sc + rr[ch][2], vv[ch][3] * sc + rr[ch][3]};
                  } else {
                    *reinterpret_cast<f32x4*>(a.out + ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx) =
                        f32x4{vv[ch][0] + rr[ch][0], vv[ch][1] + rr[ch][1], vv[ch][2] + rr[ch][2], vv[ch][3] + rr[ch][3]};
                  }
                }
              }
            }
          }
        }
      } else {
        // F(2x2, 5x5): A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1], two output rows of two pixels per tile
        const int gx = x0 + 2 * l15;   // 16 lanes x float2 = one 128-byte row segment
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cot * 64 + cb * 16 + 4 * kq + r;
          const bool cok = co < a.cout;
          const float bb = bias_r[r];
          __builtin_amdgcn_sched_barrier(0);
          f32x2 rr[2];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const int gy = y0 + 2 * tg + dy;
            rr[dy] = f32x2{0.f, 0.f};
            if (a.residual != nullptr && cok && gy < h && gx < w)
              rr[dy] = *reinterpret_cast<const f32x2*>(a.residual + ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx);
          }
          float s[2][6];
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            const float m0 = acc[w6_slot(0, q)][r], m1 = acc[w6_slot(1, q)][r], m2 = acc[w6_slot(2, q)][r], m3 = acc[w6_slot(3, q)][r],
                        m4 = acc[w6_slot(4, q)][r], m5 = acc[w6_slot(5, q)][r];
            s[0][q] = m0 + (m1 + m2) + (m3 + m4);
            s[1][q] = ((m1 - m2) + 2.f * (m3 - m4)) + m5;
          }
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            float y[2];
            y[0] = s[dy][0] + (s[dy][1] + s[dy][2]) + (s[dy][3] + s[dy][4]);
            y[1] = ((s[dy][1] - s[dy][2]) + 2.f * (s[dy][3] - s[dy][4])) + s[dy][5];
            const int gy = y0 + 2 * tg + dy;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              y[j] += bb;
              y[j] = eavsr_act(y[j], act_s);
            }
            if (cok && gy < h && gx < w) {     // w % 4 == 0 and gx even: gx + 1 < w as well
              csum[r] += y[0] + y[1];
              *reinterpret_cast<f32x2*>(a.out + ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx) =
                  f32x2{y[0] + rr[dy].x, y[1] + rr[dy].y};
            }
          }
        }
      }
      if (a.chan_partial) {
        // per output channel: 16 lanes (tiles) here and the same again in wave ^ 1 (fixed order)
        __syncthreads();   // s_red is free (the previous tile's sums were read long ago)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = csum[r];
          v += __shfl_xor(v, 8);
          v += __shfl_xor(v, 4);
          v += __shfl_xor(v, 2);
          v += __shfl_xor(v, 1);
          if (l15 == 0) s_red[tg * 64 + cb * 16 + 4 * kq + r] = v;
        }
        __syncthreads();
        if (tid < 64) {
          const int co = cot * 64 + tid;
          if (co < a.cout)
            a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + tile_lin) * a.cout + co] = s_red[tid] + s_red[64 + tid];
        }
      }
      // next tile of this workgroup
#pragma unroll
      for (int x = 0; x < NPOS; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (it + 1 < total_iters) tile_coords((it + 1) / total_chunks, bn, y0, x0, tile_lin);
    }
    W4_STAMP(6);    // epilogue: output transform, stores
  }   // flattened (tile, chunk) loop
  W4_TL(4);   // epilogue issued
#ifdef EAVSR_W4_ITSTAMP
  __syncthreads();
  if (tid < 256) g_w4_it[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = s_its[tid];
#endif
#ifdef EAVSR_W4_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W4_TL(5);   // this wave's stores acknowledged
#endif
#ifdef EAVSR_W4_STAMPS
  if (lane == 0 && (wave & 1) == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_w4_stamps[(wave >> 1) * 8 + i], st_acc[i]);
  }
#endif
}

// weight (cout, cin, R, R) -> U = G g G^T laid out [cot][cin / 4][xi / 2][c][co ^ 16 (c & 1)][xi & 1] (zero for
// co >= cout); evaluated in double and rounded once.  G[p] = scale_p * (1, p, p^2, ..) for the points 0, 1, -1, 2, -2
// (scales 1/4, -1/6, -1/6, 1/24, 1/24) and the unit vector of the highest power for the point at infinity.
template <int R>
__global__ void pack_wino6_kernel(const float* __restrict__ wt, float* __restrict__ out, int cout, int cin, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  // destination index: [cot][chunk][xi / 2][c][column][xi & 1]
  const int lo = (int)(e & 1);
  const int colsw = (int)((e >> 1) & 63);
  long u = e >> 7;
  const int c = (int)(u % CK); u /= CK;
  const int pair = (int)(u % (NPOS / 2)); u /= NPOS / 2;
  const int xi = 6 * (pair / 3) + w6_col_of(pair % 3, lo);   // the position this slot holds (w6_slot)
  const int col = colsw ^ ((c & 1) << 4);
  const int nchunks = cin / CK;
  const int chunk = (int)(u % nchunks);
  const int cot = (int)(u / nchunks);
  const int co = cot * 64 + col, ci = chunk * CK + c;
  float v = 0.f;
  if (co < cout) {
    const float* g = wt + ((size_t)co * cin + ci) * (R * R);
    const int r = xi / 6, q = xi - 6 * r;
    const double pt[5] = {0., 1., -1., 2., -2.}, sc[5] = {0.25, -1. / 6, -1. / 6, 1. / 24, 1. / 24};
    double Gr[R], Gq[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      double pr = 1., pq = 1.;
      for (int e2 = 0; e2 < k; ++e2) { pr *= pt[r < 5 ? r : 0]; pq *= pt[q < 5 ? q : 0]; }
      Gr[k] = r < 5 ? sc[r] * pr : (k == R - 1 ? 1. : 0.);
      Gq[k] = q < 5 ? sc[q] * pq : (k == R - 1 ? 1. : 0.);
    }
    double s = 0.;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j) s += Gr[i] * (double)g[i * R + j] * Gq[j];
    v = (float)s;
  }
  out[e] = v;
}

}  // namespace

#ifdef EAVSR_W4_STAMPS
extern "C" int eavsr_debug_w4_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4_stamps), sizeof(g_w4_stamps));
  if (reset) {
    unsigned long long z[32] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_w4_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

#ifdef EAVSR_W4_ITSTAMP
extern "C" int eavsr_debug_w4_itstamps(unsigned long long* host_out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4_it), sizeof(g_w4_it));
  return 0;
}
#endif

#ifdef EAVSR_W4_TIMELINE
extern "C" int eavsr_debug_w4_timeline(unsigned long long* host_out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4_tl), sizeof(g_w4_tl));
  return 0;
}
#endif

extern "C" int64_t eavsr_wino4_weight_elems(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % CK != 0) return 0;
  return (int64_t)eavsr::cdiv(cout, 64) * (cin / CK) * U_ELEMS;
}

namespace {

template <int R>
int pack_wino6(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight_wino: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % CK == 0, -1, "pack_conv_weight_wino: cin %d must be a multiple of 4", cin);
  const long total = eavsr_wino4_weight_elems(cout, cin);
  hipLaunchKernelGGL(pack_wino6_kernel<R>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream),
                     weight, packed, cout, cin, total);
  return eavsr::launch_status("pack_conv_weight_wino");
}

template <int R>
int launch_wino6(const eavsr_conv2d_desc* d, const float* weight_wino, void* stream) {
  using C = WCfg<R>;
  EAVSR_REQUIRE(d != nullptr && weight_wino != nullptr, -1, "conv_wino6: NULL descriptor / weights");
  EAVSR_REQUIRE(d->n_src >= 1 && d->n_src <= 5, -1, "conv_wino6: n_src %d not in 1..5", d->n_src);
  EAVSR_REQUIRE(d->ksize == R, -2, "conv_wino6: kernel size %d (this entry point: %d)", d->ksize, R);
  EAVSR_REQUIRE(d->out, -1, "conv_wino6: NULL out");
  EAVSR_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, -1, "conv_wino6: bad dims");
  EAVSR_REQUIRE(d->act >= 0 && d->act <= 2, -1, "conv_wino6: act %d", d->act);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_LRELU || (d->slope >= 0.f && d->slope <= 1.f), -2,
                "conv_wino6: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(t, slope t))", (double)d->slope);
  const bool fuse = d->ca_scale != nullptr;
#if !EAVSR_LAB
  EAVSR_REQUIRE(!fuse, -2, "conv_wino6: the fused channel-attention prologue is a lab-build instantiation (python -m eavsr_amd.build --lab); "
                           "use eavsr_scale_residual_f32 + a plain convolution, or desc.res_scale on the previous convolution");
#endif
  if (fuse) {
    EAVSR_REQUIRE(R == 3, -2, "conv_wino6: the fused channel-attention prologue exists for the 3x3 kernel only");
    EAVSR_REQUIRE(d->ca_x != nullptr, -1, "conv_wino6: ca_scale without ca_x");
    EAVSR_REQUIRE(d->n_src == 1 && (((uintptr_t)d->ca_x) & 15) == 0 && (d->ca_out == nullptr || (((uintptr_t)d->ca_out) & 15) == 0),
                  -2, "conv_wino6: the fused channel-attention prologue needs a single source and 16-byte aligned ca_x / ca_out");
  } else {
    EAVSR_REQUIRE(d->ca_x == nullptr && d->ca_out == nullptr, -1, "conv_wino6: ca_x / ca_out without ca_scale");
  }
  EAVSR_REQUIRE(d->w % 4 == 0, -2, "conv_wino6: w %% 4 != 0 (use eavsr_conv2d_f32)");
  constexpr uintptr_t AL = R == 3 ? 15 : 7;   // float4 / float2 row stores
  EAVSR_REQUIRE((((uintptr_t)d->out) & AL) == 0 && (d->residual == nullptr || (((uintptr_t)d->residual) & AL) == 0), -2,
                "conv_wino6: out / residual must be %d-byte aligned", (int)AL + 1);
  W4Args a;
  int csum = 0;
  for (int s = 0; s < 5; ++s) {
    a.src[s] = s < d->n_src ? d->src[s] : nullptr;
    a.src_c[s] = s < d->n_src ? d->src_c[s] : 0;
    if (s < d->n_src) {
      EAVSR_REQUIRE(d->src[s] != nullptr && d->src_c[s] > 0 && d->src_c[s] % CK == 0 && (((uintptr_t)d->src[s]) & 15) == 0, -2,
                    "conv_wino6: source %d must be 16-byte aligned with a multiple of 4 channels", s);
      csum += d->src_c[s];
    }
  }
  EAVSR_REQUIRE(csum == d->cin, -1, "conv_wino6: sources sum to %d channels, cin = %d", csum, d->cin);
  if (d->n == 0) return 0;
  a.n_src = d->n_src;
  a.wu = weight_wino;
  a.bias = d->bias; a.residual = d->residual; a.out = d->out; a.chan_partial = d->chan_partial;
  a.ca_scale = d->ca_scale; a.ca_x = d->ca_x; a.ca_out = d->ca_out;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
  a.tiles_x = eavsr::cdiv(d->w, C::TOW);
  a.tiles_y = eavsr::cdiv(d->h, C::TOH);
  a.act = d->act; a.slope = d->slope;
  a.out_shuffle = d->out_shuffle;
  EAVSR_REQUIRE(d->sum_mul == nullptr, -2, "conv3x3_wino4: sum_mul exists in eavsr_conv3x3_f32x6s only");
  a.res_scale = d->res_scale;
  a.border = d->border_pieces;
  a.border_stride = d->border_stride;
  if (d->border_pieces != nullptr) {
    EAVSR_REQUIRE(R == 3 && d->cout == 64 && d->out_shuffle == 0 && d->res_scale == nullptr && d->ca_scale == nullptr &&
                      (d->cin / CK) % 2 == 0 && w6_grouped_schedule(), -2,
                  "conv_wino6: border_pieces needs the grouped 3x3 kernel, 64 output channels, no shuffle / scaled residual / prologue");
    EAVSR_REQUIRE(d->border_stride >= a.tiles_x && d->border_stride >= 2 * a.tiles_y, -1,
                  "conv_wino6: border_stride %d < max(tiles_x %d, 2 tiles_y %d)", d->border_stride, a.tiles_x, 2 * a.tiles_y);
  }
  EAVSR_REQUIRE(d->res_scale == nullptr || (R == 3 && d->residual != nullptr && d->out_shuffle == 0 && d->ca_scale == nullptr &&
                                            (d->cin / CK) % 2 == 0 && w6_grouped_schedule()), -2,
                "conv_wino6: res_scale needs the (grouped) 3x3 kernel, a residual, no prologue, an even number of 4-channel chunks");
  EAVSR_REQUIRE(d->out_shuffle == 0 || d->out_shuffle == 2, -1, "conv_wino6: out_shuffle %d (0 or 2)", d->out_shuffle);
  if (d->out_shuffle == 2)
    EAVSR_REQUIRE(R == 3 && d->cout % 4 == 0 && d->residual == nullptr && d->chan_partial == nullptr, -2,
                  "conv_wino6: the pixel-shuffle epilogue needs the 3x3 kernel, cout %% 4 == 0, no residual, no channel sums");
  const long blocks = (long)a.tiles_x * a.tiles_y * d->n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv_wino6: too many tiles");
  EAVSR_REQUIRE((long)d->h * d->w * 16 < (1L << 31), -1, "conv_wino6: image plane too large for 32-bit tile offsets");
  EAVSR_REQUIRE(R != 3 || (long)d->h * d->w * 64 * 4 < (1L << 31), -1,
                "conv_wino6: image plane too large for the epilogue's 32-bit offsets (64 channels x h x w x 4 bytes must be < 2 GiB)");
  [[maybe_unused]] constexpr size_t LDS_FUSE = C::LDS_BYTES + 2 * C::IN_PAD * sizeof(float);
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino6_kernel<R, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
#if EAVSR_LAB
    if (R == 3 && attr_err == hipSuccess)
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino6_kernel<3, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WCfg<3>::LDS_BYTES + 2 * WCfg<3>::IN_PAD * sizeof(float)));
#endif
    if (R == 3 && attr_err == hipSuccess)
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino6_kernel<3, false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)WCfg<3>::LDS_BYTES);
    if (R == 3 && attr_err == hipSuccess)
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino6_kernel<3, false, true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)WCfg<3>::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv_wino6: hipFuncSetAttribute(%zu B of LDS): %s", C::LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  // persistent workgroups: one per CU, each walking blocks / grid.x tiles
  const long per_cot = blocks < 256 ? blocks : 256;
  dim3 grid((unsigned)per_cot, eavsr::cdiv(d->cout, 64));
  // EAVSR_WINO6_SCHED=lockstep: round 2's schedule (every wave the same program), the A/B reference of the ping-pong kernel
  if (fuse) {
#if EAVSR_LAB
    if constexpr (R == 3)
      hipLaunchKernelGGL((conv_wino6_kernel<3, true>), grid, dim3(64 * NW), LDS_FUSE, eavsr::as_stream(stream), a);
#endif
  } else {
    // the grouped schedule (transform phases of two chunks, pure GEMM iterations): 3x3, an even number of 4-channel chunks;
    // EAVSR_W4_GRP=0 keeps the duty-pair schedule (A/B switch)
    const bool grp_on = w6_grouped_schedule();
    if constexpr (R == 3) {
      if (grp_on && (d->cin / CK) % 2 == 0) {
        if (d->res_scale != nullptr)
          hipLaunchKernelGGL((conv_wino6_kernel<3, false, true, true>), grid, dim3(64 * NW), C::LDS_BYTES, eavsr::as_stream(stream), a);
        else
          hipLaunchKernelGGL((conv_wino6_kernel<3, false, true>), grid, dim3(64 * NW), C::LDS_BYTES, eavsr::as_stream(stream), a);
        return eavsr::launch_status("conv_wino6");
      }
    }
    hipLaunchKernelGGL((conv_wino6_kernel<R, false>), grid, dim3(64 * NW), C::LDS_BYTES, eavsr::as_stream(stream), a);
  }
  return eavsr::launch_status("conv_wino6");
}

}  // namespace

extern "C" int eavsr_pack_conv_weight_wino4(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream) {
  return pack_wino6<3>(weight, packed, cout, cin, stream);
}
extern "C" int eavsr_pack_conv_weight_wino5x5(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream) {
  return pack_wino6<5>(weight, packed, cout, cin, stream);
}

extern "C" int32_t eavsr_conv3x3_wino4_tiles(int32_t h, int32_t w) {
  return eavsr::cdiv(h, WCfg<3>::TOH) * eavsr::cdiv(w, WCfg<3>::TOW);
}
extern "C" int eavsr_conv3x3_wino4_border_pieces(int32_t h, int32_t w, int32_t* p_rows, int32_t* p_cols) {
  EAVSR_REQUIRE(h > 0 && w > 0 && p_rows && p_cols, -1, "conv3x3_wino4_border_pieces: bad arguments");
  *p_rows = eavsr::cdiv(w, WCfg<3>::TOW);
  *p_cols = 2 * eavsr::cdiv(h, WCfg<3>::TOH);
  return 0;
}
extern "C" int32_t eavsr_conv5x5_wino_tiles(int32_t h, int32_t w) {
  return eavsr::cdiv(h, WCfg<5>::TOH) * eavsr::cdiv(w, WCfg<5>::TOW);
}

extern "C" int eavsr_conv3x3_wino4_f32(const eavsr_conv2d_desc* d, const float* weight_wino4, void* stream) {
  return launch_wino6<3>(d, weight_wino4, stream);
}
extern "C" int eavsr_conv5x5_wino_f32(const eavsr_conv2d_desc* d, const float* weight_wino5x5, void* stream) {
  return launch_wino6<5>(d, weight_wino5x5, stream);
}

extern "C" int eavsr_wino4_schedule(void) { return w6_grouped_schedule() ? 1 : 0; }
