// 3x3 64 -> cout stride-1 "same" convolution for SMALL launches, fp32 in / fp32 out, the contraction on the bf16 matrix pipe
// with exact operand splits ("bf16x6", as conv_x6.hip): the forward and input-gradient convolutions of RCABlock
// (models/networks.py:456-464) at a training crop -- 2 x 64 x 96 x 96 per launch, 3,515 launches per training step
// (BASELINE.json configs[3], eavsrp_model.py:109-119) -- and the pyramid levels of the flow refinement at inference.
//
// Why another kernel.  At 2 x 96 x 96 a launch is 72 tiles of 8 x 32 pixels: conv3x3_small_kernel (fp32 MFMA, 32-channel halves)
// puts 144 workgroups on 256 CUs and every wave multiplies for 288 x 64 = 18.4 K cycles, two waves per SIMD: 15 us of matrix pipe
// on the busy CUs + a four-stage chunk pipeline's fill = 29.8 us per launch, 0.29 of the fp32 MFMA peak.  Nothing about the
// decomposition fixes that (1,152 wave-tiles on 1,024 SIMDs); six bf16 partial products per 16 k do the same sum in 0.375 x the
// matrix-pipe time, and -- unlike conv_x9.hip -- every input value is split ONCE, on its way into LDS.
//
// Per workgroup (512 threads; wave = one row of 32 pixels; 32 output channels):
//   * the WHOLE input patch, (8 + 2) x (32 + 2) pixels x 64 channels, is requested in the prologue by plain coalesced loads
//     (thread = patch pixel, 64 values in flight), split exactly into three bf16 planes and kept in LDS as
//     [chunk of 8 channels][plane][row][col][8 ch] (16 bytes per pixel and plane: the MFMA B operand of lane (n = lane & 31,
//     g = lane >> 5) for k-step s -- the 8 channels of pixel n shifted by ITS tap 2 s + g -- is one ds_read_b128 per plane).
//     Chunk c + 1 is split and stored under the MFMAs of chunk c.
//   * weights pre-split and pre-arranged by eavsr_pack_conv_weight_x6(ksize = 3) in A-operand order, streamed by 16-byte LDS-DMA
//     one chunk (5 k-steps: 9 taps + one zero tap) per slab into two stages; the slab barrier sits one k-step before the slab
//     changes.
//   * 40 k-steps x 6 MFMAs (v_mfma_f32_32x32x16_bf16) per wave, operands read one k-step ahead.
//   * epilogue in registers: bias (the accumulators' initial value), activation, residual / ReLU-mask (EAVSR_ACT_RELU_MASK),
//     per-tile channel sums for the channel attention.
// LDS: 130,560 (patch) + 30,720 (two weight slabs) + 1,152 = 162,432 bytes.
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int Q_NW = 8, Q_TW = 32, Q_TH = 8;
constexpr int Q_IW = Q_TW + 2, Q_IH = Q_TH + 2, Q_NPIX = Q_IW * Q_IH;      // 34 x 10 = 340 patch pixels
constexpr int Q_PLANE_B = Q_NPIX * 16;                                     // one bf16 plane of one 8-channel chunk
constexpr int Q_CHUNK_B = 3 * Q_PLANE_B;
constexpr int Q_KSTEPS = 5;                                                // tap pairs per chunk: 9 taps + a zero tap
constexpr int Q_SLAB_U4 = Q_KSTEPS * 3 * 64;                               // 16-byte elements of one chunk's A operands (32 channels)

template <int NCH> struct QCfg {
  static constexpr int PATCH_B = NCH * Q_CHUNK_B;
  static constexpr size_t LDS_BYTES = (size_t)PATCH_B + 2 * (size_t)Q_SLAB_U4 * 16 + 32 * 4 + Q_NW * 32 * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct QArgs {
  const float* x;        // (n, 8 NCH, h, w)
  const u32x4* wsplit;   // eavsr_pack_conv_weight_x6(ksize 3): [cot][chunk][k-step][plane][mt][lane] 16-byte elements
  const float* bias;
  const float* residual; // added to the output -- or, act == EAVSR_ACT_RELU_MASK, the mask source
  float* out;            // (n, cout, h, w)
  float* chan_partial;   // (n, tiles, cout) or NULL
  const float* sum_mul;  // NULL, or (n, cout, h, w): chan_partial holds the per-tile sums of out * sum_mul (the STORED values) instead
  int n, cout, h, w, tiles_x, tiles_y;
  int wmt;               // 32-channel tiles per packed `cot` (1: cout <= 32, 2 otherwise)
  int act;
  float slope;
};

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void q_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned uma = __float_as_uint(ra), umb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(uma & 0xFFFF0000u), lb = rb - __uint_as_float(umb & 0xFFFF0000u);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  mid = __builtin_amdgcn_perm(umb, uma, 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ f32x16 q_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef EAVSR_X6S_STAMPS
// diagnostic build only (tools/build_x6s_diag.sh): clock ticks per phase, summed over wave 0 of every workgroup
__device__ unsigned long long g_q_stamps[8];
#define Q_STAMP(i)                                                    \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define Q_STAMP(i) do { } while (0)
#endif

// eight channels of four consecutive pixels (one float4 per channel) -> the four pixels' 16-byte units of the three planes
__device__ __forceinline__ void q_store_quad(const f32x4 (&q)[8], unsigned char* dst) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32x4 pl[3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned h2, m2, l2;
      q_split2(q[2 * c][i], q[2 * c + 1][i], h2, m2, l2);
      pl[0][c] = h2; pl[1][c] = m2; pl[2][c] = l2;
    }
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) *reinterpret_cast<u32x4*>(dst + i * 16 + p3 * Q_PLANE_B) = pl[p3];
  }
}

// VEC: w % 4 == 0 and 16-byte aligned x / out / residual -- the patch is requested as float4 row segments (a wave's load
// instruction costs the texture path 16 cycles whatever its width: 64 dword requests per thread were 6 K cycles of a 32 K-cycle
// workgroup) and the output tile leaves through an LDS transpose as float4 row segments (16 dword stores per lane: 5.8 K cycles).
template <int NCH, bool VEC>
__global__ __launch_bounds__(512) void conv3x3_x6s_kernel(QArgs a) {
  using K = QCfg<NCH>;
#ifdef EAVSR_X6S_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smemq[];
  unsigned char* s_patch = smemq;                                            // [NCH][3 planes][10][34][16 B]
  u32x4* s_w = reinterpret_cast<u32x4*>(smemq + K::PATCH_B);                 // [2][Q_SLAB_U4]
  float* s_bias = reinterpret_cast<float*>(smemq + K::PATCH_B + 2 * Q_SLAB_U4 * 16);
  float* s_red = s_bias + 32;                                                // [8 waves][32 channels]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;

  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;                       // 32 output channels
  const int y0 = ty * Q_TH, x0 = tx * Q_TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  // ---- weight slabs: one chunk each, LDS stage = chunk & 1 --------------------------------------------------------
  const int wcot = cot / a.wmt, wsub = cot - wcot * a.wmt;
  auto issue_slab = [&](int ch) __attribute__((always_inline)) {
    const char* wsrc = reinterpret_cast<const char*>(a.wsplit + ((size_t)wcot * NCH + ch) * (size_t)(Q_SLAB_U4 * a.wmt));
    u32x4* dst = s_w + (ch & 1) * Q_SLAB_U4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int seg = i * Q_NW + wave;              // piece (k-step, plane) of this workgroup's 32 channels; 15 of them
      if (seg < Q_KSTEPS * 3)                       // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (unsigned)((seg * a.wmt + wsub) * 64 + lane) * 16u), (lptr_t)(dst + seg * 64), 16, 0, 0);
    }
  };

  // ---- patch producer, !VEC: thread t < 340 owns patch pixel t, all 64 channels; chunk c + 1 is split under chunk c's MFMAs -----
  const int pr = tid / Q_IW, pc = tid - pr * Q_IW;
  const int pgy = y0 - 1 + pr, pgx = x0 - 1 + pc;
  const bool pok = tid < Q_NPIX && pgy >= 0 && pgy < h && pgx >= 0 && pgx < w;
  float pv[NCH][8];
  auto store_patch = [&](int ch) __attribute__((always_inline)) {
    if (tid < Q_NPIX) {
      u32x4 pl[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h2, m2, l2;
        q_split2(pv[ch][2 * c], pv[ch][2 * c + 1], h2, m2, l2);
        pl[0][c] = h2; pl[1][c] = m2; pl[2][c] = l2;
      }
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) *reinterpret_cast<u32x4*>(s_patch + ch * Q_CHUNK_B + p3 * Q_PLANE_B + tid * 16) = pl[p3];
    }
  };

  // B operand of this lane: pixel (wave + ky, l31 + kx) of the patch, tap (ky, kx) = 2 s + kg.  The tap pair of a k-step is
  // (kx, kx + 1) of one kernel row, or (2 of row 0, 0 of row 1); the last k-step pairs tap 8 with a tap that does not exist:
  // its weights are zero and its lanes' operand is set to zero (0 x 0 whatever the patch holds).
  const int bbase = (wave * Q_IW + l31) * 16;
  const int b_same = bbase + (kg ? 16 : 0);
  const int b_wrap = bbase + (kg ? (Q_IW - 2) * 16 : 0);
  auto read_b = [&](int ch, int s, u32x4 (&b)[3]) __attribute__((always_inline)) {
    const int tap0 = 2 * s, ky = tap0 / 3, kx = tap0 - 3 * ky;
    const unsigned char* base = s_patch + ch * Q_CHUNK_B + ((s == Q_KSTEPS - 1 ? bbase : kx == 2 ? b_wrap : b_same) + (ky * Q_IW + kx) * 16);
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) {
      b[p3] = *reinterpret_cast<const u32x4*>(base + p3 * Q_PLANE_B);
      if (s == Q_KSTEPS - 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) b[p3][c] = kg ? 0u : b[p3][c];
      }
    }
  };
  auto read_a = [&](int ch, int s, u32x4 (&av)[3]) __attribute__((always_inline)) {
    const u32x4* ws = s_w + (ch & 1) * Q_SLAB_U4 + s * (3 * 64) + lane;
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) av[p3] = ws[p3 * 64];
  };

  // ---- prologue --------------------------------------------------------------------------------------------------
  issue_slab(0);
  if (NCH > 1) issue_slab(1);
  if constexpr (VEC) {
    // work items: (chunk, patch row, quad of interior columns 1 + 4 q .. 4 + 4 q) = 80 NCH quad items of 8 channels x float4, and
    // (chunk, patch row, left / right halo column) = 20 NCH halo items of 8 dwords.  Thread t takes quad item t; threads 0..127
    // quad item 512 + t (NCH = 8: 640 items); threads 128..128 + 20 NCH - 1 a halo item.  Everything is split and stored here.
    static_assert(80 * NCH <= 512 + 128 && 128 + 20 * NCH <= 512, "item assignment");
    auto quad_item = [&](int q, f32x4 (&v)[8], int& lds) __attribute__((always_inline)) {
      const int ch = q / 80, rem = q - ch * 80, row = rem >> 3, quad = rem & 7;
      const int gy = y0 - 1 + row, gx = x0 + 4 * quad;
      const bool ok = q < 80 * NCH && gy >= 0 && gy < h && gx < w;      // (w % 4 == 0: gx + 3 < w as well)
      const float* sp = a.x + ((size_t)bn * (NCH * 8) + ch * 8) * plane + (ok ? (size_t)gy * w + gx : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ok ? *reinterpret_cast<const f32x4*>(sp + (size_t)j * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
      lds = ch * Q_CHUNK_B + (row * Q_IW + 1 + 4 * quad) * 16;
    };
    f32x4 qa[8], qb[8];
    float hv[8];
    int lds_a, lds_b = 0, lds_h = 0;
    quad_item(tid, qa, lds_a);
    const bool has_b = 512 + tid < 80 * NCH;                              // (wave-uniform: waves 0, 1)
    const bool has_h = tid >= 128 && tid < 128 + 20 * NCH;
    if (has_b) quad_item(512 + tid, qb, lds_b);
    if (has_h) {
      const int e = tid - 128, ch = e / 20, rem = e - ch * 20, row = rem >> 1, col = (rem & 1) * (Q_IW - 1);
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const float* sp = a.x + ((size_t)bn * (NCH * 8) + ch * 8) * plane + (ok ? (size_t)gy * w + gx : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = ok ? sp[(size_t)j * plane] : 0.f;
      lds_h = ch * Q_CHUNK_B + (row * Q_IW + col) * 16;
    }
    if (tid < 32) {
      const int co = cot * 32 + tid;
      s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    Q_STAMP(0);      // requests issued
    if (tid < 80 * NCH) q_store_quad(qa, s_patch + lds_a);
    if (has_b) q_store_quad(qb, s_patch + lds_b);
    if (has_h) {
      u32x4 pl[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h2, m2, l2;
        q_split2(hv[2 * c], hv[2 * c + 1], h2, m2, l2);
        pl[0][c] = h2; pl[1][c] = m2; pl[2][c] = l2;
      }
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) *reinterpret_cast<u32x4*>(s_patch + lds_h + p3 * Q_PLANE_B) = pl[p3];
    }
  } else {
    const float* sp = a.x + (size_t)bn * (NCH * 8) * plane + (pok ? (size_t)pgy * w + pgx : 0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[ch][j] = pok ? sp[(size_t)(ch * 8 + j) * plane] : 0.f;
    if (tid < 32) {
      const int co = cot * 32 + tid;
      s_bias[tid] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    Q_STAMP(0);      // requests issued
    store_patch(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Q_STAMP(1);      // loads landed, chunk 0 split
  __syncthreads();
  Q_STAMP(2);      // first barrier
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = s_bias[(r & 3) + 8 * (r >> 2) + 4 * kg];
  u32x4 acur[3], bcur[3];
  read_a(0, 0, acur);
  read_b(0, 0, bcur);

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const bool more = ch + 1 < NCH;
#pragma unroll
    for (int ks = 0; ks < Q_KSTEPS; ++ks) {
      const bool last = ks == Q_KSTEPS - 1;
      if (last && more) {
        // the slab barrier, one k-step early: slab ch + 1 has landed (requested a chunk ago), every wave has read the last A
        // operands of slab ch (prefetched in the previous k-step), so its stage takes slab ch + 2; chunk ch + 1 of the patch
        // (stored at ks == 1) is published by the same barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + 2 < NCH) issue_slab(ch + 2);
      }
      if constexpr (!VEC)
        if (ks == 1 && more) store_patch(ch + 1);
      u32x4 anext[3], bnext[3];
      if (!last) {
        read_a(ch, ks + 1, anext);
        read_b(ch, ks + 1, bnext);
      } else if (more) {
        read_a(ch + 1, 0, anext);
        read_b(ch + 1, 0, bnext);
      }
      // the six partial products, smallest first: (A plane, B plane) = (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
      acc = q_mfma(acur[2], bcur[0], acc);
      acc = q_mfma(acur[0], bcur[2], acc);
      acc = q_mfma(acur[1], bcur[1], acc);
      acc = q_mfma(acur[1], bcur[0], acc);
      acc = q_mfma(acur[0], bcur[1], acc);
      acc = q_mfma(acur[0], bcur[0], acc);
      if (!last || more) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA, one read, ..
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!last || more) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          acur[p3] = anext[p3];
          bcur[p3] = bnext[p3];
        }
      }
    }
  }

  Q_STAMP(3);      // k-steps
  // ---- epilogue: activation, residual / mask, NCHW stores (lanes 0-31 / 32-63: 32 consecutive pixels of two channels 4 apart) ----
  // Every residual load is issued before the first store (vmcnt counts loads and stores in one queue).
  const int gx = x0 + l31, gy = y0 + wave;
  const bool pxok = gx < w && gy < h;
  const float act_s = a.act == EAVSR_ACT_RELU ? 0.f : a.act == EAVSR_ACT_LRELU ? a.slope : 1.f;
  const int co0 = cot * 32 + 4 * kg;
  if constexpr (VEC) {
    // activation and channel sums in the accumulator layout; then the tile is transposed through LDS (the patch region: eight
    // private 32-channel x 36-float tiles once every wave has read its last operands) so that residual loads and output stores
    // are float4 row segments: 4 + 4 instructions per lane instead of 16 + 16
    float vv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2);
      float v = acc[r];
      v = eavsr_act(v, act_s);      // branch-free: max(v, v s), 0 <= s <= 1
      vv[r] = v;
      if (a.chan_partial && !a.sum_mul) {
        float sum = (pxok && co0 + cu < a.cout) ? v : 0.f;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 1);
        if (l31 == 0) s_red[wave * 32 + cu + 4 * kg] = sum;
      }
    }
    const int q4 = lane & 7, rsub = lane >> 3;
    const int gx4 = x0 + 4 * q4;
    const bool pok4 = gx4 < w && gy < h;               // (w % 4 == 0: gx4 + 3 < w as well)
    f32x4 rr4[4], mm4[4];
    size_t o4[4];
    bool ok4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = cot * 32 + i * 8 + rsub;
      ok4[i] = pok4 && co < a.cout;
      o4[i] = ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx4;
      rr4[i] = (a.residual && ok4[i]) ? *reinterpret_cast<const f32x4*>(a.residual + o4[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      mm4[i] = (a.sum_mul && ok4[i]) ? *reinterpret_cast<const f32x4*>(a.sum_mul + o4[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (a.chan_partial && !a.sum_mul && tid < 32) {
      const int co = cot * 32 + tid;
      if (co < a.cout) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < Q_NW; ++k) v += s_red[k * 32 + tid];
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * a.cout + co] = v;
      }
    }
    float* tile = reinterpret_cast<float*>(s_patch) + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * kg) * 36 + l31] = vv[r];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (i * 8 + rsub) * 36 + 4 * q4);
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = a.act == EAVSR_ACT_RELU_MASK ? (rr4[i][c] > 0.f ? v[c] : 0.f) : v[c] + rr4[i][c];
      if (ok4[i]) *reinterpret_cast<f32x4*>(a.out + o4[i]) = o;
      if (a.sum_mul) {      // sum over the wave's 32 pixels of stored value x multiplier, per channel (i, rsub): 4 in the lane, 8 lanes
        float sm = (o[0] * mm4[i][0] + o[1] * mm4[i][1]) + (o[2] * mm4[i][2] + o[3] * mm4[i][3]);      // (a lane outside holds zeros in mm4)
        sm += __shfl_xor(sm, 1);
        sm += __shfl_xor(sm, 2);
        sm += __shfl_xor(sm, 4);
        if (q4 == 0) s_red[wave * 32 + i * 8 + rsub] = sm;
      }
    }
    if (a.sum_mul) {      // the tile's eight pixel rows, in wave order (deterministic)
      __syncthreads();
      if (tid < 32) {
        const int co = cot * 32 + tid;
        if (co < a.cout) {
          float v = s_red[tid];
#pragma unroll
          for (int k = 1; k < Q_NW; ++k) v += s_red[k * 32 + tid];
          a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * a.cout + co] = v;
        }
      }
    }
#ifdef EAVSR_X6S_STAMPS
    Q_STAMP(4);      // epilogue issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q_STAMP(5);      // stores acknowledged
    if (tid == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&g_q_stamps[i], st_acc[i]);
#endif
    return;
  }
  const size_t obase = ((size_t)bn * a.cout + co0) * plane + (size_t)gy * w + gx;
  float rr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int cu = (r & 3) + 8 * (r >> 2);
    rr[r] = (a.residual && pxok && co0 + cu < a.cout) ? a.residual[obase + (size_t)cu * plane] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int cu = (r & 3) + 8 * (r >> 2);
    const bool cok = co0 + cu < a.cout;
    float v = acc[r];
    v = eavsr_act(v, act_s);      // branch-free: max(v, v s), 0 <= s <= 1
    float sum = 0.f;
    if (cok && pxok) {
      sum = v;
      a.out[obase + (size_t)cu * plane] = a.act == EAVSR_ACT_RELU_MASK ? (rr[r] > 0.f ? v : 0.f) : v + rr[r];
    }
    if (a.chan_partial) {
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 8);
      sum += __shfl_xor(sum, 4);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 1);
      if (l31 == 0) s_red[wave * 32 + cu + 4 * kg] = sum;
    }
  }
#ifdef EAVSR_X6S_STAMPS
  Q_STAMP(4);      // epilogue issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Q_STAMP(5);      // stores acknowledged
  if (tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_q_stamps[i], st_acc[i]);
#endif
  if (a.chan_partial) {
    __syncthreads();
    if (tid < 32) {
      const int co = cot * 32 + tid;
      if (co < a.cout) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < Q_NW; ++k) v += s_red[k * 32 + tid];
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * a.cout + co] = v;
      }
    }
  }
}

template <int NCH, bool VEC>
int launch_q(const QArgs& a, void* stream) {
  using K = QCfg<NCH>;
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once_pd.flag[dev_], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x6s_kernel<NCH, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)K::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_f32x6s: hipFuncSetAttribute(%zu B of LDS): %s", K::LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  const long blocks = (long)a.tiles_x * a.tiles_y * a.n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_f32x6s: too many tiles");
  dim3 grid((unsigned)blocks, eavsr::cdiv(a.cout, 32));
  hipLaunchKernelGGL((conv3x3_x6s_kernel<NCH, VEC>), grid, dim3(64 * Q_NW), K::LDS_BYTES, eavsr::as_stream(stream), a);
  return eavsr::launch_status("conv3x3_f32x6s");
}

}  // namespace

#ifdef EAVSR_X6S_STAMPS
extern "C" int eavsr_debug_x6s_stamps(unsigned long long* host_out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_q_stamps), sizeof(unsigned long long) * 8);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_q_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif

extern "C" int32_t eavsr_conv3x3_x6s_tiles(int32_t h, int32_t w) {
  if (h <= 0 || w <= 0) return 0;
  return eavsr::cdiv(h, Q_TH) * eavsr::cdiv(w, Q_TW);
}

extern "C" int eavsr_conv3x3_f32x6s(const eavsr_conv2d_desc* d, const void* weight_x6, void* stream) {
  EAVSR_REQUIRE(d && weight_x6, -1, "conv3x3_f32x6s: NULL pointer");
  EAVSR_REQUIRE(d->ksize == 3 && d->n_src == 1 && d->src_c[0] == d->cin, -2, "conv3x3_f32x6s: one source, 3x3");
  EAVSR_REQUIRE(d->cin == 64, -2, "conv3x3_f32x6s: cin %d (64 only; use eavsr_conv2d_f32)", d->cin);
  EAVSR_REQUIRE(d->n >= 0 && d->cout > 0 && d->h > 0 && d->w > 0, -1, "conv3x3_f32x6s: bad dims");
  if (d->n == 0) return 0;
  EAVSR_REQUIRE(d->src[0] && d->out, -1, "conv3x3_f32x6s: NULL pointer");
  EAVSR_REQUIRE(d->ca_scale == nullptr && d->ca_x == nullptr && d->ca_out == nullptr && d->out_shuffle == 0 && d->res_scale == nullptr && d->border_pieces == nullptr, -2,
                "conv3x3_f32x6s: no channel-attention prologue, no pixel-shuffle store");
  EAVSR_REQUIRE(d->act >= 0 && d->act <= EAVSR_ACT_RELU_MASK, -1, "conv3x3_f32x6s: act %d", d->act);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_LRELU || (d->slope >= 0.f && d->slope <= 1.f), -2,
                "conv3x3_f32x6s: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)d->slope);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_RELU_MASK || (d->residual != nullptr && d->chan_partial == nullptr), -1,
                "conv3x3_f32x6s: EAVSR_ACT_RELU_MASK takes the mask source in `residual` (no channel sums)");
  EAVSR_REQUIRE(d->sum_mul == nullptr || d->act != EAVSR_ACT_RELU_MASK, -1, "conv3x3_f32x6s: sum_mul with EAVSR_ACT_RELU_MASK");
  EAVSR_REQUIRE((long)d->h * d->w < (1L << 31), -1, "conv3x3_f32x6s: image plane too large for 32-bit pixel offsets");
  QArgs a;
  a.x = d->src[0]; a.wsplit = reinterpret_cast<const u32x4*>(weight_x6); a.bias = d->bias; a.residual = d->residual;
  a.out = d->out; a.chan_partial = d->chan_partial; a.sum_mul = d->sum_mul;
  a.n = d->n; a.cout = d->cout; a.h = d->h; a.w = d->w;
  a.tiles_x = eavsr::cdiv(d->w, Q_TW); a.tiles_y = eavsr::cdiv(d->h, Q_TH);
  a.wmt = d->cout > 32 ? 2 : 1;
  a.act = d->act; a.slope = d->slope;
  const bool vec = d->w % 4 == 0 && ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.out) |
                                       reinterpret_cast<uintptr_t>(a.residual)) & 15) == 0;
  EAVSR_REQUIRE(a.sum_mul == nullptr || (a.chan_partial != nullptr && vec && (reinterpret_cast<uintptr_t>(a.sum_mul) & 15) == 0), -2,
                "conv3x3_f32x6s: sum_mul needs chan_partial, w %% 4 == 0 and 16-byte aligned tensors");
#ifdef EAVSR_X6S_NO_VEC
  return launch_q<8, false>(a, stream);
#endif
  return vec ? launch_q<8, true>(a, stream) : launch_q<8, false>(a, stream);
}
