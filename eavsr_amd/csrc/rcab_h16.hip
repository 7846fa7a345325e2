// conv -> ReLU -> conv of one RCAB in the 16-bit modes as ONE kernel (BASELINE.json configs[2] bf16 / configs[4] fp16; SURVEY.md 8a:
// a11; SURVEY.md 7 "hard parts": "in bf16 a single 3x3 64->64 conv is HBM-bound: fuse conv-ReLU-conv of an RCAB through LDS with a
// 2-px halo").
//
// Reference: RCABlock.forward, models/networks.py:461-462 (mode 'CRC'): r = conv3x3(ReLU(conv3x3(x))), both 64 -> 64, bias, zero
// padding -- the intermediate t = ReLU(conv1(x)) is zero OUTSIDE THE IMAGE for the second convolution (its own padding), not the
// first convolution evaluated there.
//
// Why a new kernel and not a mode of csrc/conv_h16.hip: that kernel keeps the 73.7 KB weight matrix resident in LDS, which leaves
// room for one patch stage per wave group and makes it latency-bound (a phase lasts one patch round trip: 0.31 / 0.39 of HBM,
// 0.28 / 0.36 of the MFMA peak).  Two weight sets + the input and the intermediate patch do not fit 160 KB at all.  Here the
// weights are STREAMED: both convolutions' k-steps travel from L2 through a two-slot LDS ring, one filter row (12 k-steps,
// 24 KB) per slot, requested one compute phase ahead; the input patch (12 x 36 pixels) and the intermediate (10 x 34) live in LDS;
// HBM sees one read of the input tile (+ halo, from L2) and one write of r per RCAB instead of two of each.
//
//   tile = 8 x 32 output pixels; conv-1 is evaluated on the 10 x 34 pixels conv-2 reads (340 "t pixels", 11 MFMA N-tiles of 32 in
//   row-major order, the last one partly empty): 22 (N-tile, M-tile) units over 8 waves -- wave w owns output-channel half m = w & 1
//   and N-tiles (w >> 1) + 4 j; conv-2: wave w owns pixel row w, both halves.  Operand conventions, LDS patch layout
//   ([row][col][8 x 16-byte blocks], block index XOR ((col >> 1) & 7)), packed weights and the epilogue's lane layout are those of
//   csrc/conv_h16.hip; the k-steps are accumulated in the same order (ky, kx, 16-channel block), so r equals the two-launch
//   result bit for bit.
//   Every global access is a buffer instruction: patch units outside the image read zeros through the range check (the
//   convolution's zero padding: no fill code, no border branch), stores of lanes without a pixel are dropped by it.
#include "common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <mutex>
#include <type_traits>

namespace {

typedef _Float16 r_h16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* r_lptr;
typedef __amdgpu_buffer_rsrc_t r_rsrc;
typedef unsigned r_u32x4 __attribute__((ext_vector_type(4)));

constexpr int RT_H = 8, RT_W = 32;                    // output tile
constexpr int RI_H = RT_H + 4, RI_W = RT_W + 4;       // input patch 12 x 36
constexpr int RM_H = RT_H + 2, RM_W = RT_W + 2;       // intermediate 10 x 34
constexpr int RI_PIX = RI_H * RI_W;                   // 432 pixels = 54 one-KiB DMA pieces exactly
constexpr int RM_PIX = RM_H * RM_W;                   // 340
constexpr int RI_BYTES = RI_PIX * 128;                // 55,296
constexpr int RM_BYTES = RM_PIX * 128;                // 43,520
constexpr int RI_SEGS = RI_BYTES / 1024;              // 54
constexpr int RG_KS = 9;                              // k-steps of one weight granule (a convolution = 4 granules)
constexpr int RW_GRAN = RG_KS * 2048;                 // 18,432 B = 18 one-KiB DMA pieces
constexpr int RW_CONV = 4 * RW_GRAN;                  // 73,728
constexpr int RW_SLOTS = 3;
constexpr int RL_IN = 0;
constexpr int RL_MID = RL_IN + RI_BYTES;
constexpr int RL_W = RL_MID + RM_BYTES;               // three ring slots
constexpr int RL_RED = RL_W + RW_SLOTS * RW_GRAN;     // [8 waves][64] channel sums
constexpr int RL_BIAS = RL_RED + 8 * 64 * 4;          // [2][64]
constexpr int R_LDS_BYTES = RL_BIAS + 2 * 64 * 4;     // 156,672
constexpr int RN1 = (RM_PIX + 31) / 32;               // 11 N-tiles of conv-1
constexpr unsigned R_OOB = 0x80000000u;

#ifdef EAVSR_RCAB_STAMPS
// diagnostic build only: shader cycles per phase, summed over wave 0 of every workgroup
__device__ unsigned long long g_rcab_stamps[16];
#define RC_STAMP(i)                                                   \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    st_acc[i] += t_ - st_last;                                        \
    st_last = t_;                                                     \
  } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif

struct RcabArgs {
  const void* x;       // (n, h, w, 64) 16-bit
  const void* wp1;     // packed weights of conv-1: [36 k-steps][2 halves][64 co][8] (eavsr_pack_conv3x3_c64_h16)
  const void* wp2;     // ... of conv-2
  const float* bias1;  // fp32 [64] or NULL
  const float* bias2;
  void* out;           // r: (n, h, w, 64) 16-bit
  float* chan_partial; // (n, gridDim.x, 64) fp32: row b of sample s = workgroup b's channel sums of r (the 16-bit values) over its
                       // tiles of that sample (zeros where it has none), or NULL
  int n, h, w, tiles_x, tiles_y, num_tiles;
  int tiles_base, tiles_rem;   // workgroup b owns tiles_base + (b < tiles_rem) CONSECUTIVE tiles (neighbours share halos in L2, and a
                               // workgroup's tiles belong to one sample except at a sample boundary)
};

template <bool BF16> __device__ __forceinline__ unsigned short r_to_h16(float v);
template <> __device__ __forceinline__ unsigned short r_to_h16<true>(float v) { return __builtin_bit_cast(unsigned short, __float2bfloat16(v)); }
template <> __device__ __forceinline__ unsigned short r_to_h16<false>(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
template <bool BF16> __device__ __forceinline__ float r_from_h16(unsigned short v);
template <> __device__ __forceinline__ float r_from_h16<true>(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
template <> __device__ __forceinline__ float r_from_h16<false>(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }

template <bool BF16>
__device__ __forceinline__ f32x16 r_mfma(const f32x4& a, const f32x4& b, const f32x16& c) {
#ifdef EAVSR_RCAB_EXP_NO_MFMA      // timing ablation (results wrong)
  f32x16 d = c;
  d[0] += a[0] + b[1];
  return d;
#endif
  if (BF16) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(r_h16x8, a), __builtin_bit_cast(r_h16x8, b), c, 0, 0, 0);
  }
}

#define R_ADD_DPP(v, ctrl) asm volatile("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))

__device__ __forceinline__ f32x4 r_lds4(unsigned addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(addr);
}

template <bool BF16>
__global__ __launch_bounds__(512, 2) void rcab_convs_h16_kernel(RcabArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(unsigned long long)(r_lptr)smem;
  float* s_red = reinterpret_cast<float*>(smem + RL_RED);
  float* s_bias = reinterpret_cast<float*>(smem + RL_BIAS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = a.h, w = a.w;
  const int m1 = wave & 1, nq = wave >> 1;            // conv-1: output-channel half and first N-tile of this wave
  const int nu1 = nq + 8 < RN1 ? 3 : 2;               // its N-tiles: nq, nq + 4, (nq + 8)

#if defined(EAVSR_RCAB_PRIO)
  if (wave < 4) __builtin_amdgcn_s_setprio(EAVSR_RCAB_PRIO);      // A/B: prefer the first-dispatched half (as csrc/dcnv2_il2.hip)
#elif defined(EAVSR_RCAB_PRIO_HIGH)
  if (wave >= 4) __builtin_amdgcn_s_setprio(EAVSR_RCAB_PRIO_HIGH);  // A/B: prefer the second-dispatched half (the arbitration loser)
#endif
  if (tid < 128) s_bias[tid] = tid < 64 ? (a.bias1 ? a.bias1[tid] : 0.f) : (a.bias2 ? a.bias2[tid - 64] : 0.f);

  const r_rsrc r_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp1), 0, RW_CONV, 0x00020000);
  const r_rsrc r_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp2), 0, RW_CONV, 0x00020000);
  const unsigned img_bytes = (unsigned)h * (unsigned)w * 128u;
  const unsigned wv = (unsigned)lane * 16u;
  // granule `q` (0..3 conv-1, 4..7 conv-2) into ring slot `slot`: 18 one-KiB pieces, piece i * 8 + wave (waves 0, 1 three, the
  // others two).  Issued one piece at a time between the MFMAs of a granule (an LDS-DMA instruction costs 60-185 issue cycles).
  auto req_w1 = [&](int q, int slot, int i) __attribute__((always_inline)) {
#ifdef EAVSR_RCAB_EXP_NO_WDMA      // timing ablation (results wrong)
    if (a.n >= 0) return;
#endif
    const r_rsrc rw = q < 4 ? r_w1 : r_w2;
    const unsigned so = (unsigned)((q & 3) * RW_GRAN);
    const int piece = i * 8 + wave;
    if (i < 2 || wave < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (r_lptr)(smem + RL_W + slot * RW_GRAN + piece * 1024), 16, wv, so + (unsigned)(piece * 1024), 0, 0);
  };
  auto req_w = [&](int q, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) req_w1(q, slot, i);
  };

  // ---- input patch: unit e = piece * 64 + lane -> pixel e >> 3 of the 12 x 36 patch, stored block e & 7 holds the logical block
  // (e & 7) ^ swz(col); piece = i * 8 + wave (54 pieces: i = 0..6, the last round waves 0..5)
  // (the per-lane geometry of a piece is recomputed where it is requested: 21 registers for ~10 instructions per piece and tile)
  const int t_first = (int)blockIdx.x * a.tiles_base + min((int)blockIdx.x, a.tiles_rem);
  auto tile_at = [&](int j, int& bn, int& ty, int& tx) __attribute__((always_inline)) {
    int t = t_first + j;
    tx = t % a.tiles_x;
    t /= a.tiles_x;
    ty = t % a.tiles_y;
    bn = t / a.tiles_y;
  };
  // the next tile's patch: context set once (resource of its image, patch origin), then one piece per call
  r_rsrc rx_n;
  int py0_n = 0, px0_n = 0;
  auto patch_ctx = [&](int j) __attribute__((always_inline)) {      // (j past the run: the last tile once more, never read)
    int bn, ty, tx;
    tile_at(j, bn, ty, tx);
    py0_n = ty * RT_H - 2;
    px0_n = tx * RT_W - 2;
    rx_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x) + (size_t)bn * img_bytes), 0,
                                             (int)img_bytes, 0x00020000);
  };
  auto req_patch1 = [&](int i) __attribute__((always_inline)) {
    if (i < 6 || wave < RI_SEGS - 48) {
      const int e = (i * 8 + wave) * 64 + lane;
      const int pp = e >> 3, sb = e & 7;
      const int pr = pp / RI_W, pc = pp - pr * RI_W;
      const int gy = py0_n + pr, gx = px0_n + pc;
      const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
      const unsigned vo = ok ? (unsigned)(gy * w + gx) * 128u + (unsigned)((sb ^ ((pc >> 1) & 7)) << 4) : R_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_n, (r_lptr)(smem + RL_IN + (i * 8 + wave) * 1024), 16, vo, 0, 0, 0);
    }
  };

  const int cnt = a.tiles_base + ((int)blockIdx.x < a.tiles_rem ? 1 : 0);   // tiles of this workgroup
  if (cnt <= 0) return;

  // ---- per-lane operand addresses ------------------------------------------------------------------------------------------
  // conv-1, N-tile u: t pixel p = n * 32 + l31 (clamped into the region), its window origin in the input patch = (tr, tc)
  unsigned b1a[3][3];         // [unit][kx]: address of the lane's operand at filter row 0, 16-channel block 0; block cb = this ^ (cb << 5)
  int trc1[3];                // (tr << 8) | tc of the lane's pixel of unit u; -1: the lane stores nothing
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int p = (nq + 4 * u) * 32 + l31;      // (waves 6, 7: the third N-tile does not exist; clamped, never stored)
    const int pc = min(p, RM_PIX - 1);
    const int tr = pc / RM_W, tc = pc - tr * RM_W;
    trc1[u] = (u < nu1 && p < RM_PIX) ? ((tr << 8) | tc) : -1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      b1a[u][kx] = lds0 + (unsigned)(RL_IN + ((tr * RI_W + tc + kx) << 7) + ((half ^ (((tc + kx) >> 1) & 7)) << 4));
  }
  // conv-2: pixel (row `wave`, column l31) of the tile reads the intermediate at (wave + ky, l31 + kx)
  unsigned b2a[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    b2a[kx] = lds0 + (unsigned)(RL_MID + ((wave * RM_W + l31 + kx) << 7) + ((half ^ (((l31 + kx) >> 1) & 7)) << 4));
  // weights: [k-step][half][co][8]: lane (l31, half) supplies row co = m * 32 + l31
  const unsigned wl = lds0 + (unsigned)(RL_W + ((half * 64 + l31) << 4));

  // Channel sums of r for the channel attention: every lane keeps the sums of ITS pixel column over the workgroup's tiles of the
  // current sample (32 registers); the cross-lane / cross-wave reduction (160 DPP additions, LDS, a barrier) runs only when the
  // sample changes or the run ends -- not once per tile as in conv_h16.hip, where it is half of the epilogue's instructions.
  float csum[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) csum[m][e] = 0.f;
  int first_bn = -1;
  auto flush_sums = [&](int bn_) __attribute__((always_inline)) {      // all waves; wave-uniform; ends with the sums cleared
#define R_DPP_STEP(ctrl)                                     \
  _Pragma("unroll") for (int m = 0; m < 2; ++m)              \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) R_ADD_DPP(csum[m][e], ctrl);
    asm volatile("s_nop 1");
    R_DPP_STEP("row_shr:1")
    R_DPP_STEP("row_shr:2")
    R_DPP_STEP("row_shr:4")
    R_DPP_STEP("row_shr:8")
    R_DPP_STEP("row_bcast:15")
#undef R_DPP_STEP
    if (l31 == 31) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) s_red[wave * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = csum[m][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave == 0) {
      float v = s_red[lane];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += s_red[k * 64 + lane];
      a.chan_partial[((size_t)bn_ * gridDim.x + blockIdx.x) * 64 + lane] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) csum[m][e] = 0.f;
  };

  // ---- prologue -------------------------------------------------------------------------------------------------------------
  // Weight granule G (9 k-steps, 18 KB; G0..G3 conv-1, G4..G7 conv-2) lives in ring slot G % 3.  The workgroup meets ONCE per
  // granule, in its MIDDLE (k-step 4): every wave has waited for its own pieces of granule G + 1 (requested half a granule
  // after the previous meeting) and everybody is past granule G - 1, whose slot the requests for G + 2 -- issued one piece at a
  // time behind the MFMAs that follow -- then overwrite.  Nothing drains at a granule boundary: the operand reads of a k-step are
  // issued one k-step ahead straight across it.  Vector-memory operations of a wave in issue order, per tile (every request is
  // UNCONDITIONAL -- past the end of the run the last tile is re-requested and never read -- so the counted waits hold everywhere):
  //   mid G0: weights G2;  G1: G3;  G2: G4;  G3: G5;  G4: G6;  G5: G7, then the NEXT tile's input patch;  G6: G0';  G7: G1';
  //   epilogue: 4 stores.   At mid Gk the youngest operations behind the request for G(k+1) are: k = 0: the 4 stores; k = 6: the
  //   patch (>= 6 pieces); otherwise none.
  patch_ctx(0);
#pragma unroll
  for (int i = 0; i < 7; ++i) req_patch1(i);
  req_w(0, 0);
  req_w(1, 1);
  int slot = 0;      // ring slot of the granule being multiplied
  const int cnt_last = cnt - 1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#ifdef EAVSR_RCAB_EXP_NO_READS     // timing ablation (results wrong): no operand reads, the waits find nothing outstanding
#define R_DSR(dst, addr, off) asm volatile("v_mov_b32 %0, %1" : "=v"(dst[0]) : "v"(addr))
#else
#define R_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#endif
#define R_FENCE() __builtin_amdgcn_sched_barrier(0)
#define R_MID(N)                                                  \
  do {                                                            \
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");     \
    __builtin_amdgcn_s_barrier();                                 \
    asm volatile("" ::: "memory");                                \
  } while (0)

  // The epilogue of conv-2 (bias, rounding, the lane pair's exchange, four 16-byte stores, per-lane channel sums) is ~250 vector
  // instructions per wave that all eight waves would execute together with the matrix pipe idle (measured: 30 % of the launch).
  // It is DEFERRED: a tile's accumulators stay in registers and its four chunks run behind the first MFMAs of the NEXT tile's
  // conv-1 (k-steps 0..3 of granule 0, ahead of that granule's meeting, so the counted waits see the stores where they were).
  f32x16 acc2[2];
  r_rsrc ro_p = r_w1;      // the pending tile's output resource, lane offset (R_OOB: no pixel) and sample
  unsigned vo_p = R_OOB;
  bool ok_p = false;
  int bn_p = -1;
  auto epi2_chunk = [&](int c) __attribute__((always_inline)) {
    const int m = c >> 1, qe = (c & 1) * 2;
    unsigned dw[2][2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int qd = qe + q2;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + 64 + m * 32 + 8 * qd + 4 * half);
      unsigned short pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pk[e] = r_to_h16<BF16>(acc2[m][4 * qd + e] + b4[e]);
        if (a.chan_partial) csum[m][4 * qd + e] += ok_p ? r_from_h16<BF16>(pk[e]) : 0.f;      // the 16-bit value the next layer reads
      }
      dw[q2][0] = (unsigned)pk[0] | ((unsigned)pk[1] << 16);
      dw[q2][1] = (unsigned)pk[2] | ((unsigned)pk[3] << 16);
    }
    // the lane pair (l31, half 0 | 1) trades 4-channel runs so that a lane stores 16 contiguous bytes (conv_h16.hip)
    const auto s0 = __builtin_amdgcn_permlane32_swap(dw[0][0], dw[1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(dw[0][1], dw[1][1], false, false);
    const r_u32x4 v4 = {s0[0], s1[0], s0[1], s1[1]};
    __builtin_amdgcn_raw_buffer_store_b128(v4, ro_p, vo_p + (unsigned)((m * 32 + 8 * (qe + half)) * 2), 0, 0);
  };

#ifdef EAVSR_RCAB_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  for (int j = 0; j < cnt; ++j) {
    int bn, ty, tx;
    tile_at(j, bn, ty, tx);
    const int jn = min(j + 1, cnt_last);
    RC_STAMP(0);      // tile bookkeeping
    // ================================================================ conv-1 on the 10 x 34 intermediate region ===============
    // Operand reads are hand-placed inline assembly with COUNTED waits (as in conv_h16.hip: left to the compiler every MFMA waits
    // `lgkmcnt(0)` behind its own fresh read -- an exposed LDS round trip per MFMA).  A k-step's four reads (weights of this
    // wave's channel half, pixels of its three N-tiles) are issued one k-step ahead, one behind each MFMA; the counter is in
    // order, so "at most two younger reads outstanding" is exactly "mine have landed".  Waves 6 and 7 own two N-tiles: their
    // third unit multiplies a clamped address and is never stored (its SIMD partner is busy with three units anyway).
    f32x16 acc1[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[u][e] = 0.f;
    // Operands are requested TWO k-steps ahead (three register sets): with one k-step the wave waited for LDS at every k-step
    // (eight waves keep ~30 reads queued: the round trip is longer than two MFMAs).  The reads are issued in a fixed order --
    // k-step i: weights, pixels of unit 0, 1, 2 = reads 4 i .. 4 i + 3 -- and every MFMA waits for "all but the N youngest", N
    // counted at compile time from that order.
    f32x4 opa[3], opb[3][3];
    int issued1 = 0;      // (folds to constants in the unrolled code)
    auto rd1 = [&](int sg, int which, unsigned wcur, unsigned wnxt, int g_now) __attribute__((always_inline)) {
      // read `which` (0: weights, 1..3: pixels of unit which - 1) of conv-1 k-step sg
      const int S = sg % 3, ky = sg / 12, kx = (sg % 12) >> 2, cb = sg & 3, gk = sg / RG_KS, ko = sg - gk * RG_KS;
      if (which == 0) {
        if (gk == g_now) R_DSR(opa[S], wcur, ko * 2048); else R_DSR(opa[S], wnxt, ko * 2048);
      } else {
        // block cb = base ^ (cb << 5), computed AT the read inside one asm statement (hoisted by the compiler the 36 addresses are
        // 27 more live registers; as a statement of its own it gets a hazard s_nop in front of the read)
        if (cb == 0) {
          R_DSR(opb[S][which - 1], b1a[which - 1][kx], ky * RI_W * 128);
        } else {
          unsigned ad;
          asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                       : "=v"(opb[S][which - 1]), "=&v"(ad) : "v"(b1a[which - 1][kx]), "i"(cb << 5), "i"(ky * RI_W * 128));
        }
      }
      ++issued1;
    };
    {
      const unsigned wb0 = wl + (unsigned)(slot * RW_GRAN) + (unsigned)(m1 * 32 * 16);
      R_FENCE();
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int q = 0; q < 4; ++q) rd1(sg, q, wb0, wb0, 0);
      R_FENCE();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nslot = slot == 2 ? 0 : slot + 1, rslot = slot == 0 ? 2 : slot - 1;      // of granules g + 1 and g + 2 (= g - 1)
      const unsigned wbase = wl + (unsigned)(slot * RW_GRAN) + (unsigned)(m1 * 32 * 16);
      const unsigned wnext = wl + (unsigned)(nslot * RW_GRAN) + (unsigned)(m1 * 32 * 16);
#pragma unroll
      for (int ks = 0; ks < RG_KS; ++ks) {
        const int sg = g * RG_KS + ks;      // k-step of the convolution
        const int S = sg % 3;
        const bool nxt = sg + 2 < 36;       // the convolution's last two k-steps prefetch nothing
        if (ks == 4) {
          RC_STAMP(1);      // conv-1 k-steps
#ifndef EAVSR_RCAB_EPI_LOCKSTEP
          if (g == 0 && wave < 4) R_MID(4); else R_MID(0);      // (waves 4..7 issue their four stores behind this meeting)
#else
          if (g == 0) R_MID(4); else R_MID(0);
#endif
          RC_STAMP(2);      // conv-1 meetings (own DMA share + the other waves)
          R_FENCE();
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          // needs reads 4 sg (weights) and 4 sg + 1 + u: everything younger may be outstanding
          // (no register operands on the wait: tied to the operand registers the compiler treats it as their producer and puts a
          // hazard s_nop in front of every MFMA; the scheduling fences on both sides keep the order)
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(issued1 - 1 - (4 * sg + 1 + u)));
          R_FENCE();
          acc1[u] = r_mfma<BF16>(opa[S], opb[S][u], acc1[u]);
          R_FENCE();
          if (u == 1 && ks >= 5 && ks <= 7) { req_w1(g + 2, rslot, ks - 5); R_FENCE(); }
          // the previous tile's epilogue (wave-uniform): the two waves of a SIMD (w, w + 4) run it at DIFFERENT k-steps -- in
          // lockstep both would leave the matrix pipe idle together
#ifndef EAVSR_RCAB_EPI_LOCKSTEP
          if (u == 2 && g == 0 && j > 0) {      // (the chunk index must be a constant: it selects accumulator registers)
            if (ks < 4) { if (wave < 4) epi2_chunk(ks); }
            else if (ks >= 5) { if (wave >= 4) epi2_chunk(ks - 5); }
            R_FENCE();
          }
#else
          if (u == 2 && g == 0 && ks < 4 && j > 0) { epi2_chunk(ks); R_FENCE(); }
#endif
          if (nxt) {
            if (u == 0) rd1(sg + 2, 0, wbase, wnext, g);
            if (u == 1) rd1(sg + 2, 1, wbase, wnext, g);
            if (u == 2) { rd1(sg + 2, 2, wbase, wnext, g); rd1(sg + 2, 3, wbase, wnext, g); }
            R_FENCE();
          }
        }
      }
      slot = nslot;
    }
    RC_STAMP(1);
    if (a.chan_partial && j > 0 && bn_p != bn) flush_sums(bn_p);      // (wave-uniform) the previous tile ended its sample
    // ---- t = ReLU(conv-1 + bias) rounded to 16 bits into the intermediate patch; zero outside the image ------------------------
    {
      const int gy0 = ty * RT_H - 1, gx0 = tx * RT_W - 1;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        {
          const int p = (nq + 4 * u) * 32 + l31;
          const bool pv = trc1[u] >= 0;
          const int tr = trc1[u] >> 8, tc = trc1[u] & 255;
          const int gy = gy0 + tr, gx = gx0 + tc;
          // a pixel of the intermediate outside the image is ZERO (the second convolution's padding): mask of the packed words
          const unsigned keep = ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w) ? 0xFFFFFFFFu : 0u;
          const unsigned pbase = lds0 + (unsigned)(RL_MID + (p << 7)) + (unsigned)(half * 8);
          const int swz = (tc >> 1) & 7;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + m1 * 32 + 8 * qd + 4 * half);
            unsigned short pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = r_to_h16<BF16>(fmaxf(acc1[u][4 * qd + e] + b4[e], 0.f));
            const unsigned d0 = ((unsigned)pk[0] | ((unsigned)pk[1] << 16)) & keep, d1 = ((unsigned)pk[2] | ((unsigned)pk[3] << 16)) & keep;
            if (pv) {
              typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
              *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>(pbase + (unsigned)(((m1 * 4 + qd) ^ swz) << 4)) = u32x2_t{d0, d1};
            }
          }
        }
      }
    }
    // ================================================================ conv-2 on the tile ========================================
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[m][e] = 0.f;
    f32x4 qb[3], qa[3][2];
    int issued2 = 0;
    auto rd2 = [&](int sg, int which, unsigned wcur, unsigned wnxt, int g_now) __attribute__((always_inline)) {
      // read `which` (0: pixels, 1 / 2: weights of channel half 0 / 1) of conv-2 k-step sg
      const int S = sg % 3, ky = sg / 12, kx = (sg % 12) >> 2, cb = sg & 3, gk = sg / RG_KS, ko = sg - gk * RG_KS;
      if (which == 0) {
        if (cb == 0) {
          R_DSR(qb[S], b2a[kx], ky * RM_W * 128);
        } else {
          unsigned ad;
          asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                       : "=v"(qb[S]), "=&v"(ad) : "v"(b2a[kx]), "i"(cb << 5), "i"(ky * RM_W * 128));
        }
      } else {
        if (gk == g_now) R_DSR(qa[S][which - 1], wcur, ko * 2048 + (which - 1) * 512);
        else R_DSR(qa[S][which - 1], wnxt, ko * 2048 + (which - 1) * 512);
      }
      ++issued2;
    };
    RC_STAMP(3);      // intermediate written
    // the intermediate is complete and visible, the conv-1 reads of the input patch are over (the one drain of a tile)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RC_STAMP(4);      // the barrier behind it
    {
      const unsigned wb0 = wl + (unsigned)(slot * RW_GRAN);
      R_FENCE();
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int q = 0; q < 3; ++q) rd2(sg, q, wb0, wb0, 0);
      R_FENCE();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nslot = slot == 2 ? 0 : slot + 1, rslot = slot == 0 ? 2 : slot - 1;
      const int rq = g < 2 ? g + 6 : g - 2;      // requested in this granule: G6, G7, then the next tile's G0, G1
      const unsigned wbase = wl + (unsigned)(slot * RW_GRAN);
      const unsigned wnext = wl + (unsigned)(nslot * RW_GRAN);
#pragma unroll
      for (int ks = 0; ks < RG_KS; ++ks) {
        const int sg = g * RG_KS + ks;
        const int S = sg % 3;
        const bool nxt = sg + 2 < 36;
        if (ks == 4) {
          RC_STAMP(5);      // conv-2 k-steps
          if (g == 2) R_MID(6); else R_MID(0);
          RC_STAMP(6);      // conv-2 meetings
          if (g == 1) patch_ctx(jn);      // the input patch is free since the barrier behind conv-1
          R_FENCE();
        }
        // k-step sg = reads 3 sg (pixels), 3 sg + 1, 3 sg + 2 (weights of channel half 0, 1)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(issued2 - 1 - (3 * sg + 1)));
        R_FENCE();
        acc2[0] = r_mfma<BF16>(qa[S][0], qb[S], acc2[0]);
        R_FENCE();
        if (nxt) { rd2(sg + 2, 0, wbase, wnext, g); R_FENCE(); }
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(issued2 - 1 - (3 * sg + 2)));
        R_FENCE();
        acc2[1] = r_mfma<BF16>(qa[S][1], qb[S], acc2[1]);
        R_FENCE();
        // this granule's requests, one piece behind an MFMA: the weights first, then (G5) the next tile's input patch
        if (ks >= 5 && ks <= 7) { req_w1(rq, rslot, ks - 5); R_FENCE(); }
        if (nxt) { rd2(sg + 2, 1, wbase, wnext, g); rd2(sg + 2, 2, wbase, wnext, g); R_FENCE(); }
      }
      if (g == 1) {      // G5's second request set: the next tile's input patch, behind the weights of G7
#pragma unroll
        for (int i = 0; i < 7; ++i) req_patch1(i);
      }
      slot = nslot;
    }
    RC_STAMP(5);
    // ---- this tile's epilogue is pending: it runs behind the next tile's first MFMAs (or after the loop) ---------------------
    {
      const int gy = ty * RT_H + wave, gx = tx * RT_W + l31;
      ok_p = gy < h && gx < w;
      ro_p = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (size_t)bn * img_bytes, 0, (int)img_bytes, 0x00020000);
      vo_p = ok_p ? (unsigned)(gy * w + gx) * 128u : R_OOB;
      bn_p = bn;
      if (first_bn < 0) first_bn = bn;
    }
  }
  // the last tile's epilogue and its sample's sums
  RC_STAMP(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) epi2_chunk(c);
  if (a.chan_partial) flush_sums(bn_p);
  RC_STAMP(7);      // last epilogue + sums
#ifdef EAVSR_RCAB_STAMPS
  if (lane == 0 && (wave == 0 || wave == 4))
    for (int i = 0; i < 8; ++i) atomicAdd(&g_rcab_stamps[(wave == 4 ? 8 : 0) + i], st_acc[i]);
#endif
#undef R_DSR
#undef R_FENCE
#undef R_MID
  // nothing of this workgroup may be in flight into its LDS when it ends (the last tile's unconditional requests)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (a.chan_partial && wave == 0) {      // zero rows for the samples this workgroup has no tile of
    int bnl, tyl, txl;
    tile_at(cnt - 1, bnl, tyl, txl);
    for (int sb = 0; sb < a.n; ++sb)
      if (sb < first_bn || sb > bnl) a.chan_partial[((size_t)sb * gridDim.x + blockIdx.x) * 64 + lane] = 0.f;
  }
}

template <bool BF16>
int launch_rcab_h16(const RcabArgs& a, int blocks, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&rcab_convs_h16_kernel<BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("rcab_convs_h16: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL(rcab_convs_h16_kernel<BF16>, dim3(blocks), dim3(512), R_LDS_BYTES, st, a);
  return eavsr::launch_status("rcab_convs_h16");
}

}  // namespace

#ifdef EAVSR_RCAB_STAMPS
extern "C" int eavsr_debug_rcab_stamps(unsigned long long* host_out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rcab_stamps), sizeof(g_rcab_stamps));
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_rcab_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

// rows per sample of eavsr_rcab_convs_h16's chan_partial: one per workgroup of the launch
extern "C" int32_t eavsr_rcab_h16_partial_rows(int32_t n, int32_t h, int32_t w) {
  const long tiles = (long)eavsr::cdiv(h, RT_H) * eavsr::cdiv(w, RT_W) * (n > 0 ? n : 1);
  return (int32_t)(tiles < 256 ? tiles : 256);
}

extern "C" int eavsr_rcab_convs_h16(const void* x, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2,
                                    void* out, float* chan_partial, int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(x && w1_packed && w2_packed && out, -1, "rcab_convs_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "rcab_convs_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0, -1, "rcab_convs_h16: bad dims");
  EAVSR_REQUIRE((long)h * w * 128 < (1L << 31), -1, "rcab_convs_h16: one image exceeds 2 GiB");
  EAVSR_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1_packed | (uintptr_t)w2_packed) & 15) == 0, -1,
                "rcab_convs_h16: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  RcabArgs a;
  a.x = x; a.wp1 = w1_packed; a.wp2 = w2_packed; a.bias1 = bias1; a.bias2 = bias2; a.out = out; a.chan_partial = chan_partial;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, RT_W);
  a.tiles_y = eavsr::cdiv(h, RT_H);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "rcab_convs_h16: too many tiles");
  a.num_tiles = (int)tiles;
  const int blocks = eavsr_rcab_h16_partial_rows(n, h, w);
  a.tiles_base = (int)(tiles / blocks);
  a.tiles_rem = (int)(tiles % blocks);
  return dtype == 2 ? launch_rcab_h16<true>(a, blocks, eavsr::as_stream(stream)) : launch_rcab_h16<false>(a, blocks, eavsr::as_stream(stream));
}
