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
//   ds_read_b128 from the NHWC tile.  k-step s = (tap, 16-channel block): 36 steps x 2 M-tiles x 2 pixel rows = 144 MFMAs
//   per wave and 8 x 32-pixel tile (four waves per tile, see the schedule at the kernel).
//   LDS image of the input patch: [row][col][8 x 16-byte blocks], block index XOR-swizzled with
//   ((col >> 1) & 7) so that the 16 lanes of a ds_read_b128 group (consecutive pixels, 128-byte stride) hit
//   16 different bank quads.  The patch arrives by 16-byte LDS-DMA, whose LDS side is lane-linear: the
//   swizzle is applied to the per-lane SOURCE address (slots outside the image are zero-filled by ds_write).
//   Two patch stages, one per wave group.
//   Epilogue: fp32 accumulators (+ bias, ReLU, optional per-tile channel sums in fp32) are rounded to
//   16 bits and stored from the registers (16 bytes = 8 consecutive channels per lane and store, after the two lanes of a
//   pixel traded 4-channel runs by v_permlane32_swap).
#include "common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <mutex>
#include <type_traits>

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
  void* out;          // (n, h, w, 64) 16-bit; ps: (n, 2 h, 2 w, 64)
  float* chan_partial;
  int n, h, w, tiles_x, tiles_y, num_tiles;
  int act;            // 0 none, 1 ReLU, 2 LeakyReLU(slope)
  float slope;
  int sum_rows;       // 0: chan_partial holds one row per tile; > 0: one row per (workgroup, wave group) and sample (2 x gridDim.x rows per
                      // sample), the sums of that group's tiles of the sample added up in the kernel -- large images, where the ONE
                      // workgroup of ca_scale would otherwise read thousands of per-tile rows
  // RCABlock's tail inside the second convolution's epilogue (round 5): out = res_x + res_scale[n][co] * (conv + bias), rounded once.
  // The attention res_scale is known BEFORE this launch: the channel means of this convolution's output are a linear function of
  // border-corrected channel sums of its INPUT (eavsr_ca_scale_pre_h16, csrc/ca.hip).  NULL: the plain epilogues.
  const void* res_x;         // (n, h, w, 64) 16-bit: the block's input (the skip path)
  const float* res_scale;    // (n, 64) fp32
  float* border;      // NULL, or [n][4][border_stride][64] fp32: sums of the OUTPUT's border lines per border tile (round 6; with chan_partial):
  int border_stride;  //   0 / 1 = image row 0 / h - 1, one piece per tile column; 2 / 3 = image column 0 / w - 1, one piece per (tile row, wave)
  int ps;             // 1: conv 64 -> 256 + PixelShuffle(2) as four 64 -> 64 slices (blockIdx.y = 2 dy + dx): slice k holds the output
                      // channels 4 c + k of the reference weight as its channel c, and its pixel (y, x) is output pixel (2 y + dy, 2 x + dx)
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

// v + (v of the lane the DPP control selects; 0 where there is none) as ONE v_add_f32_dpp (left to the compiler the adds are
// SLP-packed into v_pk_add_f32 and every DPP move becomes its own instruction)
#define H16_ADD_DPP(v, ctrl) asm volatile("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const f32x4& a, const f32x4& b, const f32x16& c) {
  if (BF16) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
}


// Ping-pong schedule (round 2): the workgroup's eight waves are two GROUPS of four (waves 0-3, 4-7 = one wave of each group on
// every SIMD).  A group owns one patch stage and walks its own tiles (the workgroup's tiles alternate between the groups); a
// wave owns two pixel rows of its group's tile (the A operands are shared by both rows: 4 operand reads per 4 MFMAs instead of
// 3 per 2).  The groups run half a tile apart: while one group is in its COMPUTE phase (operand reads + 144 MFMAs per wave) the
// other is in its MEMORY phase (epilogue of its previous tile: bias, activation, rounding, staging, stores, channel sums -- then
// the zero fills and LDS-DMA pieces of its next patch), one workgroup barrier per phase.  In the one-role-at-a-time version all
// eight waves left the matrix pipe idle for 56 % of a tile (in-kernel stamps, tools/gpu_h16_ablate.py).
// A memory phase first requests the group's next patch (its stage was consumed in the phase before), then runs the epilogue under
// that latency, straight from the accumulators (8-byte stores, channel sums by DPP adds): staging the outputs through the patch
// stage put the whole patch round trip (2+ us) at the END of every phase and the schedule was no faster than the old one.
// RESK: the instantiation with the RCAB tail in the epilogue (a.res_x / a.res_scale) -- a kernel of its own so that its registers
// do not weigh on the plain one (as one kernel with a run-time switch the patch offsets went to scratch and every patch request
// waited for the previous one: s_waitcnt vmcnt(0) in front of each)
// BORD: the instantiation that also writes the border pieces (a.border; the RCAB's first convolution) -- a kernel of its own, as RESK
template <bool BF16, bool RESK, bool BORD = false>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_h16_kernel(H16Args a) {
  static_assert(!(RESK && BORD), "border pieces are written by the plain (sums) epilogue");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_w = smem;                                  // resident weights
  unsigned char* s_p = smem + HW_BYTES;                       // two patch stages, one per group
  float* s_red = reinterpret_cast<float*>(smem + HW_BYTES + 2 * HP_SEGS * 1024);   // [group][wave][64] channel sums
  // the bias once per launch through LDS (the 512 never-written bytes behind patch stage 0: 43 one-KiB pieces hold 42.5 KiB)
  float* s_bias = reinterpret_cast<float*>(s_p + HP_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int l31 = lane & 31, half = lane >> 5;
  const int h = a.h, w = a.w;
  unsigned char* pst = s_p + grp * (HP_SEGS * 1024);

  const int slice = blockIdx.y;                     // 0 unless ps
  const int oS = a.ps ? 2 : 1, o_dy = a.ps ? slice >> 1 : 0, o_dx = a.ps ? slice & 1 : 0;
  if (tid < 64) s_bias[tid] = a.bias ? a.bias[slice * 64 + tid] : 0.f;
#ifndef EAVSR_H16_EXP_NO_WDMA
#pragma unroll 1
  for (int seg = wave; seg < HW_SEGS; seg += 8)
    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a.wp) + (size_t)slice * HW_BYTES + seg * 1024 + lane * 16),
                                     (lptr_t)(s_w + seg * 1024), 16, 0, 0);
#endif

#ifdef EAVSR_H16_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  const int cnt = ((int)a.num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
  const int cnt_g = cnt > grp ? (cnt - grp + 1) >> 1 : 0;                                      // ... of this group
  const int cnt_o = cnt > (grp ^ 1) ? (cnt - (grp ^ 1) + 1) >> 1 : 0;
  const int p_end = max(grp + 2 * cnt_g, (grp ^ 1) + 2 * cnt_o) + 1;
  auto tile_at = [&](int j, int& bn, int& ty, int& tx) __attribute__((always_inline)) {
    int t = (int)blockIdx.x + (grp + 2 * j) * (int)gridDim.x;
    tx = t % a.tiles_x;
    t /= a.tiles_x;
    ty = t % a.tiles_y;
    bn = t / a.tiles_y;
  };

  // per-lane source offset of this wave's 11 patch pieces relative to the patch origin (tile-invariant): interior tiles need one
  // add per piece instead of the division / bounds / swizzle arithmetic
  // (RESK recomputes them per tile instead: its sixteen skip-tensor registers are live while the patch is requested, and with the
  //  offsets resident one of them went to scratch -- a scratch reload + vmcnt(0) in front of the last patch piece)
  auto rel_off_of = [&](int i, int lane_) __attribute__((always_inline)) {
    const int e = (i * 4 + w4) * 64 + lane_;
    const int pp = e >> 3, sb = e & 7;
    const int r = pp / HP_W, c = pp - r * HP_W;
    return (unsigned)((r * w + c) * 128 + ((sb ^ ((c >> 1) & 7)) << 4));
  };
  unsigned rel_off[11];
  if constexpr (!RESK && !BORD) {      // (BORD recomputes them per tile as RESK does: with them resident the border code's registers spill)
#pragma unroll
    for (int i = 0; i < 11; ++i) rel_off[i] = rel_off_of(i, lane);
  }
  const bool tail_lane = ((10 * 4 + w4) * 64 + lane) >> 3 < HP_PIX;   // the last piece is partial

  f32x16 acc[2][2];   // [row][m]
  // RESK: the sample's 64 attention values go to LDS at the start of a tile's compute phase (the group's channel-sum slots, unused
  // in this mode).  The sixteen eight-byte pieces of the skip tensor that this lane's tile adds to are requested at the very START
  // of the memory phase, BEFORE the next tile's patch: a wave's requests return in order, so behind the patch (whose round trip
  // is the 5 us the phase lasts) they stalled the epilogue for that long; in front of it their ~2 us hide in the phase's slack.
  // (Holding them across the MFMAs instead spilled the patch offsets to scratch, with a vmcnt(0) in front of every patch request.)
  unsigned xq[2][2][4][2];
  float run = 0.f;     // sum_rows mode, wave 0 of a group: this group's channel sums of sample run_bn so far (lane = channel)
  int run_bn = -1;
  auto flush_rows = [&](int upto_bn) __attribute__((always_inline)) {      // rows of samples run_bn (the sums) .. upto_bn - 1 (zeros)
    const size_t row0 = (size_t)blockIdx.x * 2 + grp;
    if (run_bn >= 0) a.chan_partial[((size_t)run_bn * a.sum_rows + row0) * 64 + lane] = run;
    for (int sb = run_bn + 1; sb < upto_bn; ++sb) a.chan_partial[((size_t)sb * a.sum_rows + row0) * 64 + lane] = 0.f;
  };
  for (int p = 0; p <= p_end; ++p) {
    const int q = p - grp;
    const int j = q >> 1;
    // A memory phase issues its patch requests FIRST and its 8 output stores behind them: the counter is in order, so
    // `vmcnt(8)` at the end of the phase waits for the patch and leaves the stores in flight (their acknowledgement from HBM
    // is ~2 us that the phase used to end with).  Only when all 8 store instructions are known to have been issued: every
    // pixel of the wave's two rows inside the image (wave-uniform).  The compute phase that follows ends with vmcnt(0).
    bool stores_behind_dma = false;
    if (q >= 0 && (q & 1) == 0) {
      // ================================================================== memory phase =======================================
      if constexpr (RESK) {
        if (j >= 1 && j - 1 < cnt_g) {
          int bn, ty, tx;
          tile_at(j - 1, bn, ty, tx);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int gy = ty * HT_H + 2 * w4 + r, gx = tx * HT_W + l31;
            const bool ok = gy < h && gx < w;
            const char* xrow = reinterpret_cast<const char*>(a.res_x) + (size_t)bn * h * w * 128 + (4 * half) * 2 +
                               ((size_t)(ok ? gy : 0) * w + (ok ? gx : 0)) * 128;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t xv = *reinterpret_cast<const u32x2_t*>(xrow + (m * 32 + 8 * qd) * 2);
                xq[r][m][qd][0] = xv[0];
                xq[r][m][qd][1] = xv[1];
              }
          }
          __builtin_amdgcn_sched_barrier(0);      // (the requests stay in front of the patch's)
        }
      }
      if (j < cnt_g) {
        // ---- the patch of tile j FIRST (the stage was consumed in the last phase; its latency hides under the epilogue below):
        // slot e -> pixel e >> 3, stored block e & 7 holds the logical block (e & 7) ^ swz(col); slots outside the image are
        // zero-filled.  43 one-KiB pieces over the four waves of the group.
        int bn, ty, tx;
        tile_at(j, bn, ty, tx);
        const int y0 = ty * HT_H - 1, x0 = tx * HT_W - 1;
        const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)bn * h * w * 128;
        if (y0 >= 0 && x0 >= 0 && y0 + HP_H <= h && x0 + HP_W <= w) {      // interior tile (wave-uniform): no zero fills
          const char* org = xb + ((size_t)y0 * w + x0) * 128;
          int lane_o = lane;      // RESK: an opaque copy per tile, so that the offsets below are not hoisted back out of the tile loop
          if constexpr (RESK || BORD) asm volatile("" : "+v"(lane_o));
#pragma unroll
          for (int i = 0; i < 11; ++i) {
            const int seg = i * 4 + w4;
#ifndef EAVSR_H16_EXP_NO_DMA
            if (seg < HP_SEGS && (i < 10 || tail_lane))
              __builtin_amdgcn_global_load_lds((gptr_t)(org + ((RESK || BORD) ? rel_off_of(i, lane_o) : rel_off[i])), (lptr_t)(pst + seg * 1024), 16, 0, 0);
#endif
          }
        } else {
#pragma unroll 1
          for (int i = 0; i < 11; ++i) {
            const int seg = i * 4 + w4;
            if (seg < HP_SEGS) {
              const int e = seg * 64 + lane;
              const int pp = e >> 3, sb = e & 7;
              const int r = pp / HP_W, c = pp - r * HP_W;
              const int gy = y0 + r, gx = x0 + c;
              if (pp < HP_PIX) {
#ifndef EAVSR_H16_EXP_NO_DMA
                if (gy >= 0 && gy < h && gx >= 0 && gx < w)
                  __builtin_amdgcn_global_load_lds((gptr_t)(xb + ((size_t)gy * w + gx) * 128 + ((sb ^ ((c >> 1) & 7)) << 4)),
                                                   (lptr_t)(pst + seg * 1024), 16, 0, 0);
                else
#endif
                  *reinterpret_cast<f32x4*>(pst + e * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
              }
            }
          }
        }
        H16_STAMP(2);     // zero fills + patch DMA issue
      }
      if (j >= 1 && j - 1 < cnt_g) {
        // ---- epilogue of tile j - 1 straight from the accumulators: lane (pixel l31, half) holds 4 consecutive channels per
        // (m, qd) = one 8-byte store; the two halves write adjacent 8 bytes, the eight (m, qd) stores of a pixel fill its
        // 128-byte row (merged in L2).  No LDS staging: the patch stage is already being refilled.
        int bn, ty, tx;
        tile_at(j - 1, bn, ty, tx);
#ifndef EAVSR_H16_EXP_WAIT_STORES
        stores_behind_dma = ty * HT_H + 2 * w4 + 1 < h && tx * HT_W + 31 < w;
#endif
        const int ow = w * oS;      // output row length in pixels
        char* ob = reinterpret_cast<char*>(a.out) + (size_t)bn * (h * oS) * ow * 128;
        const float slope = a.slope;
        f32x4 bq4[2][4];     // this lane's 32 bias values: 4 consecutive channels per (m, qd)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) bq4[m][qd] = *reinterpret_cast<const f32x4*>(s_bias + m * 32 + 8 * qd + 4 * half);
        // the activation and the channel sums are launch constants: four straight-line variants (the sums alone are ~450 of the
        // ~900 vector instructions of a general epilogue, as many issue cycles as the tile's MFMAs)
        auto epilogue = [&](auto sums_c, auto act_c, auto res_c) __attribute__((always_inline)) {
          constexpr bool SUMS = decltype(sums_c)::value;
          constexpr int ACT = decltype(act_c)::value;
          constexpr bool RES = decltype(res_c)::value;
          f32x4 sq4[2][4];
          if (RES) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) sq4[m][qd] = *reinterpret_cast<const f32x4*>(s_red + grp * 256 + m * 32 + 8 * qd + 4 * half);
          }
          float csum[2][16];
          if (SUMS) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int e = 0; e < 16; ++e) csum[m][e] = 0.f;
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int gy = ty * HT_H + 2 * w4 + r, gx = tx * HT_W + l31;
            const bool ok = gy < h && gx < w;
            char* orow = ob + ((size_t)(gy * oS + o_dy) * ow + (gx * oS + o_dx)) * 128;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int qe = 0; qe < 4; qe += 2) {
                // registers 4 qd .. 4 qd + 3 are 4 consecutive output channels co = m*32 + 8 qd + 4 half + (0..3): 8 bytes.  The
                // lane pair (l31, half 0 | 1) trades runs by v_permlane32_swap -- half 0 keeps its run of qd = qe and takes its
                // partner's, half 1 those of qd = qe + 1 -- so that a lane stores 16 contiguous bytes (channels 8 qd .. 8 qd + 7)
                // and a store instruction writes 32 contiguous bytes per pixel: half as many store instructions, each touching
                // the same 32 lines (the 8-byte form cost ~400 cycles of the memory pipe per instruction)
                unsigned dw[2][2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                  const int qd = qe + q2;
                  unsigned short pk[4];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    float v = acc[r][m][4 * qd + e] + bq4[m][qd][e];
                    if (RES) {
                      const unsigned short xs_ = (unsigned short)(xq[r][m][qd][e >> 1] >> (16 * (e & 1)));
                      v = fmaf(v, sq4[m][qd][e], from_h16<BF16>(xs_));
                    }
                    if (ACT == 1) v = fmaxf(v, 0.f);
                    if (ACT == 2) v = fmaxf(v, v * slope);
                    pk[e] = to_h16<BF16>(v);
                    if (SUMS) csum[m][4 * qd + e] += ok ? from_h16<BF16>(pk[e]) : 0.f;   // the 16-bit value the next layer reads
                  }
                  dw[q2][0] = (unsigned)pk[0] | ((unsigned)pk[1] << 16);
                  dw[q2][1] = (unsigned)pk[2] | ((unsigned)pk[3] << 16);
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(dw[0][0], dw[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(dw[0][1], dw[1][1], false, false);
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v4 = {s0[0], s1[0], s0[1], s1[1]};
#ifdef EAVSR_H16_EXP_NO_STORE
                if (v4[0] == 0x12344321u && v4[1] == 0x43211234u)
#endif
                if (ok) *reinterpret_cast<u32x4_t*>(orow + (m * 32 + 8 * (qe + half)) * 2) = v4;
              }
          }
          H16_STAMP(5);     // bias / activation / rounding / output stores
          if (SUMS) {
            // round 6: the sums of the output's border lines as a by-product (a.border; for eavsr_ca_scale_pre_pieces, which had a
            // launch of its own for them on every RCAB's dependent chain).  Column 0 / w - 1 of the image: before the scan below a
            // lane's csum is the sum over ITS pixel column of the wave's two rows -- the lane on the border column stores its 32
            // values as the piece of (tile row, wave).  Rows: behind the scan (below).
            if constexpr (BORD) {
              const int cnt_last = (w - 1) - tx * HT_W;      // lane of image column w - 1 in this tile (>= 32: not in this tile)
              float* bp = a.border + (size_t)bn * 4 * a.border_stride * 64;
              if (tx == 0 && l31 == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                  for (int qd = 0; qd < 4; ++qd)
                    *reinterpret_cast<f32x4*>(bp + ((size_t)2 * a.border_stride + ty * 4 + w4) * 64 + m * 32 + 8 * qd + 4 * half) =
                        f32x4{csum[m][4 * qd], csum[m][4 * qd + 1], csum[m][4 * qd + 2], csum[m][4 * qd + 3]};
              }
              if (cnt_last >= 0 && cnt_last < HT_W && l31 == cnt_last) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                  for (int qd = 0; qd < 4; ++qd)
                    *reinterpret_cast<f32x4*>(bp + ((size_t)3 * a.border_stride + ty * 4 + w4) * 64 + m * 32 + 8 * qd + 4 * half) =
                        f32x4{csum[m][4 * qd], csum[m][4 * qd + 1], csum[m][4 * qd + 2], csum[m][4 * qd + 3]};
              }
            }
            // channel sums over this wave's 64 pixels: 32 lanes hold the same channels -> five DPP adds per value (an inclusive
            // scan inside each row of 16 lanes, then row_bcast:15), totals in lanes 31 and 63; the four waves meet in LDS and
            // wave 0 of the group adds them up after the phase barrier (fixed order)
            // step-major order (volatile asm keeps it): 31 independent instructions between two DPP operations on the same
            // register, which also covers the VALU-write -> DPP-read wait states the assembler does not insert for inline asm
#define H16_DPP_STEP(ctrl)                                     \
  _Pragma("unroll") for (int m = 0; m < 2; ++m)                \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) H16_ADD_DPP(csum[m][e], ctrl);
            asm volatile("s_nop 1");   // the last csum add may sit right in front of the first DPP read of its register
            H16_DPP_STEP("row_shr:1")
            H16_DPP_STEP("row_shr:2")
            H16_DPP_STEP("row_shr:4")
            H16_DPP_STEP("row_shr:8")
            H16_DPP_STEP("row_bcast:15")
#undef H16_DPP_STEP
            if (l31 == 31) {
#pragma unroll
              for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                  s_red[(grp * 4 + w4) * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = csum[m][e];
            }
            // row 0 / h - 1 of the image, where this wave holds it (at most two waves of a border-row tile, wave-uniform): the row's
            // 16-bit values once more from the accumulators (still live) into the registers the scan has just freed, the same scan,
            // the totals of lanes 31 / 63 as the piece of this tile column.  No register, no LDS word beside what the sums use.
            if constexpr (BORD) {
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const int gy = ty * HT_H + 2 * w4 + r;
                const bool is_top = gy == 0, is_bot = gy == h - 1;      // (wave-uniform)
                if (is_top || is_bot) {
                  const bool okx = tx * HT_W + l31 < w;
#pragma unroll
                  for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                      for (int e = 0; e < 4; ++e) {
                        float v = acc[r][m][4 * qd + e] + bq4[m][qd][e];
                        if (ACT == 1) v = fmaxf(v, 0.f);
                        if (ACT == 2) v = fmaxf(v, v * slope);
                        csum[m][4 * qd + e] = okx ? from_h16<BF16>(to_h16<BF16>(v)) : 0.f;
                      }
#define H16_DPP_STEP2(ctrl)                                    \
  _Pragma("unroll") for (int m = 0; m < 2; ++m)                \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) H16_ADD_DPP(csum[m][e], ctrl);
                  asm volatile("s_nop 1");
                  H16_DPP_STEP2("row_shr:1")
                  H16_DPP_STEP2("row_shr:2")
                  H16_DPP_STEP2("row_shr:4")
                  H16_DPP_STEP2("row_shr:8")
                  H16_DPP_STEP2("row_bcast:15")
#undef H16_DPP_STEP2
                  if (l31 == 31) {
                    float* bp = a.border + (size_t)bn * 4 * a.border_stride * 64;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                      for (int qd = 0; qd < 4; ++qd) {
                        const f32x4 v4 = f32x4{csum[m][4 * qd], csum[m][4 * qd + 1], csum[m][4 * qd + 2], csum[m][4 * qd + 3]};
                        if (is_top) *reinterpret_cast<f32x4*>(bp + ((size_t)0 * a.border_stride + tx) * 64 + m * 32 + 8 * qd + 4 * half) = v4;
                        if (is_bot) *reinterpret_cast<f32x4*>(bp + ((size_t)1 * a.border_stride + tx) * 64 + m * 32 + 8 * qd + 4 * half) = v4;
                      }
                  }
                }
              }
            }
          }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        using A0 = std::integral_constant<int, 0>;
        using A1 = std::integral_constant<int, 1>;
        using A2 = std::integral_constant<int, 2>;
        if constexpr (RESK) epilogue(F_{}, A0{}, T_{});
        else if (a.chan_partial) { if (a.act == 1) epilogue(T_{}, A1{}, F_{}); else epilogue(T_{}, A0{}, F_{}); }      // (the sums: no LeakyReLU form)
        else { if (a.act == 1) epilogue(F_{}, A1{}, F_{}); else if (a.act == 2) epilogue(F_{}, A2{}, F_{}); else epilogue(F_{}, A0{}, F_{}); }
        H16_STAMP(4);     // channel sums
      }
    } else if (q >= 1) {
      // ================================================================== compute phase ======================================
      if (a.chan_partial && j >= 1 && j - 1 < cnt_g && w4 == 0) {   // finish the channel sums of the tile stored in the last phase
        int bn, ty, tx;
        tile_at(j - 1, bn, ty, tx);
        float v = s_red[(grp * 4) * 64 + lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) v += s_red[(grp * 4 + k) * 64 + lane];
        if (a.sum_rows == 0) {
          a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * 64 + lane] = v;
        } else {      // (a group's tiles come in increasing order: the sample index never decreases)
          if (bn != run_bn) {
            flush_rows(bn);
            run = 0.f;
            run_bn = bn;
          }
          run += v;
        }
      }
      if (j < cnt_g) {
        if constexpr (RESK) {
          int bn, ty, tx;
          tile_at(j, bn, ty, tx);
          if (w4 == 0) s_red[grp * 256 + lane] = a.res_scale[(size_t)bn * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][m][e] = 0.f;
        // The 18 half taps (filter row ky, i = kx * 2 + pair of 16-channel blocks) as ONE software pipeline with hand-placed
        // operand reads: the eight `ds_read_b128` of half tap s + 1 (order per 16-channel block: pixels row 0, weights m = 0,
        // pixels row 1, weights m = 1) go one behind every MFMA of half tap s, and every MFMA that needs a new operand waits with
        // `lgkmcnt(6)` -- the counter is in order, so "at most six younger reads outstanding" is exactly "mine has landed".  The
        // reads are inline assembly because the compiler's own waits for them were `lgkmcnt(0)` behind a block of eight fresh
        // reads (an exposed LDS round trip every other half tap; only one wave of a SIMD computes at a time, nobody fills it:
        // 51 cycles per MFMA instead of 32 by the stamps).  No LDS-DMA of THIS wave is in flight here.
        f32x4 bq[2][4], aq[2][4];
        const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
        const unsigned lds_w = lds0 + (unsigned)(((half * 64) + l31) << 4), lds_w2 = lds_w + 2 * 24576;
        unsigned ba[3][4];      // pixel-operand address of this lane for (kx, block cb), row 0 of the wave, filter row 0
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            ba[kx][cb] = lds0 + (unsigned)(HW_BYTES + grp * (HP_SEGS * 1024) + (((2 * w4) * HP_W + l31 + kx) << 7) +
                                           (((cb * 2 + half) ^ (((l31 + kx) >> 1) & 7)) << 4));
#define H16_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        // read K (0..7) of half tap (KY, I) into operand set SET
#define H16_LOAD1(KY, I, SET, K)                                                                                          \
  do {                                                                                                                    \
    constexpr int kx_ = (I) >> 1, c2_ = (K) >> 2, j_ = (K) & 3, cb_ = ((I) & 1) * 2 + c2_;                                \
    if constexpr ((j_ & 1) == 0) H16_DSR(bq[SET][c2_ * 2 + (j_ >> 1)], ba[kx_][cb_], (((j_ >> 1) + (KY)) * HP_W) << 7);   \
    else if constexpr ((KY) < 2) H16_DSR(aq[SET][c2_ * 2 + (j_ >> 1)], lds_w,                                            \
                                         (KY) * 24576 + (((kx_ * 4 + cb_) * 2 * 64 + (j_ >> 1) * 32) << 4));              \
    else H16_DSR(aq[SET][c2_ * 2 + (j_ >> 1)], lds_w2, ((kx_ * 4 + cb_) * 2 * 64 + (j_ >> 1) * 32) << 4);                 \
  } while (0)
        // MFMA K of half tap (KY, I) on set SET (+ its wait), then read K of the next half tap (NKY, NI) into the other set
#ifdef EAVSR_H16_EXP_NO_MFMA
#define H16_MFMA(r_, m_, A_, B_) acc[r_][m_][0] += (A_)[0] + (B_)[1]
#else
#define H16_MFMA(r_, m_, A_, B_) acc[r_][m_] = mfma16<BF16>(A_, B_, acc[r_][m_])
#endif
#define H16_SLOT(SET, K, HASNEXT, NKY, NI)                                                                                \
  do {                                                                                                                    \
    constexpr int c2_ = (K) >> 2, m_ = ((K) >> 1) & 1, r_ = (K) & 1;                                                      \
    constexpr int need_ = ((K) & 3) == 3 ? -1 : (K) + 1;      /* index of the youngest read this MFMA needs */            \
    if constexpr (need_ >= 0)                                                                                             \
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(aq[SET][c2_ * 2 + m_]), "+v"(bq[SET][c2_ * 2 + r_])                    \
                   : "n"(7 - need_ + ((HASNEXT) ? (K) : 0)));                                                             \
    H16_MFMA(r_, m_, aq[SET][c2_ * 2 + m_], bq[SET][c2_ * 2 + r_]);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if constexpr (HASNEXT) {                                                                                              \
      H16_LOAD1(NKY, NI, (SET) ^ 1, K);                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
    }                                                                                                                     \
  } while (0)
#define H16_STEP(SET, HASNEXT, NKY, NI)                                                        \
  H16_SLOT(SET, 0, HASNEXT, NKY, NI); H16_SLOT(SET, 1, HASNEXT, NKY, NI); H16_SLOT(SET, 2, HASNEXT, NKY, NI); \
  H16_SLOT(SET, 3, HASNEXT, NKY, NI); H16_SLOT(SET, 4, HASNEXT, NKY, NI); H16_SLOT(SET, 5, HASNEXT, NKY, NI); \
  H16_SLOT(SET, 6, HASNEXT, NKY, NI); H16_SLOT(SET, 7, HASNEXT, NKY, NI)
        __builtin_amdgcn_sched_barrier(0);
        H16_LOAD1(0, 0, 0, 0); H16_LOAD1(0, 0, 0, 1); H16_LOAD1(0, 0, 0, 2); H16_LOAD1(0, 0, 0, 3);
        H16_LOAD1(0, 0, 0, 4); H16_LOAD1(0, 0, 0, 5); H16_LOAD1(0, 0, 0, 6); H16_LOAD1(0, 0, 0, 7);
        __builtin_amdgcn_sched_barrier(0);
        H16_STEP(0, true, 0, 1); H16_STEP(1, true, 0, 2); H16_STEP(0, true, 0, 3); H16_STEP(1, true, 0, 4); H16_STEP(0, true, 0, 5);
        H16_STEP(1, true, 1, 0);
        H16_STEP(0, true, 1, 1); H16_STEP(1, true, 1, 2); H16_STEP(0, true, 1, 3); H16_STEP(1, true, 1, 4); H16_STEP(0, true, 1, 5);
        H16_STEP(1, true, 2, 0);
        H16_STEP(0, true, 2, 1); H16_STEP(1, true, 2, 2); H16_STEP(0, true, 2, 3); H16_STEP(1, true, 2, 4); H16_STEP(0, true, 2, 5);
        H16_STEP(1, false, 0, 0);
#undef H16_STEP
#undef H16_SLOT
#undef H16_MFMA
#undef H16_LOAD1
#undef H16_DSR
        __builtin_amdgcn_sched_barrier(0);
        H16_STAMP(1);     // operand reads + MFMAs
      }
    }
    // this wave's DMA pieces, zero fills and sums have landed (raw barrier: __syncthreads() would drain vmcnt again)
    if (stores_behind_dma) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    H16_STAMP(0);         // waiting for the DMA and the barrier
  }
  if (a.chan_partial && a.sum_rows > 0 && w4 == 0) flush_rows(a.n);      // the last sample's sums; zero rows for the samples this group never saw
#ifdef EAVSR_H16_STAMPS
  if (tid == 0 || tid == 256)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_h16_stamps[i], st_acc[i]);
#endif
}


// ---------------------------------------------------------------------------------------------
// conv_last of the upsampling tail in the 16-bit modes (eavsrp_model.py:359-360): 3x3, 64 -> 3, 16-bit NHWC in, fp32 NCHW out
// (+ the bilinear skip image).  5,184 FLOP per pixel against 128 bytes read: a streaming kernel.  One thread per output pixel,
// the 10 x 34-pixel patch of an 8 x 32 tile in LDS (the XOR swizzle of the MFMA kernel's patch: 16 neighbouring pixels of a
// ds_read_b128 hit 16 different bank quads), the 1,728 weights -- wave-uniform -- through SCALAR loads from their 16-bit
// [tap][8-channel block][co][8] image (as LDS broadcasts they were 3/4 of the kernel's LDS traffic and the kernel was slower
// than the fp32 one), the contraction by v_dot2c_f32_{bf16,f16} (two channels per instruction, fp32 accumulate).
// ---------------------------------------------------------------------------------------------
struct L16Args {
  const void* x;          // (n, h, w, 64) 16-bit
  const void* weight;     // 16-bit [9 taps][8 blocks][3 co][8 channels]: the (3, 64, 3, 3) parameter permuted and rounded
  const float* bias;      // [3] or NULL
  const float* residual;  // (n, 3, h, w) fp32 or NULL
  float* out;             // (n, 3, h, w) fp32
  int n, h, w, tiles_x, tiles_y;
};

template <bool BF16>
__device__ __forceinline__ float dot2_h16(unsigned a, unsigned b, float c) {
  if (BF16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), c, false);
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void conv3x3_c64to3_h16_kernel(L16Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_patch[HP_BYTES];      // [row][col][8 x 16-byte blocks], blocks swizzled
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, bn = t / a.tiles_y;
  const int h = a.h, w = a.w;
  const int y0 = ty * HT_H - 1, x0 = tx * HT_W - 1;
  const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)bn * h * w * 128;
  for (int e = tid; e < HP_PIX * 8; e += 256) {
    const int pp = e >> 3, sb = e & 7;
    const int r = pp / HP_W, c = pp - r * HP_W;
    const int gy = y0 + r, gx = x0 + c;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = *reinterpret_cast<const u32x4_t*>(xb + ((size_t)gy * w + gx) * 128 + sb * 16);
    *reinterpret_cast<u32x4_t*>(s_patch + pp * 128 + ((sb ^ ((c >> 1) & 7)) << 4)) = v;
  }
  __syncthreads();
  const int r = tid >> 5, c = tid & 31;
  float acc[3] = {a.bias ? a.bias[0] : 0.f, a.bias ? a.bias[1] : 0.f, a.bias ? a.bias[2] : 0.f};
  const unsigned* __restrict__ wq = reinterpret_cast<const unsigned*>(a.weight);      // uniform addresses: scalar loads
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned char* px = s_patch + ((r + ky) * HP_W + c + kx) * 128;
      const int swz = ((c + kx) >> 1) & 7;
      const unsigned* wt = wq + (ky * 3 + kx) * (8 * 3 * 4);
#pragma unroll
      for (int sb = 0; sb < 8; ++sb) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(px + ((sb ^ swz) << 4));
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[co] = dot2_h16<BF16>(v[q], wt[(sb * 3 + co) * 4 + q], acc[co]);
      }
    }
  const int gy = ty * HT_H + r, gx = tx * HT_W + c;
  if (gy < h && gx < w) {
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const size_t o = ((size_t)bn * 3 + co) * h * w + (size_t)gy * w + gx;
      a.out[o] = acc[co] + (a.residual ? a.residual[o] : 0.f);
    }
  }
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

template <bool BF16, bool RESK = false, bool BORD = false>
int launch_conv_h16(const H16Args& a, int blocks, int slices, hipStream_t st) {
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_h16_kernel<BF16, RESK, BORD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_c64_h16: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((conv3x3_c64_h16_kernel<BF16, RESK, BORD>), dim3(blocks, slices), dim3(512), H_LDS_BYTES, st, a);
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

// rows per sample of eavsr_conv3x3_c64_h16's chan_partial: one per tile, or -- where a sample has more tiles than twice the
// persistent workgroups -- one per (workgroup, wave group): the kernel adds up each group's tiles of a sample itself
extern "C" int32_t eavsr_conv_h16_partial_rows(int32_t n, int32_t h, int32_t w) {
  const long per = (long)eavsr_conv_h16_tiles(h, w), total = per * (n > 0 ? n : 1);
  const long blocks = total < 256 ? total : 256;
  return (int32_t)(per > 2 * blocks ? 2 * blocks : per);
}

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

static int conv3x3_c64_h16_launch(const void* x, const void* weight_packed, const float* bias, void* out, float* chan_partial,
                                  int32_t n, int32_t h, int32_t w, int32_t act, float slope, int32_t ps, int32_t dtype, void* stream,
                                  const void* res_x = nullptr, const float* res_scale = nullptr, float* border = nullptr,
                                  int32_t border_stride = 0);

extern "C" int eavsr_conv3x3_c64_h16_res(const void* x, const void* weight_packed, const float* bias, void* out, const void* res_x,
                                         const float* res_scale, int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(res_x && res_scale, -1, "conv3x3_c64_h16_res: NULL pointer");
  EAVSR_REQUIRE((((uintptr_t)res_x | (uintptr_t)res_scale) & 15) == 0, -1, "conv3x3_c64_h16_res: res_x / res_scale must be 16-byte aligned");
  return conv3x3_c64_h16_launch(x, weight_packed, bias, out, nullptr, n, h, w, 0, 0.f, 0, dtype, stream, res_x, res_scale);
}

extern "C" int eavsr_conv3x3_c64_h16(const void* x, const void* weight_packed, const float* bias, void* out,
                                     float* chan_partial, int32_t n, int32_t h, int32_t w, int32_t relu,
                                     int32_t dtype, void* stream) {
  return conv3x3_c64_h16_launch(x, weight_packed, bias, out, chan_partial, n, h, w, relu ? 1 : 0, 0.f, 0, dtype, stream);
}

// eavsr_conv3x3_c64_h16 that also leaves the sums of its OUTPUT's border lines per border tile (H16Args::border): p_rows = tiles_x
// pieces per row border, p_cols = 4 tiles_y per column border (eavsr_conv_h16_border_pieces), layout [n][4][border_stride][64] fp32
extern "C" int eavsr_conv_h16_border_pieces(int32_t h, int32_t w, int32_t* p_rows, int32_t* p_cols) {
  EAVSR_REQUIRE(h > 0 && w > 0 && p_rows && p_cols, -1, "conv_h16_border_pieces: bad arguments");
  *p_rows = eavsr::cdiv(w, HT_W);
  *p_cols = 4 * eavsr::cdiv(h, HT_H);
  return 0;
}
extern "C" int eavsr_conv3x3_c64_h16_b(const void* x, const void* weight_packed, const float* bias, void* out, float* chan_partial,
                                       float* border_pieces, int32_t border_stride, int32_t n, int32_t h, int32_t w, int32_t relu,
                                       int32_t dtype, void* stream) {
  EAVSR_REQUIRE(border_pieces != nullptr, -1, "conv3x3_c64_h16_b: NULL border_pieces");
  return conv3x3_c64_h16_launch(x, weight_packed, bias, out, chan_partial, n, h, w, relu ? 1 : 0, 0.f, 0, dtype, stream, nullptr, nullptr,
                                border_pieces, border_stride);
}

extern "C" int eavsr_conv3x3_c64_h16_act(const void* x, const void* weight_packed, const float* bias, void* out, int32_t n, int32_t h,
                                         int32_t w, int32_t act, float slope, int32_t pixel_shuffle2, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(act == EAVSR_ACT_NONE || act == EAVSR_ACT_RELU || act == EAVSR_ACT_LRELU, -1, "conv3x3_c64_h16_act: act %d", act);
  EAVSR_REQUIRE(pixel_shuffle2 == 0 || pixel_shuffle2 == 1, -1, "conv3x3_c64_h16_act: pixel_shuffle2 %d", pixel_shuffle2);
  return conv3x3_c64_h16_launch(x, weight_packed, bias, out, nullptr, n, h, w, act == EAVSR_ACT_RELU ? 1 : act == EAVSR_ACT_LRELU ? 2 : 0,
                                slope, pixel_shuffle2, dtype, stream);
}

static int conv3x3_c64_h16_launch(const void* x, const void* weight_packed, const float* bias, void* out, float* chan_partial,
                                  int32_t n, int32_t h, int32_t w, int32_t act, float slope, int32_t ps, int32_t dtype, void* stream,
                                  const void* res_x, const float* res_scale, float* border, int32_t border_stride) {
  EAVSR_REQUIRE(x && weight_packed && out, -1, "conv3x3_c64_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv3x3_c64_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0, -1, "conv3x3_c64_h16: bad dims");
  EAVSR_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight_packed) & 15) == 0, -1,
                "conv3x3_c64_h16: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  H16Args a;
  a.x = x; a.wp = weight_packed; a.bias = bias; a.out = out; a.chan_partial = chan_partial;
  a.res_x = res_x; a.res_scale = res_scale;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, HT_W);
  a.tiles_y = eavsr::cdiv(h, HT_H);
  a.border = border; a.border_stride = border_stride;
  EAVSR_REQUIRE(border == nullptr || (chan_partial != nullptr && !ps && res_x == nullptr && border_stride >= a.tiles_x &&
                                      border_stride >= 4 * a.tiles_y && (((uintptr_t)border) & 15) == 0), -1,
                "conv3x3_c64_h16: border pieces need chan_partial, a 16-byte aligned buffer and border_stride >= max(tiles_x, 4 tiles_y)");
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "conv3x3_c64_h16: too many tiles");
  a.num_tiles = (int)tiles;
  a.act = act; a.slope = slope; a.ps = ps;
  {
    const int rows = eavsr_conv_h16_partial_rows(n, h, w);
    a.sum_rows = (chan_partial && !ps && rows != a.tiles_x * a.tiles_y) ? rows : 0;
  }
  const int per = ps ? 64 : 256;                      // persistent: one workgroup per CU (four slices: 64 each)
  const int blocks = tiles < per ? (int)tiles : per;
  if (res_x != nullptr)
    return dtype == 2 ? launch_conv_h16<true, true>(a, blocks, 1, eavsr::as_stream(stream))
                      : launch_conv_h16<false, true>(a, blocks, 1, eavsr::as_stream(stream));
  if (border != nullptr)
    return dtype == 2 ? launch_conv_h16<true, false, true>(a, blocks, 1, eavsr::as_stream(stream))
                      : launch_conv_h16<false, false, true>(a, blocks, 1, eavsr::as_stream(stream));
  return dtype == 2 ? launch_conv_h16<true>(a, blocks, ps ? 4 : 1, eavsr::as_stream(stream))
                    : launch_conv_h16<false>(a, blocks, ps ? 4 : 1, eavsr::as_stream(stream));
}

extern "C" int eavsr_conv3x3_c64to3_h16(const void* x, const void* weight, const float* bias, const float* residual, float* out,
                                        int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(x && weight && out, -1, "conv3x3_c64to3_h16: NULL pointer");
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "conv3x3_c64to3_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0, -1, "conv3x3_c64to3_h16: bad dims");
  EAVSR_REQUIRE((((uintptr_t)x | (uintptr_t)weight) & 15) == 0, -1, "conv3x3_c64to3_h16: x and weight must be 16-byte aligned");
  if (n == 0) return 0;
  L16Args a;
  a.x = x; a.weight = weight; a.bias = bias; a.residual = residual; a.out = out;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = eavsr::cdiv(w, HT_W);
  a.tiles_y = eavsr::cdiv(h, HT_H);
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(tiles < (1L << 31), -1, "conv3x3_c64to3_h16: too many tiles");
  if (dtype == 2)
    hipLaunchKernelGGL(conv3x3_c64to3_h16_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, eavsr::as_stream(stream), a);
  else
    hipLaunchKernelGGL(conv3x3_c64to3_h16_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, eavsr::as_stream(stream), a);
  return eavsr::launch_status("conv3x3_c64to3_h16");
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
