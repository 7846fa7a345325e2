// 3x3 convolution with a handful of output channels (2, 3, 4 or 6) on the vector ALUs
// (a3 heads 64 -> 4+2, a4 TransOffsetworelu 18 -> 2; 8f: conv_last 64 -> 3 at 4x resolution).
//
// Reference: nn.Conv2d at models/networks.py:330-331 (transform_matrix_conv + translation_conv),
// :568 (TransOffsetworelu), models/eavsrp_model.py:156 (conv_last).
//
// These layers have 2-6 output channels: an MFMA tile (32 output channels) would idle > 80 % of the matrix
// pipe and the kernel is bound by streaming the input anyway (64 input channels per output pixel).  Here a
// workgroup stages an 8-channel slab of the (8 + 2) x (64 + 2) input patch in LDS (coalesced loads), every
// thread owns 2 vertically adjacent pixels and keeps COUT x 2 accumulators; the slab's weights sit in LDS too
// and are read back as 16-byte broadcasts, so the inner loop is one ds_read_b32 per input value (4 rows x 3
// columns shared by the 2 pixels), one or two ds_read_b128 per tap and COUT FMAs per pixel and tap.
#include "common.h"

namespace {

constexpr int ST_H = 8, ST_W = 64;              // output tile per workgroup (256 threads, 2 pixels each)
constexpr int SP_H = ST_H + 2, SP_W = ST_W + 2; // patch
constexpr int SCK = 8;                          // channels per LDS slab

struct SmallArgs {
  const float* x;
  const float* wt;    // (cout, cin, 3, 3), original layout
  const float* bias;
  const float* residual;
  float* out;
  int n, cin, h, w, tiles_x, tiles_y, act;
  float slope;
};

// KZ = 1: 256 threads.  KZ = 2: 512 threads, thread group kz = tid >> 8 sums the channels [4 kz, 4 kz + 4) of every slab for the
// same pixels and the groups meet in LDS at the end -- for launches of less than two workgroups per CU (the predictor's heads on a
// 2-clip sub-batch: 230 workgroups), where a workgroup is one wave per SIMD and every LDS round trip of the channel loop is exposed.
template <int COUT, int KZ>
__global__ __launch_bounds__(256 * KZ) void conv3x3_smallco_kernel(SmallArgs a) {
  constexpr int NT = 256 * KZ;
  __shared__ float s_in[SCK][SP_H][SP_W + 1];
  constexpr int CP = COUT <= 4 ? 4 : 8;                      // weights of one (channel, tap): CP floats, 16-byte rows
  __shared__ __attribute__((aligned(16))) float s_wt[SCK][9][CP];
  const int tid = threadIdx.x;
  const int kz = tid >> 8;
  const int lx = tid & 63, ly = (tid >> 6) & 3;  // thread -> column lx, rows 2*ly and 2*ly + 1
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int y0 = ty * ST_H, x0 = tx * ST_W;
  const int h = a.h, w = a.w, cin = a.cin;
  const size_t plane = (size_t)h * w;

  float acc[COUT][2];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co][0] = acc[co][1] = 0.f;

  // The slabs are software-pipelined through registers: the global loads of slab c0 + 8 are issued before the FMAs of slab c0
  // and written to LDS after them, so a slab's HBM round trip hides behind ~1,100 vector instructions.  (A 2 x 180 x 320 launch
  // is 230 workgroups -- less than one per CU, nothing else to hide it: load -> wait -> compute per slab cost 42 us of which
  // ~25 were exposed latency.)
  constexpr int IN_N = SCK * SP_H * SP_W, IN_IT = (IN_N + NT - 1) / NT;
  constexpr int W_N = SCK * 9 * CP, W_IT = (W_N + NT - 1) / NT;
  float tin[IN_IT], tw[W_IT];
  // Slab-invariant index arithmetic once per thread (it was ~70 instructions per element and slab, more than the FMAs): the
  // clamped source offset and the LDS slot of each of this thread's patch elements, a validity bit per element (zero padding),
  // and the same for the weights.
  unsigned in_off[IN_IT], in_ok = 0u;
  unsigned short in_slot[IN_IT];
  unsigned short in_ci[IN_IT];
#pragma unroll
  for (int i = 0; i < IN_IT; ++i) {
    const int e = tid + i * NT;
    const int ci = min(e / (SP_H * SP_W), SCK - 1), rem = e - (e / (SP_H * SP_W)) * (SP_H * SP_W);
    const int r = rem / SP_W, c = rem - r * SP_W;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    if (e < IN_N && gy >= 0 && gy < h && gx >= 0 && gx < w) in_ok |= 1u << i;
    const int cgy = min(max(gy, 0), h - 1), cgx = min(max(gx, 0), w - 1);
    in_off[i] = (unsigned)(cgy * w + cgx);
    in_slot[i] = (unsigned short)((ci * SP_H + r) * (SP_W + 1) + c);
    in_ci[i] = (unsigned short)ci;
  }
  unsigned w_off[W_IT], w_ok = 0u;
  unsigned short w_ci[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int e = min(tid + i * NT, W_N - 1);
    const int co = e % CP, tap = (e / CP) % 9, ci = e / (CP * 9);
    if (tid + i * NT < W_N && co < COUT) w_ok |= 1u << i;
    w_off[i] = (unsigned)((min(co, COUT - 1) * cin + ci) * 9 + tap);
    w_ci[i] = (unsigned short)ci;
  }
  auto fetch = [&](int c0) __attribute__((always_inline)) {
    // straight-line, always-valid addresses, raw values: every load is in flight before the first is used (the zero-padding
    // select happens in commit(); a select here would wait for the load at once)
    const float* xb = a.x + ((size_t)bn * cin + c0) * plane;
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) tin[i] = xb[(size_t)min((int)in_ci[i], cin - c0 - 1) * plane + in_off[i]];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) tw[i] = a.wt[w_off[i] + (unsigned)(min(c0, cin - 1 - (int)w_ci[i]) * 9)];
  };
  auto commit = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IN_IT; ++i)
      if (i < IN_IT - 1 || tid + i * NT < IN_N)
        (&s_in[0][0][0])[in_slot[i]] = ((in_ok >> i) & 1u) && c0 + (int)in_ci[i] < cin ? tin[i] : 0.f;
#pragma unroll
    for (int i = 0; i < W_IT; ++i)
      if (i < W_IT - 1 || tid + i * NT < W_N)
        (&s_wt[0][0][0])[tid + i * NT] = ((w_ok >> i) & 1u) && c0 + (int)w_ci[i] < cin ? tw[i] : 0.f;
  };
  fetch(0);
  for (int c0 = 0; c0 < cin; c0 += SCK) {
    __syncthreads();  // the previous slab has been consumed
    commit(c0);
    __syncthreads();
    if (c0 + SCK < cin) fetch(c0 + SCK);
    const int nci = min(SCK, cin - c0);
    for (int ci = kz * (SCK / KZ); ci < min(nci, (kz + 1) * (SCK / KZ)); ++ci) {   // wave-uniform trip count
      // 4 input rows x 3 columns around the two pixels (rows 2ly .. 2ly+3 of the patch)
      float v[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = s_in[ci][2 * ly + r][lx + c];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float wv[CP];
#pragma unroll
          for (int q = 0; q < CP / 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&s_wt[ci][ky * 3 + kx][4 * q]);
            wv[4 * q] = t4[0]; wv[4 * q + 1] = t4[1]; wv[4 * q + 2] = t4[2]; wv[4 * q + 3] = t4[3];
          }
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            acc[co][0] += wv[co] * v[ky][kx];
            acc[co][1] += wv[co] * v[ky + 1][kx];
          }
        }
    }
  }
  if (KZ > 1) {
    // the channel groups meet in LDS (the last slab's image is dead): one group at a time parks its partial sums, group 0 adds
    float* s_acc = &s_in[0][0][0];
    static_assert(256 * 2 * 6 <= SCK * SP_H * (SP_W + 1), "partial sums must fit the slab image");
    for (int k = 1; k < KZ; ++k) {
      __syncthreads();
      if (kz == k) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          s_acc[(co * 2 + 0) * 256 + (tid & 255)] = acc[co][0];
          s_acc[(co * 2 + 1) * 256 + (tid & 255)] = acc[co][1];
        }
      }
      __syncthreads();
      if (kz == 0) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          acc[co][0] += s_acc[(co * 2 + 0) * 256 + tid];
          acc[co][1] += s_acc[(co * 2 + 1) * 256 + tid];
        }
      }
    }
    if (kz > 0) return;
  }
  const int gx = x0 + lx;
  if (gx < w) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gy = y0 + 2 * ly + p;
      if (gy < h) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          float v = acc[co][p] + (a.bias ? a.bias[co] : 0.f);
          v = eavsr_act(v, a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope);   // branch-free: max(v, v s), 0 <= s <= 1
          const size_t o = ((size_t)bn * COUT + co) * plane + (size_t)gy * w + gx;
          if (a.residual) v += a.residual[o];
          a.out[o] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same convolution shaped to run BESIDE a resident Winograd workgroup of the other stream (round 3).
//
// The what-if timings of the two-stream step (tools/gpu_whatif.py) put the small-cout convolutions at -9.2 ms of 240 when they
// are removed: with 112-184 registers per lane they cannot share a CU with a conv_wino6 workgroup (which leaves 48 registers per
// lane and 28 KB of LDS), so every launch waits for convolution-free CUs and then displaces the other stream's convolutions.  This
// variant is built for the leftover: 256 threads (one wave per SIMD), 23 KB of LDS, and as few vector registers as the compiler
// can be talked into:
//   * the input slabs go global -> LDS by LDS-DMA (no staging registers): 4 channels x 10 rows x 72 columns per stage, the patch
//     starting 4 columns left of the tile so that a row is eighteen aligned 16-byte units (lanes outside the image copy zeros),
//     two stages, the next slab requested right behind the barrier that retires its stage;
//   * the weights never touch a vector register or LDS: packed [ci][kx][ky][co] (eavsr_pack_smallco_weight), they are wave-uniform
//     scalar loads and enter the FMAs as scalar operands;
//   * the 3 x 3 window of a channel is read column by column (4 values live instead of 12).
// Same products as conv3x3_smallco_kernel, channels in order; a channel's nine taps are summed column by column (the other kernel
// goes row by row), so the two agree to rounding, not bit for bit.  Which one runs depends on the image (w % 4, alignment) and
// on EAVSR_SMALLCO, never on the batch.
typedef const __attribute__((address_space(1))) void* sl_gptr_t;
typedef __attribute__((address_space(3))) void* sl_lptr_t;
__device__ __attribute__((aligned(16))) float g_smallco_zero[4];

constexpr int SL_C = 4;                                            // channels per stage
__host__ __device__ constexpr int sl_colblock(int cout) { return cout <= 2 ? 8 : cout <= 4 ? 16 : 32; }   // floats per (ci, kx)

// TH x TW output pixels per workgroup, PXR vertically adjacent pixels per thread: 8 x 64 x 2 (256 threads) for launches of many
// workgroups, 4 x 32 x 1 (128 threads, four times the workgroups) for the pyramid-level heads, whose 230 tiles of 8 x 64 would be
// less than one workgroup per CU with nothing to hide a slab's round trip behind.
template <int COUT, int TH, int TW, int PXR>
__global__ __launch_bounds__((TH / PXR) * TW) void conv3x3_smallco_lite_kernel(SmallArgs a) {
  constexpr int NT = (TH / PXR) * TW, NWV = NT / 64;
  constexpr int PW = TW + 8, PH = TH + 2;                          // patch: 4 columns left of the tile (16-byte units), 1 row above
  constexpr int STAGE = SL_C * PH * PW;                            // floats
  constexpr int SEGS = (STAGE / 4 + 63) / 64;                      // one-KiB pieces
  constexpr int PAD = SEGS * 256;
  constexpr int P_IT = (SEGS + NWV - 1) / NWV;                     // pieces per wave
  constexpr int CB = sl_colblock(COUT);
  static_assert(TW % 4 == 0 && NT % 64 == 0 && (TW == 32 || TW == 64), "tile shape");
  __shared__ __attribute__((aligned(16))) float s_in[2][PAD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned bid = (unsigned)eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned q1 = bid / (unsigned)a.tiles_x, tx = bid - q1 * (unsigned)a.tiles_x;
  const unsigned bn = q1 / (unsigned)a.tiles_y, ty = q1 - bn * (unsigned)a.tiles_y;
  const int y0 = (int)ty * TH, x0 = (int)tx * TW;
  const int h = a.h, w = a.w, cin = a.cin;
  const unsigned plane4 = (unsigned)(h * w) * 4u;                   // bytes per channel plane (the launcher checks the range)

  // this wave's pieces of a stage: 16-byte unit e4 -> (channel, row, unit of the row): byte offset inside the 4-channel block,
  // 0xFFFFFFFF = zero padding (rows / columns outside the image, the unused tail of the last piece)
  unsigned voff[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int e4 = (i * NWV + wave) * 64 + lane;
    const int ci = e4 / (PH * (PW / 4));
    const int rem = e4 - ci * (PH * (PW / 4));
    const int r = rem / (PW / 4), c4 = rem - r * (PW / 4);
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * c4;
    const bool ok = e4 < STAGE / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;      // w % 4 == 0: a unit is inside or outside as a whole
    voff[i] = ok ? (unsigned)ci * plane4 + (unsigned)(gy * w + gx) * 4u : 0xFFFFFFFFu;
  }
  const char* const zero_src = reinterpret_cast<const char*>(g_smallco_zero);
  const char* const xb = reinterpret_cast<const char*>(a.x) + (size_t)bn * cin * plane4;      // wave-uniform
  const unsigned total4 = (unsigned)cin * plane4;
  auto issue = [&](int c0, int stage) __attribute__((always_inline)) {
    const unsigned cbase = (unsigned)c0 * plane4;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      if (i * NWV + wave < SEGS) {     // wave-uniform
        const unsigned off = cbase + voff[i];
        const bool ok = voff[i] != 0xFFFFFFFFu && off < total4;      // channels past cin (the last stage of 18 -> 2) read zeros
        const char* src = ok ? xb + off : zero_src;
        __builtin_amdgcn_global_load_lds((sl_gptr_t)src, (sl_lptr_t)(&s_in[stage][(i * NWV + wave) * 256]), 16, 0, 0);
      }
    }
  };

  float acc[COUT][PXR];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < PXR; ++p) acc[co][p] = 0.f;
  const int lx = tid % TW, ly = tid / TW;
  // packed [ci][kx][CB] ([ky][co] in the first 3 COUT floats), read through the CONSTANT address space: a wave-uniform address
  // there is a scalar load (through the global pointer the compiler must assume the kernel's own stores may alias the weights and
  // uses vector loads: 13 x 16 bytes per lane and channel, and the registers to hold them)
  typedef const __attribute__((address_space(4))) float* sl_cptr_t;
  const sl_cptr_t wpk = (sl_cptr_t)(uintptr_t)a.wt;

  issue(0, 0);
  for (int c0 = 0, stage = 0; c0 < cin; c0 += SL_C, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // this stage's patch is complete; every wave is done with the other stage
    if (c0 + SL_C < cin) issue(c0 + SL_C, stage ^ 1);
    const int nci = min(SL_C, cin - c0);
    for (int ci = 0; ci < nci; ++ci) {
      const float* pin = &s_in[stage][(ci * PH + PXR * ly) * PW + lx + 3];
      const sl_cptr_t wc = wpk + (size_t)(c0 + ci) * 3 * CB;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float v[PXR + 2];
#pragma unroll
        for (int r = 0; r < PXR + 2; ++r) v[r] = pin[r * PW + kx];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float wv = wc[kx * CB + ky * COUT + co];            // wave-uniform: a scalar load, a scalar operand
#pragma unroll
            for (int p = 0; p < PXR; ++p) acc[co][p] += wv * v[ky + p];
          }
      }
    }
  }
  const int gx = x0 + lx;
  if (gx < w) {
    const float act_s = a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope;
    // every bias and every residual value FIRST, in one batch (uniform branches on the two nullable pointers): loaded where
    // they are used each sat behind its own branch and a vmcnt(0) -- up to 12 dependent round trips at the end of a workgroup
    float bv[COUT], rv[PXR][COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) bv[co] = 0.f;
    if (a.bias) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) bv[co] = a.bias[co];
    }
#pragma unroll
    for (int p = 0; p < PXR; ++p)
#pragma unroll
      for (int co = 0; co < COUT; ++co) rv[p][co] = 0.f;
    if (a.residual) {
#pragma unroll
      for (int p = 0; p < PXR; ++p) {
        const int gy = min(y0 + PXR * ly + p, h - 1);      // (clamped: the value of a row below the image is not used)
#pragma unroll
        for (int co = 0; co < COUT; ++co) rv[p][co] = a.residual[((size_t)bn * COUT + co) * (plane4 / 4) + (size_t)gy * w + gx];
      }
    }
#pragma unroll
    for (int p = 0; p < PXR; ++p) {
      const int gy = y0 + PXR * ly + p;
      if (gy < h) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          float v = acc[co][p] + bv[co];
          v = eavsr_act(v, act_s);
          a.out[((size_t)bn * COUT + co) * (plane4 / 4) + (size_t)gy * w + gx] = v + rv[p][co];
        }
      }
    }
  }
}

// (cout, cin, 3, 3) -> [ci][kx][CB]: element ky * cout + co of a column block is w[co][ci][ky][kx]; the rest zero
__global__ void pack_smallco_kernel(const float* __restrict__ wt, float* __restrict__ out, int cout, int cin, int cb) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= cin * 3 * cb) return;
  const int j = e % cb, kx = (e / cb) % 3, ci = e / (3 * cb);
  float v = 0.f;
  if (j < 3 * cout) {
    const int ky = j / cout, co = j - ky * cout;
    v = wt[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
  }
  out[e] = v;
}

}  // namespace

extern "C" int eavsr_conv3x3_smallco_f32(const float* x, const float* weight, const float* bias,
                                         const float* residual, float* out, int32_t n, int32_t cin, int32_t h,
                                         int32_t w, int32_t cout, int32_t act, float slope, void* stream) {
  EAVSR_REQUIRE(x && weight && out, -1, "conv3x3_smallco: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0, -1, "conv3x3_smallco: bad dims");
  EAVSR_REQUIRE(cout == 2 || cout == 3 || cout == 4 || cout == 6, -2,
                "conv3x3_smallco: %d output channels unsupported (2, 3, 4, 6); use eavsr_conv2d_f32", cout);
  EAVSR_REQUIRE(act >= 0 && act <= 2, -1, "conv3x3_smallco: act %d", act);
  EAVSR_REQUIRE(act != EAVSR_ACT_LRELU || (slope >= 0.f && slope <= 1.f), -2,
                "conv3x3_smallco: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)slope);
  if (n == 0) return 0;
  SmallArgs a;
  a.x = x; a.wt = weight; a.bias = bias; a.residual = residual; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.act = act; a.slope = slope;
  a.tiles_x = eavsr::cdiv(w, ST_W);
  a.tiles_y = eavsr::cdiv(h, ST_H);
  const long blocks = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_smallco: too many tiles");
  hipStream_t st = eavsr::as_stream(stream);
  dim3 grid((unsigned)blocks);
  // few workgroups per CU: split the channel loop over two thread groups (one more wave per SIMD to hide LDS latency).  The two
  // groups sum the channels in another order than the single group, so the choice depends on the IMAGE size only, never on the
  // batch: a clip's output bits must not depend on how many clips share its launch (256 tiles per sample = two sub-batch clips
  // of 180 x 320 stay below two workgroups per CU)
  const bool split = (long)a.tiles_x * a.tiles_y < 256 && cin >= 16;
#define EAVSR_SMALLCO(CO)                                                                                  \
  if (split) hipLaunchKernelGGL((conv3x3_smallco_kernel<CO, 2>), grid, dim3(512), 0, st, a);               \
  else hipLaunchKernelGGL((conv3x3_smallco_kernel<CO, 1>), grid, dim3(256), 0, st, a)
  switch (cout) {
    case 2: EAVSR_SMALLCO(2); break;
    case 3: EAVSR_SMALLCO(3); break;
    case 4: EAVSR_SMALLCO(4); break;
    default: EAVSR_SMALLCO(6); break;
  }
#undef EAVSR_SMALLCO
  return eavsr::launch_status("conv3x3_smallco");
}

extern "C" int64_t eavsr_smallco_packed_elems(int32_t cout, int32_t cin) {
  if (!(cout == 2 || cout == 3 || cout == 4 || cout == 6) || cin < 1) return 0;      // the kernel instances
  return (int64_t)cin * 3 * sl_colblock(cout);
}

extern "C" int eavsr_pack_smallco_weight(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_smallco_weight: NULL pointer");
  // only the output counts the kernels are instantiated for: the packing kernel reads weight[co] for co < cout (cout
  // 1 / 5 were packed as 2 / 6 and read one output channel past the weight)
  EAVSR_REQUIRE((cout == 2 || cout == 3 || cout == 4 || cout == 6) && cin >= 1, -2, "pack_smallco_weight: cout %d (2, 3, 4 or 6), cin %d",
                cout, cin);
  const int kc = cout;
  const int cb = sl_colblock(kc);
  const int total = cin * 3 * cb;
  // the column blocks are laid out for the kernel's output count kc: ky * kc + co
  hipLaunchKernelGGL(pack_smallco_kernel, dim3((total + 255) / 256), dim3(256), 0, eavsr::as_stream(stream), weight, packed,
                     kc, cin, cb);
  return eavsr::launch_status("pack_smallco_weight");
}

// The co-resident variant (see conv3x3_smallco_lite_kernel): `weight_packed` from eavsr_pack_smallco_weight; cout in {2, 3, 4, 6};
// w % 4 == 0, x 16-byte aligned, h * w * cin * 4 < 2^32.  Other shapes: eavsr_conv3x3_smallco_f32.
extern "C" int eavsr_conv3x3_smallco_lite_f32(const float* x, const float* weight_packed, const float* bias, const float* residual,
                                              float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t act,
                                              float slope, void* stream) {
  EAVSR_REQUIRE(x && weight_packed && out, -1, "conv3x3_smallco_lite: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0, -1, "conv3x3_smallco_lite: bad dims");
  EAVSR_REQUIRE(cout == 2 || cout == 3 || cout == 4 || cout == 6, -2, "conv3x3_smallco_lite: cout %d (2, 3, 4 or 6)", cout);
  EAVSR_REQUIRE(act >= 0 && act <= 2, -1, "conv3x3_smallco_lite: act %d", act);
  EAVSR_REQUIRE(act != EAVSR_ACT_LRELU || (slope >= 0.f && slope <= 1.f), -2, "conv3x3_smallco_lite: leaky-ReLU slope %g outside [0, 1]",
                (double)slope);
  EAVSR_REQUIRE(w % 4 == 0 && (((uintptr_t)x) & 15) == 0, -2, "conv3x3_smallco_lite: w %% 4 == 0 and a 16-byte aligned input");
  EAVSR_REQUIRE((long)h * w * 4 * cin < (1L << 32), -2, "conv3x3_smallco_lite: sample too large for 32-bit byte offsets");
  if (n == 0) return 0;
  SmallArgs a;
  a.x = x; a.wt = weight_packed; a.bias = bias; a.residual = residual; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.act = act; a.slope = slope;
  // 8 x 64 tiles while they give every CU several workgroups; 4 x 32 tiles (a quarter of the pixels, 128 threads) below that:
  // the choice depends on the image size only (a clip's bits never depend on the batch -- and here not on the tile either: a
  // pixel's sum is the same sequence of operations in both shapes)
  const bool small_tiles = (long)eavsr::cdiv(w, ST_W) * eavsr::cdiv(h, ST_H) < 1024;
#ifndef EAVSR_SL_TH
#define EAVSR_SL_TH 4
#define EAVSR_SL_TW 32
#endif
  const int th = small_tiles ? EAVSR_SL_TH : ST_H, tw = small_tiles ? EAVSR_SL_TW : ST_W;
  a.tiles_x = eavsr::cdiv(w, tw);
  a.tiles_y = eavsr::cdiv(h, th);
  const long blocks = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_smallco_lite: too many tiles");
  hipStream_t st = eavsr::as_stream(stream);
  dim3 grid((unsigned)blocks);
#define EAVSR_SMALLCO_LITE(CO)                                                                                         \
  if (small_tiles) hipLaunchKernelGGL((conv3x3_smallco_lite_kernel<CO, EAVSR_SL_TH, EAVSR_SL_TW, 1>), grid, dim3(EAVSR_SL_TH * EAVSR_SL_TW), 0, st, a); \
  else hipLaunchKernelGGL((conv3x3_smallco_lite_kernel<CO, ST_H, ST_W, 2>), grid, dim3(256), 0, st, a)
  switch (cout) {
    case 2: EAVSR_SMALLCO_LITE(2); break;
    case 3: EAVSR_SMALLCO_LITE(3); break;
    case 4: EAVSR_SMALLCO_LITE(4); break;
    default: EAVSR_SMALLCO_LITE(6); break;
  }
#undef EAVSR_SMALLCO_LITE
  return eavsr::launch_status("conv3x3_smallco_lite");
}
