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
          v = fmaxf(v, v * (a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope));   // branch-free: max(v, v s), 0 <= s <= 1
          const size_t o = ((size_t)bn * COUT + co) * plane + (size_t)gy * w + gx;
          if (a.residual) v += a.residual[o];
          a.out[o] = v;
        }
      }
    }
  }
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
