// Channel attention of the residual channel-attention blocks (a11).
//
// Reference: CALayer.forward models/networks.py:444-447 (AdaptiveAvgPool2d(1) -> 1x1 conv c -> c/16
// -> ReLU -> 1x1 conv -> Sigmoid -> x * y) and RCABlock.forward :461-464 (res * y + x).
// The global mean needs a grid-wide reduction between conv2 and the scale; the conv kernel leaves
// per-tile channel sums (no atomics, fixed order), ca_scale finishes the mean + the 64->4->64 MLP for
// every sample in one tiny launch, scale_residual applies `r * y + x` as a float4 stream.
#include "common.h"

namespace {

// one workgroup per sample; 1024 threads = Q groups of c threads, group q sums tiles q, q+Q, ... of its
// channel (coalesced over channels; four independent partial sums so that the loads of a thread are in flight
// together -- the launch is pure latency: 60..240 tiles of 64 floats per sample), fixed-order LDS reduction over
// the groups, then the tiny MLP.
constexpr int CA_THREADS = 1024;
__global__ __launch_bounds__(CA_THREADS) void ca_scale_kernel(const float* __restrict__ partial, int tiles, float inv_hw,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ scale, float* __restrict__ mean_out, int c, int cr) {
  extern __shared__ float sm[];  // part[Q*c] then mean[c] then hidden[cr]
  const int Q = CA_THREADS / c;
  float* part = sm;
  float* mean = sm + Q * c;
  float* hid = mean + c;
  const int bn = blockIdx.x;
  const int tid = threadIdx.x;
  const int q = tid / c, ch = tid - q * c;
  if (q < Q) {
    const float* p = partial + (size_t)bn * tiles * c + ch;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = q;
    // sixteen loads in flight where the tiles allow it (configs[4]: 1020 tiles per sample, 64 per thread -- four dependent batches of
    // four loads each were 22 us of pure latency per launch); every accumulator still receives its addends in the same order
    for (; t + 15 * Q < tiles; t += 16 * Q) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(t + j * Q) * c];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; t + 3 * Q < tiles; t += 4 * Q) {
      const float v0 = p[(size_t)t * c], v1 = p[(size_t)(t + Q) * c], v2 = p[(size_t)(t + 2 * Q) * c],
                  v3 = p[(size_t)(t + 3 * Q) * c];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; t < tiles; t += Q) s0 += p[(size_t)t * c];
    part[q * c + ch] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < c) {
    float s = 0.f;
    for (int k = 0; k < Q; ++k) s += part[k * c + tid];
    mean[tid] = s * inv_hw;
    if (mean_out) mean_out[(size_t)bn * c + tid] = s * inv_hw;      // the training step's backward wants it (no second reduction)
  }
  __syncthreads();
  if (tid < cr) {
    float v = b1[tid];
    for (int k = 0; k < c; ++k) v += w1[tid * c + k] * mean[k];
    hid[tid] = fmaxf(v, 0.f);
  }
  __syncthreads();
  if (tid < c) {
    float v = b2[tid];
    for (int j = 0; j < cr; ++j) v += w2[tid * cr + j] * hid[j];
    scale[(size_t)bn * c + tid] = 1.f / (1.f + expf(-v));
  }
}

// out = r * scale[n,c] + x ; hw % 4 == 0 fast path with 16-B accesses
template <bool VEC>
__global__ __launch_bounds__(256) void scale_residual_kernel(const float* __restrict__ r,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ x, float* __restrict__ out,
                                                             int hw) {
  const int nc = blockIdx.y;
  const float s = scale[nc];
  const size_t base = (size_t)nc * hw;
  if (VEC) {
    const int hw4 = hw >> 2;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(r + base);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
    f32x4* o4 = reinterpret_cast<f32x4*>(out + base);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw4; i += gridDim.x * 256) {
      const f32x4 a = r4[i], b = x4[i];
      f32x4 o;
      o[0] = a[0] * s + b[0]; o[1] = a[1] * s + b[1]; o[2] = a[2] * s + b[2]; o[3] = a[3] * s + b[3];
      o4[i] = o;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256)
      out[base + i] = r[base + i] * s + x[base + i];
  }
}

// The whole RCAB tail in ONE launch: out = r * sigmoid(MLP(mean_hw(r))) + x, from the conv's per-tile channel sums.
// Round 2's version (1024-thread workgroups, eight 16-byte pieces per thread in registers) was bit-identical and 3 ms less kernel
// time per forward, and made the two-stream step SLOWER (261 -> 274 ms): its workgroups filled every CU and evicted the other
// stream's convolution workgroups.  This is the version shaped to run BESIDE a resident Winograd workgroup (which leaves 48 vector
// registers per lane and 28 KB of LDS on its CU): 256 threads, < 32 registers, 4.6 KB of LDS, one pair of 16-byte loads in flight
// per thread.  Every workgroup finishes the mean + the MLP of its sample itself -- the fixed-order reduction of ca_scale_kernel
// replayed by 256 threads (each takes the role of four of its 1024), so every workgroup of a sample gets the identical scale and
// the result equals ca_scale + scale_residual bit for bit -- and then streams its slice: float4 columns [p0, p0 + len) of EVERY
// channel.  The partial sums are 29 KB per sample, L2-resident, re-read by each workgroup of the sample.
constexpr int CT_THREADS = 256;
__global__ __launch_bounds__(CT_THREADS) void ca_tail_kernel(const float* __restrict__ r, const float* __restrict__ partial,
                                                            int tiles, float inv_hw, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, const float* __restrict__ x,
                                                            float* __restrict__ out, int c, int cr, int hw4, int slice,
                                                            float* __restrict__ scale_out, float* __restrict__ mean_out) {
  extern __shared__ float sm[];  // part[Q*c] then mean[c] then hidden[cr] then scale[c]
  const int Q = CA_THREADS / c;  // the grouping of ca_scale_kernel (1024 logical threads), whatever this kernel's width
  float* part = sm;
  float* mean = sm + Q * c;
  float* hid = mean + c;
  float* s_scale = hid + cr;
  const int bn = blockIdx.y;
  const int tid = threadIdx.x;
  // Training form (scale_out != NULL; round 6): everything the phases below read is requested HERE, in front of the first barrier --
  // the MLP's parameters (64 channels, <= 8 hidden units) and the slice's first four pieces of r and x per thread.  Behind their
  // barriers they were the second, third and fourth round trip of a launch that sits on every RCAB's dependent chain (9.0 us per
  // launch at a 96 x 96 crop); the arithmetic and its order are unchanged.
  const bool pre = scale_out != nullptr && c == 64 && cr <= 8;
  float pw1[64], pw2[8], pb1 = 0.f, pb2 = 0.f;
  f32x4 pa[4], pbx[4];
  size_t po[4];
  int pch[4];
  const int p0_ = blockIdx.x * slice, len_ = min(slice, hw4 - p0_), total_ = len_ * c;
  if (pre) {
    if (tid < cr) {
      pb1 = b1[tid];
#pragma unroll
      for (int k = 0; k < 64; ++k) pw1[k] = w1[tid * 64 + k];
    }
    if (tid < 64) {
      pb2 = b2[tid];
#pragma unroll
      for (int j = 0; j < 8; ++j) pw2[j] = j < cr ? w2[tid * cr + j] : 0.f;
    }
    const f32x4* r4p = reinterpret_cast<const f32x4*>(r) + (size_t)bn * c * hw4 + p0_;
    const f32x4* x4p = reinterpret_cast<const f32x4*>(x) + (size_t)bn * c * hw4 + p0_;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = tid + u * CT_THREADS;
      const bool ok = f < total_;
      pch[u] = ok ? f / len_ : 0;
      po[u] = ok ? (size_t)pch[u] * hw4 + (f - pch[u] * len_) : 0;
      pa[u] = r4p[po[u]];
      pbx[u] = x4p[po[u]];
    }
  }
  // 1. mean + MLP (ca_scale_kernel's arithmetic, one logical thread at a time)
  for (int lt = tid; lt < Q * c; lt += CT_THREADS) {
    const int q = lt / c, ch = lt - q * c;
    const float* p = partial + (size_t)bn * tiles * c + ch;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = q;
    for (; t + 3 * Q < tiles; t += 4 * Q) {
      const float v0 = p[(size_t)t * c], v1 = p[(size_t)(t + Q) * c], v2 = p[(size_t)(t + 2 * Q) * c],
                  v3 = p[(size_t)(t + 3 * Q) * c];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; t < tiles; t += Q) s0 += p[(size_t)t * c];
    part[q * c + ch] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  for (int ch = tid; ch < c; ch += CT_THREADS) {
    float s = 0.f;
    for (int k = 0; k < Q; ++k) s += part[k * c + ch];
    mean[ch] = s * inv_hw;
  }
  __syncthreads();
  if (pre) {
    if (tid < cr) {
      float v = pb1;
#pragma unroll
      for (int k = 0; k < 64; ++k) v += pw1[k] * mean[k];
      hid[tid] = fmaxf(v, 0.f);
    }
    __syncthreads();
    if (tid < 64) {
      float v = pb2;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cr) v += pw2[j] * hid[j];
      s_scale[tid] = 1.f / (1.f + expf(-v));
    }
  } else {
    if (tid < cr) {
      float v = b1[tid];
      for (int k = 0; k < c; ++k) v += w1[tid * c + k] * mean[k];
      hid[tid] = fmaxf(v, 0.f);
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += CT_THREADS) {
      float v = b2[ch];
      for (int j = 0; j < cr; ++j) v += w2[ch * cr + j] * hid[j];
      s_scale[ch] = 1.f / (1.f + expf(-v));
    }
  }
  __syncthreads();
  // (training: the backward of CALayer needs the attention and the channel means -- written once per sample)
  if (blockIdx.x == 0 && scale_out != nullptr)
    for (int ch = tid; ch < c; ch += CT_THREADS) {
      scale_out[(size_t)bn * c + ch] = s_scale[ch];
      if (mean_out != nullptr) mean_out[(size_t)bn * c + ch] = mean[ch];
    }
  // 2. out = r * scale + x over the slice: flat piece f -> channel f / len, column f % len (kept incrementally)
  const int p0 = blockIdx.x * slice, len = min(slice, hw4 - p0);
  const int total = len * c;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(r) + (size_t)bn * c * hw4 + p0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + (size_t)bn * c * hw4 + p0;
  f32x4* o4 = reinterpret_cast<f32x4*>(out) + (size_t)bn * c * hw4 + p0;
  int ch = tid / len, col = tid - ch * len;
  const int dch = CT_THREADS / len, dcol = CT_THREADS - dch * len;
  if (scale_out != nullptr) {
    // the training step's form (one stream, nothing to co-reside with): four pieces per thread in flight instead of one
    int f_first = tid;
    if (pre) {      // the first four pieces arrived while the phases above ran
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tid + u * CT_THREADS < total) {
          const float sv = s_scale[pch[u]];
          f32x4 v;
          v[0] = pa[u][0] * sv + pbx[u][0]; v[1] = pa[u][1] * sv + pbx[u][1]; v[2] = pa[u][2] * sv + pbx[u][2]; v[3] = pa[u][3] * sv + pbx[u][3];
          o4[po[u]] = v;
        }
      f_first = tid + 4 * CT_THREADS;
      ch = f_first / len;      // (ch, col) of piece f_first
      col = f_first - ch * len;
    }
    for (int f = f_first; f < total; f += 4 * CT_THREADS) {
      f32x4 a[4], b[4];
      size_t o[4];
      float sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = f + u * CT_THREADS < total;
        o[u] = ok ? (size_t)ch * hw4 + col : 0;
        sv[u] = ok ? s_scale[ch] : 0.f;
        a[u] = r4[o[u]];
        b[u] = x4[o[u]];
        ch += dch; col += dcol;
        if (col >= len) { col -= len; ++ch; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (f + u * CT_THREADS < total) {
          f32x4 v;
          v[0] = a[u][0] * sv[u] + b[u][0]; v[1] = a[u][1] * sv[u] + b[u][1]; v[2] = a[u][2] * sv[u] + b[u][2]; v[3] = a[u][3] * sv[u] + b[u][3];
          o4[o[u]] = v;
        }
    }
    return;
  }
  for (int f = tid; f < total; f += CT_THREADS) {
    const size_t o = (size_t)ch * hw4 + col;
    const f32x4 a = r4[o], b = x4[o];
    const float s = s_scale[ch];
    f32x4 v;
    v[0] = a[0] * s + b[0]; v[1] = a[1] * s + b[1]; v[2] = a[2] * s + b[2]; v[3] = a[3] * s + b[3];
    o4[o] = v;
    ch += dch; col += dcol;
    if (col >= len) { col -= len; ++ch; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The attention of an RCAB BEFORE its second convolution runs (16-bit modes, round 5).  r = conv2(t) + b2 is linear in t, so the
// channel means of r -- all the attention needs (networks.py:444-447) -- follow from sums of t:
//   sum_o r[co][o] = hw b2[co] + sum_{ci, ky, kx} W2[co][ci][ky][kx] S[ci][ky][kx],
//   S[ci][ky][kx] = sum of t[ci] over the pixels p whose output pixel p - (ky - 1, kx - 1) lies in the image
//                 = T - R(ky) - C(kx) + X(ky, kx)      (zero padding by inclusion / exclusion)
// with T the plane sum (the first convolution's epilogue already leaves it as per-tile channel sums), R(0) / R(2) the sums of the
// last / first image row, C(0) / C(2) of the last / first column, X the corner pixel both exclude (R(1) = C(1) = 0).  The second
// convolution can then apply x + scale * r in its own epilogue (eavsr_conv3x3_c64_h16_res) and scale_residual_h16 -- three
// 128-byte-per-pixel streams per RCAB, 8 % / 13 % of the configs[2] / [4] step -- disappears.  W2 is rounded to the 16-bit type
// here as the convolution's packed weights are; t is the 16-bit tensor the convolution reads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CP_SEGS = 8;      // segments per border line (workgroups of the border-sum launch: 4 x CP_SEGS per sample)

template <bool BF16> __device__ __forceinline__ float cp_from_h16(unsigned short v) {
  if (BF16) return __builtin_bit_cast(float, (unsigned)v << 16);
  return (float)__builtin_bit_cast(_Float16, v);
}
template <bool BF16> __device__ __forceinline__ float cp_round_h16(float v) {
  if (BF16) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const unsigned r = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;      // round to nearest even (finite inputs)
    return __builtin_bit_cast(float, r);
  }
  return (float)(_Float16)v;
}

// element (bn, y, x, ch) of t -- DT 1 / 2: 16-bit NHWC (fp16 / bf16); DT 0: fp32 NCHW (the fp32 path of configs[1] / [3])
template <int DT> __device__ __forceinline__ float cp_load(const void* t, int bn, int y, int x, int ch, int h, int w) {
  if (DT == 0) return reinterpret_cast<const float*>(t)[(((size_t)bn * 64 + ch) * h + y) * w + x];
  return cp_from_h16<DT == 2>(reinterpret_cast<const unsigned short*>(t)[(((size_t)bn * h + y) * w + x) * 64 + ch]);
}

// out[((bn * 4 + b) * CP_SEGS + s) * 64 + ch] = sum over segment s of border line b (0 top row, 1 bottom row, 2 left column,
// 3 right column) of t[bn][..][ch].  256 threads = 64 channels x 4 pixel lanes; a pixel's 64 channels are 128 contiguous bytes.
template <int DT>
__global__ __launch_bounds__(256) void h16_border_sums_kernel(const void* __restrict__ t, float* __restrict__ out, int h, int w) {
  __shared__ float red[4][64];
  const int ch = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int sgm = blockIdx.x, b = blockIdx.y, bn = blockIdx.z;
  const int len = b < 2 ? w : h;
  const int lo = (int)((long)len * sgm / CP_SEGS), hi = (int)((long)len * (sgm + 1) / CP_SEGS);
  auto at = [&](int i) __attribute__((always_inline)) {
    return b == 0 ? cp_load<DT>(t, bn, 0, i, ch, h, w) : b == 1 ? cp_load<DT>(t, bn, h - 1, i, ch, h, w)
         : b == 2 ? cp_load<DT>(t, bn, i, 0, ch, h, w) : cp_load<DT>(t, bn, i, w - 1, ch, h, w);
  };
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = lo + pl;
  for (; i + 12 < hi; i += 16) {
    const float v0 = at(i), v1 = at(i + 4), v2 = at(i + 8), v3 = at(i + 12);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; i < hi; i += 4) s0 += at(i);
  red[pl][ch] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0) out[(((size_t)bn * 4 + b) * CP_SEGS + sgm) * 64 + ch] = (red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch]);
}

// one workgroup of 1024 threads per sample, 64 channels
template <int DT>
__global__ __launch_bounds__(1024) void ca_scale_pre_kernel(const float* __restrict__ partial, int rows, const float* __restrict__ border,
                                                            int p_rows, int p_cols, int p_stride,
                                                            const void* __restrict__ t, int h, int w,
                                                            const float* __restrict__ wc, const float* __restrict__ bc,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            float* __restrict__ scale, int cr) {
  constexpr int C = 64, Q = 1024 / C;
  __shared__ float part[Q * C];
  __shared__ float T[C], B[4][C], Bq[16][C], X[4][C], S[9][C], mean[C], hid[C];
  const int bn = blockIdx.x, tid = threadIdx.x;
  // every parameter this thread will need, requested now (each was a round trip of its own behind a barrier)
  const float bc_v = bc ? bc[tid >> 4] : 0.f;
  const int hj = tid >> 6;                                   // hidden unit of this wave (cr <= 16 of them in one pass)
  const float w1_v = hj < cr ? w1[hj * C + (tid & 63)] : 0.f, b1_v = hj < cr ? b1[hj] : 0.f;
  float w2_v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w2_v[j] = (tid < C && j < cr) ? w2[tid * cr + j] : 0.f;
  const float b2_v = tid < C ? b2[tid] : 0.f;
  {      // the plane sums of t from the first convolution's per-tile (or per-workgroup) channel sums: ca_scale_kernel's reduction
    const int q = tid / C, ch = tid - q * C;
    const float* p = partial + (size_t)bn * rows * C + ch;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = q;
    for (; r + 31 * Q < rows; r += 32 * Q) {      // (512 per-workgroup rows at 540 x 960: one round trip instead of two)
      float v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = p[(size_t)(r + j * Q) * C];
#pragma unroll
      for (int j = 0; j < 32; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; r + 15 * Q < rows; r += 16 * Q) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(r + j * Q) * C];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; r + 3 * Q < rows; r += 4 * Q) {
      const float v0 = p[(size_t)r * C], v1 = p[(size_t)(r + Q) * C], v2 = p[(size_t)(r + 2 * Q) * C], v3 = p[(size_t)(r + 3 * Q) * C];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; r < rows; r += Q) s0 += p[(size_t)r * C];
    part[q * C + ch] = (s0 + s1) + (s2 + s3);
  }
  // border lines: the pieces of border b (0 top row, 1 bottom row: p_rows pieces; 2 left column, 3 right column: p_cols pieces) -- the
  // segments of the border-sum launch, or what the first convolution's epilogue left per border tile (round 6) -- added in a fixed
  // order: a wave takes a border (rows) or a seventh of one (columns), sixteen requests in flight per lane, the waves meet in LDS
  {
    // (border and part are the same for a whole wave: the bounds below are scalar branches, a request that no lane needs is not
    // issued -- sixteen exec-masked requests per wave cost the one memory pipe of the CU 2 us.)  Waves 0 / 1 take the row borders
    // whole, waves 2 .. 8 / 9 .. 15 a seventh of a column border each: at 540 x 960 a column border holds 272 pieces against 30 of
    // a row border, and as four equal quarters per border (rounds 5-6) the columns were five request round trips; three now.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ch = tid & 63;
    const int b = wv < 2 ? wv : 2 + (wv - 2) / 7, part = wv < 2 ? 0 : (wv - 2) % 7, nparts = wv < 2 ? 1 : 7;
    const int cnt = b < 2 ? p_rows : p_cols;
    const float* p = border + (((size_t)bn * 4 + b) * p_stride) * C + ch;
    float s0 = 0.f, s1 = 0.f;
    for (int k = part; k < cnt; k += 16 * nparts) {      // sixteen predicated requests at once
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = 0.f;
        if (k + nparts * j < cnt) v[j] = p[(size_t)(k + nparts * j) * C];
      }
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        s0 += v[j] + v[j + 1];
        s1 += v[j + 2] + v[j + 3];
      }
    }
    Bq[wv][ch] = s0 + s1;
  }
  if (tid < 4 * C) {      // corners: 0 = (0, 0), 1 = (0, w - 1), 2 = (h - 1, 0), 3 = (h - 1, w - 1)
    const int k = tid >> 6, ch = tid & 63;
    X[k][ch] = cp_load<DT>(t, bn, (k >> 1) ? h - 1 : 0, (k & 1) ? w - 1 : 0, ch, h, w);
  }
  // this thread's 36 weights of the contraction below as nine 16-byte requests, issued BEHIND the sums' requests (a wave's requests
  // return in order) and in flight across the next three barriers (as 36 dependent dwords in front of their use they were nine L2
  // round trips of the one workgroup the whole GPU waits for: 11.6 us per launch at 540 x 960)
  f32x4 wq[9];
  {
    const f32x4* wr4 = reinterpret_cast<const f32x4*>(wc + (size_t)(tid >> 4) * 576) + (tid & 15);
#pragma unroll
    for (int i = 0; i < 9; ++i) wq[i] = wr4[16 * i];
  }
  __syncthreads();
  if (tid < C) {
    float v = 0.f;
    for (int k = 0; k < Q; ++k) v += part[k * C + tid];
    T[tid] = v;
  } else if (tid >= 4 * C && tid < 8 * C) {
    const int b = (tid >> 6) - 4, ch = tid & 63;
    // (a border's waves in wave order: deterministic)
    float v;
    if (b < 2) {
      v = Bq[b][ch];
    } else {
      const int w0 = 2 + 7 * (b - 2);
      v = ((Bq[w0][ch] + Bq[w0 + 1][ch]) + (Bq[w0 + 2][ch] + Bq[w0 + 3][ch])) + ((Bq[w0 + 4][ch] + Bq[w0 + 5][ch]) + Bq[w0 + 6][ch]);
    }
    B[b][ch] = v;
  }
  __syncthreads();
  if (tid < 9 * C) {      // S[tap][ci] = T - R(ky) - C(kx) + X(ky, kx)
    const int tap = tid >> 6, ci = tid & 63, ky = tap / 3, kx = tap - 3 * ky;
    const float R = ky == 0 ? B[1][ci] : ky == 2 ? B[0][ci] : 0.f;      // ky = 0 excludes the LAST row, ky = 2 the first
    const float Cc = kx == 0 ? B[3][ci] : kx == 2 ? B[2][ci] : 0.f;
    const float Xc = (ky == 1 || kx == 1) ? 0.f : X[(ky == 0 ? 2 : 0) + (kx == 0 ? 1 : 0)][ci];
    S[tap][ci] = (T[ci] - R) - Cc + Xc;
  }
  __syncthreads();
  {      // mean of r: 64 outputs x 576 products, 16 threads per output (one shuffle tree inside 16 consecutive lanes)
    const int co = tid >> 4, pt = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 4 * (pt + 16 * i) + e;      // k = ci * 9 + tap (the weight's own order)
        const int ci = k / 9, tap = k - 9 * ci;
        v += (DT == 0 ? wq[i][e] : cp_round_h16<DT == 2>(wq[i][e])) * S[tap][ci];
      }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    if (pt == 0) mean[co] = bc_v + v / ((float)h * (float)w);
  }
  __syncthreads();
  // hidden unit j by wave j, wave j + 16, ..: one request per lane and a shuffle tree (four threads walking 64 dependent products each
  // were most of what was left of the launch)
  if (hj < cr) {
    float v = w1_v * mean[tid & 63];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) hid[hj] = fmaxf(v + b1_v, 0.f);
  }
  __syncthreads();
  if (tid < C) {
    float v = b2_v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += j < cr ? w2_v[j] * hid[j] : 0.f;
    scale[(size_t)bn * C + tid] = 1.f / (1.f + expf(-v));
  }
}

}  // namespace

extern "C" int64_t eavsr_ca_scale_pre_ws_floats(int32_t n) { return n > 0 ? (int64_t)n * 4 * CP_SEGS * 64 : 0; }

// pieces == nullptr: the border lines are summed by a launch of h16_border_sums_kernel into `workspace`; otherwise `pieces` holds what
// the first convolution's epilogue left ([n][4][p_stride][64], p_rows / p_cols pieces per row / column border) and there is ONE launch
static int ca_scale_pre_launch(const void* t, const float* chan_partial, int32_t rows, const float* conv_weight, const float* conv_bias,
                               const float* w1, const float* b1, const float* w2, const float* b2, float* scale, float* workspace,
                               int32_t n, int32_t h, int32_t w, int32_t cr, int32_t dtype, void* stream,
                               const float* pieces = nullptr, int32_t p_rows = 0, int32_t p_cols = 0, int32_t p_stride = 0) {
  EAVSR_REQUIRE(t && chan_partial && conv_weight && w1 && b1 && w2 && b2 && scale && (workspace || pieces), -1, "ca_scale_pre: NULL pointer");
  EAVSR_REQUIRE(pieces == nullptr || (p_rows > 0 && p_cols > 0 && p_stride >= p_rows && p_stride >= p_cols), -1,
                "ca_scale_pre: border pieces need 0 < p_rows, p_cols <= p_stride");
  EAVSR_REQUIRE(dtype >= 0 && dtype <= 2, -1, "ca_scale_pre: dtype %d (0 = f32 NCHW, 1 = f16, 2 = bf16 NHWC)", dtype);
  EAVSR_REQUIRE(n >= 0 && h > 0 && w > 0 && rows > 0 && cr > 0 && cr <= 8 && n <= 65535, -1, "ca_scale_pre: bad dims (1..8 hidden units)");
  EAVSR_REQUIRE(((uintptr_t)conv_weight & 15) == 0, -1, "ca_scale_pre: conv_weight must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t st = eavsr::as_stream(stream);
#define CP_LAUNCH(DT_)                                                                                                             \
  do {                                                                                                                             \
    if (pieces == nullptr) {                                                                                                      \
      hipLaunchKernelGGL(h16_border_sums_kernel<DT_>, dim3(CP_SEGS, 4, n), dim3(256), 0, st, t, workspace, h, w);                   \
      hipLaunchKernelGGL(ca_scale_pre_kernel<DT_>, dim3(n), dim3(1024), 0, st, chan_partial, rows, workspace, CP_SEGS, CP_SEGS,     \
                         CP_SEGS, t, h, w, conv_weight, conv_bias, w1, b1, w2, b2, scale, cr);                                     \
    } else {                                                                                                                       \
      hipLaunchKernelGGL(ca_scale_pre_kernel<DT_>, dim3(n), dim3(1024), 0, st, chan_partial, rows, pieces, p_rows, p_cols,          \
                         p_stride, t, h, w, conv_weight, conv_bias, w1, b1, w2, b2, scale, cr);                                    \
    }                                                                                                                              \
  } while (0)
  if (dtype == 2) CP_LAUNCH(2);
  else if (dtype == 1) CP_LAUNCH(1);
  else CP_LAUNCH(0);
#undef CP_LAUNCH
  return eavsr::launch_status("ca_scale_pre");
}

extern "C" int eavsr_ca_scale_pre_h16(const void* t, const float* chan_partial, int32_t rows, const float* conv_weight,
                                      const float* conv_bias, const float* w1, const float* b1, const float* w2, const float* b2,
                                      float* scale, float* workspace, int32_t n, int32_t h, int32_t w, int32_t cr, int32_t dtype,
                                      void* stream) {
  EAVSR_REQUIRE(dtype == 1 || dtype == 2, -1, "ca_scale_pre_h16: dtype %d (1 = f16, 2 = bf16)", dtype);
  return ca_scale_pre_launch(t, chan_partial, rows, conv_weight, conv_bias, w1, b1, w2, b2, scale, workspace, n, h, w, cr, dtype, stream);
}

extern "C" int eavsr_ca_scale_pre_f32(const float* t, const float* chan_partial, int32_t tiles, const float* conv_weight,
                                      const float* conv_bias, const float* w1, const float* b1, const float* w2, const float* b2,
                                      float* scale, float* workspace, int32_t n, int32_t h, int32_t w, int32_t cr, void* stream) {
  return ca_scale_pre_launch(t, chan_partial, tiles, conv_weight, conv_bias, w1, b1, w2, b2, scale, workspace, n, h, w, cr, 0, stream);
}

extern "C" int eavsr_ca_scale_mean_f32(const float* chan_partial, int32_t tiles, int32_t hw, const float* w1,
                                       const float* b1, const float* w2, const float* b2, float* scale, float* mean_out,
                                       int32_t n, int32_t c, int32_t cr, void* stream) {
  EAVSR_REQUIRE(chan_partial && w1 && b1 && w2 && b2 && scale, -1, "ca_scale: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && cr > 0 && tiles > 0 && hw > 0, -1, "ca_scale: bad dims");
  EAVSR_REQUIRE(c <= 256 && cr <= 256, -2, "ca_scale: c=%d / cr=%d unsupported (<= 256)", c, cr);
  if (n == 0) return 0;
  const int Q = CA_THREADS / c;
  hipLaunchKernelGGL(ca_scale_kernel, dim3(n), dim3(CA_THREADS), (size_t)(Q * c + c + cr) * sizeof(float), eavsr::as_stream(stream),
                     chan_partial, tiles, 1.0f / (float)hw, w1, b1, w2, b2, scale, mean_out, c, cr);
  return eavsr::launch_status("ca_scale");
}

extern "C" int eavsr_ca_scale_f32(const float* chan_partial, int32_t tiles, int32_t hw, const float* w1,
                                  const float* b1, const float* w2, const float* b2, float* scale, int32_t n,
                                  int32_t c, int32_t cr, void* stream) {
  return eavsr_ca_scale_mean_f32(chan_partial, tiles, hw, w1, b1, w2, b2, scale, nullptr, n, c, cr, stream);
}

extern "C" int eavsr_scale_residual_f32(const float* r, const float* scale, const float* x, float* out, int32_t n,
                                        int32_t c, int32_t hw, void* stream) {
  EAVSR_REQUIRE(r && scale && x && out, -1, "scale_residual: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hw > 0, -1, "scale_residual: bad dims");
  EAVSR_REQUIRE((long)n * c <= 65535, -1, "scale_residual: n*c too large");
  if (n * c == 0) return 0;
  const bool vec = (hw % 4) == 0 && (((uintptr_t)r | (uintptr_t)x | (uintptr_t)out) & 15) == 0;
  const int work = vec ? hw / 4 : hw;
  int bx = eavsr::cdiv(work, 256);
  if (bx > 64) bx = 64;
  dim3 grid(bx, n * c);
  if (vec)
    hipLaunchKernelGGL(scale_residual_kernel<true>, grid, dim3(256), 0, eavsr::as_stream(stream), r, scale, x, out, hw);
  else
    hipLaunchKernelGGL(scale_residual_kernel<false>, grid, dim3(256), 0, eavsr::as_stream(stream), r, scale, x, out, hw);
  return eavsr::launch_status("scale_residual");
}

extern "C" int eavsr_ca_tail_stats_f32(const float* r, const float* chan_partial, int32_t tiles, const float* w1, const float* b1,
                                       const float* w2, const float* b2, const float* x, float* out, float* scale_out,
                                       float* mean_out, int32_t n, int32_t c, int32_t cr, int32_t hw, void* stream);
extern "C" int eavsr_ca_tail_f32(const float* r, const float* chan_partial, int32_t tiles, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* x, float* out, int32_t n, int32_t c,
                                 int32_t cr, int32_t hw, void* stream) {
  return eavsr_ca_tail_stats_f32(r, chan_partial, tiles, w1, b1, w2, b2, x, out, nullptr, nullptr, n, c, cr, hw, stream);
}

extern "C" int eavsr_ca_tail_stats_f32(const float* r, const float* chan_partial, int32_t tiles, const float* w1, const float* b1,
                                       const float* w2, const float* b2, const float* x, float* out, float* scale_out,
                                       float* mean_out, int32_t n, int32_t c, int32_t cr, int32_t hw, void* stream) {
  EAVSR_REQUIRE(r && chan_partial && w1 && b1 && w2 && b2 && x && out, -1, "ca_tail: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && cr > 0 && tiles > 0 && hw > 0 && n <= 65535, -1, "ca_tail: bad dims");
  EAVSR_REQUIRE(c <= 256 && cr <= 256, -2, "ca_tail: c=%d / cr=%d unsupported (<= 256)", c, cr);
  EAVSR_REQUIRE(hw % 4 == 0 && ((((uintptr_t)r) | ((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0, -2,
                "ca_tail: hw %% 4 == 0 and 16-byte aligned tensors (otherwise eavsr_ca_scale_f32 + eavsr_scale_residual_f32)");
  if (n == 0) return 0;
  const int hw4 = hw / 4;
  // ~256 workgroups per sample (each re-reduces the sample's per-tile sums: 29 KB of L2 reads at c = 64, 115 tiles), at
  // least 32 float4 columns per workgroup
  int slice = eavsr::cdiv(hw4, 256);
  if (slice < 32) slice = 32;
  const int bx = eavsr::cdiv(hw4, slice);
  const int Q = CA_THREADS / c;
  hipLaunchKernelGGL(ca_tail_kernel, dim3(bx, n), dim3(CT_THREADS), (size_t)(Q * c + c + cr + c) * sizeof(float),
                     eavsr::as_stream(stream), r, chan_partial, tiles, 1.0f / (float)hw, w1, b1, w2, b2, x, out, c, cr, hw4, slice,
                     scale_out, mean_out);
  return eavsr::launch_status("ca_tail");
}

// The same attention with the border lines of t taken from the pieces the FIRST convolution's epilogue wrote (round 6: eavsr_conv2d_desc
// .border_pieces; eavsr_conv3x3_c64_h16_b): ONE launch per RCAB instead of two on its dependent chain.  pieces: [n][4][p_stride][64]
// fp32, border 0 / 1 = top / bottom row (p_rows pieces), 2 / 3 = left / right column (p_cols pieces); dtype 0 = fp32 NCHW t, 1 / 2 =
// fp16 / bf16 NHWC t (the four corner pixels are still read from t).
extern "C" int eavsr_ca_scale_pre_pieces(const void* t, const float* chan_partial, int32_t rows, const float* pieces, int32_t p_rows,
                                         int32_t p_cols, int32_t p_stride, const float* conv_weight, const float* conv_bias,
                                         const float* w1, const float* b1, const float* w2, const float* b2, float* scale, int32_t n,
                                         int32_t h, int32_t w, int32_t cr, int32_t dtype, void* stream) {
  EAVSR_REQUIRE(pieces != nullptr, -1, "ca_scale_pre_pieces: NULL pieces");
  return ca_scale_pre_launch(t, chan_partial, rows, conv_weight, conv_bias, w1, b1, w2, b2, scale, nullptr, n, h, w, cr, dtype, stream,
                             pieces, p_rows, p_cols, p_stride);
}
