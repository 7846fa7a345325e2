// Backward kernels of the streaming / sampling ops of the hot path (config 4, training).
//
// Reference: autograd through models/networks.py:597-631 (MultiAdSTN.forward), :432-464 (CALayer,
// RCABlock), :298-348 (AdaptBlock*), models/eavsrp_model.py:218-220 -- i.e. the ATen backward of
// grid_sample / interpolate / elementwise ops the reference runs implicitly in loss.backward()
// (eavsrp_model.py:109-113).  All fp32.  Scatter-type gradients (flow_warp / resize w.r.t. their input)
// use float atomics (one dword per lane, neighbouring lanes hit neighbouring addresses); they are not
// bitwise reproducible run to run, like the ATen kernels they replace.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// activation backward: g = dy * act'(y), y = the activation's OUTPUT (relu: y > 0; lrelu: slope)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ g, long count, float slope) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) g[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// ---------------------------------------------------------------------------------------------
// per-(n,c) reduction over the plane: out[nc] = sum_hw a (* b).  One workgroup per (n,c).
// ---------------------------------------------------------------------------------------------
// blockDim.x = 256 or 1024 (planes of >= 4,096 pixels: a 96 x 96 training crop is 2,304 float4 per plane -- nine dependent
// round trips per thread of 256, one and a fraction per thread of 1,024; the launch is latency, not bandwidth)
__global__ __launch_bounds__(1024) void plane_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int hw, float scale) {
  __shared__ float red[16];
  const size_t base = (size_t)blockIdx.x * hw;
  const int nt = blockDim.x;
  float s0 = 0.f, s1 = 0.f;
  if ((hw & 3) == 0) {   // whole float4 groups, 16-byte aligned planes: two independent load streams per thread
    const float4* a4 = reinterpret_cast<const float4*>(a + base);
    const float4* b4 = b ? reinterpret_cast<const float4*>(b + base) : nullptr;
    const int q = hw >> 2;
    for (int i = threadIdx.x; i < q; i += 2 * nt) {
      const int j = i + nt;
      const float4 x0 = a4[i], x1 = j < q ? a4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (b4) {
        const float4 y0 = b4[i], y1 = j < q ? b4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s0 += (x0.x * y0.x + x0.y * y0.y) + (x0.z * y0.z + x0.w * y0.w);
        s1 += (x1.x * y1.x + x1.y * y1.y) + (x1.z * y1.z + x1.w * y1.w);
      } else {
        s0 += (x0.x + x0.y) + (x0.z + x0.w);
        s1 += (x1.x + x1.y) + (x1.z + x1.w);
      }
    }
  } else if (b != nullptr) {
    for (int i = threadIdx.x; i < hw; i += nt) s0 += a[base + i] * b[base + i];
  } else {
    for (int i = threadIdx.x; i < hw; i += nt) s0 += a[base + i];
  }
  float s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = (red[0] + red[1]) + red[2] + red[3];
    for (int k = 4; k < (nt >> 6); ++k) t += red[k];
    out[blockIdx.x] = t * scale;
  }
}

// per-channel reduction over batch and plane (bias gradients): out[c] = sum_seg sum_n sum_hw a_seg[n,c,hw].  One workgroup of
// 1024 threads per channel, fixed summation order.  Up to CS_MAX_SEG tensors of one shape per launch (the uses of one bias
// across the frames of the recurrence: see conv_wgrad.hip), their pointers by value in the kernel arguments.
constexpr int CS_MAX_SEG = 8;
struct ChanSumArgs {
  const float* av[CS_MAX_SEG];
  float* out;
  int nseg, n, c, hw, accumulate;
};
__global__ __launch_bounds__(1024) void channel_sum_kernel(ChanSumArgs g) {
  __shared__ float red[16];
  const int ch = blockIdx.x;
  const int n = g.n, c = g.c, hw = g.hw;
  float s0 = 0.f, s1 = 0.f;
  for (int sg = 0; sg < g.nseg; ++sg) {
    const float* a = g.av[sg];
    for (int b = 0; b < n; ++b) {
      const float* p = a + ((size_t)b * c + ch) * hw;
      if ((hw & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const int q = hw >> 2;
        for (int i = threadIdx.x; i < q; i += 2048) {
          const int j = i + 1024;
          const float4 x0 = p4[i], x1 = j < q ? p4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          s0 += (x0.x + x0.y) + (x0.z + x0.w);
          s1 += (x1.x + x1.y) + (x1.z + x1.w);
        }
      } else {
        for (int i = threadIdx.x; i < hw; i += 1024) s0 += p[i];
      }
    }
  }
  float s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    g.out[ch] = g.accumulate ? g.out[ch] + t : t;
  }
}

// ---------------------------------------------------------------------------------------------
// scale_residual backward (out = r * s[n,c] + x):  dr = d * s + dmean[n,c] (the mean's broadcast
// gradient, already divided by hw, or NULL);  dx = d is the caller's alias.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_bwd_kernel(const float* __restrict__ d, const float* __restrict__ s,
                                                        const float* __restrict__ dmean, float* __restrict__ dr,
                                                        int hw) {
  const int nc = blockIdx.y;
  const float sc = s[nc];
  const float add = dmean ? dmean[nc] : 0.f;
  const size_t base = (size_t)nc * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) dr[base + i] = d[base + i] * sc + add;
}

// ---------------------------------------------------------------------------------------------
// channel-attention MLP, forward from the mean and backward (networks.py:436-447):
//   hid = relu(W1 m + b1); s = sigmoid(W2 hid + b2)
// backward: given ds -> dz2 = ds * s (1 - s); dW2 += dz2 hid^T; db2 += dz2; dhid = W2^T dz2;
//   dz1 = dhid * (hid > 0); dW1 += dz1 m^T; db1 += dz1; dm = W1^T dz1.   One workgroup, loops over n
//   (deterministic accumulation order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ca_mlp_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ ds,
                                                         float* __restrict__ dmean, float* __restrict__ dw1,
                                                         float* __restrict__ db1, float* __restrict__ dw2,
                                                         float* __restrict__ db2, int n, int c, int cr, int staged) {
  extern __shared__ float sm[];  // m[c] hid[cr] dz2[c] dz1[cr] (w1[cr*c] w2[c*cr] when `staged`)
  float* m = sm;
  float* hid = m + c;
  float* dz2 = hid + cr;
  float* dz1 = dz2 + c;
  const int tid = threadIdx.x;
  // the two weight matrices are read 2n times each: keep them in LDS when they are small (64 x 4 on the path)
  const float* W1 = w1;
  const float* W2 = w2;
  if (staged) {
    float* s1 = dz1 + cr;
    float* s2 = s1 + cr * c;
    for (int i = tid; i < cr * c; i += 256) { s1[i] = w1[i]; s2[i] = w2[i]; }
    W1 = s1;
    W2 = s2;
  }
  if (n == 0) {   // this kernel owns the parameter gradients
    for (int i = tid; i < cr * c; i += 256) { dw1[i] = 0.f; dw2[i] = 0.f; }
    for (int i = tid; i < cr; i += 256) db1[i] = 0.f;
    for (int i = tid; i < c; i += 256) db2[i] = 0.f;
  }
  __syncthreads();
  for (int b = 0; b < n; ++b) {
    const bool first = b == 0;   // the first clip writes the parameter gradients, the others add (fixed order)
    for (int i = tid; i < c; i += 256) m[i] = mean[(size_t)b * c + i];
    __syncthreads();
    for (int j = tid; j < cr; j += 256) {
      float v = b1[j];
      for (int k = 0; k < c; ++k) v += W1[j * c + k] * m[k];
      hid[j] = fmaxf(v, 0.f);
    }
    __syncthreads();
    for (int i = tid; i < c; i += 256) {
      float v = b2[i];
      for (int j = 0; j < cr; ++j) v += W2[i * cr + j] * hid[j];
      const float s = 1.f / (1.f + expf(-v));
      dz2[i] = ds[(size_t)b * c + i] * s * (1.f - s);
    }
    __syncthreads();
    for (int j = tid; j < cr; j += 256) {
      float v = 0.f;
      for (int i = 0; i < c; ++i) v += W2[i * cr + j] * dz2[i];
      dz1[j] = hid[j] > 0.f ? v : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < c * cr; i += 256) {
      const int ci = i / cr, j = i - ci * cr;   // dw2[ci][j]
      const float g2 = dz2[ci] * hid[j];
      dw2[i] = first ? g2 : dw2[i] + g2;
      const int j1 = i / c, k = i - j1 * c;      // dw1[j1][k]
      const float g1 = dz1[j1] * m[k];
      dw1[i] = first ? g1 : dw1[i] + g1;
    }
    for (int i = tid; i < c; i += 256) {
      db2[i] = first ? dz2[i] : db2[i] + dz2[i];
      float v = 0.f;
      for (int j = 0; j < cr; ++j) v += W1[j * c + i] * dz1[j];
      dmean[(size_t)b * c + i] = v;
    }
    for (int j = tid; j < cr; j += 256) db1[j] = first ? dz1[j] : db1[j] + dz1[j];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The whole backward of the RCAB tail out = r * s + x, s = sigmoid(W2 relu(W1 mean_hw(r) + b1) + b2) (networks.py:444-447,
// 463-464) behind the plane sums ds[n,c] = sum_hw d r, as ONE launch (round 5): it used to be ca_mlp_bwd (one workgroup, 17-23 us
// of dependent global round trips) -> an ATen multiplication by 1 / hw -> scale_residual_bwd, plus four ATen additions that
// summed the MLP's parameter gradients over the uses of a block -- ~56 us per block, 845 blocks per training step of
// configs[3].  Here every streaming workgroup (one per (n, c) plane) recomputes the 64 -> CR -> 64 MLP of its sample in wave 0
// (lane = channel; the CR hidden sums by wave reductions) and applies dr = d * s[n,c] + dmean[n,c] / hw to its plane; ONE
// extra workgroup evaluates the parameter gradients over all samples in a fixed order and writes or ADDS them to the caller's
// buffers (autograd.grad_sink: no per-use tensors, no ATen sums).  c = 64, cr <= 8 (the path's CALayer: 64 / 16 = 4).
// ---------------------------------------------------------------------------------------------
struct RcabBwdArgs {
  const float* d;        // (n, 64, hw) gradient of the block's output
  const float* scale;    // (n, 64) the forward's s
  const float* mean;     // rows == 0: (n, 64) the forward's mean_hw(r);  rows > 0: (n, rows, 64) partial channel SUMS of r (the conv
                         // epilogue's per-tile sums), added up here and scaled by 1 / hw
  int rows;
  const float* w1;       // (cr, 64)
  const float* b1;       // (cr)
  const float* w2;       // (64, cr)
  const float* b2;       // (64)
  const float* ds;       // (n, 64) sum_hw d * r, or -- ds_rows > 0 -- (n, ds_rows, 64) partial sums (a convolution's per-tile rows)
  int ds_rows;
  float* dr;             // (n, 64, hw)
  float* dw1; float* db1; float* dw2; float* db2;
  int n, cr, hw, accumulate;
  float inv_hw;
};

// mean of channel `lane` of sample b (wave-wide: lane = channel)
__device__ __forceinline__ float rb_mean(const RcabBwdArgs& a, int b, int lane) {
  if (a.rows == 0) return a.mean[(size_t)b * 64 + lane];
  const float* p = a.mean + (size_t)b * a.rows * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  int t = 0;
  for (; t + 1 < a.rows; t += 2) {
    s0 += p[(size_t)t * 64];
    s1 += p[(size_t)(t + 1) * 64];
  }
  if (t < a.rows) s0 += p[(size_t)t * 64];
  return (s0 + s1) * a.inv_hw;
}

// sum_hw d * r of channel `lane` of sample b for EVERY thread of the 256-thread workgroup (uniform call: two barriers when the sums
// come as partial rows).  Rows: the four waves take the rows q, q + 4, .. with eight requests in flight each and meet in LDS -- one
// round trip instead of a chain of them in one wave (36 rows per sample at a 96 x 96 crop: 4.4 us per launch when wave 0 walked them)
__device__ __forceinline__ float rb_ds_block(const RcabBwdArgs& a, int b, float* sds /* [4][64] */, int tid) {
  const int q = tid >> 6, lane = tid & 63;
  if (a.ds_rows == 0) return a.ds[(size_t)b * 64 + lane];
  const float* p = a.ds + (size_t)b * a.ds_rows * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  for (int r0 = q; r0 < a.ds_rows; r0 += 32) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (r0 + 4 * j < a.ds_rows) ? p[(size_t)(r0 + 4 * j) * 64] : 0.f;
    s0 += (v[0] + v[1]) + (v[2] + v[3]);
    s1 += (v[4] + v[5]) + (v[6] + v[7]);
  }
  sds[q * 64 + lane] = s0 + s1;
  __syncthreads();
  const float r = (sds[lane] + sds[64 + lane]) + (sds[128 + lane] + sds[192 + lane]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ float rb_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int CR>
__global__ __launch_bounds__(256) void rcab_tail_bwd_kernel(RcabBwdArgs a) {
  constexpr int C = 64;
  __shared__ float sh[2 + 4 * C];
  __shared__ float sds[4 * C];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nplanes = a.n * C;
  if ((int)blockIdx.x < nplanes) {
    const int b = blockIdx.x / C, ch = blockIdx.x - b * C;
    // the plane's first RB_PRE float4 per thread are requested BEFORE the barriers (they depend neither on the row sums nor on the
    // MLP that wave 0 computes), every pass issues all of its loads before its first store (d and dr may alias as far as the compiler knows:
    // load / store / load chains were nine dependent round trips per thread at a 96 x 96 crop)
    // (the partial rows of sum_hw d r first: a wave's requests return in order, and these few lines are what the MLP waits for)
    float dv[8];
    if (a.ds_rows > 0) {
      const float* p = a.ds + (size_t)b * a.ds_rows * 64 + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[j] = ((tid >> 6) + 4 * j < a.ds_rows) ? p[(size_t)((tid >> 6) + 4 * j) * 64] : 0.f;
    }
    // (and everything the MLP reads: behind the row sums' barrier these were a second round trip)
    const float pm = rb_mean(a, b, lane), pb2 = a.b2[lane];
    float pw1[CR], pb1[CR], pw2[CR];
#pragma unroll
    for (int j = 0; j < CR; ++j) {
      pw1[j] = a.w1[j * C + lane];
      pb1[j] = a.b1[j];
      pw2[j] = a.w2[lane * CR + j];
    }
    constexpr int RB_PRE = 9;
    const size_t base = (size_t)blockIdx.x * a.hw;
    const bool vec = (a.hw & 3) == 0;
    const int q = a.hw >> 2;
    const float4* d4 = reinterpret_cast<const float4*>(a.d + base);
    float4 pre[RB_PRE];
    if (vec) {
#pragma unroll
      for (int u = 0; u < RB_PRE; ++u) {
        const int i = tid + u * 256;
        pre[u] = i < q ? d4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float dsv;
    if (a.ds_rows > 0) {      // (rb_ds_block with its first eight requests per thread issued above)
      float s0 = (dv[0] + dv[1]) + (dv[2] + dv[3]), s1 = (dv[4] + dv[5]) + (dv[6] + dv[7]);
      const float* p = a.ds + (size_t)b * a.ds_rows * 64 + lane;
      for (int r0 = (tid >> 6) + 32; r0 < a.ds_rows; r0 += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r0 + 4 * j < a.ds_rows) ? p[(size_t)(r0 + 4 * j) * 64] : 0.f;
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
        s1 += (v[4] + v[5]) + (v[6] + v[7]);
      }
      sds[(tid >> 6) * 64 + lane] = s0 + s1;
      __syncthreads();
      dsv = (sds[lane] + sds[64 + lane]) + (sds[128 + lane] + sds[192 + lane]);
    } else {
      dsv = a.ds[(size_t)b * 64 + lane];
    }
    if (tid < 64) {      // wave 0: the MLP of sample b, lane = channel
      const float m = pm;
      float hid[CR], z2 = pb2;
#pragma unroll
      for (int j = 0; j < CR; ++j) {
        hid[j] = fmaxf(rb_wave_sum(pw1[j] * m) + pb1[j], 0.f);
        z2 += pw2[j] * hid[j];
      }
      const float sg = 1.f / (1.f + expf(-z2));
      const float dz2 = dsv * sg * (1.f - sg);
      float dm = 0.f;
#pragma unroll
      for (int j = 0; j < CR; ++j) {
        const float dz1 = hid[j] > 0.f ? rb_wave_sum(pw2[j] * dz2) : 0.f;
        dm += pw1[j] * dz1;
      }
      if (lane == ch) {
        sh[0] = a.scale[(size_t)b * C + ch];
        sh[1] = dm * a.inv_hw;
      }
    }
    __syncthreads();
    const float sc = sh[0], add = sh[1];
    if (vec) {
      float4* o4 = reinterpret_cast<float4*>(a.dr + base);
#pragma unroll
      for (int u = 0; u < RB_PRE; ++u) {
        const int i = tid + u * 256;
        if (i < q) o4[i] = make_float4(pre[u].x * sc + add, pre[u].y * sc + add, pre[u].z * sc + add, pre[u].w * sc + add);
      }
      for (int i0 = RB_PRE * 256; i0 < q; i0 += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + tid + u * 256;
          v[u] = i < q ? d4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + tid + u * 256;
          if (i < q) o4[i] = make_float4(v[u].x * sc + add, v[u].y * sc + add, v[u].z * sc + add, v[u].w * sc + add);
        }
      }
    } else {
      for (int i = tid; i < a.hw; i += 256) a.dr[base + i] = a.d[base + i] * sc + add;
    }
    return;
  }
  // ---- the parameter gradients: one workgroup, samples in order (deterministic) -----------------------------------------
  float* m = sh + 2;      // [C]
  float* hidv = m + C;    // [CR] (C reserved)
  float* dz2 = hidv + C;  // [C]
  float* dz1 = dz2 + C;   // [CR]
  const float qb2 = a.b2[lane];
  float qw1[CR], qb1[CR], qw2[CR];
#pragma unroll
  for (int j = 0; j < CR; ++j) {
    qw1[j] = a.w1[j * C + lane];
    qb1[j] = a.b1[j];
    qw2[j] = a.w2[lane * CR + j];
  }
  // what the four parameter gradients are added to (accumulate), requested before anything else; the sums over the samples in
  // registers (sample order: deterministic)
  constexpr int NU = (C * CR + 255) / 256;
  float odw2[NU], odw1[NU], adw2[NU], adw1[NU], odb2 = 0.f, odb1 = 0.f, adb2 = 0.f, adb1 = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = tid + 256 * u;
    odw2[u] = (a.accumulate && i < C * CR) ? a.dw2[i] : 0.f;
    odw1[u] = (a.accumulate && i < C * CR) ? a.dw1[i] : 0.f;
    adw2[u] = adw1[u] = 0.f;
  }
  if (a.accumulate && tid < C) odb2 = a.db2[tid];
  if (a.accumulate && tid < CR) odb1 = a.db1[tid];
  for (int b = 0; b < a.n; ++b) {
    const float mv = rb_mean(a, b, lane);      // (requested in front of the row sums: one round trip for both)
    const float dsv = rb_ds_block(a, b, sds, tid);
    if (tid < C) m[tid] = mv;
    __syncthreads();
    if (tid < 64) {
      float z2 = qb2;
#pragma unroll
      for (int j = 0; j < CR; ++j) {
        const float hj = fmaxf(rb_wave_sum(qw1[j] * m[lane]) + qb1[j], 0.f);
        if (lane == 0) hidv[j] = hj;
        z2 += qw2[j] * hj;
      }
      const float sg = 1.f / (1.f + expf(-z2));
      const float g2 = dsv * sg * (1.f - sg);
      dz2[lane] = g2;
#pragma unroll
      for (int j = 0; j < CR; ++j) {
        const float t = rb_wave_sum(qw2[j] * g2);
        if (lane == 0) dz1[j] = t;
      }
    }
    __syncthreads();
    if (tid < CR) dz1[tid] = hidv[tid] > 0.f ? dz1[tid] : 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = tid + 256 * u;
      if (i < C * CR) {
        const int ci = i / CR, j = i - ci * CR;   // dw2[ci][j]
        adw2[u] += dz2[ci] * hidv[j];
        const int j1 = i / C, k = i - j1 * C;      // dw1[j1][k]
        adw1[u] += dz1[j1] * m[k];
      }
    }
    if (tid < C) adb2 += dz2[tid];
    if (tid < CR) adb1 += dz1[tid];
    __syncthreads();
  }
  // the samples' sums leave once (they were a store and a dependent load of the same address per sample)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = tid + 256 * u;
    if (i < C * CR) {
      a.dw2[i] = odw2[u] + adw2[u];
      a.dw1[i] = odw1[u] + adw1[u];
    }
  }
  if (tid < C) a.db2[tid] = odb2 + adb2;
  if (tid < CR) a.db1[tid] = odb1 + adb1;
}

// ---------------------------------------------------------------------------------------------
// flow_warp backward (zeros padding; the border mode only occurs inside the frozen SPyNet).
// One thread per pixel, all channels: dx gets 4 atomic adds per channel, dflow is summed in registers.
// d(sample position)/d(flow) = 1 (the reference's normalise / un-normalise pair cancels).
// ---------------------------------------------------------------------------------------------
// Round 5: one workgroup = 64 pixels of a row x FWB_SL channel slices (threadIdx.y); a thread walks c / FWB_SL channels, so a
// pixel's 4 c atomics are in flight from FWB_SL waves instead of queued behind one another in one thread (81 -> see DESIGN.md 6),
// and the slices' d(flow) partial sums meet in LDS (fixed order).
constexpr int FWB_SL = 8;
__global__ __launch_bounds__(64 * FWB_SL) void flow_warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                                    const float* __restrict__ flow2,
                                                                    const float* __restrict__ dout, float* __restrict__ dx,
                                                                    float* __restrict__ dflow, int c, int h, int w) {
  __shared__ float s_g[2][FWB_SL][64];
  const int px = blockIdx.x * 64 + threadIdx.x;
  const int py = blockIdx.y;
  const int sl = threadIdx.y;
  const int bn = blockIdx.z;
  const bool live = px < w;
  const size_t plane = (size_t)h * w;
  const size_t fo = (size_t)bn * 2 * plane + (size_t)py * w + min(px, w - 1);
  float fx = flow[fo], fy = flow[fo + plane];
  if (flow2) { fx += flow2[fo]; fy += flow2[fo + plane]; }
  const float gx = (float)px + fx, gy = (float)py + fy;
  const float nx = 2.0f * gx / (float)max(w - 1, 1) - 1.0f;
  const float ny = 2.0f * gy / (float)max(h - 1, 1) - 1.0f;
  float ix = ((nx + 1.0f) / 2.0f) * (float)(w - 1);
  float iy = ((ny + 1.0f) / 2.0f) * (float)(h - 1);
  ix = fminf(fmaxf(ix, -4.0f), (float)w + 4.0f);
  iy = fminf(fmaxf(iy, -4.0f), (float)h + 4.0f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x1 >= 0) & (x1 < w);
  const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y1 >= 0) & (y1 < h);
  const bool v_nw = vx0 & vy0 & live, v_ne = vx1 & vy0 & live, v_sw = vx0 & vy1 & live, v_se = vx1 & vy1 & live;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  const int i_nw = cy0 * w + cx0, i_ne = cy0 * w + cx1, i_sw = cy1 * w + cx0, i_se = cy1 * w + cx1;
  const float w_nw = wx0 * wy0, w_ne = wx1 * wy0, w_sw = wx0 * wy1, w_se = wx1 * wy1;
  float gix = 0.f, giy = 0.f;
  const int cper = (c + FWB_SL - 1) / FWB_SL;
  const int c_lo = sl * cper, c_hi = min(c, c_lo + cper);
  for (int cc = c_lo; cc < c_hi; ++cc) {
    const size_t cb = ((size_t)bn * c + cc) * plane;
    const float g = live ? dout[cb + (size_t)py * w + px] : 0.f;
    if (dx != nullptr) {
      if (v_nw) atomicAdd(dx + cb + i_nw, g * w_nw);
      if (v_ne) atomicAdd(dx + cb + i_ne, g * w_ne);
      if (v_sw) atomicAdd(dx + cb + i_sw, g * w_sw);
      if (v_se) atomicAdd(dx + cb + i_se, g * w_se);
    }
    if (dflow != nullptr) {
      const float a = v_nw ? x[cb + i_nw] : 0.f, b = v_ne ? x[cb + i_ne] : 0.f;
      const float cv = v_sw ? x[cb + i_sw] : 0.f, d = v_se ? x[cb + i_se] : 0.f;
      gix += g * ((b - a) * wy0 + (d - cv) * wy1);
      giy += g * ((cv - a) * wx0 + (d - b) * wx1);
    }
  }
  if (dflow != nullptr) {      // (uniform over the workgroup)
    s_g[0][sl][threadIdx.x] = gix;
    s_g[1][sl][threadIdx.x] = giy;
    __syncthreads();
    if (sl < 2 && live) {
      float v = s_g[sl][0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < FWB_SL; ++k) v += s_g[sl][k][threadIdx.x];
      dflow[fo + (sl ? plane : 0)] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// resize_bilinear_ac backward: din (+)= scale * scatter(dout)   (din pre-zeroed by the caller)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                            int hin, int win, int hout, int wout, float rh, float rw,
                                                            float scale) {
  const int ox = blockIdx.x * 64 + threadIdx.x;
  const int oy = blockIdx.y * 4 + threadIdx.y;
  const int nc = blockIdx.z;
  if (ox >= wout || oy >= hout) return;
  const float sy = rh * (float)oy, sx = rw * (float)ox;
  const int y0 = min((int)sy, hin - 1), x0 = min((int)sx, win - 1);
  const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float g = dout[(size_t)nc * hout * wout + (size_t)oy * wout + ox] * scale;
  float* p = din + (size_t)nc * hin * win;
  atomicAdd(p + y0 * win + x0, g * ly0 * lx0);
  atomicAdd(p + y0 * win + x1, g * ly0 * lx1);
  atomicAdd(p + y1 * win + x0, g * ly1 * lx0);
  atomicAdd(p + y1 * win + x1, g * ly1 * lx1);
}

// pyramid backward: din = 0.25 * dd2[y/2][x/2] + (inner 2x2 of each 4x4 block) 0.25 * dd4[y/4][x/4]
__global__ __launch_bounds__(256) void pyramid_bwd_kernel(const float* __restrict__ dd2, const float* __restrict__ dd4,
                                                          float* __restrict__ din, int h, int w) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int nc = blockIdx.z;
  if (x >= w || y >= h) return;
  float v = 0.25f * dd2[(size_t)nc * (h / 2) * (w / 2) + (size_t)(y >> 1) * (w / 2) + (x >> 1)];
  const int ry = y & 3, rx = x & 3;
  if ((ry == 1 || ry == 2) && (rx == 1 || rx == 2))
    v += 0.25f * dd4[(size_t)nc * (h / 4) * (w / 4) + (size_t)(y >> 2) * (w / 4) + (x >> 2)];
  din[(size_t)nc * h * w + (size_t)y * w + x] = v;
}

// affine_offsets backward: d(heads) from d(offset) [, d(mask) with the saved mask = sigmoid(logit)]
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ doff, const float* __restrict__ dmask,
                                                         const float* __restrict__ mask, float* __restrict__ dheads,
                                                         int D, int hw, int head_c) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int g = blockIdx.y, bn = blockIdx.z;
  const float* op = doff + ((size_t)bn * D * 18 + (size_t)g * 18) * hw + p;
  float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f, ty = 0.f, tx = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float ry = (float)(k / 3 - 1), rx = (float)(k % 3 - 1);
    const float gy = op[(size_t)(2 * k) * hw], gx = op[(size_t)(2 * k + 1) * hw];
    t00 += gy * ry; t01 += gy * rx; t10 += gx * ry; t11 += gx * rx;
    ty += gy; tx += gx;
  }
  float* hp = dheads + (size_t)bn * head_c * hw + p;
  hp[(size_t)(g * 4 + 0) * hw] = t00;
  hp[(size_t)(g * 4 + 1) * hw] = t01;
  hp[(size_t)(g * 4 + 2) * hw] = t10;
  hp[(size_t)(g * 4 + 3) * hw] = t11;
  hp[(size_t)(4 * D + g * 2 + 0) * hw] = ty;
  hp[(size_t)(4 * D + g * 2 + 1) * hw] = tx;
  if (dmask != nullptr) {
    const size_t mo = ((size_t)bn * D * 9 + (size_t)g * 9) * hw + p;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float s = mask[mo + (size_t)k * hw];
      hp[(size_t)(6 * D + g * 9 + k) * hw] = dmask[mo + (size_t)k * hw] * s * (1.f - s);
    }
  }
}

}  // namespace

extern "C" int eavsr_act_bwd_f32(const float* dy, const float* y, float* g, int64_t count, int32_t act, float slope,
                                 void* stream) {
  EAVSR_REQUIRE(dy && y && g, -1, "act_bwd: NULL pointer");
  EAVSR_REQUIRE(act == EAVSR_ACT_RELU || act == EAVSR_ACT_LRELU, -1, "act_bwd: act %d", act);
  if (count <= 0) return 0;
  long blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, eavsr::as_stream(stream), dy, y, g,
                     (long)count, act == EAVSR_ACT_RELU ? 0.f : slope);
  return eavsr::launch_status("act_bwd");
}

extern "C" int eavsr_plane_sum_f32(const float* a, const float* b, float* out, int32_t nc, int32_t hw, float scale,
                                   void* stream) {
  EAVSR_REQUIRE(a && out, -1, "plane_sum: NULL pointer");
  EAVSR_REQUIRE(nc >= 0 && hw > 0, -1, "plane_sum: bad dims");
  if (nc == 0) return 0;
  hipLaunchKernelGGL(plane_sum_kernel, dim3(nc), dim3(hw >= 4096 ? 1024 : 256), 0, eavsr::as_stream(stream), a, b, out, hw, scale);
  return eavsr::launch_status("plane_sum");
}

extern "C" int eavsr_channel_sum_multi_f32(const void* const* a_list, int32_t nseg, float* out, int32_t n, int32_t c, int32_t hw,
                                           int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(a_list && out, -1, "channel_sum: NULL pointer");
  EAVSR_REQUIRE(nseg >= 1 && nseg <= CS_MAX_SEG, -1, "channel_sum: %d segments (1..%d)", nseg, CS_MAX_SEG);
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hw > 0, -1, "channel_sum: bad dims");
  if (c == 0) return 0;
  ChanSumArgs g;
  for (int s = 0; s < CS_MAX_SEG; ++s) {
    g.av[s] = reinterpret_cast<const float*>(a_list[s < nseg ? s : 0]);
    EAVSR_REQUIRE(g.av[s], -1, "channel_sum: NULL segment pointer");
  }
  g.out = out; g.nseg = nseg; g.n = n; g.c = c; g.hw = hw; g.accumulate = accumulate;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(c), dim3(1024), 0, eavsr::as_stream(stream), g);
  return eavsr::launch_status("channel_sum");
}

extern "C" int eavsr_channel_sum_f32(const float* a, float* out, int32_t n, int32_t c, int32_t hw, int32_t accumulate,
                                     void* stream) {
  EAVSR_REQUIRE(a && out, -1, "channel_sum: NULL pointer");
  const void* al[1] = {a};
  return eavsr_channel_sum_multi_f32(al, 1, out, n, c, hw, accumulate, stream);
}

extern "C" int eavsr_scale_residual_bwd_f32(const float* d, const float* scale, const float* dmean, float* dr,
                                            int32_t n, int32_t c, int32_t hw, void* stream) {
  EAVSR_REQUIRE(d && scale && dr, -1, "scale_residual_bwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hw > 0 && (long)n * c <= 65535, -1, "scale_residual_bwd: bad dims");
  if (n * c == 0) return 0;
  int bx = eavsr::cdiv(hw, 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(scale_bwd_kernel, dim3(bx, n * c), dim3(256), 0, eavsr::as_stream(stream), d, scale, dmean, dr, hw);
  return eavsr::launch_status("scale_residual_bwd");
}

extern "C" int eavsr_ca_mlp_bwd_f32(const float* mean, const float* w1, const float* b1, const float* w2,
                                    const float* b2, const float* dscale, float* dmean, float* dw1, float* db1,
                                    float* dw2, float* db2, int32_t n, int32_t c, int32_t cr, void* stream) {
  EAVSR_REQUIRE(mean && w1 && b1 && w2 && b2 && dscale && dmean && dw1 && db1 && dw2 && db2, -1, "ca_mlp_bwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && cr > 0 && c <= 1024 && cr <= 1024, -1, "ca_mlp_bwd: bad dims");
  const int staged = (size_t)c * cr <= 4096;
  hipLaunchKernelGGL(ca_mlp_bwd_kernel, dim3(1), dim3(256),
                     (size_t)(2 * c + 2 * cr + (staged ? 2 * c * cr : 0)) * sizeof(float), eavsr::as_stream(stream), mean,
                     w1, b1, w2, b2, dscale, dmean, dw1, db1, dw2, db2, n, c, cr, staged);
  return eavsr::launch_status("ca_mlp_bwd");
}

extern "C" int eavsr_rcab_tail_bwd_f32(const float* d, const float* scale, const float* mean, const float* w1, const float* b1,
                                       const float* w2, const float* b2, const float* dscale, float* dr, float* dw1, float* db1,
                                       float* dw2, float* db2, int32_t n, int32_t c, int32_t cr, int32_t hw, int32_t mean_rows,
                                       int32_t dscale_rows, int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(d && scale && mean && w1 && b1 && w2 && b2 && dscale && dr && dw1 && db1 && dw2 && db2, -1, "rcab_tail_bwd: NULL pointer");
  EAVSR_REQUIRE(c == 64 && (cr == 4 || cr == 8 || cr == 2 || cr == 1), -2,
                "rcab_tail_bwd: %d channels / %d hidden units unsupported (64 channels, 1 / 2 / 4 / 8 hidden units)", c, cr);
  EAVSR_REQUIRE(n >= 1 && hw > 0 && (long)n * c < 65535 && mean_rows >= 0 && dscale_rows >= 0, -1, "rcab_tail_bwd: bad dims");
  EAVSR_REQUIRE((((uintptr_t)d | (uintptr_t)dr) & 15) == 0 || (hw & 3), -2, "rcab_tail_bwd: d / dr must be 16-byte aligned");
  RcabBwdArgs a;
  a.d = d; a.scale = scale; a.mean = mean; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.ds = dscale; a.dr = dr;
  a.dw1 = dw1; a.db1 = db1; a.dw2 = dw2; a.db2 = db2;
  a.n = n; a.cr = cr; a.hw = hw; a.accumulate = accumulate; a.inv_hw = 1.0f / (float)hw; a.rows = mean_rows; a.ds_rows = dscale_rows;
  hipStream_t st = eavsr::as_stream(stream);
  const dim3 grid(n * c + 1), block(256);
  switch (cr) {
    case 1: hipLaunchKernelGGL(rcab_tail_bwd_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(rcab_tail_bwd_kernel<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(rcab_tail_bwd_kernel<4>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(rcab_tail_bwd_kernel<8>, grid, block, 0, st, a); break;
  }
  return eavsr::launch_status("rcab_tail_bwd");
}

extern "C" int eavsr_flow_warp_bwd_f32(const float* x, const float* flow, const float* flow2, const float* dout,
                                       float* dx, float* dflow, int32_t n, int32_t c, int32_t h, int32_t w,
                                       void* stream) {
  EAVSR_REQUIRE(x && flow && dout, -1, "flow_warp_bwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && h > 0 && w > 0 && n <= 65535, -1, "flow_warp_bwd: bad dims");
  if (n == 0 || c == 0) return 0;
  EAVSR_REQUIRE(h <= 65535, -1, "flow_warp_bwd: image too tall for the launch grid");
  dim3 grid(eavsr::cdiv(w, 64), h, n), block(64, FWB_SL, 1);
  hipLaunchKernelGGL(flow_warp_bwd_kernel, grid, block, 0, eavsr::as_stream(stream), x, flow, flow2, dout, dx, dflow, c,
                     h, w);
  return eavsr::launch_status("flow_warp_bwd");
}

extern "C" int eavsr_resize_bilinear_ac_bwd_f32(const float* dout, float* din, int32_t n, int32_t c, int32_t hin,
                                                int32_t win, int32_t hout, int32_t wout, float scale, void* stream) {
  EAVSR_REQUIRE(dout && din, -1, "resize_bilinear_ac_bwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hin > 0 && win > 0 && hout > 0 && wout > 0 && (long)n * c <= 65535, -1,
                "resize_bilinear_ac_bwd: bad dims");
  if (n * c == 0) return 0;
  const float rh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
  const float rw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
  dim3 grid(eavsr::cdiv(wout, 64), eavsr::cdiv(hout, 4), n * c), block(64, 4, 1);
  hipLaunchKernelGGL(resize_ac_bwd_kernel, grid, block, 0, eavsr::as_stream(stream), dout, din, hin, win, hout, wout,
                     rh, rw, scale);
  return eavsr::launch_status("resize_bilinear_ac_bwd");
}

extern "C" int eavsr_pyramid_bwd_f32(const float* ddown2, const float* ddown4, float* din, int32_t nc, int32_t h,
                                     int32_t w, void* stream) {
  EAVSR_REQUIRE(ddown2 && ddown4 && din, -1, "pyramid_bwd: NULL pointer");
  EAVSR_REQUIRE(nc >= 0 && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0 && nc <= 65535, -1, "pyramid_bwd: bad dims");
  if (nc == 0) return 0;
  dim3 grid(eavsr::cdiv(w, 64), eavsr::cdiv(h, 4), nc), block(64, 4, 1);
  hipLaunchKernelGGL(pyramid_bwd_kernel, grid, block, 0, eavsr::as_stream(stream), ddown2, ddown4, din, h, w);
  return eavsr::launch_status("pyramid_bwd");
}

extern "C" int eavsr_affine_offsets_bwd_f32(const float* doffset, const float* dmask, const float* mask, float* dheads,
                                            int32_t n, int32_t D, int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(doffset && dheads, -1, "affine_offsets_bwd: NULL pointer");
  EAVSR_REQUIRE((dmask == nullptr) == (mask == nullptr), -1, "affine_offsets_bwd: dmask and mask go together");
  EAVSR_REQUIRE(n >= 0 && D > 0 && h > 0 && w > 0 && D <= 65535 && n <= 65535, -1, "affine_offsets_bwd: bad dims");
  if (n == 0) return 0;
  const int hw = h * w;
  dim3 grid(eavsr::cdiv(hw, 256), D, n);
  hipLaunchKernelGGL(affine_bwd_kernel, grid, dim3(256), 0, eavsr::as_stream(stream), doffset, dmask, mask, dheads, D,
                     hw, dmask ? 15 * D : 6 * D);
  return eavsr::launch_status("affine_offsets_bwd");
}
