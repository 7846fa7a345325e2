// Backward of DCNv2 and of the predictor's grouped 3x3 convolutions (config 4, training).
//
// DCNv2 backward follows the structure of the published mmcv 1.x op (modulated_deform_conv backward:
// columns = im2col(x, offset, mask); dW = dOut . columns^T; dcolumns = W^T . dOut; then col2im for the
// input gradient and col2im_coord for the offset / mask gradients) -- the two GEMMs run on the MFMA
// kernels (eavsr_conv_wgrad_f32 with k = 1 and eavsr_conv2d_f32 with k = 1 on the column tensor), the two
// samplers are here.  Training patches are small (96 x 96), so the column tensor (n, cin*9, h, w) in HBM
// is affordable in this direction, unlike in the forward kernel.
#include "common.h"

namespace {

struct Samp {
  bool in;
  float w1, w2, w3, w4;     // corner weights with validity folded in
  float hh, hw, lh, lw;
  bool v1, v2, v3, v4;
  int i1, i2, i3, i4;
};

__device__ __forceinline__ Samp make_samp(float py, float px, int h, int w) {
  Samp s;
  s.in = py > -1.f && px > -1.f && py < (float)h && px < (float)w;
  const float fy0 = floorf(py), fx0 = floorf(px);
  s.lh = py - fy0; s.lw = px - fx0; s.hh = 1.f - s.lh; s.hw = 1.f - s.lw;
  const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)w);
  const int hh_i = hl + 1, wh_i = wl + 1;
  const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
  s.v1 = s.in && t_ok && l_ok; s.v2 = s.in && t_ok && r_ok; s.v3 = s.in && b_ok && l_ok; s.v4 = s.in && b_ok && r_ok;
  s.w1 = s.v1 ? s.hh * s.hw : 0.f; s.w2 = s.v2 ? s.hh * s.lw : 0.f;
  s.w3 = s.v3 ? s.lh * s.hw : 0.f; s.w4 = s.v4 ? s.lh * s.lw : 0.f;
  const int cy0 = min(max(hl, 0), h - 1), cy1 = min(max(hh_i, 0), h - 1);
  const int cx0 = min(max(wl, 0), w - 1), cx1 = min(max(wh_i, 0), w - 1);
  s.i1 = cy0 * w + cx0; s.i2 = cy0 * w + cx1; s.i3 = cy1 * w + cx0; s.i4 = cy1 * w + cx1;
  return s;
}

// columns[n][c*9 + k][px] = bilinear(x[n,c], p + tap_k + offset) * mask      one thread per (n, g, k, px)
template <int CPG>      // > 0: channels per deformable group at compile time (the channel loop unrolls: 4 CPG loads in flight)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                         const float* __restrict__ mask, float* __restrict__ col,
                                                         int c, int h, int w, int dg) {
  const int hw = h * w;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int g = blockIdx.y / 9, k = blockIdx.y % 9, bn = blockIdx.z;
  const int cpg = CPG > 0 ? CPG : c / dg;
  const int gy = p / w, gx = p - gy * w;
  const float oy = offset[((size_t)bn * dg * 18 + g * 18 + 2 * k) * hw + p];
  const float ox = offset[((size_t)bn * dg * 18 + g * 18 + 2 * k + 1) * hw + p];
  const float m = mask[((size_t)bn * dg * 9 + g * 9 + k) * hw + p];
  const Samp s = make_samp((float)(gy - 1 + k / 3) + oy, (float)(gx - 1 + k % 3) + ox, h, w);
#pragma unroll
  for (int cc = 0; cc < cpg; ++cc) {
    const int ch = g * cpg + cc;
    const float* q = x + ((size_t)bn * c + ch) * hw;
    float v = s.w1 * q[s.i1];
    v += s.w2 * q[s.i2];
    v += s.w3 * q[s.i3];
    v += s.w4 * q[s.i4];
    col[((size_t)bn * c * 9 + (size_t)ch * 9 + k) * hw + p] = v * m;
  }
}

// from dcolumns: dx (atomics), doffset, dmask                                  one thread per (n, g, k, px)
// CPG > 0: the channels per deformable group as a compile-time constant (8 in the model) -- the channel loop unrolls and its
// 5 CPG loads are all requested before the first use (the runtime loop was CPG dependent round trips per thread)
template <int CPG>
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                         const float* __restrict__ mask, const float* __restrict__ dcol,
                                                         float* __restrict__ dx, float* __restrict__ doffset,
                                                         float* __restrict__ dmask, int c, int h, int w, int dg) {
  const int hw = h * w;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int g = blockIdx.y / 9, k = blockIdx.y % 9, bn = blockIdx.z;
  const int cpg = CPG > 0 ? CPG : c / dg;
  const int gy = p / w, gx = p - gy * w;
  const size_t oi = ((size_t)bn * dg * 18 + g * 18 + 2 * k) * hw + p;
  const size_t mi = ((size_t)bn * dg * 9 + g * 9 + k) * hw + p;
  const float oy = offset[oi], ox = offset[oi + hw], m = mask[mi];
  const Samp s = make_samp((float)(gy - 1 + k / 3) + oy, (float)(gx - 1 + k % 3) + ox, h, w);
  float gm = 0.f, gpy = 0.f, gpx = 0.f;
  if constexpr (CPG > 0) {
    float dcv[CPG], a1[CPG], a2[CPG], a3[CPG], a4[CPG];
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {      // (clamped indices: every load is in range; validity is applied to the values)
      const int ch = g * CPG + cc;
      const float* q = x + ((size_t)bn * c + ch) * hw;
      dcv[cc] = dcol[((size_t)bn * c * 9 + (size_t)ch * 9 + k) * hw + p];
      a1[cc] = q[s.i1]; a2[cc] = q[s.i2]; a3[cc] = q[s.i3]; a4[cc] = q[s.i4];
    }
#pragma unroll
    for (int cc = 0; cc < CPG; ++cc) {
      const float b1 = s.v1 ? a1[cc] : 0.f, b2 = s.v2 ? a2[cc] : 0.f, b3 = s.v3 ? a3[cc] : 0.f, b4 = s.v4 ? a4[cc] : 0.f;
      const float dc = dcv[cc];
      const float val = s.hh * s.hw * b1 + s.hh * s.lw * b2 + s.lh * s.hw * b3 + s.lh * s.lw * b4;
      gm += dc * val;
      const float dv = dc * m;
      gpy += dv * ((b3 - b1) * s.hw + (b4 - b2) * s.lw);
      gpx += dv * ((b2 - b1) * s.hh + (b4 - b3) * s.lh);
      if (dx != nullptr) {
        float* dq = dx + ((size_t)bn * c + g * CPG + cc) * hw;
        if (s.v1) atomicAdd(dq + s.i1, dv * s.w1);
        if (s.v2) atomicAdd(dq + s.i2, dv * s.w2);
        if (s.v3) atomicAdd(dq + s.i3, dv * s.w3);
        if (s.v4) atomicAdd(dq + s.i4, dv * s.w4);
      }
    }
  } else {
    for (int cc = 0; cc < cpg; ++cc) {
      const int ch = g * cpg + cc;
      const float dc = dcol[((size_t)bn * c * 9 + (size_t)ch * 9 + k) * hw + p];
      const float* q = x + ((size_t)bn * c + ch) * hw;
      const float a1 = s.v1 ? q[s.i1] : 0.f, a2 = s.v2 ? q[s.i2] : 0.f, a3 = s.v3 ? q[s.i3] : 0.f, a4 = s.v4 ? q[s.i4] : 0.f;
      const float val = s.hh * s.hw * a1 + s.hh * s.lw * a2 + s.lh * s.hw * a3 + s.lh * s.lw * a4;
      gm += dc * val;
      const float dv = dc * m;
      gpy += dv * ((a3 - a1) * s.hw + (a4 - a2) * s.lw);
      gpx += dv * ((a2 - a1) * s.hh + (a4 - a3) * s.lh);
      if (dx != nullptr) {
        float* dq = dx + ((size_t)bn * c + ch) * hw;
        if (s.v1) atomicAdd(dq + s.i1, dv * s.w1);
        if (s.v2) atomicAdd(dq + s.i2, dv * s.w2);
        if (s.v3) atomicAdd(dq + s.i3, dv * s.w3);
        if (s.v4) atomicAdd(dq + s.i4, dv * s.w4);
      }
    }
  }
  doffset[oi] = s.in ? gpy : 0.f;
  doffset[oi + hw] = s.in ? gpx : 0.f;
  dmask[mi] = gm;
}

// ---------------------------------------------------------------------------------------------
// grouped 3x3 conv with ONE output channel per group and cpg (1 or 2) input channels per group: the
// `concat` (depthwise) and `concat2` layers of AdaptBlock2_3x3 / AdaptBlockOffset (networks.py:290-291,
// 327-328) in their un-fused training form.  act: 0 none, 2 LeakyReLU(slope).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gconv3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                           const float* __restrict__ b, float* __restrict__ out,
                                                           int co, int cpg, int h, int w, float slope, int act) {
  const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
  const int o = blockIdx.z % co, bn = blockIdx.z / co;
  if (px >= w || py >= h) return;
  float v = b ? b[o] : 0.f;
  for (int j = 0; j < cpg; ++j) {
    const float* q = x + ((size_t)bn * co * cpg + (size_t)o * cpg + j) * h * w;
    const float* wk = wt + ((size_t)o * cpg + j) * 9;
    float qv[9];      // clamped addresses, all nine loads first (see gconv3x3_dgrad_kernel)
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = py + ky - 1, xx = px + kx - 1;
        ok[ky * 3 + kx] = yy >= 0 && yy < h && xx >= 0 && xx < w;
        qv[ky * 3 + kx] = q[(size_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
      }
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (ok[k]) v += wk[k] * qv[k];
  }
  if (act == EAVSR_ACT_LRELU) v = v > 0.f ? v : v * slope;
  out[((size_t)bn * co + o) * h * w + (size_t)py * w + px] = v;
}

// dx[n, o*cpg + j, y, x] = sum_tap w[o][j][tap] * g[n, o, y - ky + 1, x - kx + 1]
__global__ __launch_bounds__(256) void gconv3x3_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ wt,
                                                             float* __restrict__ dx, int co, int cpg, int h, int w) {
  const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
  const int ci = blockIdx.z % (co * cpg), bn = blockIdx.z / (co * cpg);
  if (px >= w || py >= h) return;
  const int o = ci / cpg, j = ci - o * cpg;
  const float* q = g + ((size_t)bn * co + o) * h * w;
  const float* wk = wt + ((size_t)o * cpg + j) * 9;
  // the nine taps with clamped addresses, all loads first (as nine conditional loads each sat behind its own branch and wait)
  float qv[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = py - ky + 1, xx = px - kx + 1;
      ok[ky * 3 + kx] = yy >= 0 && yy < h && xx >= 0 && xx < w;
      qv[ky * 3 + kx] = q[(size_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
    }
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (ok[k]) v += wk[k] * qv[k];      // (a select: the same sum in the same order as before)
  dx[((size_t)bn * co * cpg + ci) * h * w + (size_t)py * w + px] = v;
}

// dw[o][j][tap] = sum_{n,y,x} g[n,o,y,x] x[n,o*cpg+j,y+ky-1,x+kx-1] ; db[o] = sum g   one workgroup per (o, j)
__global__ __launch_bounds__(1024) void gconv3x3_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              float* __restrict__ dw, float* __restrict__ db, int n,
                                                              int co, int cpg, int h, int w, int accumulate) {
  __shared__ float red[16][10];
  const int o = blockIdx.x / cpg, j = blockIdx.x - o * cpg;
  float acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.f;
  const int hw = h * w;
  for (int bn = 0; bn < n; ++bn) {
    const float* gq = g + ((size_t)bn * co + o) * hw;
    const float* xq = x + ((size_t)bn * co * cpg + (size_t)o * cpg + j) * hw;
    // two pixels per pass: twenty loads in flight per thread (a 96 x 96 crop is nine passes of one pixel: nine round trips)
    for (int p0 = threadIdx.x; p0 < hw; p0 += 2048) {
      float gv[2], xv[2][9];
      bool ok[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = p0 + u * 1024;
        const bool in = p < hw;
        const int pc = in ? p : 0;
        const int py = pc / w, px = pc - py * w;
        gv[u] = in ? gq[pc] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {      // clamped addresses, all loads first (nine conditional loads were nine round trips)
            const int yy = py + ky - 1, xx = px + kx - 1;
            ok[u][ky * 3 + kx] = in && yy >= 0 && yy < h && xx >= 0 && xx < w;
            xv[u][ky * 3 + kx] = xq[(size_t)min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
          }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {      // (pixel p0 before pixel p0 + 1024: the order of the one-pixel loop)
        acc[9] += gv[u];
#pragma unroll
        for (int k = 0; k < 9; ++k)
          if (ok[u][k]) acc[k] += gv[u] * xv[u][k];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    float v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
    // accumulate: the uses of the weight across the frames of the recurrence add into one buffer (autograd.grad_sink)
    if (threadIdx.x < 9) {
      float* d = dw + ((size_t)o * cpg + j) * 9 + threadIdx.x;
      *d = accumulate ? *d + v : v;
    } else if (j == 0 && db != nullptr) {
      db[o] = accumulate ? db[o] + v : v;
    }
  }
}

}  // namespace

extern "C" int eavsr_dcnv2_im2col_f32(const float* x, const float* offset, const float* mask, float* columns,
                                      int32_t n, int32_t c, int32_t h, int32_t w, int32_t deform_groups, void* stream) {
  EAVSR_REQUIRE(x && offset && mask && columns, -1, "dcnv2_im2col: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && deform_groups > 0 && c % deform_groups == 0 && n <= 65535 &&
                    deform_groups * 9 <= 65535, -1, "dcnv2_im2col: bad dims");
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(h * w, 256), deform_groups * 9, n);
  if (c / deform_groups == 8)
    hipLaunchKernelGGL(dcn_im2col_kernel<8>, grid, dim3(256), 0, eavsr::as_stream(stream), x, offset, mask, columns, c, h, w,
                       deform_groups);
  else
    hipLaunchKernelGGL(dcn_im2col_kernel<0>, grid, dim3(256), 0, eavsr::as_stream(stream), x, offset, mask, columns, c, h, w,
                       deform_groups);
  return eavsr::launch_status("dcnv2_im2col");
}

extern "C" int eavsr_dcnv2_col2im_f32(const float* x, const float* offset, const float* mask, const float* dcolumns,
                                      float* dx, float* doffset, float* dmask, int32_t n, int32_t c, int32_t h,
                                      int32_t w, int32_t deform_groups, void* stream) {
  EAVSR_REQUIRE(x && offset && mask && dcolumns && doffset && dmask, -1, "dcnv2_col2im: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0 && deform_groups > 0 && c % deform_groups == 0 && n <= 65535 &&
                    deform_groups * 9 <= 65535, -1, "dcnv2_col2im: bad dims");
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(h * w, 256), deform_groups * 9, n);
  if (c / deform_groups == 8)
    hipLaunchKernelGGL(dcn_col2im_kernel<8>, grid, dim3(256), 0, eavsr::as_stream(stream), x, offset, mask, dcolumns, dx,
                       doffset, dmask, c, h, w, deform_groups);
  else
    hipLaunchKernelGGL(dcn_col2im_kernel<0>, grid, dim3(256), 0, eavsr::as_stream(stream), x, offset, mask, dcolumns, dx,
                       doffset, dmask, c, h, w, deform_groups);
  return eavsr::launch_status("dcnv2_col2im");
}

extern "C" int eavsr_gconv3x3_fwd_f32(const float* x, const float* weight, const float* bias, float* out, int32_t n,
                                      int32_t cout, int32_t cpg, int32_t h, int32_t w, int32_t act, float slope,
                                      void* stream) {
  EAVSR_REQUIRE(x && weight && out, -1, "gconv3x3_fwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cout > 0 && cpg > 0 && h > 0 && w > 0 && (long)n * cout <= 65535, -1, "gconv3x3_fwd: bad dims");
  EAVSR_REQUIRE(act == EAVSR_ACT_NONE || act == EAVSR_ACT_LRELU, -1, "gconv3x3_fwd: act %d", act);
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(w, 64), eavsr::cdiv(h, 4), n * cout), block(64, 4, 1);
  hipLaunchKernelGGL(gconv3x3_fwd_kernel, grid, block, 0, eavsr::as_stream(stream), x, weight, bias, out, cout, cpg, h, w,
                     slope, act);
  return eavsr::launch_status("gconv3x3_fwd");
}

extern "C" int eavsr_gconv3x3_bwd_acc_f32(const float* g, const float* x, const float* weight, float* dx, float* dweight,
                                          float* dbias, int32_t n, int32_t cout, int32_t cpg, int32_t h, int32_t w,
                                          int32_t accumulate, void* stream);
extern "C" int eavsr_gconv3x3_bwd_f32(const float* g, const float* x, const float* weight, float* dx, float* dweight,
                                      float* dbias, int32_t n, int32_t cout, int32_t cpg, int32_t h, int32_t w,
                                      void* stream) {
  return eavsr_gconv3x3_bwd_acc_f32(g, x, weight, dx, dweight, dbias, n, cout, cpg, h, w, 0, stream);
}

extern "C" int eavsr_gconv3x3_bwd_acc_f32(const float* g, const float* x, const float* weight, float* dx, float* dweight,
                                          float* dbias, int32_t n, int32_t cout, int32_t cpg, int32_t h, int32_t w,
                                          int32_t accumulate, void* stream) {
  EAVSR_REQUIRE(g && x && weight && dx && dweight, -1, "gconv3x3_bwd: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cout > 0 && cpg > 0 && h > 0 && w > 0 && (long)n * cout * cpg <= 65535, -1,
                "gconv3x3_bwd: bad dims");
  if (n == 0) return 0;
  hipStream_t st = eavsr::as_stream(stream);
  dim3 grid(eavsr::cdiv(w, 64), eavsr::cdiv(h, 4), n * cout * cpg), block(64, 4, 1);
  hipLaunchKernelGGL(gconv3x3_dgrad_kernel, grid, block, 0, st, g, weight, dx, cout, cpg, h, w);
  hipLaunchKernelGGL(gconv3x3_wgrad_kernel, dim3(cout * cpg), dim3(1024), 0, st, g, x, dweight, dbias, n, cout, cpg, h, w,
                     accumulate ? 1 : 0);
  return eavsr::launch_status("gconv3x3_bwd");
}
