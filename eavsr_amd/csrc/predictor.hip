// MultiAdaSTN offset / mask predictor pieces (a3, a6).
//
// (1) adapt_frontend: the `concat` + `concat2` stack shared by AdaptBlock2_3x3 (networks.py:327-328,
//     336) and AdaptBlockOffset (networks.py:290-291,300):
//        t1 = LeakyReLU_0.2(depthwise3x3(cat(x, h_hr)))      (2c channels)
//        t2 = LeakyReLU_0.2(conv3x3(t1, groups = c))         (c channels, 2 inputs each)
//     Output channel o depends only on cat-channels 2o and 2o+1, so one workgroup handles one
//     (tile, o): the two input channels come into LDS once with a 2-pixel halo, t1 is produced in LDS
//     with a 1-pixel halo (zero outside the image: t1 is zero-padded by the second conv, it is NOT
//     lrelu(bias) there), t2 goes straight to HBM.  HBM-bound: 2c floats in, c floats out per pixel,
//     no intermediate round trip (the reference writes/reads cat and t1: 5x the traffic).
//
// (2) affine_offsets: networks.py:302-311 / :338-346.  Per pixel and deformable group the 2x2 transform
//     T and the translation t become the 9 sampling offsets  T . R - R + t  with the regular grid R,
//     written in mmcv channel order g*18 + 2k + {0: y, 1: x}; the mask logits get their sigmoid
//     (networks.py:313-314).  Pure streaming.
#include "common.h"

namespace {

constexpr int FT_H = 16, FT_W = 64;  // output tile per workgroup
constexpr int FP_H = FT_H + 4, FP_W = FT_W + 4;
constexpr int FM_H = FT_H + 2, FM_W = FT_W + 2;

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : v * 0.2f; }

__global__ __launch_bounds__(256) void adapt_frontend_kernel(
    const float* __restrict__ x, const float* __restrict__ hh, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ out, int c, int h, int w, int tiles_x) {
  __shared__ float s_in[2][FP_H][FP_W];
  __shared__ float s_mid[2][FM_H][FM_W];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int o = blockIdx.y, bn = blockIdx.z;
  const int y0 = ty * FT_H, x0 = tx * FT_W;
  const size_t plane = (size_t)h * w;

  // issue every load of the patch first, then write LDS (a load / wait / write loop would serialise ~11 HBM
  // latencies per workgroup)
  constexpr int P_N = 2 * FP_H * FP_W, P_IT = (P_N + 255) / 256;
  float tin[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int e = tid + i * 256;
    const int ch = min(e / (FP_H * FP_W), 1);
    const int rem = e - ch * (FP_H * FP_W);
    const int r = rem / FP_W, cc = rem - r * FP_W;
    const int gy = y0 - 2 + r, gx = x0 - 2 + cc;
    const int cat_c = 2 * o + ch;
    const float* src = cat_c < c ? x : hh;
    const int sc = cat_c < c ? cat_c : cat_c - c;
    const bool ok = e < P_N && gy >= 0 && gy < h && gx >= 0 && gx < w;
    const int cgy = min(max(gy, 0), h - 1), cgx = min(max(gx, 0), w - 1);
    const float v = src[((size_t)bn * c + sc) * plane + (size_t)cgy * w + cgx];
    tin[i] = ok ? v : 0.f;
  }
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int e = tid + i * 256;
    if (i < P_IT - 1 || e < P_N) (&s_in[0][0][0])[e] = tin[i];
  }
  __syncthreads();
  for (int e = tid; e < 2 * FM_H * FM_W; e += 256) {
    const int ch = e / (FM_H * FM_W);
    const int rem = e - ch * (FM_H * FM_W);
    const int r = rem / FM_W, cc = rem - r * FM_W;
    const int gy = y0 - 1 + r, gx = x0 - 1 + cc;
    float v = 0.f;
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
      const float* wk = w1 + (size_t)(2 * o + ch) * 9;
      v = b1[2 * o + ch];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v += wk[ky * 3 + kx] * s_in[ch][r + ky][cc + kx];
      v = lrelu02(v);
    }
    s_mid[ch][r][cc] = v;
  }
  __syncthreads();
  const float* wk2 = w2 + (size_t)o * 18;
  const float bias2 = b2[o];
  for (int e = tid; e < FT_H * FT_W; e += 256) {
    const int r = e / FT_W, cc = e - r * FT_W;
    const int gy = y0 + r, gx = x0 + cc;
    if (gy < h && gx < w) {
      float v = bias2;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) v += wk2[ch * 9 + ky * 3 + kx] * s_mid[ch][r + ky][cc + kx];
      out[((size_t)bn * c + o) * plane + (size_t)gy * w + gx] = lrelu02(v);
    }
  }
}

// one thread per (n, g, pixel)
__global__ __launch_bounds__(256) void affine_offsets_kernel(const float* __restrict__ heads,
                                                             float* __restrict__ offset,
                                                             float* __restrict__ mask, int D, int hw,
                                                             int head_c) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int g = blockIdx.y, bn = blockIdx.z;
  const float* hp = heads + (size_t)bn * head_c * hw + p;
  const float t00 = hp[(size_t)(g * 4 + 0) * hw];
  const float t01 = hp[(size_t)(g * 4 + 1) * hw];
  const float t10 = hp[(size_t)(g * 4 + 2) * hw];
  const float t11 = hp[(size_t)(g * 4 + 3) * hw];
  const float try_ = hp[(size_t)(4 * D + g * 2 + 0) * hw];
  const float trx = hp[(size_t)(4 * D + g * 2 + 1) * hw];
  float* op = offset + ((size_t)bn * D * 18 + (size_t)g * 18) * hw + p;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float ry = (float)(k / 3 - 1), rx = (float)(k % 3 - 1);
    // (T . R)[:,k] - R[:,k] + t     (matmul then subtract then add, networks.py:304-311)
    const float oy = (t00 * ry + t01 * rx) - ry + try_;
    const float ox = (t10 * ry + t11 * rx) - rx + trx;
    op[(size_t)(2 * k) * hw] = oy;
    op[(size_t)(2 * k + 1) * hw] = ox;
  }
  if (mask != nullptr) {
    float* mp = mask + ((size_t)bn * D * 9 + (size_t)g * 9) * hw + p;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float z = hp[(size_t)(6 * D + g * 9 + k) * hw];
      mp[(size_t)k * hw] = 1.f / (1.f + expf(-z));
    }
  }
}

}  // namespace

extern "C" int eavsr_adapt_frontend_f32(const float* x, const float* h_hr, const float* w1, const float* b1,
                                        const float* w2, const float* b2, float* out, int32_t n, int32_t c,
                                        int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(x && h_hr && w1 && b1 && w2 && b2 && out, -1, "adapt_frontend: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0, -1, "adapt_frontend: bad dims");
  EAVSR_REQUIRE(c % 2 == 0, -2, "adapt_frontend: channel count %d must be even", c);
  EAVSR_REQUIRE(c <= 65535 && n <= 65535, -1, "adapt_frontend: grid too large");
  if (n == 0) return 0;
  const int tiles_x = eavsr::cdiv(w, FT_W), tiles_y = eavsr::cdiv(h, FT_H);
  dim3 grid(tiles_x * tiles_y, c, n);
  hipLaunchKernelGGL(adapt_frontend_kernel, grid, dim3(256), 0, eavsr::as_stream(stream), x, h_hr, w1, b1, w2,
                     b2, out, c, h, w, tiles_x);
  return eavsr::launch_status("adapt_frontend");
}

extern "C" int eavsr_affine_offsets_f32(const float* heads, float* offset, float* mask, int32_t n, int32_t D,
                                        int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(heads && offset, -1, "affine_offsets: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && D > 0 && h > 0 && w > 0, -1, "affine_offsets: bad dims");
  EAVSR_REQUIRE((long)h * w < (1L << 31) && D <= 65535 && n <= 65535, -1, "affine_offsets: too large");
  if (n == 0) return 0;
  const int hw = h * w;
  const int head_c = mask ? 15 * D : 6 * D;
  dim3 grid(eavsr::cdiv(hw, 256), D, n);
  hipLaunchKernelGGL(affine_offsets_kernel, grid, dim3(256), 0, eavsr::as_stream(stream), heads, offset, mask, D,
                     hw, head_c);
  return eavsr::launch_status("affine_offsets");
}
