// Resampling / elementwise glue of MultiAdSTN.forward and EAVSRP.forward (a5, a12).
// All of these are tiny streaming kernels; they exist so that the whole alignment step stays on the
// caller's stream without host-side meshgrids, H2D copies or library round trips.
#include "common.h"

namespace {

// F.interpolate(mode='bilinear', align_corners=True) as at networks.py:600-601,608,613, with the
// reference's pre-add (offset_p2 + offset_p1_up2), scale factor (/4, /2, *2) and post-add
// (offset_down2 + offset_p1_up2) folded in.  Index arithmetic follows ATen's upsample_bilinear2d:
// ratio = (in - 1) / (out - 1), src = ratio * dst, i0 = (int)src, lambda1 = src - i0.
__global__ __launch_bounds__(256) void resize_bilinear_ac_kernel(
    const float* __restrict__ in, const float* __restrict__ in2, const float* __restrict__ addend,
    float* __restrict__ out, int hin, int win, int hout, int wout, float rh, float rw, float scale) {
  const int ox = blockIdx.x * 64 + threadIdx.x;
  const int oy = blockIdx.y * 4 + threadIdx.y;
  const int nc = blockIdx.z;
  if (ox >= wout || oy >= hout) return;
  const float sy = rh * (float)oy, sx = rw * (float)ox;
  const int y0 = min((int)sy, hin - 1), x0 = min((int)sx, win - 1);
  const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const size_t base = (size_t)nc * hin * win;
  const float* p = in + base;
  float v00 = p[y0 * win + x0], v01 = p[y0 * win + x1], v10 = p[y1 * win + x0], v11 = p[y1 * win + x1];
  if (in2 != nullptr) {
    const float* q = in2 + base;
    v00 += q[y0 * win + x0];
    v01 += q[y0 * win + x1];
    v10 += q[y1 * win + x0];
    v11 += q[y1 * win + x1];
  }
  float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  v *= scale;
  const size_t o = (size_t)nc * hout * wout + (size_t)oy * wout + ox;
  if (addend != nullptr) v += addend[o];
  out[o] = v;
}

// eavsrp_model.py:218-220: F.interpolate(scale_factor=0.5 / 0.25, bilinear, align_corners=False)
// on an (h,w) divisible by 4: src = (dst + 0.5) / s - 0.5 lands exactly half way between two pixels,
// so down2 = mean of the 2x2 block at (2d, 2d+1) and down4 = mean of the 2x2 block at (4d+1, 4d+2).
// One thread per down4 pixel: reads a 4x4 input block, writes four down2 values and one down4 value.
__global__ __launch_bounds__(256) void pyramid_kernel(const float* __restrict__ in, float* __restrict__ d2,
                                                      float* __restrict__ d4, int h, int w) {
  const int h4 = h >> 2, w4 = w >> 2;
  const int x4 = blockIdx.x * 64 + threadIdx.x;
  const int y4 = blockIdx.y * 4 + threadIdx.y;
  const int nc = blockIdx.z;
  if (x4 >= w4 || y4 >= h4) return;
  const float* p = in + (size_t)nc * h * w + (size_t)(4 * y4) * w + 4 * x4;
  float v[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 row = *reinterpret_cast<const f32x4*>(p + (size_t)r * w);  // w % 4 == 0 -> 16-B aligned
    v[r][0] = row[0]; v[r][1] = row[1]; v[r][2] = row[2]; v[r][3] = row[3];
  }
  const int w2 = w >> 1;
  float* q = d2 + (size_t)nc * (h >> 1) * w2 + (size_t)(2 * y4) * w2 + 2 * x4;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
      q[(size_t)r * w2 + c] = 0.5f * (0.5f * v[2 * r][2 * c] + 0.5f * v[2 * r][2 * c + 1]) +
                              0.5f * (0.5f * v[2 * r + 1][2 * c] + 0.5f * v[2 * r + 1][2 * c + 1]);
  d4[(size_t)nc * h4 * w4 + (size_t)y4 * w4 + x4] =
      0.5f * (0.5f * v[1][1] + 0.5f * v[1][2]) + 0.5f * (0.5f * v[2][1] + 0.5f * v[2][2]);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  const float* __restrict__ c, float* __restrict__ out,
                                                  long count) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    float v = a[i] + b[i];
    if (c != nullptr) v += c[i];
    out[i] = v;
  }
}

}  // namespace

extern "C" int eavsr_resize_bilinear_ac_f32(const float* in, const float* in2, const float* addend, float* out,
                                            int32_t n, int32_t c, int32_t hin, int32_t win, int32_t hout,
                                            int32_t wout, float scale, void* stream) {
  EAVSR_REQUIRE(in && out, -1, "resize_bilinear_ac: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, -1, "resize_bilinear_ac: bad dims");
  EAVSR_REQUIRE((long)n * c <= 65535, -1, "resize_bilinear_ac: n*c too large");
  if (n * c == 0) return 0;
  const float rh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
  const float rw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
  dim3 grid(eavsr::cdiv(wout, 64), eavsr::cdiv(hout, 4), n * c), block(64, 4, 1);
  hipLaunchKernelGGL(resize_bilinear_ac_kernel, grid, block, 0, eavsr::as_stream(stream), in, in2, addend, out,
                     hin, win, hout, wout, rh, rw, scale);
  return eavsr::launch_status("resize_bilinear_ac");
}

extern "C" int eavsr_pyramid_f32(const float* in, float* down2, float* down4, int32_t nc, int32_t h, int32_t w,
                                 void* stream) {
  EAVSR_REQUIRE(in && down2 && down4, -1, "pyramid: NULL pointer");
  EAVSR_REQUIRE(nc >= 0 && h > 0 && w > 0, -1, "pyramid: bad dims");
  EAVSR_REQUIRE(h % 4 == 0 && w % 4 == 0, -2,
                "pyramid: h=%d w=%d must be divisible by 4 (the reference's .view at eavsrp_model.py:223-224 "
                "requires it)", h, w);
  EAVSR_REQUIRE(nc <= 65535 * 64, -1, "pyramid: n*c too large");
  EAVSR_REQUIRE(((uintptr_t)in & 15) == 0, -1, "pyramid: input must be 16-byte aligned");
  if (nc == 0) return 0;
  // grid.z is limited to 65535: fold the surplus into repeated launches
  const int per = 65535;
  for (int z0 = 0; z0 < nc; z0 += per) {
    const int zn = (nc - z0) < per ? (nc - z0) : per;
    dim3 grid(eavsr::cdiv(w / 4, 64), eavsr::cdiv(h / 4, 4), zn), block(64, 4, 1);
    hipLaunchKernelGGL(pyramid_kernel, grid, block, 0, eavsr::as_stream(stream), in + (size_t)z0 * h * w,
                       down2 + (size_t)z0 * (h / 2) * (w / 2), down4 + (size_t)z0 * (h / 4) * (w / 4), h, w);
  }
  return eavsr::launch_status("pyramid");
}

extern "C" int eavsr_add_f32(const float* a, const float* b, const float* c, float* out, int64_t count,
                             void* stream) {
  EAVSR_REQUIRE(a && b && out, -1, "add: NULL pointer");
  EAVSR_REQUIRE(count >= 0, -1, "add: negative count");
  if (count == 0) return 0;
  long blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, eavsr::as_stream(stream), a, b, c, out,
                     (long)count);
  return eavsr::launch_status("add");
}
