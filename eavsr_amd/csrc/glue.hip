// Elementwise / resampling glue either side of the hot path (f1 SPyNet, f2 encoder + upsampling tail): the
// handful of ATen kernels that were still inside the timed forward in round 2 (avg_pool2d, upsample_bilinear2d, the
// normalisation's sub / div, cat) as plain HIP streaming kernels on the caller's stream.  Arithmetic follows the ATen
// kernels the reference reaches (same operation order), so the SPyNet flows keep their goldens.
#include "common.h"

namespace {

// (x - mean[c]) / std[c]: SPyNet.compute_flow eavsrp_model.py:436-437, ContrasExtractorLayer.forward networks.py:550
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ in, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, float* __restrict__ out, int c, int hw) {
  const int nc = blockIdx.y;
  const float m = mean[nc % c], s = stdv[nc % c];
  const size_t base = (size_t)nc * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) out[base + i] = (in[base + i] - m) / s;
}

// F.avg_pool2d(x, 2, 2, count_include_pad=False) on even h, w (eavsrp_model.py:450-462): ((a + b) + c) + d, then / 4, the
// accumulation order of ATen's avg_pool2d kernel
__global__ __launch_bounds__(256) void avg_pool2_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w) {
  const int wo = w >> 1, ho = h >> 1;
  const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y, nc = blockIdx.z;
  if (ox >= wo || oy >= ho) return;
  const float* p = in + (size_t)nc * h * w + (size_t)(2 * oy) * w + 2 * ox;
  const float v = ((p[0] + p[1]) + p[w]) + p[w + 1];
  out[(size_t)nc * ho * wo + (size_t)oy * wo + ox] = v / 4.f;
}

// F.interpolate(x, size, mode='bilinear', align_corners=False) (eavsrp_model.py:499-509 SPyNet, :158,359 the x4 / x2 skip):
// ATen's upsample_bilinear2d: scale = in / out, src = scale (dst + 0.5) - 0.5 clamped at 0, i0 = (int)src, i1 = i0 + (i0 < in - 1),
// l1 = src - i0, l0 = 1 - l1; value = h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11).  Channel ch of `cmul` channels is then
// multiplied by m0 (ch = 0) / m1 (ch = 1): the flow rescaling of eavsrp_model.py:519-521 (cmul = 0: none).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int hin,
                                                              int win, int hout, int wout, float rh, float rw, int cmul, float m0,
                                                              float m1) {
  const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y, nc = blockIdx.z;
  if (ox >= wout || oy >= hout) return;
  const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)sy, hin - 1), x0 = min((int)sx, win - 1);
  const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* p = in + (size_t)nc * hin * win;
  float v = ly0 * (lx0 * p[y0 * win + x0] + lx1 * p[y0 * win + x1]) + ly1 * (lx0 * p[y1 * win + x0] + lx1 * p[y1 * win + x1]);
  if (cmul > 0) {
    const int ch = nc % cmul;
    v *= ch == 0 ? m0 : ch == 1 ? m1 : 1.f;
  }
  out[(size_t)nc * hout * wout + (size_t)oy * wout + ox] = v;
}

// torch.cat([a, b, c], 1) of (n, ca | cb | cc, h, w) (eavsrp_model.py:486: the 8-channel input of a SPyNet level)
__global__ __launch_bounds__(256) void concat3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, float* __restrict__ out, int ca, int cb, int cc,
                                                      int hw) {
  const int ct = ca + cb + cc;
  const int nc = blockIdx.y, n = nc / ct, ch = nc - n * ct;
  const float* src = ch < ca ? a + ((size_t)n * ca + ch) * hw : ch < ca + cb ? b + ((size_t)n * cb + (ch - ca)) * hw
                                                                             : c + ((size_t)n * cc + (ch - ca - cb)) * hw;
  float* dst = out + (size_t)nc * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

extern "C" int eavsr_normalize_f32(const float* in, const float* mean, const float* stdv, float* out, int32_t n, int32_t c,
                                   int32_t hw, void* stream) {
  EAVSR_REQUIRE(in && mean && stdv && out, -1, "normalize: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && hw > 0 && (long)n * c <= 65535, -1, "normalize: bad dims");
  if (n == 0) return 0;
  int bx = eavsr::cdiv(hw, 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(normalize_kernel, dim3(bx, n * c), dim3(256), 0, eavsr::as_stream(stream), in, mean, stdv, out, c, hw);
  return eavsr::launch_status("normalize");
}

extern "C" int eavsr_avg_pool2_f32(const float* in, float* out, int32_t nc, int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(in && out, -1, "avg_pool2: NULL pointer");
  EAVSR_REQUIRE(nc >= 0 && h > 0 && w > 0 && nc <= 65535, -1, "avg_pool2: bad dims");
  EAVSR_REQUIRE(h % 2 == 0 && w % 2 == 0, -2, "avg_pool2: h=%d w=%d must be even", h, w);
  if (nc == 0) return 0;
  dim3 grid(eavsr::cdiv(w / 2, 64), eavsr::cdiv(h / 2, 4), nc), block(64, 4, 1);
  hipLaunchKernelGGL(avg_pool2_kernel, grid, block, 0, eavsr::as_stream(stream), in, out, h, w);
  return eavsr::launch_status("avg_pool2");
}

extern "C" int eavsr_resize_bilinear_f32(const float* in, float* out, int32_t n, int32_t c, int32_t hin, int32_t win,
                                         int32_t hout, int32_t wout, int32_t cmul, float m0, float m1, void* stream) {
  EAVSR_REQUIRE(in && out, -1, "resize_bilinear: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && hin > 0 && win > 0 && hout > 0 && wout > 0 && (long)n * c <= 65535, -1, "resize_bilinear: bad dims");
  EAVSR_REQUIRE(cmul == 0 || cmul == c, -1, "resize_bilinear: cmul %d must be 0 or the channel count %d", cmul, c);
  if (n * c == 0) return 0;
  const float rh = (float)hin / (float)hout, rw = (float)win / (float)wout;      // ATen: area_pixel_compute_scale without a scale factor
  dim3 grid(eavsr::cdiv(wout, 64), eavsr::cdiv(hout, 4), n * c), block(64, 4, 1);
  hipLaunchKernelGGL(resize_bilinear_kernel, grid, block, 0, eavsr::as_stream(stream), in, out, hin, win, hout, wout, rh, rw, cmul,
                     m0, m1);
  return eavsr::launch_status("resize_bilinear");
}

extern "C" int eavsr_concat3_f32(const float* a, int32_t ca, const float* b, int32_t cb, const float* c, int32_t cc, float* out,
                                 int32_t n, int32_t hw, void* stream) {
  EAVSR_REQUIRE(a && b && c && out, -1, "concat3: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && ca > 0 && cb > 0 && cc > 0 && hw > 0 && (long)n * (ca + cb + cc) <= 65535, -1, "concat3: bad dims");
  if (n == 0) return 0;
  int bx = eavsr::cdiv(hw, 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(concat3_kernel, dim3(bx, n * (ca + cb + cc)), dim3(256), 0, eavsr::as_stream(stream), a, b, c, out, ca, cb,
                     cc, hw);
  return eavsr::launch_status("concat3");
}
