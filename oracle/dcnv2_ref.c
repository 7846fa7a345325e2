/*
 * oracle/dcnv2_ref.c -- plain-C restatement of the two samplers of the EAVSR hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/eavsr_oracle.py): loaded by tests/ and by
 * __graft_entry__.smoke() as a checker, never by the product.
 *
 *  - eavsr_ref_dcnv2: modulated deformable convolution as consumed at
 *    /root/reference/models/networks.py:627-630.  The arithmetic lives in mmcv-full 1.x
 *    (mmcv/ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh, version not pinned by the
 *    reference and absent from /root/reference): this follows the published algorithm --
 *    modulated_deformable_im2col (+ dmcn_im2col_bilinear) then W . col + b.  PARITY UNPINNED against
 *    the mmcv binary; anchored on the call site, the channel layout AdaptBlockOffset emits
 *    (networks.py:286-288,303-315) and known-answer identities (tests/test_oracle_dcn.py).
 *  - eavsr_ref_flow_warp: networks.py:699-739 -> torch grid_sample(bilinear, align_corners=True,
 *    zeros | border) restated with explicit coordinate arithmetic.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

static float bilinear_zero(const float* im, int h, int w, float y, float x) {
  int y0 = (int)floorf(y), x0 = (int)floorf(x);
  int y1 = y0 + 1, x1 = x0 + 1;
  float ly = y - (float)y0, lx = x - (float)x0;
  float hy = 1.f - ly, hx = 1.f - lx;
  float v1 = (y0 >= 0 && x0 >= 0) ? im[y0 * w + x0] : 0.f;
  float v2 = (y0 >= 0 && x1 <= w - 1) ? im[y0 * w + x1] : 0.f;
  float v3 = (y1 <= h - 1 && x0 >= 0) ? im[y1 * w + x0] : 0.f;
  float v4 = (y1 <= h - 1 && x1 <= w - 1) ? im[y1 * w + x1] : 0.f;
  return hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
}

/* x (n,c,h,w); offset (n,dg*2*k*k,ho,wo); mask (n,dg*k*k,ho,wo); weight (co,c,k,k); bias (co) or NULL;
 * out (n,co,ho,wo).  groups = 1. */
int eavsr_ref_dcnv2(const float* x, const float* offset, const float* mask, const float* weight,
                    const float* bias, float* out, int n, int c, int h, int w, int co, int k,
                    int stride, int pad, int dil, int dg) {
  const int ho = (h + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int wo = (w + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int K = k * k, cpg = c / dg;
  float* col = (float*)malloc(sizeof(float) * (size_t)c * K);
  if (!col) return -1;
  for (int b = 0; b < n; ++b)
    for (int oy = 0; oy < ho; ++oy)
      for (int ox = 0; ox < wo; ++ox) {
        for (int ch = 0; ch < c; ++ch) {
          const int g = ch / cpg;
          const float* im = x + ((size_t)b * c + ch) * h * w;
          for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
              const int t = i * k + j;
              const float dy = offset[(((size_t)b * dg + g) * 2 * K + 2 * t) * ho * wo + (size_t)oy * wo + ox];
              const float dx = offset[(((size_t)b * dg + g) * 2 * K + 2 * t + 1) * ho * wo + (size_t)oy * wo + ox];
              const float m = mask[(((size_t)b * dg + g) * K + t) * ho * wo + (size_t)oy * wo + ox];
              const float py = (float)(oy * stride - pad + i * dil) + dy;
              const float px = (float)(ox * stride - pad + j * dil) + dx;
              float v = 0.f;
              if (py > -1 && px > -1 && py < h && px < w) v = bilinear_zero(im, h, w, py, px);
              col[ch * K + t] = v * m;
            }
        }
        for (int o = 0; o < co; ++o) {
          double acc = bias ? bias[o] : 0.0; /* double accumulation: a tighter yardstick than fp32 */
          const float* wr = weight + (size_t)o * c * K;
          for (int q = 0; q < c * K; ++q) acc += (double)wr[q] * (double)col[q];
          out[(((size_t)b * co + o) * ho + oy) * wo + ox] = (float)acc;
        }
      }
  free(col);
  return 0;
}

/* flow NCHW (n,2,h,w): ch0 = x displacement, ch1 = y; border != 0 -> padding_mode='border' */
int eavsr_ref_flow_warp(const float* x, const float* flow, float* out, int n, int c, int h, int w, int border) {
  for (int b = 0; b < n; ++b)
    for (int y = 0; y < h; ++y)
      for (int xx = 0; xx < w; ++xx) {
        float px = (float)xx + flow[((size_t)b * 2 + 0) * h * w + (size_t)y * w + xx];
        float py = (float)y + flow[((size_t)b * 2 + 1) * h * w + (size_t)y * w + xx];
        if (border) {
          px = fminf((float)(w - 1), fmaxf(px, 0.f));
          py = fminf((float)(h - 1), fmaxf(py, 0.f));
        }
        for (int ch = 0; ch < c; ++ch) {
          const float* im = x + ((size_t)b * c + ch) * h * w;
          float v = 0.f;
          if (px > -1 && py > -1 && px < w && py < h) v = bilinear_zero(im, h, w, py, px);
          out[((size_t)b * c + ch) * h * w + (size_t)y * w + xx] = v;
        }
      }
  return 0;
}
