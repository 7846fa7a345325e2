// Modulated deformable convolution for the configurations OFF the reference's path (any kernel size, stride,
// padding, dilation, conv groups and deformable groups of mmcv.ops.modulated_deform_conv2d's signature,
// networks.py:575-583).  The reference itself only ever runs 3x3 / stride 1 / pad 1 / dilation 1 / groups 1 /
// 8 deformable groups (eavsrp_model.py:143) -- that is eavsr_dcnv2_f32.  This kernel exists so that the drop-in
// accepts the whole signature (e.g. MultiAdSTN's own default deformable_groups=64, networks.py:576); it is a
// straightforward one-thread-per-output-pixel kernel (8 output channels in registers, sampling positions shared
// by the channels of a deformable group), not a tuned one.  Forward only.
#include "common.h"

namespace {

struct GArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* weight;   // (cout, cin / groups, kh, kw), original layout
  const float* bias;
  float* out;
  int n, cin, h, w, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, ho, wo;
};

constexpr int GCO = 8;  // output channels per thread

__global__ __launch_bounds__(128) void dcnv2_generic_kernel(GArgs a) {
  const int px = blockIdx.x * 128 + threadIdx.x;
  const int howo = a.ho * a.wo;
  if (px >= howo) return;
  const int co0 = blockIdx.y * GCO, bn = blockIdx.z;
  const int oy = px / a.wo, ox = px - oy * a.wo;
  const int K = a.kh * a.kw;
  const int cin_g = a.cin / a.groups, cout_g = a.cout / a.groups;
  const int grp = co0 / cout_g;                 // conv group of this thread's output channels (GCO divides cout_g)
  const int cpg = a.cin / a.dg;                 // channels per deformable group
  const size_t plane = (size_t)a.h * a.w;
  float acc[GCO];
#pragma unroll
  for (int j = 0; j < GCO; ++j) acc[j] = 0.f;
  const float* xb = a.x + (size_t)bn * a.cin * plane;
  const float* offb = a.offset + (size_t)bn * a.dg * 2 * K * howo + px;
  const float* mkb = a.mask + (size_t)bn * a.dg * K * howo + px;
  for (int k = 0; k < K; ++k) {
    const int ki = k / a.kw, kj = k - ki * a.kw;
    int cur_g = -1;
    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mcur = 0.f;
    int i1 = 0, i2 = 0, i3 = 0, i4 = 0;
    for (int cl = 0; cl < cin_g; ++cl) {
      const int c = grp * cin_g + cl;
      const int g = c / cpg;
      if (g != cur_g) {   // new deformable group: sampling position of this tap
        cur_g = g;
        const float py = (float)(oy * a.sh - a.ph + ki * a.dh) + offb[(size_t)(g * 2 * K + 2 * k) * howo];
        const float pxf = (float)(ox * a.sw - a.pw + kj * a.dw) + offb[(size_t)(g * 2 * K + 2 * k + 1) * howo];
        const bool in = py > -1.f && pxf > -1.f && py < (float)a.h && pxf < (float)a.w;
        const float m = in ? mkb[(size_t)(g * K + k) * howo] : 0.f;
        const float fy0 = floorf(py), fx0 = floorf(pxf);
        const float lh = py - fy0, lw = pxf - fx0, hh = 1.f - lh, hw = 1.f - lw;
        const int hl = (int)fminf(fmaxf(fy0, -2.f), (float)a.h), wl = (int)fminf(fmaxf(fx0, -2.f), (float)a.w);
        const int hh_i = hl + 1, wh_i = wl + 1;
        const bool t_ok = hl >= 0, b_ok = hh_i <= a.h - 1, l_ok = wl >= 0, r_ok = wh_i <= a.w - 1;
        w1 = (in & t_ok & l_ok) ? hh * hw : 0.f;
        w2 = (in & t_ok & r_ok) ? hh * lw : 0.f;
        w3 = (in & b_ok & l_ok) ? lh * hw : 0.f;
        w4 = (in & b_ok & r_ok) ? lh * lw : 0.f;
        const int cy0 = min(max(hl, 0), a.h - 1), cy1 = min(max(hh_i, 0), a.h - 1);
        const int cx0 = min(max(wl, 0), a.w - 1), cx1 = min(max(wh_i, 0), a.w - 1);
        i1 = cy0 * a.w + cx0; i2 = cy0 * a.w + cx1; i3 = cy1 * a.w + cx0; i4 = cy1 * a.w + cx1;
        mcur = m;   // col = (bilinear sample) * mask, in this order
      }
      const float* p = xb + (size_t)c * plane;
      float v = w1 * p[i1];
      v += w2 * p[i2];
      v += w3 * p[i3];
      v += w4 * p[i4];
      v *= mcur;
      const float* wp = a.weight + ((size_t)co0 * cin_g + cl) * K + k;
#pragma unroll
      for (int j = 0; j < GCO; ++j)
        if (co0 + j < a.cout) acc[j] += wp[(size_t)j * cin_g * K] * v;
    }
  }
#pragma unroll
  for (int j = 0; j < GCO; ++j) {
    const int co = co0 + j;
    if (co < a.cout) a.out[((size_t)bn * a.cout + co) * howo + px] = acc[j] + (a.bias ? a.bias[co] : 0.f);
  }
}

}  // namespace

extern "C" int eavsr_dcnv2_generic_f32(const float* x, const float* offset, const float* mask, const float* weight,
                                       const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w,
                                       int32_t cout, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w,
                                       int32_t pad_h, int32_t pad_w, int32_t dil_h, int32_t dil_w, int32_t groups,
                                       int32_t deform_groups, void* stream) {
  EAVSR_REQUIRE(x && offset && mask && weight && out, -1, "dcnv2_generic: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && kh > 0 && kw > 0, -1, "dcnv2_generic: bad dims");
  EAVSR_REQUIRE(stride_h > 0 && stride_w > 0 && dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0, -1,
                "dcnv2_generic: bad stride / dilation / padding");
  EAVSR_REQUIRE(groups > 0 && cin % groups == 0 && cout % groups == 0, -1, "dcnv2_generic: groups %d", groups);
  EAVSR_REQUIRE(deform_groups > 0 && cin % deform_groups == 0, -1, "dcnv2_generic: deform_groups %d", deform_groups);
  EAVSR_REQUIRE(groups == 1 || (cout / groups) % GCO == 0, -2,
                "dcnv2_generic: with conv groups > 1 the output channels per group must be a multiple of %d", GCO);
  GArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.weight = weight; a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.kh = kh; a.kw = kw;
  a.sh = stride_h; a.sw = stride_w; a.ph = pad_h; a.pw = pad_w; a.dh = dil_h; a.dw = dil_w;
  a.groups = groups; a.dg = deform_groups;
  a.ho = (h + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  a.wo = (w + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  EAVSR_REQUIRE(a.ho > 0 && a.wo > 0, -1, "dcnv2_generic: empty output");
  EAVSR_REQUIRE((long)h * w < (1L << 31) && (long)a.ho * a.wo < (1L << 31), -1, "dcnv2_generic: plane too large");
  EAVSR_REQUIRE(n <= 65535 && eavsr::cdiv(cout, GCO) <= 65535, -1, "dcnv2_generic: grid too large");
  if (n == 0) return 0;
  dim3 grid(eavsr::cdiv(a.ho * a.wo, 128), eavsr::cdiv(cout, GCO), n);
  hipLaunchKernelGGL(dcnv2_generic_kernel, grid, dim3(128), 0, eavsr::as_stream(stream), a);
  return eavsr::launch_status("dcnv2_generic");
}
