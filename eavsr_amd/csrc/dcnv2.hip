// Fused modulated deformable convolution (DCNv2) forward (a7).
//
// Reference: mmcv.ops.modulated_deform_conv2d called at models/networks.py:627-630 with the
// ModulatedDeformConv2d parameters of networks.py:575-583 (64 -> 64, 3x3, stride 1, pad 1,
// dilation 1, groups 1, deform_groups 8).  mmcv runs a per-sample im2col into a (cin*9, h*w) column
// buffer in HBM (132.7 MB at 180x320), an SGEMM and a bias pass.  Here the column tile never leaves
// the CU: per 4x32-pixel tile and per 8-channel chunk the sampler writes col[c][tap][px] into LDS and
// the 64 x (8*9) x 128 contraction runs on v_mfma_f32_32x32x2_f32 straight from LDS.
//
// Semantics restated from the published mmcv 1.x algorithm (modulated_deformable_im2col):
//   tap k = 3 i + j samples at p = (y - 1 + i + dy, x - 1 + j + dx), dy = offset[g*18 + 2k],
//   dx = offset[g*18 + 2k + 1], g = c / (cin / dg); the sample is taken iff -1 < p_y < h and
//   -1 < p_x < w and is a corner-wise zero-padded bilinear interpolation
//   (v = hh*hw*v1 + hh*lw*v2 + lh*hw*v3 + lh*lw*v4), col = v * mask[g*9 + k], out = W . col + b.
//
// HBM traffic per pixel (fp32, cin = cout = 64, dg = 8): 64 + 144 + 72 + 64 floats = 1376 B; the input
// patch is re-read through L1/L2 by the 9 taps (data-dependent gathers, neighbouring lanes hit the
// same lines).  The per-(pixel, tap) corner indices / weights are computed once and reused by the 8
// channels of the deformable group.
#include "common.h"

namespace {

struct DcnArgs {
  const float* x;
  const float* offset;
  const float* mask;
  const float* wp;
  const float* bias;
  float* out;
  int n, cin, h, w, cout, dg, cpg, cin_pad, tiles_x, tiles_y;
};

constexpr int DT_H = 4, DT_W = 32, DT_PX = DT_H * DT_W;  // 128 pixels per workgroup, one row per wave
constexpr int DCK = 8;                                   // channels per chunk
constexpr int DKK = 9;

template <int MT>
__global__ __launch_bounds__(256, 2) void dcnv2_kernel(DcnArgs a) {
  constexpr int CO = 32 * MT;
  __shared__ float s_col[DCK * DKK * DT_PX];                             // 36,864 B
  __shared__ __attribute__((aligned(16))) float s_w[DCK * DKK * CO];    // 18,432 B (MT = 2)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * DT_H, x0 = tx * DT_W;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const float* bcol = s_col + half * (DKK * DT_PX) + wave * 32 + l31;
  const float* acol = s_w + half * (DKK * CO) + l31;

  for (int c0 = 0; c0 < a.cin; c0 += DCK) {
    const int g = c0 / a.cpg;
    __syncthreads();
    // ---- sampler: col[c][tap][px] for the 8 channels of this chunk ---------------------------
    for (int item = tid; item < DKK * DT_PX; item += 256) {
      const int tap = item >> 7;  // wave-uniform
      const int p = item & (DT_PX - 1);
      const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
      float vals[DCK];
#pragma unroll
      for (int c = 0; c < DCK; ++c) vals[c] = 0.f;
      if (gy < h && gx < w) {
        const size_t pix = (size_t)gy * w + gx;
        const float* offp = a.offset + ((size_t)bn * a.dg * 18 + (size_t)g * 18 + 2 * tap) * plane + pix;
        const float oy = offp[0];
        const float ox = offp[plane];
        const float mk = a.mask[((size_t)bn * a.dg * 9 + (size_t)g * 9 + tap) * plane + pix];
        const int ti = tap / 3, tj = tap - 3 * ti;
        const float py = (float)(gy - 1 + ti) + oy;
        const float px = (float)(gx - 1 + tj) + ox;
        if (py > -1.f && px > -1.f && py < (float)h && px < (float)w) {
          const float fy0 = floorf(py), fx0 = floorf(px);
          const int hl = (int)fy0, wl = (int)fx0;
          const int hh_i = hl + 1, wh_i = wl + 1;
          const float lh = py - fy0, lw = px - fx0;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const bool t_ok = hl >= 0, b_ok = hh_i <= h - 1, l_ok = wl >= 0, r_ok = wh_i <= w - 1;
          const float w1 = (t_ok & l_ok) ? hh * hw : 0.f;
          const float w2 = (t_ok & r_ok) ? hh * lw : 0.f;
          const float w3 = (b_ok & l_ok) ? lh * hw : 0.f;
          const float w4 = (b_ok & r_ok) ? lh * lw : 0.f;
          const int cy0 = max(hl, 0), cy1 = min(hh_i, h - 1);
          const int cx0 = max(wl, 0), cx1 = min(wh_i, w - 1);
          const int i1 = cy0 * w + cx0, i2 = cy0 * w + cx1, i3 = cy1 * w + cx0, i4 = cy1 * w + cx1;
          const float* xp = a.x + ((size_t)bn * a.cin + c0) * plane;
#pragma unroll
          for (int c = 0; c < DCK; ++c) {
            const float* q = xp + (size_t)c * plane;
            float v = w1 * q[i1];
            v += w2 * q[i2];
            v += w3 * q[i3];
            v += w4 * q[i4];
            vals[c] = v * mk;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < DCK; ++c) s_col[(c * DKK + tap) * DT_PX + p] = vals[c];
    }
    // ---- weight slab of this chunk -----------------------------------------------------------
    {
      const f32x4* wsrc =
          reinterpret_cast<const f32x4*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)c0) * (DKK * CO));
      f32x4* wdst = reinterpret_cast<f32x4*>(s_w);
      for (int e = tid; e < DCK * DKK * CO / 4; e += 256) wdst[e] = wsrc[e];
    }
    __syncthreads();
    // ---- contraction: 36 k-steps of (channel pair, tap) ----------------------------------------
#pragma unroll
    for (int tap = 0; tap < DKK; ++tap) {
#pragma unroll
      for (int cp = 0; cp < DCK / 2; ++cp) {
        const float b = bcol[(cp * 2 * DKK + tap) * DT_PX];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float av = acol[(cp * 2 * DKK + tap) * CO + m * 32];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[m], 0, 0, 0);
        }
      }
    }
  }

  const int gy = y0 + wave, gx = x0 + l31;
  if (gy < h && gx < w) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.cout) {
          const float b = a.bias ? a.bias[co] : 0.f;
          a.out[((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx] = acc[m][r] + b;
        }
      }
  }
}

}  // namespace

extern "C" int eavsr_dcnv2_f32(const float* x, const float* offset, const float* mask,
                               const float* weight_packed, const float* bias, float* out, int32_t n,
                               int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t deform_groups,
                               void* stream) {
  EAVSR_REQUIRE(x && offset && mask && weight_packed && out, -1, "dcnv2: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && deform_groups > 0, -1, "dcnv2: bad dims");
  EAVSR_REQUIRE(cin % deform_groups == 0, -1, "dcnv2: cin %d not divisible by deform_groups %d", cin,
                deform_groups);
  const int cpg = cin / deform_groups;
  EAVSR_REQUIRE(cpg % DCK == 0, -2,
                "dcnv2: %d channels per deformable group unsupported (must be a multiple of 8; the reference "
                "uses 64 channels / 8 groups)", cpg);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "dcnv2: plane too large");
  if (n == 0) return 0;
  DcnArgs a;
  a.x = x; a.offset = offset; a.mask = mask; a.wp = weight_packed; a.bias = bias; a.out = out;
  a.n = n; a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.dg = deform_groups; a.cpg = cpg;
  a.cin_pad = cin;  // multiple of 8 == the 3x3 packing chunk
  a.tiles_x = eavsr::cdiv(w, DT_W);
  a.tiles_y = eavsr::cdiv(h, DT_H);
  const long blocks = (long)a.tiles_x * a.tiles_y * n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "dcnv2: too many tiles");
  const int CO = cout <= 32 ? 32 : 64;
  dim3 grid((unsigned)blocks, eavsr::cdiv(cout, CO));
  if (CO == 32)
    hipLaunchKernelGGL(dcnv2_kernel<1>, grid, dim3(256), 0, eavsr::as_stream(stream), a);
  else
    hipLaunchKernelGGL(dcnv2_kernel<2>, grid, dim3(256), 0, eavsr::as_stream(stream), a);
  return eavsr::launch_status("dcnv2");
}
