// flow_warp: bilinear warp of a feature map by a pixel-unit flow (a1, a2).
//
// Reference: models/networks.py:699-739 (flow NCHW) and models/eavsrp_model.py:587-626 (flow
// NHWC) -> torch.nn.functional.grid_sample(mode='bilinear', align_corners=True,
// padding_mode in {'zeros','border'}).  The reference builds a meshgrid on the host, adds the
// flow, normalises to [-1,1] and lets grid_sample un-normalise again; here the coordinate
// arithmetic is done in registers with the SAME sequence of fp32 operations so the sampling
// positions round identically, and nothing but x, flow and out touches HBM.
//
// HBM-bound: (c + 2 + c) * 4 bytes per pixel.  One thread per pixel column (64 lanes along W so
// every load/store row segment is coalesced), 16 channels per thread; the 4 corner weights and
// indices are computed once per pixel and reused across channels.
#include "common.h"

namespace {

constexpr int kChanPerThread = 16;

template <int PADMODE>
__global__ __launch_bounds__(256) void flow_warp_kernel(
    const float* __restrict__ x, const float* __restrict__ flow, const float* __restrict__ flow2,
    float* __restrict__ out, int n, int c, int h, int w, long fs_n, long fs_c, long fs_y, long fs_x,
    int c_chunks, int tiles_x, int tiles_y) {
  // linear tile id -> XCD-contiguous order: (x tile, y tile, channel chunk, sample), x fastest
  int lid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % tiles_x;
  lid /= tiles_x;
  const int ty = lid % tiles_y;
  lid /= tiles_y;
  const int c0 = (lid % c_chunks) * kChanPerThread;
  const int bn = lid / c_chunks;
  const int px = tx * 64 + threadIdx.x;
  const int py = ty * 4 + threadIdx.y;
  if (px >= w || py >= h) return;

  const long fo = (long)bn * fs_n + (long)py * fs_y + (long)px * fs_x;
  float fx = flow[fo];
  float fy = flow[fo + fs_c];
  if (flow2 != nullptr) {
    fx += flow2[fo];
    fy += flow2[fo + fs_c];
  }
  // reference: grid + flow, then 2*g/max(size-1,1) - 1   (networks.py:727-731)
  const float gx = (float)px + fx;
  const float gy = (float)py + fy;
  const float nx = 2.0f * gx / (float)max(w - 1, 1) - 1.0f;
  const float ny = 2.0f * gy / (float)max(h - 1, 1) - 1.0f;
  // grid_sample, align_corners=True: ((coord + 1) / 2) * (size - 1)
  float ix = ((nx + 1.0f) / 2.0f) * (float)(w - 1);
  float iy = ((ny + 1.0f) / 2.0f) * (float)(h - 1);
  if (PADMODE == EAVSR_PAD_BORDER) {
    ix = fminf((float)(w - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(h - 1), fmaxf(iy, 0.0f));
  }
  // keep the int conversion defined for wild flows (everything out of range reads as zero)
  ix = fminf(fmaxf(ix, -4.0f), (float)w + 4.0f);
  iy = fminf(fmaxf(iy, -4.0f), (float)h + 4.0f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x1 >= 0) & (x1 < w);
  const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y1 >= 0) & (y1 < h);
  const float w_nw = (vx0 & vy0) ? wx0 * wy0 : 0.0f;
  const float w_ne = (vx1 & vy0) ? wx1 * wy0 : 0.0f;
  const float w_sw = (vx0 & vy1) ? wx0 * wy1 : 0.0f;
  const float w_se = (vx1 & vy1) ? wx1 * wy1 : 0.0f;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  const int i_nw = cy0 * w + cx0, i_ne = cy0 * w + cx1, i_sw = cy1 * w + cx0, i_se = cy1 * w + cx1;

  const size_t plane = (size_t)h * w;
  const float* xp = x + ((size_t)bn * c + c0) * plane;
  float* op = out + ((size_t)bn * c + c0) * plane + (size_t)py * w + px;
  const int cend = min(kChanPerThread, c - c0);
  // One channel (4 gathers) in flight per thread: 46 vector registers instead of 62 with four -- slower on its own, but a wave
  // then fits on a SIMD BESIDE the two 232-register waves of a resident Winograd convolution workgroup of the other stream (512
  // registers per SIMD lane; above 48 the kernel only runs on convolution-free CUs).  Step 262.1 -> 260.3 ms (A/B, one box).
#pragma unroll 1
  for (int cc = 0; cc < cend; ++cc) {
    const float* p = xp + (size_t)cc * plane;
    float v = p[i_nw] * w_nw;
    v += p[i_ne] * w_ne;
    v += p[i_sw] * w_sw;
    v += p[i_se] * w_se;
    op[(size_t)cc * plane] = v;
  }
}

// Two feature maps warped by the SAME flow in one launch (networks.py:621 and :623: `nbr_feat_l[0]` and `feat_prop` are
// both warped by the refined offset): the flow is read and the corner weights are computed once per (pixel, channel
// chunk) as before, but one launch covers both tensors.  The second output may be written in the "IL8" layout
// [n][c/8][h][w][8] that eavsr_dcnv2_il_f32 samples from (the warp of feat_prop feeds DCNv2 and nothing else).
template <int b_il8>      // second output: 0 NCHW fp32, 1 IL8 fp32, 2 IL8 fp16, 3 IL8 bf16
__global__ __launch_bounds__(256) void flow_warp_pair_kernel(
    const float* __restrict__ xa, const float* __restrict__ xb, const float* __restrict__ flow,
    const float* __restrict__ flow2, float* __restrict__ outa, float* __restrict__ outb, int n, int c, int h, int w,
    int c_chunks, int tiles_x, int tiles_y) {
  int lid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % tiles_x;
  lid /= tiles_x;
  const int ty = lid % tiles_y;
  lid /= tiles_y;
  const int chunk = lid % (2 * c_chunks);
  const int bn = lid / (2 * c_chunks);
  const bool second = chunk >= c_chunks;
  const int c0 = (second ? chunk - c_chunks : chunk) * kChanPerThread;
  const int px = tx * 64 + threadIdx.x;
  const int py = ty * 4 + threadIdx.y;
  if (px >= w || py >= h) return;
  const size_t plane = (size_t)h * w;
  const size_t fo = (size_t)bn * 2 * plane + (size_t)py * w + px;
  float fx = flow[fo];
  float fy = flow[fo + plane];
  if (flow2 != nullptr) {
    fx += flow2[fo];
    fy += flow2[fo + plane];
  }
  // the reference's op sequence (networks.py:727-731 + grid_sample align_corners=True), as flow_warp_kernel
  const float gx = (float)px + fx;
  const float gy = (float)py + fy;
  const float nx = 2.0f * gx / (float)max(w - 1, 1) - 1.0f;
  const float ny = 2.0f * gy / (float)max(h - 1, 1) - 1.0f;
  float ix = ((nx + 1.0f) / 2.0f) * (float)(w - 1);
  float iy = ((ny + 1.0f) / 2.0f) * (float)(h - 1);
  ix = fminf(fmaxf(ix, -4.0f), (float)w + 4.0f);
  iy = fminf(fmaxf(iy, -4.0f), (float)h + 4.0f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x1 >= 0) & (x1 < w);
  const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y1 >= 0) & (y1 < h);
  const float w_nw = (vx0 & vy0) ? wx0 * wy0 : 0.0f;
  const float w_ne = (vx1 & vy0) ? wx1 * wy0 : 0.0f;
  const float w_sw = (vx0 & vy1) ? wx0 * wy1 : 0.0f;
  const float w_se = (vx1 & vy1) ? wx1 * wy1 : 0.0f;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  const int i_nw = cy0 * w + cx0, i_ne = cy0 * w + cx1, i_sw = cy1 * w + cx0, i_se = cy1 * w + cx1;

  const float* xp = (second ? xb : xa) + ((size_t)bn * c + c0) * plane;
  const int cend = min(kChanPerThread, c - c0);
  if (second && b_il8) {
    // c % 8 == 0 (checked on the host): this thread owns whole octets; 32 contiguous bytes per (pixel, octet)
    for (int o = 0; o < cend / 8; ++o) {
      // all 32 corner loads of the octet first, THEN the blends (a scheduling fence between them): in the 16-bit forms the
      // compiler otherwise reuses one temporary and waits for every load before issuing the next (2x the kernel's time)
      float l[8][4];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* p = xp + (size_t)(o * 8 + e) * plane;
        l[e][0] = p[i_nw]; l[e][1] = p[i_ne]; l[e][2] = p[i_sw]; l[e][3] = p[i_se];
      }
      __builtin_amdgcn_sched_barrier(0);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = l[e][0] * w_nw;
        t += l[e][1] * w_ne;
        t += l[e][2] * w_sw;
        t += l[e][3] * w_se;
        v[e] = t;
      }
      const size_t unit = (((size_t)bn * (c / 8) + (c0 / 8 + o)) * h + py) * w + px;
      if (b_il8 == 1) {
        f32x4* op = reinterpret_cast<f32x4*>(outb + unit * 8);
        op[0] = f32x4{v[0], v[1], v[2], v[3]};
        op[1] = f32x4{v[4], v[5], v[6], v[7]};
      } else {      // 2: fp16, 3: bf16 -- one 16-byte unit per (pixel, octet), each value rounded once
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        typedef float f2_ __attribute__((ext_vector_type(2)));
        u32x4_ pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (b_il8 == 3) {
            typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{v[2 * e], v[2 * e + 1]}, b2_));
          } else {
            typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
            pk[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{v[2 * e], v[2 * e + 1]}, h2_));
          }
        }
        reinterpret_cast<u32x4_*>(outb)[unit] = pk;
      }
    }
    return;
  }
  float* op = (second ? outb : outa) + ((size_t)bn * c + c0) * plane + (size_t)py * w + px;
#pragma unroll 1
  for (int cc = 0; cc < cend; ++cc) {
    const float* p = xp + (size_t)cc * plane;
    float v = p[i_nw] * w_nw;
    v += p[i_ne] * w_ne;
    v += p[i_sw] * w_sw;
    v += p[i_se] * w_se;
    op[(size_t)cc * plane] = v;
  }
}

// The other grid_sample modes the reference's signature admits (interpolation='nearest', padding_mode=
// 'reflection', align_corners=False; networks.py:699-739 passes them straight to F.grid_sample).  Off the hot path:
// one kernel with run-time switches, same thread mapping, coordinate arithmetic as aten's grid_sampler.
__device__ __forceinline__ float reflect_coord(float in, int twice_low, int twice_high) {
  if (twice_low == twice_high) return 0.f;
  const float mn = (float)twice_low / 2.f;
  const float span = (float)(twice_high - twice_low) / 2.f;
  in = fabsf(in - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  return (flips % 2 == 0) ? extra + mn : span - extra + mn;
}

__global__ __launch_bounds__(256) void flow_warp_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ flow, const float* __restrict__ flow2,
    float* __restrict__ out, int n, int c, int h, int w, long fs_n, long fs_c, long fs_y, long fs_x,
    int c_chunks, int tiles_x, int tiles_y, int padmode, int nearest, int align) {
  int lid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % tiles_x;
  lid /= tiles_x;
  const int ty = lid % tiles_y;
  lid /= tiles_y;
  const int c0 = (lid % c_chunks) * kChanPerThread;
  const int bn = lid / c_chunks;
  const int px = tx * 64 + threadIdx.x;
  const int py = ty * 4 + threadIdx.y;
  if (px >= w || py >= h) return;
  const long fo = (long)bn * fs_n + (long)py * fs_y + (long)px * fs_x;
  float fx = flow[fo], fy = flow[fo + fs_c];
  if (flow2 != nullptr) {
    fx += flow2[fo];
    fy += flow2[fo + fs_c];
  }
  const float nx = 2.0f * ((float)px + fx) / (float)max(w - 1, 1) - 1.0f;
  const float ny = 2.0f * ((float)py + fy) / (float)max(h - 1, 1) - 1.0f;
  float ix = align ? ((nx + 1.0f) / 2.0f) * (float)(w - 1) : ((nx + 1.0f) * (float)w - 1.0f) / 2.0f;
  float iy = align ? ((ny + 1.0f) / 2.0f) * (float)(h - 1) : ((ny + 1.0f) * (float)h - 1.0f) / 2.0f;
  if (padmode == EAVSR_PAD_REFLECTION) {
    ix = align ? reflect_coord(ix, 0, 2 * (w - 1)) : reflect_coord(ix, -1, 2 * w - 1);
    iy = align ? reflect_coord(iy, 0, 2 * (h - 1)) : reflect_coord(iy, -1, 2 * h - 1);
  }
  if (padmode != EAVSR_PAD_ZEROS) {
    ix = fminf((float)(w - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(h - 1), fmaxf(iy, 0.0f));
  }
  ix = fminf(fmaxf(ix, -4.0f), (float)w + 4.0f);
  iy = fminf(fmaxf(iy, -4.0f), (float)h + 4.0f);
  const size_t plane = (size_t)h * w;
  const float* xp = x + ((size_t)bn * c + c0) * plane;
  float* op = out + ((size_t)bn * c + c0) * plane + (size_t)py * w + px;
  const int cend = min(kChanPerThread, c - c0);
  if (nearest) {
    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);   // round half to even, as aten
    const bool ok = xn >= 0 && xn < w && yn >= 0 && yn < h;
    const int idx = min(max(yn, 0), h - 1) * w + min(max(xn, 0), w - 1);
    for (int cc = 0; cc < cend; ++cc) op[(size_t)cc * plane] = ok ? xp[(size_t)cc * plane + idx] : 0.f;
    return;
  }
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x1 >= 0) & (x1 < w);
  const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y1 >= 0) & (y1 < h);
  const float w_nw = (vx0 & vy0) ? wx0 * wy0 : 0.0f;
  const float w_ne = (vx1 & vy0) ? wx1 * wy0 : 0.0f;
  const float w_sw = (vx0 & vy1) ? wx0 * wy1 : 0.0f;
  const float w_se = (vx1 & vy1) ? wx1 * wy1 : 0.0f;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  const int i_nw = cy0 * w + cx0, i_ne = cy0 * w + cx1, i_sw = cy1 * w + cx0, i_se = cy1 * w + cx1;
  for (int cc = 0; cc < cend; ++cc) {
    const float* p = xp + (size_t)cc * plane;
    float v = p[i_nw] * w_nw;
    v += p[i_ne] * w_ne;
    v += p[i_sw] * w_sw;
    v += p[i_se] * w_se;
    op[(size_t)cc * plane] = v;
  }
}

}  // namespace

extern "C" int eavsr_flow_warp_f32(const float* x, const float* flow, const float* flow2, float* out,
                                   int32_t n, int32_t c, int32_t h, int32_t w, int32_t flow_layout,
                                   int32_t padding_mode, void* stream) {
  EAVSR_REQUIRE(x && flow && out, -1, "flow_warp: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && h >= 0 && w >= 0, -1, "flow_warp: negative dimension");
  EAVSR_REQUIRE(flow_layout == EAVSR_FLOW_NCHW || flow_layout == EAVSR_FLOW_NHWC, -1,
                "flow_warp: flow_layout %d", flow_layout);
  const int padmode = padding_mode & 3;
  const int nearest = (padding_mode & EAVSR_WARP_NEAREST) != 0, align = (padding_mode & EAVSR_WARP_NO_ALIGN_CORNERS) == 0;
  EAVSR_REQUIRE(padmode <= EAVSR_PAD_REFLECTION && (padding_mode & ~(3 | EAVSR_WARP_NEAREST | EAVSR_WARP_NO_ALIGN_CORNERS)) == 0, -2,
                "flow_warp: padding_mode / flags 0x%x unsupported", padding_mode);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "flow_warp: plane too large");
  if (n == 0 || c == 0 || h == 0 || w == 0) return 0;
  long fs_n, fs_c, fs_y, fs_x;
  if (flow_layout == EAVSR_FLOW_NCHW) {
    fs_n = 2L * h * w; fs_c = (long)h * w; fs_y = w; fs_x = 1;
  } else {
    fs_n = 2L * h * w; fs_c = 1; fs_y = 2L * w; fs_x = 2;
  }
  const int c_chunks = eavsr::cdiv(c, kChanPerThread);
  const int tiles_x = eavsr::cdiv(w, 64), tiles_y = eavsr::cdiv(h, 4);
  const long nblk = (long)tiles_x * tiles_y * c_chunks * n;
  EAVSR_REQUIRE(nblk < (1L << 31), -1, "flow_warp: too many tiles");
  dim3 grid((unsigned)nblk), block(64, 4, 1);
  if (nearest || !align || padmode == EAVSR_PAD_REFLECTION)
    hipLaunchKernelGGL(flow_warp_generic_kernel, grid, block, 0, eavsr::as_stream(stream), x, flow, flow2, out, n, c, h, w,
                       fs_n, fs_c, fs_y, fs_x, c_chunks, tiles_x, tiles_y, padmode, nearest, align);
  else if (padmode == EAVSR_PAD_ZEROS)
    hipLaunchKernelGGL(flow_warp_kernel<EAVSR_PAD_ZEROS>, grid, block, 0, eavsr::as_stream(stream), x, flow,
                       flow2, out, n, c, h, w, fs_n, fs_c, fs_y, fs_x, c_chunks, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL(flow_warp_kernel<EAVSR_PAD_BORDER>, grid, block, 0, eavsr::as_stream(stream), x, flow,
                       flow2, out, n, c, h, w, fs_n, fs_c, fs_y, fs_x, c_chunks, tiles_x, tiles_y);
  return eavsr::launch_status("flow_warp");
}

extern "C" int eavsr_flow_warp_pair_f32(const float* xa, const float* xb, const float* flow, const float* flow2, float* outa,
                                        float* outb, int32_t n, int32_t c, int32_t h, int32_t w, int32_t outb_il8,
                                        void* stream) {
  EAVSR_REQUIRE(xa && xb && flow && outa && outb, -1, "flow_warp_pair: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c >= 0 && h >= 0 && w >= 0, -1, "flow_warp_pair: negative dimension");
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "flow_warp_pair: plane too large");
  EAVSR_REQUIRE(outb_il8 >= 0 && outb_il8 <= 3, -1, "flow_warp_pair: outb_il8 %d (0 NCHW, 1 IL8 fp32, 2 IL8 fp16, 3 IL8 bf16)", outb_il8);
  EAVSR_REQUIRE(!outb_il8 || (c % 8 == 0 && (((uintptr_t)outb) & 15) == 0), -2,
                "flow_warp_pair: the IL8 output needs c %% 8 == 0 and a 16-byte aligned buffer");
  if (n == 0 || c == 0 || h == 0 || w == 0) return 0;
  const int c_chunks = eavsr::cdiv(c, kChanPerThread);
  const int tiles_x = eavsr::cdiv(w, 64), tiles_y = eavsr::cdiv(h, 4);
  const long nblk = (long)tiles_x * tiles_y * c_chunks * 2 * n;
  EAVSR_REQUIRE(nblk < (1L << 31), -1, "flow_warp_pair: too many tiles");
  const dim3 grid((unsigned)nblk), block(64, 4, 1);
  hipStream_t st = eavsr::as_stream(stream);
#define EAVSR_WARP_PAIR(M) hipLaunchKernelGGL(flow_warp_pair_kernel<M>, grid, block, 0, st, xa, xb, flow, flow2, outa, outb, n, c, h, w, c_chunks, tiles_x, tiles_y)
  if (outb_il8 == 0) EAVSR_WARP_PAIR(0);
  else if (outb_il8 == 1) EAVSR_WARP_PAIR(1);
  else if (outb_il8 == 2) EAVSR_WARP_PAIR(2);
  else EAVSR_WARP_PAIR(3);
#undef EAVSR_WARP_PAIR
  return eavsr::launch_status("flow_warp_pair");
}
