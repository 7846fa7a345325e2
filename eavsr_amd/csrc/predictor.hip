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

#include <mutex>

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

// The same operation for w % 4 == 0 and 16-byte aligned tensors (every call of the model): register-tiled.  The patch starts 4
// columns left of the tile, so it arrives as aligned 16-byte loads (3 per thread instead of 11 scalar ones with their index
// arithmetic); each stage evaluates groups of 4 adjacent columns -- a stencil row is one ds_read_b128 + one b128 / b64 for 4
// outputs x 3 taps instead of one ds_read_b32 per FMA.  2 x 64 x 180 x 320: 50 -> see DESIGN.md 4g.
typedef float pf32x2 __attribute__((ext_vector_type(2)));
constexpr int GP_W = FT_W + 8;        // patch columns x0 - 4 .. x0 + 67 (72, 16-byte rows)
constexpr int GM_W = FT_W + 4;        // mid columns x0 - 1 .. x0 + 66 (68: 17 groups of 4; the last two are never used)
__global__ __launch_bounds__(256) void adapt_frontend_v4_kernel(
    const float* __restrict__ x, const float* __restrict__ hh, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ out, int c, int h, int w, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float s_in[2][FP_H][GP_W];
  __shared__ __attribute__((aligned(16))) float s_mid[2][FM_H][GM_W];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int o = blockIdx.y, bn = blockIdx.z;
  const int y0 = ty * FT_H, x0 = tx * FT_W;
  const size_t plane = (size_t)h * w;

  // patch: 2 channels x 20 rows x 18 quads, every load issued before the first LDS write
  constexpr int Q_N = 2 * FP_H * (GP_W / 4), Q_IT = (Q_N + 255) / 256;   // 720 quads, 3 per thread
  f32x4 tq[Q_IT];
#pragma unroll
  for (int i = 0; i < Q_IT; ++i) {
    const int e = min(tid + i * 256, Q_N - 1);
    const int ch = e / (FP_H * (GP_W / 4));
    const int rem = e - ch * (FP_H * (GP_W / 4));
    const int r = rem / (GP_W / 4), q = rem - r * (GP_W / 4);
    const int gy = y0 - 2 + r, gx = x0 - 4 + 4 * q;
    const int cat_c = 2 * o + ch;
    const float* src = cat_c < c ? x : hh;
    const int sc = cat_c < c ? cat_c : cat_c - c;
    const int cgy = min(max(gy, 0), h - 1), cgx = min(max(gx, 0), w - 4);
    tq[i] = *reinterpret_cast<const f32x4*>(src + ((size_t)bn * c + sc) * plane + (size_t)cgy * w + cgx);
  }
#pragma unroll
  for (int i = 0; i < Q_IT; ++i) {
    const int e = tid + i * 256;
    if (i < Q_IT - 1 || e < Q_N) {
      const int ch = e / (FP_H * (GP_W / 4));
      const int rem = e - ch * (FP_H * (GP_W / 4));
      const int r = rem / (GP_W / 4), q = rem - r * (GP_W / 4);
      const int gy = y0 - 2 + r, gx = x0 - 4 + 4 * q;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;      // w % 4 == 0: a quad is inside or outside as a whole
      *reinterpret_cast<f32x4*>(&s_in[ch][r][4 * q]) = ok ? tq[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  // t1: mid column m <-> gx = x0 - 1 + m needs patch columns m + 2 .. m + 4; group g = columns 4g .. 4g + 3
  for (int e = tid; e < 2 * FM_H * (GM_W / 4); e += 256) {
    const int ch = e / (FM_H * (GM_W / 4));
    const int rem = e - ch * (FM_H * (GM_W / 4));
    const int r = rem / (GM_W / 4), g = rem - r * (GM_W / 4);
    const int gy = y0 - 1 + r;
    const float* wk = w1 + (size_t)(2 * o + ch) * 9;
    const float bias = b1[2 * o + ch];
    float v[4] = {bias, bias, bias, bias};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&s_in[ch][r + ky][4 * g]);        // patch columns 4g .. 4g + 3
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&s_in[ch][r + ky][4 * g + 4]);    //               4g + 4 .. 4g + 7
      const float p[6] = {a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};                     // columns 4g + 2 .. 4g + 7
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float wv = wk[ky * 3 + kx];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += wv * p[j + kx];
      }
    }
    f32x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gx = x0 - 1 + 4 * g + j;
      o4[j] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? lrelu02(v[j]) : 0.f;   // zero padding of the second conv
    }
    *reinterpret_cast<f32x4*>(&s_mid[ch][r][4 * g]) = o4;
  }
  __syncthreads();
  // t2: one thread = 4 adjacent output pixels (16 rows x 16 groups = 256 threads); out column j needs mid columns j .. j + 2
  {
    const int r = tid >> 4, g = tid & 15;
    const int gy = y0 + r, gx = x0 + 4 * g;
    if (gy < h && gx < w) {
      const float* wk2 = w2 + (size_t)o * 18;
      const float bias2 = b2[o];
      float v[4] = {bias2, bias2, bias2, bias2};
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(&s_mid[ch][r + ky][4 * g]);
          const pf32x2 b2v = *reinterpret_cast<const pf32x2*>(&s_mid[ch][r + ky][4 * g + 4]);
          const float p[6] = {a4[0], a4[1], a4[2], a4[3], b2v[0], b2v[1]};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wk2[ch * 9 + ky * 3 + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += wv * p[j + kx];
          }
        }
      *reinterpret_cast<f32x4*>(out + ((size_t)bn * c + o) * plane + (size_t)gy * w + gx) =
          f32x4{lrelu02(v[0]), lrelu02(v[1]), lrelu02(v[2]), lrelu02(v[3])};
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
  const bool vec = w % 4 == 0 && w >= 4 && ((((uintptr_t)x) | ((uintptr_t)h_hr) | ((uintptr_t)out)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(adapt_frontend_v4_kernel, grid, dim3(256), 0, eavsr::as_stream(stream), x, h_hr, w1, b1, w2,
                       b2, out, c, h, w, tiles_x);
  else
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

#if EAVSR_LAB      // one kernel per pyramid level: built, correct, not faster (docs/history 4g) -- lab build only
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// (3) flow_level: one pyramid level of MultiAdSTN's residual-flow refinement as ONE kernel (networks.py:604-619):
//        f      = concat2(concat(cat(x, h_hr)))                      AdaptBlock2_3x3 front end (networks.py:327-328,336)
//        heads  = conv3x3(f; transform_matrix_conv ++ translation_conv)     64 -> 4 + 2          (networks.py:330-331,337)
//        off18  = T . R - R + t                                       (networks.py:338-346)
//        p      = conv3x3(off18; TransOffsetworelu.conv_first)        18 -> 2, no activation     (networks.py:566-571)
//     Round 1 ran this as four launches (adapt_frontend, conv3x3 64->6, affine_offsets, conv3x3 18->2) whose intermediates
//     (64, 6 and 18 channels) all went through HBM; here a workgroup reads the 2 x 64 input channels of its tile (+ 4 px
//     halo) once and writes 2 channels.  Channel o of f depends on cat channels 2o, 2o+1 only, so f is produced one
//     channel at a time in LDS and folded at once into the six head accumulators that every thread keeps in registers
//     for its pixels; off18 then lives in LDS for the last convolution.  Every stage is zero outside the IMAGE (each conv
//     of the chain zero-pads its own input), which the staged tiles reproduce explicitly.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LT_H = 16, LT_W = 32;           // output tile
// Every stage lives in an LDS image of 24 rows x 48 columns with the SAME origin (y0 - 4, x0 - 8): stage k of the chain is
// valid one row / column further in than stage k-1, and every stencil is evaluated on aligned groups of four columns (one
// ds_read_b128 + two ds_read_b32 per stencil row feed 4 outputs x 3 taps), columns 4 .. 43.  Groups at the rim compute a
// few values nobody reads.
constexpr int LP = 48, LR = 24;               // pitch, rows
constexpr int LG0 = 1, LGN = 10;              // column groups 1 .. 10  (columns 4 .. 43)
constexpr int LOP = 40;                       // pitch of the off18 image (columns 4 .. 43 -> 0 .. 39), rows 3 .. 20 -> 0 .. 17

// 3 rows x 6 columns around an aligned group of four: v[r][0..5] = img[row + r][col4 - 1 .. col4 + 4]
__device__ __forceinline__ void ld_3x6(const float* img, int pitch, float (&v)[3][6]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const f32x4 m = *reinterpret_cast<const f32x4*>(img + r * pitch);
    v[r][0] = img[r * pitch - 1];
    v[r][1] = m[0]; v[r][2] = m[1]; v[r][3] = m[2]; v[r][4] = m[3];
    v[r][5] = img[r * pitch + 4];
  }
}

__global__ __launch_bounds__(256) void flow_level_kernel(
    const float* __restrict__ x, const float* __restrict__ hh, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ wh, const float* __restrict__ bh,
    const float* __restrict__ wt, const float* __restrict__ bt, float* __restrict__ out, int c, int h, int w, int tiles_x,
    int tiles_y) {
  // LDS: the three stage images, the off18 image, and ALL per-channel weights of the chain (staged once per workgroup:
  // fetching each channel's 93 coefficients by scalar loads put four exposed SMEM round trips into every channel step)
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float (*s_in)[LR * LP] = reinterpret_cast<float (*)[LR * LP]>(lsm);
  float (*s_mid)[LR * LP] = reinterpret_cast<float (*)[LR * LP]>(lsm + 2 * LR * LP);
  float* s_f = lsm + 4 * LR * LP;
  float (*s_off)[18 * LOP] = reinterpret_cast<float (*)[18 * LOP]>(lsm + 5 * LR * LP);
  float* s_wgt = lsm + 5 * LR * LP + 18 * 18 * LOP;      // per f-channel o, 96 floats: [w1 pair 18][b1 pair 2][w2 18][b2 1][pad 3][wh 6 x 9]
  const int tid = threadIdx.x;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int bn = bid / tiles_y;
  const int y0 = ty * LT_H, x0 = tx * LT_W;
  const int gy0 = y0 - 4, gx0 = x0 - 8;        // image coordinates of LDS (row 0, column 0)
  const size_t plane = (size_t)h * w;

  // the LDS images start as zeros: rim columns / rows that no stage writes must hold finite values
  for (int e = tid; e < 2 * LR * LP; e += 256) { (&s_in[0][0])[e] = 0.f; (&s_mid[0][0])[e] = 0.f; }
  for (int e = tid; e < LR * LP; e += 256) s_f[e] = 0.f;
  for (int e = tid; e < c * 96; e += 256) {
    const int o = e / 96, j = e - o * 96;
    float v = 0.f;
    if (j < 18) v = w1[(size_t)(2 * o) * 9 + j];                 // cat channels 2o, 2o+1: 9 taps each (contiguous)
    else if (j < 20) v = b1[2 * o + (j - 18)];
    else if (j < 38) v = w2[(size_t)o * 18 + (j - 20)];
    else if (j == 38) v = b2[o];
    else if (j >= 42) v = wh[((size_t)((j - 42) / 9) * c + o) * 9 + (j - 42) % 9];
    s_wgt[e] = v;
  }

  // input patch: 2 channels x 24 rows x columns 4 .. 43 (= x0 - 4 .. x0 + 35): 1920 values, 7.5 per thread; the offsets
  // are the same for every channel pair
  constexpr int P_N = 2 * LR * 40, P_IT = (P_N + 255) / 256;
  int poff[P_IT], pdst[P_IT];      // offset inside a plane (-1: outside the image), LDS destination
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int e = tid + i * 256;
    const int ch = e >= LR * 40 ? 1 : 0;
    const int rem = e - ch * (LR * 40);
    const int r = rem / 40, cc = rem - r * 40;
    const int gy = gy0 + r, gx = gx0 + 4 + cc;
    poff[i] = (e < P_N && gy >= 0 && gy < h && gx >= 0 && gx < w) ? gy * w + gx : -1;
    pdst[i] = e < P_N ? ch * (LR * LP) + r * LP + 4 + cc : -1;
  }
  auto load_pair = [&](int o, float (&t)[P_IT]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int ch = (tid + i * 256) >= LR * 40 ? 1 : 0;
      const int cat_c = 2 * o + ch;
      const float* src = cat_c < c ? x + ((size_t)bn * c + cat_c) * plane : hh + ((size_t)bn * c + (cat_c - c)) * plane;
      t[i] = poff[i] >= 0 ? src[poff[i]] : 0.f;
    }
  };

  // work items: (row, group of four columns)
  //   t1    rows 1 .. 22, 10 groups, 2 channels: 440 items      f rows 2 .. 21: 200 items      heads rows 3 .. 20: 180 items
  const bool h_item = tid < 18 * LGN;
  const int h_r = 3 + tid / LGN, h_g = LG0 + tid % LGN;
  float hacc[6][4];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) hacc[k][j] = 0.f;

  float tin[P_IT];
  load_pair(0, tin);
  for (int o = 0; o < c; ++o) {
    __syncthreads();      // the previous channel's passes are done with s_in / s_mid / s_f (and the zero fill is complete)
#pragma unroll
    for (int i = 0; i < P_IT; ++i)
      if (pdst[i] >= 0) (&s_in[0][0])[pdst[i]] = tin[i];
    if (o + 1 < c) load_pair(o + 1, tin);      // next channel pair: in flight under this channel's arithmetic
    __syncthreads();
    // t1 = lrelu(depthwise 3x3 + b1), zero outside the image
    for (int it = tid; it < 2 * 22 * LGN; it += 256) {
      const int ch = it >= 22 * LGN ? 1 : 0;
      const int rem = it - ch * (22 * LGN);
      const int r = 1 + rem / LGN, g = LG0 + rem % LGN;
      float v[3][6];
      ld_3x6(&s_in[ch][(r - 1) * LP + 4 * g], LP, v);
      const float* wk = s_wgt + o * 96 + ch * 9;      // wave-uniform LDS addresses: broadcast reads
      const float bb = s_wgt[o * 96 + 18 + ch];
      f32x4 res;
      const int gy = gy0 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = bb;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a += wk[ky * 3 + kx] * v[ky][j + kx];
        const int gx = gx0 + 4 * g + j;
        res[j] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? lrelu02(a) : 0.f;
      }
      *reinterpret_cast<f32x4*>(&s_mid[ch][r * LP + 4 * g]) = res;
    }
    __syncthreads();
    // f_o = lrelu(grouped 3x3 over the two t1 channels + b2), zero outside the image
    if (tid < 20 * LGN) {
      const int r = 2 + tid / LGN, g = LG0 + tid % LGN;
      const float* wk2 = s_wgt + o * 96 + 20;
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = s_wgt[o * 96 + 38];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float v[3][6];
        ld_3x6(&s_mid[ch][(r - 1) * LP + 4 * g], LP, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a[j] += wk2[ch * 9 + ky * 3 + kx] * v[ky][j + kx];
      }
      f32x4 res;
      const int gy = gy0 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gx = gx0 + 4 * g + j;
        res[j] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? lrelu02(a[j]) : 0.f;
      }
      *reinterpret_cast<f32x4*>(&s_f[r * LP + 4 * g]) = res;
    }
    __syncthreads();
    // heads += wh[:, o] * f_o  (3x3, six outputs for the item's four pixels)
    if (h_item) {
      float v[3][6];
      ld_3x6(&s_f[(h_r - 1) * LP + 4 * h_g], LP, v);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float* wk = s_wgt + o * 96 + 42 + k * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float ww = wk[ky * 3 + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) hacc[k][j] += ww * v[ky][j + kx];
          }
      }
    }
  }
  // off18 = T . R - R + t per position, zero outside the image (the padding of TransOffsetworelu's conv)
  if (h_item) {
    const int gy = gy0 + h_r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gx = gx0 + 4 * h_g + j;
      const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const float t00 = hacc[0][j] + bh[0], t01 = hacc[1][j] + bh[1], t10 = hacc[2][j] + bh[2], t11 = hacc[3][j] + bh[3];
      const float try_ = hacc[4][j] + bh[4], trx = hacc[5][j] + bh[5];
      const int e = (h_r - 3) * LOP + 4 * (h_g - LG0) + j;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float ry = (float)(k / 3 - 1), rx = (float)(k % 3 - 1);
        // (T . R)[:,k] - R[:,k] + t     (matmul then subtract then add, networks.py:338-346)
        const float oy = (t00 * ry + t01 * rx) - ry + try_;
        const float ox = (t10 * ry + t11 * rx) - rx + trx;
        s_off[2 * k][e] = in ? oy : 0.f;
        s_off[2 * k + 1][e] = in ? ox : 0.f;
      }
    }
  }
  __syncthreads();
  // p = conv3x3(off18) + bt: rows 4 .. 19, columns 8 .. 39 = off18-image rows 1 .. 16, columns 4 .. 35; 128 items of 4 pixels
  if (tid < LT_H * (LT_W / 4)) {
    const int r = tid / (LT_W / 4), g = tid % (LT_W / 4);      // output row r, pixels 4 g .. 4 g + 3 of the tile
    const int gy = y0 + r, gx = x0 + 4 * g;
    float p0[4], p1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { p0[j] = bt[0]; p1[j] = bt[1]; }
    for (int ch = 0; ch < 18; ++ch) {
      float v[3][6];
      ld_3x6(&s_off[ch][r * LOP + 4 + 4 * g], LOP, v);      // rows r .. r + 2 of the off18 image = image rows gy - 1 .. gy + 1
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float wa = wt[(0 * 18 + ch) * 9 + ky * 3 + kx], wb = wt[(1 * 18 + ch) * 9 + ky * 3 + kx];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            p0[j] += wa * v[ky][j + kx];
            p1[j] += wb * v[ky][j + kx];
          }
        }
    }
    if (gy < h) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (gx + j < w) {
          out[((size_t)bn * 2 + 0) * plane + (size_t)gy * w + gx + j] = p0[j];
          out[((size_t)bn * 2 + 1) * plane + (size_t)gy * w + gx + j] = p1[j];
        }
    }
  }
}

}  // namespace

extern "C" int eavsr_flow_level_f32(const float* x, const float* h_hr, const float* w1, const float* b1, const float* w2,
                                    const float* b2, const float* w_heads, const float* b_heads, const float* w_trans,
                                    const float* b_trans, float* out, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  EAVSR_REQUIRE(x && h_hr && w1 && b1 && w2 && b2 && w_heads && b_heads && w_trans && b_trans && out, -1, "flow_level: NULL pointer");
  EAVSR_REQUIRE(n >= 0 && c > 0 && h > 0 && w > 0, -1, "flow_level: bad dims");
  EAVSR_REQUIRE(c % 2 == 0, -2, "flow_level: channel count %d must be even", c);
  EAVSR_REQUIRE((long)h * w < (1L << 31), -1, "flow_level: plane too large");
  if (n == 0) return 0;
  const int tiles_x = eavsr::cdiv(w, LT_W), tiles_y = eavsr::cdiv(h, LT_H);
  const long nblk = (long)tiles_x * tiles_y * n;
  EAVSR_REQUIRE(nblk < (1L << 31), -1, "flow_level: too many tiles");
  const size_t lds = (size_t)(5 * LR * LP + 18 * 18 * LOP + c * 96) * sizeof(float);
  EAVSR_REQUIRE(lds <= 160 * 1024, -2, "flow_level: %d channels need %zu bytes of LDS", c, lds);
  if (lds > 64 * 1024) {
    static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
    const int dev_ = eavsr::current_device();
    static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
    hipError_t& attr_err = attr_err_pd[dev_];
    std::call_once(once_pd.flag[dev_], [&] {
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&flow_level_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024);
    });
    if (attr_err != hipSuccess) {
      eavsr::set_error("flow_level: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
      return (int)attr_err;
    }
  }
  hipLaunchKernelGGL(flow_level_kernel, dim3((unsigned)nblk), dim3(256), lds, eavsr::as_stream(stream), x, h_hr, w1, b1, w2, b2,
                     w_heads, b_heads, w_trans, b_trans, out, c, h, w, tiles_x, tiles_y);
  return eavsr::launch_status("flow_level");
}
#endif  // EAVSR_LAB
