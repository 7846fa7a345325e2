// 3x3 stride-1 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores.
// Same operands / epilogue / tensors as conv2d_mfma_kernel<3, 2, true> (conv_mfma.hip) for the residual backbone
// (networks.py:456-458,478; eavsrp_model.py:381), 2.25x fewer multiplications:
//
//   Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A        per 4x4 input tile d, 3x3 filter g   (Lavin & Gray 2016)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// i.e. 16 independent GEMMs (one per transform-domain position xi): M_xi[co, t] = sum_ci U_xi[co, ci] V_xi[ci, t], all
// fp32 (v_mfma_f32_32x32x2_f32); the transforms are additions and multiplications by 1/2.  This is what cuDNN / MIOpen
// run for an fp32 3x3 convolution by default -- fp32 arithmetic throughout, different rounding than the direct sum
// (tests/test_hip_ops.py compares both against fp64).
//
// Per workgroup (512 threads = 8 waves, one per CU): 8 x 32 output pixels = 4 x 16 Winograd tiles, 64 output channels.
//   The GEMMs run on v_mfma_f32_16x16x4_f32 (same FLOP rate as 32x32x2) so that ONE wave can hold all 16 transform
//   positions of a (16 output channels x 16 tiles) block in 64 registers: wave w owns output channels 16 (w >> 1) ..
//   and the tile rows 2 (w & 1), 2 (w & 1) + 1 (16 tiles each) -- 128 accumulator registers -- and the output transform needs no
//   exchange between waves (a first version with 32x32x2 tiles, two positions per wave, spent 22 % of its time
//   moving M through LDS).
//   per chunk of 8 input channels (two LDS stages each for the patch, U and V; one barrier per chunk):
//     LDS-DMA: the (8 x 10 x 40) fp32 input patch (zero padding = never-written zero-initialised LDS) and the
//              pre-transformed weight slab U[16][8][64] (eavsr_pack_conv_weight_wino)          (HBM / L2 -> LDS)
//     input transform: thread (c, tile) reads its 4 x 4 patch, 32 additions, writes V[16][8][64]   (LDS -> LDS)
//     GEMM: per position and 4-channel k-step: 1 + 2 ds_read_b32, 2 MFMAs; U and V rows carry an XOR swizzle
//           (column ^ 16 (c & 1)) so that the four k-rows a wave reads at once fall on different banks
//   epilogue in registers: output transform (24 additions per (co, tile)), + bias, activation, + residual, float2
//   row stores, optional per-tile channel sums (deterministic).
#include "common.h"

#include <mutex>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) float g_wino_zero[4];   // source of the zero-padding DMA lanes

struct WnArgs {
  const float* src[5];
  int src_c[5];
  int n_src;
  const float* wu;        // [cot][cin / 8][16][8][64]
  const float* bias;
  const float* residual;
  float* out;
  float* chan_partial;
  const float* ca_scale;  // FUSE: effective input = src[0] * ca_scale[n, c] + ca_x
  const float* ca_x;
  float* ca_out;          // FUSE: optional copy of the effective input (the next residual stream)
  int n, h, w, cin, cout, tiles_x, tiles_y;
  int act;
  float slope;
};

constexpr int CK = 8, NW = 8;
constexpr int TOH = 8, TOW = 32;                      // output tile: 4 x 16 Winograd tiles; a row is 32 px = one 128-byte line
constexpr int MARG = 4;                               // patch starts 4 columns left of the tile: 16-byte DMA pieces
constexpr int IH = TOH + 2, IW = TOW + 2 * MARG;      // 10 x 40
constexpr int IN_ELEMS = CK * IH * IW;                // 3200 floats
constexpr int IN_SEGS = (IN_ELEMS + 255) / 256;       // 13 one-KiB pieces (the last one half)
constexpr int IN_PAD = IN_SEGS * 256;
constexpr int IN_IT = (IN_SEGS + NW - 1) / NW;        // 2
constexpr int UV = 16 * CK * 64;                      // floats of one U (or V) chunk: 8192 = 32 KB
constexpr int W_SEGS = UV / 256;                      // 32 pieces
constexpr int W_IT = W_SEGS / NW;                     // 4
constexpr int OFF_U = 2 * IN_PAD;                     // LDS map: [patch 0][patch 1][U 0][U 1][V 0][V 1][channel sums]
constexpr int OFF_V = OFF_U + 2 * UV;
constexpr int LDS_MAIN = OFF_V + 2 * UV;              // 154.6 KB
constexpr int LDS_FLOATS = LDS_MAIN + 128;
constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
// FUSE (channel-attention prologue, RCABlock tail folded into the next conv): two patches (r, x) per stage, ONE V stage
// (two barriers per chunk): [r 0][r 1][x 0][x 1][U 0][U 1][V][scale (<= 256 channels)][channel sums]
constexpr int F_OFF_X = 2 * IN_PAD, F_OFF_U = 4 * IN_PAD, F_OFF_V = F_OFF_U + 2 * UV, F_OFF_Y = F_OFF_V + UV;
constexpr int F_MAXC = 256;
constexpr int F_LDS_FLOATS = F_OFF_Y + F_MAXC + 128;
constexpr size_t F_LDS_BYTES = (size_t)F_LDS_FLOATS * sizeof(float);

template <bool FUSE>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(WnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int O_U = FUSE ? F_OFF_U : OFF_U;
  float* s_v = smem + (FUSE ? F_OFF_V : OFF_V);
  float* s_y = smem + F_OFF_Y;                       // FUSE only
  float* s_red = smem + (FUSE ? F_OFF_Y + F_MAXC : LDS_MAIN);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  eavsr_stagger_priority(wave);

  // Persistent workgroups: workgroup b walks the tiles b, b + gridDim.x, ... (XCD-contiguous order) as ONE flattened
  // sequence of (tile, chunk) iterations, so the LDS-DMA stream never drains at a tile boundary and the epilogue of
  // a tile runs while the first chunks of the next one are already in flight (the per-tile fixed cost was 23 % of
  // the kernel).
  const int cot = blockIdx.y;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;
  const int total_tiles = a.tiles_x * a.tiles_y * a.n;
  const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto tile_coords = [&](int k, int& bn_, int& y0_, int& x0_, int& lin_) __attribute__((always_inline)) {
    int t = eavsr_xcd_remap((int)blockIdx.x + k * (int)gridDim.x, total_tiles);
    const int tx_ = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty_ = t % a.tiles_y;
    bn_ = t / a.tiles_y;
    y0_ = ty_ * TOH;
    x0_ = tx_ * TOW;
    lin_ = ty_ * a.tiles_x + tx_;
  };

  // acc[b][xi]: M_xi of output channels 16 cb + 4 (lane >> 4) + r, tile 16 (tb0 + b) + (lane & 15)
  const int l15 = lane & 15, kq = lane >> 4;
  const int cb = wave >> 1, tb0 = 2 * (wave & 1);
  float bias_r[4];   // this lane's four output channels are the same for every tile of the launch
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = cot * 64 + cb * 16 + 4 * kq + r;
    bias_r[r] = (co < a.cout && a.bias != nullptr) ? a.bias[co] : 0.f;
  }
  f32x4 acc[2][16];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[b][x] = f32x4{0.f, 0.f, 0.f, 0.f};

  int total_chunks = 0;
  for (int s = 0; s < a.n_src; ++s) total_chunks += a.src_c[s] / CK;
  const int total_iters = my_tiles * total_chunks;

  // ---- prefetch stream (runs up to two chunks ahead of the compute stream, across tile boundaries) -------------
  int p_k = 0, p_cs = 0, p_cc0 = 0, p_bn = 0, p_y0 = 0, p_x0 = 0, p_lin = 0;
  unsigned voff[IN_IT];   // per-lane byte offsets of this wave's patch pieces; 0xFFFFFFFF: zero padding
  auto p_setup_tile = [&]() __attribute__((always_inline)) {
    tile_coords(p_k, p_bn, p_y0, p_x0, p_lin);
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      const int e4 = seg * 64 + lane;
      const int ci = e4 / (IH * (IW / 4));
      const int rem = e4 - ci * (IH * (IW / 4));
      const int r = rem / (IW / 4);
      const int c4 = rem - r * (IW / 4);
      const int gy = p_y0 - 1 + r, gx = p_x0 - MARG + 4 * c4;
      const bool ok = e4 < IN_ELEMS / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;
      voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)gy * w + gx) * 4) : 0xFFFFFFFFu;
    }
  };
  p_setup_tile();
  // ---- compute stream -----------------------------------------------------------------------------------------
  int bn = 0, y0 = 0, x0 = 0, tile_lin = 0;
  tile_coords(0, bn, y0, x0, tile_lin);
  if (FUSE) {
    for (int c = tid; c < a.cin; c += 64 * NW) s_y[c] = a.ca_scale[(size_t)bn * a.cin + c];
  }

  // the patch of the NEXT chunk of the prefetch stream -> patch stage.  Every piece is always issued -- lanes outside
  // the image copy 16 zero bytes -- so a stage never keeps data of the previous tile.
  auto issue_patch = [&](int stage) {
    float* s_in = smem + stage * IN_PAD;
    const int sc = a.src_c[p_cs];
    const char* sp = reinterpret_cast<const char*>(a.src[p_cs] + ((size_t)p_bn * sc + p_cc0) * plane);
    const char* xp = FUSE ? reinterpret_cast<const char*>(a.ca_x + ((size_t)p_bn * sc + p_cc0) * plane) : nullptr;
    const char* zp = reinterpret_cast<const char*>(g_wino_zero);
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      if (seg < IN_SEGS) {   // wave-uniform
        const bool ok = voff[i] != 0xFFFFFFFFu;
        __builtin_amdgcn_global_load_lds((gptr_t)(ok ? sp + voff[i] : zp), (lptr_t)(s_in + seg * 256), 16, 0, 0);
        if (FUSE)
          __builtin_amdgcn_global_load_lds((gptr_t)(ok ? xp + voff[i] : zp), (lptr_t)(s_in + F_OFF_X + seg * 256), 16, 0, 0);
      }
    }
    p_cc0 += CK;   // advance over the virtual concatenation of the sources, then over this workgroup's tiles
    if (p_cc0 >= a.src_c[p_cs]) {
      ++p_cs;
      p_cc0 = 0;
      if (p_cs >= a.n_src) {
        p_cs = 0;
        ++p_k;
        if (p_k < my_tiles) p_setup_tile();
      }
    }
  };
  auto issue_u = [&](int g, int stage) {
    float* s_u = smem + O_U + stage * UV;
    const char* usrc = reinterpret_cast<const char*>(a.wu + ((size_t)cot * (a.cin / CK) + (size_t)g) * UV);
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = i * NW + wave;
      __builtin_amdgcn_global_load_lds((gptr_t)(usrc + (unsigned)(seg * 64 + lane) * 16u), (lptr_t)(s_u + seg * 256), 16, 0, 0);
    }
  };

  // input-transform job of this thread: channel tc of the chunk, Winograd tile tt = 16 tty + ttx
  const int tc = tid >> 6, tt = tid & 63;
  const int tty = tt >> 4, ttx = tt & 15;
  const int poff = tc * (IH * IW) + (2 * tty) * IW + (MARG - 1) + 2 * ttx;   // top-left of its 4 x 4 patch
  float* vdst = s_v + tc * 64 + (tt ^ ((tc & 1) << 4));                      // + xi * (CK * 64); swizzled column

  // input transform V = B^T d B of this thread's (channel, tile): patch stage ps -> V stage vs
  auto transform = [&](int ps, int vs, int chunk) __attribute__((always_inline)) {
#ifndef EAVSR_WINO_EXP_NOTRANSFORM   // timing ablations only (tools/gpu_wino_ablate.py): results are wrong
    const float* pp = smem + ps * IN_PAD + poff;
    float* vd = vdst + vs * UV;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[r][q] = pp[r * IW + q];
    if (FUSE) {
      // effective input r * scale[n, c] + x (RCABlock tail, networks.py:447,463-464); its interior 2 x 2 pixels are
      // this tile's share of the side output (the next block's residual stream)
      const float sc_ = s_y[chunk * CK + tc];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[r][q] = fmaf(d[r][q], sc_, pp[F_OFF_X + r * IW + q]);
      if (a.ca_out != nullptr && cot == 0) {
        const int gx = x0 + 2 * ttx;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int gy = y0 + 2 * tty + dy;
          if (gy < h && gx < w)
            *reinterpret_cast<f32x2*>(a.ca_out + ((size_t)bn * a.cin + chunk * CK + tc) * plane + (size_t)gy * w + gx) =
                f32x2{d[1 + dy][1], d[1 + dy][2]};
        }
      }
    }
    float t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // B^T d: rows
      t[0][q] = d[0][q] - d[2][q];
      t[1][q] = d[1][q] + d[2][q];
      t[2][q] = d[2][q] - d[1][q];
      t[3][q] = d[1][q] - d[3][q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // (.) B: columns
      vd[(r * 4 + 0) * (CK * 64)] = t[r][0] - t[r][2];
      vd[(r * 4 + 1) * (CK * 64)] = t[r][1] + t[r][2];
      vd[(r * 4 + 2) * (CK * 64)] = t[r][2] - t[r][1];
      vd[(r * 4 + 3) * (CK * 64)] = t[r][1] - t[r][3];
    }
#endif
  };

  // Pipeline (one barrier per chunk): iteration j multiplies chunk j (V[j&1], U[j&1]) right after transforming chunk
  // j+1 (patch[(j+1)&1] -> V[(j+1)&1]); the weight slab runs one chunk ahead of its GEMM, the input patch two.
  // FUSE (one V stage): iteration j transforms chunk j, barrier, multiplies it; everything runs one chunk ahead.
  // j runs over ALL (tile, chunk) pairs of this workgroup.
  issue_patch(0);
  issue_u(0, 0);
  if (!FUSE) {
    if (total_iters > 1) issue_patch(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform(0, 0, 0);
  }
  int chunk = 0;   // chunk of iteration `it` within its tile
  for (int it = 0; it < total_iters; ++it) {
    // U(it) and the patch the next transform needs have landed; every wave is done with the GEMM of iteration it-1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int chunk_n = chunk + 1 == total_chunks ? 0 : chunk + 1;
    auto issue_dma = [&]() __attribute__((always_inline)) {
#ifndef EAVSR_WINO_EXP_NODMA
      if (it + 1 < total_iters) issue_u(chunk_n, (it + 1) & 1);
      if (FUSE) {
        if (it + 1 < total_iters) issue_patch((it + 1) & 1);
      } else {
        if (it + 2 < total_iters) issue_patch(it & 1);
      }
#endif
    };
    // The two waves of a SIMD (w and w + 4) issue their DMA pieces at different points of the iteration: if all eight
    // waves issue right after the barrier, every wave of the CU stalls in the vector-memory issue at once and the
    // matrix pipe idles; staggered, one wave of each pair is in its MFMAs while the other issues.
    const bool dma_late = !FUSE && wave >= 4;
    if (!dma_late) issue_dma();
    if (FUSE) {
      transform(it & 1, 0, chunk);
      __syncthreads();
    } else if (it + 1 < total_iters) {
      transform((it + 1) & 1, (it + 1) & 1, chunk_n);
    }
    // ---- the 16 GEMMs of this wave's two blocks: M_xi[co, t] += sum over the chunk's 8 channels U_xi[co, c] V_xi[c, t]
    const float* su = smem + O_U + (it & 1) * UV;
    const int sw = (kq & 1) << 4;                       // swizzle of the row this lane reads (c = 4 ks + kq)
    const float* ua = su + kq * 64 + ((cb * 16 + l15) ^ sw);
    const float* sv = s_v + (FUSE ? 0 : (it & 1) * UV);
    const float* vb0 = sv + kq * 64 + ((tb0 * 16 + l15) ^ sw);
    const float* vb1 = sv + kq * 64 + ((tb0 * 16 + 16 + l15) ^ sw);
    // operands run three steps ahead of the MFMAs (32 steps = 16 positions x 2 k-steps): an LDS read takes longer than
    // the 64 cycles of a step's two MFMAs
    constexpr int NSTEP = 16 * (CK / 4), AHEAD = 3;
    auto row_of = [](int i) { return ((i / (CK / 4)) * CK + 4 * (i % (CK / 4))) * 64; };
    float av[AHEAD + 1], b0[AHEAD + 1], b1[AHEAD + 1];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) { av[i] = ua[row_of(i)]; b0[i] = vb0[row_of(i)]; b1[i] = vb1[row_of(i)]; }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      if (i == NSTEP / 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (dma_late) issue_dma();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i + AHEAD < NSTEP) {
        av[(i + AHEAD) % (AHEAD + 1)] = ua[row_of(i + AHEAD)];
        b0[(i + AHEAD) % (AHEAD + 1)] = vb0[row_of(i + AHEAD)];
        b1[(i + AHEAD) % (AHEAD + 1)] = vb1[row_of(i + AHEAD)];
      }
      const int xi = i / (CK / 4), cur = i % (AHEAD + 1);
      acc[0][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur], b0[cur], acc[0][xi], 0, 0, 0);
      acc[1][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur], b1[cur], acc[1][xi], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the 3 reads of step i + AHEAD
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // the 2 MFMAs of step i
    }
    chunk = chunk_n;
    if (chunk != 0) continue;   // the tile is not finished yet

  // ---- epilogue, all in registers: lane (kq, l15) holds M_xi[co = 16 cb + 4 kq + r][tile] for every xi ----------
  {
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    // Every residual load is issued before the first store (the bias was read once at kernel start): loads and
    // stores share the vmcnt queue and the output may alias, so a load placed after a store waits for that
    // store's acknowledgement - eight to twenty-four serialised round trips per tile otherwise.
    const int gx = x0 + 2 * l15;   // tile block = one row of 16 tiles: 16 lanes x float2 = one 128-byte line
    f32x2 rr[2][4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int co = cot * 64 + cb * 16 + 4 * kq + r;
          const int gy = y0 + 2 * (tb0 + b) + dy;
          rr[b][r][dy] = f32x2{0.f, 0.f};
          if (a.residual != nullptr && co < a.cout && gy < h && gx < w)
            rr[b][r][dy] = *reinterpret_cast<const f32x2*>(a.residual + ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx);
        }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * 64 + cb * 16 + 4 * kq + r;
        float s0[4], s1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // A^T m: rows
          s0[q] = acc[b][0 + q][r] + acc[b][4 + q][r] + acc[b][8 + q][r];
          s1[q] = acc[b][4 + q][r] - acc[b][8 + q][r] - acc[b][12 + q][r];
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2];
        y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2];
        y[1][1] = s1[1] - s1[2] - s1[3];
        const bool cok = co < a.cout;
        const float bb = bias_r[r];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int gy = y0 + 2 * (tb0 + b) + dy;
          float v0 = y[dy][0] + bb, v1 = y[dy][1] + bb;
          { const float as_ = a.act == EAVSR_ACT_NONE ? 1.f : a.act == EAVSR_ACT_RELU ? 0.f : a.slope; v0 = eavsr_act(v0, as_); v1 = eavsr_act(v1, as_); }   // branch-free: max(v, v s)
          if (cok && gy < h && gx < w) {     // w % 4 == 0 and gx even: gx + 1 < w as well
            const size_t o = ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx;
            csum[r] += v0 + v1;
            *reinterpret_cast<f32x2*>(a.out + o) = f32x2{v0 + rr[b][r][dy].x, v1 + rr[b][r][dy].y};
          }
        }
      }
    }
    if (a.chan_partial) {
      // per output channel: 16 lanes (tiles) x 2 blocks here, and the same again in wave ^ 1 (fixed order)
      __syncthreads();   // s_red is free (previous tile's sums were read long ago)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = csum[r];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (l15 == 0) s_red[(wave & 1) * 64 + cb * 16 + 4 * kq + r] = v;
      }
      __syncthreads();
      if (tid < 64) {
        const int co = cot * 64 + tid;
        if (co < a.cout)
          a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + tile_lin) * a.cout + co] = s_red[tid] + s_red[64 + tid];
      }
    }
    // next tile of this workgroup
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[b][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it + 1 < total_iters) {
      tile_coords((it + 1) / total_chunks, bn, y0, x0, tile_lin);
      if (FUSE) {   // visible after the barrier at the top of the next iteration; nobody reads s_y until then
        for (int c = tid; c < a.cin; c += 64 * NW) s_y[c] = a.ca_scale[(size_t)bn * a.cin + c];
      }
    }
  }
  }   // flattened (tile, chunk) loop
}


// weight (cout, cin, 3, 3) -> U = G g G^T laid out [cot][cin / 8][xi][c][co] (zero for co >= cout)
__global__ void pack_wino_kernel(const float* __restrict__ wt, float* __restrict__ out, int cout, int cin, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int col = (int)(e & 63);
  long u = e >> 6;
  const int c = (int)(u % CK); u /= CK;
  const int xi = (int)(u % 16); u /= 16;
  const int nchunks = cin / CK;
  const int chunk = (int)(u % nchunks);
  const int cot = (int)(u / nchunks);
  const int co = cot * 64 + col, ci = chunk * CK + c;
  float v = 0.f;
  if (co < cout) {
    const float* g = wt + ((size_t)co * cin + ci) * 9;
    const int r = xi >> 2, q = xi & 3;
    // row r of G g (a 3-vector), then column q of (.) G^T
    float gr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float g0 = g[0 * 3 + j], g1 = g[1 * 3 + j], g2 = g[2 * 3 + j];
      gr[j] = r == 0 ? g0 : (r == 1 ? 0.5f * (g0 + g1 + g2) : (r == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    }
    v = q == 0 ? gr[0] : (q == 1 ? 0.5f * (gr[0] + gr[1] + gr[2]) : (q == 2 ? 0.5f * (gr[0] - gr[1] + gr[2]) : gr[2]));
  }
  // XOR swizzle of the column (see the GEMM loop): element (xi, c, co) lives at column co ^ 16 (c & 1)
  out[e - col + (col ^ ((c & 1) << 4))] = v;
}

}  // namespace

extern "C" int64_t eavsr_wino_weight_elems(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || cin % CK != 0) return 0;
  return (int64_t)eavsr::cdiv(cout, 64) * (cin / CK) * UV;
}

extern "C" int eavsr_pack_conv_weight_wino(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight_wino: NULL pointer");
  EAVSR_REQUIRE(cout > 0 && cin > 0 && cin % CK == 0, -1, "pack_conv_weight_wino: cin %d must be a multiple of 8", cin);
  const long total = eavsr_wino_weight_elems(cout, cin);
  hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eavsr::as_stream(stream), weight, packed,
                     cout, cin, total);
  return eavsr::launch_status("pack_conv_weight_wino");
}

extern "C" int32_t eavsr_conv3x3_wino_tiles(int32_t h, int32_t w) { return eavsr::cdiv(h, TOH) * eavsr::cdiv(w, TOW); }

extern "C" int eavsr_conv3x3_wino_f32(const eavsr_conv2d_desc* d, const float* weight_wino, void* stream) {
  EAVSR_REQUIRE(d != nullptr && weight_wino != nullptr, -1, "conv3x3_wino: NULL descriptor / weights");
  EAVSR_REQUIRE(d->n_src >= 1 && d->n_src <= 5, -1, "conv3x3_wino: n_src %d not in 1..5", d->n_src);
  EAVSR_REQUIRE(d->ksize == 3, -2, "conv3x3_wino: kernel size %d (3 only)", d->ksize);
  EAVSR_REQUIRE(d->out, -1, "conv3x3_wino: NULL out");
  EAVSR_REQUIRE(d->out_shuffle == 0 && d->res_scale == nullptr && d->border_pieces == nullptr && d->sum_mul == nullptr, -2, "conv3x3_wino: the pixel-shuffle and scaled-residual epilogues exist in eavsr_conv3x3_wino4_f32 only");
  EAVSR_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, -1, "conv3x3_wino: bad dims");
  EAVSR_REQUIRE(d->act >= 0 && d->act <= 2, -1, "conv3x3_wino: act %d", d->act);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_LRELU || (d->slope >= 0.f && d->slope <= 1.f), -2,
                "conv3x3_wino: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)d->slope);
  const bool fuse = d->ca_scale != nullptr;
  if (fuse) {
    EAVSR_REQUIRE(d->ca_x != nullptr, -1, "conv3x3_wino: ca_scale without ca_x");
    EAVSR_REQUIRE(d->n_src == 1 && d->cin <= F_MAXC && (((uintptr_t)d->ca_x) & 15) == 0 &&
                      (d->ca_out == nullptr || (((uintptr_t)d->ca_out) & 7) == 0), -2,
                  "conv3x3_wino: the fused channel-attention prologue needs a single source, cin <= %d and aligned ca_x / "
                  "ca_out", F_MAXC);
  } else {
    EAVSR_REQUIRE(d->ca_x == nullptr && d->ca_out == nullptr, -1, "conv3x3_wino: ca_x / ca_out without ca_scale");
  }
  EAVSR_REQUIRE(d->w % 4 == 0, -2, "conv3x3_wino: w %% 4 != 0 (use eavsr_conv2d_f32)");
  EAVSR_REQUIRE((((uintptr_t)d->out) & 7) == 0 && (d->residual == nullptr || (((uintptr_t)d->residual) & 7) == 0), -2,
                "conv3x3_wino: out / residual must be 8-byte aligned");
  WnArgs a;
  int csum = 0;
  for (int s = 0; s < 5; ++s) {
    a.src[s] = s < d->n_src ? d->src[s] : nullptr;
    a.src_c[s] = s < d->n_src ? d->src_c[s] : 0;
    if (s < d->n_src) {
      EAVSR_REQUIRE(d->src[s] != nullptr && d->src_c[s] > 0 && d->src_c[s] % CK == 0 && (((uintptr_t)d->src[s]) & 15) == 0, -2,
                    "conv3x3_wino: source %d must be 16-byte aligned with a multiple of 8 channels", s);
      csum += d->src_c[s];
    }
  }
  EAVSR_REQUIRE(csum == d->cin, -1, "conv3x3_wino: sources sum to %d channels, cin = %d", csum, d->cin);
  if (d->n == 0) return 0;
  a.n_src = d->n_src;
  a.wu = weight_wino;
  a.bias = d->bias; a.residual = d->residual; a.out = d->out; a.chan_partial = d->chan_partial;
  a.ca_scale = d->ca_scale; a.ca_x = d->ca_x; a.ca_out = d->ca_out;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
  a.tiles_x = eavsr::cdiv(d->w, TOW);
  a.tiles_y = eavsr::cdiv(d->h, TOH);
  a.act = d->act; a.slope = d->slope;
  const long blocks = (long)a.tiles_x * a.tiles_y * d->n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv3x3_wino: too many tiles");
  EAVSR_REQUIRE((long)d->h * d->w * 16 < (1L << 31), -1, "conv3x3_wino: image plane too large for 32-bit tile offsets");
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (attr_err == hipSuccess)
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv3x3_wino: hipFuncSetAttribute(%zu B of LDS): %s", LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  // persistent workgroups: one per CU (the kernel needs > 80 KB of LDS), each walking blocks / grid.x tiles
  const long per_cot = blocks < 256 ? blocks : 256;
  dim3 grid((unsigned)per_cot, eavsr::cdiv(d->cout, 64));
  if (fuse) hipLaunchKernelGGL(conv3x3_wino_kernel<true>, grid, dim3(64 * NW), F_LDS_BYTES, eavsr::as_stream(stream), a);
  else hipLaunchKernelGGL(conv3x3_wino_kernel<false>, grid, dim3(64 * NW), LDS_BYTES, eavsr::as_stream(stream), a);
  return eavsr::launch_status("conv3x3_wino");
}
