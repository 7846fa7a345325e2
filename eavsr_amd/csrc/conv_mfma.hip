// Dense stride-1 convolution as an implicit GEMM on the fp32 matrix cores (a9, a10,
// a11 convs, predictor heads of a3/a4/a6; also the 7x7 / encoder / upsample convs either side).
//
// Reference: torch.nn.Conv2d forward at eavsrp_model.py:146,313-314,381 and networks.py:293-295,
// 330-331,456-458,478,568 (cuDNN / MKLDNN behind PyTorch).
//
// GEMM view per image: out[co, px] = sum_{ci,tap} W[co, ci, tap] * in[ci, px + tap],
//   M = cout (32 or 64 per workgroup), N = pixels (32 x 32 tile per workgroup), K = cin * k * k.
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD = the fp32 vector peak but with
// one operand VGPR per 4096 FLOP): lane l supplies A[i = l & 31][k = l >> 5] = weight of output
// channel i for input channel (2 cp + (l >> 5)) at one tap, and B[k = l >> 5][j = l & 31] = the
// input pixel j of that channel shifted by the tap.  The accumulator holds
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
//
// Per workgroup (512 threads = 8 waves, two per SIMD; one workgroup per CU, 32 x 32 pixel tile):
//   for each chunk of CK input channels, two LDS stages: the LDS-DMA (global_load_lds, no VGPRs, no
//   ds_write) of chunk i+1 is in flight behind the MFMA loop of chunk i, one barrier per chunk:
//     the (16 + k - 1) x (32 + k - 1) x CK input patch (zero padded at the image border) and the
//     CK x k*k x CO weight slab (contiguous in the packed layout)               (HBM / L2 -> LDS)
//     every wave: for tap, channel pair: 2*MT + 4 ds_read_b32, MT*4 MFMAs       (LDS -> MFMA)
//   epilogue from the accumulators: + bias, activation, + residual, coalesced 128-B row stores,
//   optional per-tile channel sums for the channel attention (deterministic, no atomics).
// LDS reads are conflict-free by construction: the 32 lanes of a half-wave read 32 consecutive floats.
// The input is the virtual concatenation of up to 5 sources, so torch.cat never materialises.
#include "common.h"

#include <mutex>

namespace {

struct ConvArgs {
  const float* src[5];
  int src_c[5];
  int n_src;
  const float* wp;
  const float* bias;
  const float* residual;
  float* out;
  float* chan_partial;
  const float* ca_scale;  // FUSE: effective input = src[0] * ca_scale[n, c] + ca_x
  const float* ca_x;
  float* ca_out;          // FUSE: optional copy of the effective input (the next residual stream)
  int n, h, w, cin, cout, cin_pad, tiles_x, tiles_y;
  int act;
  float slope;
};

// Lanes of the input-patch DMA that fall outside the image (zero padding) or outside the channel
// range read this zero word instead (an LDS-DMA cannot produce a constant by itself).
__device__ float g_zero_word[4];

template <int KS> struct ChunkOf { static constexpr int value = (KS == 1) ? 16 : (KS == 3) ? 8 : (KS == 5) ? 4 : 2; };

// VEC = true : the input patch moves as 16-byte DMA pieces; needs w % 4 == 0, 16-byte aligned sources and
//              every source a multiple of CK channels.  The patch then starts 4 columns left of the tile
//              (a whole float4) instead of PAD columns, and its per-lane source offsets are computed once
//              per workgroup (the patch geometry is the same for every chunk; only the channel base moves).
// VEC = false: dword pieces with per-piece address arithmetic; any width / alignment / ragged last source.
// FUSE = true: channel-attention prologue (RCABlock tail, networks.py:447,463-464, fused into the next conv):
//              two input patches (r, x) per stage and B = r * scale + x formed on the way to the MFMA;
//              chunks of 4 channels so that two stages still fit the 160 KiB of LDS.
// NT_  : rows per wave (4, 2 or 1): the tile is 8 NT_ rows x 32 columns.  Small images launch too few 32-row
//        tiles to occupy 256 CUs (a training crop of 2 x 96 x 96 has 18), so the host picks the tile height that
//        minimises rounds x (rows + fixed cost) -- eavsr_conv2d_tile_rows().
template <int KS, int MT, bool VEC, bool FUSE = false, int NT_ = 4>
struct ConvCfg {
  static constexpr int CK = FUSE ? 4 : ChunkOf<KS>::value;
  static constexpr int NIN = FUSE ? 2 : 1;                 // input patches per stage
  static constexpr int NT = NT_;
  static constexpr int TH = 8 * NT_, TW = EAVSR_CONV_TW;
  static constexpr int PAD = KS / 2, KK = KS * KS;
  static constexpr int MARG = VEC ? (PAD ? 4 : 0) : PAD;  // columns staged left/right of the tile
  static constexpr int IH = TH + KS - 1, IW = TW + 2 * MARG;
  static constexpr int CO = 32 * MT;
  static constexpr int IN_ELEMS = CK * IH * IW;
  static constexpr int PIECE = VEC ? 256 : 64;            // floats moved by one wave-level DMA
  static constexpr int IN_SEGS = (IN_ELEMS + PIECE - 1) / PIECE;
  static constexpr int IN_PAD = IN_SEGS * PIECE;
  static constexpr int NW = 8;                             // waves per workgroup (one NT-row strip each)
  static constexpr int IN_IT = (IN_SEGS + NW - 1) / NW;    // pieces per wave
  static constexpr int W_ELEMS = CK * KK * CO;
  static constexpr int W_SEGS = (W_ELEMS + 255) / 256;     // one wave-level dwordx4 DMA = 256 floats
  static constexpr int W_PAD = W_SEGS * 256;
  static constexpr int BUF = NIN * IN_PAD + W_PAD;         // floats per pipeline stage
  static constexpr int W_IT = (W_SEGS + NW - 1) / NW;
  static constexpr int LDS_FLOATS = 2 * BUF + NW * CO + (FUSE ? 256 : 0);
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int KS, int MT, bool VEC, bool FUSE, int NT_>
__global__ __launch_bounds__(512, 2) void conv2d_mfma_kernel(ConvArgs a) {
  using Cfg = ConvCfg<KS, MT, VEC, FUSE, NT_>;
  static_assert(!FUSE || VEC, "the fused channel-attention prologue exists for the 16-byte DMA path only");
  constexpr int CK = Cfg::CK, NT = Cfg::NT, TH = Cfg::TH, TW = Cfg::TW, PAD = Cfg::PAD, KK = Cfg::KK;
  constexpr int MARG = Cfg::MARG, IH = Cfg::IH, IW = Cfg::IW, CO = Cfg::CO;
  constexpr int IN_ELEMS = Cfg::IN_ELEMS, IN_SEGS = Cfg::IN_SEGS, IN_PAD = Cfg::IN_PAD, IN_IT = Cfg::IN_IT;
  constexpr int W_ELEMS = Cfg::W_ELEMS, W_SEGS = Cfg::W_SEGS, W_IT = Cfg::W_IT, BUF = Cfg::BUF;
  constexpr int NW = Cfg::NW, NTHR = 64 * NW;
  // ALL LDS in one array: [stage 0: input patch | weight slab][stage 1: ...][channel-sum scratch]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_red = smem + 2 * BUF;
  float* s_y = s_red + NW * CO;  // FUSE: ca_scale[bn, 0..cin)
  constexpr int NIN = Cfg::NIN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  eavsr_stagger_priority(wave);
  const int l31 = lane & 31, half = lane >> 5;

  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = blockIdx.y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  // chunk cursor over the virtual concatenation of the sources
  int cs = 0, cc0 = 0, cbase = 0;
  int total_chunks = 0;
  for (int s = 0; s < a.n_src; ++s) total_chunks += (a.src_c[s] + CK - 1) / CK;

  // VEC: per-lane byte offsets of this wave's pieces inside one CK-channel slab (0xFFFFFFFF = the piece
  // lies in the zero padding: never moved, its LDS words stay at the zeros written below)
  unsigned voff[VEC ? IN_IT : 1];
  if (VEC) {
    f32x4* z = reinterpret_cast<f32x4*>(smem);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < (2 * BUF) / 4; e += NTHR) z[e] = zero;
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = i * NW + wave;
      const int e4 = seg * 64 + lane;
      const int ci = e4 / (IH * (IW / 4));
      const int rem = e4 - ci * (IH * (IW / 4));
      const int r = rem / (IW / 4);
      const int c4 = rem - r * (IW / 4);
      const int gy = y0 - PAD + r, gx = x0 - MARG + 4 * c4;
      const bool ok = seg < IN_SEGS && e4 < IN_ELEMS / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;
      voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)gy * w + gx) * 4) : 0xFFFFFFFFu;
    }
    if (FUSE) {
      for (int c = tid; c < a.cin; c += NTHR) s_y[c] = a.ca_scale[(size_t)bn * a.cin + c];
    }
    __syncthreads();  // the zero fill is complete before any DMA may land
  }

  // ---- LDS-DMA of one chunk (input patch + weight slab) into pipeline stage `stage` ----------------
  // global_load_lds: per-lane global address, LDS destination = wave-uniform base + lane * size, no
  // VGPR staging and no ds_write.  Wave w moves pieces w, w+NW, w+2NW, ...
  auto issue_chunk = [&](int stage) {
    float* s_in = smem + stage * BUF;
    float* s_w = s_in + NIN * IN_PAD;
    const int sc = a.src_c[cs];
    const float* sp = a.src[cs] + ((size_t)bn * sc + cc0) * plane;
    if (VEC) {
#pragma unroll
      for (int i = 0; i < IN_IT; ++i) {
        const int seg = i * NW + wave;
        if (voff[i] != 0xFFFFFFFFu) {
          __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(sp) + voff[i]),
                                           (lptr_t)(s_in + seg * 256), 16, 0, 0);
          if (FUSE) {
            const float* xp = a.ca_x + ((size_t)bn * sc + cc0) * plane;
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(xp) + voff[i]),
                                             (lptr_t)(s_in + IN_PAD + seg * 256), 16, 0, 0);
          }
        }
      }
    } else {
      const int nvalid = min(CK, sc - cc0);
      // rolled on purpose: unrolling makes hipcc materialise every lane address / mask up front (spills)
#pragma unroll 1
      for (int i = 0; i < IN_IT; ++i) {
        const int seg = i * NW + wave;
        if (seg < IN_SEGS) {  // wave-uniform
          const int e = seg * 64 + lane;
          const int ci = e / (IH * IW);
          const int rem = e - ci * (IH * IW);
          const int r = rem / IW;
          const int cc = rem - r * IW;
          const int gy = y0 - PAD + r, gx = x0 - MARG + cc;
          const bool ok = e < IN_ELEMS && ci < nvalid && gy >= 0 && gy < h && gx >= 0 && gx < w;
          const float* p = ok ? sp + ((size_t)ci * plane + (size_t)gy * w + gx) : g_zero_word;
          __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(s_in + seg * 64), 4, 0, 0);
        }
      }
    }
    const char* wsrc = reinterpret_cast<const char*>(
        a.wp + ((size_t)cot * a.cin_pad + (size_t)(cbase + cc0)) * (KK * CO));
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = i * NW + wave;
      if (seg < W_SEGS) {  // wave-uniform
        const unsigned e4 = (unsigned)min(seg * 64 + lane, W_ELEMS / 4 - 1);  // a ragged tail re-reads its last 16 B
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + e4 * 16u), (lptr_t)(s_w + seg * 256), 16, 0, 0);
      }
    }
  };
  auto advance = [&]() {
    cc0 += CK;
    if (cc0 >= a.src_c[cs]) {
      cbase += a.src_c[cs];
      ++cs;
      cc0 = 0;
    }
  };

  issue_chunk(0);
  for (int it = 0; it < total_chunks; ++it) {
    // chunk `it` has landed (this wave's DMAs: vmcnt; the other waves': the barrier); everyone is also
    // done computing on the stage that the next DMA overwrites.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + 1 < total_chunks) {
      advance();
#ifndef EAVSR_CONV_EXP_NODMA   // timing ablation only (tools/gpu_conv_ablate.py): 173 -> 162 us without the DMA,
                               // 167 us without the output stores.  Handing the pieces out from inside the tap loop
                               // (2 per tap) instead of here changes nothing: the cost is per piece, not the burst.
      issue_chunk((it + 1) & 1);  // in flight behind the MFMA loop below
#endif
    }
    const float* bin = smem + (it & 1) * BUF + half * (IH * IW) + (wave * NT) * IW + (MARG - PAD) + l31;
    const float* ain = smem + (it & 1) * BUF + NIN * IN_PAD + half * (KK * CO) + l31;
    float yv[FUSE ? CK / 2 : 1];
    if (FUSE) {
      const int cur_c0 = it * CK;  // single source: chunk `it` covers channels it*CK ..
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) yv[cp] = s_y[cur_c0 + 2 * cp + half];
      if (a.ca_out != nullptr && cot == 0) {
        // side output: the effective input of this chunk's channels for this wave's 4 x 32 interior pixels
        const float* rin = smem + (it & 1) * BUF + (wave * NT + PAD) * IW + MARG + l31;
        const int gxs = x0 + l31;
#pragma unroll
        for (int j = 0; j < (CK * NT) / 2; ++j) {
          const int item = 2 * j + half;
          const int ci = item / NT, t = item - ci * NT;
          const int gys = y0 + wave * NT + t;
          const float v = fmaf(rin[ci * (IH * IW) + t * IW], s_y[cur_c0 + ci], rin[IN_PAD + ci * (IH * IW) + t * IW]);
          if (gys < h && gxs < w)
            a.ca_out[((size_t)bn * a.cin + cur_c0 + ci) * plane + (size_t)gys * w + gxs] = v;
        }
      }
    }
#pragma unroll(KS <= 3 ? KS : 1)
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
          float av[MT], bv[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m) av[m] = ain[(cp * 2 * KK + tap) * CO + m * 32];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int o = cp * 2 * (IH * IW) + (t + ky) * IW + kx;
            bv[t] = FUSE ? fmaf(bin[o], yv[cp], bin[IN_PAD + o]) : bin[o];
          }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[t], acc[m][t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const int gx = x0 + l31;
  const bool xok = gx < w;
  float csum[MT][16];
  // the lane's 16 MT bias values in one batch (a uniform branch on the nullable pointer): loaded where they are used, each sat
  // behind its own branch and a vmcnt(0) -- 32 dependent round trips in the epilogue of a workgroup
  float bvs[MT][16];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) bvs[m][r] = 0.f;
  if (a.bias) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bvs[m][r] = a.bias[min(co, a.cout - 1)];      // (clamped: rows >= cout are never stored)
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool cok = co < a.cout;
      const float b = bvs[m][r];
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gy = y0 + wave * NT + t;
        float v = acc[m][t][r] + b;
        v = eavsr_act(v, a.act == EAVSR_ACT_RELU ? 0.f : a.act == EAVSR_ACT_LRELU ? a.slope : 1.f);   // branch-free: max(v, v s), 0 <= s <= 1
#ifdef EAVSR_CONV_EXP_NOSTORE   // timing ablation only (tools/gpu_conv_ablate.py)
        if (cok && xok && gy < h && v == 12345.678f) {
#else
        if (cok && xok && gy < h) {
#endif
          const size_t o = ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx;
          sum += v;
          if (a.act == EAVSR_ACT_RELU_MASK) v = a.residual[o] > 0.f ? v : 0.f;      // ReLU's backward mask (the forward output)
          else if (a.residual) v += a.residual[o];
          a.out[o] = v;
        }
      }
      csum[m][r] = sum;
    }
  }
  if (a.chan_partial) {
    // reduce over the 32 pixel lanes of each half-wave, then over the 4 waves (fixed order)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = csum[m][r];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (l31 == 0) s_red[wave * CO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = v;
      }
    __syncthreads();
    if (tid < CO) {
      const int co = cot * CO + tid;
      if (co < a.cout) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < NW; ++k) v += s_red[k * CO + tid];
        const int tile = ty * a.tiles_x + tx;
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + tile) * a.cout + co] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3, 8-row tiles (NT = 1), 16-byte DMA path: the small-problem variant.  A training crop (2 x 96 x 96) or a pyramid
// level launches fewer tiles than there are CUs, each wave has only 36 MFMAs per 8-channel chunk (1 us), and with the
// two-stage pipeline above every chunk exposed the LDS-DMA latency: 4.4 us per chunk measured against 2 us of MFMAs
// (tools/gpu_conv_small_sweep.py).  Same tile geometry, operands, packed weights and epilogue; what changes:
//   * FOUR LDS stages; the DMA of chunk it+3 is issued while chunk it is multiplied, behind a COUNTED vmcnt that leaves
//     the pieces of the two chunks in between in flight
//   * every DMA piece is always issued (zero-padding lanes copy 16 zero bytes, surplus slots repeat the last piece), so
//     the number of outstanding pieces per wave and chunk is a compile-time constant and no LDS zero-fill is needed
// ---------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_f4[4];

template <int MT>
struct SmallCfg {
  static constexpr int CK = 8, NW = 8, NSTAGE = 4;
  static constexpr int TH = 8, TW = EAVSR_CONV_TW, MARG = 4, PAD = 1, KK = 9;
  static constexpr int IH = TH + 2, IW = TW + 2 * MARG;        // 10 x 40
  static constexpr int CO = 32 * MT;
  static constexpr int IN_ELEMS = CK * IH * IW;                // 3200 floats
  static constexpr int IN_SEGS = (IN_ELEMS + 255) / 256;       // 13 pieces (the last one half)
  static constexpr int IN_PAD = IN_SEGS * 256;
  static constexpr int IN_IT = (IN_SEGS + NW - 1) / NW;        // 2 per wave
  static constexpr int W_ELEMS = CK * KK * CO;
  static constexpr int W_SEGS = (W_ELEMS + 255) / 256;         // 18 pieces (MT = 2)
  static constexpr int W_IT = (W_SEGS + NW - 1) / NW;          // 3 per wave
  static constexpr int PIECES = IN_IT + W_IT;                  // DMA instructions per wave and chunk
  static constexpr int BUF = IN_PAD + W_SEGS * 256;
  static constexpr int LDS_FLOATS = NSTAGE * BUF + NW * CO;
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

// SPLIT: the weight is packed for the 64-wide output tile but each workgroup computes one 32-channel half of it
// (blockIdx.y = 2 * tile + half) - twice the workgroups for problems that leave most of the 256 CUs idle.
template <int MT, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_small_kernel(ConvArgs a) {
  static_assert(!SPLIT || MT == 1, "a split tile is one 32-channel M-tile");
  using Cfg = SmallCfg<MT>;
  constexpr int CK = Cfg::CK, NW = Cfg::NW, NSTAGE = Cfg::NSTAGE, TH = Cfg::TH, TW = Cfg::TW, MARG = Cfg::MARG, PAD = Cfg::PAD;
  constexpr int KK = Cfg::KK, IH = Cfg::IH, IW = Cfg::IW, CO = Cfg::CO, IN_ELEMS = Cfg::IN_ELEMS, IN_SEGS = Cfg::IN_SEGS;
  constexpr int IN_PAD = Cfg::IN_PAD, IN_IT = Cfg::IN_IT, W_ELEMS = Cfg::W_ELEMS, W_SEGS = Cfg::W_SEGS, W_IT = Cfg::W_IT;
  constexpr int PIECES = Cfg::PIECES, BUF = Cfg::BUF;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_red = smem + NSTAGE * BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  eavsr_stagger_priority(wave);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = eavsr_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int bn = bid / a.tiles_y;
  const int cot = SPLIT ? blockIdx.y >> 1 : blockIdx.y;
  const int co_base = SPLIT ? blockIdx.y * 32 : blockIdx.y * CO;   // first output channel of this workgroup
  constexpr int WCO = SPLIT ? 64 : CO;                              // row length of the packed weight
  const int y0 = ty * TH, x0 = tx * TW;
  const int h = a.h, w = a.w;
  const size_t plane = (size_t)h * w;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  int cs = 0, cc0 = 0, cbase = 0;   // cursor of the ISSUE stream over the virtual concatenation of the sources
  int total_chunks = 0;
  for (int s = 0; s < a.n_src; ++s) total_chunks += a.src_c[s] / CK;

  unsigned voff[IN_IT];
#pragma unroll
  for (int i = 0; i < IN_IT; ++i) {
    const int seg = min(i * NW + wave, IN_SEGS - 1);
    const int e4 = seg * 64 + lane;
    const int ci = e4 / (IH * (IW / 4));
    const int rem = e4 - ci * (IH * (IW / 4));
    const int r = rem / (IW / 4);
    const int c4 = rem - r * (IW / 4);
    const int gy = y0 - PAD + r, gx = x0 - MARG + 4 * c4;
    const bool ok = e4 < IN_ELEMS / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w;
    voff[i] = ok ? (unsigned)(((size_t)ci * plane + (size_t)gy * w + gx) * 4) : 0xFFFFFFFFu;
  }
  auto issue_chunk = [&](int stage) {
    float* s_in = smem + stage * BUF;
    float* s_w = s_in + IN_PAD;
    const int sc = a.src_c[cs];
    const char* sp = reinterpret_cast<const char*>(a.src[cs] + ((size_t)bn * sc + cc0) * plane);
    const char* zp = reinterpret_cast<const char*>(g_zero_f4);
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      const int seg = min(i * NW + wave, IN_SEGS - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(voff[i] != 0xFFFFFFFFu ? sp + voff[i] : zp), (lptr_t)(s_in + seg * 256), 16, 0, 0);
    }
    const char* wsrc = reinterpret_cast<const char*>(a.wp + ((size_t)cot * a.cin_pad + (size_t)(cbase + cc0)) * (KK * WCO) +
                                                     (SPLIT ? (blockIdx.y & 1) * 32 : 0));
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int seg = min(i * NW + wave, W_SEGS - 1);
      const unsigned e4 = (unsigned)min(seg * 64 + lane, W_ELEMS / 4 - 1);
      const unsigned goff = SPLIT ? (e4 >> 3) * (WCO * 4u) + (e4 & 7) * 16u : e4 * 16u;   // 8 float4 per 32-float row
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + goff), (lptr_t)(s_w + seg * 256), 16, 0, 0);
    }
    cc0 += CK;
    if (cc0 >= a.src_c[cs]) {
      cbase += a.src_c[cs];
      ++cs;
      cc0 = 0;
    }
  };

#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i)
    if (i < total_chunks) issue_chunk(i);
  for (int it = 0; it < total_chunks; ++it) {
    // chunk `it` must have landed; the pieces of the (up to) two younger chunks stay in flight
    const int younger = min(total_chunks - 1 - it, NSTAGE - 2);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... for every wave; and everyone is done with the stage chunk it+3 overwrites (chunk it-1's)
    if (it + NSTAGE - 1 < total_chunks) issue_chunk((it + NSTAGE - 1) % NSTAGE);
    const float* bin = smem + (it % NSTAGE) * BUF + half * (IH * IW) + wave * IW + (MARG - PAD) + l31;
    const float* ain = smem + (it % NSTAGE) * BUF + IN_PAD + half * (KK * CO) + l31;
    // the 12 operands of tap t+1 (4 channel pairs x (2 M-tiles + 1 row)) are read before the 8 MFMAs of tap t: with one
    // row per wave every MFMA pair needs fresh A operands, and a read / wait / multiply sequence per pair left the
    // matrix pipe idle half of the time
    float av[2][CK / 2][MT], bv[2][CK / 2];
    auto load_tap = [&](int tap, int slot) __attribute__((always_inline)) {
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) {
        bv[slot][cp] = bin[cp * 2 * (IH * IW) + ky * IW + kx];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[slot][cp][m] = ain[(cp * 2 * KK + tap) * CO + m * 32];
      }
    };
    load_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      if (tap + 1 < KK) load_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tap & 1][cp][m], bv[tap & 1][cp], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (CK / 2) * (MT + 1), 0);   // the next tap's reads first
      __builtin_amdgcn_sched_group_barrier(0x008, (CK / 2) * MT, 0);         // then this tap's MFMAs
    }
  }

  // ---- epilogue (as conv2d_mfma_kernel with one row per wave) ------------------------------------------
  // Every bias / residual load is issued before the first store: vmcnt counts loads and stores in one queue and
  // the output pointer may alias, so a load placed after a store waits for that store's acknowledgement - sixteen
  // serialised round trips per wave cost more than the whole main loop of a 64-channel launch.
  const int gx = x0 + l31, gy = y0 + wave;
  const bool pok = gx < w && gy < h;
  float bb[MT][16], rr[MT][16];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool cok = co < a.cout;
      bb[m][r] = (cok && a.bias) ? a.bias[co] : 0.f;
      rr[m][r] = (cok && pok && a.residual) ? a.residual[((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx] : 0.f;
    }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int co = co_base + col;
      const bool cok = co < a.cout;
      float v = acc[m][r] + bb[m][r];
      v = eavsr_act(v, a.act == EAVSR_ACT_RELU ? 0.f : a.act == EAVSR_ACT_LRELU ? a.slope : 1.f);   // branch-free: max(v, v s), 0 <= s <= 1
      float sum = 0.f;
      if (cok && pok) {
        const size_t o = ((size_t)bn * a.cout + co) * plane + (size_t)gy * w + gx;
        sum = v;
        // EAVSR_ACT_RELU_MASK: `residual` is the forward output of a ReLU whose backward mask this input-gradient convolution applies
        a.out[o] = a.act == EAVSR_ACT_RELU_MASK ? (rr[m][r] > 0.f ? v : 0.f) : v + rr[m][r];
      }
      if (a.chan_partial) {
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 1);
        if (l31 == 0) s_red[wave * CO + col] = sum;
      }
    }
  }
  if (a.chan_partial) {
    __syncthreads();
    if (tid < CO) {
      const int co = co_base + tid;
      if (co < a.cout) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < NW; ++k) v += s_red[k * CO + tid];
        a.chan_partial[((size_t)bn * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx) * a.cout + co] = v;
      }
    }
  }
}

template <int MT, bool SPLIT = false>
int launch_small(const ConvArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = SmallCfg<MT>;
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_small_kernel<MT, SPLIT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv2d: hipFuncSetAttribute(%zu B of LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  if (SPLIT) grid.y *= 2;
  hipLaunchKernelGGL((conv3x3_small_kernel<MT, SPLIT>), grid, dim3(64 * Cfg::NW), Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("conv2d");
}

// weight (cout,cin,k,k) -> [cout_tile][cin_pad][k*k][CO], zero padded
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int cout, int cin,
                                   int kk, int cin_pad, int CO, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % CO);
  long r = i / CO;
  const int tap = (int)(r % kk);
  r /= kk;
  const int ci = (int)(r % cin_pad);
  const int cot = (int)(r / cin_pad);
  const int co = cot * CO + col;
  p[i] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * kk + tap] : 0.f;
}

inline int chunk_of(int ks) { return ks == 1 ? 16 : ks == 3 ? 8 : ks == 5 ? 4 : 2; }
inline int co_tile_of(int cout) { return cout <= 32 ? 32 : 64; }

template <int KS, int MT, bool VEC, bool FUSE = false, int NT_ = 4>
int launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = ConvCfg<KS, MT, VEC, FUSE, NT_>;
  static eavsr::PerDeviceOnce once_pd;   // hipFuncSetAttribute is per device: once per (kernel, device)
  const int dev_ = eavsr::current_device();
  std::once_flag& once = once_pd.flag[dev_];
  static hipError_t attr_err_pd[eavsr::kMaxDevices] = {};
  hipError_t& attr_err = attr_err_pd[dev_];
  std::call_once(once, [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<KS, MT, VEC, FUSE, NT_>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  });
  if (attr_err != hipSuccess) {
    eavsr::set_error("conv2d: hipFuncSetAttribute(%zu B of LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(attr_err));
    return (int)attr_err;
  }
  hipLaunchKernelGGL((conv2d_mfma_kernel<KS, MT, VEC, FUSE, NT_>), grid, dim3(64 * Cfg::NW), Cfg::LDS_BYTES, st, a);
  return eavsr::launch_status("conv2d");
}

template <int KS, int NT_>
int launch_ks(const ConvArgs& a, dim3 grid, int CO, bool vec, hipStream_t st) {
  if (vec) return CO == 32 ? launch_one<KS, 1, true, false, NT_>(a, grid, st) : launch_one<KS, 2, true, false, NT_>(a, grid, st);
  return CO == 32 ? launch_one<KS, 1, false, false, NT_>(a, grid, st) : launch_one<KS, 2, false, false, NT_>(a, grid, st);
}

// rows per wave for an (n, h, w) problem: minimise rounds over the 256 CUs x (rows per tile + fixed per-tile cost).
// 7x7 (SPyNet only) keeps the 32-row tile.
inline int tile_rows_of(int n, int h, int w, int ksize) {
  if (ksize == 7) return 4;
  int best = 4;
  long best_cost = -1;
  for (int nt = 4; nt >= 1; nt >>= 1) {
    const long tiles = (long)n * eavsr::cdiv(h, 8 * nt) * eavsr::cdiv(w, EAVSR_CONV_TW);
    const long rounds = (tiles + 255) / 256;
    const long cost = rounds * (2 * nt + 1);   // fixed cost = half a row-unit
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nt; }
  }
  return best;
}

template <int KS>
int launch_nt(const ConvArgs& a, dim3 grid, int CO, bool vec, int nt, hipStream_t st) {
  if (nt == 1) return launch_ks<KS, 1>(a, grid, CO, vec, st);
  if (nt == 2) return launch_ks<KS, 2>(a, grid, CO, vec, st);
  return launch_ks<KS, 4>(a, grid, CO, vec, st);
}

}  // namespace

extern "C" int32_t eavsr_conv2d_ck(int32_t ksize) { return chunk_of(ksize); }

extern "C" int32_t eavsr_conv2d_tile_rows(int32_t n, int32_t h, int32_t w, int32_t ksize) {
  return 8 * tile_rows_of(n > 0 ? n : 1, h, w, ksize);
}

extern "C" int32_t eavsr_conv2d_tiles(int32_t n, int32_t h, int32_t w, int32_t ksize) {
  return eavsr::cdiv(h, eavsr_conv2d_tile_rows(n, h, w, ksize)) * eavsr::cdiv(w, EAVSR_CONV_TW);
}

extern "C" int64_t eavsr_packed_weight_elems(int32_t cout, int32_t cin, int32_t ksize) {
  if (cout <= 0 || cin <= 0 || !(ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7)) return -1;
  const int CO = co_tile_of(cout), ck = chunk_of(ksize);
  const int64_t cots = eavsr::cdiv(cout, CO), cin_pad = (int64_t)eavsr::cdiv(cin, ck) * ck;
  return cots * cin_pad * ksize * ksize * CO;
}

extern "C" int eavsr_pack_conv_weight_f32(const float* weight, float* packed, int32_t cout, int32_t cin,
                                          int32_t ksize, void* stream) {
  EAVSR_REQUIRE(weight && packed, -1, "pack_conv_weight: NULL pointer");
  const int64_t total = eavsr_packed_weight_elems(cout, cin, ksize);
  EAVSR_REQUIRE(total > 0, -1, "pack_conv_weight: bad shape cout=%d cin=%d k=%d", cout, cin, ksize);
  const int CO = co_tile_of(cout), ck = chunk_of(ksize);
  const int cin_pad = eavsr::cdiv(cin, ck) * ck;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     eavsr::as_stream(stream), weight, packed, cout, cin, ksize * ksize, cin_pad, CO, (long)total);
  return eavsr::launch_status("pack_conv_weight");
}

extern "C" int eavsr_conv2d_f32(const eavsr_conv2d_desc* d, void* stream) {
  EAVSR_REQUIRE(d != nullptr, -1, "conv2d: NULL descriptor");
  EAVSR_REQUIRE(d->n_src >= 1 && d->n_src <= 5, -1, "conv2d: n_src %d not in 1..5", d->n_src);
  EAVSR_REQUIRE(d->ksize == 1 || d->ksize == 3 || d->ksize == 5 || d->ksize == 7, -2,
                "conv2d: kernel size %d unsupported (1,3,5,7)", d->ksize);
  EAVSR_REQUIRE(d->weight_packed && d->out, -1, "conv2d: NULL weight/out");
  EAVSR_REQUIRE(d->out_shuffle == 0 && d->res_scale == nullptr && d->border_pieces == nullptr && d->sum_mul == nullptr, -2, "conv2d: the pixel-shuffle and scaled-residual epilogues exist in eavsr_conv3x3_wino4_f32 only");
  EAVSR_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, -1, "conv2d: bad dims");
  EAVSR_REQUIRE(d->act >= 0 && d->act <= EAVSR_ACT_RELU_MASK, -1, "conv2d: act %d", d->act);
  EAVSR_REQUIRE(d->act != EAVSR_ACT_RELU_MASK || (d->residual != nullptr && d->chan_partial == nullptr && d->ca_scale == nullptr), -1,
                "conv2d: EAVSR_ACT_RELU_MASK takes the mask source in `residual` (no channel sums, no fused prologue)");
  EAVSR_REQUIRE(d->act != EAVSR_ACT_LRELU || (d->slope >= 0.f && d->slope <= 1.f), -2,
                "conv2d: leaky-ReLU slope %g outside [0, 1] (the epilogue evaluates max(v, slope v))", (double)d->slope);
  const int ck = chunk_of(d->ksize);
  ConvArgs a;
  int csum = 0;
  for (int s = 0; s < 5; ++s) {
    a.src[s] = s < d->n_src ? d->src[s] : nullptr;
    a.src_c[s] = s < d->n_src ? d->src_c[s] : 0;
    if (s < d->n_src) {
      EAVSR_REQUIRE(d->src[s] != nullptr && d->src_c[s] > 0, -1, "conv2d: source %d is empty", s);
      EAVSR_REQUIRE(s == d->n_src - 1 || d->src_c[s] % ck == 0, -2,
                    "conv2d: source %d has %d channels, not a multiple of %d (only the last may be ragged)", s,
                    d->src_c[s], ck);
      csum += d->src_c[s];
    }
  }
  EAVSR_REQUIRE(csum == d->cin, -1, "conv2d: sources sum to %d channels, cin = %d", csum, d->cin);
  if (d->n == 0) return 0;
  const int CO = co_tile_of(d->cout);
  a.n_src = d->n_src;
  a.wp = d->weight_packed;
  a.bias = d->bias;
  a.residual = d->residual;
  a.out = d->out;
  a.chan_partial = d->chan_partial;
  a.ca_scale = d->ca_scale;
  a.ca_x = d->ca_x;
  a.ca_out = d->ca_out;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
  a.cin_pad = eavsr::cdiv(d->cin, ck) * ck;
  const int nt = tile_rows_of(d->n, d->h, d->w, d->ksize);
  a.tiles_x = eavsr::cdiv(d->w, EAVSR_CONV_TW);
  a.tiles_y = eavsr::cdiv(d->h, 8 * nt);
  a.act = d->act;
  a.slope = d->slope;
  const long blocks = (long)a.tiles_x * a.tiles_y * d->n;
  EAVSR_REQUIRE(blocks < (1L << 31), -1, "conv2d: too many tiles");
  EAVSR_REQUIRE((long)d->h * d->w * 16 < (1L << 31), -1, "conv2d: image plane too large for 32-bit tile offsets");
  dim3 grid((unsigned)blocks, eavsr::cdiv(d->cout, CO));
  hipStream_t st = eavsr::as_stream(stream);
  // 16-byte DMA path: whole float4 groups are either inside or outside the image and every piece is aligned
  bool vec = (d->w % 4) == 0;
  for (int s = 0; s < d->n_src; ++s)
    vec = vec && (((uintptr_t)d->src[s]) & 15) == 0 && (d->src_c[s] % ck) == 0;
  if (d->ca_scale != nullptr) {
    EAVSR_REQUIRE(d->ca_x != nullptr, -1, "conv2d: ca_scale without ca_x");
    EAVSR_REQUIRE(d->ksize == 3 && d->n_src == 1 && vec && (((uintptr_t)d->ca_x) & 15) == 0 && d->cin % 4 == 0 &&
                      d->cin <= 256 && CO == 64 && d->cout <= 64,
                  -2, "conv2d: the fused channel-attention prologue needs a single 16-byte aligned source, k = 3, "
                      "w %% 4 == 0, cin %% 4 == 0, cin <= 256 and 33..64 output channels; use eavsr_scale_residual_f32 + "
                      "a plain conv otherwise");
    EAVSR_REQUIRE(nt == 4, -2, "conv2d: the fused channel-attention prologue exists for the 32-row tile only "
                               "(this problem size runs %d-row tiles)", 8 * nt);
    return launch_one<3, 2, true, true>(a, grid, st);
  }
  EAVSR_REQUIRE(d->ca_x == nullptr && d->ca_out == nullptr, -1, "conv2d: ca_x / ca_out without ca_scale");
  switch (d->ksize) {
    case 1: return launch_nt<1>(a, grid, CO, vec, nt, st);
    case 3:
      if (nt == 1 && vec) {
        if (CO == 32) return launch_small<1>(a, grid, st);
        // fewer than half of the CUs busy: one 32-channel half tile per workgroup
        if (blocks * grid.y <= 128) return launch_small<1, true>(a, grid, st);
        return launch_small<2>(a, grid, st);
      }
      return launch_nt<3>(a, grid, CO, vec, nt, st);
    case 5: return launch_nt<5>(a, grid, CO, vec, nt, st);
    default: return launch_ks<7, 4>(a, grid, CO, vec, st);
  }
}
