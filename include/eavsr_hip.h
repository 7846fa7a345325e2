/*
 * eavsr_hip.h -- C ABI of libeavsr_hip.so: MI355X (gfx950) kernels for the EAVSR inter-frame
 * alignment + feature-propagation hot path.
 *
 * Boundary contract:
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - the caller allocates and owns every buffer (inputs, outputs, workspaces); the library
 *     never allocates or frees device memory and keeps no pointer after return.
 *   - every call is asynchronous on the given stream; no internal synchronisation, no
 *     default-stream use.  Safe from several host threads (no mutable global state besides a
 *     thread-local error string).
 *   - return value: 0 = OK, <0 = argument / shape / unsupported-configuration error found on
 *     the host (nothing was launched), >0 = hipError_t of the failed launch.
 *     eavsr_last_error() returns the message of the last failing call on this thread.
 *   - all tensors are contiguous fp32 NCHW unless a parameter says otherwise.
 *
 * Each entry point names the reference interface (file:line under /root/reference) it replaces.
 */
#ifndef EAVSR_HIP_H
#define EAVSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAVSR_ABI_VERSION 30

/* activation codes for eavsr_conv2d_f32 */
#define EAVSR_ACT_NONE 0
#define EAVSR_ACT_RELU 1
#define EAVSR_ACT_LRELU 2 /* negative slope in desc.slope */
/* eavsr_conv2d_f32 only (ABI 26): out = desc.residual > 0 ? conv + bias : 0 -- the backward mask of a ReLU whose forward output is
 * given in `residual` (which is NOT added), fused into the input-gradient convolution that produces the masked gradient
 * (RCABlock's conv -> ReLU -> conv, models/networks.py:461-462, in loss.backward()) */
#define EAVSR_ACT_RELU_MASK 3

/* padding modes of eavsr_flow_warp_f32 (torch grid_sample padding_mode) */
#define EAVSR_PAD_ZEROS 0
#define EAVSR_PAD_BORDER 1
#define EAVSR_PAD_REFLECTION 2
/* or-ed into padding_mode for the grid_sample variants off the reference's path (networks.py:699-739 forwards
 * `interpolation` and `align_corners` to F.grid_sample): nearest-neighbour sampling, align_corners=False */
#define EAVSR_WARP_NEAREST 0x10
#define EAVSR_WARP_NO_ALIGN_CORNERS 0x20

/* flow layouts of eavsr_flow_warp_f32 */
#define EAVSR_FLOW_NCHW 0 /* networks.py:699-739      (n,2,h,w): ch0 = x disp, ch1 = y disp */
#define EAVSR_FLOW_NHWC 1 /* eavsrp_model.py:587-626  (n,h,w,2): [..,0] = x,  [..,1] = y     */

int eavsr_abi_version(void);
/* "eavsr-hip <semver> gfx950" */
const char* eavsr_version(void);
/* message of the last failing call on the calling thread ("" if none) */
const char* eavsr_last_error(void);
/* 0: the product build (every entry point above the EXPERIMENTAL section); 1: the lab build (ABI 28) */
int eavsr_lab_build(void);

/* Verifies on the device that v_mfma_f32_32x32x2_f32 has the operand / accumulator lane layout the
 * kernels assume (asymmetric integer data).  scratch: >= 8192 floats of device memory.  The result
 * (0 = layout OK, otherwise the number of mismatching elements) is written to scratch[0] as float;
 * the caller synchronises and reads it. */
int eavsr_selftest_mfma_f32(float* scratch, void* stream);

/* ---- a1 / a2: flow_warp -------------------------------------------------------------------
 * replaces networks.flow_warp (models/networks.py:699-739) and eavsrp_model.flow_warp
 * (models/eavsrp_model.py:587-626): bilinear grid_sample, align_corners=True, of x by a pixel-unit
 * flow.  out[n,c,y,x] = bilinear(x[n,c], y + fy, x + fx), corner-wise zero padding (ZEROS) or
 * coordinate clamping (BORDER).  flow2 (nullable, same layout) is added to flow first
 * (networks.py:610,615 warp by a sum of two flows).  The modes the reference never uses on the path
 * (REFLECTION, EAVSR_WARP_NEAREST, EAVSR_WARP_NO_ALIGN_CORNERS) run a generic kernel; forward only.      */
int eavsr_flow_warp_f32(const float* x, const float* flow, const float* flow2, float* out,
                        int32_t n, int32_t c, int32_t h, int32_t w,
                        int32_t flow_layout, int32_t padding_mode, void* stream);

/* networks.py:621 + :623: two feature maps (the neighbour's features and the propagated features) warped by the SAME
 * NCHW flow (+ optional flow2, summed first), bilinear, zeros padding, align_corners=True, in ONE launch.  outa is NCHW;
 * outb is NCHW (outb_il8 = 0), the IL8 layout [n][c/8][h][w][8] that eavsr_dcnv2_il_f32 samples from (outb_il8 = 1,
 * c % 8 == 0), or IL8 rounded to fp16 (outb_il8 = 2) / bf16 (outb_il8 = 3) for eavsr_dcnv2_il16. */
int eavsr_flow_warp_pair_f32(const float* xa, const float* xb, const float* flow, const float* flow2, float* outa,
                             float* outb, int32_t n, int32_t c, int32_t h, int32_t w, int32_t outb_il8, void* stream);


/* ---- a7: DCNv2 ------------------------------------------------------------------------------
 * replaces mmcv.ops.modulated_deform_conv2d as called at models/networks.py:627-630 (module
 * parameters from the ModulatedDeformConv2d base class, networks.py:575-583).
 * Fused sampler + modulation + (cout x cin*9) contraction + bias; no column buffer in HBM.
 * offset: (n, dg*18, h, w) channel g*18 + 2k + {0: dy, 1: dx};  mask: (n, dg*9, h, w) channel g*9+k.
 * weight_packed: from eavsr_pack_conv_weight_f32(weight(cout,cin,3,3)).
 * Supported: 3x3, stride 1, pad 1, dilation 1, groups 1, (cin/dg) % 8 == 0, cout <= 64 per launch
 * tile (any cout; tiles of 64).  Anything else returns -2.                                       */
int eavsr_dcnv2_f32(const float* x, const float* offset, const float* mask,
                    const float* weight_packed, const float* bias, float* out,
                    int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                    int32_t deform_groups, void* stream);

/* Same operation with the 576-deep fp32 contraction carried by the bf16 matrix pipe ("bf16x9"): every fp32 operand
 * is split EXACTLY into three bf16 terms (hi + mid + lo reproduces all 24 significant bits) and all nine partial
 * products are accumulated in fp32 -- no operand is rounded; only the accumulation order differs from the fma
 * chain of eavsr_dcnv2_f32.  weight_x9: eavsr_dcn_weight_x9_bytes(cout, cin) bytes written by
 * eavsr_pack_dcn_weight_x9 from weight(cout,cin,3,3).  Additionally requires w % 4 == 0 and a 16-byte aligned x
 * (returns -2 otherwise: call eavsr_dcnv2_f32).  The packed slabs are what eavsr_dcnv2_il_f32 reads; the NCHW kernel itself
 * (eavsr_dcnv2_f32x9) is in the EXPERIMENTAL section at the end of this header (lab build only). */
int64_t eavsr_dcn_weight_x9_bytes(int32_t cout, int32_t cin);
int eavsr_pack_dcn_weight_x9(const float* weight, void* weight_x9, int32_t cout, int32_t cin, void* stream);

/* Round-2 hot-path form of the same operation (csrc/dcnv2_il.hip).  Differences from eavsr_dcnv2_f32x9:
 *   x_il8   the sampled feature map in "IL8" layout [n][cin/8][h][w][8] (the 8 channels of a deformable group interleaved
 *           per pixel): written directly by eavsr_flow_warp_pair_f32 (the warp of networks.py:623 that precedes the call)
 *           or by eavsr_nchw_to_il8_f32 from a plain NCHW tensor.
 *   heads   0: offset (n, dg*18, h, w) / mask (n, dg*9, h, w) exactly as mmcv's signature;
 *           2 (eavsr_dcnv2_il2_f32 only): as 1, but the 9 dg mask channels already went through the sigmoid (eavsr_conv_f32x6 with
 *              sigmoid_from = 6 dg);
 *           1: `offset_or_heads` is the (n, 15*dg, h, w) output of AdaptBlockOffset's three 5x5 convolutions
 *              (transform g*4+{0..3}, translation 4dg + g*2+{0,1}, mask logits 6dg + g*9+k) and the kernel applies
 *              networks.py:302-315 itself (offset = T.R - R + t, mask = sigmoid): de_offset / mask never exist in HBM.
 *   nprod   9: exact bf16x9 contraction; 6: the three partial products below 2^-23 of the result are dropped
 *           (error per product <= one fp32 rounding).
 * weight_x9 from eavsr_pack_dcn_weight_x9.  Requires (cin/dg) % 8 == 0, 16-byte aligned x_il8; any h, w. */
int eavsr_nchw_to_il8_f32(const float* x, float* out_il8, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
int eavsr_dcnv2_il_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_x9,
                       const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                       int32_t deform_groups, int32_t nprod, int32_t heads, void* stream);

/* Round-4 schedule of eavsr_dcnv2_il_f32 (csrc/dcnv2_il2.hip): same operation, arguments, operand splits and products;
 * the two 8-channel groups of a pair are contracted as 18 taps = nine full k-steps (no padded tenth half k-step) in one
 * software pipeline that runs across group, pair and tile boundaries.  Replaces the modulated_deform_conv2d call of
 * models/networks.py:627-630 (heads = 1: together with networks.py:302-315).  weight_il2: eavsr_dcn_weight_il2_bytes(cout,
 * cin) bytes written by eavsr_pack_dcn_weight_il2 from weight(cout,cin,3,3).  Requires cin % 16 == 0 in addition to the
 * requirements of eavsr_dcnv2_il_f32 (-2 otherwise: call eavsr_dcnv2_il_f32).  Results equal eavsr_dcnv2_il_f32's up to
 * the re-association of the k-steps (and, heads = 1, <= 1 ulp of the mask: the sigmoid's reciprocal is v_rcp_f32). */
int64_t eavsr_dcn_weight_il2_bytes(int32_t cout, int32_t cin);
int eavsr_pack_dcn_weight_il2(const float* weight, void* weight_il2, int32_t cout, int32_t cin, void* stream);
int eavsr_dcnv2_il2_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_il2,
                        const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                        int32_t deform_groups, int32_t nprod, int32_t heads, void* stream);

/* 16-bit form of eavsr_dcnv2_il_f32 for the 16-bit mode (BASELINE.json configs[2] bf16 / configs[4] fp16; csrc/dcnv2_il16.hip):
 * x_il8_h16 is the IL8 layout in 16 bits ([n][cin/8][h][w][8] bf16 / fp16: eavsr_flow_warp_pair_f32 with outb_il8 = 2 | 3, or
 * eavsr_nchw_to_il8_h16), the weights are rounded once to the same 16-bit type (eavsr_pack_dcn_il16_weight), the bilinear blend
 * and the accumulation are fp32, the blended sample is rounded once to 16 bits for the MFMA, offsets / mask / heads / bias / out
 * stay fp32.  dtype: 1 = fp16, 2 = bf16.  Judged by PSNR against the fp32 oracle, as BASELINE.json says for these configs. */
int64_t eavsr_dcn_il16_weight_bytes(int32_t cout, int32_t cin);
int eavsr_pack_dcn_il16_weight(const float* weight, void* weight_il16, int32_t cout, int32_t cin, int32_t dtype, void* stream);
int eavsr_nchw_to_il8_h16(const float* x, void* out_il8_h16, int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype, void* stream);
int eavsr_dcnv2_il16(const void* x_il8_h16, const float* offset_or_heads, const float* mask, const void* weight_il16,
                     const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                     int32_t deform_groups, int32_t dtype, int32_t heads, void* stream);

/* The rest of mmcv.ops.modulated_deform_conv2d's signature (any kernel size, stride, padding, dilation, conv groups,
 * deformable groups; networks.py:575-583 declares them, the reference never uses them): a plain one-thread-per-pixel
 * kernel on the ORIGINAL weight layout (cout, cin/groups, kh, kw).  offset (n, dg*2*kh*kw, ho, wo), mask
 * (n, dg*kh*kw, ho, wo), out (n, cout, ho, wo).  Forward only.                                                    */
int eavsr_dcnv2_generic_f32(const float* x, const float* offset, const float* mask, const float* weight,
                            const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w,
                            int32_t cout, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w,
                            int32_t pad_h, int32_t pad_w, int32_t dil_h, int32_t dil_w, int32_t groups,
                            int32_t deform_groups, void* stream);

/* ---- dense convolution (a9, a10, a11 convs; predictor heads a3/a4/a6; callers f1/f2) ----------
 * replaces torch.nn.Conv2d forward (stride 1, padding k/2, dilation 1, groups 1) as used at
 * eavsrp_model.py:146,313-314 (fusion 1x1), :381 (input conv), networks.py:456-458 (RCAB convs),
 * :293-295,:330-331 (predictor heads), :568 (TransOffsetworelu), SPyNet 7x7 convs
 * (eavsrp_model.py:534-574), encoder / upsample convs.  Implicit GEMM on v_mfma_f32_32x32x2_f32
 * (exact fp32 fma chain).
 * The input is the virtual channel-concatenation of up to 5 NCHW sources (replaces torch.cat at
 * eavsrp_model.py:313,322,353): every source but the last must have channels % eavsr_conv2d_ck(ksize) == 0.
 * out = act(conv + bias) + residual.                                                           */
typedef struct eavsr_conv2d_desc {
  const float* src[5];
  int32_t src_c[5];
  int32_t n_src;
  int32_t ksize; /* 1, 3, 5 or 7 */
  const float* weight_packed; /* eavsr_pack_conv_weight_f32 */
  const float* bias;          /* [cout] or NULL */
  const float* residual;      /* (n,cout,h,w) or NULL, added after the activation */
  float* out;                 /* (n,cout,h,w) */
  /* optional: per-tile channel sums of `out` for the channel attention (networks.py:444-445):
   * chan_partial[(n * tiles + tile) * cout + co], tiles = eavsr_conv2d_tiles(n,h,w,ksize); NULL = off */
  float* chan_partial;
  int32_t n, h, w, cin, cout;
  int32_t act;
  float slope;
  /* optional fused channel-attention prologue (RCABlock tail, networks.py:447,463-464, folded into the
   * NEXT conv): the effective input is src[0] * ca_scale[n,c] + ca_x; when ca_out != NULL it is also
   * written there (it is the next block's residual stream).  Needs n_src == 1, ksize == 3, w % 4 == 0,
   * 16-byte aligned src[0] / ca_x, cin % 4 == 0, 33..64 output channels; otherwise the call returns -2
   * and the caller uses eavsr_scale_residual_f32 + a plain conv.  NULL = off. */
  const float* ca_scale; /* (n, cin) */
  const float* ca_x;     /* (n, cin, h, w) */
  float* ca_out;         /* (n, cin, h, w) or NULL */
  /* 0: out is (n, cout, h, w).  2: out is F.pixel_shuffle(conv, 2) = (n, cout/4, 2h, 2w) written by the epilogue itself (the
   * upsampling tail, eavsrp_model.py:343-347: channel 4c+2i+j -> out[c][2y+i][2x+j]); eavsr_conv3x3_wino4_f32 only, cout % 4 == 0,
   * no residual, no channel sums -- every other entry point returns -2 for a non-zero value. */
  int32_t out_shuffle;
  /* NULL, or (n, cout) fp32 (ABI 27; eavsr_conv3x3_wino4_f32 only, with `residual`): out = residual + res_scale[n][co] * act(conv
   * + bias) -- RCABlock's tail `res * y + x` (models/networks.py:463-464) as the second convolution's epilogue; the attention
   * comes from eavsr_ca_scale_pre_f32 BEFORE the launch.  Every other entry point returns -2 for a non-NULL value. */
  const float* res_scale;
  /* NULL, or [n][4][border_stride][64] fp32 (ABI 29; eavsr_conv3x3_wino4_f32 only, 64 output channels): the epilogue also leaves the
   * sums of the OUTPUT's border lines per border tile -- border 0 / 1 = image row 0 / h - 1, one piece per tile column
   * (eavsr_conv3x3_wino4_border_pieces: p_rows); border 2 / 3 = image column 0 / w - 1, one piece per (tile row, wave row) (p_cols)
   * -- for eavsr_ca_scale_pre_pieces: RCABlock's attention (models/networks.py:444-447) needs them of the second convolution's
   * input, and summing them was a launch of its own on every block's dependent chain.  Every other entry point returns -2. */
  float* border_pieces;
  int32_t border_stride; /* >= max(p_rows, p_cols) */
  /* NULL, or (n, cout, h, w) fp32 (ABI 30; eavsr_conv3x3_f32x6s only, with chan_partial, w % 4 == 0, 16-byte aligned tensors):
   * chan_partial then holds the per-tile channel sums of out * sum_mul -- of the values the launch STORES (after activation and
   * residual) -- instead of the sums of act(conv + bias): the plane sums `sum_hw d r` that the backward of RCABlock's tail
   * (models/networks.py:463-464) starts with, taken where d is produced (the previous block's input-gradient convolution) instead
   * of by a launch of their own.  Every other entry point returns -2 for a non-NULL value. */
  const float* sum_mul;
} eavsr_conv2d_desc;

/* sizeof(eavsr_conv2d_desc) as THIS library was compiled (ABI 28): a binding compares it with its own struct at load time, so
 * that a descriptor that grew without an ABI bump (ABI 27 did, within one round) can never be read past its end. */
size_t eavsr_conv2d_desc_size(void);
int eavsr_conv2d_f32(const eavsr_conv2d_desc* desc, void* stream);

/* The same convolution by Winograd F(4x4, 3x3): 36 transform-domain GEMMs per 6x6 input tile, 4x fewer
 * multiplications than the direct sum (1.78x fewer than F(2x2, 3x3)), fp32 throughout; the larger transform costs
 * accuracy (~1e-5 of the output scale against ~4e-7).  Same descriptor and epilogue as eavsr_conv2d_f32 (desc->weight_packed is
 * ignored); requires ksize 3, w % 4 == 0, 16-byte aligned sources of a multiple of 4 channels, out / residual 16-byte aligned,
 * chan_partial has eavsr_conv3x3_wino4_tiles(h, w) rows per sample (8 x 64-pixel tiles).  The fused channel-attention
 * prologue (ca_scale, ca_x, optional ca_out: input = src[0] * ca_scale[n, c] + ca_x, single source) is applied in the
 * input transform.  weight_wino4: eavsr_wino4_weight_elems(cout, cin) floats from eavsr_pack_conv_weight_wino4. */
int64_t eavsr_wino4_weight_elems(int32_t cout, int32_t cin);
int eavsr_pack_conv_weight_wino4(const float* weight, float* weight_wino4, int32_t cout, int32_t cin, void* stream);
int32_t eavsr_conv3x3_wino4_tiles(int32_t h, int32_t w);
/* pieces per row border / per column border that desc.border_pieces receives for an (h, w) image */
int eavsr_conv3x3_wino4_border_pieces(int32_t h, int32_t w, int32_t* p_rows, int32_t* p_cols);
int eavsr_conv3x3_wino4_f32(const eavsr_conv2d_desc* desc, const float* weight_wino4, void* stream);

/* 5x5 stride-1 "same" convolution by Winograd F(2x2, 5x5) - the same 6x6-tile pipeline (same points and input
 * transform), 2x2 outputs per tile, 2.78x fewer multiplications than the direct sum: the 5x5 offset / mask heads of
 * AdaptBlockOffset (networks.py:283-285, 306-308).  Descriptor as above with ksize 5; sources a multiple of 4 channels
 * and 16-byte aligned, w % 4 == 0, out / residual 8-byte aligned, ca_* NULL; chan_partial has
 * eavsr_conv5x5_wino_tiles(h, w) rows per sample (4 x 32-pixel tiles).  weight: eavsr_wino4_weight_elems(cout, cin)
 * floats written by eavsr_pack_conv_weight_wino5x5 from the (cout, cin, 5, 5) weight. */
int eavsr_pack_conv_weight_wino5x5(const float* weight, float* weight_wino5x5, int32_t cout, int32_t cin, void* stream);
int32_t eavsr_conv5x5_wino_tiles(int32_t h, int32_t w);
int eavsr_conv5x5_wino_f32(const eavsr_conv2d_desc* desc, const float* weight_wino5x5, void* stream);
/* input-channel chunk the kernel for this kernel size works in (sources must be multiples of it) */
int32_t eavsr_conv2d_ck(int32_t ksize);
/* rows of the spatial tile (32, 16 or 8; 32 columns) the kernel runs an (n, h, w) problem in: small images get
 * shorter tiles so that the launch still covers the 256 CUs; and the number of tiles per image (= rows of
 * chan_partial per sample).  The fused channel-attention prologue exists for 32-row tiles only (-2 otherwise). */
int32_t eavsr_conv2d_tile_rows(int32_t n, int32_t h, int32_t w, int32_t ksize);
int32_t eavsr_conv2d_tiles(int32_t n, int32_t h, int32_t w, int32_t ksize);
/* number of floats of the packed weight buffer for (cout, cin, ksize) */
int64_t eavsr_packed_weight_elems(int32_t cout, int32_t cin, int32_t ksize);
/* weight (cout,cin,k,k) -> packed [cout_tile][cin_pad][k*k][co_in_tile], zero padded */
int eavsr_pack_conv_weight_f32(const float* weight, float* packed, int32_t cout, int32_t cin,
                               int32_t ksize, void* stream);

/* 3x3 convolutions with 2, 3, 4 or 6 output channels (predictor heads networks.py:330-331, TransOffsetworelu :568,
 * conv_last eavsrp_model.py:156) on the vector ALUs: an MFMA tile would idle > 80 % of the matrix pipe.  weight is
 * the ORIGINAL (cout, cin, 3, 3) layout.  out = act(conv + bias) + residual.  Other cout: returns -2. */
int eavsr_conv3x3_smallco_f32(const float* x, const float* weight, const float* bias, const float* residual,
                              float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t act,
                              float slope, void* stream);

/* The same convolution (cout in {2, 3, 4, 6}) shaped to run beside a resident Winograd workgroup of another stream: 256 threads,
 * <= 40 vector registers, 24 KB of LDS, weights as scalar operands from the packed form [ci][kx][block] made by
 * eavsr_pack_smallco_weight (eavsr_smallco_packed_elems floats).  w % 4 == 0, 16-byte aligned input; same reference call sites as
 * eavsr_conv3x3_smallco_f32 (models/networks.py:330-331,568, models/eavsrp_model.py:156). */
int64_t eavsr_smallco_packed_elems(int32_t cout, int32_t cin);
int eavsr_pack_smallco_weight(const float* weight, float* packed, int32_t cout, int32_t cin, void* stream);
int eavsr_conv3x3_smallco_lite_f32(const float* x, const float* weight_packed, const float* bias, const float* residual,
                                   float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t act,
                                   float slope, void* stream);

/* 7x7 and 5x5 stride-1 "same" convolutions, fp32 in / out, the contraction on the bf16 matrix pipe with both operands split exactly
 * into three bf16 terms and the six partial products >= 2^-16 of the result kept ("bf16x6": <= 2 * 2^-24 relative per product, one
 * fp32 rounding; no operand is rounded):
 *   ksize 7: SPyNet's basic module (models/eavsrp_model.py:398-431: 8 -> 32 -> 64 -> 32 -> 16 -> 2, ReLU between; called per
 *            pyramid level from compute_flow, models/eavsrp_model.py:433-488);
 *   ksize 5: AdaptBlockOffset's three heads as one 64 -> 15 D convolution (models/networks.py:289-315).
 * cin % 8 == 0; any cout.  `weight_x6`: eavsr_conv_weight_x6_bytes(ksize, cout, cin) bytes written by eavsr_pack_conv_weight_x6
 * from the (cout, cin, ksize, ksize) weight.  act / slope as eavsr_conv2d_f32.
 * FINITE INPUTS: the exact split forms residuals a - trunc_bf16(a); for a = +-inf that is inf - inf = NaN, so an infinity in
 * x or the weights becomes NaN in every output of its receptive field (the fp32-MFMA and Winograd kernels and the reference give
 * +-inf there); values below 2^-110 lose their mid / lo terms to bf16 denormals.  The same holds for every bf16-split kernel of
 * this library (eavsr_dcnv2_f32x9, eavsr_dcnv2_il_f32, eavsr_dcnv2_il2_f32, eavsr_conv3x3_f32x9).                            */
size_t eavsr_conv_weight_x6_bytes(int32_t ksize, int32_t cout, int32_t cin);
int eavsr_pack_conv_weight_x6(const float* weight, void* packed, int32_t ksize, int32_t cout, int32_t cin, void* stream);
/* The same packed form for the INPUT-GRADIENT convolution of a stride-1 "same" convolution with forward weight `weight`
 * (cout_w, cin_w, k, k) (ABI 27): the transposed, flipped weight W'[ci][co][tap] = W[co][ci][k*k - 1 - tap] is read in place, no
 * materialised copy (loss.backward() of every nn.Conv2d of RCABlock, models/networks.py:456-464; eavsrp_model.py:109-113).  The
 * result packs a (cin_w output, cout_w input)-channel convolution: eavsr_conv_weight_x6_bytes(ksize, cin_w, cout_w) bytes. */
int eavsr_pack_conv_weight_x6_dgrad(const float* weight, void* packed, int32_t ksize, int32_t cout_w, int32_t cin_w, void* stream);
/* `count` weights of one square shape (c, c, ksize, ksize) packed in ceil(count / 48) launches (ABI 30): weights[t] -> packed[t]
 * (eavsr_conv_weight_x6_bytes(ksize, c, c) bytes each), transposed[t] != 0: the input-gradient form.  HOST arrays of device pointers,
 * read at the call (pointers travel by value in the kernel arguments: the launch captures into a graph).  The training step's
 * 3x3 64 -> 64 weights (models/networks.py:456-458) and their input-gradient forms, once per step. */
int eavsr_pack_conv_weight_x6_multi(const float* const* weights, void* const* packed, const int32_t* transposed, int32_t count,
                                    int32_t ksize, int32_t c, void* stream);
int eavsr_conv_f32x6(const float* x, const void* weight_x6, const float* bias, float* out, int32_t n, int32_t cin, int32_t cout,
                     int32_t h, int32_t w, int32_t ksize, int32_t act, float slope, int32_t sigmoid_from, void* stream);
/* The 3x3 form for SMALL launches (ABI 27): 64 input channels, one source, the same exact bf16x6 arithmetic; the descriptor's bias,
 * activation (incl. EAVSR_ACT_RELU_MASK), residual and per-tile channel sums (eavsr_conv3x3_x6s_tiles(h, w) rows per sample, 8 x 32
 * pixel tiles) -- no channel-attention prologue, no pixel-shuffle store (-2: call eavsr_conv2d_f32).  RCABlock's convolutions
 * (models/networks.py:456-464) and their input-gradient convolutions at a training crop (2 x 64 x 96 x 96 per launch,
 * eavsrp_model.py:109-119), where a launch is fewer workgroups than the GPU has CUs and its time is one workgroup's chain of
 * matrix instructions: six bf16 products take 0.375 x the fp32 MFMA's cycles.  weight_x6 = eavsr_pack_conv_weight_x6(ksize 3). */
int32_t eavsr_conv3x3_x6s_tiles(int32_t h, int32_t w);
int eavsr_conv3x3_f32x6s(const eavsr_conv2d_desc* desc, const void* weight_x6, void* stream);
/* sigmoid_from: -1, or a multiple of 8: output channels >= sigmoid_from leave through the sigmoid instead of `act` -- the mask
 * head of AdaptBlockOffset (mask = torch.sigmoid(mask_conv(..)), models/networks.py:313-314) when the three heads run as one
 * convolution; eavsr_dcnv2_il2_f32 (heads = 2) then takes the masks as they are. */

/* Which schedule eavsr_conv3x3_wino4_f32 runs for plain 3x3 launches with an even number of 4-channel chunks: 1 = grouped
 * (transform phases of two chunks, pure GEMM iterations; default), 0 = duty pair (EAVSR_W4_GRP=0).  Chosen once per process from
 * the environment; reported so that recorded measurements state which variant ran.                               */
int eavsr_wino4_schedule(void);

/* ---- a11: channel attention -----------------------------------------------------------------
 * CALayer (models/networks.py:432-447): scale[n,c] = sigmoid(W2 . relu(W1 . mean_hw(r) + b1) + b2)
 * from the per-tile sums produced by eavsr_conv2d_f32(chan_partial).  w1 (cr,c), w2 (c,cr).      */
int eavsr_ca_scale_f32(const float* chan_partial, int32_t tiles, int32_t hw,
                       const float* w1, const float* b1, const float* w2, const float* b2,
                       float* scale, int32_t n, int32_t c, int32_t cr, void* stream);
/* the same with the sample means (n, c) written to mean_out (nullable; ABI 26): the training step's backward of the RCAB tail
 * (eavsr_rcab_tail_bwd_f32) takes them */
int eavsr_ca_scale_mean_f32(const float* chan_partial, int32_t tiles, int32_t hw, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* scale, float* mean_out, int32_t n, int32_t c, int32_t cr,
                            void* stream);
/* RCABlock tail (networks.py:447,464): out = r * scale[n,c] + x */
int eavsr_scale_residual_f32(const float* r, const float* scale, const float* x, float* out,
                             int32_t n, int32_t c, int32_t hw, void* stream);
/* Both of the above in one launch (networks.py:444-447,463-464): out = r * sigmoid(W2 . relu(W1 . mean_hw(r) + b1) + b2) + x from
 * the conv's per-tile channel sums; every workgroup finishes the (fixed-order) mean + MLP of its sample itself, then streams its
 * slice.  hw % 4 == 0, 16-byte aligned r / x / out (otherwise -2: use the two calls above). */
int eavsr_ca_tail_f32(const float* r, const float* chan_partial, int32_t tiles, const float* w1, const float* b1,
                      const float* w2, const float* b2, const float* x, float* out, int32_t n, int32_t c, int32_t cr,
                      int32_t hw, void* stream);
/* The same launch also writing the attention scale_out (n, c) and, when mean_out != NULL, the channel means (n, c) -- what the
 * backward of CALayer needs (training step, ABI 27); scale_out == NULL: eavsr_ca_tail_f32. */
int eavsr_ca_tail_stats_f32(const float* r, const float* chan_partial, int32_t tiles, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* x, float* out, float* scale_out, float* mean_out,
                            int32_t n, int32_t c, int32_t cr, int32_t hw, void* stream);

/* ---- a3 / a6 front end ----------------------------------------------------------------------
 * AdaptBlock2_3x3 / AdaptBlockOffset `concat` + `concat2` (models/networks.py:290-291,300 and
 * :327-328,336): cat(x,h_hr) -> depthwise 3x3 + LeakyReLU(0.2) -> grouped 3x3 (2 in / 1 out,
 * groups=c) + LeakyReLU(0.2), fused in one pass.  w1 (2c,1,3,3) b1 (2c) w2 (c,2,3,3) b2 (c).       */
int eavsr_adapt_frontend_f32(const float* x, const float* h_hr, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* out,
                             int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/* affine -> sampling offsets (models/networks.py:302-311 and :338-346) and mask sigmoid (:313-314).
 * heads: (n, 6*D [+ 9*D], h, w) = [transform 4D | translation 2D | mask logits 9D] (the three head
 * convs run as one conv with concatenated weights).  offset: (n, 18*D, h, w) in mmcv order;
 * mask: (n, 9*D, h, w) = sigmoid, or NULL when the heads carry no mask logits (D = 1 blocks).      */
int eavsr_affine_offsets_f32(const float* heads, float* offset, float* mask,
                             int32_t n, int32_t D, int32_t h, int32_t w, void* stream);

/* ---- a5 / a12 resampling glue ---------------------------------------------------------------
 * out = scale * bilinear_align_corners(in [+ in2]) [+ addend]   (F.interpolate(..., align_corners=True)
 * at models/networks.py:600-601,608,613 with the /4, /2, *2 factors and the adds at :610,:613).   */
int eavsr_resize_bilinear_ac_f32(const float* in, const float* in2, const float* addend, float* out,
                                 int32_t n, int32_t c, int32_t hin, int32_t win,
                                 int32_t hout, int32_t wout, float scale, void* stream);
/* feature pyramid (models/eavsrp_model.py:218-220): F.interpolate(scale 0.5 / 0.25, bilinear,
 * align_corners=False) == 2x2 box means of pixels (2d,2d+1) and (4d+1,4d+2).  h % 4 == w % 4 == 0. */
int eavsr_pyramid_f32(const float* in, float* down2, float* down4,
                      int32_t nc, int32_t h, int32_t w, void* stream);
/* out = a + b [+ c] elementwise (flow sums at networks.py:619, eavsrp_model.py:309,323) */
int eavsr_add_f32(const float* a, const float* b, const float* c, float* out, int64_t count,
                  void* stream);

/* ---- f1 / f2 glue either side of the path (round 3: the ATen kernels that were left inside the forward) -----------
 * out = (in - mean[c]) / std[c]: SPyNet.compute_flow models/eavsrp_model.py:436-437, ContrasExtractorLayer.forward
 * models/networks.py:550 (mean / std: c device floats). */
int eavsr_normalize_f32(const float* in, const float* mean, const float* std, float* out,
                        int32_t n, int32_t c, int32_t hw, void* stream);
/* F.avg_pool2d(x, kernel_size=2, stride=2, count_include_pad=False), models/eavsrp_model.py:450-462; h, w even. */
int eavsr_avg_pool2_f32(const float* in, float* out, int32_t nc, int32_t h, int32_t w, void* stream);
/* F.interpolate(x, size=(hout, wout), mode='bilinear', align_corners=False): models/eavsrp_model.py:499-509 (SPyNet's
 * resize to a multiple of 32 and back) and nn.Upsample(scale_factor, 'bilinear') at :158,359 (the LR skip).  cmul = c
 * multiplies channel 0 by m0 and channel 1 by m1 afterwards (the flow rescaling of :519-521); cmul = 0: no scaling. */
int eavsr_resize_bilinear_f32(const float* in, float* out, int32_t n, int32_t c, int32_t hin, int32_t win,
                              int32_t hout, int32_t wout, int32_t cmul, float m0, float m1, void* stream);
/* torch.cat([a, b, c], dim=1) of (n, ca | cb | cc, h, w): the 8-channel input of a SPyNet level, models/eavsrp_model.py:486 */
int eavsr_concat3_f32(const float* a, int32_t ca, const float* b, int32_t cb, const float* c, int32_t cc, float* out,
                      int32_t n, int32_t hw, void* stream);

/* ============================================================================================
 * Backward entry points (training step).  They replace the ATen / cuDNN / mmcv
 * backward kernels autograd runs for the same modules in the reference's loss.backward()
 * (models/eavsrp_model.py:109-113).  Same boundary rules as above.
 * ============================================================================================ */

/* g = dy * act'(y) with y the activation OUTPUT (ReLU / LeakyReLU fused into eavsr_conv2d_f32) */
int eavsr_act_bwd_f32(const float* dy, const float* y, float* g, int64_t count, int32_t act, float slope,
                      void* stream);
/* out[nc] = scale * sum_hw a[nc,hw] (* b[nc,hw] when b != NULL): bias gradients, channel means
 * (CALayer avg_pool, networks.py:445) and d(scale) of the RCAB tail */
int eavsr_plane_sum_f32(const float* a, const float* b, float* out, int32_t nc, int32_t hw, float scale,
                        void* stream);
/* out[c] = sum_n sum_hw a[n,c,hw]: the bias gradient of every nn.Conv2d on the path (autograd's sum over
 * (0, 2, 3) in the reference, models/eavsrp_model.py:109-113); n == 0 gives zeros; accumulate != 0 adds to out
 * (a parameter used once per frame sums its per-use gradients in place) */
int eavsr_channel_sum_f32(const float* a, float* out, int32_t n, int32_t c, int32_t hw, int32_t accumulate,
                          void* stream);
/* ... over up to 8 tensors of one shape in one launch (ABI 26; a_list: HOST array of nseg device pointers): the bias gradient of
 * the segments of eavsr_conv_wgrad_multi_f32 */
int eavsr_channel_sum_multi_f32(const void* const* a_list, int32_t nseg, float* out, int32_t n, int32_t c, int32_t hw,
                                int32_t accumulate, void* stream);
/* backward of out = r * scale[n,c] + x w.r.t. r:  dr = d * scale[n,c] + dmean[n,c] (dmean nullable) */
int eavsr_scale_residual_bwd_f32(const float* d, const float* scale, const float* dmean, float* dr,
                                 int32_t n, int32_t c, int32_t hw, void* stream);
/* backward of the channel-attention MLP scale = sigmoid(W2 relu(W1 mean + b1) + b2) (networks.py:436-447):
 * writes dmean (n,c) and the four parameter gradients (overwritten, summed over n in a fixed order) */
int eavsr_ca_mlp_bwd_f32(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* dscale, float* dmean, float* dw1, float* db1, float* dw2, float* db2,
                         int32_t n, int32_t c, int32_t cr, void* stream);
/* The whole backward of the RCAB tail out = r * s + x, s = sigmoid(W2 relu(W1 mean_hw(r) + b1) + b2) (networks.py:444-447,463-464)
 * behind the plane sums dscale[n,c] = sum_hw d r (eavsr_plane_sum_f32), as one launch (ABI 26): dr = d * scale[n,c] +
 * dmean[n,c] / hw with dmean from the MLP's backward, and the four parameter gradients summed over n in a fixed order, written
 * (accumulate = 0) or ADDED (accumulate != 0) to dw1 (cr, 64), db1 (cr), dw2 (64, cr), db2 (64).  c = 64, cr in {1, 2, 4, 8};
 * other shapes: eavsr_ca_mlp_bwd_f32 + eavsr_scale_residual_bwd_f32.  mean_rows = 0: `mean` is (n, 64); mean_rows > 0: `mean` is the
 * convolution epilogue's (n, mean_rows, 64) partial channel SUMS of r, added up and divided by hw here.  dscale_rows (ABI 30) = 0:
 * `dscale` is (n, 64); > 0: `dscale` is (n, dscale_rows, 64) partial sums of d r -- the rows eavsr_conv3x3_f32x6s leaves with
 * desc.sum_mul = r where it produces d -- added up here in row order: no plane-sum launch. */
int eavsr_rcab_tail_bwd_f32(const float* d, const float* scale, const float* mean, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* dscale, float* dr, float* dw1, float* db1,
                            float* dw2, float* db2, int32_t n, int32_t c, int32_t cr, int32_t hw, int32_t mean_rows,
                            int32_t dscale_rows, int32_t accumulate, void* stream);
/* backward of eavsr_flow_warp_f32 (zeros padding, NCHW flow): dx (pre-zeroed, accumulated with float
 * atomics; NULL = skip) and dflow (n,2,h,w; NULL = skip).  flow2 as in the forward. */
int eavsr_flow_warp_bwd_f32(const float* x, const float* flow, const float* flow2, const float* dout,
                            float* dx, float* dflow, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* backward of eavsr_resize_bilinear_ac_f32 w.r.t. its input: din (pre-zeroed) += scale * scatter(dout) */
int eavsr_resize_bilinear_ac_bwd_f32(const float* dout, float* din, int32_t n, int32_t c, int32_t hin,
                                     int32_t win, int32_t hout, int32_t wout, float scale, void* stream);
/* backward of eavsr_pyramid_f32 */
int eavsr_pyramid_bwd_f32(const float* ddown2, const float* ddown4, float* din, int32_t nc, int32_t h,
                          int32_t w, void* stream);
/* backward of eavsr_affine_offsets_f32: d(heads) from d(offset) and, for blocks with a mask, d(mask) and the
 * saved mask (= sigmoid(logits)) */
int eavsr_affine_offsets_bwd_f32(const float* doffset, const float* dmask, const float* mask, float* dheads,
                                 int32_t n, int32_t D, int32_t h, int32_t w, void* stream);

/* conv weight gradient for a (<= 64 output channels) x (<= 64 input channels) block:
 *   dweight[co0+co][ci_dst0+ci][tap] (+)= sum_{n,y,x} dy[n,co0+co,y,x] * x[n,ci0+ci,y+ky-P,x+kx-P]
 * dweight is the full (cout_total, cin_total, k, k) gradient; x is one source of the virtual concatenation
 * (cin_src channels) whose channel ci0 lands at input channel ci_dst0 of the weight.  ksize 1, 3 or 5.
 * workspace: eavsr_conv_wgrad_blocks(n,h,w,ksize) * 64*64*ksize*ksize floats.  accumulate != 0 adds. */
int32_t eavsr_conv_wgrad_blocks(int32_t n, int32_t h, int32_t w, int32_t ksize);
int eavsr_conv_wgrad_f32(const float* dy, const float* x, float* dweight, float* workspace, int32_t n, int32_t h,
                         int32_t w, int32_t cout_total, int32_t co0, int32_t cin_src, int32_t ci0,
                         int32_t cin_total, int32_t ci_dst0, int32_t ksize, int32_t accumulate, void* stream);
/* ksize 1 over ALL 64-channel blocks of one source in one launch + one reduction (ABI 27): dweight[co0+co][ci_dst0 + ci] (+)=
 * sum dy[.., co0+co, ..] x[.., ci, ..] for ci = 0 .. cin_src-1.  DCNv2's weight gradient = this against the 576-channel column
 * tensor of eavsr_dcnv2_im2col_f32.  workspace: eavsr_conv_wgrad_blocks(n,h,w,1) * ceil(cin_src / 64) * 64*64 floats. */
int eavsr_conv_wgrad_span_f32(const float* dy, const float* x, float* dweight, float* workspace, int32_t n, int32_t h, int32_t w,
                              int32_t cout_total, int32_t co0, int32_t cin_src, int32_t cin_total, int32_t ci_dst0,
                              int32_t accumulate, void* stream);
/* The same over up to 8 SEGMENTS in one launch (ABI 26): dy_list[s] / x_list[s] are HOST arrays of device pointers to nseg (dY, X)
 * pairs of identical shapes -- the uses of one weight across the frames of the recurrence (models/eavsrp_model.py:271-324: the
 * same backbone / alignment weights at every time step), whose gradients autograd would compute and sum one use at a time
 * (loss.backward(), models/eavsrp_model.py:109-113).  n is the batch of ONE segment; workspace:
 * eavsr_conv_wgrad_blocks(n * nseg, h, w, ksize) * 64*64*ksize*ksize floats.  The pointers are read on the host at the call. */
/* ksize 3 with w % 4 == 0 and 16-byte aligned tensors runs on the bf16 matrix pipe with both operands split exactly into three
 * bf16 terms (ABI 27; six partial products, fp32 accumulation: no operand rounded); eavsr_wgrad3_mode() = 1.  EAVSR_WGRAD3=fp32 in
 * the environment keeps the fp32-MFMA kernel (0; chosen once per process, reported so that measurements state which ran). */
int eavsr_wgrad3_mode(void);
int eavsr_conv_wgrad_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                               float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total, int32_t co0,
                               int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0, int32_t ksize,
                               int32_t accumulate, void* stream);
/* The same with the BIAS gradient riding along (ABI 27): dbias (cout_total,) or NULL.  dbias[co0 .. co0 + 63] (+)= sum over
 * segments, images and pixels of dy -- inside the bf16x6 3x3 kernel (which stages dY anyway); every other kernel launches
 * eavsr_channel_sum_multi_f32 over ALL channels with the co0 == 0 call.  Pass dbias with ONE (source, ci0) per co0 block.
 * workspace: eavsr_conv_wgrad_blocks(n * nseg, h, w, ksize) * (64*64*ksize*ksize + 64) floats. */
int eavsr_conv_wgrad_bias_multi_f32(const void* const* dy_list, const void* const* x_list, int32_t nseg, float* dweight,
                                    float* dbias, float* workspace, int32_t n, int32_t h, int32_t w, int32_t cout_total,
                                    int32_t co0, int32_t cin_src, int32_t ci0, int32_t cin_total, int32_t ci_dst0, int32_t ksize,
                                    int32_t accumulate, void* stream);

/* DCNv2 backward samplers (the two GEMMs run on eavsr_conv_wgrad_f32 / eavsr_conv2d_f32 with k = 1):
 * columns (n, c*9, h, w) = im2col(x, offset, mask);  from dcolumns: dx (pre-zeroed, atomics; NULL = skip),
 * doffset (n, dg*18, h, w), dmask (n, dg*9, h, w). */
/* DCNv2 backward on the sampler's side (ABI 29; csrc/dcn_bwd.hip): the whole backward of modulated_deform_conv2d (models/networks.py:
 * 627-630 under loss.backward(), models/eavsrp_model.py:109-119) without a column tensor: the column gradient W^T . dY stays in the MFMA
 * accumulators of a (deformable group, 4 x 16-pixel tile) unit, d_offset / d_mask are reduced over the group's 8 channels in the wave,
 * dx is added -- without atomics -- into per-wave LDS windows that leave the unit as one 16 x 32-cell slab and are gathered per dx
 * cell afterwards (samples further than 6 rows / 8 columns from their pixel go to dx_il8 by global atomics), dW = dY . col^T runs on
 * the same matrix instructions from a per-wave LDS tile of re-sampled columns and is reduced from per-workgroup slabs in a fixed order.
 * 64 -> 64 channels, 8 deformable groups, 3x3 (-2 otherwise: eavsr_dcnv2_im2col_f32 / eavsr_dcnv2_col2im_f32 below take any
 * configuration).
 * x_il8: the input in the IL8 layout (eavsr_nchw_to_il8_f32); dx_il8: IL8 gradient buffer, ADDED to (zero it for a plain gradient), or
 * NULL (eavsr_il8_to_nchw_f32 turns it into NCHW); dweight (64, 64, 3, 3) written or, accumulate_dw != 0, added to; workspace:
 * eavsr_dcnv2_bwd_workspace_floats floats, 16-byte aligned (dW slabs + packed W^T + 2 KB of dx slab per pixel).  The bias gradient is
 * eavsr_channel_sum_f32 of dy. */
int32_t eavsr_dcnv2_bwd_grid(int32_t n, int32_t h, int32_t w);
int64_t eavsr_dcnv2_bwd_workspace_floats(int32_t n, int32_t h, int32_t w);
int eavsr_dcnv2_bwd_f32(const float* x_il8, const float* offset, const float* mask, const float* weight, const float* dy, float* dx_il8,
                        float* doffset, float* dmask, float* dweight, float* workspace, int32_t n, int32_t cin, int32_t h, int32_t w,
                        int32_t cout, int32_t deform_groups, int32_t accumulate_dw, void* stream);
int eavsr_il8_to_nchw_f32(const float* x_il8, float* out, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
int eavsr_dcnv2_im2col_f32(const float* x, const float* offset, const float* mask, float* columns, int32_t n,
                           int32_t c, int32_t h, int32_t w, int32_t deform_groups, void* stream);
int eavsr_dcnv2_col2im_f32(const float* x, const float* offset, const float* mask, const float* dcolumns,
                           float* dx, float* doffset, float* dmask, int32_t n, int32_t c, int32_t h, int32_t w,
                           int32_t deform_groups, void* stream);

/* un-fused training form of the predictor front end: grouped 3x3 conv with one output channel per group and
 * cpg input channels per group (`concat`: cpg = 1, `concat2`: cpg = 2; networks.py:290-291,327-328),
 * optional LeakyReLU; and its backward (g = gradient w.r.t. the pre-activation output). */
int eavsr_gconv3x3_fwd_f32(const float* x, const float* weight, const float* bias, float* out, int32_t n,
                           int32_t cout, int32_t cpg, int32_t h, int32_t w, int32_t act, float slope, void* stream);
int eavsr_gconv3x3_bwd_f32(const float* g, const float* x, const float* weight, float* dx, float* dweight,
                           float* dbias, int32_t n, int32_t cout, int32_t cpg, int32_t h, int32_t w, void* stream);
/* accumulate != 0: dweight / dbias are added to (the uses of the weight across the frames of the recurrence, ABI 27) */
int eavsr_gconv3x3_bwd_acc_f32(const float* g, const float* x, const float* weight, float* dx, float* dweight,
                               float* dbias, int32_t n, int32_t cout, int32_t cpg, int32_t h, int32_t w, int32_t accumulate,
                               void* stream);

/* ============================================================================================
 * 16-bit residual backbone (BASELINE.json configs[2] bf16 / configs[4] fp16; SURVEY.md 8a: a10, a11).
 * dtype: 1 = fp16, 2 = bf16.  Activations are NHWC 16-bit (n, h, w, 64): one pixel = 128 contiguous bytes;
 * accumulation, bias and the channel-attention statistics stay fp32.  Replaces the 3x3 64->64 convs of
 * RCABlock / RCAGroup (models/networks.py:456-458,478) and the RCAB tail (:447,463-464) when the caller
 * opts into a 16-bit backbone.  All pointers 16-byte aligned.
 * ============================================================================================ */
int32_t eavsr_conv_h16_tiles(int32_t h, int32_t w);
/* weight (64,64,3,3) fp32 -> 64*576 16-bit values in MFMA fragment order */
int eavsr_pack_conv3x3_c64_h16(const float* weight, void* packed, int32_t dtype, void* stream);
/* out = [relu](conv3x3(x) + bias); chan_partial (nullable): fp32 (n, eavsr_conv_h16_partial_rows(n,h,w), 64) partial channel
 * sums of the output as stored (after the rounding to 16 bits: what the next layer reads) -- one row per tile, or, where a sample
 * has more tiles than twice the persistent workgroups (large images), one row per (workgroup, wave group) with that group's tiles
 * of the sample already added up; either way the rows of a sample sum to its channel sums, in a fixed order */
int32_t eavsr_conv_h16_partial_rows(int32_t n, int32_t h, int32_t w);
int eavsr_conv3x3_c64_h16(const void* x, const void* weight_packed, const float* bias, void* out,
                          float* chan_partial, int32_t n, int32_t h, int32_t w, int32_t relu, int32_t dtype,
                          void* stream);
/* RCABlock's tail inside its second convolution (ABI 27): out = res_x + res_scale[n][co] * (conv(x) + bias), rounded once to the
 * 16-bit type -- `res * y + x` of models/networks.py:463-464 without a launch of its own.  res_scale (n, 64) fp32 comes from
 * eavsr_ca_scale_pre_h16 BEFORE this launch: the channel means of a convolution's output are a linear function of border-corrected
 * channel sums of its input.  res_x: the block's input, (n, h, w, 64) 16-bit. */
int eavsr_conv3x3_c64_h16_res(const void* x, const void* weight_packed, const float* bias, void* out, const void* res_x,
                              const float* res_scale, int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream);
/* eavsr_conv3x3_c64_h16 that also leaves the sums of its OUTPUT's four border lines per border tile (ABI 29; the RCAB's first
 * convolution: eavsr_ca_scale_pre_pieces takes them instead of a border-sum launch).  border_pieces: [n][4][border_stride][64] fp32,
 * border 0 / 1 = image row 0 / h - 1 with p_rows pieces (one per 32-pixel tile column), 2 / 3 = image column 0 / w - 1 with p_cols
 * pieces (one per (8-row tile, wave of two rows)); border_stride >= max(p_rows, p_cols); chan_partial required. */
int eavsr_conv_h16_border_pieces(int32_t h, int32_t w, int32_t* p_rows, int32_t* p_cols);
int eavsr_conv3x3_c64_h16_b(const void* x, const void* weight_packed, const float* bias, void* out, float* chan_partial,
                            float* border_pieces, int32_t border_stride, int32_t n, int32_t h, int32_t w, int32_t relu,
                            int32_t dtype, void* stream);
/* scale[n][co] = sigmoid(W2 relu(W1 mean_hw(conv(t) + conv_bias) + b1) + b2) (CALayer, models/networks.py:444-447) from the per-tile
 * channel sums of t that eavsr_conv3x3_c64_h16(.., relu, chan_partial) left (rows = eavsr_conv_h16_partial_rows) and the border rows
 * / columns / corners of t (16-bit NHWC), WITHOUT running the convolution: sum_o conv(t)[co][o] = sum W[co][ci][ky][kx] (T[ci] -
 * R(ky) - C(kx) + X(ky,kx)).  conv_weight (64, 64, 3, 3) fp32 is rounded to `dtype` as the convolution's packed weights are.
 * workspace: eavsr_ca_scale_pre_ws_floats(n) floats.  Two launches (border sums, then one workgroup per sample). */
/* The fp32 NCHW form (64 channels): t (n, 64, h, w) fp32, chan_partial (n, tiles, 64) from the first convolution (relu + channel
 * sums), the second convolution's fp32 weights as they are. */
int eavsr_ca_scale_pre_f32(const float* t, const float* chan_partial, int32_t tiles, const float* conv_weight, const float* conv_bias,
                           const float* w1, const float* b1, const float* w2, const float* b2, float* scale, float* workspace,
                           int32_t n, int32_t h, int32_t w, int32_t cr, void* stream);
int64_t eavsr_ca_scale_pre_ws_floats(int32_t n);
/* The same attention in ONE launch (ABI 29): the border lines of t come as the pieces the FIRST convolution's epilogue wrote
 * (desc.border_pieces of eavsr_conv3x3_wino4_f32, layout [n][4][p_stride][64], p_rows / p_cols pieces per row / column border) instead
 * of from a border-sum launch.  dtype 0: t fp32 NCHW; 1 / 2: fp16 / bf16 NHWC (only its four corner pixels are read). */
int eavsr_ca_scale_pre_pieces(const void* t, const float* chan_partial, int32_t rows, const float* pieces, int32_t p_rows,
                              int32_t p_cols, int32_t p_stride, const float* conv_weight, const float* conv_bias, const float* w1,
                              const float* b1, const float* w2, const float* b2, float* scale, int32_t n, int32_t h, int32_t w,
                              int32_t cr, int32_t dtype, void* stream);
int eavsr_ca_scale_pre_h16(const void* t, const float* chan_partial, int32_t rows, const float* conv_weight, const float* conv_bias,
                           const float* w1, const float* b1, const float* w2, const float* b2, float* scale, float* workspace,
                           int32_t n, int32_t h, int32_t w, int32_t cr, int32_t dtype, void* stream);
/* The upsampling tail in the 16-bit modes (models/eavsrp_model.py:343-360; eavsrpx2_model.py likewise).
 * eavsr_conv3x3_c64_h16_act: out = act(conv3x3(x) + bias), act = EAVSR_ACT_NONE | RELU | LRELU(slope) -- conv_hr (:355-357) with
 *   pixel_shuffle2 = 0.  pixel_shuffle2 = 1: the 64 -> 256 convolution + nn.PixelShuffle(2) + activation of upsample1 / upsample2
 *   (:343-352) as four 64 -> 64 slices of the same kernel: weight_packed = FOUR packed matrices, slice k = 2 dy + dx holding the
 *   reference weight's output channels 4 c + k as its channel c (c = 0..63), bias likewise (4 x 64 floats); out is (n, 2 h, 2 w, 64)
 *   16-bit NHWC, output pixel (2 y + dy, 2 x + dx) = slice k's pixel (y, x) -- the shuffle is the store pattern.
 * eavsr_conv3x3_c64to3_h16: conv_last (:359-360): x 16-bit NHWC (n, h, w, 64); weight: the (3, 64, 3, 3) parameter permuted to
 *   [ky][kx][8-channel block][co][8 channels] and rounded to the 16-bit type by the caller (1,728 values: w.view(3, 8, 8, 3, 3)
 *   .permute(3, 4, 1, 0, 2)); fp32 accumulation, out fp32 NCHW (n, 3, h, w) = conv + bias + residual (nullable: the bilinear skip). */
int eavsr_conv3x3_c64_h16_act(const void* x, const void* weight_packed, const float* bias, void* out, int32_t n, int32_t h,
                              int32_t w, int32_t act, float slope, int32_t pixel_shuffle2, int32_t dtype, void* stream);
int eavsr_conv3x3_c64to3_h16(const void* x, const void* weight, const float* bias, const float* residual, float* out,
                             int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream);
/* Generic 3x3 convolution of the 16-bit modes (csrc/conv3_h16.hip): same descriptor as eavsr_conv2d_f32 (fp32 NCHW sources as a
 * virtual concatenation, fp32 NCHW out = act(conv + bias)), the operands rounded once to the 16-bit type, fp32 accumulation.
 * Takes the 3x3 convolutions that are not the 64 -> 64 NHWC backbone kernel when the caller opted into a 16-bit mode: the first
 * convolution of each residual backbone (models/eavsrp_model.py:375-381 over the concatenated feature lists) and the encoder's
 * layers (models/networks.py:522-552).  ksize 3, every source's channel count a multiple of 16, w % 4 == 0, 16-byte aligned
 * sources; no residual / channel sums / channel-attention prologue / pixel shuffle (-2).  weight_packed: eavsr_pack_conv3x3_h16g
 * (cout, cin, 3, 3) fp32 -> eavsr_conv3x3_h16g_weight_bytes(cout, cin) bytes, a cache of the library version that made it. */
int64_t eavsr_conv3x3_h16g_weight_bytes(int32_t cout, int32_t cin);
int eavsr_pack_conv3x3_h16g(const float* weight, void* packed, int32_t cout, int32_t cin, int32_t dtype, void* stream);
int eavsr_conv3x3_h16g_f32(const eavsr_conv2d_desc* desc, int32_t dtype, void* stream);
/* SPyNet's 7x7 layers (models/eavsrp_model.py:398-431) in the 16-bit modes: the kernel of eavsr_conv_f32x6 (csrc/conv_x6.hip) with ONE
 * operand plane -- fp32 NCHW in and out, operands rounded once (nearest even) to dtype 1 = fp16 / 2 = bf16, one product per operand
 * pair instead of six, fp32 accumulation.  ksize 7, cin % 8 == 0; the packed weight (eavsr_pack_conv_weight_h16x1,
 * eavsr_conv_weight_h16x1_bytes) is a cache of the library version that made it. */
size_t eavsr_conv_weight_h16x1_bytes(int32_t ksize, int32_t cout, int32_t cin);
int eavsr_pack_conv_weight_h16x1(const float* weight, void* packed, int32_t ksize, int32_t cout, int32_t cin, int32_t dtype, void* stream);
int eavsr_conv_h16x1(const float* x, const void* weight_h16x1, const float* bias, float* out, int32_t n, int32_t cin, int32_t cout,
                     int32_t h, int32_t w, int32_t ksize, int32_t act, float slope, int32_t dtype, void* stream);
/* The predictor's three 5x5 heads (transform_matrix_conv ++ translation_conv ++ mask_conv, models/networks.py:283-285,
 * 298-301) in the 16-bit modes (csrc/conv5_h16.hip): x 16-bit NHWC (n, h, w, 64) -- the front-end feature through
 * eavsr_nchw_f32_to_nhwc_h16 --, weight (cout, 64, 5, 5) fp32 rounded once by the pack call (cout <= 128), fp32 accumulation,
 * out fp32 NCHW (n, cout, h, w) = conv5x5(x, pad 2) + bias: the `heads` operand of eavsr_dcnv2_il16 / eavsr_affine_offsets_f32. */
int64_t eavsr_conv5x5_c64_h16_weight_bytes(void);
int eavsr_pack_conv5x5_c64_h16(const float* weight, void* packed, int32_t cout, int32_t dtype, void* stream);
int eavsr_conv5x5_c64_h16(const void* x, const void* weight_packed, const float* bias, float* out, int32_t n, int32_t h,
                          int32_t w, int32_t cout, int32_t dtype, void* stream);
/* fp32 NCHW -> 16-bit NHWC, and back with an optional fp32 NCHW residual added (RCAGroup's `+ x`) */
int eavsr_nchw_f32_to_nhwc_h16(const float* in, void* out, int32_t n, int32_t c, int32_t hw, int32_t dtype,
                               void* stream);
int eavsr_nhwc_h16_to_nchw_f32(const void* in, const float* residual, float* out, int32_t n, int32_t c,
                               int32_t hw, int32_t dtype, void* stream);
/* NHWC 16-bit RCAB tail: out = r * scale[n,c] + x  (scale fp32) */
int eavsr_scale_residual_h16(const void* r, const float* scale, const void* x, void* out, int32_t n,
                             int32_t c, int32_t hw, int32_t dtype, void* stream);

/* ============================================================================================
 * EXPERIMENTAL -- exported by the LAB build only (`python -m eavsr_amd.build --lab`, -DEAVSR_LAB=1; eavsr_lab_build() == 1).
 * Schedules that were built, measured against the stable ones above and retired; kept because DESIGN.md / docs/history quote
 * their A/B figures.  Same boundary contract; no stability promise; a default build does not contain them.
 * (Also lab-only, without an entry point of its own: the channel-attention prologue of eavsr_conv3x3_wino4_f32 -- desc.ca_scale
 * with that entry point returns -2 on a default build.)
 * ============================================================================================ */
/* One pyramid level of MultiAdSTN's residual-flow refinement as ONE kernel (networks.py:604-619): AdaptBlock2_3x3
 * (front end `concat` + `concat2`, the 3x3 heads transform_matrix_conv (4) ++ translation_conv (2), the affine -> 18 offsets
 * expansion, networks.py:334-348) followed by TransOffsetworelu's 3x3 conv 18 -> 2 (networks.py:566-571).
 * x, h_hr (n, c, h, w); w1 (2c,1,3,3) b1 (2c); w2 (c,2,3,3) b2 (c); w_heads (6, c, 3, 3) b_heads (6); w_trans (2, 18, 3, 3)
 * b_trans (2); out (n, 2, h, w).  Replaces eavsr_adapt_frontend_f32 + eavsr_conv3x3_smallco_f32 (64 -> 6) +
 * eavsr_affine_offsets_f32 + eavsr_conv3x3_smallco_f32 (18 -> 2) and their three HBM round trips. */
int eavsr_flow_level_f32(const float* x, const float* h_hr, const float* w1, const float* b1, const float* w2,
                         const float* b2, const float* w_heads, const float* b_heads, const float* w_trans,
                         const float* b_trans, float* out, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/* The round-1 NCHW DCNv2 with all nine bf16 partial products (csrc/dcnv2_x9.hip): arguments of eavsr_dcnv2_f32, weight_x9 from
 * eavsr_pack_dcn_weight_x9 (stable: eavsr_dcnv2_il_f32 reads the same slabs); w % 4 == 0, 16-byte aligned x, else -2. */
int eavsr_dcnv2_f32x9(const float* x, const float* offset, const float* mask,
                      const void* weight_x9, const float* bias, float* out,
                      int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                      int32_t deform_groups, void* stream);

/* The same operation, arguments and arithmetic as eavsr_dcnv2_il_f32, scheduled wave-specialised (csrc/dcnv2_ws.hip): four
 * sampler waves (positions, LDS gathers, blend, 3-way bf16 split -> LDS stage) and four contractor waves (MFMAs, LDS-DMA,
 * stores) per workgroup, so that the two waves of a SIMD use different issue ports. */
int eavsr_dcnv2_ws_f32(const float* x_il8, const float* offset_or_heads, const float* mask, const void* weight_x9,
                       const float* bias, float* out, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                       int32_t deform_groups, int32_t nprod, int32_t heads, void* stream);

/* Opt-in "bf16x9" form of the 3x3 convolution (same descriptor, same tensors, same epilogue): both fp32 operands
 * are split EXACTLY into three bf16 terms and all nine partial products are accumulated in fp32 on
 * v_mfma_f32_32x32x16_bf16 (see eavsr_dcnv2_f32x9).  weight_x9 comes from eavsr_pack_dcn_weight_x9 on the
 * (cout, cin, 3, 3) weight (desc->weight_packed is ignored).  Requires ksize 3, w % 4 == 0, 16-byte aligned sources
 * with channels % 8 == 0, no fused channel-attention prologue, and a problem size that runs 32-row tiles
 * (eavsr_conv2d_tile_rows); returns -2 otherwise and the caller uses eavsr_conv2d_f32.                          */
int eavsr_conv3x3_f32x9(const eavsr_conv2d_desc* desc, const void* weight_x9, void* stream);

/* The 3x3 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores (same descriptor, tensors and epilogue;
 * 2.25x fewer multiplications; fp32 arithmetic throughout, as cuDNN / MIOpen run fp32 3x3 convolutions by default).
 * weight_wino: eavsr_wino_weight_elems(cout, cin) floats written by eavsr_pack_conv_weight_wino from the
 * (cout, cin, 3, 3) weight (desc->weight_packed is ignored).  8 x 32-pixel tiles: chan_partial has
 * eavsr_conv3x3_wino_tiles(h, w) rows per sample.  Requires ksize 3, w % 4 == 0, 16-byte aligned sources with
 * channels % 8 == 0; the fused channel-attention prologue (ca_scale / ca_x / ca_out of the descriptor) is applied in
 * the input transform and needs a single source with cin <= 256.  Returns -2 otherwise (call eavsr_conv2d_f32).  */
int64_t eavsr_wino_weight_elems(int32_t cout, int32_t cin);
int eavsr_pack_conv_weight_wino(const float* weight, float* weight_wino, int32_t cout, int32_t cin, void* stream);
int32_t eavsr_conv3x3_wino_tiles(int32_t h, int32_t w);
int eavsr_conv3x3_wino_f32(const eavsr_conv2d_desc* desc, const float* weight_wino, void* stream);

/* conv3x3 -> ReLU -> conv3x3 of one RCAB (RCABlock.forward, models/networks.py:461-462, mode 'CRC') as ONE launch in the 16-bit
 * modes (ABI 26): r = conv2(ReLU(conv1(x) + bias1)) + bias2 on 16-bit NHWC tensors, both weights in the packed form of
 * eavsr_pack_conv3x3_c64_h16, the intermediate rounded to 16 bits in LDS (never in HBM) and zero outside the image (the second
 * convolution's own padding).  r is bit-identical to two eavsr_conv3x3_c64_h16 launches.  chan_partial (nullable): (n,
 * eavsr_rcab_h16_partial_rows(n, h, w), 64) fp32, one row per workgroup of the launch and sample (zeros where a workgroup has no
 * tile of the sample); the rows of a sample add up to the channel sums of r over its pixels. */
int32_t eavsr_rcab_h16_partial_rows(int32_t n, int32_t h, int32_t w);
int eavsr_rcab_convs_h16(const void* x, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2,
                         void* out, float* chan_partial, int32_t n, int32_t h, int32_t w, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAVSR_HIP_H */
