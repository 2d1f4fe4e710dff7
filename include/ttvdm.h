/*
 * ttvdm.h -- C ABI of libttvdm.so: the MI355X (gfx950) kernels behind the SVD denoise hot path
 * of Kiteretsu77/This_and_That_VDM.
 *
 * The reference has NO native boundary for this path: it is eager PyTorch calling stock
 * Conv2d/Conv3d/Linear/GroupNorm/LayerNorm/SDPA through diffusers==0.25.1.  The
 * entry points below are therefore the operators that path lowers to; each cites
 * the reference site(s) whose work it replaces.  Citations are relative to /root/reference.
 *
 * Conventions
 *   - plain C: pointers + sizes only.  Every pointer is a DEVICE pointer unless marked host.
 *   - no ownership transfer, no allocation, no host sync inside: stream-ordered and stateless,
 *     hence safe to capture in a hipGraph and thread-safe per stream.
 *   - return 0 on success, a negative TT_E* code otherwise (never throws).
 *   - activations are token-major ("NHWC"): [frames*batch, h*w, C], C contiguous.
 *   - `dtype` selects the storage type of activations/weights: TT_BF16 or TT_F16; all
 *     accumulation, norm statistics, softmax and small vectors (bias, FiLM rows) are fp32.
 *     TT_F32 is the reference-precision mode: the SAME kernels instantiated on fp32 storage with the
 *     exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate).  It exists so the
 *     whole launch sequence can be checked against the fp32 CPU reference at rtol 1e-3 / atol 1e-4
 *     (16-bit operand rounding alone exceeds that); it is not the benchmarked path.
 */
#ifndef TTVDM_H
#define TTVDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tt_stream_t; /* hipStream_t */

enum { TT_BF16 = 0, TT_F16 = 1, TT_F32 = 2 };
enum { TT_OK = 0, TT_EINVAL = -1, TT_EUNSUPPORTED = -2, TT_ELAUNCH = -3 };

/* library/ABI version and target arch string ("gfx950"). */
int tt_abi_version(void);
const char* tt_target_arch(void);
/* text of the last error on the calling thread (host pointer, valid until the next call). */
const char* tt_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * tt_gemm: out = epilogue( gather(A) x W^T ).  One kernel family for every dense contraction:
 *   mode 0  Linear            nn.Linear sites: transformer_temporal.py:326,373 (proj_in/out) and the
 *                             diffusers Attention/FeedForward linears built at :240-261; 1x1 convs:
 *                             ResnetBlock2D.conv_shortcut, temporal_controlnet.py:614-622 zero-convs.
 *   mode 1  Conv2d 3x3 pad 1  ResnetBlock2D.conv1/conv2 (unet_3d_blocks.py:2094 etc.),
 *                             Downsample2D stride 2 (:2117-2123), Upsample2D nearest x2 + conv
 *                             (:2223,2332; the upsample is an index map, never materialised),
 *                             conv_in / conv_in_concat / conv_out
 *                             (unet_spatio_temporal_condition.py:455,528; temporal_controlnet.py:580).
 *   mode 2  Conv3d (3,1,1)    TemporalResnetBlock.conv1/conv2 (diffusers; reached from
 *                             unet_3d_blocks.py:1891-2316): 3 taps along the frame axis.
 * W is [n, taps*(k0+k1)] row-major with k index (tap, source, channel); two sources implement
 * torch.cat([h, skip], dim=1) (unet_3d_blocks.py:2242,2352) without a concat buffer.
 * Epilogue, in this order (every term optional):
 *   acc *= 1/sigma of the LayerNorm-folded operand row (ln_fold, below)
 *   v = (acc + bias[n]) * acc_scale + rowvec[m / rowvec_rows][n]        (row index taken modulo rowvec_mod when that is > 0)
 *   geglu: v = v_value * gelu_erf(v_gate)      (W rows pre-interleaved in 16-row groups: 8 value, 8 gate)
 *   v += residual[m][n];  v = alpha*blend[m][n] + (1-alpha)*v   (AlphaBlender, video branch)
 * `out` may alias `residual` (same pointer and stride: in-place update of the hidden states -- every element's residual is
 * read by the lane that writes it, in the tile kernels and in the split-K reduction); it must not alias a0 / a1 / blend.
 * ---------------------------------------------------------------------------------------------- */
typedef struct TtGemmArgs {
  const void* a0; const void* a1;      /* activation sources; a1 NULL if k1 == 0 */
  int32_t k0, k1;                      /* channels per tap taken from a0 / a1 (multiples of 8) */
  int64_t lda0, lda1;                  /* row strides, elements */
  const void* w; int64_t ldw;          /* [n, taps*(k0+k1)] */
  int32_t m, n;                        /* n multiple of 4 */
  int32_t mode;                        /* 0 linear, 1 conv3x3, 2 tconv3 */
  int32_t nimg, hin, win, hout, wout, stride, upsample;   /* mode 1 (hin/win = stored input size) */
  int32_t frames, hw;                  /* mode 2: row = (b*frames + f)*hw + p */
  const float* bias;
  float acc_scale;
  const float* rowvec; int32_t rowvec_rows; int64_t ld_rowvec;
  int32_t geglu;
  const void* residual; int64_t ld_res;
  const void* blend; int64_t ld_blend; float alpha;
  void* out; int64_t ldo; int32_t out_f32;
  int32_t out_col_hw, out_col_hwp;     /* if hw>0: column c -> (c/hw)*hwp + c%hw (padded V^T sequences) */
  int32_t dtype;
  void* ws; int64_t ws_bytes;          /* optional caller-owned scratch for split-K (tt_gemm_ws_bytes); NULL = never split */
  /* Fused LayerNorm of one operand (nn.LayerNorm sites inside Basic/TemporalBasicTransformerBlock, reached from
   * transformer_temporal.py:342-365; mode 0, k1 == 0, k0 = the LayerNorm width):
   *   ln_fold 1   acc[m][n] is multiplied by 1/sqrt(var(a0 row m) + ln_eps)      -> out = LN(A) W^T
   *   ln_fold 2   acc[m][n] is multiplied by 1/sqrt(var(w  row n) + ln_eps)      -> out = A LN(W)^T  (swapped V^T projection)
   * before bias / acc_scale / the other epilogue terms.  The statistics are taken inside the K loop from the operand
   * fragments themselves (no separate pass).  The CALLER folds the affine part: the un-normalised operand is multiplied by
   * (weight * gamma), centred over k so that each weight row sums to zero (then x W^T == (x - mean) W^T), and beta
   * enters through `bias` (this_and_that_vdm_amd/packing.py:fold_layernorm). */
  int32_t ln_fold; float ln_eps;
  /* out_fp8 != 0: `out` receives OCP e4m3 bytes (row stride ldo BYTES, saturating conversion) instead of 16-bit values:
   * the Q | K and V^T operands of tt_attention's fp8 path (BASELINE config 5).  Plain mode-0 linears only (ln_fold and
   * out_col_hw are allowed; no geglu / residual / blend / rowvec / out_f32); dtype stays the 16-bit type of a0 / w. */
  int32_t out_fp8;
  /* ABI 7.  rowvec_mod > 0: row m takes rowvec[(m / rowvec_rows) % rowvec_mod] -- a PERIODIC row vector.  rowvec_rows = 1,
   * rowvec_mod = 2 gives even and odd rows their own vector: the temporal transformer block's output projection serves the rows of
   * both context classes (quirk Q3 pairs query pixel p with context p mod 2; the class whose context is all zeros also receives the
   * cross-attention's to_out bias, transformer_temporal.py:342-365 via TemporalBasicTransformerBlock) in ONE launch instead of one
   * launch per class on strided row views.  0: the plain form above. */
  int32_t rowvec_mod;
  /* ABI 8.  stats_out != NULL: beside its output the kernel writes, per row tile of R = tt_gemm_stats_rows(args) output rows and per
   * column, the sum and the sum of squares of the values it STORES (after rounding to the storage type):
   *     stats_out[(t * 2 + 0) * n + c] = sum_{r in tile t} out[r][c]        stats_out[(t * 2 + 1) * n + c] = sum of squares
   * fp32, [ceil(m / R)][2][n] floats, no atomics (fixed summation order: bit-reproducible).  This is the producer half of GroupNorm
   * (nn.GroupNorm(32) in ResnetBlock2D / TemporalResnetBlock / TransformerSpatioTemporalModel.norm -- unet_3d_blocks.py:1891-2316,
   * transformer_temporal.py:323-326): the tensor a GroupNorm reads is the output of a conv / Linear launch, whose epilogue has every
   * value in registers; tt_groupnorm_tiles turns the tile sums into the normalised tensor in ONE pass over it, without a
   * statistics pass.  Only routes and shapes for which tt_gemm_stats_rows(args) > 0 (TT_EUNSUPPORTED otherwise). */
  float* stats_out;
  /* rows of the consumer's GroupNorm segment (one image: h*w; the frames of a video: frames*h*w), a hint that lets the route pick a
   * statistics tile height R that divides it: the split-K routes (coarse UNet levels; the sums are taken by the reduction kernel, any
   * height works) use the largest divisor of stats_seg up to 128; the tiled template falls back from its tile height (128) to the
   * 32 / 64 rows of one wave row when only that divides stats_seg (448-row images, 1568-row videos).  0: the route's tile height. */
  int32_t stats_seg;
  /* ABI 9.  gn_out != NULL: the launch ALSO writes act(GroupNorm_32(out)) to gn_out (same storage type, row stride ld_gn elements), with
   * statistics per segment of stats_seg output rows (one image, or the frames of a video), gamma / beta fp32 [n], SiLU when gn_silu != 0
   * -- the GroupNorm that reads this launch's output (ResnetBlock2D.norm2 after conv1, TemporalResnetBlock.norm1 / norm2 --
   * unet_3d_blocks.py:1891-2316 via diffusers ResnetBlock2D / TemporalResnetBlock).  Only where the launch ends in the split-K
   * reduction pass anyway (the two coarsest UNet levels: 784 / 3136 rows), which then owns a whole (segment, group) per block: it sums
   * the fp32 slabs, applies the epilogue, stores `out`, and normalises the stored values from registers -- one launch instead of the
   * reduction plus a GroupNorm launch.  tt_gemm_gn_fused(args) != 0 says whether `args` qualifies (TT_EUNSUPPORTED otherwise);
   * exclusive with stats_out. */
  void* gn_out; int64_t ld_gn;
  const float* gn_gamma; const float* gn_beta;
  float gn_eps; int32_t gn_silu;
  /* ABI 11.  TT_F32 with tt_gemm_set_f32_split(1) only: an operand that is a constant of the request (packed weights) may be handed over
   * PRE-SPLIT, so the kernel does not convert it again in every launch: bit 0 = a0 / a1, bit 1 = w.  A pre-split matrix has the shape,
   * strides and element size of the fp32 matrix it replaces; every aligned group of 4 consecutive k-elements (16 bytes) holds the eight
   * fp16 values  h0 h1 h2 h3 l0 l1 l2 l3  with  h = fp16(x 2^-8), l = fp16((x - 2^8 h) 2^3)  of its four fp32 values
   * (this_and_that_vdm_amd/packing.py:presplit_f32).  Not together with the LayerNorm statistics of that operand (ln_fold 1 with
   * bit 0, ln_fold 2 with bit 1).  0 = plain fp32 operands. */
  int32_t presplit;
} TtGemmArgs;
int tt_gemm(const TtGemmArgs* args, tt_stream_t stream);
/* 1 if tt_gemm serves `args` with gn_out (a split-K plan, 16-bit storage, stats_seg rows per segment dividing m and short enough for one
 * block -- up to 816 rows at 1280 channels, 1632 at 640 --, n / 32 channels per group a multiple of 4, plain row-major 16-bit output), else 0.
 * Host-only; set ws / ws_bytes as for tt_gemm. */
int32_t tt_gemm_gn_fused(const TtGemmArgs* args);
/* rows per statistics tile R if tt_gemm can serve `args` with stats_out (see stats_seg; m must be a multiple of R -- on the tiled
 * template a ragged last tile is fine, its rows beyond m add nothing); 0: this problem's route has no statistics epilogue (GEGLU,
 * fp8 / fp32 / transposed outputs, the persistent kernels, a split-K route without stats_seg) -- leave stats_out NULL.  Host-only;
 * call with ws / ws_bytes set as for tt_gemm (the plan depends on them). */
int32_t tt_gemm_stats_rows(const TtGemmArgs* args);
/* which tile configuration tt_gemm will use for this problem: cfg[0..6] = BM, BN, BK, ring stages, waves along M,
 * waves along N, split-K factor.  Lets a profiler name the kernel instance
 * (gemm_kernel<dtype, BM, BN, BK, stages, wavesM, wavesN, mode>) a launch maps to.  Host-only, no launch. */
int tt_gemm_plan(const TtGemmArgs* args, int32_t cfg[7]);
/* bytes of fp32 scratch tt_gemm would like for this problem (0 = no split-K planned).  Problems with few output
 * tiles and a long K (convs at the coarsest UNet levels) are split over K; the slabs are summed in a fixed order,
 * so results stay bit-reproducible.  Without (enough) workspace the un-split plan runs instead. */
size_t tt_gemm_ws_bytes(const TtGemmArgs* args);
/* tuning knob: force tile configuration `cfg` (index into the table in gemm.hip) for every tt_gemm call of this
 * process; -1 restores the built-in heuristic.  Also settable by the TT_GEMM_CFG environment variable. */
int tt_gemm_set_tile_override(int32_t cfg);
/* tuning knob: route the 320 x 320 linears with m >= 4096 (mode 0, one source, bias / residual / self-blend epilogue) to
 * the persistent W-in-registers streaming kernel: 0 never, 1 always, 2 (the default) from 131 072 rows on -- 64x112 latents, where the
 * big-tile kernel walks 3.06 rounds of tiles; at 32x56 it is faster alone and slower inside the step.  Also TT_GEMM_SQ320=0|1|2. */
int tt_gemm_set_streaming_square(int32_t on);
/* tuning knob for the N = 320 t big-tile kernels (gemm_w320.hip), for A/B measurements and tests; also TT_GEMM_W320=0|1|2|3|4:
 *   0  keep these problems on the tiled kernels;
 *   1  (default) problems with >= 180 row tiles of 256 (the finest UNet level: ResnetBlock2D / temporal convs, shortcuts, proj_in/out,
 *      to_out, FF2, LayerNorm-folded Q projections -- svd/diffusion_arch/unet_3d_blocks.py:2094,2212,2311,
 *      svd/diffusion_arch/transformer_temporal.py:323-376) go to the 256 x 320 kernel; conv3x3 problems with fewer rows but >= 180
 *      tiles of 128 rows (the second level at 32x56 latents) to its 128 x 320 variant (measured: +9..16 % on those convs, nothing
 *      on the linears / temporal convs of that level, which therefore stay on the tiled kernels); long-K problems of the two
 *      coarsest levels (FF2 at 3136 rows, conv3x3 at 3136 / 784 rows: 100 / 28 tiles of 128 x 320) go to the variant with the K
 *      loop split over 2..16 workgroups per tile (fp32 slabs in the tt_gemm_ws_bytes workspace, summed in a fixed order);
 *   2  the 256 x 320 kernel only;    3  the 128 x 320 variant for every gather mode (its linear / temporal-conv / LayerNorm paths);
 *   4  as 1 without the split-K route. */
int tt_gemm_set_big_tile(int32_t on);
/* ABI 11 (round 6).  How TT_F32 launches of tt_gemm form their products (process-wide; also TT_F32_SPLIT=0|1 in the environment):
 *   0  (default) exact-fp32 MFMA, v_mfma_f32_32x32x2_f32: bitwise an fmaf chain in k order, 157 TFLOP/s peak;
 *   1  "split16": every fp32 operand x is split on the fly into two fp16 parts on different binary scales, h = fp16(x 2^-8) and
 *      l = fp16((x - 2^8 h) 2^3), and a b ~ 2^16 a_h b_h + 2^5 (a_h b_l + a_l b_h) runs as three v_mfma_f32_32x32x16_f16 into two fp32
 *      accumulators -- max(2^-22 |x|, 2^-28) per operand, operands up to |x| < 2^24 (beyond that the hi part overflows to inf),
 *      3 x 8 passes per 32 x 32 x 16 block instead of 8 x 16.  Storage, epilogues,
 *      LayerNorm statistics and accumulation stay fp32; the mode meets the north-star tolerance (rtol 1e-3 / atol 1e-4 against the
 *      reference's CPU fp32 forward, svd/unet_spatio_temporal_condition.py:363-536) at about half of the exact mode's step time.
 *      tt_attention on fp32 storage (head_dim 64) follows the same switch; few-tile long-K problems take a split-K plan in this mode
 *      (fp32 slabs in the tt_gemm_ws_bytes workspace, summed in a fixed order). */
int tt_gemm_set_f32_split(int32_t on);

/* ------------------------------------------------------------------------------------------------
 * tt_conv3x3: Conv2d 3x3 / stride 1 / pad 1 with the GroupNorm (+SiLU) of its INPUT fused in -- the ResnetBlock2D convs
 * (norm1 -> SiLU -> conv1 (+ time-embedding FiLM), norm2 -> SiLU -> conv2 (+ shortcut); reached from
 * unet_3d_blocks.py:1891-2316) without a normalised copy of the activation and with the input staged in LDS as a spatial
 * patch with halo, read 9 times (once per tap) from there:
 *   out[p][n] = bias[n] + rowvec[p / rowvec_rows][n] + residual[p][n] + sum_{tap,c} act(x[p+tap][c]) W[n][(tap, c)]
 *   act(v) = silu?(v * gn_scale[image][c] + gn_shift[image][c])     (gn_* = outputs of tt_groupnorm_stats; NULL: identity)
 * Halo pixels outside the image contribute exact zeros (zero padding of the ACTIVATED tensor, as in the reference).
 * x0 | x1: token-major sources (virtual channel concat, c0 and c1 multiples of 64); W as for tt_gemm mode 1.
 * Serves the image sizes tt_conv3x3_supported() accepts (h x w tiled by 16x8, 8x16 or 8x14 rectangles); everything else
 * (stride 2, upsampling, tiny images, TT_F32) stays on tt_groupnorm_apply + tt_gemm mode 1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct TtConvArgs {
  const void* x0; const void* x1; int32_t c0, c1; int64_t ld0, ld1;
  const void* w; int64_t ldw;                 /* [n, 9*(c0+c1)], k index (tap, source, channel) */
  int32_t nimg, h, w_img, n;
  const float* gn_scale; const float* gn_shift; int32_t silu;     /* fp32 [nimg, c0+c1] each, or both NULL */
  const float* bias;
  const float* rowvec; int32_t rowvec_rows; int64_t ld_rowvec;
  const void* residual; int64_t ld_res;
  void* out; int64_t ldo;
  int32_t dtype;                              /* TT_BF16 or TT_F16 */
} TtConvArgs;
int tt_conv3x3_supported(int32_t h, int32_t w, int32_t c0, int32_t c1, int32_t n, int32_t dtype);
int tt_conv3x3(const TtConvArgs* args, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * tt_attention: softmax(Q K^T / sqrt(d)) V with online softmax on MFMA tiles; replaces
 * F.scaled_dot_product_attention inside diffusers AttnProcessor2_0 for
 *   mask 0  spatial self-attention            (BasicTransformerBlock.attn1; transformer_temporal.py:353)
 *   mask 1  spatial cross-attention, S<=pad   (attn2; context of batch n/frames)
 *   mask 2  temporal cross-attention          (TemporalBasicTransformerBlock.attn2, :361-365) with the
 *           reference's (hw,B)-flattened context pairing reproduced: query (b,p) sees context
 *           (b*hw + p) % ctx_batches  (transformer_temporal.py:316-319).
 * q [nseq*lq, ldq] (+ head*d), k [rows, ldk] (+ head*d), vt = V transposed [heads*d, ldvt].
 * ---------------------------------------------------------------------------------------------- */
typedef struct TtAttnArgs {
  const void* q; int64_t ldq;
  const void* k; int64_t ldk;
  const void* vt; int64_t ldvt;
  void* out; int64_t ldo;
  int32_t nseq, lq, heads, head_dim;   /* head_dim 64 or 128 */
  int32_t mask;                        /* 0,1,2 as above */
  int32_t lk;                          /* valid keys per sequence/context */
  int32_t k_seq_stride, v_seq_stride;  /* rows of k / columns of vt per sequence (mask 0) or per context (1,2) */
  int32_t frames, ctx_batches;         /* masks 1,2 */
  int32_t dtype;
  int32_t batch0;                      /* masks 1,2: batch index of sequence 0 (a launch may cover a sub-range of the batch) */
  /* fp8 != 0 (mask 0 only): q, k, vt hold OCP e4m3 bytes (strides in bytes = elements; written by tt_gemm out_fp8),
   * QK^T and PV run on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales) with P = 8 * exp2(s c - m) converted to e4m3, m a
   * lazily updated reference that keeps P inside e4m3's range (the factor cancels against the row sum); softmax statistics
   * and the output accumulation stay fp32, `out` is `dtype` (16-bit).  Scales are 1: the
   * operands are LayerNorm-ed projections, far inside e4m3's +-448.  Its tolerance is e4m3's (3 mantissa bits), see
   * tests/test_ops_gpu.py::test_attention_fp8. */
  int32_t fp8;
  /* Fused query projection (ABI 6; masks 1 / 2, head_dim 64, 16-bit dtype): when qx != NULL, `q` is ignored (may be NULL) and every
   * block computes its own queries  Q = LN(x rows) Wq^T + bq  in front of the key loop -- the nn.Linear to_q of the cross-attention
   * (BasicTransformerBlock.attn2 / TemporalBasicTransformerBlock.attn2, reached from transformer_temporal.py:353-365) and the
   * LayerNorm in front of it, without a Q tensor or a launch of their own.  qx [nseq*lq rows, ldqx] holds the UN-normalised hidden
   * states (rows addressed like q: query r of sequence s is row s*lq + r), qc = the LayerNorm width = columns of wq.
   * wq [heads*64, ldwq] and bq [heads*64] are the LayerNorm-folded projection (packing.fold_layernorm) with bits 2 and 3 of the
   * row index swapped inside every group of 16 rows (packing.permute_q_rows): the projection's MFMA accumulators then ARE the
   * Q^T operand fragments of the score product.  qx and wq hold elements of `dtype`, bq is fp32; all three start on 16-byte
   * boundaries (TT_EINVAL otherwise), wq spans heads*64 rows of qc elements. */
  const void* qx; int64_t ldqx;
  const void* wq; int64_t ldwq;
  const float* bq;
  int32_t qc; float ln_eps;
  /* ABI 10.  v_rows != 0 (mask 0, head_dim 64, 16-bit dtype): `vt` holds V ITSELF -- [key rows, ldvt] (+ head*d), rows addressed like k
   * (sequence s starts at row s * v_seq_stride) -- as it leaves a fused Q | K | V projection (nn.Linear to_q / to_k / to_v of
   * BasicTransformerBlock.attn1 in ONE tt_gemm launch; transformer_temporal.py:353 via diffusers Attention): no V^T projection launch.
   * The kernel transposes on the way out of LDS (ds_read_b64_tr_b16). */
  int32_t v_rows;
} TtAttnArgs;
int tt_attention(const TtAttnArgs* args, tt_stream_t stream);

/* temporal self-attention over the frame axis (seq = frames <= 32), one 16/32-lane group per
 * (batch, pixel, head); replaces TemporalBasicTransformerBlock.attn1 incl. its two
 * [(B F),hw,C] <-> [(B hw),F,C] permute copies (diffusers; reached from transformer_temporal.py:361).
 * qkv [batch*frames*hw, ldqkv] holds Q | K | V at column offsets 0, C, 2C. */
int tt_temporal_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int32_t batch, int32_t frames,
                          int32_t hw, int32_t heads, int32_t head_dim, int32_t dtype, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) split in statistics -> affine (+SiLU); nn.GroupNorm sites: ResnetBlock2D /
 * TemporalResnetBlock norm1/norm2, transformer_temporal.py:234,323, unet...:244,526.
 * `frames_per_group` = 1 for per-image statistics, = F for the temporal block (stats over F*h*w).
 * Two sources = virtual channel concat.  ws must hold tt_groupnorm_ws_bytes().
 * tt_groupnorm_stats writes per-(image, channel) scale/shift fp32 [nimg, C] each:
 *     y = x*scale + shift  ==  (x-mean)*rstd*gamma + beta
 * ---------------------------------------------------------------------------------------------- */
size_t tt_groupnorm_ws_bytes(int32_t nimg, int32_t hw, int32_t c);
int tt_groupnorm_stats(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                       int32_t frames_per_group, const float* gamma, const float* beta, float eps,
                       float* scale, float* shift, void* ws, size_t ws_bytes, int32_t dtype, tt_stream_t stream);
/* GroupNorm(32) (+SiLU) of x [nseg * seg_rows, c] from the tile sums its producer left (TtGemmArgs.stats_out, stat_rows = the
 * tt_gemm_stats_rows of that launch): ONE pass over x, one launch, for per-image statistics (seg_rows = h*w) and for the cross-frame
 * statistics of TemporalResnetBlock (seg_rows = frames*h*w) alike.  seg_rows must be a multiple of stat_rows (a segment is a whole
 * number of statistics tiles): tt_groupnorm_tiles_supported.  Replaces the statistics pass of tt_groupnorm_small / tt_groupnorm_stats
 * for nn.GroupNorm sites whose input is the direct output of one tt_gemm launch (unet_3d_blocks.py:1891-2316). */
int tt_groupnorm_tiles_supported(int32_t seg_rows, int32_t c, int32_t stat_rows, int32_t dtype);
int tt_groupnorm_tiles(const void* x, int32_t c, const float* stats, int32_t stat_rows, int32_t nseg, int32_t seg_rows,
                       const float* gamma, const float* beta, float eps, int32_t silu, void* y, int64_t ldy, int32_t dtype,
                       tt_stream_t stream);
/* y[n,p,0:c0+c1] = act(x*scale+shift), act = SiLU if silu else identity; y has row stride ldy (>= c0+c1,
 * extra columns untouched). */
int tt_groupnorm_apply(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                       const float* scale, const float* shift, int32_t silu, void* y, int64_t ldy,
                       int32_t dtype, tt_stream_t stream);

/* Statistics + apply in ONE launch for per-image GroupNorm (frames_per_group = 1): y = act(group_norm(x0 | x1)), no workspace, no
 * scale/shift arrays.  Images of 256 to 4096 rows: one block per (image, slice of consecutive groups) -- a group's statistics need only
 * its own channels, so blocks never exchange anything and x is read from HBM once (the apply pass re-reads the slice from L2).
 * Smaller images (at most 640 KiB per image: the 8x14 / 4x7 levels at 256x448): one block per image and row part, each recomputing
 * the image's statistics.  Same result as tt_groupnorm_stats + tt_groupnorm_apply up to fp32 summation order; bit-reproducible.
 * ResnetBlock2D norm1 / norm2 (diffusers resnet.py), TransformerSpatioTemporalModel.norm (transformer_temporal.py:323),
 * conv_norm_out (unet...:526). */
int tt_groupnorm_small_supported(int32_t hw, int32_t c, int32_t dtype);
int tt_groupnorm_small(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                       const float* gamma, const float* beta, float eps, int32_t silu, void* y, int64_t ldy,
                       int32_t dtype, tt_stream_t stream);

/* LayerNorm over the last dim (eps, affine); nn.LayerNorm inside Basic/TemporalBasicTransformerBlock.
 * Optional fused frame-position embedding (transformer_temporal.py:358-359): when rowvec != NULL,
 * x' = x + rowvec[(row / rows_per_vec) % nvec] is written to xsum_out and normalised. */
int tt_layernorm(const void* x, int64_t ldx, int32_t rows, int32_t c, const float* gamma, const float* beta,
                 float eps, const float* rowvec, int32_t rows_per_vec, int32_t nvec, void* xsum_out,
                 void* y, int64_t ldy, int32_t dtype, tt_stream_t stream);

/* Weight packing on the device (packing.zero_sum_round; the LayerNorm-folded projections of Basic / TemporalBasicTransformerBlock reached
 * from transformer_temporal.py:342-365): rounds the rows of w (fp32 [n, k], each summing to ~0) to the 16-bit `dtype` such that every ROUNDED
 * row sums to exactly zero -- binade by binade from `hi` down to max(lo, hi - 48) (the largest / smallest binade exponent of the rounded
 * non-zero values of the whole matrix), up to round(|row sum| / ulp) elements of a binade move by one ulp.  One block per row; the result is
 * bit-identical to the host implementation.  k <= 16384. */
int tt_zero_sum_round(const float* w, int64_t ldw, int32_t n, int32_t k, int32_t hi, int32_t lo, void* out, int64_t ldo,
                      int32_t dtype, tt_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * small dense layers on <=32 rows, fp32 activations (time / add / FiLM / frame-position MLPs:
 * unet...:416-432, ResnetBlock2D.time_emb_proj, transformer_temporal.py:338):
 *   y[r][n] = act_out( sum_k act_in(x[r][k]) * W[n][k] + bias[n] ),  act: 0 none, 1 SiLU.
 * W is `dtype`, x/y/bias fp32.  accumulate != 0 adds into y.  k % 8 == 0 and k <= 8192 (the activated rows of x are staged
 * in LDS four at a time: 4 x k fp32; TT_EUNSUPPORTED beyond -- the path's widest MLP input is 1280).
 * ---------------------------------------------------------------------------------------------- */
int tt_small_linear(const float* x, int64_t ldx, int32_t rows, int32_t k, const void* w, int64_t ldw, int32_t n,
                    const float* bias, int32_t act_in, int32_t act_out, int32_t accumulate, float* y, int64_t ldy,
                    int32_t dtype, tt_stream_t stream);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, shift=0): out[r] = [cos(t_r*f_i) | sin(t_r*f_i)], fp32.
 * `t` device fp32 [rows]. */
int tt_timestep_embedding(const float* t, int32_t rows, int32_t dim, float* out, int64_t ldo, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * denoise-loop glue (pipeline_stable_video_diffusion_controlnet.py:630-635,704-709):
 * tt_prep_model_input: x[b,f,p,0:4] = latents[f,:,p] * c_in (CFG: same latents for every b),
 *   x[...,4:8] = image_latents[b,f,:,p], x[...,8:12] = cond[f,:,p] (if cond), rest of cpad zero.
 *   latents/image_latents/cond are fp32 NCHW-per-frame as the pipeline holds them.
 * tt_cfg_euler_step: eps[b,f,p,0:4] fp32 token-major (conv_out) ->
 *   v = u + g_f*(c-u);  x0 = v*(-s/sqrt(s^2+1)) + x/(s^2+1);  x += (x-x0)/s*(s_next-s)   (in place, fp32)
 * tt_cfg3_euler_step: the use_instructpix2pix variant (:698-702), eps batch of 3 in the reference's order
 *   (first-frame e1, cond c, uncond u):  v = u + g_f*(c-u) + image_guidance_scale*(c-e1), then the same Euler update.
 * sigma scalars live on the device (sigmas[step], sigmas[step+1]) so a captured graph can be replayed
 * for every step.
 * ---------------------------------------------------------------------------------------------- */
int tt_prep_model_input(const float* latents, const float* image_latents, const float* cond, const float* sigmas,
                        int32_t step, int32_t batch, int32_t frames, int32_t h, int32_t w, int32_t cpad,
                        void* x, int32_t dtype, tt_stream_t stream);
int tt_cfg_euler_step(const float* eps, int32_t ld_eps, float* latents, const float* guidance, const float* sigmas,
                      int32_t step, int32_t batch, int32_t frames, int32_t h, int32_t w, tt_stream_t stream);
int tt_cfg3_euler_step(const float* eps, int32_t ld_eps, float* latents, const float* guidance, float image_guidance_scale,
                       const float* sigmas, int32_t step, int32_t frames, int32_t h, int32_t w, tt_stream_t stream);

/* layout plumbing at the drop-in boundary: NCHW (any float dtype code below) <-> token-major.
 * src_kind/dst_kind: 0 = dtype (bf16/f16), 1 = fp32. */
int tt_nchw_to_tokens(const void* src, int32_t src_f32, int32_t nimg, int32_t c, int32_t hw, void* dst, int64_t ld_dst,
                      int32_t dtype, tt_stream_t stream);
int tt_tokens_to_nchw(const void* src, int32_t src_f32, int64_t ld_src, int32_t nimg, int32_t c, int32_t hw, void* dst,
                      int32_t dst_f32, int32_t dtype, tt_stream_t stream);
/* y = a + b*scale (dtype, elementwise, n multiple of 8): ControlNet residual add, unet...:485-491,501-502. */
int tt_add_scaled(const void* a, const void* b, float scale, void* y, int64_t n, int32_t dtype, tt_stream_t stream);
/* y[r][c] = x[r][c] + rowvec[(r / rows_per_vec) % nvec][c]: the frame-position embedding added to the hidden states before
 * the temporal transformer block (transformer_temporal.py:358-359; the block's norm_in is folded into its first GEMM).
 * x, y `dtype` [rows, c] (c multiple of 8), rowvec fp32 [nvec, c]. */
int tt_add_rowvec(const void* x, int64_t ldx, int32_t rows, int32_t c, const float* rowvec, int64_t ld_rowvec,
                  int32_t rows_per_vec, int32_t nvec, void* y, int64_t ldy, int32_t dtype, tt_stream_t stream);

/* y[r, 0:cols] = softmax(x[r, 0:cols]) over fp32 scores, y[r, cols:cols_pad] = 0; y in `dtype` storage.  The attention of the
 * temporal VAE decoder's mid block (one head of 512 channels over h*w tokens per frame: diffusers
 * autoencoder_kl_temporal_decoder.py MidBlockTemporalDecoder, reached from
 * svd/pipeline_stable_video_diffusion_controlnet.py:257-283) runs as  scores = tt_gemm(Q, K, out_f32)  ->  tt_softmax_rows
 * ->  tt_gemm(P, V^T): head_dim 512 is outside tt_attention's 64 / 128. */
int tt_softmax_rows(const float* x, int64_t ldx, int32_t rows, int32_t cols, void* y, int64_t ldy, int32_t cols_pad,
                    int32_t dtype, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TTVDM_H */
