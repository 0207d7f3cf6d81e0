/*
 * hallo_amd.h -- C ABI of libhallo_amd.so: the gfx950 (MI355X) operator library behind the
 * Hallo denoising hot path.
 *
 * The reference (fudan-generative-vision/hallo) has no C/FFI boundary: its operator seam is
 * the diffusers attention-processor protocol and torch module calls.  Each entry point below
 * names the reference call sites it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - extern "C", plain device pointers and sizes, no torch types.
 *   - return 0 on success, -22 (EINVAL) on bad arguments, <= -1000 for a HIP launch error.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream.
 *   - no allocation inside; the caller owns all buffers for the duration of the call.
 *   - dtype: HALLO_F16 / HALLO_BF16 select the storage type of activations and weights;
 *     all accumulation is fp32.
 *   - activations are token-major ("NHWC"): [frames, H*W, C] with C contiguous.
 */
#ifndef HALLO_AMD_H
#define HALLO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALLO_F16 0
#define HALLO_BF16 1

#define HALLO_ACT_NONE 0
#define HALLO_ACT_SILU 1
#define HALLO_ACT_RELU 2
#define HALLO_ACT_GELU 3      /* erf GELU after the residual add (hallo_gemm only) */
#define HALLO_ACT_GELU_PRE 4  /* erf GELU of alpha * (acc + bias), BEFORE the residual add (hallo_gemm only) */

int hallo_abi_version(void);

/* Tuning / A-B switch (not part of the numerical contract): "gemm_variant" = 0 register-staged 128x128 kernel,
 * 1 / 2 direct-to-LDS 128x128 kernel with 1 / 2 LDS stages, 3 auto among those, 4 / 5 force the 256x320 / 128x320
 * big-tile kernel wherever applicable, 6 auto over all (default);
 * "split_k" = 0 / 1 (auto, default); "gn_fused" = 0 / 1 (single-launch GroupNorm for small feature maps, default 1);
 * "v3_min_tiles" = smallest grid the auto rule gives to the big-tile kernel; round 4: "gemm4" = 0 off / 1 auto rule / 2 every
 * problem csrc/gemm4.hip covers, "gemm4_min_nk", "gemm_stage_min_tiles", "split_k_max" (cap of the split-K factor).
 * Round 6: "splitk_nt" = 0 / 1 / 2 (non-temporal split-K slab stores (+ loads): A/B), "fp8_mx" = 1 (default) / 0 (hallo_gemm_fp8 on the MX-rate
 * scaled MFMA with unit block scales, or on the non-scaled bf16-rate form); "gemm_variant" 7 / 8 (persistent big tile) no longer exist.
 * Returns -22 for unknown names / values. */
int hallo_set_option(const char* name, int value);
/* Read an option back; "last_gemm_kernel" = the kernel the last hallo_gemm / hallo_conv3x3_nhwc call launched, as
 * 1000 * f + 100 * k + 10 * mode + s: k = 1 gemm_kernel / 2 gemm2_kernel / 3 gemm3_kernel, mode = 0 gemm / 1 conv3x3 /
 * 2 geglu, s = LDS stages (gemm2) or TM (gemm3), f = fused-LayerNorm form of gemm2 (0 / 1 / 2); + 10000 * e, e = row_parts epilogue form of gemm2 (0 / 1 / 2).  Used by bench.py to report achieved rates per kernel SYMBOL.  -22 = unknown. */
int hallo_get_option(const char* name);
/* ABI v8: the names of every option hallo_set_option accepts, comma-separated (static storage).  A host that caches launch
 * sequences (hipGraphs of a UNet evaluation) keys them on the VALUES of all of these. */
const char* hallo_option_names(void);

/* ------------------------------------------------------------------------------------------
 * hallo_gemm: C[M,N] = act( alpha * rowscale[m] * (A[M,K] . W[N,K]^T + bias) + residual )
 * Replaces torch Linear / 1x1 Conv2d everywhere on the path:
 *   Attention.to_q/to_k/to_v/to_out (diffusers, imported hallo/models/attention.py:22-23),
 *   Transformer3DModel.proj_in/proj_out (hallo/models/transformer_3d.py:199,242),
 *   TemporalTransformer3DModel.proj_in/proj_out (hallo/models/motion_module.py:295,306),
 *   ResnetBlock3D.conv_shortcut / time_emb_proj (hallo/models/resnet.py:390-408),
 *   zero_conv_full/face/lip + mask multiply + motion_scale sum (hallo/models/attention.py:846-903),
 *   TimestepEmbedding (hallo/models/unet_3d.py:588), ImageProjModel / AudioProjModel Linears.
 * geglu=1: W holds 2N rows (value rows then gate rows), output is value * gelu_erf(gate)
 *   = diffusers FeedForward/GEGLU (hallo/models/attention.py:601,905; motion_module.py:420).
 * batch>1 with element strides gives a strided-batched GEMM (VAE mid-block attention).
 * K, lda, ldb must be multiples of 8; pointers 16-byte aligned.
 */
typedef struct hallo_gemm_desc {
  const void* A; const void* B; void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int batch;
  int64_t stride_a, stride_b, stride_c, stride_r;
  const void* bias;          /* [N] (geglu: [2N]); [M] if bias_per_row */
  int bias_per_row;
  const void* bias2;         /* [M / bias2_rows_per_group, N], e.g. per-frame time embedding */
  int bias2_rows_per_group;
  int64_t bias2_ld;          /* row pitch of bias2 in elements, 0 = N (a column slice of a wider buffer) */
  const float* rowscale;     /* [M] fp32, e.g. the audio attention masks */
  const void* residual;      /* [M,N] with leading dimension ldr (may alias C) */
  int64_t ldr;
  float alpha;
  int act;
  int geglu;
  int out_f32;               /* write C as fp32 instead of dtype */
  int dtype;
  void* workspace;           /* optional fp32 scratch for split-K / stream-K partial sums (small and mid-size grids); may be null.
                              * The caller ZERO-INITIALISES it once (hipMemset) and gives it to one stream at a time: its last
                              * 64 KB hold the arrival counters of the stream-K kernel (csrc/gemm4.hip), zero between launches;
                              * >= 42 MB for that kernel's stream-K tail (it is skipped with less) */
  int64_t workspace_bytes;
  /* ABI v2: output columns n < lead_cols (multiple of 8; 0 = none) are multiplied by lead_alpha on top of alpha /
   * rowscale, before the residual add.  Used for the q part of Attention.to_q / fused q|k|v projections: q leaves the
   * GEMM already scaled by head_dim^-0.5 * log2(e) (one rounding, like the reference's own rounding of q), and
   * hallo_attention(q_prescaled = 1) exponentiates raw scores. */
  int lead_cols;
  float lead_alpha;
  /* ABI v2: fused LayerNorm (nn.LayerNorm in front of to_q|to_k|to_v, to_q, GEGLU: hallo/models/attention.py:563-601,
   * 784-905; motion_module.py:387-423).  When ln_colsum != NULL, A is the UN-normalised activation, W must be
   * gamma-scaled (W[n,k] * gamma[k], rounded once), bias must include beta . W^T, and ln_colsum[n] = sum_k W[n,k] of
   * the scaled, rounded W in fp32 ([N]; geglu: [2N]).  The kernel reduces every row of A to mean / rstd over K while it
   * feeds the MFMAs and writes act(alpha * ... (rstd * (acc - mean * ln_colsum[n]) + bias + bias2) ...): LayerNorm costs
   * no pass over HBM.  batch = 1, dtype output, no bias_per_row. */
  const float* ln_colsum;
  float ln_eps;
  const float* ln_stats;     /* optional [M][2] fp32 (mean, rstd) from hallo_row_stats; NULL: computed inside the K loop */
  /* ABI v7 (round 5): LayerNorm statistics from the PRODUCER of a tensor instead of a pass over it (hallo_row_stats).
   * row_parts (out, optional): fp32 [M][ceil(N / 64)][2] -- (sum, sum of squares) of every output row over each 64-column block,
   *   taken from the ROUNDED values written to C (what a later nn.LayerNorm over C's rows would read).  The 128 x 128 kernel
   *   emits them from its epilogue registers; every other routing target fills the same layout with one extra pass over C
   *   (the cost of hallo_row_stats), so the contract holds for any problem.  batch = 1, dtype output, no geglu.
   * ln_parts (in, with ln_colsum): when > 0, ln_stats is such a buffer -- [M][ln_parts][2] partial sums over the K columns of A
   *   (ln_parts = ceil(K / 64) for a hallo_gemm producer) -- and the kernel reduces them to mean / rstd (eps = ln_eps) for its
   *   rows in its prologue, in slot order.  0: ln_stats is [M][2] (mean, rstd) as before (hallo_row_stats, hallo_face_xattn_stats).
   *   Round 6: ln_parts > 0 must equal ceil(K / 64) and be <= 32 (a tile's rows x ln_parts pairs are staged in the operand LDS), else -22. */
  float* row_parts;
  int ln_parts;
  /* ABI v7: 1 = the caller guarantees that the last 64 KB of `workspace` were zero before the first launch that used this workspace
   * and have been written by nothing but hallo_gemm since (the stream-K kernel's arrival counters; every launch restores them).
   * 0 (what a v6-style caller that hands over uninitialised scratch gets): K-split tails are not used -- whole tiles only. */
  int workspace_zeroed;
  /* ABI v9 (round 6): HEAD-MAJOR K / V straight from the fused to_q|to_k|to_v projection of the 320-channel level (8 heads x 40:
   * hallo/models/mutual_self_attention.py:253-284, attention.py:828-831).  With kv_out != NULL the output columns n >= kv_col0 are
   * not written to C but to kv_out: column kv_col0 + t * 320 + h * 40 + d of row m (t = 0: K, 1: V) goes to
   *   kv_out + t * kv_tensor_stride + (((m / kv_rows_per_image) * 8 + h) * kv_rows_per_image + m % kv_rows_per_image) * 40 + d
   * i.e. [image][head][row][40] tensors, the layout hallo_attention reads with kv1_hs = kv_rows_per_image * 40 and a row stride of
   * 40: a 64-key tile of a head is then ONE contiguous 5 KB piece instead of 64 80-byte pieces at a 1920-byte pitch (8-12 % of the
   * attention kernel's time, profiles/r6_attn40_headmajor.txt).  C keeps the columns below kv_col0 (ldc >= kv_col0).  Only the
   * row-stationary K = 320 kernel has this epilogue: ask hallo_gemm_kv_split_ok() first; any other routing returns -22. */
  void* kv_out;
  int kv_col0;
  int kv_rows_per_image;
  int64_t kv_tensor_stride;
} hallo_gemm_desc;
int hallo_gemm(const hallo_gemm_desc* d, void* stream);
/* 1 when hallo_gemm would run an [M, N] x K problem with fused LayerNorm and lead_cols = kv_col0 on the kernel that implements
 * kv_out (N - kv_col0 must be 640: K and V of 8 heads x 40), under the current routing options. */
int hallo_gemm_kv_split_ok(int M, int N, int K, int kv_col0);

/* ------------------------------------------------------------------------------------------
 * hallo_conv3x3_nhwc: implicit-GEMM 3x3 convolution on token-major activations.
 *   y[n,oy,ox,:] = act( alpha * (sum_{ky,kx,c} x[n, oy*stride+ky-pad_t, ox*stride+kx-pad_l, c]
 *                                 * w[:,ky,kx,c] + bias + bias2[n-group]) + residual )
 * w is [Cout, 3, 3, Cin] (the torch [Cout,Cin,3,3] weight permuted once at load time).
 * upsample=1 folds a nearest 2x upsample of x into the gather (Upsample3D, resnet.py:166-183).
 * Replaces InflatedConv3d.forward (hallo/models/resnet.py:50-66) at conv_in/conv_out
 * (unet_3d.py:603,710), ResnetBlock3D.conv1/conv2 (resnet.py:388,405), Downsample3D
 * (resnet.py:250), FaceLocator (face_locator.py:94-113) and the AutoencoderKL convs.
 * Cin must be a multiple of 8 (pad the channel axis with zeros otherwise).
 */
typedef struct hallo_conv_desc {
  const void* x; const void* w; void* y;
  int n_img, H, W, Cin, Cout, OH, OW;
  int stride, pad_t, pad_l, upsample;
  const void* bias;          /* [Cout] or null */
  const void* bias2;         /* [n_groups, Cout] or null (time embedding per batch entry) */
  int bias2_rows_per_group;  /* output rows (pixels) per bias2 row */
  int64_t bias2_ld;          /* row pitch of bias2 in elements, 0 = Cout */
  const void* residual;      /* [n_img*OH*OW, Cout] (ld = ldr) or null */
  int64_t ldr;
  int64_t ldy;               /* leading dimension of y, 0 = Cout */
  float alpha;
  int act;
  int dtype;
  void* workspace;           /* optional split-K scratch, as in hallo_gemm_desc */
  int64_t workspace_bytes;
} hallo_conv_desc;
int hallo_conv3x3_nhwc(const hallo_conv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_attention: flash-style softmax(Q K^T * scale) V with up to two key/value segments.
 * Replaces diffusers AttnProcessor2_0 -> F.scaled_dot_product_attention at:
 *   - the reference-augmented spatial self-attention (K/V = [self ; ReferenceNet bank],
 *     hallo/models/mutual_self_attention.py:253-263) including the CFG rule that the
 *     unconditional batch entries attend to themselves only (:264-284): segment 2 is skipped
 *     for batch index < kv2_first_batch; the bank entry of batch row b is (b / kv2_batch_div) %
 *     kv2_batch_mod, which covers the reference's tiled frame->bank mapping (:235-247, bank row
 *     n % 2 under CFG) as div=1, mod=2;
 *   - the audio block's spatial self-attention (hallo/models/attention.py:828-831);
 *   - the face-token cross-attention (mutual_self_attention.py:296-303) and the three
 *     hierarchical audio cross-attentions (attention.py:846-884) with Lkv = 4 / 32.
 * Layouts (element strides): q[b, i, h*hd + d] at q + b*q_bs + i*q_rs + h*hd + d; same for
 * k1/v1 (Lkv1 rows) and k2/v2 (Lkv2 rows; batch index b / kv2_batch_div), o (row stride o_rs).
 * head_dim in {40, 80, 160}.  heads*head_dim contiguous per row.
 */
typedef struct hallo_attn_desc {
  const void* q; const void* k1; const void* v1; const void* k2; const void* v2; void* o;
  int batch, heads, head_dim, Lq, Lkv1, Lkv2;
  int64_t q_bs, q_rs, k1_bs, k1_rs, v1_bs, v1_rs, k2_bs, k2_rs, v2_bs, v2_rs, o_bs, o_rs;
  int kv2_batch_div;    /* segment-2 batch index = (b / kv2_batch_div) % kv2_batch_mod */
  int kv2_batch_mod;    /* <= 0: no modulo */
  int kv2_first_batch;  /* batches below this index skip segment 2 (CFG uncond half) */
  float scale;          /* head_dim^-0.5 */
  int dtype;
  /* Optional fp32 output row scale (ABI v2): o[b, q, head h] *= o_rowscale[(h / o_rowscale_head_div) * o_rowscale_stride
   * + b * Lq + q].  Carries `motion_scale[i] * mask_i[level]` of the hierarchical audio cross-attention
   * (hallo/models/attention.py:853-903) so that the three branches run as ONE launch over 3 x heads heads
   * (head_div = heads per branch, stride = batch * Lq); null = no scaling. */
  const float* o_rowscale;
  int o_rowscale_head_div;      /* <= 0: all heads share one scale vector */
  int64_t o_rowscale_stride;
  int q_prescaled;              /* 1: q already carries scale * log2(e) (hallo_gemm lead_alpha); `scale` is ignored */
  /* ABI v9: head strides of segment 1 / 2 in elements; 0 = head_dim (the heads of a row are adjacent, every layout above).
   * kvN_hs = LkvN * head_dim with kN_rs = vN_rs = head_dim is the head-major layout hallo_gemm's kv_out writes.  head_dim 40 with
   * q_prescaled only (the LDS-DMA kernel); anything else with a non-zero stride returns -22. */
  int64_t kv1_hs, kv2_hs;
} hallo_attn_desc;
int hallo_attention(const hallo_attn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_temporal_attention: per-pixel attention over the frame axis (F' <= 32 tokens).
 * Replaces VersatileAttention.forward (hallo/models/motion_module.py:553-609): the
 * "(b f) d c -> (b d) f c" rearranges, AttnProcessor SDPA over f and the inverse rearrange.
 * qkv: [B*F', HW, 3*C] rows of [q | k | v] (the fused projection of norm(x)+PE), out: [B*F', HW, C].
 */
int hallo_temporal_attention(const void* qkv, void* out, int B, int F, int HW, int C, int heads,
                             float scale, int dtype, void* stream);
/* ABI v8 (round 6): the same attention over F' = lead + F_clip temporal positions per batch entry, for tensors whose frame rows are
 * stored in TWO segments: the `lead` leading positions of ALL batch entries first (entry b's at frame rows [b * lead, (b + 1) * lead)),
 * then the remaining F' - lead positions of entry b at frame rows B * lead + b * (F' - lead) + ...  The leading positions are the
 * ReferenceNet features of the motion frames that the reference concatenates in time in front of every clip before a motion
 * module and slices off after it (hallo/models/unet_3d_blocks.py:696-748, 1148-1202): with them at the front of the buffer the
 * clip rows of a whole batch are ONE contiguous [B * F_clip, HW, C] block -- the audio module's output projection writes
 * straight into it and the row-wise tail of the motion module runs on it without a gather, for any batch size (CFG pairs,
 * batches of independent clips).  Key order per pixel is unchanged (leading positions first), so results equal the interleaved
 * layout's bit for bit.  lead = 0 is hallo_temporal_attention. */
int hallo_temporal_attention_lead(const void* qkv, void* out, int B, int F, int lead, int HW, int C, int heads,
                                  float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_groupnorm_nhwc: per-frame GroupNorm (+ optional SiLU) on [n_img, HW, C].
 * Replaces InflatedGroupNorm (hallo/models/resnet.py:88-101) + SiLU in ResnetBlock3D
 * (resnet.py:385-386,396,402), conv_norm_out (unet_3d.py:708-709), Transformer3DModel.norm
 * (transformer_3d.py:197, eps 1e-6), TemporalTransformer3DModel.norm (motion_module.py:290)
 * and the AutoencoderKL norms.  workspace: fp32 [n_img * chunks * groups * 2] with
 * chunks = hallo_groupnorm_chunks(HW).
 */
int hallo_groupnorm_chunks(int HW);
int hallo_groupnorm_nhwc(const void* x, void* y, const void* gamma, const void* beta, float* workspace,
                         int n_img, int HW, int C, int groups, float eps, int silu, int dtype, void* stream);
/* ABI v8 (round 6): the same GroupNorm over a channel CONCATENATION read in place: the C channels of a row are the C1 channels of
 * x [n_img, HW, C1] followed by the C - C1 channels of x2 [n_img, HW, C - C1]; y is [n_img, HW, C].  Replaces
 * `torch.cat([hidden_states, res_hidden_states], dim=1)` + norm1 of the up-block resnets (hallo/models/unet_3d_blocks.py:1131,1373;
 * resnet.py:385): the concatenated tensor is never written.  Bit-identical to hallo_groupnorm_nhwc on the materialised concatenation
 * (same per-column sums, same reduction order).  C1 % 8 == 0; C1 == C (x2 ignored) is hallo_groupnorm_nhwc. */
int hallo_groupnorm_nhwc2(const void* x, int C1, const void* x2, void* y, const void* gamma, const void* beta, float* workspace,
                          int n_img, int HW, int C, int groups, float eps, int silu, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_layernorm: row LayerNorm (eps 1e-5 default in torch) with optional positional
 * encoding add: y[r,:] = LN(x[r,:]) * gamma + beta + pe[(r / pe_rows_per_pos) % pe_len, :].
 * Replaces nn.LayerNorm at hallo/models/attention.py:563,586,601,812,835,905 and
 * motion_module.py:408,420, fused with PositionalEncoding.forward (motion_module.py:459-461).
 * pe is fp32 [pe_len, C] or null.
 */
int hallo_layernorm(const void* x, void* y, const void* gamma, const void* beta, const float* pe,
                    int rows, int C, float eps, int pe_rows_per_pos, int pe_len, int dtype, void* stream);

/* hallo_softmax_rows: y[r,:] = softmax(scale * x[r,:]), x fp32 [rows, cols], y dtype.
 * Used by the AutoencoderKL mid-block attention (1 head, head_dim 512). */
int hallo_softmax_rows(const float* x, void* y, int rows, int cols, float scale, int dtype, void* stream);

/* hallo_copy2d: dst[r, 0:width] = src[r, 0:width] for r < rows (element pitches).  Used for the skip
 * concatenation (unet_3d_blocks.py:1131,1373) and the motion-frame concat (unet_3d_blocks.py:740). */
int hallo_copy2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t rows, int width,
                 int dtype, void* stream);

/* hallo_nchw_to_nhwc: x [n, C, HW] (fp32 if src_f32 else dtype) -> y [n, HW, Cpad] dtype, channels
 * >= C zero-filled.  hallo_nhwc_to_nchw_f32: x [n, HW, ldx] dtype -> y fp32 [n, C, HW] with
 * y = clamp(x*mul + add, lo, hi) (decode_latents post-processing, face_animate.py:243). */
int hallo_nchw_to_nhwc(const void* x, void* y, int n, int C, int HW, int Cpad, int src_f32, int dtype, void* stream);
int hallo_nhwc_to_nchw_f32(const void* x, float* y, int n, int C, int HW, int64_t ldx, float mul, float add,
                           float lo, float hi, int dtype, void* stream);

/* hallo_timestep_embedding: diffusers Timesteps(dim, flip_sin_to_cos=True, shift=0)
 * (hallo/models/unet_3d.py:184-185,582): out[b, :] = [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(1e4)*i/half). */
int hallo_timestep_embedding(const float* t, void* out, int batch, int dim, int dtype, void* stream);

/* hallo_cfg_ddim_step: classifier-free guidance combine + DDIM (eta = 0) update
 * (hallo/animate/face_animate.py:415-420; diffusers DDIMScheduler.step).  `cfg` is a flag word:
 *   HALLO_DDIM_CFG (1) guidance, HALLO_DDIM_PRED_EPSILON (2) / HALLO_DDIM_PRED_SAMPLE (4) prediction_type
 *   (neither: v_prediction, what configs/inference/default.yaml:82 sets), HALLO_DDIM_CLIP_SAMPLE (8) clip x0 to [-1, 1].
 *   v      = guidance ? uncond + gs * (cond - uncond) : model_out
 *   v-pred:  x0 = sqrt(a_t) * x - sqrt(1 - a_t) * v ; eps = sqrt(a_t) * v + sqrt(1 - a_t) * x
 *   epsilon: eps = v ; x0 = (x - sqrt(1 - a_t) * v) / sqrt(a_t)      sample: x0 = v ; eps = (x - sqrt(a_t) * v) / sqrt(1 - a_t)
 *   clip_sample: x0 = clamp(x0, -1, 1) (eps is not recomputed: use_clipped_model_output = False)
 *   x_prev = sqrt(a_prev) * x0 + sqrt(1 - a_prev) * eps
 * model_out: dtype [B*F, HW, ldm] token-major (cond batch follows uncond batch when cfg=1),
 * latents: fp32 [F, HW, C] token-major, updated in place; next_in: dtype [F, HW, ldn] (the next
 * UNet input, channels >= C zero) written for both CFG halves by the caller's layout. */
#define HALLO_DDIM_CFG 1
#define HALLO_DDIM_PRED_EPSILON 2
#define HALLO_DDIM_PRED_SAMPLE 4
#define HALLO_DDIM_CLIP_SAMPLE 8
int hallo_cfg_ddim_step(const void* model_out, int64_t ldm, float* latents, void* next_in, int64_t ldn,
                        int rows, int C, int cfg, float guidance_scale, float alpha_t, float alpha_prev,
                        int dtype, void* stream);

/* hallo_row_stats: per-row LayerNorm statistics stats[r] = (mean, 1/sqrt(var + eps)) of x [rows, C] (two-pass, fp32),
 * the only pass over x that nn.LayerNorm still costs when its affine is folded into the consuming hallo_gemm
 * (ln_colsum / ln_stats).  C % 8 == 0, C <= 1536. */
int hallo_row_stats(const void* x, float* stats, int64_t rows, int C, float eps, int dtype, void* stream);

/* hallo_gemm_fuses_row_stats (ABI v4): 1 when hallo_gemm, called with ln_colsum set and ln_stats = NULL on a problem of
 * this shape (contiguous A / W, dtype output, bias2 with `bias2_rows_per_group` rows per group or 0, `lead_cols`), takes the
 * row-stationary kernel (csrc/gemm_rs.hip: K = 320 / 640, the A rows of a workgroup held in registers) that derives the
 * LayerNorm statistics from its resident A rows -- the caller can then skip hallo_row_stats.  0: pass ln_stats (or accept
 * the in-K-loop statistics of the tiled kernel).  Pure function of its arguments and of hallo_set_option("gemm_rs"). */
int hallo_gemm_fuses_row_stats(int M, int N, int K, int geglu, int bias2_rows_per_group, int lead_cols);

/* hallo_gemm4_schedule (ABI v6, round 4): the work deal csrc/gemm4.hip (exact-fit kernel: 128 x 160 tiles, one persistent workgroup per
 * CU) would use for a plain M x N x K hallo_gemm problem -- sched[0..7] = tiles_m, tiles_n, K steps of 64, workgroups G,
 * data-parallel rounds dp (workgroup w owns tiles j * G + w, j < dp, XCD-remapped), tail tiles R = tiles - dp * G, parts per tail
 * tile (1: whole tiles; > 1: each tail tile's K loop dealt over that many workgroups, partial tiles reduced in K order by the
 * last arriver), K steps per part.  Pure function of its arguments and of the device's CU count (256 without a device), so the
 * deal can be checked on a host without a GPU (tests/test_host_cpu.py).  Returns 1 / 0 (the kernel does not cover the problem:
 * K % 64, too little workspace for a split tail) / -22; `force_parts` is reserved (pass 0).  Which problems hallo_gemm actually
 * gives to the kernel is the routing rule behind hallo_set_option("gemm4", ...). */
int hallo_gemm4_schedule(int M, int N, int K, int64_t workspace_bytes, int force_parts, int* sched);

/* ------------------------------------------------------------------------------------------
 * ABI v4: fp8 (OCP e4m3) projections -- BASELINE.json configs[4] "fp8 MFMA QKV/out projections with bf16 accumulate":
 * the diffusers Attention.to_q / to_k / to_v / to_out Linears (hallo/models/mutual_self_attention.py:253-303,
 * hallo/models/attention.py:828-884, hallo/models/motion_module.py:553-609) with both operands quantised.
 *
 * hallo_quant_rows_fp8: x [rows, C] (row stride ldx, dtype) -> q [rows, C] e4m3 bytes (contiguous) and scale[rows] fp32 with
 *   x[r, c] ~= scale[r] * decode(q[r, c]),  scale[r] = max_c |x[r, c]| / 448 (1 for an all-zero row).
 *   gamma / beta non-NULL: y = LayerNorm(x; gamma, beta, eps) rounded to dtype is quantised instead (the norm in front of
 *   to_q|k|v), x is read once.  C % 8 == 0, C <= 1536.  Weights are quantised once with the same call (one scale per row of
 *   W = per output channel).
 * hallo_gemm_fp8: C[m, n] = alpha * lead(n) * (a_scale[m] * w_scale[n] * sum_k decode(A[m,k]) decode(B[n,k]) + bias[n]) +
 *   residual[m, n], fp32 accumulation on v_mfma_f32_32x32x16_fp8_fp8, output in `dtype` (fp16 / bf16);
 *   lead(n) = lead_alpha for n < lead_cols else 1 (the q columns of a fused q|k|v projection carry the softmax scale).
 *   K % 16 == 0, N % 8 == 0, lda / ldb multiples of 16 bytes, A / B / C 16-byte aligned.
 *   Round 6: the contraction is v_mfma_scale_f32_32x32x64_f8f6f4 with unit (E8M0 = 127) block scales -- the fp8-rate instruction; the per-row /
 *   per-channel scales above are applied to the fp32 accumulators as before (hallo_set_option("fp8_mx", 0): the non-scaled 32x32x16 form). */
int hallo_quant_rows_fp8(const void* x, int64_t ldx, void* q, float* scale, int64_t rows, int C, const void* gamma,
                         const void* beta, float eps, int dtype, void* stream);
typedef struct {
  const void* A; const void* B; void* C;     /* A [M, K] e4m3, B [N, K] e4m3, C [M, N] dtype */
  int M, N, K;
  int64_t lda, ldb, ldc;                     /* lda / ldb in bytes (= elements), ldc in elements */
  const float* a_scale;                      /* [M] */
  const float* w_scale;                      /* [N] */
  const void* bias;                          /* [N] dtype or NULL */
  const void* residual; int64_t ldr;         /* [M, N] dtype or NULL */
  float alpha;
  int lead_cols; float lead_alpha;
  int dtype;
} hallo_gemm_fp8_desc;
int hallo_gemm_fp8(const hallo_gemm_fp8_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * ABI v5: hallo_ff320 -- diffusers FeedForward(activation_fn="geglu") of a 320-wide transformer block with its LayerNorm and
 * residual, `ff(norm3(x)) + x` (hallo/models/attention.py:601,905; hallo/models/motion_module.py:420), as ONE kernel:
 *   y[m, :] = res[m, :] + b2 + W2 . GEGLU(W1' . LN(x[m, :]) + b1'),   GEGLU(v | g) = v * gelu_erf(g)
 * with the LayerNorm affine folded by the caller (W1' = W1 * gamma, b1' = b1 + W1 . beta; layernorm = 0: x is used as is).
 * The 1280-wide intermediate never reaches memory (csrc/gemm_ff.hip).  x / res / y: [M, 320] dtype with row strides ldx /
 * ldr / ldy (elements; ldx, ldy % 8 == 0, ldr % 4 == 0; x, y 16-byte aligned); y may alias x and res.  b2: [320] dtype.
 * wpack: hallo_ff320_pack_bytes() bytes, 80 images of 32 KB, image s = the weights of intermediate columns c0 = 16 s .. +15:
 *   [0, 20480)      W1' rows: 5 sub-tiles t of [32 rows][64 k]: row r < 16 = value row c0 + r, r >= 16 = gate row 1280 + c0 +
 *                   r - 16; the 16-byte piece pc (k = 64 t + 8 pc .. +7) of row r at byte 4096 t + 128 r + 16 (pc ^ ((r >> 1) & 7))
 *   [20480, 30720)  W2: row n (0..319) at byte 32 n; its two 16-byte halves h hold the 8 k-slots e <-> column
 *                   c0 + (e & 3) + 8 (e >> 2) + 4 h (the accumulator layout of the first MFMA), half h at 16 (h ^ ((n >> 3) & 1))
 *   [30720, 30848)  fp32 [2 halves h][16]: b1'[c] * 1.1774100 for the 8 columns of half h, then b1'[1280 + c] * 0.8493218
 *                   (the gelu_u scales, csrc/common.h); the rest of the image is padding
 * (hallo_amd/ops.py ff320_pack builds it).  hallo_set_option("ff_fused", 1) makes hallo_amd's FeedForward call it; the default is
 * 0 -- on MI355X the kernel measures 219 us against 193 us for the two hallo_gemm launches at 65536 rows (csrc/gemm_ff.hip). */
int64_t hallo_ff320_pack_bytes(void);
int hallo_ff320(const void* x, int64_t ldx, const void* res, int64_t ldr, void* y, int64_t ldy, const void* wpack,
                const void* b2, int64_t M, int layernorm, float ln_eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_face_xattn: y = x + to_out(SDPA(to_q(LayerNorm(x)), K_face, V_face)) for a cross-attention over H*T = 32
 * (head, token) pairs -- norm2 + attn2 + residual of the spatial transformer block
 * (hallo/models/mutual_self_attention.py:286-303; 4 face tokens x 8 heads) in ONE pass over x.
 * The per-clip constants fold the projections and the LayerNorm affine (fp32 host math, then rounded once):
 *   sg  [nb][32][C]  dtype  gamma_c * c0 * sum_d Wq[h*hd+d, c] K[b, t, h*hd+d],  (h,t) = h*T + t, c0 = hd^-0.5 * log2(e)
 *   g   [nb][32]     fp32   sum_c sg[c]            b [nb][32] fp32   sum_c beta_c * (sg[c] / gamma_c)
 *   owp [nb][C][32]  dtype  sum_d Wo[c, h*hd+d] V[b, t, h*hd+d] stored as [c][ks2][hi][e] with
 *                           (h,t) = 8*(2*ks2 + (e >> 2)) + 4*hi + (e & 3)     (the MFMA k-slot order of the kernel)
 *   bo  [C] dtype           to_out bias
 * x, y: [rows, C] (y may alias x); batch entry of a row = row / rows_per_batch (a multiple of 32); C % 32 == 0.
 */
int hallo_face_xattn(const void* x, void* y, const void* sg, const float* g, const float* b, const void* owp,
                     const void* bo, int64_t rows, int C, int64_t rows_per_batch, float eps, int dtype, void* stream);
/* ABI v7: the same, and stats_out (optional) [rows][2] fp32 = (mean, rstd) over C of every OUTPUT row as written (rounded), with
 * LayerNorm eps = stats_eps: what hallo_row_stats(y) would give, from the kernel's own epilogue -- the statistics of norm3 in front
 * of the block's feed-forward (hallo/models/attention.py:586-601) without a pass over y.  (The kernel already holds whole rows.) */
int hallo_face_xattn_stats(const void* x, void* y, const void* sg, const float* g, const float* b, const void* owp,
                           const void* bo, int64_t rows, int C, int64_t rows_per_batch, float eps, float* stats_out,
                           float stats_eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * hallo_frames_to_uint8: decoded frames, planar fp32 [frames, channels, hw] in [0, 1], to interleaved uint8
 * [frames, hw, channels] = np.clip(x * 255, 0, 255).astype(np.uint8) of tensor_to_video
 * (hallo/utils/util.py:308-312).  Byte-exact with the numpy expression on the same fp32 input; done on the device so
 * that the D2H copy / the 8-GPU all-gather of a clip moves 4x fewer bytes.
 */
int hallo_frames_to_uint8(const float* x, uint8_t* y, int frames, int channels, int64_t hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * ABI v3: wav2vec2 audio front-end: Wav2VecModel.forward (hallo/models/wav2vec.py:42-109) =
 * transformers Wav2Vec2FeatureEncoder -> linear_interpolation to the video frame rate (wav2vec.py:196-209) ->
 * Wav2Vec2FeatureProjection -> Wav2Vec2Encoder with all 12 hidden states kept (audio_processor.py:105-129).
 *
 * hallo_w2v_conv0_gn_gelu: the first feature-encoder layer, Conv1d(1 -> C, k, stride, bias = False) ->
 *   GroupNorm(C groups: per-channel statistics over time, biased variance) -> erf GELU, on a normalised fp32 waveform
 *   wave[n_samples].  y is the token-major activation [L0, C] in `dtype`, L0 = (n_samples - k) / stride + 1.
 *   w is fp32 [C][k] (conv.weight[:, 0, :]), gamma / beta fp32 [C].  The conv is recomputed, not stored: pass 1 reduces
 *   the statistics (deterministic two-level reduction, combined in fp64), pass 2 normalises and writes y once.
 *   workspace: hallo_w2v_conv0_workspace(n_samples, C, k, stride) bytes of fp32 scratch.
 *   C % 8 == 0, k <= 16, stride <= 8.  The remaining feature-encoder layers are hallo_gemm calls over overlapping row
 *   windows of y (lda = stride * C, K = k * C) with act = HALLO_ACT_GELU.
 * hallo_lerp_rows: y[t, :] = lerp of x rows at src = t * (in_rows - 1) / (out_rows - 1), i.e.
 *   F.interpolate(mode = "linear", align_corners = True) along time with ATen's fp32 index arithmetic. C % 8 == 0.
 */
int64_t hallo_w2v_conv0_workspace(int64_t n_samples, int C, int k, int stride);
int hallo_w2v_conv0_gn_gelu(const float* wave, int64_t n_samples, const float* w, const float* gamma, const float* beta,
                            void* y, float* workspace, int C, int k, int stride, float eps, int dtype, void* stream);
int hallo_lerp_rows(const void* x, void* y, int in_rows, int out_rows, int C, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HALLO_AMD_H */
