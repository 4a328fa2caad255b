/*
 * lcm_hip.h -- C ABI of the MI355X (gfx950) LCM Stable-Diffusion-1.5 hot path.
 *
 * The reference has no FFI of its own: its hot path is one Python call,
 *   self.pipe(prompt=..., width=..., height=..., num_inference_steps=..., guidance_scale=..., generator=...)
 * at backends/cuda_worker.py:221-229 (diffusers StableDiffusionPipeline + LCMScheduler), behind the
 * PipelineWorker Protocol of backends/base.py:29-39.  This header is the boundary a maintainer binds
 * (ctypes stub in INTEGRATION.md) to replace the arithmetic under that call.  Each entry point names the
 * operator of the reference's op graph it replaces (diffusers module, reached from the call site above;
 * numpy twin of the glue in backends/rknnlcm.py where one exists).
 *
 * Conventions
 *   - all device pointers; activations are fp16 "pixel-major" (NHWC == [B*H*W, C] row major) unless stated;
 *     latents / noise / eps-state are fp32 NCHW (the request-side layout of backends/rknnlcm.py:424).
 *   - weights are fp16, row = output channel, k-contiguous; 3x3 weights are [Cout][ky][kx][Cin].
 *   - every function enqueues on `stream` (hipStream_t passed as void*), never synchronises, and
 *     returns 0 on success or a negative LCM_E* / positive hipError_t code; lcm_last_error() gives text.
 *   - nothing here allocates: callers own all buffers (graph-capture safe).
 */
#ifndef LCM_HIP_H
#define LCM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LCM_OK 0
#define LCM_EINVAL (-1)   /* shape / alignment precondition violated */
#define LCM_ENODEV (-2)   /* no gfx950 device */

#define LCM_EPI_NONE 0
#define LCM_EPI_GEGLU 1   /* out[m][j] = x*gelu(g); weight rows interleaved x/g in blocks of 16 */
#define LCM_EPI_QUICK_GELU 2  /* x * sigmoid(1.702 x) after bias (CLIPMLP, hidden_act quick_gelu) */
#define LCM_EPI_GELU 3        /* exact gelu after bias */

const char* lcm_last_error(void);
int lcm_version(void);
/* device_count / arch string of device `dev` (buf >= 64 bytes) */
int lcm_device_info(int dev, char* arch_buf, int buf_len, int* cu_count, uint64_t* hbm_bytes);

/* ---- dense contraction (torch.nn.Linear / 1x1 Conv2d inside UNet2DConditionModel, AutoencoderKL) ----
 * out[z][m][n] = out_scale * sum_k A[z][m][k] * W[z][n][k]  (+bias[n]) (+rowadd[m / rows_per_batch][n]) (+res[m][n])
 * A may be split along k over two sources (fused torch.cat of the up-block skip: k < K1 from A, else A2).
 * Preconditions: K % 64 == 0, K1 % 64 == 0, N % 64 == 0 (N % 32 for GEGLU pairs), 16-byte aligned rows.
 */
int lcm_gemm_f16(const void* A, int lda, const void* A2, int lda2, int K1,
                 const void* W, const void* bias, const void* rowadd, int ld_rowadd, int rows_per_batch,
                 const void* res, int ldr, void* out, int ldo,
                 int M, int N, int K, int epilogue, float out_scale,
                 int batch, int64_t strideA, int64_t strideW, int64_t strideO, int img_rows,
                 void* stats_out, int64_t stats_bytes, int* slabs_per_image, void* stream);
/* img_rows = output rows PER IMAGE when the M rows stack several independent requests (M % img_rows == 0; 0: M is
 * one image).  It keys everything that decides the fp32 summation order (see "Determinism" below).
 *
 * Fused GroupNorm statistics (all three contraction entry points): stats_out != NULL asks the epilogue to also
 * write the per-channel (sum, sum of squares) of the fp16 values it stores, per CANONICAL 32-pixel slab of the
 * output (32 consecutive rows of a GEMM; a 2x16 -- 4x8 for images up to 16 wide -- pixel patch of a 3x3 convolution;
 * a split-K reduce slab): fp32 [B * slabs_per_image][N][2].  Slabs are a property of the output tensor, never of
 * the tile shape.  *slabs_per_image returns how many slabs each image got; 0 means the launch could not produce them
 * (img_rows % 32 != 0, GEGLU/batched launch, N > 2048): use lcm_groupnorm_f16 on the output instead.
 * Consumer: lcm_groupnorm_from_stats_f16.  stats_bytes = size of the buffer behind stats_out: a launch whose slabs would not
 * fit is refused with LCM_EINVAL BEFORE anything is enqueued (round 2 wrote past an undersized buffer: a GPU memory fault on
 * odd image sizes).  lcm_stats_bytes(M, N, img_rows) returns a size that always suffices. */
int64_t lcm_stats_bytes(int M, int N, int img_rows);

/* fp32 scratch for deterministic split-K (deep-K, small-M layers).  The caller owns the memory; it is
 * registered per current device and must outlive every later launch (graph replays included).  With no workspace
 * registered the library never splits (a deployment either registers one, always, or never: the pipeline always
 * does).  With one registered, a launch whose canonical K partition does not fit FAILS (LCM_EINVAL) -- it is never
 * silently run with fewer parts, because that would change the numbers.  bytes >= 4*splits*M*N of the largest split
 * layer (64 MiB covers SD1.5 at batch 1, 384 MiB batch 8). */
int lcm_set_workspace(void* ptr, int64_t bytes);
/* a workspace of its own for the launches of one stream: two sampler passes in flight on two streams ("lanes") must not share
 * split-K slabs.  Looked up before the device-wide workspace.  An entry is OWNED by the pointer that registered it:
 * (ptr, bytes > 0) registers -- LCM_EINVAL if the stream already carries another owner's workspace; (ptr, 0) forgets the entry
 * only if it still holds ptr; (NULL, 0) forgets it whoever owns it. */
int lcm_set_stream_workspace(void* stream, void* ptr, int64_t bytes);
/* a hipStream_t of the caller's own (hipStreamNonBlocking).  Frameworks hand streams out of small recycled pools (torch: 32 per
 * device), so two long-lived owners can end up keyed on one handle; a lane of the sampler takes its streams from here. */
int lcm_stream_create(void** stream_out);
int lcm_stream_destroy(void* stream);

/* launch heuristics of the contraction kernels (0 keeps a value): workgroups a split-K launch aims for, the
 * maximum number of K splits, and the workgroup count below which a larger tile is passed over.  These feed the
 * canonical-partition heuristic: changing them changes fp32 summation order for shapes without a plan entry. */
int lcm_set_tuning(int target_wgs, int max_splits, int min_wgs);
/* where the canonical K partition may have more than one part: images of at most `max_rows_per_image` output rows
 * (default 4096), at most `max_parts` parts (default 8).  A deployment-wide constant: it is part of what decides the
 * fp32 summation order. */
int lcm_set_split_policy(int max_rows_per_image, int max_parts);
/* How a launch whose canonical partition has several parts runs: split over workgroups + reduce launch (fp32 slabs), or
 * SEGMENTED in one workgroup (the parts accumulated separately and added in part order, in registers).  Same bits either
 * way; 0 (default) = segmented when the unsplit launch already fills the chip (batched requests), 1 = always segmented,
 * 2 = always split. */
int lcm_set_seg_mode(int mode);

/* Determinism.  The fp32 summation order of every output element is fixed by the K partition of its layer, and
 * that partition is a function of the PER-IMAGE problem only: (kind, rows per image, N, K[, output width]) ->
 * splits, from the plan entry of that per-image shape if one was set, else from a fixed heuristic.  Tile shape,
 * pipeline depth and kernel variant (the plan entry of the TOTAL shape, or the occupancy heuristic) never change a
 * bit of the result, and the fused statistics are slab-canonical.  Hence: same request + seed => same bytes, alone
 * or inside a batch of any size, in any process, under any tuning of tile / variant (the reference contract
 * tests/test_sdxl_worker.py:171-198, extended to micro-batches).
 *
 * Per-shape launch plans, normally loaded from the table shipped with the package (tuned offline): kind 0 =
 * lcm_gemm_f16 (aux = batch), 1 = row-gather conv (aux = 1), 2 = LDS-halo conv (aux = (W_out << 1) | has_gn).
 * bm in {64,128}, bn in {64,128,160}; variant as below (-1 = auto).  `splits` is honoured only through the entry
 * whose M equals the rows per image of a launch (it then IS that shape's canonical partition). */
int lcm_plan_set(int kind, int M, int N, int K, int aux, int bm, int bn, int splits, int variant);
int lcm_plan_clear(void);
/* the K partition a contraction of this per-image shape runs with (ph = 1: phase-decomposed upsample conv) */
int lcm_canonical_splits(int kind, int m_img, int N, int K, int aux, int ph);

/* ---- LayerNorm -> Linear as one contraction (BasicTransformerBlock: norm1 -> attn1.to_q|k|v, norm2 -> attn2.to_q,
 * norm3 -> ff.net.0.proj) ----
 *   LN(x) W^T + b  =  rstd[m] * (sum_k x[m][k] W'[n][k] - mean[m] * ln_g[n]) + ln_c[n]
 * with W' = gamma (*) W in fp16, ln_g[n] = sum_k W'[n][k] and ln_c[n] = sum_k beta[k] W[n][k] + b[n] in fp32 (packed once
 * by the host).  The row statistics (sum, sum of squares over K, fp32) are accumulated from the A fragments while the
 * kernel walks K, so the normalisation costs no launch and no extra pass over the activations.  K is never split.
 * epilogue: LCM_EPI_NONE or LCM_EPI_GEGLU (ln_g / ln_c in the packed row order of W).  img_rows as lcm_gemm_f16. */
int lcm_gemm_ln_f16(const void* A, int lda, const void* W, const void* ln_g, const void* ln_c, float eps,
                    void* out, int ldo, int M, int N, int K, int epilogue, int img_rows, void* stream);

/* FeedForward of a BasicTransformerBlock in ONE launch:  out = x + Linear_2( GEGLU( Linear_1( LayerNorm(x) ) ) )
 * = lcm_gemm_ln_f16(epilogue GEGLU) followed by lcm_gemm_f16(bias b2, residual x), bit for bit, with the [M, 4C] intermediate
 * kept in registers (replaces diffusers' FeedForward(activation_fn="geglu") under norm3 and the residual add).
 * W1 [8C][C]: gamma (*) W of ff.net.0.proj, value / gate rows interleaved in blocks of 16; ln_g / ln_c [8C] fp32 as for
 * lcm_gemm_ln_f16; W2 [C][4C]: ff.net.2 with its columns in the stored order of the GEGLU output ("operand order": channel
 * 16 P + 4 q + j at column 32 (P >> 1) + 8 q + 4 (P & 1) + j -- the order in which lcm_gemm_*_f16's GEGLU epilogue stores it).
 * out may alias x (every workgroup reads and writes only its own 128 rows).  C = 320 only; a layer whose canonical K
 * partition of the second product has parts (img_rows small) is refused with LCM_EINVAL: use the two launches. */
int lcm_mlp_geglu_f16(const void* x, int ldx, const void* W1, const void* ln_g, const void* ln_c, float eps,
                      const void* W2, const void* b2, void* out, int ldo, int M, int C, int img_rows, void* stream);
/* refresh ln_g (= row sums of the live fp16 W') and ln_c (= c_base + alpha * c_delta; c_out / c_delta may be NULL) after
 * a style LoRA re-merged W' in place */
int lcm_ln_fold_refresh(const void* W, int N, int K, const void* c_base, const void* c_delta, float alpha,
                        void* g_out, void* c_out, void* stream);

/* 1: short-K GEMM launches with more tiles than the chip holds let each workgroup walk several n-tiles with a
 * continuous LDS-DMA pipeline (no ramp / drain per tile); 0 (default): one tile per workgroup.  Bit-identical. */
int lcm_set_persist_n(int on);

/* contraction kernel variant: 0 = register-staged double buffer, 2/3/4 = LDS-DMA pipeline with that many stages */
int lcm_set_kernel_variant(int variant);

/* tile shape the contraction kernels use for an [M x N] output (BM*1000 + BN); for profiling / docs */
int lcm_gemm_tile_config(int M, int N, int batch);

/* ---- 3x3 convolution, padding 1 (ResnetBlock2D.conv1/conv2, Downsample2D, Upsample2D.conv) ----
 * implicit GEMM over K = 9*Cin on MFMA; in: [B,Hin,Win,Cin]; stride 1|2; ups=1 reads the input through a
 * nearest-2x upsample (F.interpolate(scale_factor=2) fused into the loader).  ups=2 computes the same
 * Upsample2D (interpolate -> conv) as four 2x2 PHASE convolutions on the low-resolution input -- output pixel
 * (2y+py, 2x+px) only sees input rows {y-1+py, y+py} / columns {x-1+px, x+px} -- with W the phase-packed weights
 * [4 = py*2+px][Cout][2][2][Cin] (3x3 taps that land on one input pixel pre-summed): 16 instead of 36 multiply-adds
 * per output element and input channel.  Epilogue as lcm_gemm_f16.
 * Odd targets (Upsample2D called with output_size = an odd-sized skip: F.interpolate(size=(2h-1, 2w-1), nearest),
 * which is the 2x result minus its last row / column, then conv with ZERO padding of that cropped image): ups = 1 | 4
 * (output height 2*Hin-1) | 8 (output width 2*Win-1), with the plain 3x3 weights -- the pre-summed phase weights of
 * ups=2 do not hold for the border outputs, ups=2 with a crop flag is refused.
 * Preconditions: Cin % 64 == 0, Cout % 64 == 0.
 */
int lcm_conv3x3_f16(const void* in, const void* W, const void* bias,
                    const void* rowadd, int ld_rowadd, const void* res, void* out,
                    int B, int Hin, int Win, int Cin, int Cout, int stride, int ups,
                    void* stats_out, int64_t stats_bytes, int* slabs_per_image, void* stream);

/* ---- fused GroupNorm(+SiLU) -> 3x3 convolution, stride 1 (ResnetBlock2D norm1->act->conv1, norm2->act->conv2) ----
 * LDS-halo implicit GEMM (csrc/conv_halo.hip).  Input = channel concat [in | in2] (in2 NULL: single source; fused
 * torch.cat of the skip).  gn_scale/gn_shift: fp32 [B][C1+C2] from lcm_groupnorm_affine_f16 (NULL: plain conv);
 * silu=1 applies SiLU after the affine; zero padding is applied AFTER the normalisation, as in the reference
 * graph.  ups=1 reads the (raw) input through a nearest-2x upsample; ups=2 as in lcm_conv3x3_f16 (gn_scale must be NULL).
 * Epilogue as lcm_gemm_f16.
 * Preconditions: C1, C2, Cout multiples of 64.
 */
int lcm_conv3x3_gn_f16(const void* in, int C1, const void* in2, int C2, const void* gn_scale, const void* gn_shift,
                       int silu, const void* W, const void* bias, const void* rowadd, int ld_rowadd, const void* res,
                       void* out, int B, int Hin, int Win, int Cout, int ups, void* stats_out, int64_t stats_bytes,
                       int* slabs_per_image, void* stream);
/* GroupNorm statistics folded into per-(image, channel) fp32 scale/shift tables [B][C1+C2] for the call above;
 * ws as for lcm_groupnorm_f16. */
int lcm_groupnorm_affine_f16(const void* x, int C1, const void* x2, int C2, const void* gamma, const void* beta,
                             void* scale_out, void* shift_out, int B, int HW, int groups, float eps, void* ws,
                             void* stream);
/* launches of the LDS-halo conv with fewer workgroups than this use its pipelined variant (3-stage weight ring,
 * double-buffered halo) instead of the single-buffer high-occupancy one; default 768 */
int lcm_set_halo_pipe_threshold(int wgs);
/* GroupNorm-fused convolution (lcm_conv3x3_gn_f16 with scale / shift): 1 = the raw halo of the next 64-channel chunk
 * is fetched into registers under the taps of the current one (measured slower: 202 VGPRs, two workgroups per CU instead of three), 0 (default) = fetched where it is consumed.  Bit-neutral. */
int lcm_set_halo_prefetch(int on);
/* bit 0 (default on): the LDS-halo convolution's plain (unsplit) launches with a residual fetch the residual tile by LDS-DMA and
 * store the result tile in whole rows through an LDS image of the tile; bit 1 (default off: measured neutral): plain GEMMs with a
 * residual too; 0: 8-byte pieces per lane everywhere.  Bit-neutral. */
int lcm_set_staged_epilogue(int on);
/* 1 (default): stride-1 3x3 convolutions use the LDS-halo kernel; 0: the row-gather implicit GEMM everywhere */
int lcm_set_conv_impl(int impl);

/* ---- 3x3 convolution from the fp32 NCHW latent (UNet conv_in; VAE post_quant_conv+decoder.conv_in) ----
 * in: fp32 [B,4,H,W]; optional pre-transform z = pre_w(4x4 fp32, row=out) * (in * in_scale) + pre_b
 * (AutoencoderKL: latents / scaling_factor -> post_quant_conv, backends/rknnlcm.py:614).
 * W: fp16 [Cout][9][4]; out: fp16 [B,H,W,Cout].  Cout % 16 == 0.  The fp32 input enters the MFMA as fp16 hi + fp16 lo parts
 * (two K slots per value against the same weight): input precision ~22 bits, fp32 accumulation.
 */
int lcm_conv3x3_c4_f32in(const void* in, const void* pre_w, const void* pre_b, float in_scale,
                         const void* W, const void* bias, void* out, int B, int H, int Wd, int Cout, void* stream);

/* ---- 3x3 convolution to a few channels (UNet conv_out -> eps; VAE decoder.conv_out -> RGB) ----
 * in: fp16 [B,H,W,Cin], W: fp16 [Cout][9][Cin], Cout <= 4, Cin % 8 == 0.
 * mode 0: out fp32 [B,H,W,Cout];  mode 1: out u8 [B,H,W,Cout] = rint(clamp(y/2+0.5,0,1)*255)
 * (VaeImageProcessor.postprocess; backends/rknnlcm.py:223,232-236,259); out_f32 optional float copy (NHWC).
 */
int lcm_conv3x3_smalln(const void* in, const void* W, const void* bias, void* out, void* out_f32,
                       int B, int H, int Wd, int Cin, int Cout, int mode, void* stream);
/* The same behind a fused GroupNorm-apply (+SiLU when silu != 0): in is the RAW tensor, gn_scale / gn_shift fp32 [B][Cin] the
 * tables of lcm_groupnorm_from_stats_f16 (out == NULL form) -- conv_norm_out -> SiLU -> conv_out of the UNet and of the
 * AutoencoderKL decoder without the normalised tensor in memory.  Cin % 64 == 0 (MFMA kernel; the choice of kernel depends on
 * Cin only).  gn_scale == NULL: identical to lcm_conv3x3_smalln. */
int lcm_conv3x3_smalln_gn(const void* in, const void* gn_scale, const void* gn_shift, int silu, const void* W,
                          const void* bias, void* out, void* out_f32, int B, int H, int Wd, int Cin, int Cout,
                          int mode, void* stream);

/* ---- GroupNorm (+SiLU) over [B,HW,C1(+C2)] (ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out) ----
 * x2 != NULL normalises the channel concatenation [x | x2] (fused torch.cat of the skip) and writes the
 * concatenated tensor.  ws: fp32 workspace of lcm_groupnorm_ws_bytes().  Deterministic (no atomics).
 */
int64_t lcm_groupnorm_ws_bytes(int B, int HW, int C, int groups);
int lcm_groupnorm_f16(const void* x, int C1, const void* x2, int C2, const void* gamma, const void* beta,
                      void* out, int B, int HW, int groups, float eps, int silu, void* ws, void* stream);

/* GroupNorm (+SiLU) of [x | x2] from producer-written statistics (see lcm_gemm_f16): stats1 = [B*P1][C1][2],
 * stats2 = [B*P2][C2][2] (x2/stats2 NULL: single source).  ws: >= 8*B*(C1+C2) bytes.  No pass over the data for
 * the statistics: one finalize launch (per image x group) + the apply launch.  out == NULL stops after the finalize
 * and leaves the folded per-(image, channel) fp32 tables in ws (scale [B][C1+C2], then shift [B][C1+C2]) for
 * lcm_conv3x3_gn_f16, which applies them while staging its input (x / x2 may then be NULL; C2 is taken as given). */
/* Tensors of at most this many bytes (default 8 MiB) run lcm_groupnorm_from_stats_f16 as ONE launch (each workgroup
 * re-derives its group's statistics, then applies its pixel slice): small-batch passes are launch-latency bound. */
int lcm_set_gn_fused_bytes(int64_t bytes);
int lcm_groupnorm_from_stats_f16(const void* x, int C1, const void* x2, int C2, const void* stats1, int P1,
                                 const void* stats2, int P2, const void* gamma, const void* beta, void* out,
                                 int B, int HW, int groups, float eps, int silu, void* ws, void* stream);

/* ---- LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3) ---- */
int lcm_layernorm_f16(const void* x, const void* gamma, const void* beta, void* out, int M, int C, float eps,
                      void* stream);

/* ---- fused attention softmax(scale*Q K^T) V (Attention in BasicTransformerBlock.attn1/attn2) ----
 * Q: rows b*Sq+s, element (h*d + i) at Q[row*ldq + ...]; K,V likewise with Sk rows per batch; out [B*Sq][ldo].
 * d in {40, 64, 80, 160} (UNet / CLIP heads: 32x32x16-MFMA kernel, O for the whole head in registers) or, without causal
 * mask, d = 512 (AutoencoderKL mid-block attention, one head of 512: 16x16x32-MFMA kernel, 16 query rows per wave,
 * V^T fragments by ds_read_b64_tr_b16).  Online softmax in fp32, no S x S matrix in memory; a query row's result does not
 * depend on B or on the other rows.  scale > 0: the softmax scale (diffusers: d^-0.5).  scale <= 0: Q already carries
 * scale * log2(e) (the caller folded both into the projection that produced Q) and the logits are used as they are.
 */
int lcm_attention_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* out, int ldo,
                      int B, int heads, int Sq, int Sk, int d, float scale, int causal, void* stream);
/* query rows per attention workgroup: 0 = by grid size, 4 = 128, 8 = 256 (streaming kernel only), 2 = 64 (register-staged
 * kernel only; measured slower at batch 1: every workgroup re-stages all K/V tiles).  Bit-neutral within a kernel. */
int lcm_set_attention_waves(int waves);
/* Which kernel serves the long non-causal sequences (Sk >= 128, d in {40, 64, 80}: the UNet's self-attention at the 64^2 / 32^2
 * levels, SDXL): 1 (default) = the streaming kernel (K/V tiles by LDS-DMA into a double buffer, Q pre-scaled, running max
 * carried in the padding k-slots of the QK^T MFMA for d = 40, deferred rescale), 0 = the register-staged kernel that serves
 * everything else.  Which kernel runs is a function of (d, Sk, causal) only -- never of B -- so a request's bits do not depend
 * on the batch; the two kernels differ in rounding (Q scaling in fp16, deferred max). */
int lcm_set_attention_impl(int impl);
/* 1 (default): self-attention over 1024..4096 keys splits the keys of a query block over two wave groups of the workgroup
 * (merged at the end; keyed on the sequence length alone, so batch-invariant); 0: never (A/B switch, changes those bits). */
int lcm_set_attention_ksplit(int on);
/* causal != 0: keys after the query are masked (CLIPTextModel's causal attention mask, transformers; the text
 * encoder call of the pipeline, twin backends/rknnlcm.py:266-367). */

/* ---- CLIPTextEmbeddings: out[b*S+s] = token_embedding[ids[b*S+s]] + position_embedding[s]; ids int32 ---- */
int lcm_embed_tokens_f16(const void* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int S, int D,
                         int vocab, void* stream);

/* ---- row softmax in place over [rows][n] fp16 (AutoencoderKL mid-block attention, d=512 single head) ---- */
int lcm_softmax_rows_f16(void* x, int rows, int n, int ld, void* stream);
/* ---- [z][R][C] -> [z][C][R] transpose, fp16 ---- */
int lcm_transpose_f16(const void* in, int ldi, void* out, int ldo, int R, int C, int batch,
                      int64_t stride_in, int64_t stride_out, void* stream);

/* ---- small-M linear (TimestepEmbedding, ResnetBlock2D.time_emb_proj): M <= 16 ----
 * out[m][n] = act_out( sum_k act_in(x[m][k]) W[n][k] + bias[n] + res[m][n] ), act = SiLU when flagged.
 */
int lcm_linear_smallm_f16(const void* x, int ldx, const void* W, const void* bias, const void* res, int ldr,
                          void* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream);
/* The same with any M and with x / res given for fewer rows than M: row m reads x[m % x_rows] and res[m % res_rows]
 * (the time-embedding MLP of all sampler steps in one launch per layer: rows step-major (step, image), the guidance embedding
 * and SDXL's added embedding given once per image).  A row's result does not depend on M. */
int lcm_linear_rows_f16(const void* x, int ldx, int x_rows, const void* W, const void* bias, const void* res, int ldr,
                        int res_rows, void* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream);

/* ---- Timesteps(flip_sin_to_cos=True, freq_shift=0): out fp16 [B][dim] = [cos | sin](t * f) ---- */
int lcm_timestep_embedding(float t, void* out, int B, int dim, void* stream);
/* nsteps <= 64 timesteps (host array) at once: out fp16 [nsteps][B][dim] */
int lcm_timestep_embedding_steps(const float* t_host, int nsteps, void* out, int B, int dim, void* stream);

/* ---- LCMScheduler.step (backends/rknnlcm.py:596-599), epsilon prediction, fp32 state ----
 * eps: fp32 NHWC [B,h,w,4] (conv_out); eps_uncond != NULL applies classifier-free guidance first.
 * lat (in/out): fp32 NCHW [B,4,h,w]; noise fp32 NCHW (ignored when last).  coef = {sqrt_alpha_t, sqrt_beta_t,
 * c_skip, c_out, sqrt_alpha_prev, sqrt_beta_prev}.
 */
int lcm_scheduler_step(const void* eps, const void* eps_uncond, float guidance, void* lat, const void* noise,
                       const float* coef6, int last, int B, int h, int w, void* stream);

/* ---- adaptive_avg_pool2d(lat,(8,8)) -> fp16 [B,4,8,8] (run_job_with_latents, backends/cuda_worker.py:299-304) */
int lcm_latents_pool8(const void* lat, void* out_f16, int B, int h, int w, void* stream);

/* Live per-launch timing of the MFMA kernels: between begin/end every contraction / attention launch is bracketed by
 * HIP events on its launch stream (main kernel only).  lcm_profile_end synchronises those events and writes one
 * "kernel instantiation<TAB>milliseconds" line per launch, in launch order; returns the number of lines. */
int lcm_profile_begin(int max_launches);
int lcm_profile_end(char* out, int64_t cap);
/* profiling aid: hold the stream busy for `usec` (<= 2 s) so queued launches run back to back */
int lcm_debug_spin(int usec, void* stream);
/* measurement only (tools/seam_cost.py): n_barriers grid-wide seams (release, one monotonic counter, bounded relaxed poll, acquire,
 * re-read of another workgroup's 128-byte record) inside ONE launch of `workgroups` <= 256 co-resident workgroups; state = device
 * memory of >= 16 + workgroups * 128 bytes (word 1 reads 1 afterwards if a spin gave up).  Never on the product path. */
int lcm_debug_grid_barrier(int workgroups, int n_barriers, void* state, void* stream);

/* ---- RGB8 -> PNG on the host (no GPU work): the ``img.save(buf, format="PNG")`` that closes run_job
 * (backends/cuda_worker.py:234-239).  rgb = `height` scanlines of `width` RGB8 pixels `pitch` bytes apart; the image is cut into
 * `stripes` deflate segments (dynamic Huffman over literals + distance-1 runs, filter "Up") compressed in parallel; the bytes
 * written depend on (image, stripes) only.  out must hold lcm_png_bound(width, height, stripes) bytes. */
long long lcm_png_bound(int width, int height, int stripes);
int lcm_png_encode_rgb8(const void* rgb, int width, int height, long long pitch, int stripes, void* out, long long out_cap,
                        long long* out_len);

/* ---- AutoencoderKL.tiled_decode glue (vae.enable_tiling(), backends/cuda_worker.py:91): decoded tiles are fp32
 * pixel-major [B,h,w,3].  blend: vertical=1 -> b[y] = a[ah-extent+y]*(1-y/extent) + b[y]*(y/extent) for y < extent (aw == bw);
 * vertical=0 -> the same along x (ah == bh).  place_tile: crop [0:ch, 0:cw] of a tile into the RGB8 image at (oy, ox) with
 * the (x/2+0.5).clamp * 255 -> rint conversion (optional fp32 copy). */
int lcm_vae_blend_f32(const void* a, int ah, int aw, void* b, int bh, int bw, int B, int extent, int vertical, void* stream);
int lcm_vae_place_tile(const void* tile, int th, int tw, void* out_u8, void* out_f32, int H, int W, int B,
                       int oy, int ox, int ch, int cw, void* stream);

/* ---- LoRA style merge: out = base + alpha * delta over n fp16 elements (n % 8 == 0); out may alias the live weight.
 * Replaces pipe.set_adapters([name],[weight]) / disable_lora() of backends/cuda_worker.py:165-196 (weights are
 * re-merged in place, so captured graphs stay valid). */
int lcm_axpy_f16(const void* base, const void* delta, float alpha, void* out, int64_t n, void* stream);

/* ---- hipGraph capture of the 4-step sampler loop + VAE ---- */
int lcm_graph_begin(void* stream);
int lcm_graph_end(void* stream, void** graph_exec_out);
int lcm_graph_launch(void* graph_exec, void* stream);
int lcm_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif
