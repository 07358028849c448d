/* libomlm_hip.so -- C ABI of the MI355X (gfx950) kernels behind the open-musiclm
 * TokenConditionedTransformer hot path.
 *
 * The reference (zhvng/open-musiclm) is pure Python/PyTorch and has NO FFI / plugin / custom-op
 * interface for this path; its only fused-operator seam is the optional
 * xformers.ops.memory_efficient_attention call at open_musiclm/transformer.py:298.  Each entry point
 * below therefore names the reference *Python* code whose arithmetic it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add at each of those sites.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (omlm_last_error() gives the thread-local message);
 *   - all pointers are DEVICE pointers unless stated otherwise; the caller owns every buffer, including
 *     workspaces; the library never allocates device memory, never synchronises, never changes the device;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns at once;
 *   - dtype codes: 0 = fp32, 1 = bf16, 2 = fp16 (IEEE half).  For GEMM / attention OPERANDS, fp32 selects the "bf16x3" split
 *     (hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulation), bf16 / fp16 the single-pass modes
 *     (v_mfma_f32_32x32x16_bf16 / _f16, same rate; precision modes "bf16x3" / "bf16" / "fp16" of the Python layer).  A call
 *     uses ONE 16-bit type: bf16 operands give fp32 or bf16 outputs, fp16 operands fp32 or fp16 outputs.  The library holds
 *     a bf16 and an fp16 copy of every 16-bit kernel; these entry points dispatch on the code;
 *   - row-major everywhere; "ld" = row pitch in elements.
 *
 * Not in this library: the data-parallel gradient exchange.  SURVEY.md 8b sketches an `omlm_allreduce_flat` export; it does not exist --
 * the exchange is ONE torch.distributed (backend "nccl" = RCCL over xGMI) SUM all-reduce of the flat fp32 gradient buffer per optimizer
 * step, issued by the Python host between the captured micro-step and omlm_adamw_clip_step (open_musiclm_amd/parallel.py:
 * DataParallel.allreduce_sum_; the reference: accelerate / DDP bucketed all-reduces on every micro-batch backward, trainer.py:154-155,439).
 * A collective is a host-side call on a communicator the host owns; nothing of it would gain from crossing this C ABI.
 */
#ifndef OMLM_H
#define OMLM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int         omlm_version(void);
const char* omlm_last_error(void);
void        omlm_set_error(const char* msg);

/* C[m,n] = alpha * sum_k A(m,k) B(n,k) (+ Cin[m,n]).  Replaces every nn.Linear / einsum contraction on the path:
 * to_q / to_kv / to_out (transformer.py:203-212,254,333), ConvFeedForward Linear layers (:144,:149), the
 * per-quantizer logit heads einsum 'q c d, b n q d -> b n q c' (open_musiclm.py:173,180), RelativePositionBias
 * Linear layers (transformer.py:48-53) and the autograd backward of each.
 * a_kmajor/b_kmajor: operand stored [K, M] / [K, N] instead of [M, K] / [N, K].
 * a_map / b_map / c_map (optional, int32): physical row of each logical row (k row for k-major operands);
 * c_map < 0 skips the row.  a_rows / b_rows: physical row counts (bounds for out-of-range zero fill).
 * workspace / workspace_bytes (optional; 16-bit operands): caller-owned scratch for the deterministic split-K of the peeled tail -- the m-tile
 * rows behind the last full round of 256 x 256 tiles have their K range cut into slices that fill the machine, each slice stores its partial
 * tile to its own plane of the workspace, a reduction kernel adds the planes in a fixed order with the residual and writes the output type.
 * omlm_gemm_tail_workspace_bytes(M, N) bounds what an M x N output needs; a smaller buffer or NULL / 0 keeps the one-launch tail.  The library
 * keeps no pointer: the buffer belongs to the call, one buffer per stream that launches GEMMs concurrently.  $OMLM_GEMM_TAIL_SPLIT=0: off. */
int omlm_gemm(const void* A, const void* B, void* C, const float* Cin,
              const int* a_map, const int* b_map, const int* c_map, long long a_rows, long long b_rows,
              int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
              int a_kmajor, int b_kmajor, int in_dtype, int out_dtype, float alpha,
              void* workspace, long long workspace_bytes, void* stream);
long long omlm_gemm_tail_workspace_bytes(int M, int N);

/* LayerNorm (transformer.py:24-31: F.layer_norm, learnable gamma, beta == 0, eps 1e-5).
 * fwd: y = LN(x)*gamma in out_dtype (pitch ldy); xcast (optional) = cast(x) for the K/V projection, which the
 *      reference feeds with the UN-normalised input (kv_input bound at :228 before the pre-norm at :250).
 * bwd: dx = dx_scale * (dres + LN^T(dy)); dxcast (optional) = cast(dx); dgamma += sum_rows dy * xhat.  workspace (optional,
 *      omlm_layernorm_bwd_workspace_bytes): per-workgroup dgamma partials instead of contended atomics. */
int omlm_layernorm_fwd(const float* x, const float* gamma, void* y, void* xcast, float* mean, float* rstd,
                       int M, int D, int ldy, float eps, int out_dtype, void* stream);
/* the same forward with the result as hi/lo planes of the 16-bit type out_dtype (1 = bf16, 2 = fp16): y = rne16(v), y_lo = rne16(v - y), both
 * at pitch ldy (precision "fp16ff": the FF-in omlm_gemm_planes16 reads both, the backward reads y) */
int omlm_layernorm_fwd_planes(const float* x, const float* gamma, void* y, void* y_lo, float* mean, float* rstd,
                              int M, int D, int ldy, float eps, int out_dtype, void* stream);
long long omlm_layernorm_bwd_workspace_bytes(int D);
int omlm_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* dres, float* dx, void* dxcast, float* dgamma, float* workspace, int M, int D,
                       float dx_scale, int cast_dtype, int dy_dtype, void* stream);   /* dy_dtype: 0 fp32, 1 bf16 / 2 fp16 (GEMM epilogue output) */
/* the same with a second residual-gradient term dres2 [M, D] (optional) in the type cast_dtype names (dxcast may be null): the K/V
 * projection's input gradient reaches the attention LayerNorm's backward as a 16-bit GEMM output instead of being added to the fp32
 * residual gradient by the GEMM's own epilogue (146 MB read + 146 MB written per layer at B = 32). */
int omlm_layernorm_bwd2(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* dres, const void* dres2, float* dx, void* dxcast, float* dgamma, float* workspace, int M, int D,
                        float dx_scale, int cast_dtype, int dy_dtype, void* stream);

/* q/k l2-normalise * learned per-dim scale, v pass-through (transformer.py:265-271; utils.py:68-69), dim_head 64. */
int omlm_qk_norm_fwd(const float* q_raw, const float* kv_raw, const float* q_scale, const float* k_scale,
                     void* q, void* k, void* v, int M, int H, int out_dtype, void* stream);
int omlm_qk_norm_bwd(const float* dq, const float* dk, const float* dv, const float* q_raw, const float* kv_raw,
                     const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw,
                     float* dq_scale, float* dk_scale, int M, int H, int out_dtype, void* stream);

/* Attention q / k projections with the l2-norm + learned scale of transformer.py:265-271 folded into the GEMM epilogue (16-bit modes):
   C [M, N] = per 64-wide head h < groups: (A B^T)[:, h] / max(|.|, 1e-12) * scale[0..63]; heads >= groups pass through; columns
   >= c2_col0 go to C2 (pitch ldc2) when C2 is given (v of the k | v projection); norm_out [M, ldnorm] fp32 receives the norms.
   A [M, K], B [N, K] row-major, dtype 1 = bf16 / 2 = fp16 (operands and outputs).  Replaces to_q / to_kv + l2norm + scale
   (transformer.py:254,265-271) without the fp32 pre-norm tensors. */
int omlm_gemm_qknorm(const void* A, const void* B, void* C, void* C2, int c2_col0, int ldc2, const float* scale, float* norm_out,
                     int ldnorm, int groups, long long a_rows, long long b_rows, int M, int N, int K, int lda, int ldb, int ldc,
                     int dtype, void* stream);
/* its backward, from the normalised 16-bit q / k and the saved norms (qn [M, H], kn [M]) instead of the fp32 pre-norm projections;
   outputs as omlm_qk_norm_bwd (dq_raw [M, H*64], dkv_raw [M, 128] 16-bit; dq_scale / dk_scale [64] fp32, accumulated). */
int omlm_qk_norm_bwd2(const float* dq, const float* dk, const float* dv, const void* q, const void* k, const float* qn, const float* kn,
                      const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw, float* dq_scale, float* dk_scale,
                      int M, int H, int dtype, void* stream);

/* Causal multi-query attention with rel-pos bias table and key mask (transformer.py:303-331; the same logical
 * inputs as the xformers seam at :275-301, without materialising attn_bias).  bias: [N, bias_ld] fp32, row = i-j,
 * column = head (the un-gathered MLP output of RelativePositionBias, :60-64).  keymask: [B, N] uint8, 1 = attend.
 * lse: [B, H, N] (log2 domain).  bwd: dq [B*N, H*64], dk, dv [B*N, 64] fp32 overwritten; dbias += ; delta scratch [B,H,N]. */
int omlm_mqa_attn_fwd(const void* q, const void* k, const void* v, const float* bias, const float* biasT,
                      const unsigned char* keymask, void* out, float* lse, int B, int N, int H, float scale, int bias_ld,
                      int dtype, void* stream);
/* biasT: the same table transposed to [ceil8(H)][ld'] with 64 leading zeros per row, zero tail, pre-multiplied by log2(e): the
 * layout the bf16 kernels stream per key tile (omlm_attn_bias_table_floats floats; bias == NULL gives an all-zero table).
 * Built once per forward for all layers; fp32 ("bf16x3") operands read `bias` directly and ignore biasT. */
long long omlm_attn_bias_table_floats(int N, int H);
/* q_scale / k_scale (64 floats each, optional: the learned scales of transformer.py:269-271) or qk_bound > 0 give the bound
 * |q.k| <= max_d |q_scale_d k_scale_d| that lets the bf16 forward exponentiate against a fixed reference point
 * (m_h = scale log2e bound + max bias_h, subtracted from the table) instead of a running maximum; neither: online softmax.
 * p_max_log2: 0 for bf16 / fp32 attention operands; 15 for IEEE half operands -- the reference point is lowered by 15 so that the
 * probability numerators span half's normal range (2^-13 .. 2^15) and the fixed form is selected while scale log2e 2 bound + the
 * table's range < 28 (wider: the flag in the table stays 0 and the forward runs its online-softmax kernel). */
int omlm_attn_bias_prepare(const float* bias, float* biasT, int N, int H, int bias_ld, const float* q_scale,
                           const float* k_scale, float qk_bound, float scale, int p_max_log2, void* stream);
/* The tables of `layers` attention layers over ONE rel-pos table in ONE launch (transformer.py:402-405 computes the bias once per forward and
 * hands it to every layer; here each layer's copy carries that layer's reference point): biasT[l] from q_scale[l] / k_scale[l].
 * biasT / q_scale / k_scale: HOST arrays (length `layers`) of DEVICE pointers; q_scale and k_scale both given or both NULL (then qk_bound). */
int omlm_attn_bias_prepare_group(const float* bias, float* const* biasT, int layers, int N, int H, int bias_ld,
                                 const float* const* q_scale, const float* const* k_scale, float qk_bound, float scale,
                                 int p_max_log2, void* stream);
/* dbias_ws (optional, omlm_mqa_attn_bwd_workspace_bytes(B, N, H) bytes, contents irrelevant on entry and exit): the dQ kernel leaves
 * each wave's d(bias) bins there with plain stores and a small reduction adds them into dbias; without it every wave adds its bins into
 * dbias with device-scope atomics (measured 290 us per layer slower at B = 8, N = 1817, H = 16). */
long long omlm_mqa_attn_bwd_workspace_bytes(int B, int N, int H);
int omlm_mqa_attn_bwd(const void* q, const void* k, const void* v, const float* bias, const float* biasT,
                      const unsigned char* keymask, const void* out, const void* dout, const float* lse, float* delta,
                      float* dq, float* dk, float* dv, float* dbias, float* dbias_ws,
                      int B, int N, int H, float scale, int bias_ld, int dtype, void* stream);

/* Middle of ConvFeedForward: CausalDSConv -> GEGLU -> LayerNorm(F) -> Dropout (transformer.py:122-148).
 * h1: [M, 2*Fp] (value half cols [0,F), gate half cols [Fp, Fp+F)); h2: [M, Fp]; convw: taps re-packed tap-major and
 * padded, [3, 2*Fp] in h1's column layout (from the reference ds_conv.weight [2F,1,3]); gamma padded to [Fp] with zeros; both in
 * the operand dtype of h1 (they are re-read by every row, so in bf16 mode they travel as bf16 like every other operand);
 * dconv is accumulated in the reference layout [2F, 3].  rows are b*nseq + t.  Dropout mask = Philox-4x32-10(seed', element
 * index / 8): one 16-bit draw per element, kept iff draw >= p * 65536; seed' = seed + *seed_dev * golden-ratio (seed_dev optional device word: lets a captured HIP graph draw a new mask
 * on every replay).  With drop_bits given (or p == 0) both operand dtypes take the second-generation kernels (csrc/ffmid2.hip: a thread owns a few
 * channel pairs and walks down a strip of rows -- conv window, taps and gamma in fp32 registers, packed fp32 arithmetic, and a
 * backward that fuses dropout^T .. conv^T in one pass behind a row-sum prepass: no du round trip); their keep-mask comes from a
 * 32-bit integer hash of (seed', element index / 8) with the same 16-bit draws, and always reaches the backward through drop_bits.  drop_bits (optional, [M, Fp/8] bytes): the forward stores the keep-mask, 1 bit per element, and the
 * backward reads it instead of regenerating it (null: the backward regenerates the mask from the same (seed, salt) pair).
 * gh (optional, [M, Fp] in the operand dtype): the forward stores the normalised GEGLU output (before gamma and dropout); handed
 * to the backward, its LayerNorm^T sums and d(gamma) are formed from it without recomputing the conv and the GELU. */
int omlm_ffmid_fwd(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd,
                   int M, int nseq, int F, int Fp, float eps, float p, unsigned long long seed,
                   const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, int dtype, void* stream);
/* The same forward on hi/lo planes of the 16-bit type `dtype` (1 = bf16, 2 = fp16; precision "fp16ff"): h1 (what omlm_gemm_planes16 left), the
 * conv taps and gamma are read as hi + lo (each *_lo plane in its hi plane's layout), h2 leaves as planes (h2 = rne16(y), h2_lo = rne16(y - h2))
 * for the FF-out omlm_gemm_planes16; gh, statistics and keep bits as above.  Strip kernels only: Fp <= 4096, drop_bits required when p > 0.
 * The backward is omlm_ffmid_bwd on the hi planes. */
int omlm_ffmid_fwd_planes(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                          void* h2, void* h2_lo, float* mean, float* rstd, int M, int nseq, int F, int Fp, float eps, float p,
                          unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, int dtype, void* stream);
long long omlm_ffmid_bwd_workspace_bytes(int F, int Fp);
int omlm_ffmid_bwd(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* mean,
                   const float* rstd, void* du_tmp, void* dh1, float* dgamma, float* dconv, float* workspace,
                   int M, int nseq, int F, int Fp, float p, unsigned long long seed,
                   const unsigned long long* seed_dev, const unsigned char* drop_bits, const void* gh, int dtype, void* stream);
int omlm_colsum_accumulate(const float* part, float* out, int P, int C, int ldp, void* stream);
/* out_i[c] += sum_p part_i[p * ldp_i + c] for `count` problems in ONE launch: the d(gamma) partial rows of every LayerNorm of a backward
 * pass (omlm_layernorm_bwd2 with dgamma NULL and a workspace leaves its [min(M, 2048), D] partial rows there instead of summing them). */
typedef struct omlm_colsum_desc { const float* part; float* out; int P, C, ldp; } omlm_colsum_desc;
int omlm_colsum_group(const omlm_colsum_desc* problems, int count, void* stream);
/* 1 (default, or $OMLM_FFMID_IMPL): the column-strip kernels where their preconditions hold (Fp <= 4096, drop_bits present when
 * p > 0, and gh present for the backward); 0: the wave-per-row kernels (A/B runs, tests). */
int omlm_ffmid_set_impl(int impl);

/* Embedding gather + start-token interleave + concat (open_musiclm.py:123-145; utils.get_embeds :126-143) and its
 * transpose with the grad_shrink factor (utils.py:60-61).  tables/starts/pos: HOST arrays of device pointers. */
/* table_rows / pos_rows (optional HOST arrays, length nseq): row counts of the tables; an id or position past its table is
 * skipped like a pad (never an out-of-bounds access) and ORed into *err_flag (optional DEVICE int: bit 0 token id, bit 1
 * position, bit 2 label >= V from the cross entropy) -- torch's embedding raises a device assert in the same situation. */
int omlm_embed_gather_fwd(const int* ids, const int* seg, const int* posidx,
                          const float* const* tables, const float* const* starts, const float* const* pos,
                          int nseq, float* out, int B, int N, int D,
                          const long long* table_rows, const long long* pos_rows, int* err_flag, void* stream);
int omlm_embed_gather_bwd(const int* ids, const int* seg, const int* posidx,
                          float* const* dtables, float* const* dstarts, float* const* dpos,
                          int nseq, const float* dx, int B, int N, int D, float alpha,
                          const long long* table_rows, const long long* pos_rows, int* err_flag, void* stream);

/* Training-batch preparation as one launch: eos append + labels + last-token drop + conditioning pad / eos masking + key mask
   (TokenConditionedTransformerWrapper.forward, open_musiclm.py:340-376), per-quantizer offsets + start markers + concatenation
   (TokenConditionedTransformer.forward, :116-130,:139-145) and the forgetful mask (generate_mask_with_prob, utils.py:49-56: the n_drop
   largest of scores[b, 1:] are dropped).  ids[s]: int64 [B, len[s]]; labels[s]: int32 [B, len[s] + 1] or null; ids32 int32 [B, N]
   (-2 = start token, offsets applied); keymask uint8 [B, N]; scores fp32 [B, N] or null (no forgetful mask);
   N = sum_s (len[s] + 1) + (nseq - 1). */
int omlm_prepare_train_batch(const long long* const* ids, int* const* labels, const int* len, const int* eos, const int* Q,
                             const int* codebook, int nseq, int B, int pad_id, const float* scores, int n_drop,
                             int* ids32, unsigned char* keymask, int N, void* stream);

/* F.cross_entropy(logits, labels) pieces (open_musiclm.py:401-405): per-row lse + NLL sum; (softmax-onehot)*coef*g. */
int omlm_cross_entropy_fwd(const float* logits, const int* labels, float* row_lse, float* nll_sum,
                           int R, int V, int ld, int* err_flag, void* stream);
int omlm_cross_entropy_bwd(const float* logits, const int* labels, const float* row_lse, const float* gscale,
                           float coef, void* dlogits, int R, int V, int ld, int ldd, int out_dtype, void* stream);

/* clip_grad_norm_ + Adam/AdamW on flat buffers (optimizer.py:10-34; trainer.py:444-447). */
/* out[0] += sum g^2; partials (optional, >= 2048 floats) selects the bit-reproducible two-pass form (identical clip on every replica). */
int omlm_sumsq_accumulate(const float* g, long long n, float* out, float* partials, void* stream);
int omlm_adamw_clip_step(float* p, float* g, float* m, float* v, void* p16, long long n,
                         float lr, float beta1, float beta2, float eps, float wd, int step,
                         float gscale, const float* gnorm_sq, float max_norm, int decoupled, int zero_grad, int p16_dtype,
                         const float* ls_state, void* stream);
/* p16 (optional): 16-bit shadow of p in p16_dtype (1 bf16 / 2 fp16); a non-finite *gnorm_sq skips the update (gradients still cleared).
   ls_state (optional, precision "fp16"): 5 floats on the device {loss scale, good steps since its last change, skipped steps, applied
   steps, scale of the last finished step}: gradients are divided by ls_state[0] and the bias corrections use step = ls_state[3] + 1
   instead of `step`; omlm_loss_scale_update copies ls_state[0] to ls_state[4] before it halves / doubles the scale. */
/* after the last parameter group of a step: non-finite *gnorm_sq -> scale = max(scale * backoff, scale_min), skipped += 1; else applied += 1
   and after `interval` consecutive good steps scale = min(scale * growth, scale_max).  (No reference counterpart: the reference trains in
   fp32, trainer.py:444-447; this is torch.cuda.amp.GradScaler's rule kept on the device so that the captured step never reads the norm.) */
int omlm_loss_scale_update(float* ls_state, const float* gnorm_sq, float growth, float backoff, int interval,
                           float scale_min, float scale_max, void* stream);

/* operand casts / weight repack */
int omlm_cast_pad(const float* src, void* dst, long long R, int C, int ld_src, int ld_dst, int out_dtype, void* stream);
/* dst[c, r] = cast(src[r, c]): k-contiguous W^T copies for the input-gradient GEMMs (the autograd transpose of
 * nn.Linear, transformer.py:203-212,144,149), refreshed once per optimizer step. */
int omlm_transpose_cast(const float* src, void* dst, int R, int C, int ld_src, int ld_dst, int out_dtype, void* stream);
/* The per-step weight re-packs of a model as ONE launch: problem i casts src [R, C] (pitch ld_src) into dst [R, ld_dst] with zero pad
 * columns (omlm_cast_pad), or -- transpose != 0 -- writes dst[c, r] = src[r, c] (omlm_transpose_cast; pad entries untouched).
 * lo != 0: dst receives the LO PLANE of the cast, rne16(v - rne16(v)), instead of the cast itself (the hi/lo weight planes of precision
 * "fp16ff": omlm_gemm_planes16, omlm_ffmid_fwd_planes). */
typedef struct omlm_cast_pad_desc { const float* src; void* dst; int R, C, ld_src, ld_dst, transpose, lo; } omlm_cast_pad_desc;
int omlm_cast_pad_group(const omlm_cast_pad_desc* problems, int count, int out_dtype, void* stream);

/* fp32-grade GEMM ("bf16x3") on bf16 hi/lo operand planes through the bf16 LDS-DMA tile kernels: A and B point at bf16 hi planes
 * laid out as omlm_gemm takes bf16 operands, the lo plane of each lies a_plane_bytes / b_plane_bytes behind it
 * (omlm_split_planes: hi = fp32 truncated to bf16, lo = RNE(x - hi)).  One launch, k-loop three times as long:
 * hi*hi + hi*lo + lo*hi with fp32 accumulation -- the products the fp32 path of omlm_gemm forms, at the bf16 kernels' rate.
 * Row maps as in omlm_gemm; k-row maps (k-major operand with a map) are not supported. */
int omlm_gemm_planes(const void* A, long long a_plane_bytes, const void* B, long long b_plane_bytes, void* C, const float* Cin,
                     const int* a_map, const int* b_map, const int* c_map, long long a_rows, long long b_rows,
                     int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                     int a_kmajor, int b_kmajor, int out_dtype, float alpha, void* workspace, long long workspace_bytes, void* stream);
int omlm_split_planes(const float* x, void* planes, long long n, long long plane_elems, void* stream);
/* Precision "fp16ff" (round 5): the forward of the two ConvFeedForward linears (transformer.py:144,149 -- 86-88 % of the fp16 logits-error
 * variance, profiles/r05_error_budget.md) on hi/lo planes of the 16-bit operand type `dtype` (1 = bf16, 2 = fp16).  C = A B^T (+ Cin), A [M, K]
 * and B [N, K] row-major, every *_lo plane in its hi plane's layout (separate allocations are fine); one launch, hi*hi + hi*lo + lo*hi with
 * fp32 accumulation.  C_lo NULL: C is fp32 (Cin optional).  C_lo given: C = rne16(v) and C_lo = rne16(v - C) in `dtype` (no Cin) -- the next
 * forward kernel reads the un-rounded value hi + lo, the 16-bit backward reads C alone.  a_map / c_map (optional): row maps as in omlm_gemm
 * (the logit heads of the same mode, open_musiclm.py:163-186). */
int omlm_gemm_planes16(const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, void* C_lo, const float* Cin,
                       const int* a_map, const int* c_map, long long a_rows, long long b_rows, int M, int N, int K,
                       int lda, int ldb, int ldc, int ldcin, int dtype, void* workspace, long long workspace_bytes, void* stream);
/* Round 6: the same contraction for IEEE-half planes with the two CORRECTION products on fp8 at twice the matrix rate (csrc/gemm_mx.hip):
 *   C = A_hi B_hi^T (v_mfma_f32_32x32x16_f16)  +  2^(ea + eb - 11) (A_hi8 B_lo8^T + A_lo8 B_hi8^T)  (v_mfma_scale_f32_32x32x64_f8f6f4)
 * A [M, K] / B [N, K]: the half hi planes (pitch lda / ldb elements).  A8 / B8: the operand's two fp8 (e4m3) planes [hi8 | lo8] at the half plane's
 * row pitch in BYTES (2 lda / 2 ldb; element k of a row at byte k; bytes [K, ceil128(K)) zero), the lo8 plane a8_stride / b8_stride bytes behind
 * the hi8 plane, rows padded to a multiple of 256 (both planes readable in full).  a_scale / b_scale: one E8M0 byte per row,
 * hi8 = fp8(hi 2^-(byte - 127)), lo8 = fp8(lo 2^-(byte - 127 - 11)) (omlm_layernorm_fwd_mx / omlm_ffmid_fwd_mx / omlm_quant_rows_mx write all
 * of it).  2 x the k-tiles of a plain GEMM instead of omlm_gemm_planes16's 3 x; what the fp8 corrections leave in the logits:
 * profiles/r06_error_budget_fp8corr.md.  C_lo given: planes out (no Cin) -- C the half hi plane rne16(v), C_lo the remainder v - C as half
 * (c_lo_bf8 == 0) or as bf8 (e5m2) BYTES at the same element pitch ldc (c_lo_bf8 != 0: the upper byte of a half, same exponent range, no
 * scale; C + C_lo ~ v to 2^-14 -- h1 of the ConvFeedForward forward, read back by omlm_ffmid_fwd_mx: half the lo plane's bytes written and
 * read); else fp32 (+ Cin).  workspace as in omlm_gemm (omlm_gemm_mx16_workspace_bytes(M, N, K)).  K % 64 == 0. */
int omlm_gemm_mx16(const void* A, const void* A8, long long a8_stride, const unsigned char* a_scale,
                   const void* B, const void* B8, long long b8_stride, const unsigned char* b_scale,
                   void* C, void* C_lo, int c_lo_bf8, const float* Cin, long long a_rows, long long b_rows,
                   int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                   void* workspace, long long workspace_bytes, void* stream);
long long omlm_gemm_mx16_workspace_bytes(int M, int N, int K);
/* Producers of omlm_gemm_mx16's operand form (fp16 only).  Each writes the half hi plane exactly as its plain / planes sibling does, plus the fp8
 * planes [hi8 | lo8] (row pitch = 2 x the half plane's pitch in elements, in bytes; lo8 plane `*_stride` bytes behind hi8; bytes behind the row's
 * last element zero up to the next multiple of 128) and one E8M0 scale byte per row, 2^e >= 2^-8 x a bound of the row's largest entry:
 *   omlm_layernorm_fwd_mx : transformer.py:24-31 in front of FF-in; bound = sqrt(D) max|gamma| (what a LayerNorm output cannot exceed)
 *   omlm_ffmid_fwd_mx     : omlm_ffmid_fwd_planes with h2 in this form; bound = sqrt(F) max|gamma / keep| the same way.  h1_lo: the lo plane
 *                           of h1 as bf8 BYTES at h1's element pitch (what omlm_gemm_mx16 writes with c_lo_bf8 != 0)
 *   omlm_quant_rows_mx    : fp32 weights (all problems of a model in one launch); hi = rne_half(w), exact row maximum; rows' tails untouched
 *                           (zero-fill the buffer once) */
int omlm_layernorm_fwd_mx(const float* x, const float* gamma, void* y, void* y8, long long y8_stride, unsigned char* scale8,
                          float* mean, float* rstd, int M, int D, int ldy, float eps, void* stream);
int omlm_ffmid_fwd_mx(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                      void* h2, void* h2_8, long long h2_8_stride, unsigned char* scale8, float* mean, float* rstd, int M, int nseq, int F, int Fp,
                      float eps, float p, unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, void* stream);
typedef struct omlm_quant_rows_desc { const float* src; unsigned char* dst8; long long lo_stride; unsigned char* scale8; int R, C, ld_src, ld8; } omlm_quant_rows_desc;
int omlm_quant_rows_mx(const omlm_quant_rows_desc* problems, int count, void* stream);
/* All weight-gradient contractions of a backward pass in one launch (autograd of nn.Linear, transformer.py:203-212,144,149:
 * dW += dY^T X).  Problem i: C_i [M_i, N_i] fp32 (accumulated, +=; c_map optional: physical row of logical row m, < 0 skips)
 * from 16-bit k-major operands A_i [K_i, M_i] (pitch lda) and B_i [K_i, N_i] (pitch ldb); pitches are multiples of 8 elements.
 * splits: K-splits per output tile (0: chosen for whole machine rounds; > 1 accumulates with fp32 atomics). */
typedef struct omlm_gemm_wgrad_desc {
    const void* A; const void* B; float* C; const int* c_map;
    int M, N, K, lda, ldb, ldc;
} omlm_gemm_wgrad_desc;
int omlm_gemm_wgrad_group(const omlm_gemm_wgrad_desc* problems, int count, int splits, int dtype, void* stream);   /* dtype: operands of ALL problems, 1 bf16 / 2 fp16 */

/* RelativePositionBias MLP helpers (transformer.py:55-64): SiLU layers around omlm_gemm. */
int omlm_relpos_first_fwd(const float* w0, const float* b0, float* pre, float* z, int n, int Hd, void* stream);
int omlm_relpos_first_bwd(const float* ds, float* dw0, int n, int Hd, void* stream);
int omlm_bias_silu_fwd(const float* a, const float* b, float* pre, float* z, long long R, int C, void* stream);
int omlm_silu_bwd(const float* dz, const float* pre, float* ds, long long total, void* stream);
int omlm_bias_add(const float* a, const float* b, float* out, int R, int C, int ld, void* stream);
/* The whole MLP as ONE forward launch and TWO backward launches (round 5; Hd = 256 or 512, H <= 16: every shipped config), replacing the
 * 21 launches of the layer-by-layer path above.  RelativePositionBias.forward (transformer.py:55-64) restricted to the n causal distances
 * 0 .. n - 1: table[r, h] = net(r)[h].  w0 = net.0.0.weight viewed [Hd]; W1 / W2 = net.1.0 / net.2.0 weights [Hd, Hd]; W3 = net.3.weight
 * [H, Hd].  pre* / z* [n, Hd] are the layers' pre-activations / SiLU outputs, written by the forward when non-null (all or none) and
 * read by the backward; table / dtable are [n, ldb] (ldb <= 16, pad columns zero).  The backward ACCUMULATES into the g* buffers
 * (deterministic: one owner thread per element, rows summed in ascending order); scratch holds 3 * n * Hd floats. */
int omlm_relpos_mlp_fwd(const float* w0, const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, float* pre0, float* z0, float* pre1, float* z1, float* pre2, float* z2,
                        float* table, int n, int Hd, int H, int ldb, void* stream);
int omlm_relpos_mlp_bwd(const float* dtable, const float* W1, const float* W2, const float* W3, const float* pre0, const float* z0,
                        const float* pre1, const float* z1, const float* pre2, const float* z2, float* scratch, float* gw0, float* gb0,
                        float* gW1, float* gb1, float* gW2, float* gb2, float* gW3, float* gb3, int n, int Hd, int H, int ldb, void* stream);

/* Nearest-codeword kernels: ClapQuantized.quantize -> ResidualVQ eval path (clap_quantized.py:75-87) and
 * HfHubertWithKmeans assign (hf_hubert_kmeans.py:87).  codebooks_T: [nstage][D][C] (transposed); indices int32 [n, nstage].
 * omlm_rvq_encode / _strided use the distance form of vector-quantize-pytorch's EuclideanCodebook -- argmax(-cdist(x, embed)),
 * first maximum: dist = sqrt(max((|x|^2 + |e|^2) - 2 x.e, 0)) in fp32, IEEE root (distances whose roots round equal are a tie for
 * the lowest index); pinned bit for bit against torch.cdist on exactly representable inputs (tests/rvq_cases.py).
 * omlm_nearest_centroid uses sum_d (x_d - c_d)^2 (pinned against sklearn.MiniBatchKMeans.predict).  Ties -> lowest index. */
int omlm_rvq_encode(const float* x, const float* codebooks_T, int* indices, float* residual_out,
                    int n, int D, int C, int nstage, void* stream);
int omlm_nearest_centroid(const float* x, const float* centroids_T, int* indices, int n, int D, int C, void* stream);
int omlm_rvq_encode_strided(const float* x, const float* codebook_T, int* indices, int idx_stride, float* residual_out,
                            int n, int D, int C, void* stream);
/* Fitting side of the residual VQ (reference call sites: trainer.py:689-736 -> clap_quantized.py:75-84 with rq.train(True); the
 * arithmetic is vector-quantize-pytorch's EuclideanCodebook, un-vendored: oracle.rvq_fit_step restates the published update rules;
 * the assignment inside it is the -cdist form above).
 * accumulate: counts[k] += #rows assigned to k, sums[k, :] += those rows (caller zeroes both; indices[i * idx_stride]).
 * kmeans_update: means[k] = sums[k] / counts[k] where counts[k] > 0 (Lloyd step), means_T [D, K] refreshed.
 * ema_update: cluster_size = d cs + (1-d) counts; embed_avg = d avg + (1-d) sums; embed = embed_avg / Laplace-smoothed sizes
 * ((cs + eps) / (sum cs + K eps) * sum cs); embed_T [D, K] refreshed; total_scratch: one device float. */
int omlm_vq_accumulate(const float* x, const int* indices, int idx_stride, float* counts, float* sums, int n, int D, int K, void* stream);
int omlm_vq_kmeans_update(float* means, float* means_T, const float* counts, const float* sums, int K, int D, void* stream);
int omlm_vq_ema_update(float* cluster_size, float* embed_avg, float* embed, float* embed_T, const float* counts, const float* sums,
                       float* total_scratch, int K, int D, float decay, float eps, void* stream);

/* AR sampler: eos suppression + top_k(thres) + gumbel_sample (open_musiclm.py:309-316; utils.py:65-84). */
int omlm_sample_topk_gumbel(const float* logits, const float* uniform, long long* out, int B, int V, int ld,
                            int k, float temperature, int forbid_last, void* stream);

/* The same sampler in graph-replayable form: uniform_base [steps, B, V] and hist [steps, B] (optional) are indexed by the
 * DEVICE counter *step_dev; out [B] is the fixed buffer the next decode step embeds from. */
int omlm_sample_topk_gumbel_at(const float* logits, const float* uniform_base, const int* step_dev, long long* out,
                               long long* hist, int B, int V, int ld, int k, float temperature, int forbid_last, void* stream);

/* Sampler + embedding gather of the sampled id, x[b, :] = emb_table[id_b + emb_row_offset] (rows clamped to [0, emb_rows);
 * open_musiclm.py:123-134): the decode step's first launch folded into the sampler (omlm_decode_step then runs with
 * emb_table == NULL). */
int omlm_sample_embed_at(const float* logits, const float* uniform_base, const int* step_dev, long long* out, long long* hist,
                         int B, int V, int ld, int k, float temperature, int forbid_last,
                         const float* emb_table, long long emb_row_offset, long long emb_rows, float* x, int D, void* stream);

/* KV-cached AR decode step: ONE new row (index *pos_dev) per sample through all L layers and the logit head of the quantizer
 * that row predicts -- replaces the reference's full re-forward per sampled id (wrapper.generate, open_musiclm.py:301-321;
 * the trunk is strictly causal, so the logits are the same).  State owned by the caller, all fp32:
 *   Kc/Vc[l]  [B, Nmax, 64]   l2-normalised keys / values of rows < pos (row `pos` is appended by this call)
 *   hist[l]   [B, 2, 2*Fp]    FF-in outputs of rows pos-2, pos-1 (conv state; advanced by this call)
 * Pointer-array members are HOST arrays of L device pointers.  w_dtype 0: fp32 weights, 1: bf16 operand copies (then
 * round_bf16 = 1 rounds the activations the batched path keeps in bf16).  W1p [2*Fp, D] / W2p [D, Fp] / convw [3, 2*Fp] /
 * mid_gamma [Fp] are the padded layouts of omlm_ffmid_fwd.  emb_table (optional): x = emb_table[ids[b] + emb_row_offset]
 * first (open_musiclm.py:123-134); otherwise x must already hold the new row.  head_W (optional) [V1, D]: logits
 * [B, ldV] = LN(x_L) head_W^T.  B <= 8; 16-bit weights with D = 1024 and ln_parts given: B <= 16 (matrix-core step kernels). */
typedef struct omlm_decode_args {
    int B, D, H, L, F, Fp, Nmax, w_dtype, round_bf16, nsplit;     /* nsplit >= ceil(Nmax / 64): attention key ranges */
    float eps, scale;
    const int* pos_dev;                                            /* DEVICE int: index of the row computed by this step */
    const void* const* Wq; const void* const* Wkv; const void* const* Wo; const void* const* W1p; const void* const* W2p;
    const float* const* attn_gamma; const float* const* q_scale; const float* const* k_scale;
    const float* const* ffin_gamma; const float* const* convw; const float* const* mid_gamma;
    float* const* Kc; float* const* Vc; float* const* hist;
    const float* bias_table; int bias_ld;
    const float* final_gamma; const void* head_W; int V1; int ldV;
    const float* emb_table; long long emb_row_offset; long long emb_rows;
    float* x; float* x1; float* q; float* parts; float* u; float* logits;   /* scratch: parts [B, nsplit, H, 66] */
    int* advance_pos; int* advance_step;   /* optional DEVICE counters (+= 1) bumped by the step's last kernel: pass pos_dev (and the
                                            * sampler's step counter) here instead of launching omlm_decode_advance */
    float* ln_parts;                       /* optional scratch, 3 * OMLM_DECODE_LN_PARTS(D, Fp) floats: the batched matrix-core kernels
                                            * (2 <= B <= 8) leave per-workgroup partial sums of their outputs there, and the kernel that
                                            * applies the next LayerNorm adds them up instead of re-reducing every sample's row; null:
                                            * every consumer reduces the rows itself */
    /* precision "fp16ff" (optional; all NULL: the plain 16-bit step): lo planes of the FF-in / FF-out / head weights in the layout of
     * W1p / W2p / head_W.  With them those three launches read W = hi + lo and keep LayerNorm outputs and h1 un-rounded -- the arithmetic of
     * the batched forward's omlm_gemm_planes16 (open_musiclm.py:299-319 evaluated fp32-grade where the error budget puts the error).
     * B >= 2: needs the matrix-core kernels (16-bit weights, D = 1024, ln_parts given, L >= 1), Fp <= 3072. */
    const void* const* W1p_lo; const void* const* W2p_lo; const void* head_W_lo;
    /* optional scratch of the batched FF-out launch (2 <= B <= 16, matrix-core kernels, ln_parts given): with it a tile of 16 output rows
     * is cut into four k-slices (256 workgroups instead of 64) that meet through fp32 slabs, added in slice order by the last to arrive.
     * splitk_ws: OMLM_DECODE_SPLITK_FLOATS(D) floats, contents irrelevant; splitk_cnt: ceil(D / 16) ints, ZERO before the first step
     * (every step leaves them zero).  One decode stream at a time per scratch pair.  NULL: one workgroup per tile walks the whole row. */
    float* splitk_ws; int* splitk_cnt;
} omlm_decode_args;
#define OMLM_DECODE_SPLITK_FLOATS(D) (4 * (((D) + 15) / 16) * 256)
#define OMLM_DECODE_LN_PARTS(D, Fp) ((((D) + 15) / 16 > ((Fp) + 7) / 8 ? ((D) + 15) / 16 : ((Fp) + 7) / 8) * 32)
int omlm_decode_step(const omlm_decode_args* args, const long long* ids, void* stream);
/* *pos_dev += 1, *step_dev += 1 (either may be null): keeps the row / sampler-step counters on the device so that a
 * captured step can be replayed. */
int omlm_decode_advance(int* pos_dev, int* step_dev, void* stream);

/* hardware probe (tests only): raw ds_read_b64_tr_b16 result for a linear LDS image, 64 lanes x 4 int16 */
int omlm_probe_tr16(short* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
