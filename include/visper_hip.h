/* visper_hip.h — C ABI of libvisper_hip.so: the MI355X (gfx950) kernels behind the VisPer-LM
 * pre-training step (NTP + per-layer embedding distillation).
 *
 * The reference (SHI-Labs/VisPer-LM) has NO native code and no FFI: every op below replaces a
 * PyTorch/HF call site on the hot path.  Each prototype cites the reference
 * call site it stands in for (paths relative to /root/reference; "HF:" = transformers==4.41.1, the
 * reference's pinned dependency).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers + sizes; all tensors are device pointers owned by the caller (PyTorch caching
 *    allocator); the library never allocates, frees or retains device memory — also not for its own bookkeeping: the two kernels that need
 *    device-side counters (dynamic GEMM tile claims, the distillation loss's reduction tree) take them as caller-owned blocks
 *    (`sched_ws`, `counters`).  Only the vp_comm_* communicator object holds state.
 *  - bf16 tensors are `void*` (raw uint16 bit patterns); fp32 are `float*`; ld* = leading dimension in
 *    ELEMENTS; every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *  - return 0 on success, negative VP_ERR_* otherwise; vp_last_error_string() (thread-local) explains.
 *    Nothing throws or aborts across the ABI.
 */
#ifndef VISPER_HIP_H
#define VISPER_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vp_stream_t; /* hipStream_t */

#define VP_OK 0
#define VP_ERR_BAD_ARG (-1)
#define VP_ERR_UNSUPPORTED_SHAPE (-2)
#define VP_ERR_HIP (-3)

/* epilogue / activation kinds */
#define VP_EPI_NONE 0
#define VP_EPI_GELU 1       /* erf GELU: multimodal_projector/builder.py:57, resampler.py:14 */
#define VP_EPI_QUICK_GELU 2 /* HF: CLIPMLP quick_gelu */
#define VP_EPI_RELU 3       /* aux_heads/da_v2_head.py:331-335 build_mlp */

const char* vp_last_error_string(void);
int vp_version(void);
int vp_device_info(int* cu_count, int* wave_size, long* lds_bytes_per_cu);

/* ---- GEMM: C[M,N] = epi(A[M,K] . B[N,K]^T + bias[N]) + residual[M,N]   (bf16 in, fp32 MFMA accumulate)
 * replaces every nn.Linear / F.linear on the path: HF LlamaAttention/LlamaMLP q,k,v,o,gate,up,down
 * (HF modeling_llama.py), CLIP q,k,v,out,fc1,fc2 (HF modeling_clip.py), lm_head (ola_llama.py:121),
 * mm_projector (multimodal_projector/builder.py:53-60), resampler proj_in/to_q/to_kv/to_out/FF/proj_out
 * (multimodal_projector/resampler.py:9-16,40-44,186-190), depth MLPs (aux_heads/da_v2_head.py:439-442).
 * out_f32=1 writes fp32 (used for weight gradients). force_generic: 0 = auto: the one-wave-per-SIMD 256x256 kernel (code 8) for aligned
 * large problems without bias / activation (M, N multiples of 256, K of 128), its general variant (code 14: bias / activation / residual
 * epilogues, any M >= 256) for such launches from 64 tiles on (erf-GELU launches stay on the 8-phase kernel), the 128x128 kernel for small
 * problems, the bounds-checked generic kernel when K%64 != 0 or rows are not 16-B aligned.  1 = generic, 2 = 128-tile, 3 = the simple
 * persistent 256-tile kernel (kept as the A/B reference), 7 = the 256x256 8-phase ping-pong kernel (the auto choice of rounds 1-2),
 * 8 / 14 = see above, 13 = 4-phase variant of the 8-phase kernel.  Every code computes the same result (the tests compare them bit for bit);
 * any other value is VP_ERR_BAD_ARG. */
int vp_gemm_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                 const void* bias, const void* residual, long ldr, int epilogue, int out_f32, int force_generic, int* sched_ws,
                 vp_stream_t stream);

/* sched_ws (vp_gemm_bf16, vp_gemm_bf16_swiglu, vp_gemm_tn_bf16): NULL = the persistent kernels walk their tiles statically (single GPU:
 * nothing else runs beside the GEMMs).  Non-NULL = vp_gemm_sched_workspace_bytes() bytes of CALLER-OWNED device memory, zeroed once by the
 * caller: the persistent 8-phase kernels then claim their tiles from per-XCD counters in it, so a CU held by a concurrent kernel (an RCCL
 * collective) only costs its own share instead of stalling the static grid (the reference leaves this to DeepSpeed's stream overlap,
 * scripts/zero2.json "overlap_comm").  A launch leaves the block zeroed again: one block serves every launch of one stream; launches that may
 * overlap (different streams) need different blocks.  The library keeps no pointer after the call returns. */
long vp_gemm_sched_workspace_bytes(void);

/* Fused SwiGLU GEMMs for the decoder MLP (HF LlamaMLP.forward: down_proj(act_fn(gate_proj(x)) * up_proj(x))).  The fused
 * gate/up tensor keeps gate and up interleaved in 8-column chunks (g0..7 | u0..7 | g8..15 | ...), which is also the layout
 * vp_swiglu_fwd / vp_swiglu_bwd use.  mode 1: C[M,N] = A B^T (gate_up) and C2[M,N/2] = silu(gate)*up.  mode 2: d_act = A B^T
 * stays on chip, aux = gate_up[M,2N], C[M,2N] = d_gate_up.  M, N multiples of 256, K of 64; else VP_ERR_UNSUPPORTED_SHAPE.
 * mode 1 with aux != NULL: aux is an fp32 [M] row scale applied to the accumulators (gate_up = bf16(acc * scale): RMSNorm's 1/rms when
 * gamma is folded into the frozen weight); one-wave-per-SIMD kernel only (K % 128 == 0, >= 192 tiles), else VP_ERR_UNSUPPORTED_SHAPE. */
int vp_gemm_bf16_swiglu(int mode, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                        void* C2, long ldc2, const void* aux, long ldaux, int* sched_ws, vp_stream_t stream);

/* Residual GEMM that also emits the next RMSNorm's statistics (replaces: HF LlamaDecoderLayer.forward's `residual + o_proj(...)` /
 * `residual + mlp(...)` followed by a separate pass over the stream for the norm, reached from ola_llama.py:105-115): C[M,N] = A B^T + residual
 * exactly as vp_gemm_bf16, plus sumsq_part[M, N/16] (fp32): per-row sums of squares of the result (fp32, before the last bf16 rounding) over
 * 16-column groups; vp_rstd_from_sumsq finishes them.  One-wave-per-SIMD kernel only: M, N multiples of 256, K of 128, 16-byte aligned
 * rows; otherwise VP_ERR_UNSUPPORTED_SHAPE. */
int vp_gemm_bf16_sumsq(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, const void* residual, long ldr,
                       float* sumsq_part, vp_stream_t stream);

/* rstd[row] = rsqrt(sum_j part[row, j] / H + eps) (HF LlamaRMSNorm.forward's variance / rsqrt), fixed summation order.  nparts % 4 == 0. */
int vp_rstd_from_sumsq(int M, int nparts, const float* part, int H, float eps, float* rstd, vp_stream_t stream);

/* QKV projection of a decoder layer with rotate-half RoPE in the GEMM epilogue (replaces: HF LlamaAttention.forward's q/k/v_proj followed by
 * apply_rotary_pos_emb, reached from ola_vlm/model/language_model/ola_llama.py:105-115): C[M,N] = (row_scale (.) A) B^T with the columns
 * < rope_cols (whole 128-wide heads: q and k of the fused qkv weight) rotated exactly as vp_gemm_bf16 + vp_rope would (same rounding points,
 * bit-identical).  cos_t / sin_t: fp32 [S, 64]; pos: int32 [M] position ids or NULL (row % S).  row_scale: fp32 [M] or NULL; it multiplies the
 * fp32 accumulator before the bf16 rounding (RMSNorm's 1/rms when gamma is folded into the frozen weight).  head_dim 128, N % 256 == 0,
 * K % 128 == 0, M >= 256 (any M: the last row tile is range-checked), 16-byte aligned rows; otherwise VP_ERR_UNSUPPORTED_SHAPE. */
int vp_gemm_bf16_rope(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, const float* row_scale,
                      int rope_cols, const float* cos_t, const float* sin_t, const int* pos, int S, vp_stream_t stream);

/* Weight gradient without transposes: C[M,N] (+)= A[K,M]^T B[K,N], both operands contraction-major (autograd of nn.Linear,
 * grad_weight = grad_output.t() @ input: A = dY[tokens,out], B = X[tokens,in]).  Same 8-phase kernel structure with transposing
 * LDS reads.  M, N multiples of 256, K of 64, 16-byte aligned rows; else VP_ERR_UNSUPPORTED_SHAPE (the caller transposes and
 * uses vp_gemm_bf16).  out_f32=1 writes fp32; accumulate=1 (fp32 only) adds into C (chunked lm_head gradient). */
int vp_gemm_tn_bf16(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int out_f32,
                    int accumulate, int* sched_ws, vp_stream_t stream);

/* (measurement / development entry points — vp_debug_* — are NOT part of this ABI: include/visper_hip_debug.h, built with -DVP_DEBUG) */

int vp_transpose_bf16(int rows, int cols, const void* in, long ld_in, void* out, long ld_out, vp_stream_t stream);
/* `batch` independent [rows, cols] -> [cols, rows] transposes in ONE launch (matrix b at in + b * batch_stride_in / out + b * batch_stride_out,
 * strides in elements, multiples of 8): the (B, C, 24, 24) segmentation targets re-laid to the prediction's (B, 576, C) order once per step
 * (base_ola_vlm.py:289-320 compares preds [B, C, 24, 24] with targets of the same layout; the loss kernel streams both linearly). */
int vp_transpose_batched_bf16(int batch, int rows, int cols, const void* in, long batch_stride_in, long ld_in, void* out, long batch_stride_out,
                              long ld_out, vp_stream_t stream);

/* ---- norms.  HF LlamaRMSNorm (modeling_llama.py:53-68); nn.LayerNorm in CLIP and the resampler
 * (resampler.py:12,37-38,189).  rstd/mean: fp32 [M] saved for backward. */
int vp_rmsnorm_fwd(int M, int H, const void* x, long ldx, const void* w, float eps, void* y, long ldy, float* rstd,
                   vp_stream_t stream);
int vp_rmsnorm_bwd(int M, int H, const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                   void* dx, long ld, vp_stream_t stream);
int vp_layernorm_fwd(int M, int H, const void* x, long ldx, const void* w, const void* b, float eps, void* y, long ldy,
                     float* mean, float* rstd, vp_stream_t stream);
int vp_layernorm_bwd_dx(int M, int H, const void* dy, const void* x, const void* w, const float* mean,
                        const float* rstd, const void* dres, void* dx, long ld, vp_stream_t stream);
int vp_layernorm_bwd_wb_partial(int M, int H, const void* dy, const void* x, const float* mean, const float* rstd,
                                float* pw, float* pb, long ld, int rows_per_block, vp_stream_t stream);

/* ---- elementwise.  RoPE rotate-half (HF modeling_llama.py:138-160), SwiGLU (HF LlamaMLP :175-177),
 * GELU / ReLU fwd+bwd for the trainable projector/heads, residual adds. */
int vp_rope(long T, int S, int nheads, int head_dim, void* x, long ld, const float* cos_t, const float* sin_t,
            const int* pos, int inverse, vp_stream_t stream);
/* interleaved = 1: gate / up in 8-column chunks (the frozen-LLM layout of the fused GEMM epilogues); 0: [gate | up] halves
 * (trainable LLM, where gate_proj / up_proj stay contiguous views of the flat parameter buffer) */
int vp_swiglu_fwd(long M, int F, const void* gate_up, long ldg, void* out, long ldo, int interleaved, vp_stream_t stream);
int vp_swiglu_bwd(long M, int F, const void* dact, long ldd, const void* gate_up, void* dgate_up, long ldg, int interleaved,
                  vp_stream_t stream);
/* dst[idx[r], :] += src[r, :] (fp32 atomics, idx < 0 skips): embed_tokens gradient when the LLM is trainable (IFT stage) */
int vp_scatter_add_rows(long n, int H, const void* src, long lds, const int* idx, float* dst, vp_stream_t stream);
int vp_act_fwd(int kind, long n, const void* x, void* y, vp_stream_t stream);
int vp_act_bwd(int kind, long n, const void* dy, const void* x, void* dx, vp_stream_t stream);
int vp_add_bf16(long n, const void* a, const void* b, void* out, vp_stream_t stream);
int vp_add2d_bf16(long R, int C, void* dst, long ldd, const void* src, long lds, vp_stream_t stream);
int vp_copy2d_bf16(long R, int C, void* dst, long ldd, const void* src, long lds, vp_stream_t stream);
/* zero a device range on `stream` (replaces torch's `.zero_()` / `torch.zeros` of gradient buffers and scatter targets) */
int vp_memset_zero(void* ptr, long bytes, vp_stream_t stream);
/* depthwise 7x7 conv, NHWC, zero pad 3 — timm ConvNeXtBlock.conv_dw of the CLIP-ConvNeXt-XXL tower
 * (multimodal_encoder/clip_convnext_encoder.py:161-165).  w is tap-major [49, C]. */
int vp_dwconv7x7_nhwc(int B, int H, int W, int C, const void* x, const void* w, const void* bias, void* y, vp_stream_t stream);
/* bias gradients (column sums), two deterministic stages */
int vp_colsum_partial(long M, int N, const void* x, long ld, float* part, int rows_per_block, vp_stream_t stream);
int vp_colsum_finish(int nslab, int N, const float* part, float* out, float scale, int accumulate, vp_stream_t stream);

/* ---- sequence splice (ola_arch.py:224-254 append_special_tokens, :345-429 embed/concat/pad) as a row
 * gather driven by a host-built index table, and its backward as a gather-sum. */
int vp_gather_rows(long n_out, int H, const void* const* srcs, const long* lds, int nsrc, const int* kind,
                   const int* row, void* out, long ldo, vp_stream_t stream);
int vp_gather_sum_rows(long n_out, int cnt, int H, const void* src, long lds, int src_f32, const int* idx, float scale,
                       void* out, long ldo, int out_f32, int accumulate, vp_stream_t stream);

int vp_cast_f32_to_bf16(long n, const float* x, void* y, vp_stream_t stream);
int vp_cast_bf16_to_f32(long n, const void* x, float* y, int accumulate, vp_stream_t stream);
/* dst[idx[r], :] = (float)src[r, :] for n rows of H bf16 (idx NULL: row r; idx < 0: row skipped): the bf16 lm_head logits of a row chunk widened
 * into their rows of the fp32 [B*S, V] `logits` the reference's forward returns (ola_llama.py:121-122 `logits = self.lm_head(hidden_states);
 * logits = logits.float()`).  H % 8 == 0, lds % 8 == 0, ldd % 4 == 0, 16-byte aligned bases. */
int vp_scatter_rows_bf16_to_f32(long n, int H, const void* src, long lds, const int* idx, float* dst, long ldd, vp_stream_t stream);
int vp_sum_f32(long n, const float* x, float* out, float scale, vp_stream_t stream);
/* out[0] = sum x[i]^2 (global gradient norm for clip_grad_norm_ semantics); part = workspace of vp_sumsq_nblk(n) floats */
int vp_sumsq_nblk(long n);
int vp_sumsq_f32(long n, const float* x, float* part, float* out, vp_stream_t stream);

/* ---- frozen DPT depth decoder (da_v2_head.py:182-321, run under no_grad at base_ola_vlm.py:462-470; output `depth_preds`).
 * NHWC bf16.  3x3 convs = vp_im2col3x3_nhwc (pad 1, stride 1|2, optional input ReLU of ResidualConvUnit; column order
 * (ky, kx, c)) + vp_gemm_bf16; ConvTranspose2d(k = stride) = vp_gemm_bf16 to [pixels, k*k*C] + vp_pixel_shuffle_nhwc;
 * F.interpolate(mode="bilinear", align_corners=True | False) = vp_bilinear_nhwc; (x - min) / (max - min) per image = vp_minmax_norm. */
int vp_im2col3x3_nhwc(int B, int H, int W, int C, int stride, int relu_in, const void* x, void* col, vp_stream_t stream);
int vp_bilinear_nhwc(int B, int H, int W, int C, int Ho, int Wo, int align_corners, const void* x, void* y, vp_stream_t stream);
int vp_pixel_shuffle_nhwc(int B, int H, int W, int k, int C, const void* x, void* y, vp_stream_t stream);
int vp_minmax_norm(int B, long n, const void* x, void* y, vp_stream_t stream);

/* ---- attention.  HF LlamaAttention eager path (modeling_llama.py:191-214: QK^T/sqrt(d) + causal mask,
 * fp32 softmax, PV), HF CLIPAttention (non-causal), PerceiverAttention (resampler.py:46-75; scale d^-1/4
 * on q and k == d^-1/2 on the product).  Tensors are [B,S,H,D] views (strides in elements, multiples of 8; 16-byte
 * aligned bases).  lse: fp32 [B,Hq,Sq] (log2 domain).  vp_attn_bwd's `delta` is a caller-owned fp32 workspace of
 * 3*B*Hq*Sq floats (row dots dO.O, then interleaved (lse, delta) pairs for the DMA-fed dK/dV kernel). */
int vp_attn_fwd(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k,
                long k_bs, long k_ts, const void* v, long v_bs, long v_ts, void* o, long o_bs, long o_ts, float* lse,
                const int* kv_len, int causal, int window, float scale, vp_stream_t stream);
/* forward with additive fp32 score biases: Swin window attention of the frozen segmentation teacher (HF modeling_swin.py SwinAttention:
 * relative position bias per head [Hq,Sq,Skv] + shifted-window mask [bias_nb,Sq,Skv] indexed by batch % bias_nb); either may be NULL */
int vp_attn_fwd_bias(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k,
                     long k_bs, long k_ts, const void* v, long v_bs, long v_ts, void* o, long o_bs, long o_ts, float* lse,
                     const int* kv_len, int causal, int window, float scale, const float* bias_h, const float* bias_b, int bias_nb,
                     vp_stream_t stream);
int vp_attn_bwd(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k,
                long k_bs, long k_ts, const void* v, long v_bs, long v_ts, const void* o, long o_bs, long o_ts,
                const float* lse, const void* dout, long do_bs, long do_ts, void* dq, long dq_bs, long dq_ts, void* dk,
                long dk_bs, long dk_ts, void* dv, long dv_bs, long dv_ts, float* delta, const int* kv_len, int causal,
                int window, float scale, vp_stream_t stream);

/* vp_attn_bwd with the RoPE backward of dq / dk fused into the stores (HF LlamaAttention.forward rotates q, k with
 * apply_rotary_pos_emb before SDPA, modeling_llama.py / modeling_phi3.py; autograd rotates dq / dk back).  D = 128 or 96, causal only.
 * rope_cos / rope_sin: fp32 [positions, D / 2] as for vp_rope; rope_pos: int32 [B, S] position ids or NULL (position = row index).
 * Bit-identical to vp_attn_bwd followed by vp_rope(inverse = 1) on dq and dk. */
int vp_attn_bwd_rope(int B, int Hq, int Hkv, int Sq, int Skv, int D, const void* q, long q_bs, long q_ts, const void* k, long k_bs,
                     long k_ts, const void* v, long v_bs, long v_ts, const void* o, long o_bs, long o_ts, const float* lse,
                     const void* dout, long do_bs, long do_ts, void* dq, long dq_bs, long dq_ts, void* dk, long dk_bs, long dk_ts,
                     void* dv, long dv_bs, long dv_ts, float* delta, const int* kv_len, int causal, int window, float scale,
                     const float* rope_cos, const float* rope_sin, const int* rope_pos, vp_stream_t stream);

/* ---- losses.  NTP CE (ola_llama.py:121-136: logits.float(), shifted CrossEntropyLoss, ignore -100);
 * embedding distillation (base_ola_vlm.py:289-320 _emb_loss; ola_utils.py:108-125
 * calculate_contrastive_loss; :96-106 dist_collect -> tgt_all is the rank-ordered all-gather). */
int vp_ce_fwd_bwd(long rows, int V, void* logits, long ld, const long* labels, float* row_loss, float grad_scale,
                  int write_grad, vp_stream_t stream);
/* vp_emb_loss_fwd: ONE launch (streaming MFMA/dot2 pass + deterministic last-block tree + the B x Bw softmax); out3 = {emb_loss, sl1,
 * contrastive} exactly as _emb_loss returns them (not yet multiplied by the task weight); coef (2B + B*Bw + 1 floats) carries the
 * backward coefficients and d loss / d logit_scale in its last slot.  workspace: vp_emb_loss_workspace(B, Bw, D) floats, uninitialised.
 * counters: vp_emb_loss_counter_bytes() bytes of CALLER-OWNED device memory holding the tree's ticket counters, zeroed once by the caller;
 * every launch leaves the block zeroed, so one block serves all launches of one stream; launches that may overlap (different streams) need
 * different blocks.  The library owns no device memory and keeps no pointer after the call returns.
 * 0 < B <= 64 local predictions, B <= Bw <= 1024 gathered targets (rank-ordered, rank r's rows at r*B), D % 8 == 0; pred [B,D] and
 * tgt_all [Bw,D] bf16 row-major, 16-byte aligned.  logit_scale NULL = no contrastive term. */
long vp_emb_loss_counter_bytes(void);
long vp_emb_loss_workspace(int B, int Bw, long D);
int vp_emb_loss_fwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* mask,
                    const float* logit_scale, float w_contrastive, float* out3, float* coef, float* workspace, unsigned* counters,
                    vp_stream_t stream);
/* The same for `ntask` (<= 8) distillation heads in ONE launch each way (blockIdx.z = head): the heads of a step share B, Bw and rank and
 * differ in D, pointers, workspace and contrastive weight (host arrays of length ntask).  The reference calls _emb_loss once per head and
 * layer (base_ola_vlm.py:445-534). */
int vp_emb_loss_fwd_multi(int ntask, int B, int Bw, const long* D, int rank, const void* const* pred, const void* const* tgt_all,
                          const float* const* mask, const float* const* logit_scale, const float* w_contrastive, float* const* out3,
                          float* const* coef, float* const* workspace, unsigned* counters, vp_stream_t stream);
int vp_emb_loss_bwd_multi(int ntask, int B, int Bw, const long* D, int rank, const void* const* pred, const void* const* tgt_all,
                          const float* const* coef, const float* grad_out, void* const* dpred, vp_stream_t stream);
int vp_emb_loss_bwd(int B, int Bw, long D, int rank, const void* pred, const void* tgt_all, const float* coef,
                    float grad_out, void* dpred, vp_stream_t stream);

/* ---- optimizer.  HF Trainer `adamw_torch` (ola_vlm_train.py:124; torch.optim.AdamW semantics) fused over
 * the flat fp32 master buffer, bf16 shadow refreshed in the same pass. */
int vp_adamw(long n, float* p, const float* g, float* m, float* v, void* bf16_shadow, float lr, float beta1, float beta2,
             float eps, float weight_decay, int step, float grad_scale, vp_stream_t stream);

/* ---- data-parallel exchange points, RCCL over xGMI with the library's own side stream + hipEvent fences.  Replaces
 * DeepSpeed ZeRO-2's gradient reduction (scripts/zero2.json:16-22, launched by scripts/train/pretrain.sh:15) and diffdist's
 * all_gather of the contrastive targets (ola_utils.py:96-106).  One process per GPU; rank 0 creates the unique id
 * (vp_comm_unique_id_bytes() bytes) and distributes it out of band; vp_comm_init is collective.  RCCL is resolved at run time
 * (the librccl already loaded in the process, e.g. PyTorch's, else librccl.so[.1]).
 *   vp_comm_allreduce_async: in-place SUM of `count` elements (dtype 0 = fp32, 1 = bf16, 2 = int32) on the communicator's side
 *     stream, ordered after the work already queued on `compute_stream`; returns at once (overlaps the rest of the backward pass).
 *   vp_comm_wait: `compute_stream` waits on the device for every all-reduce issued since the last wait; the host never blocks.
 *   vp_comm_allgather: recv[world*count] = rank-ordered concatenation of send[count], on `stream`. */
int vp_comm_unique_id_bytes(void);
int vp_comm_unique_id(void* id_out);
int vp_comm_init(int rank, int world, const void* id, void** comm_out);
int vp_comm_allreduce_async(void* comm, void* buf, long count, int dtype, vp_stream_t compute_stream);
int vp_comm_wait(void* comm, vp_stream_t compute_stream);
int vp_comm_allgather(void* comm, const void* send, void* recv, long count, int dtype, vp_stream_t stream);
int vp_comm_info(void* comm, int* rank, int* world);
int vp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
