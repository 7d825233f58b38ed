/* snx.h -- C ABI of libsnx.so: the MI355X-native (gfx950) SPLADE-ModernBERT training hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI on this path -- its boundary is the
 * Python class API of src/model/splade_modern.py and src/model/losses.py, which dispatches
 * implicitly to aten/cuBLAS/SDPA kernels -- so each entry point below cites the reference
 * statement(s) whose device work it replaces.  `ref:` = /root/reference, `hf:` =
 * transformers/models/modernbert/modeling_modernbert.py (the third-party module the reference
 * wraps at ref:src/model/splade_modern.py:38,69-73).
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host];
 *   - caller owns all buffers; nothing is allocated, freed or synchronised inside;
 *   - work is enqueued on `stream` and is ordered with other work on that stream; the operator entry
 *     points are re-entrant across streams (no hidden state).  snx_model_backward additionally forks its
 *     weight-gradient GEMMs onto ONE process-wide internal stream (event fork/join, joined before it
 *     returns: callers still see everything ordered on `stream`); it is therefore not meant to be called
 *     from several host threads at once (one process drives one GPU).  SNX_BWD_OVERLAP=0 keeps everything
 *     on `stream`;
 *   - return 0 on success, a positive hipError_t if a launch failed, or a negative SNX_E_* code
 *     when the arguments violate a kernel's shape assumptions (checked on the host BEFORE any
 *     launch -- a mis-shaped call never reaches the GPU);
 *   - bf16 tensors are raw uint16 storage (`void*`), row-major, contiguous; "T" is the number
 *     of token rows (all sequences of a call laid end to end), sequence s owns rows
 *     cu_seqlens[s] .. cu_seqlens[s+1]-1 (int32, nseq+1 entries, cu_seqlens[0]=0,
 *     cu_seqlens[nseq]=T); `mask` is the reference's attention_mask flattened to [T] (int64,
 *     1 = token, 0 = padding).
 *   - constraints: head_dim == 64, hidden % 256 == 0 (<= 1024), intermediate % 64 == 0, GEMM K % 64 == 0,
 *     sequence length <= 65535.
 */
#ifndef SNX_H_
#define SNX_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if !defined(__HIP__) && !defined(HIP_INCLUDE_HIP_HIP_RUNTIME_API_H)
typedef struct ihipStream_t* hipStream_t;
#endif

#define SNX_E_SHAPE (-2)
#define SNX_E_ARG (-3)
#define SNX_FWD_SAVE_FOR_BACKWARD 1

/* Architecture constants (ref:huggingface/v33/config.json; hf configuration_modernbert.py:113-162).
 * layer l is a global-attention layer iff l % global_every == 0; `window` is the HALF window
 * (local_attention / 2): key j visible to query i iff |i-j| <= window. */
typedef struct snx_model_desc {
  int32_t vocab, hidden, inter, layers, heads, head_dim;
  int32_t global_every, window, pad_id, reserved0;
  float ln_eps, reserved1;
} snx_model_desc;

int snx_version(void);

/* Number of parameter tensors in the canonical order used by `params` / `grads` arrays:
 *   tok_embeddings.weight, embeddings.norm.weight, then per layer [attn_norm.weight (l>=1)],
 *   attn.Wqkv.weight, attn.Wo.weight, mlp_norm.weight, mlp.Wi.weight, mlp.Wo.weight, then
 *   final_norm.weight, head.dense.weight, head.norm.weight, decoder.bias
 * (= the reference state-dict order, the tied decoder.weight counted once). */
int32_t snx_param_count(const snx_model_desc* d);

/* ---- whole-model entry points (native layer loop) --------------------------------------- */

/* bf16 copies of the fp32 master weights ([out,in] and transposed), refreshed after every
 * optimizer step.  Replaces autocast's per-forward weight casts (ref:train_v33_ddp.py:337). */
size_t snx_weight_cache_bytes(const snx_model_desc* d);
int snx_weight_cache_refresh(const snx_model_desc* d, const void* const* params /*[host] fp32 device ptrs*/,
                             void* cache, hipStream_t stream);

/* Activation arena size for one forward over T token rows (save_for_bwd=0: inference plan). */
size_t snx_model_workspace_bytes(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t save_for_bwd);
size_t snx_model_bwd_workspace_bytes(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t max_seqlen);
/* byte offset, inside a save_for_bwd arena, of the packed arg-max keys u32 [nseq, vocab]
 * (bf16 bits of relu(logit) << 16 | 0xFFFF - row): lets a caller inspect the max-pool routing. */
size_t snx_model_keys_offset(const snx_model_desc* d, int32_t T, int32_t nseq);

/* SPLADEModernBERT.forward (ref:src/model/splade_modern.py:50-88):
 *   ids,mask [T] int64; pos [T] int32 (position of each row inside its sequence);
 *   rope_* [max_pos][32][2] fp32 (cos,sin) tables for theta_global / theta_local (hf:136-163);
 *   -> sparse [nseq, vocab] fp32, token_weights [T] fp32; `saved` = arena (see above). */
int snx_model_forward(const snx_model_desc* d, const void* const* params /*[host]*/, const void* wcache,
                      const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                      const float* rope_global, const float* rope_local, void* saved, float* sparse,
                      float* token_weights, const int32_t* groups /*[host] or NULL*/, int32_t T, int32_t nseq,
                      int32_t max_seqlen, int32_t flags, hipStream_t stream);
/* `groups` (optional): {n, (seq_begin, nseq, max_len) x n} -- consecutive sequence groups of different
 * maximum length laid end to end in ONE call (e.g. the query, positive and negative batches of a
 * training micro-step); NULL = one group of nseq sequences. */
/* One PASS of a micro-step into a row range of a larger arena: the reference calls model(...) three times per micro-step
 * (query, positive, negative: ref:src/train/cli/train_v33_ddp.py:339-343) and back-propagates once.  `saved`, `sparse_all`
 * [nseq_plan, vocab] and `token_weights_all` [T_plan] are laid out for the WHOLE micro-step (snx_model_workspace_bytes(T_plan,
 * nseq_plan, 1)); this call fills token rows [row0, row0 + T) and sequences [seq0, seq0 + nseq) from the pass's own
 * ids / mask / pos / cu_seqlens (cu_seqlens[0] = 0), groups = NULL, flags must save for backward.  After the last pass the
 * arena equals what one snx_model_forward over all rows (with the passes as sequence groups) leaves behind, so ONE
 * snx_model_backward(_units) over (T_plan, nseq_plan) with the concatenated ids / mask / pos / global cu_seqlens runs the
 * micro-step's backward on full-size launches.  row0 = seq0 = 0, T = T_plan, nseq = nseq_plan is snx_model_forward. */
int snx_model_forward_range(const snx_model_desc* d, const void* const* params /*[host]*/, const void* wcache,
                            const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                            const float* rope_global, const float* rope_local, void* saved, float* sparse_all,
                            float* token_weights_all, const int32_t* groups /*NULL for a true sub-range*/, int32_t T_plan,
                            int32_t nseq_plan, int32_t row0, int32_t seq0, int32_t T, int32_t nseq, int32_t max_seqlen,
                            int32_t flags, hipStream_t stream);

/* Backward of the above (the autograd graph of ref:src/model/splade_modern.py:69-86 and of the HF
 * encoder): g_sparse [nseq, vocab] fp32 = dL/d sparse_repr; every grads[i] (fp32, same shape as
 * params[i]) is ACCUMULATED into (+=).  token_weights is treated as non-differentiable. */
int snx_model_backward(const snx_model_desc* d, const void* const* params /*[host]*/, void* const* grads /*[host]*/,
                       const void* wcache, const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens,
                       const int32_t* pos, const float* rope_global, const float* rope_local, const void* saved,
                       const float* g_sparse, void* scratch, const int32_t* groups /*[host] or NULL, as forward*/,
                       int32_t T, int32_t nseq, int32_t max_seqlen, hipStream_t stream);

/* The same backward as a chain of layers + 2 UNITS in execution order: unit 0 = SPLADE tail + tied decoder +
 * head + final norm, unit 1 + i = encoder layer (layers - 1 - i), unit layers + 1 = embeddings.  Consecutive
 * calls over [unit_begin, unit_end) ranges that together cover [0, layers + 2), with the SAME scratch buffer,
 * equal one snx_model_backward.  When a range returns, the gradients of its parameters are complete (except
 * the tied embedding matrix, which receives the decoder's share in unit 0 and the embedding's in the last):
 * `notify` (nullable) is made to wait for them (launch stream AND the internal weight-gradient stream), so
 * that a data-parallel caller can all-reduce that slice there while later units still run -- the overlap the
 * reference gets from DDP's bucketed reducer (ref:src/train/cli/train_v33_ddp.py:539-544,363-364). */
int snx_model_backward_units(const snx_model_desc* d, const void* const* params /*[host]*/,
                             void* const* grads /*[host]*/, const void* wcache, const int64_t* ids,
                             const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                             const float* rope_global, const float* rope_local, const void* saved,
                             const float* g_sparse, void* scratch, const int32_t* groups /*[host] or NULL*/,
                             int32_t T, int32_t nseq, int32_t max_seqlen, int32_t unit_begin, int32_t unit_end,
                             hipStream_t notify, hipStream_t stream);

/* ... over the first T rows / nseq sequences of an arena and scratch laid out for (T_plan, nseq_plan): the backward of a
 * micro-step that placed fewer passes than planned (snx_model_forward_range); `scratch` needs
 * snx_model_bwd_workspace_bytes(T_plan, nseq_plan, max_seqlen). */
int snx_model_backward_units_range(const snx_model_desc* d, const void* const* params /*[host]*/,
                                   void* const* grads /*[host]*/, const void* wcache, const int64_t* ids,
                                   const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                                   const float* rope_global, const float* rope_local, const void* saved,
                                   const float* g_sparse, void* scratch, const int32_t* groups /*[host] or NULL*/,
                                   int32_t T_plan, int32_t nseq_plan, int32_t T, int32_t nseq, int32_t max_seqlen,
                                   int32_t unit_begin, int32_t unit_end, hipStream_t notify, hipStream_t stream);

/* ---- fp32 execution (csrc/f32_path.hip): what the reference computes OUTSIDE torch.autocast -- a bare
 * SPLADEModernBERT.forward (ref:src/model/splade_modern.py:50-88), its inference encoder (ref:benchmark/encoders.py:
 * 309-345) and the fp32 leg of the tolerance protocol.  Same contract as the entry points above with fp32 weights
 * (`params` themselves: no weight cache), fp32 activations and contraction, no bf16 cast point; any hidden size,
 * even head_dim <= 64, any intermediate size (the tiny parity configuration runs here).  rope_* tables are
 * [max_pos][head_dim / 2][2].  The precision path, not the throughput path. */
size_t snx_model_workspace_bytes_f32(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t save_for_bwd);
size_t snx_model_bwd_workspace_bytes_f32(const snx_model_desc* d, int32_t T);
int snx_model_forward_f32(const snx_model_desc* d, const void* const* params /*[host]*/, const int64_t* ids,
                          const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos, const float* rope_global,
                          const float* rope_local, void* saved, float* sparse, float* token_weights, int32_t T,
                          int32_t nseq, int32_t flags, hipStream_t stream);
int snx_model_backward_f32(const snx_model_desc* d, const void* const* params /*[host]*/, void* const* grads /*[host]*/,
                           const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                           const float* rope_global, const float* rope_local, const void* saved, const float* g_sparse,
                           void* scratch, int32_t T, int32_t nseq, hipStream_t stream);
/* fp32 GEMM on v_mfma_f32_32x32x2_f32 with strided operands: C[m,n] (+)= (R[m,n]) + sum_k A[m a_row + k a_k] B[n b_row + k b_k]
 * (nn.Linear in fp32: forward, dX and dW are the same kernel with different strides). */
int snx_gemm_f32(const float* A, int64_t a_row, int64_t a_k, const float* B, int64_t b_row, int64_t b_k, float* C, int64_t ldc,
                 const float* R, int64_t ldr, int32_t M, int32_t N, int32_t K, int32_t accumulate, hipStream_t stream);

/* ---- inference post-processing (ref:benchmark/encoders.py:309-345 NeuralSparseEncoderV33._encode_batch) ---- */
/* Per row of rep [B,V] fp32: entries with rep > 0 and allowed[v] != 0 survive.  k > 0 and more than k survivors:
 * the k largest, weight descending, ties lowest id first (out_sorted[b] = 1); otherwise all survivors in id
 * order (out_sorted[b] = 0).  out_val / out_idx [B,cap] (cap >= min(k,V), or >= V when k <= 0), out_cnt [B].
 * k <= 16384. */
int snx_sparse_topk(const float* rep, const uint8_t* allowed, float* out_val, int32_t* out_idx, int32_t* out_cnt,
                    int32_t* out_sorted, int32_t B, int32_t V, int32_t k, int32_t cap, hipStream_t stream);

/* ---- SPLADELossV33 (ref:src/model/losses.py:183-297) ------------------------------------- */
/* dims [host] = {B, Bp, k, V, label_off, bf16_mm}: q [B,V], p [Bp,V] (Bp > B: all-gathered
 * positives for cross-GPU in-batch negatives, own rows start at label_off), n [B*k,V]; bf16_mm=1
 * rounds the operands of the in-batch mm to bf16 as autocast does (ref:losses.py:155).
 * hp [host] = {temperature, lambda_q(t), lambda_d(t), lambda_neg(t), lambda_margin_mse, lambda_kd, kd_temperature}
 * (the lambda schedule ref:losses.py:75-90 is evaluated by the host caller).
 * tpos [B], tneg [B*k] teacher scores or NULL (MarginMSE, ref:losses.py:92-134).
 * tscores [B,B] teacher score matrix or NULL (KL distillation against the rank's own positives, ref:losses.py:239-253:
 * batchmean KL(softmax(tscores / T_kd) || softmax(q p^T / T_kd)), active when lambda_kd > 0 and tscores != NULL).
 * out9 = {loss, infonce, flops_q, flops_d, flops_neg, margin_mse, nonzero_q, nonzero_d, kd}. */
size_t snx_loss_workspace_bytes(int32_t B, int32_t Bp, int32_t k, int32_t V);
int snx_loss_fwd(const float* q, const float* p, const float* n, const float* tpos, const float* tneg,
                 const float* tscores, const float* hp /*[host]*/, const int32_t* dims /*[host]*/, void* workspace,
                 float* out9, hipStream_t stream);
/* gout = dL/dloss (device scalar); dq [B,V], dp [Bp,V], dn [B*k,V] are overwritten.  use_kd != 0: the forward that
 * filled `workspace` was given tscores with lambda_kd > 0. */
int snx_loss_bwd(const float* q, const float* p, const float* n, const float* gout, const float* hp /*[host]*/,
                 const int32_t* dims /*[host]*/, void* workspace, int32_t use_kd, float* dq, float* dp, float* dn,
                 hipStream_t stream);

/* ---- individual ops (used by the entry points above; exported for parity tests) ---------- */

/* fp32 -> bf16 (autocast weight / activation cast) and fp32 [R,C] -> bf16 [C,R]. */
int snx_cast_bf16(const float* in, void* out, int64_t n, hipStream_t stream);
int snx_cast_transpose_bf16(const float* in, void* out, int32_t R, int32_t C, hipStream_t stream);

/* nn.Linear under autocast (hf:271,300,90-91,490): C[M,N] = A[M,K] B[N,K]^T, bf16, fp32 acc. */
int snx_gemm_nt_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, hipStream_t stream);
/* ... fused with the fp32 residual add of hf:331-332: Hout = Hin + bf16(A B^T). */
int snx_gemm_nt_resid(const void* A, const void* B, const float* Hin, float* Hout, int32_t M, int32_t N, int32_t K,
                      hipStream_t stream);
/* ... Wqkv fused with apply_rotary_pos_emb (hf:271-280): columns < rope_cols (q and k) rotated. */
int snx_gemm_nt_rope(const void* A, const void* B, void* C, const float* rope_tab, const int32_t* pos,
                     int32_t rope_cols, int32_t M, int32_t N, int32_t K, hipStream_t stream);
/* The same with the (cos, sin) row of every token resolved beforehand: rope_rows [M][32][2] fp32 = rope_tab[pos[row]]
 * (snx_rope_rows, once per forward pass and theta; NULL = resolve through pos inside the kernel).  Lets the 256x256
 * kernel's write-back read a row's 256 bytes without the dependent position load.  Same results. */
int snx_rope_rows(const float* cos_sin_tab, const int32_t* pos, float* rope_rows, int32_t T, hipStream_t stream);
int snx_gemm_nt_rope_rows(const void* A, const void* B, void* C, const float* rope_tab, const int32_t* pos,
                          const float* rope_rows, int32_t rope_cols, int32_t M, int32_t N, int32_t K, hipStream_t stream);
/* ... Wi fused with GeGLU (hf:90-91).  B = Wi rows in the interleaved order of snx_cast_geglu_interleave;
 * U [M,N] = Wi output in that column order (saved for backward), Y [M,N/2] = gelu(a) * g. */
int snx_gemm_nt_geglu_fwd(const void* A, const void* B_interleaved, void* U, void* Y, int32_t M, int32_t N,
                          int32_t K, hipStream_t stream);
/* ... dX of mlp.Wo fused with the GeGLU backward: dy = A B^T [M,N=I]; dU [M,2N] (interleaved). */
int snx_gemm_nt_geglu_bwd(const void* A, const void* B, const void* U, void* dU, int32_t M, int32_t N, int32_t K,
                          hipStream_t stream);
/* Dispatch of the five entry points above: from `min_m` rows on (default 8,192; env SNX_NT256_MIN_M) and N % 64 == 0
 * they run the 256x256 persistent kernel (csrc/gemm_nt256.hip), otherwise the 128x128 kernel (csrc/gemm.hip); same
 * results bit for bit (both sum k in the same order).  on = 0 (env SNX_NT256=0) keeps everything on the 128x128
 * kernel, 1 = the default shape policy (wide outputs with the plain / RoPE / GeGLU-forward epilogues), 2 = every
 * eligible shape (tests, A/B); min_m <= 0 leaves the threshold unchanged.  Process-wide host state. */
int snx_nt256_configure(int32_t on, int32_t min_m);
/* fp32 Wi [2I,C] -> bf16 interleaved copy out [2I,C] and/or its transpose out_t [C,2I]: every 64-row
 * group = [a rows 32q..32q+31 | g rows 32q..32q+31] (so a and its gate meet in one lane of the GEMM). */
int snx_cast_geglu_interleave(const float* in, void* out, void* out_t, int32_t I, int32_t C, hipStream_t stream);
/* weight gradient of a Linear: dW[N,K] += dY[M,N]^T X[M,K]  (N, K multiples of 128).
 * ORDERED REDUCTION (round 5; process switch "det_reduce", default 1): the token range is split over workgroups as
 * before, but every workgroup stores its partial tile into the caller-owned workspace `ws` and a second small kernel adds
 * a tile's partials to dW in a FIXED order -- two runs give the same bits, whatever order the workgroups finish in (the
 * role of torch's deterministic cuBLAS reduction; the float-atomic flush of rounds 1-4 stays behind "det_reduce" = 0).
 * `ws` needs snx_gemm_tn_workspace_bytes() bytes (0 for single-writer schedules; a bound that holds for every setting of
 * the process switches), is scratch (no state between calls) and may be shared by launches on ONE stream; launches on
 * different streams need workspaces of their own.  Missing or too small: SNX_E_ARG. */
typedef struct snx_tn_problem {
  const void* dY; /* [M, N] bf16 */
  const void* X;  /* [M, K] bf16 */
  float* dW;      /* [N, K] fp32, += */
  int32_t N, K;
  int32_t interleaved; /* dY columns in the interleaved GeGLU order (N = 2I) */
  int32_t reserved;
} snx_tn_problem;
size_t snx_gemm_tn_workspace_bytes(const snx_tn_problem* probs /*[host]; pointers unused*/, int32_t nprob, int32_t M);
int snx_gemm_tn_accum(const void* dY, const void* X, float* dW, int32_t M, int32_t N, int32_t K, void* ws,
                      size_t ws_bytes, hipStream_t stream);
/* same with dY's columns in the interleaved GeGLU order; dW rows land in the natural Wi order. */
int snx_gemm_tn_accum_interleaved(const void* dY, const void* X, float* dW, int32_t M, int32_t N, int32_t K, void* ws,
                                  size_t ws_bytes, hipStream_t stream);
/* up to 4 weight-gradient problems over the SAME M token rows (the four Linears of one encoder layer, whose
 * nn.Linear backward torch runs as four GEMMs) in one launch.  From 8,192 token rows on: the 256x256 persistent kernel
 * (csrc/gemm_tn256.hip: one workgroup per CU, one flush per workgroup); below, and for the ragged
 * rest of M % 64 rows: the concatenated 128x128 output tiles fill whole rounds of the resident workgroups
 * (csrc/gemm.hip).  N, K multiples of 128. */
int snx_gemm_tn_accum_group(const snx_tn_problem* probs /*[host]*/, int32_t nprob, int32_t M, void* ws, size_t ws_bytes,
                            hipStream_t stream);
/* Process-wide launch hint (host state, read at launch time): leave `n` CUs (0..128, rounded up to a multiple of 8)
 * to other kernels.  The persistent weight-gradient kernel takes one whole CU per workgroup; while RCCL's channel
 * workgroups run an overlapped gradient exchange (the role of DDP's reducer, ref:src/train/cli/train_v33_ddp.py:539-544)
 * a 256-workgroup launch would run its last workgroups as a second wave, so it launches 256 - n instead (its
 * schedule balances any count).  The token partition changes with the count: results differ from the 256-workgroup
 * launch's by fp32 summation order (each count is bit-reproducible by itself). */
/* Process-wide switches of the library (csrc/config.h).  The library reads no environment variable; the Python binding
 * maps its SNX_* variables onto these keys once, at load time (snx/_lib.py), tests and tools call them directly.
 * Keys (default): nt256 (1; 0 off, 2 every eligible shape), nt256_min_m (8192), nt256_coldeal (1: the 64-row units left
 * over after the whole rounds of 256x256 tiles are dealt along column runs, one short tile per workgroup; 0: in tile order;
 * same bits), tn256 (1), tn256_min_m (8192),
 * dec256 (1), dec256_min_t (2048), bwd_overlap (1), side_prio (1), attn_streaming (0), attn_bwd_onepass (1),
 * attn_interleave (0; 1: the workgroups of a launch's sequence groups interleaved in proportion instead of group by group),
 * splade_dh_panels (64), splade_dw_last (2: the routed decoder backward runs its activation half first, gradient rows and bucket lists non-temporal), f32_gemm64 (0), f32_attn_rows (0), wcache_per_tensor (0), resid_in_ln (1: the Wo GEMMs store
 * bf16 and the residual add happens inside the following LayerNorm; 0: in the GEMMs' fp32 epilogue, same bits),
 * stream_nt (271 = 15 + 256; bitmask of non-temporal accesses for streams nobody reads soon: 1 LayerNorm forward's loads of h and y,
 * 2 its store of h_out, 4 LayerNorm backward's loads of the saved h and of dy, 8 the GeGLU-forward GEMM's stores of the saved
 * u, 256 the weight-gradient GEMM's ordered reduce; 16 / 32 / 64 / 128: measured-level or losing variants kept for A/B -- a
 * cache hint, same bits), nt256_rev (0),
 * det_reduce (1: weight gradients -- Linear dW, LayerNorm dw, embedding rows -- summed in a fixed order through the callers'
 * workspaces, bit-reproducible; 0: float atomics in arrival order); diagnostics builds (-DSNX_DIAG) add
 * gemm_cg, gemm_dbg, gemm_mid, tn_splits, nt256_cg, nt256_dbg, nt256_force, tn256_tail_pct, tn256_dbg.  Unknown key or
 * value out of range: SNX_E_ARG. */
int snx_configure(const char* key, int32_t value);
int snx_config_get(const char* key, int32_t* value);
/* The SNX_EXTRA_HIPCC_FLAGS the library was compiled with ("" for the product build); snx/_lib.py refuses a library
 * built with a timing-only diagnostics macro (wrong results by design) unless SNX_ALLOW_DIAG_LIB=1. */
const char* snx_build_flags(void);
int snx_set_reserved_cus(int32_t n);
int snx_get_reserved_cus(void);

/* LayerNorm without bias (hf:61,312,314,420,487), fp32 in -> bf16 out. */
int snx_ln_fwd(const float* h, const float* w, void* x_out, int32_t T, int32_t H, float eps, hipStream_t stream);
/* h_out = h + float(y) (y [T,H] bf16: a Linear's output joining the fp32 residual stream, hf:331-332), x_out = bf16(LN(h_out)):
 * the residual add done inside the LayerNorm that follows it (process switch "resid_in_ln"). */
int snx_ln_fwd_add(const float* h, const void* y, const float* w, float* h_out, void* x_out, int32_t T, int32_t H,
                   float eps, hipStream_t stream);
/* ModernBertEmbeddings.forward (hf:64-71): h = LN(E[ids]) fp32, x0 = bf16(h). */
int snx_embed_ln_fwd(const int64_t* ids, const float* E, const float* w, float* h_out, void* x0_out, int32_t T,
                     int32_t H, float eps, hipStream_t stream);
/* ModernBertPredictionHead tail (hf:489-490): LN(gelu(d)). */
int snx_gelu_ln_fwd(const void* d, const float* w, void* x_out, int32_t T, int32_t H, float eps, hipStream_t stream);
/* backward of the three above; dh (+)= dx, dw += ... (overwrite=1: dh = dx); dh_bf16 (nullable)
 * also receives bf16(dh), the gradient of the next bf16 branch output (saves a cast pass).
 * `ws` (snx_ln_bwd_workspace_bytes / snx_embed_ln_bwd_workspace_bytes; scratch, one stream at a time): the blocks'
 * partial dw rows, added to dw in block order by a second kernel; for the embeddings also the dx rows and the per-id
 * token lists from which gradE[id] += sum_t dx[t] is formed in ascending token order (nn.Embedding's backward, hf:64-71,
 * pad rows skipped).  With "det_reduce" = 0 `ws` may be NULL (float atomics, arrival order). */
size_t snx_ln_bwd_workspace_bytes(int32_t T, int32_t H);
size_t snx_embed_ln_bwd_workspace_bytes(int32_t T, int32_t H, int32_t V);
int snx_ln_bwd(const void* dy, const float* h, const float* w, float* dh, void* dh_bf16, float* dw, int32_t T,
               int32_t H, float eps, int32_t overwrite, void* ws, size_t ws_bytes, hipStream_t stream);
int snx_embed_ln_bwd(const float* dh, const int64_t* ids, const float* E, const float* w, float* gradE, float* dw,
                     int32_t T, int32_t H, int32_t V, float eps, int32_t pad_id, void* ws, size_t ws_bytes,
                     hipStream_t stream);
int snx_gelu_ln_bwd(const void* dy, const void* d, const float* w, void* dd, float* dw, int32_t T, int32_t H,
                    float eps, void* ws, size_t ws_bytes, hipStream_t stream);

/* apply_rotary_pos_emb (hf:196-219) in place on the q and k thirds of qkv [T,3,heads,64]. */
int snx_rope_inplace(void* qkv, const float* cos_sin_tab, const int32_t* pos, int32_t T, int32_t heads,
                     int32_t inverse, hipStream_t stream);

/* ModernBertMLP GeGLU (hf:90-91): y = gelu(u[:, :I]) * u[:, I:], and its backward. */
int snx_geglu_fwd(const void* u, void* y, int32_t T, int32_t I, hipStream_t stream);
int snx_geglu_bwd(const void* u, const void* dy, void* du, int32_t T, int32_t I, hipStream_t stream);

/* Attention (hf:286-297 -> SDPA; masks masking_utils.py:141-150).  window < 0: global layer. */
int snx_attn_fwd(const void* qkv, const int32_t* cu_seqlens, const int64_t* mask, void* out, float* lse, int32_t T,
                 int32_t nseq, int32_t max_seqlen, int32_t heads, int32_t head_dim, int32_t window,
                 hipStream_t stream);
/* Process-wide choice of the backward for sequence groups of <= 256 tokens: 1 (default) the one-pass kernel
 * (csrc/attention_1p.hip: every score formed once, dQ / dK / dV from one launch), 0 the dQ + dK/dV kernel pair
 * (csrc/attention_unit.hip; the second opinion of the parity tests).  Longer groups always stream tile by tile. */
int snx_attn_configure(int32_t bwd_onepass);
/* _ex: `groups` [host] = {n, (seq_begin, nseq, max_len) x n}, n <= 8, consecutive sequence groups with their own
 * maximum length (NULL: one group of max_seqlen) -- only sizes the launch, results are identical. */
int snx_attn_fwd_ex(const void* qkv, const int32_t* cu_seqlens, const int64_t* mask, void* out, float* lse,
                    const int32_t* groups, int32_t T, int32_t nseq, int32_t max_seqlen, int32_t heads,
                    int32_t head_dim, int32_t window, hipStream_t stream);
/* rope_tab/pos (both or neither): also apply the backward of apply_rotary_pos_emb to dq, dk. */
int snx_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                 const int64_t* mask, float* delta_scratch /*[heads,T]*/, void* dqkv, const float* rope_tab,
                 const int32_t* pos, int32_t T, int32_t nseq, int32_t max_seqlen, int32_t heads, int32_t head_dim,
                 int32_t window, hipStream_t stream);
int snx_attn_bwd_ex(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                    const int64_t* mask, float* delta_scratch /*[heads,T]*/, void* dqkv, const float* rope_tab,
                    const int32_t* pos, const int32_t* groups, int32_t T, int32_t nseq, int32_t max_seqlen,
                    int32_t heads, int32_t head_dim, int32_t window, hipStream_t stream);

/* Tied decoder GEMM + SPLADE tail fused (hf:550 + ref:src/model/splade_modern.py:76-86).  From 2,048 token rows on
 * the 256x192 persistent kernel (csrc/decoder256.hip: it zeroes `keys`, builds its row tables in `scratch` and ends
 * with a finalize pass), below the 128x128 kernel (csrc/splade_head.hip); same outputs, any mask.  `scratch`:
 * snx_splade_head_scratch_bytes(T, V) bytes (T = rows of the whole token buffer). */
size_t snx_splade_head_scratch_bytes(int32_t T, int32_t V);
int snx_decoder_splade_fwd_ex(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                              const int64_t* mask, float* sparse, uint32_t* keys, float* token_weights, void* scratch,
                              int32_t T, int32_t nseq, int32_t max_seqlen, int32_t V, int32_t K, int32_t finalize,
                              hipStream_t stream);
int snx_decoder_splade_fwd(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                           const int64_t* mask, float* sparse, uint32_t* keys, float* token_weights, void* scratch,
                           int32_t T, int32_t nseq, int32_t max_seqlen, int32_t V, int32_t K, hipStream_t stream);
/* arg-max-routed backward: dHd [T,H] bf16 (overwritten), gradE [V,H] += , gradb [V] += ;
 * scratch: snx_splade_bwd_scratch_bytes() bytes (per-row bucket lists). */
size_t snx_splade_bwd_scratch_bytes(int32_t nseq, int32_t max_seqlen, int32_t V);
int snx_splade_bwd(const float* g, const uint32_t* keys, const void* Hd, const void* W, const int32_t* cu_seqlens,
                   void* dHd, float* gradE, float* gradb, void* scratch, int32_t T, int32_t nseq,
                   int32_t max_seqlen, int32_t V, int32_t H, hipStream_t stream);

/* ---- fused optimizer step (ref:src/train/cli/train_v33_ddp.py:367-374: clip_grad_norm_ + AdamW) ----
 * All four arrays are flat fp32 [n] (16-byte aligned).  hp [host] = {lr, beta1, beta2, eps, weight_decay,
 * max_norm (<= 0: no clipping)}; `step` is the 1-based optimizer step (bias correction); elements in
 * [nodecay_begin, nodecay_end) get weight_decay 0 (the reference's no-decay group = decoder.bias);
 * norm_out [1] receives the pre-clip global L2 norm; scratch: snx_adamw_scratch_bytes(). */
size_t snx_adamw_scratch_bytes(void);
int snx_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const float* hp /*[host]*/, int64_t step, int64_t nodecay_begin, int64_t nodecay_end,
                        float* norm_out, void* scratch, hipStream_t stream);

/* ---- optional per-kernel-class timing inside the model entry points (HIP events recorded on
 * the launch stream around every kernel class; off by default).  snx_prof_read synchronises on
 * the recorded events and returns, per class, elapsed ms, launch count and algorithmic work
 * (FLOPs for MFMA-bound classes, bytes for HBM-bound ones). */
int snx_prof_enable(int32_t on);
int32_t snx_prof_num_classes(void);
const char* snx_prof_class_name(int32_t i);
int snx_prof_read(double* ms /*[host]*/, int64_t* launches /*[host]*/, double* work /*[host]*/);

#ifdef __cplusplus
}
#endif
#endif /* SNX_H_ */
