/* tethys_mi.h — C ABI of libtethys_mi.so: the MI355X (gfx950) kernels behind the
 * tethys-speech data-parallel training step.
 *
 * The reference (hyunnnchoi/tethys-speech) has NO plugin / FFI interface: its hot path is
 * stock TensorFlow ops called from speech_jobs/whisper_dist.py and
 * speech_jobs/wav2vec2_dist.py.  Each entry point below therefore cites the
 * TensorFlow call site(s) in those files whose device arithmetic it replaces
 * ("W:" = speech_jobs/whisper_dist.py, "V:" = speech_jobs/wav2vec2_dist.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - no allocation, no synchronisation, no hidden state inside any call: the caller owns
 *     every buffer including workspaces (sizes documented per call);
 *   - return value: 0 = TMI_OK, negative = TMI_ERR_*; never throws, never exits;
 *   - dtype enum: TMI_F32 / TMI_BF16; every reduction accumulates in fp32.
 */
#ifndef TETHYS_MI_H
#define TETHYS_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMI_OK 0
#define TMI_ERR_INVALID (-1)   /* bad argument (shape, alignment, dtype combination) */
#define TMI_ERR_LAUNCH (-2)    /* hipLaunchKernel reported an error */
#define TMI_ERR_UNSUPPORTED (-3)

#define TMI_F32 0
#define TMI_BF16 1

/* ABI version, bumped on any signature change. */
int tmi_abi_version(void);
/* Last HIP error string seen by a launch in this thread (host pointer, static storage). */
const char* tmi_last_error(void);
/* Reproducible reductions, process-wide, read at launch time (returns the previous setting).  On: tmi_colsum runs one
 * workgroup per column group and every library-chosen split-K that has to fall back to fp32 atomics is capped at two
 * contributions per element (a + b = b + a; splits through workspace slabs are ordered anyway), so no result of those
 * kernels depends on arrival order; callers pair it with the workspace (fixed-order fold) form of tmi_layernorm_bwd.
 * Scope: the kernels of the WHISPER step - the Wav2Vec2 kernels that sum with atomics (GroupNorm parameter and codebook
 * gradients) are not covered.  With both, a Whisper step is bit-reproducible (the reference,
 * TensorFlow on GPU, is not run-to-run reproducible either: no file:line to cite - this is a property the parity tests
 * use, tests/test_whisper_step_gpu.py). */
int tmi_set_deterministic(int on);

/* ------------------------------------------------------------------------------------
 * Strided, batched GEMM with fused epilogue.  Replaces every tf.keras.layers.Dense call
 * (W:89-92 q/k/v/out_proj, W:194-197 fc1/fc2, W:545 lm_head) and, through overlapping-row
 * addressing of a channels-last padded buffer, tf.keras.layers.Conv1D (W:311-312) and the
 * backward matmuls tape.gradient (W:833) derives from them.
 *
 *   for b in [0,nbatch):  C_b[m,n] = epi( sum_{kb<kbatch} sum_{k<K} A(b,kb,m,k) * B(b,kb,k,n) )
 *   A(b,kb,m,k) = A[b*a_sb + kb*a_skb + m*a_sm + k*a_sk]      (element strides)
 *   B(b,kb,k,n) = B[b*b_sb + kb*b_skb + k*b_sk + n*b_sn]
 *   C_b[m,n]    = C[b*c_sb + m*ldc + n]
 * epi(v), applied in this order:
 *   v += bias[b*bias_sb + n]          (bias f32 or NULL; bias_sb = 0 shares one bias over the batch)
 *   if n < scale_cols: v *= scale     (W:141: q = (xWq+b) * head_dim^-0.5)
 *   v += C_old                        (if accumulate != 0)
 *   if aux_out: aux_out[..] = v       (pre-activation saved for backward; layout as C)
 *   if act == 1: v = gelu_erf(v)      (tf.keras.activations.gelu, W:195,333,336)
 *   if aux_in:  v *= gelu_erf'(aux_in[..])   (backward through GELU; layout as C)
 *   if resid:   v += resid[b*r_sb + m*r_ld + n]  (residual add W:228,234 / PE add W:339)
 * splitk > 1 partitions the (kb,k) range over extra workgroups and accumulates with
 * fp32 atomics into a PRE-ZEROED fp32 C (no epilogue terms allowed); splitk == 0 lets the
 * library choose (it only splits epilogue-free fp32-output GEMMs, i.e. weight gradients, and
 * then C must be pre-zeroed); splitk == 1 never splits.
 * workspace (optional scratch in device memory, contents irrelevant, owned by the call while it
 * runs on its stream; TMI_GEMM_WORKSPACE_MIN bytes always suffice): when given, a library-chosen
 * split (splitk == 0) of a long reduction is reduced WITHOUT atomics — every split runs the epilogue
 * into its own fp32 slab there and a streaming pass sums the slabs into C, so C needs no zeroing
 * and fp32 atomics (~0.5 TB/s on MI355X) leave the path.
 * in_dtype: type of A and B; out_dtype: type of C, aux_*, resid.  Valid pairs:
 * (F32,F32), (BF16,BF16), (BF16,F32).  fp32 inputs use the exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32); bf16 inputs use v_mfma_f32_32x32x16_bf16.
 */
#define TMI_GEMM_WORKSPACE_MIN (96ll << 20)
typedef struct tmi_gemm_desc {
  const void* A; const void* B; void* C;
  int64_t M, N, K;
  int64_t a_sm, a_sk, b_sk, b_sn, ldc;
  int64_t nbatch, a_sb, b_sb, c_sb;
  int64_t kbatch, a_skb, b_skb;
  const float* bias; int64_t bias_sb;
  int64_t scale_cols; float scale;
  int32_t accumulate;
  int32_t act;
  void* aux_out; const void* aux_in;
  const void* resid; int64_t r_ld, r_sb;
  int32_t splitk;
  int32_t in_dtype, out_dtype;
  void* workspace; int64_t workspace_bytes;
  /* Dropout on the epilogue value, before the residual add (W:205: x + Dropout(fc2(..)); V:396, V:431):
   * C = resid + (keep ? v / (1 - p) : 0) with the generator of tmi_dropout over the [M, N] output (row m, column n; N <= 2^17).
   * dropout_p == 0 is off.  nbatch must be 1. */
  float dropout_p; uint64_t dropout_seed;
  /* A second, OUTER batch level (0 / 1 = none): problem z = b2 * nbatch + b1 reads A + b1*a_sb + b2*a_sb2 (likewise B, C).
   * For operands whose two batch indices are not one stride apart - per-(sample, head) products on [B, T, H*hd] tensors
   * (W:147-167 in the fp32 parity mode: one launch instead of one per sample).  Only with plain epilogues (scale /
   * accumulate), nbatch * nbatch2 <= 65535; never on the bf16 fast path. */
  int64_t nbatch2, a_sb2, b_sb2, c_sb2;
} tmi_gemm_desc;
int tmi_gemm(const tmi_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last axis of x[rows, C].  Replaces
 * tf.keras.layers.LayerNormalization(epsilon=1e-5) (W:214,216,245,249,253,322,392) and
 * its gradient.  mean/rstd are fp32 [rows], saved for backward.
 * Backward: dgamma[C] / dbeta[C] are ACCUMULATED into (the caller zeroes them); dx = (accumulate_dx ? dx : 0) + dLN/dx.
 * `workspace` (fp32, 16-byte aligned, >= tmi_layernorm_bwd_workspace_bytes(rows, C, emit) bytes, the caller's): every
 * workgroup stores its partial column sums there and a second launch on the same stream adds them in workgroup order -
 * no atomics, bit-reproducible, no slow-down beside other kernels.  workspace == NULL: one fp32 atomic per column per
 * workgroup (order-dependent last bits; serialises at the memory side under load).
 */
int tmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int64_t rows, int64_t C, float eps,
                      int32_t dtype, void* stream);
/* Dropout(LayerNorm(x)) in one pass and its gradient (V:296, V:560, V:779: `layer_norm` then `dropout`): the forward applies
 * the flat dropout generator (tmi_dropout: stream 0, row, column of the [rows, C] output) to y on store; the backward
 * applies the same mask to dy on load (the same seed), then runs tmi_layernorm_bwd.  dropout_p = 0: the plain entries. */
int tmi_layernorm_dropout_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              int64_t rows, int64_t C, float eps, float dropout_p, uint64_t dropout_seed, int32_t dtype,
                              void* stream);
int tmi_layernorm_dropout_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                              void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C, int32_t accumulate_dx,
                              float dropout_p, uint64_t dropout_seed, float* workspace, int64_t workspace_bytes,
                              int32_t dtype, void* stream);
int64_t tmi_layernorm_bwd_workspace_bytes(int64_t rows, int64_t C, int32_t emit);
int tmi_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows,
                      int64_t C, int32_t accumulate_dx, float* workspace, int64_t workspace_bytes,
                      int32_t dtype, void* stream);

/* tmi_layernorm_bwd that also emits what the Dense layer BELOW this LayerNorm's residual stream needs from dx (the
 * gradient this kernel writes is that layer's dy): colsum[c] += sum over rows of dy' (its bias gradient), where
 * dy' = dx, or - when that layer's output went through Dropout (W:205, V:396, V:431) - dy' = mask * dx / (1 - p), which is
 * also written to `masked` [rows][C] (the generator of tmi_dropout over [rows, C] with `dropout_seed`).  masked == NULL:
 * no copy, column sums of dx; masked != NULL with dropout_p == 0: column sums of dx and a plain second copy of dx in
 * `masked` (a snapshot for a weight gradient launched after dx has been overwritten).  Replaces a tmi_dropout and a
 * tmi_colsum pass over dx. */
int tmi_layernorm_bwd_emit(const void* dy, const void* x, const float* gamma, const float* mean,
                           const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows,
                           int64_t C, int32_t accumulate_dx, float* colsum, void* masked,
                           float dropout_p, uint64_t dropout_seed, float* workspace,
                           int64_t workspace_bytes, int32_t dtype, void* stream);

/* out[n] += sum over rows of dY[rows, N] (row stride ld): the bias gradient of a Dense /
 * Conv1D layer.  Accumulates with fp32 atomics (one per column per workgroup); the caller
 * zeroes `out`. */
int tmi_colsum(const void* dy, int64_t ld, float* out, int64_t rows, int64_t N,
               int32_t dtype, void* stream);
/* nbatch matrices dy + b * dy_sb (elements) into nbatch vectors out + b * out_sb in one launch: the bias gradients of the
 * same Dense layer in every transformer layer (deferred, batched weight gradients). */
int tmi_colsum_batched(const void* dy, int64_t ld, int64_t dy_sb, float* out, int64_t out_sb,
                       int64_t rows, int64_t N, int64_t nbatch, int32_t dtype, void* stream);

/* dx = dy * gelu_erf'(u), elementwise over n elements (backward of W:336 where the GELU
 * output feeds the positional add rather than a GEMM). */
int tmi_gelu_bwd(const void* dy, const void* u, void* dx, int64_t n, int32_t dtype, void* stream);
/* The same over nbatch spans of n elements whose starts are dy_sb / u_sb / dx_sb elements apart (one
 * launch for the per-sample spans of a padded buffer). */
/* Dropout (tf.keras.layers.Dropout in training, W:205 / W:342 / W:411) over a [rows, cols] tensor, cols even and
 * <= 2^17:
 *   out[r,c] = (resid ? resid[r,c] : 0) + (keep(seed, r, c) ? in[r,c] * 65536/(65536-thr) : 0),
 * thr = round(p * 65536): element (r, c) is dropped when its 16-bit draw (low / high half, for even / odd c, of one
 * 32-bit hash of (row key of (seed, r), c>>1)) is below thr.  Nothing is stored: the backward applies the same call (same seed) to the
 * incoming gradient.  TF's RNG stream cannot be reproduced, so the mask differs from the reference's; the oracle
 * restates this generator (oracle/dropout.py).  out may alias in. */
int tmi_dropout(const void* in, int64_t ld_in, const void* resid, int64_t ld_res, void* out, int64_t ld_out,
                int64_t rows, int64_t cols, float p, uint64_t seed, int32_t dtype, void* stream);
int tmi_gelu_bwd_batched(const void* dy, const void* u, void* dx, int64_t n, int64_t nbatch, int64_t dy_sb,
                         int64_t u_sb, int64_t dx_sb, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Launch plans: record the launches of one training step once, replay them from ONE call.
 * Replaces what `@tf.function` gives the reference's step (W:818-819: the first call traces the step, every later call
 * replays the traced graph with no Python between its ops).  While a plan records (tmi_plan_begin .. tmi_plan_end, on the
 * calling thread) every launching entry point of this header appends itself, with its arguments copied by value, to the
 * plan and then runs as usual: the recorded step is a real step.  tmi_plan_replay issues the recorded calls in order -
 * the same entry points, argument checks and dispatch rules, no host code in between - adding `seed_delta` to every
 * recorded dropout seed (site seeds are base + step * K + site * K': pass (step_now - step_recorded) * K mod 2^64) and
 * `step_delta` to the `step` argument of tmi_adam_step / _rows / _segments.
 * Ordering between streams is part of the plan: the host reports each event record / stream wait it makes while recording
 * (tmi_plan_note_event_record / tmi_plan_note_stream_wait, raw hipEvent_t / hipStream_t handles that must outlive the
 * plan) and the replay repeats them with hipEventRecord / hipStreamWaitEvent.  tmi_plan_note_callback records a host
 * function called in sequence (what a step does between launches that is not this library's: an RCCL collective).
 * Memory operations of a step go through tmi_memset_async / tmi_memset2d_async / tmi_memcpy_async (device to device)
 * so that they are recorded too.  tmi_plan_size: what = 0 nodes, 1 launches.  One recorder per thread; a plan is
 * replayed by one thread at a time.
 */
typedef struct tmi_plan tmi_plan;
int tmi_plan_create(tmi_plan** out);
int tmi_plan_destroy(tmi_plan* plan);
int tmi_plan_begin(tmi_plan* plan);
int tmi_plan_end(tmi_plan* plan);
int tmi_plan_replay(tmi_plan* plan, uint64_t seed_delta, int64_t step_delta);
int64_t tmi_plan_size(const tmi_plan* plan, int32_t what);
int tmi_plan_note_event_record(void* event, void* stream);
int tmi_plan_note_stream_wait(void* stream, void* event);
int tmi_plan_note_callback(void (*fn)(void));
int tmi_memset_async(void* dst, int32_t value, int64_t bytes, void* stream);
int tmi_memset2d_async(void* dst, int64_t pitch_bytes, int32_t value, int64_t width_bytes, int64_t rows, void* stream);
int tmi_memcpy_async(void* dst, const void* src, int64_t bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Materialised-score softmax (fp32 "parity mode" attention, the reference's own graph
 * shape: W:147-167).  s[rows, Tk] in place; row r belongs to query i = r % Tq.
 * mask_mode 0: none.  mask_mode 1: the reference decoder's additive mask
 * (1 - (1 - band_part(ones,-1,0))) * -1e9 (W:416-418, W:152-153): key j <= i gets -1e9
 * added in fp32, keys j > i are left alone.
 * Backward: ds = p * (dp - sum_j p*dp), dp in place.
 */
int tmi_softmax_fwd(float* s, int64_t rows, int64_t Tq, int64_t Tk, int32_t mask_mode, void* stream);
int tmi_softmax_bwd(const float* p, float* dp, int64_t rows, int64_t Tk, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused (flash-style) multi-head attention, bf16 in / fp32 accumulate, head_dim 64.
 * Replaces W:147-171 (scores, mask, softmax, probs @ v, head merge) without ever
 * materialising [B,H,Tq,Tk].  q/k/v/o are addressed as  ptr[b*sb + t*st + h*64 + d]
 * (element strides), so q, k, v may be column slices of one fused QKV buffer.
 * The query is expected pre-scaled (tmi_gemm scale_cols), as in W:141.
 * stats: fp32 [B,H,Tq,2] = (row max m, 1/row sum) — kept separately because with the
 * reference's -1e9 mask a fully masked row has m = -1e9, where m + log(l) is not
 * representable in fp32.
 * Backward is two kernels (no atomics, deterministic): dq pass owns query rows, dkv pass
 * owns key rows; both recompute p from q, k and stats.  delta[B,H,Tq] = rowsum(do * o) is
 * produced by tmi_attn_bwd itself into `delta` (fp32 workspace).
 */
typedef struct tmi_attn_desc {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
  float* stats;
  int64_t B, H, Tq, Tk;
  int32_t mask_mode;           /* as tmi_softmax_fwd */
  /* backward only */
  const void* d_o; void* dq; void* dk; void* dv;
  int64_t do_sb, do_st, dq_sb, dq_st, dk_sb, dk_st, dv_sb, dv_st;
  float* delta;
  float dq_scale;              /* dq is multiplied by this on store (chain rule of W:141) */
  float score_scale;           /* scores = (q . k) * score_scale (V:349 divides AFTER q.k^T); 0 means 1 */
  /* Dropout on the attention probabilities (W:160), training mode: 0 <= dropout_p < 1, 0 = off.  The keep
   * mask is a function of (dropout_seed, b, head, q, k) (tmi_common.h: tmi_keep_attn; oracle/dropout.py keep_attention);
   * the row sums that normalise the probabilities are taken before dropping, as tf.nn.softmax -> Dropout does.
   * TensorFlow draws the mask once and keeps it for the gradient (W:160): so does this pair of entry points -
   * tmi_attn_fwd evaluates the generator and stores 1 keep bit per score into `drop_mask`, tmi_attn_bwd reads the
   * bits and never hashes.  drop_mask: 16-byte aligned device buffer of tmi_attn_dropmask_bytes(B, H, Tq, Tk) bytes
   * owned by the caller from the forward call to the backward call of the same descriptor; required when
   * dropout_p > 0 (TMI_ERR_INVALID otherwise), ignored when dropout_p == 0.  Layout (private to the two entry
   * points): u32 words [B*H][ceil(Tk/64)][2][ceil(Tq/128)*128], one word = the 32 keys a forward lane holds of one
   * 64-key tile. */
  float dropout_p;
  uint64_t dropout_seed;
  void* drop_mask;
  int64_t drop_mask_bytes;
  /* Optional fp32 scratch (16-byte aligned, tmi_attn_workspace_bytes(B, H, Tq) bytes; NULL = none).  With it, the forward
   * and dQ passes of a short query side against a long key side (one query tile, >= 8 key tiles of 64, mask_mode 0: the
   * cross-attention of W:255-301) split the keys over 2-8 workgroups per (batch, head) and fold the partials in a second
   * launch: the lone 100-row query tile is otherwise one latency chain of 24 key tiles on 96 of 256 CUs. */
  void* workspace;
  int64_t workspace_bytes;
  /* tmi_attn_bwd only.  0 or 3: both passes (ONE launch when Tq <= 256 and Tk <= 256 - the dK/dV workgroups then compute the
   * row sums themselves, in the dQ pass's order: the outputs and `delta` are bit for bit those of the two launches); 1: the dQ
   * pass alone (it also fills `delta`); 2: the dK/dV pass alone, after a dQ pass of the same descriptor has filled `delta`.  Lets a caller put the dK/dV pass of a cross-attention - whose results
   * nothing on the decoder's backward chain waits for (W:255-301: dK, dV feed the shared k/v projections) - on another stream. */
  int32_t bwd_passes;
} tmi_attn_desc;
int64_t tmi_attn_workspace_bytes(int64_t B, int64_t H, int64_t Tq);
int64_t tmi_attn_dropmask_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk);
int tmi_attn_fwd(const tmi_attn_desc* d, void* stream);
int tmi_attn_bwd(const tmi_attn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Decoder token embedding + positional encoding (W:405-408) with the teacher-forcing
 * shift of W:559-563 folded in: id(b,t) = t == 0 ? start_id : labels[b, t-1].
 *   out[b,t,:] = table[id(b,t), :] + pe[t, :]
 * Backward (gradient of tf.gather, densified): for every distinct id, dtable[id,:] =
 * sum over its positions of dy, summed in position order (deterministic, no atomics).
 * dtable must be zero on entry for rows that are not touched.
 */
int tmi_embed_fwd(const int32_t* labels, const float* table, const float* pe, void* out,
                  int64_t B, int64_t S, int64_t D, int32_t start_id, int32_t dtype, void* stream);
int tmi_embed_bwd(const int32_t* labels, const void* dy, float* dtable, int64_t B, int64_t S,
                  int64_t D, int32_t start_id, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Shifted sparse softmax cross-entropy over logits[B*S, ld] (W:585-600):
 * row (b,t) with t < S-1 is scored against labels[b, t+1]; row t = S-1 is unused.
 *   row_loss[b*S+t] = logsumexp(logits) - logits[target]      (0 for unused rows)
 *   logits <- dlogits = (softmax - onehot) * grad_scale       (0 for unused rows and for
 *                                                              the pad columns [V, ld))
 * grad_scale is 1 / (B*(S-1)) for the reference's reduce_mean.  tmi_mean_rows folds
 * row_loss into loss[0] = sum(row_loss) * scale.
 */
int tmi_xent_fwd_bwd(void* logits, int64_t ld, const int32_t* labels, float* row_loss,
                     int64_t B, int64_t S, int64_t V, float grad_scale, int32_t dtype, void* stream);
/* The same cross-entropy for logits that are the output of an LM head (W:579-600: logits = decoder_out . lm_head): x[B*S, d]
 * (row stride x_ld) and w (element (k, n) at w[k*w_sk + n*w_sn]) are the operands tmi_gemm produced `logits` from.  bf16
 * logits have lost the low bits of the target logit (ulp 0.03 at |z| in [4, 8)) and the loss is lse - z_target, so the row's
 * workgroup recomputes z_target = x[row, :] . w[:, target] in fp32 for row_loss; lse and the gradient come from `logits` as
 * in tmi_xent_fwd_bwd.  On the headline golden this takes the bf16 loss-curve error from 7.7e-4 to 2.0e-4
 * (profiles/r04_bf16_margin.txt).  fp32 logits: identical to tmi_xent_fwd_bwd (x, w are validated and ignored). */
int tmi_linear_xent(const void* x, int64_t x_ld, const void* w, int64_t w_sk, int64_t w_sn, int64_t d, void* logits, int64_t ld,
                    const int32_t* labels, float* row_loss, int64_t B, int64_t S, int64_t V, float grad_scale, int32_t dtype,
                    void* stream);
int tmi_sum_scale(const float* x, float* out, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-tensor Adam over a flat fp32 arena (tf.keras.optimizers.Adam, W:901; V:1271-1275).
 *   g' = g * gscale                      (gscale: 1 for Whisper's SUM, 1/N for V:1231)
 *   m <- b1 m + (1-b1) g' ; v <- b2 v + (1-b2) g'^2
 *   eps_mode 0 (TF/Keras-V2): p <- p - lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
 *   eps_mode 1 (torch):       p <- p - lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
 *   weight_decay (decoupled, AdamW) defaults to 0 in the reference.
 * bf16_mirror (optional, may be NULL): bf16 copy of the updated parameters, same flat
 * indexing — the k-contiguous / natural-layout weight operand of the bf16 GEMMs.
 * zero_grad != 0: g is overwritten with zeros once consumed (the next backward accumulates into it; saves the
 * separate fill pass).  max_blocks > 0 caps the grid: a slice of the arena updated beside other kernels (the
 * optimizer running bucket by bucket under backward) must leave them the CUs.
 * Algorithmic traffic: 28 B/param (read p,g,m,v; write p,m,v) + 2 B/param for the mirror (+ 4 with zero_grad).
 */
int tmi_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, int32_t eps_mode,
                  float weight_decay, float gscale, void* bf16_mirror, int32_t zero_grad,
                  int32_t max_blocks, void* stream);
/* The same update over an embedding table [nrows][row_len] (tf.keras.layers.Embedding, W:382), skipping idle rows:
 * a row whose gradient is all zero and that has never been updated (active[r] == 0: m = v = 0) has an exactly-zero
 * Adam update, so only its gradient is read.  active: one byte per row, zero-initialised by the caller and kept
 * across steps (set by the kernel the first time a row sees a gradient).  weight_decay must be 0. */
int tmi_adam_step_rows(float* p, float* g, float* m, float* v, int64_t nrows, int64_t row_len,
                       unsigned char* active, float lr, float beta1, float beta2, float eps,
                       int32_t step, int32_t eps_mode, float weight_decay, float gscale,
                       void* bf16_mirror, int32_t zero_grad, void* stream);
/* tmi_adam_step over the reference's variables (chunks: int64 device array [nchunks][3] = (first element, end element,
 * variable index) tiling the arena in pieces that never cross a variable; sumsq indexed by variable) with the gradient clipping of V:1243 (tf.clip_by_global_norm, before the update) and V:1274 (Keras
 * clipnorm = tf.clip_by_norm per variable) applied as a per-variable factor of g, computed from ONE
 * tmi_segment_sumsq pass over the raw gradients (sumsq[nseg]):
 *   c_g = clip_global / max(||g||, clip_global),  c_s = clip_each / max(c_g * ||g_s||, clip_each),  g' = g * gscale * c_g * c_s
 * (clip_global == 0 / clip_each == 0 switch a stage off).  No clipped copy of the gradients is written. */
int tmi_adam_step_segments(float* p, float* g, float* m, float* v, const int64_t* chunks,
                           int64_t nchunks, const float* sumsq, int64_t nseg, float clip_global, float clip_each, float lr,
                           float beta1, float beta2, float eps, int32_t step, int32_t eps_mode,
                           float weight_decay, float gscale, void* bf16_mirror, int32_t zero_grad,
                           int32_t max_blocks, void* stream);
/* (max_blocks > 0 caps the grid, as in tmi_adam_step: for a slice of the table that runs beside other work.  The global
 * norm always comes from the whole sumsq[nseg], so a sub-table of chunks is the same update restricted to its rows.) */
/* The step-dependent scalars of tmi_adam_step, computed on the host exactly as it does:
 * out3 = {step_size, vcorr_inv_sqrt, 1 - lr*weight_decay}. */
int tmi_adam_scalars(float lr, float beta1, float beta2, int32_t step, int32_t eps_mode,
                     float weight_decay, float* out3);
/* tmi_adam_step with those three scalars read from DEVICE memory (dev_scalars[3]): nothing in the
 * launch changes from step to step, so it can be part of a captured HIP graph. */
int tmi_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                      float beta2, float eps, const float* dev_scalars, int32_t eps_mode,
                      float gscale, void* bf16_mirror, void* stream);


/* Gradient exchange staging (C1, W:834 / V:1246: the implicit cross-replica SUM of every gradient; the
 * exchange itself is RCCL through torch.distributed, these are the device-side passes around it).
 * tmi_grad_pack:   dst[i] = bf16(src[i] * scale) - an fp32 arena slice onto a bf16 wire buffer.
 * tmi_grad_unpack: dst[i] = scale * sum_{p < nparts} src[p*part_stride + i], src bf16 or fp32, fp32 sum:
 *   the widening of a reduced bf16 bucket (nparts = 1) and the local reduction of the nparts pieces a mesh
 *   reduce-scatter (all-to-all over the xGMI links) delivers. */
int tmi_grad_pack(const float* src, void* dst, int64_t n, float scale, void* stream);
int tmi_grad_unpack(const void* src, int32_t src_dtype, int64_t nparts, int64_t part_stride, float* dst,
                    int64_t n, float scale, void* stream);

/* bf16 shadows of fp32 master weights: dst[r*ldd + c] = bf16(src[r*lds + c]) and the
 * transposed form dst[c*ldd + r] = bf16(src[r*lds + c]); pad columns [cols, ldd) of the
 * plain form are written as zero. */
int tmi_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows,
                  int64_t cols, void* stream);
int tmi_transpose_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd,
                            int64_t rows, int64_t cols, void* stream);

/* features [B, C, T] fp32 (W:792 layout) -> channels-last, time-padded
 * out[b, pad_left + t, c] of shape [B, T + pad_left + pad_right, C] (pad rows zeroed):
 * the tf.transpose of W:329 fused with the "same" padding of W:311. */
int tmi_feat_to_channels_last(const float* feats, void* out, int64_t B, int64_t C, int64_t T,
                              int64_t pad_left, int64_t pad_right, int32_t dtype, void* stream);

/* sum of squares of n fp32 values into out[0] (+= if accumulate): tf.clip_by_global_norm
 * (V:1243) first stage. */
int tmi_sumsq(const float* x, float* out, int64_t n, int32_t accumulate, void* stream);

/* ====================================================================================
 * Wav2Vec2 pre-training path (speech_jobs/wav2vec2_dist.py, "V:")
 * ==================================================================================== */

/* GroupNormalization (V:140-196; contiguous channel groups, statistics over (time, C/G) per
 * (batch, group), biased variance, per-channel affine) fused with the exact-erf GELU that
 * follows it in every conv layer of the feature encoder (V:283-288):
 *   y[b,t,c] = gelu(gamma[c] * (x - mean[b,g]) * rstd[b,g] + beta[c])
 * x is [B][T][C] with batch stride x_sb, y likewise with y_sb (so y may be the padded input
 * buffer of the next Conv1D).  stats: fp32 [B,G,2] = (mean, rstd), saved for backward.
 * part: fp32 workspace [B * tmi_groupnorm_chunks(T) * G * 2].
 * Backward: dx = d/dx of the above given dy (GELU' recomputed from x and stats);
 * dgamma/dbeta are ACCUMULATED (atomics; caller zeroes); sums: fp32 workspace [B,G,2]. */
int64_t tmi_groupnorm_chunks(int64_t T);
int tmi_groupnorm_gelu_fwd(const void* x, int64_t x_sb, const float* gamma, const float* beta,
                           void* y, int64_t y_sb, float* stats, float* part, int64_t B, int64_t T,
                           int64_t C, int64_t G, float eps, int32_t dtype, void* stream);
int tmi_groupnorm_gelu_bwd(const void* x, int64_t x_sb, const void* dy, int64_t dy_sb,
                           const float* gamma, const float* beta, const float* stats, void* dx,
                           int64_t dx_sb, float* dgamma, float* dbeta, float* part, float* sums,
                           int64_t B, int64_t T, int64_t C, int64_t G, int32_t dtype, void* stream);

/* Layout packs that turn the grouped positional Conv1D (V:271-277: k = 128, groups = 16,
 * "same") into ONE batched tmi_gemm over overlapping rows:
 *   tmi_group_pack:   x [B*T][C] -> xg [G][B*Tp][C/G], batch b's rows at [b*Tp + pad_left, +T),
 *                     every other row zero (the "same" padding, and zero rows for the gradient);
 *   tmi_group_unpack: out[b*T+t][g*Cg+j] = yg[g][b*Tp + t + row_off][j] (+ bias[c]) (+ resid);
 *   tmi_posconv_pack_weights: Keras kernel w [k][C/G][C] fp32 -> forward operand
 *                     wf[g][(kk,i)][o] = w[kk][i][g*Cg+o] and backward operand
 *                     wb[g][(kk',o)][i] = w[k-1-kk'][i][g*Cg+o] (both [G][k*Cg][Cg], dtype). */
int tmi_group_pack(const void* x, void* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t Tp,
                   int64_t pad_left, int32_t dtype, void* stream);
int tmi_group_unpack(const void* yg, const float* bias, const void* resid, void* out, int64_t B,
                     int64_t T, int64_t C, int64_t G, int64_t Tp, int64_t row_off, int32_t dtype,
                     void* stream);
int tmi_posconv_pack_weights(const float* w, void* wf, void* wb, int64_t k, int64_t Cg, int64_t G,
                             int32_t dtype, void* stream);

/* Hard vector quantiser (V:581-667): per row and group, squared L2 to every code computed as
 * sum((h-c)^2) in fp32, tf.argmin (first index on ties), q = the chosen code;
 * perplexity[0] = mean_g exp(-sum_c p log(p+1e-10)), p = clip(mean one-hot, 1e-10, 1).
 * h, q: [rows][G*gd] (dtype); codebook fp32 [G][Nc][gd]; idx int32 [rows][G].
 * tmi_vq_bwd: dcodebook[g][idx][:] += dq (the only gradient path of the quantiser, V:638). */
int tmi_vq_nearest(const void* h, const float* codebook, int32_t* idx, void* q, float* perplexity,
                   int64_t rows, int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream);
/* The same quantiser with the code choice given (idx is an INPUT, clamped to [0, Nc)): q = the chosen codes, the same
 * perplexity.  For replaying a recorded code sequence (teacher-forced parity runs of the bf16 path). */
int tmi_vq_assign(const float* codebook, const int32_t* idx, void* q, float* perplexity, int64_t rows,
                  int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream);
int tmi_vq_bwd(const int32_t* idx, const void* dq, float* dcodebook, int64_t rows, int64_t G,
               int64_t Nc, int64_t gd, int32_t dtype, void* stream);

/* First feature-encoder layer as a filter bank (V:283-288, i = 0: Conv1D(C, kernel 10, stride 5, "same", no bias) on
 * the single-channel audio, then GroupNorm (V:140-196) and exact GELU), never materialising the conv output: both
 * passes of the forward and both passes of the backward recompute it from the raw audio with the taps in registers.
 *   audio fp32 [B][Tin] (batch stride a_sb); w fp32 [k][C] (the Keras kernel [k, 1, C]); T = ceil(Tin / stride) output
 *   steps, pad_left = the "same" padding on the left; y / dy [B][T][C] of `dtype` with their own batch strides.
 *   fwd: stats[B][G][2] = (mean, rstd) of u = conv(audio), y = gelu(gamma * xhat + beta).
 *   bwd: dW[k][C] += d/dw, dgamma[C] += .., dbeta[C] += ..  (no input gradient: the input is data).
 *   part: fp32 scratch of tmi_fir_chunks(T)*B*G*2 floats; sums: [B][G][2]; wpart: tmi_fir_gn_workspace_floats(). */
int64_t tmi_fir_chunks(int64_t T);
int64_t tmi_fir_gn_workspace_floats(int64_t B, int64_t T, int64_t C);
int tmi_fir_groupnorm_gelu_fwd(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                               int64_t k, int64_t stride, const float* gamma, const float* beta, void* y,
                               int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C,
                               int64_t G, float eps, int32_t dtype, void* stream);
int tmi_fir_groupnorm_gelu_bwd(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                               int64_t k, int64_t stride, const void* dy, int64_t dy_sb, const float* gamma,
                               const float* beta, const float* stats, float* dW, float* dgamma, float* dbeta,
                               float* part, float* sums, float* wpart, int64_t B, int64_t T, int64_t C,
                               int64_t G, int32_t dtype, void* stream);

/* Contrastive loss (V:866-899; whisper_single.py:745-787) on S [B][T][T] fp32 = all-pairs <h_t, q_t'>
 * (a tmi_gemm): row (b,t) has logits [S[t][t], S[t][idx[0..Nn)]] / temperature and label 0, where
 * idx = neg + b*neg_sb + t*neg_st (element strides): V:908-937 draws one row of indices per batch row
 * (neg [B][Nn]: neg_sb = Nn, neg_st = 0); whisper_single.py:789-839 one row per time step, shared by
 * the batch (neg [T][Nn]: neg_sb = 0, neg_st = Nn).  Indices may repeat and may equal t.
 * row_loss[b*T+t] = logsumexp - logit_0; S is REPLACED by d(sum row_loss)/dS * grad_scale. */
int tmi_contrastive_fwd_bwd(float* S, const int32_t* neg, int64_t neg_sb, int64_t neg_st,
                            float* row_loss, int64_t B, int64_t T, int64_t Nn, float temperature,
                            float grad_scale, void* stream);

/* Gradient clipping.  seg_off: int64 device array [nseg+1] of element offsets into g.
 * tmi_segment_sumsq: out[s] = sum g[seg_off[s]:seg_off[s+1]]^2.
 * tmi_segment_clip:  g[seg s] *= clip / max(sqrt(sumsq[s]), clip).
 * One segment covering the arena = tf.clip_by_global_norm (V:1243); one segment per variable
 * = Keras clipnorm (V:1274). */
int tmi_segment_sumsq(const float* g, const int64_t* seg_off, float* out, int64_t nseg, void* stream);
/* The same sums over the chunk table of tmi_adam_step_segments ([nchunks][3] = lo, hi, segment index): equal pieces of work
 * instead of one workgroup row per segment.  out[nseg] is zeroed first. */
int tmi_segment_sumsq_chunks(const float* g, const int64_t* chunks, int64_t nchunks, float* out, int64_t nseg, void* stream);
int tmi_segment_clip(float* g, const int64_t* seg_off, const float* sumsq, int64_t nseg, float clip,
                     void* stream);

/* out[0] = nan_to_zero(a[0] + w * b[0]) * scale  (V:1220-1231: contrastive + 0.1 * (-perplexity),
 * NaN guard, / num_replicas) without a host round trip. */
int tmi_loss_combine(const float* a, const float* b, float w, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Log-mel front end (speech_jobs/whisper_dist.py:739-766, extract_fbank_features; dead in the
 * reference's training path).  The windowed DFT of the frames is a tmi_gemm
 * (frames = overlapping rows of the waveform, a_sm = hop; B = [n_fft, 2*n_bins] Hann-weighted
 * cos | -sin); this entry turns spec[frames, ld_spec] = [re | im] into
 * log(|X|^2 . mel[n_bins, n_mels] + eps), written frame-major out[f*ld_out + m] (the reference's
 * layout, W:764) or channels-first out[m*ld_out + f] (what the encoder reads).  fp32.
 */
int tmi_logmel_from_spectrum(const float* spec, int64_t ld_spec, const float* mel, float* out,
                             int64_t frames, int32_t n_bins, int32_t n_mels, float eps,
                             int32_t channels_first, int64_t ld_out, void* stream);

/* Diagnostics only: copies the five phase counters of the stamped GEMM build (TMI_GEMM_DBG=8: cycles
 * spent issuing the LDS-DMA, in MFMA + fragment reads, in the vmcnt wait, in the barrier, and the
 * iteration count, from one workgroup) to host memory.  Host-synchronous; not part of the data path. */
int tmi_debug_gemm_stamps(unsigned long long* out5);

#ifdef __cplusplus
}
#endif
#endif /* TETHYS_MI_H */
