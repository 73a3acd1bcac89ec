/* p5hip.h -- C ABI of libp5hip.so: the MI355X-native T5 training + constrained-beam-search path that
 * replaces OpenP5's `P5_T5` model object (the seam is the Python `model` handed to the runner,
 * /root/reference/src/src_t5/main.py:184-206).
 *
 * The reference has no FFI of its own (pure Python over torch + HF transformers).  The entry points below are
 * therefore what a binding for this path needs, one per reference call it replaces:
 *
 *   p5_forward            <- P5_T5.forward(input_ids, whole_word_ids, attention_mask, labels) -> per-token NLL
 *                            (model/P5_T5.py:275-386, called at runner/DistributedRunner.py:63-70)
 *   p5_forward_loss       <- the same + the runner's masked-mean loss       (DistributedRunner.py:72-77)
 *   p5_backward           <- loss.backward()                               (DistributedRunner.py:80)
 *   p5_grad_sumsq +
 *   p5_adamw_step         <- clip_grad_norm_ + AdamW.step + scheduler      (DistributedRunner.py:81,85-86;
 *                                                                           SingleRunner.py:191-217)
 *   p5_generate           <- P5_T5.generate(..., prefix_allowed_tokens_fn, num_beams)  (DistributedRunner.py:361-371)
 *   p5_param_table        <- state_dict()/load_state_dict() key layout     (utils/utils.py:119-129)
 *
 * plus per-kernel entry points (p5_op_*) used by the parity tests.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (torch's allocator); nothing is allocated,
 * freed or synchronised inside (graph-capturable); `stream` is a hipStream_t passed as void*; return value
 * 0 = ok, negative = error (message via p5_last_error()).  dtype: 0 = fp32 parity mode, 1 = bf16 fast mode.
 */
#ifndef P5HIP_H
#define P5HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct P5Config {
  int vocab_size, d_model, d_kv, d_ff, n_enc_layers, n_dec_layers, n_heads;
  int rel_buckets, rel_max_distance, whole_word_size;
  int gated_gelu;      /* 0: ReLU FFN (t5-small/base/large), 1: gated gelu_new (v1.1 / Flan) */
  int dtype;           /* 0 fp32, 1 bf16 */
  float eps, dropout;
  int pad_id, eos_id;
} P5Config;

typedef struct P5Engine P5Engine;

const char* p5_last_error(void);
/* process-wide tuning knobs (tests / benchmarks): "gemm_tile" = 0|64|128|256, "gemm_v2" = 0|2|3|4 (hand-pipelined loop for forced
 * 128x128 tiles), "gemm_ring", "gemm_small_ring", "gemm_ksdma", "gemm_xcd_rect", "decode_fused" = 0|1 */
int p5_set_option(const char* name, int value);
int p5_abi_version(void);
/* In-run kernel profiler (measurement aid, bench.py): between p5_profile_begin() and p5_profile_end() every kernel launch of the library is
 * bracketed by two HIP events on its stream; p5_profile_end synchronises and writes a JSON array of {"kernel" (name + launch grid), "launches",
 * "total_us", "flops" (algorithmic FLOPs of the GEMM / attention launches, 0 elsewhere)} into `report` (NUL-terminated, `cap` bytes).
 * Durations include the dispatch gap of each launch (~1-2 us), i.e. they are upper bounds of the rocprofv3 kernel durations. */
int p5_profile_begin(void);
int p5_profile_end(char* report, int cap);
int p5_is_emulator(void);   /* 1 only for the test-only host emulation build under tests/emu */

/* ---- engine lifetime + parameter arena layout ---- */
int p5_engine_create(const P5Config* cfg, P5Engine** out);
int p5_engine_destroy(P5Engine* e);
int64_t p5_param_count(const P5Engine* e);
/* idx-th tensor of the arena in HF state-dict naming; returns 0, or 1 when idx is past the end */
int p5_param_table(const P5Engine* e, int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols);
/* params/grads: fp32 arenas of p5_param_count elements; shadow: bf16 arena (dtype=1) or NULL;
 * lut_enc/lut_dec: int32 [2*lut_half+1] bucket of rel=key-query (bidirectional / unidirectional);
 * rng_state: uint32[2] {seed, step} */
int p5_engine_bind(P5Engine* e, float* params, float* grads, void* shadow, const int* lut_enc, const int* lut_dec,
                   int lut_half, uint32_t* rng_state);
int p5_refresh_shadow(P5Engine* e, void* stream);
/* Optional (bf16 mode): a caller-owned buffer of p5_transposed_bytes(e) bytes that holds W^T of every 2-D layer weight at the
 * same arena offset.  When bound, the data gradients dx = dy W (the "autograd of nn.Linear" half of loss.backward(),
 * DistributedRunner.py:80) read W^T as a K-contiguous operand and run on the forward GEMM kernel instead of the
 * K-strided-operand variant.  p5_refresh_transposed after every parameter update (it runs on the side stream when one is bound
 * and the next backward waits for it). */
int64_t p5_transposed_bytes(const P5Engine* e);
int p5_engine_bind_transposed(P5Engine* e, void* buf, void* stream);
int p5_refresh_transposed(P5Engine* e, void* stream);
/* The next backward ADDS to what the gradient arena holds instead of starting a new sum: the 2nd.. micro-batch of a gradient-
 * accumulation group, or an arena the caller has just zero-filled itself on the stream that backward will use.  One-shot.
 * Without it a backward starts a new sum: it stores every Linear gradient and clears the atomically accumulated ones itself
 * (p5_engine_discard_grads), or clears the whole arena first where the storing path does not apply (fp32 engine). */
int p5_engine_grads_zeroed(P5Engine* e);
/* optimizer.zero_grad() (DistributedRunner.py:93) done by the engine: the fill is issued on the side stream when one is bound,
 * ordered after everything `stream` holds so far (the optimizer step that read the gradients), so that it overlaps the next
 * forward; the next backward waits for it.  Nothing else may read the gradient arena before that backward. */
int p5_engine_clear_grads(P5Engine* e, void* stream);
/* optimizer.zero_grad(set_to_none=True) (DistributedRunner.py:93): the gradients are dead until the next backward; no device work.
   (p5_engine_clear_grads is the eager, set_to_none=False form.) */
int p5_engine_discard_grads(P5Engine* e);
/* optional second stream: weight-gradient GEMMs run on it, one sub-layer behind the dgrad chain (NULL = single stream) */
int p5_engine_set_side_stream(P5Engine* e, void* side_stream);

/* ---- training step pieces ---- */
int64_t p5_train_workspace_bytes(const P5Engine* e, int B, int L, int T);
/* nll_out: fp32 [B*T]; keeps activations in ws for p5_backward */
int p5_forward(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
               const int64_t* labels, int B, int L, int T, int training, float* nll_out, void* ws, int64_t ws_bytes,
               void* stream);
/* p5_forward + the runner's masked-mean loss (DistributedRunner.py:72-77) in one call: loss_out[0] = mean_b(sum_t nll*m / max(sum_t m, 1)),
 * m = (output_attention != 0), computed behind the cross-entropy kernel.  A following p5_backward / p5_backward_stage with
 * dnll == NULL back-propagates d(loss) = 1 (the CE backward derives the per-token weights from the mask itself). */
int p5_forward_loss(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
                    const int64_t* labels, const int64_t* output_attention /* [B,T] */, int B, int L, int T, int training,
                    float* nll_out /* [B*T] */, float* loss_out /* [1] */, void* ws, int64_t ws_bytes, void* stream);
/* stages: 0 = head + decoder-final, 1..n_dec = decoder layers (top first), then decoder embedding,
 * then encoder-final + encoder layers (top first), last = encoder embedding.  p5_backward runs them all. */
int p5_backward_num_stages(const P5Engine* e);
int p5_backward_stage(P5Engine* e, const float* dnll, int stage, void* stream);
int p5_backward(P5Engine* e, const float* dnll, void* stream);
/* LAYOUT ONLY: the arena range [begin,end) of the parameters whose gradients `stage` computes.  It does NOT say when they are final:
 * with the default two-layer weight-gradient groups a stage's gradients may be written by a LATER stage's grouped launch.  A data-parallel
 * caller exchanges the range reported by p5_backward_final_range after each stage (below), never this one. */
int p5_backward_stage_range(const P5Engine* e, int stage, int64_t* begin, int64_t* end);
/* What a data-parallel caller exchanges after each p5_backward_stage call: the gradient range that became FINAL with that call -- the
 * union of the stage ranges whose kernels have all been launched (empty while a two-layer weight-gradient group of the encoder is still
 * filling up: the staged backward issues the same grouped launches as p5_backward; p5_backward_stage_pairs(e, 0) restores one launch per
 * layer and one range per stage).  Ranges are contiguous and walk the arena from the back. */
int p5_backward_final_range(const P5Engine* e, int64_t* begin, int64_t* end);
int p5_backward_stage_pairs(P5Engine* e, int on);
/* The staged backward as ONE call: all stages are enqueued on `stream`; ranges[2k], ranges[2k+1] = the k-th gradient range that became final
 * (arena offsets, in completion order), *n_ranges their number (<= max_ranges, <= p5_backward_num_stages).  Behind each one an event is
 * recorded on `stream` (and, when the engine has a side stream, a second one there: weight gradients of the range may have been launched
 * on it): p5_backward_staged_wait(e, k, comm_stream) makes `comm_stream` wait for both (hipStreamWaitEvent, no host wait), after
 * which the caller enqueues the exchange of range k there -- DDP's bucketed all-reduce overlapped with the rest of the backward
 * (/root/reference/src/src_t5/main.py:158-160 wraps the model in DDP) without one host round trip per stage. */
int p5_backward_staged(P5Engine* e, const float* dnll, void* stream, int64_t* ranges, int max_ranges, int* n_ranges);
int p5_backward_staged_wait(P5Engine* e, int k, void* comm_stream);
/* The exchange for a host without torch.distributed ("an p5_allreduce_* shim over RCCL taking an ncclComm_t created once per
 * process"; replaces what DDP's reducer would do at /root/reference/src/src_t5/runner/DistributedRunner.py:26).  `nccl_comm` is the
 * caller's ncclComm_t; the library has no link-time RCCL dependency and calls the ncclAllReduce already loaded in the process (the one
 * that created the communicator), else librccl.so.1.
 *   p5_allreduce_range(e, k, comm, bf16_scratch, comm_stream): waits on `comm_stream` for range k of the last p5_backward_staged (both of
 *     its events when weight gradients ran on the engine's side stream) and all-reduces (SUM) that slice of the gradient arena in place
 *     there; bf16_scratch != NULL (room for the range's elements as bf16): cast -> all-reduce in bf16 -> widen back (half the xGMI bytes).
 *     The mean over ranks is NOT taken here: pass grad_scale = 1 / world to p5_adamw_step.
 *   p5_allreduce_sum(buf, count, dtype, comm, stream): any device buffer in place; dtype 0 = f32, 1 = bf16, 2 = f64, 3 = i64 (the metric
 *     sums of DistributedRunner.py:389-395). */
int p5_allreduce_range(P5Engine* e, int k, void* nccl_comm, void* bf16_scratch, void* comm_stream);
int p5_allreduce_sum(void* buf, int64_t count, int dtype, void* nccl_comm, void* stream);

/* out_partials: float[1024], fully overwritten; p5_adamw_step sums them in a fixed order (bit-identical on every rank) */
int p5_grad_sumsq(const float* grads, int64_t n, float* out_partials, void* stream);
int p5_adamw_step(float* params, const float* grads, float* m, float* v, void* shadow_bf16, int64_t n,
                  const float* sumsq /* float[1024] from p5_grad_sumsq, or NULL = no clipping */, double max_norm, double grad_scale, double lr, double beta1,
                  double beta2, double eps, double weight_decay /* doubles, as the reference's Python floats: derived scalars are rounded once */, int step_t,
                  void* stream);
/* The same step over the ENGINE's bound arenas (params, grads, bf16 shadow).  With the transposed / norm-folded copies bound
 * (p5_engine_bind_transposed, bf16 training) the update ALSO writes W^T and W diag(ln) -- the 2-D layer weights are updated in 64 x 64
 * tiles that are transposed through LDS, a projection behind a T5LayerNorm is multiplied by the norm weight's NEW value -- so the
 * caller skips p5_refresh_transposed (*copies_fresh = 1; 0 when the flat path ran: fp32 engine, no copy bound, side stream, option
 * "adam_tiles" 0).  Per element the arithmetic is p5_adamw_step's: parameters, moments and all three copies are bit-identical. */
int p5_engine_adamw_step(P5Engine* e, float* m, float* v, const float* sumsq, double max_norm, double grad_scale, double lr, double beta1,
                         double beta2, double eps, double weight_decay, int step_t, int* copies_fresh, void* stream);

/* ---- generation ---- */
/* Optional: a caller-owned buffer of p5_decode_fold_count(e) elements of the compute dtype.  When bound, p5_generate folds every
 * decoder RMSNorm but the first into the GEMMs around it (norm weight multiplied into the consuming projection, row statistic
 * carried between GEMM epilogues): 18 of 73 launches per decode step fewer for T5-small.  Call p5_refresh_decode_fold after
 * the parameters change (and after p5_refresh_shadow / an optimizer step). */
int64_t p5_decode_fold_count(const P5Engine* e);
int p5_engine_bind_decode_fold(P5Engine* e, void* buf);
int p5_refresh_decode_fold(P5Engine* e, void* stream);
int64_t p5_generate_workspace_bytes(const P5Engine* e, int B, int L, int K, int max_len, int max_children, int excluded_words);
/* trie in CSR: child_off[n_nodes+1], child_tok/child_node[n_edges]; node 0 = empty prefix.
 * out_seq int32 [B,K,max_len] (pad-filled, starts with pad=decoder start), out_score fp32 [B,K], out_len int32 [B,K].
 * excluded_nodes: optional uint32 bitmap [B, excluded_words] over trie node ids; bit n of row b set = node n does not
 * exist in item b's trie (the per-user history exclusion of the filtered protocol, DistributedRunner.py:286-297,
 * without building one trie per user).  NULL / 0 = nothing excluded.
 * Replaces P5_T5.generate(...) = HF beam search + PrefixConstrainedLogitsProcessor (DistributedRunner.py:361-371).
 * Enqueues the whole search and returns WITHOUT synchronising: HF's stop test is taken on the device, so no step reads
 * anything back.  max_len bounds the number of decode steps enqueued (max_len - 1): pass min(max_length, depth of the trie).
 * Limits: 1 <= K <= 64 beams, 2 <= max_len <= 128; the trie may be any DAG in this CSR form (an appended trie, generation_trie.py:19-21,
 * is grafted by the caller -- openp5_amd/trie.py::CompiledTrie.from_trie). */
int p5_generate(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
                int B, int L, int K, int max_len, const int* child_off, const int* child_tok, const int* child_node,
                const int* roots /* [B] empty-prefix node per batch item, or NULL = node 0 */,
                const uint32_t* excluded_nodes, int excluded_words, int max_children, int* out_seq, float* out_score, int* out_len, void* ws, int64_t ws_bytes, void* stream);
/* Forced-prefix fast-forward (openp5_amd/csrc/p5_decode.h): when the first n tokens after the decoder start token are the same for EVERY
 * item of the trie (OpenP5 item ids all start with "<dataset> item _"), the first n beam-search steps have one allowed token each and
 * are computed as ONE teacher-forced decoder pass over n positions per user instead of n decode steps -- same numbers, n - 1 steps saved.
 * tokens[i] / nodes[i]: the i-th forced token and the trie node it leads to (HOST arrays, n <= 16).  One-shot: applies to the next
 * p5_decode_begin / p5_generate / p5_generate_draft on this engine; ignored with per-item roots, n < 2, or option "gen_ff" = 0.
 * The caller guarantees that the chain is really forced for every item of the batch (no excluded node on it). */
int p5_generate_set_forced_prefix(P5Engine* e, const int* tokens, const int* nodes, int n);
/* ---- verified generation: the bf16 search proposes, an fp32 pass decides (openp5_amd/csrc/p5_verify.h) ----
 * The reference ranks by the fp32 scores of HF beam search (DistributedRunner.py:361-387, utils/evaluate.py:37-58).  Protocol, two engines
 * over the SAME master parameter arena (a bf16 one for the draft, an fp32 one -- dtype 0 -- for the verification), one stream:
 *   p5_verify_begin   (fp32 engine)   -> lays out the verification workspace for this batch shape (host only)
 *   p5_verify_encode  (fp32 engine)   -> fp32 encoder pass + cross-attention K/V; p5_verify_encoder_output() = its fp32 [B*L, d_model]
 *   p5_generate_set_encoder_output + p5_generate_draft (bf16 engine, beam width Kw = K + a few): the draft starts from THAT encoder
 *                                        output (rounded once) instead of running its own encoder; its results are ignored, `hist` = what
 *                                        the search kept alive at every step
 *   p5_verify_plan    (fp32 engine)   -> the distinct live prefixes of every user ("rows"); p5_verify_plan_header()[0] = the largest
 *                                        row count of any user -- the ONE number the host reads
 *   p5_verify_run     (fp32 engine, rows_per_user >= that number, multiple of 16 recommended)
 *                                     -> one teacher-forced fp32 decoder pass over all rows, full-vocabulary log-sum-exp and the trie
 *                                        children's log-probabilities per row, then HF's beam search of the REAL width K replayed on
 *                                        those numbers.  out_* as p5_generate; out_missing int32 [B]: 1 = the replay needed a prefix the
 *                                        draft had dropped, or a value of the pass left the range of the split products -- that user's
 *                                        result is NOT the fp32 search's and the caller must re-run the user through p5_generate on the
 *                                        fp32 engine (openp5_amd/model.py does).
 * A returned, unflagged list is the fp32 search's list: no bf16 number takes part in any decision or score.  The GEMMs of the fp32 passes
 * multiply on the f16 matrix cores from a two-term fp16 split of every fp32 operand (x = hi + lo / 4096: 22 mantissa bits, the lo x lo
 * term dropped, ~2^-22 relative per product, valid for |x| < 2^15 -- csrc/p5_gemm.h; option "verify_split" 0 = exact fp32 MFMAs).  A user
 * any of whose final hidden rows is non-finite or outside that range (an operand of the pass overflowed the split) is flagged through
 * out_missing exactly like a user with a missing prefix.  Limits: K <= 22, Kw <= 64, rows_per_user <= 512 and <= p5_verify_row_capacity.  A forced prefix set on the fp32 engine (p5_generate_set_forced_prefix) before p5_verify_begin
 * lets the replay skip the forced steps as the draft does. */
int64_t p5_generate_history_count(int B, int K, int max_len);      /* ints in `hist` for a draft of beam width K */
int p5_generate_draft(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
                      int B, int L, int K, int max_len, const int* child_off, const int* child_tok, const int* child_node, const int* roots,
                      const uint32_t* excluded_nodes, int excluded_words, int max_children, int* out_seq, float* out_score, int* out_len,
                      int* hist, void* ws, int64_t ws_bytes, void* stream);
int64_t p5_verify_workspace_bytes(const P5Engine* e, int B, int L, int K, int Kw, int max_len, int max_children, int excluded_words);
int p5_verify_begin(P5Engine* e, int B, int L, int K, int Kw, int max_len, const int* child_off, const int* child_tok, const int* child_node,
                    const int* roots, int max_children, int excluded_words, void* ws, int64_t ws_bytes);
int p5_verify_encode(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, void* stream);
const void* p5_verify_encoder_output(const P5Engine* e);     /* device fp32 [B*L, d_model], valid after p5_verify_encode until the next p5_verify_begin */
/* one-shot: the next p5_decode_begin / p5_generate / p5_generate_draft on `e` takes this fp32 encoder output [B*L, d_model] (cast to the
 * engine's dtype) instead of running its encoder */
int p5_generate_set_encoder_output(P5Engine* e, const float* enc_out_f32);
int p5_verify_plan(P5Engine* e, const int* hist, void* stream);
int p5_verify_row_capacity(int Kw, int max_len);           /* rows per user the workspace of a (Kw, max_len) verification holds (a multiple of 16) */
const int* p5_verify_plan_header(const P5Engine* e);     /* device int[4]: max rows per user, draft steps, total rows, overflow */
int p5_verify_run(P5Engine* e, int rows_per_user, const uint32_t* excluded_nodes, int* out_seq, float* out_score, int* out_len,
                  int* out_missing, void* stream);
/* Device-time brackets of p5_generate for benchmarks: p5_generate_timing(e, 1, NULL, NULL) arms it; after a p5_generate call,
 * p5_generate_timing(e, enable, &encode_ms, &decode_ms) WAITS for that call to finish and returns the time between its start and
 * its first decode step (encoder pass + cross-attention K/V projection + beam state) and the time of the decode loop itself. */
int p5_generate_timing(P5Engine* e, int enable, float* encode_ms, float* decode_ms);
/* The same search step by step (p5_generate = begin + (max_len - 1) x step + finish), for callers that interleave their own
 * work with the steps or want to stop early:
 *   p5_decode_begin   encoder, cross-attention K/V of every decoder layer, beam state (HF `_expand_inputs_for_generation`,
 *                     P5_T5.py:542-578, without physically repeating the encoder states num_beams times);
 *   p5_decode_step    one step: decoder over B*K rows with the KV cache, tied head, log-softmax over the full vocabulary,
 *                     trie mask, top-2K, BeamSearchScorer bookkeeping (HF generation/utils.py:3384-3483).  A no-op once the
 *                     search has stopped or max_len - 1 steps have run;
 *   p5_decode_done_flag  device pointer to an int that becomes 1 when the search has stopped (poll it asynchronously if
 *                     steps are enqueued one by one); NULL outside begin..finish;
 *   p5_decode_finish  writes the K best finished hypotheses per item (same outputs as p5_generate).
 * All arrays passed to p5_decode_begin must stay valid until p5_decode_finish; ws is p5_generate_workspace_bytes. */
int p5_decode_begin(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
                    int B, int L, int K, int max_len, const int* child_off, const int* child_tok, const int* child_node,
                    const int* roots, const uint32_t* excluded_nodes, int excluded_words, int max_children, void* ws, int64_t ws_bytes,
                    void* stream);
int p5_decode_step(P5Engine* e, void* stream);
const int* p5_decode_done_flag(const P5Engine* e);
int p5_decode_finish(P5Engine* e, int* out_seq, float* out_score, int* out_len, void* stream);
/* encoder only (JointEncoder.forward, P5_T5.py:74-204) */
int p5_encode(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask,
              int B, int L, void* enc_out /* T [B*L, d] */, void* ws, int64_t ws_bytes, void* stream);

/* ---- per-kernel entry points (parity tests) ---- */
int p5_op_gemm(int dtype, const void* A, const void* Bm, void* C, const void* aux, int M, int N, int K, int lda, int ldb,
               int ldc, int ldaux, int a_ks, int b_ks, int epi, int c_f32, int splitk, float alpha,
               const uint32_t* rng_state, uint32_t site, float drop_p, void* stream);
/* Persistent ring GEMM (openp5_amd/csrc/p5_gemm4.h), bf16 operands: up to 8 problems C[M,N] (+)= A B^T in ONE launch (the weight
 * gradients of a layer; nn.Linear autograd, DistributedRunner.py:80).  ks = 0: A [M, lda], B [N, ldb] (reduction dim contiguous);
 * ks = 1: A [K, lda], B [K, ldb] (reduction dim strided: dW = dy^T x).  epi as p5_op_gemm (0 store, 1 relu(+dropout), 2 residual +
 * dropout, 3 mask by aux > 0, 4 fp32 atomic add, 6 fp32 C += without split-K, 5 / 7 gated-GELU forward / backward, below).  tile_cfg 0 = 128x128, 1 = 256x128, 2 = 128x256, 3 = 256x128 loader / compute waves, 4 = 128x128 loader / compute waves.
 * rowss / ssq_out: optional T5LayerNorm statistics carried through the epilogue (row sum of squares in, sum of squares of the stored
 * row out), NULL = off.  `probs` is a HOST array. */
typedef struct P5GemmProblem {
  const void *A, *B; void* C; const void* aux;
  int M, N, K, lda, ldb, ldc, ldaux, epi, c_f32, splitk;
  float alpha;
  const float* rowss; float rowss_eps; float* ssq_out;
  int rowss_nt, ssq_nt;    /* > 0: the statistics are [rows, nt] partial sums (one per 64 columns), summed in index order / stored per tile; 0: one value per row */
  /* gated-GELU FFN (T5 v1.1, HF modeling_t5.py:97-123) fused into the GEMMs around it (tile_cfg 1, ks 0, whole 256x128 tiles):
   * epi 5 = forward: B = [wi_0; wi_1] ([2F, K], read gate-interleaved), N = 2F, C = h = dropout(gelu_new(u0) * u1) [M, F] (ldc), C2 = u =
   * [u0 | u1] [M, 2F] (ldc2) kept for the backward, gate_F = F;  epi 7 = backward: B = Wo^T [F, K], N = F, aux = u [M, 2F] (ldaux),
   * C = du = [dh u1 gelu'(u0) | dh gelu(u0)] [M, 2F] (ldc), dh = the product with the dropout mask of h re-applied; gate_F = 0 */
  void* C2; int ldc2, gate_F;
  /* epi 10 = T5LayerNorm backward (HF modeling_t5.py:59-72 under autograd) in the epilogue of the data-gradient GEMM that produces the
   * norm's input gradient (tile_cfg 4, ks 0, M % 128 == 0, N % 128 == 0, N = d_model): acc = dn = dOut W; aux = x [M, N] (ldaux) the
   * sub-layer's input rows, rowss / rowss_nt their partial sums of squares, nb_w [N] the norm weight, nb_dot [M, nb_dot_nt] partial sums
   * of <dOut, Out> per row (= sum_j (dn w)_j xh_j), nb_rin [M, N] fp32 incoming residual gradient.  Outputs: nb_rout [M, N] fp32 =
   * rstd (dn w - xh mean_j(dn w xh)) + nb_rin; C (ldc) = dropout(nb_rout) in bf16; C2 (ldc2, may be NULL) = w * round(x rstd);
   * nb_dw [M / 64, N] partial rows of the norm-weight gradient sum_rows dn xh.
   * epi 3 with ssq_out (tile_cfg 1): ssq_out[row][64-column group] = sum of C * aux / alpha (row sums of <d pre, pre>). */
  const float* nb_dot; int nb_dot_nt;
  const float* nb_rin; float* nb_rout;
  const float* nb_w; float* nb_dw;
} P5GemmProblem;
int p5_op_gemm_group(int tile_cfg, int ks, int nprob, const P5GemmProblem* probs, const uint32_t* rng_state, uint32_t site, float drop_p,
                     void* stream);
int p5_op_rmsnorm_fwd(int dtype, void* y, float* rstd, const void* x, const float* w, int rows, int d, float eps, void* stream);
int p5_op_rmsnorm_bwd(int dtype, float* dres_out, void* dy_next, float* dw, const void* dy, const void* x, const float* w,
                      const float* rstd, const float* dres_in, int rows, int d,
                      float* dw_partial /* scratch float[1024*d]: the engine's partial-sum mode; NULL = atomics */, void* stream);
int p5_op_attn_fwd(int dtype, const void* Q, const void* K, const void* V, void* O, float* lse, const float* rel_table,
                   const int* lut, int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk,
                   int ldv, int ldo, int causal, const uint32_t* rng_state, uint32_t site, float drop_p, void* stream);
/* d_rel_table [rel_buckets, H] (+=, may be NULL): the gradient of the relative-bias table is reduced WITHOUT fp32 atomics -- every
 * workgroup stores into its own slot of d_rel_scratch (caller-provided, room for float[B * ceil(Lq / 64)][rel_buckets * H]; the slots the launch uses are written in full, nothing is cleared) and the
 * slots are summed in index order, so the result is bit-reproducible. */
int p5_op_attn_bwd(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                   float* Dvec, void* dQ, void* dK, void* dV, const float* rel_table, float* d_rel_table, float* d_rel_scratch,
                   int rel_buckets, const int* lut, int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk,
                   int ldv, int ldo, int lddq, int lddk, int lddv, int causal, const uint32_t* rng_state, uint32_t site, float drop_p,
                   void* stream);
/* the same, and dot_out [B * Lq, H] = <dQ, Q> + <dK, K> + <dV, V> per token and head from the values as stored: the row sums the
 * T5LayerNorm-backward epilogue of the qkv data-gradient GEMM consumes (P5GemmProblem epi 10).  bf16 self-attention with
 * 16 < Lq == Lk <= 128 (the fused backward kernel) only; NULL = p5_op_attn_bwd. */
int p5_op_attn_bwd_dot(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                       float* Dvec, void* dQ, void* dK, void* dV, const float* rel_table, float* d_rel_table, float* d_rel_scratch,
                       int rel_buckets, const int* lut, int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk,
                       int ldv, int ldo, int lddq, int lddk, int lddv, int causal, const uint32_t* rng_state, uint32_t site, float drop_p,
                       float* dot_out, void* stream);
int p5_op_ce_fwd(float* nll, float* lse, const float* logits, const int64_t* labels, int rows, int V, int ldl, void* stream);
/* decode-step projection over a few hundred rows (p5_decode2.h): C = A W^T, W = T [N, ldw].  amode 0: A = T [M, lda];
 * amode 1: A = fp32 residual stream [M, K], normalised with T5LayerNorm weight `ln` by the kernel itself.
 * epi: 0 store T (alpha), 1 relu store T, 2 fp32 atomic accumulate (split-K), 3 store fp32 (alpha) */
int p5_op_skinny_gemm(int dtype, int amode, const void* A, int lda, const float* ln, const void* W, int ldw, void* C, int ldc,
                      int M, int N, int K, int epi, float alpha, float eps, void* stream);
/* decode-step cross-attention of the Kb beams of each of B items (p5_decode2.h): q T [B*Kb, H*64], kv T [B*L, 2*H*64] (K then V),
 * mask int64 [B, L]; zero position bias (HF modeling_t5.py:336-343).  variant 3 = matrix-core kernel, 2 = scalar kernel */
int p5_op_dec_cross_attn(int dtype, int variant, void* out, const void* q, const void* kv, const int64_t* mask, int B, int H, int Kb,
                         int L, void* stream);
int p5_op_tr_probe(void* out64x4_u16, const void* in256_u16, void* stream);  /* ds_read_b64_tr_b16 semantics probe */

#ifdef __cplusplus
}
#endif
#endif
