// p5_lib.hip -- host side of libp5hip.so: kernel launchers, the T5 engine (forward / backward / generate
// orchestration over one HIP stream) and the C ABI declared in include/p5hip.h.
//
// The engine replaces the Python object `P5_T5` (model/P5_T5.py:207) + HF T5ForConditionalGeneration + autograd:
// it owns no memory; parameters live in ONE flat fp32 arena (plus a bf16 shadow in fast mode), gradients in a
// second arena of the same layout (so clip + AdamW are two flat kernels and DDP buckets are contiguous
// ranges), activations in a caller-provided workspace.
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#ifndef P5_EMU
#include <dlfcn.h>
#endif

#include "p5_host.h"
#include "p5_elem.h"
#include "p5_embed.h"
#include "p5_decode.h"
#include "p5_decode2.h"
#include "p5_verify.h"
#include "../../include/p5hip.h"

thread_local std::string g_p5_err;

// =====================================================================================================
// launchers
// =====================================================================================================
// tuning knobs (p5_set_option / environment) of the engine and the decode step; the GEMM / attention families keep theirs in their own units
static int g_opt_wgrad_group = getenv("P5_WGRAD_GROUP") ? atoi(getenv("P5_WGRAD_GROUP")) : 1;   // layer-grouped deferred weight gradients (bf16)
static int g_opt_decode_fused = getenv("P5_DECODE_FUSED") ? atoi(getenv("P5_DECODE_FUSED")) : 1;   // RMSNorm folded into the decode-step GEMMs
static int g_opt_decode_v2 = getenv("P5_DECODE_V2") ? atoi(getenv("P5_DECODE_V2")) : 1;   // latency-shaped decode step (p5_decode2.h)
static int g_opt_dec_nb = getenv("P5_DEC_NB") ? atoi(getenv("P5_DEC_NB")) : 0;           // skinny GEMM: forced column-tile width (0 = auto)
static int g_opt_dec_kw = getenv("P5_DEC_KW") ? atoi(getenv("P5_DEC_KW")) : 0;           // skinny GEMM: forced K range per workgroup (0 = auto)
static int g_opt_dec_fuseq = getenv("P5_DEC_FUSEQ") ? atoi(getenv("P5_DEC_FUSEQ")) : 1;   // cross-attention computes its own q projection
static int g_opt_dgrad_t = getenv("P5_DGRAD_T") ? atoi(getenv("P5_DGRAD_T")) : 1;       // data gradients on the transposed weight copy when one is bound
static int g_opt_dec_cross = getenv("P5_DEC_CROSS") ? atoi(getenv("P5_DEC_CROSS")) : 3;   // 3 = MFMA cross-attention, 2 = scalar score / PV loops
static int g_opt_dec_head = getenv("P5_DEC_HEAD") ? atoi(getenv("P5_DEC_HEAD")) : 1;      // 1 = streaming head (no [R, V] logits), 0 = GEMM + score kernel
static int g_opt_dec_atomic = getenv("P5_DEC_ATOMIC") ? atoi(getenv("P5_DEC_ATOMIC")) : 0;   // 1 = round-2..4 decode step: residual stream updated with fp32 atomics by K-split workgroups (not bit-reproducible)
static int g_opt_adam_tiles = getenv("P5_ADAM_TILES") ? atoi(getenv("P5_ADAM_TILES")) : 1;       // p5_engine_adamw_step: tile-wise update that also writes W^T and W diag(ln)
static int g_opt_gen_ff = getenv("P5_GEN_FF") ? atoi(getenv("P5_GEN_FF")) : 1;             // forced-prefix fast-forward (p5_decode.h): 0 = every step is a decode step
static int g_opt_dec_head_nv = getenv("P5_DEC_HEAD_NV") ? atoi(getenv("P5_DEC_HEAD_NV")) : 0;   // streaming head: forced E rows per workgroup (0 = auto)

__global__ __launch_bounds__(256) void p5_shift_right_kernel(int64_t* out, const int64_t* labels, int B, int T, int64_t start) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * T) return;
  const int t = i % T;
  int64_t v = (t == 0) ? start : labels[i - 1];
  if (v == -100) v = start;
  out[i] = v;
}

__global__ __launch_bounds__(64) void p5_tr_probe_kernel(unsigned short* out, const unsigned short* in) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  // 16-lane group g reads the [4][16] block starting at element g*64, lane i -> row i/4, cols (i%4)*4..
  const u32x2 r = lds_tr16_b64(&lds[(l >> 4) * 64 + (l & 15) * 4]);
  out[l * 4 + 0] = (unsigned short)(r[0] & 0xFFFF);
  out[l * 4 + 1] = (unsigned short)(r[0] >> 16);
  out[l * 4 + 2] = (unsigned short)(r[1] & 0xFFFF);
  out[l * 4 + 3] = (unsigned short)(r[1] >> 16);
}

// =====================================================================================================
// engine
// =====================================================================================================
static constexpr int P5_MAX_STAGED = 160;     // final gradient ranges of one staged backward (<= stages: n_dec + n_enc + 4)
static constexpr int P5_NSETS = 8;      // >= 2 x (sub-layers of a decoder layer), >= 2 x (sub-layers of the encoder layers grouped into one launch)
struct ParamInfo { std::string name; int64_t off; int rows, cols; };
struct AttnOff { int64_t q, k, v, o, ln; };
struct LayerOff { AttnOff sa, ca; int64_t wi, wo, ff_ln; int64_t begin, end; };

struct LayerSave {
  void *x_sa, *n_sa, *qkv, *o_sa; float *rstd_sa, *lse_sa;
  void *x_ca, *n_ca, *q_ca, *kv_ca, *o_ca; float *rstd_ca, *lse_ca;
  void *x_ff, *n_ff, *u_ff, *h_ff; float *rstd_ff;
  float *ssq_sa, *ssq_ca, *ssq_ff;     // [rows, d/64] partial sums of squares of x_sa / x_ca / x_ff (norm folded into the GEMMs)
  uint32_t* keep_sa;                   // dropout keep masks of the self-attention probabilities (P5AttnArgs::keep_bits) or nullptr
};

struct Bump {
  char* base; size_t off;
  void* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

struct GraphKey { int B, L, K, max_len, max_c, excl_words; const void *ws, *trie, *trie_tok, *trie_node, *roots, *P, *S, *fold, *hist; int sz, fused; };

struct GenWs {
  void* kv_cross[64];   // per decoder layer: T [B*L, ldkv] -- column slices of ONE [B*L, n_dec*2*inner] block in the latency-shaped path
  int ldkv;             // (all layers' K/V projections are a single GEMM, as in training), separate [B*L, 2*inner] blocks otherwise
  void* cache[64];      // per decoder layer: T [max_len, R, 2*inner]
  void *xa, *xb, *n, *qkv, *q, *o, *h, *hn;
  float* x32;             // [R, d] fp32 residual stream of the decode step (v2: updated in place with atomics)
  float *logits, *cand, *row_top_score; int *n_cand, *row_top_c;
  float *part_m, *part_s; int* cand_key;   // streaming head: per (row, vocabulary tile) max / sum exp; candidate keys of pools beyond the LDS budget
  int64_t* mask_copy;
  float* ssq;             // [3 * n_dec_layers + 1][R] row sums of squares of the residual stream entering each norm (fused path)
  uint32_t* excluded;     // [B, excl_words] copy of the caller's per-item excluded-node bitmap (stable address for the graph)
  int64_t* ff_labels; float* ff_nll;     // forced-prefix pass: its labels [B, F] and per-token NLL
  P5BeamState st;
};

// state of the search between p5_decode_begin and p5_decode_finish (the workspace layout, the trie, the step counter)
struct GenCtx {
  bool active = false;
  GenWs w;
  int B = 0, L = 0, K = 0, max_len = 0, max_c = 0, excl_words = 0, steps = 0, steps0 = 0;     // steps0: steps covered by the forced-prefix pass
  const int *child_off = nullptr, *child_tok = nullptr, *child_node = nullptr, *roots = nullptr;
  char* ws = nullptr;
};

struct P5Engine {
  P5Config c;
  int inner;
  int64_t off_E, off_WW, off_enc_rel, off_dec_rel, off_enc_fln, off_dec_fln, n_params, off_small_end;
  std::vector<LayerOff> enc, dec;
  std::vector<ParamInfo> table;
  float* P = nullptr; float* G = nullptr; void* S = nullptr;
  const int* lut_enc = nullptr; const int* lut_dec = nullptr; int lut_half = 0;
  uint32_t* rng = nullptr;
  // ---- saved state of the last forward ----
  int B = 0, L = 0, T = 0, training = 0, M = 0, Md = 0, Vp = 0;
  const int64_t *ids = nullptr, *ww = nullptr, *mask = nullptr, *labels = nullptr, *out_attn = nullptr;
  std::vector<LayerSave> es, ds;
  void *enc_x0 = nullptr, *enc_xf = nullptr, *enc_out = nullptr; float* enc_rstd_f = nullptr;
  int64_t* dec_ids = nullptr; void *dec_x0 = nullptr, *dec_xf = nullptr, *dec_hn = nullptr; float* dec_rstd_f = nullptr;
  float *logits = nullptr, *lse_tok = nullptr, *ssq_scratch = nullptr;
  float *ce_part = nullptr, *ce_lab = nullptr, *ce_g = nullptr;      // logit-free cross-entropy: per-row partial (max, sum exp) pairs, label logits, NLL gradients
  bool ce_free_fwd = false;       // the last training forward took the logit-free head (the backward recomputes the logits tile by tile)
  void* Sf = nullptr;              // folded bf16 weight copy W diag(ln) for q/k/v, wi, cross-attention q (same arena offsets; behind St)
  float *dres_a = nullptr, *dres_b = nullptr, *d_enc = nullptr, *Dvec = nullptr, *dres_cur = nullptr, *rel_partial = nullptr;
  void *dy = nullptr, *dn = nullptr, *dqkv = nullptr, *dO = nullptr, *dh = nullptr, *du = nullptr, *dlogits = nullptr, *dkv = nullptr;
  // two sets of the temporaries the wgrad GEMMs read, alternated per sub-layer, so the side stream can run one
  // sub-layer behind the dgrad chain without a write-after-read hazard
  // (P5_NSETS sets: with the weight gradients of a whole layer deferred into ONE grouped launch at the end of the layer's backward --
  // p5_gemm4.h -- the temporaries of all its sub-layers, and of the next layer's that run meanwhile, must stay intact)
  void *dy2[P5_NSETS] = {}, *dh2[P5_NSETS] = {}, *du2[P5_NSETS] = {}, *dqkv2[P5_NSETS] = {}, *dkv2[P5_NSETS] = {};
  P5ReduceMulti nr_pending;               // norm-weight partial sums of the current backward stage, reduced by one launch at its end
  std::vector<P5GemmArgs> wg_pending;     // deferred weight-gradient problems (bf16, token count a multiple of 64)
  unsigned wg_sets = 0;                   // bit p: a pending problem reads temporaries of set p
  bool whole_backward = false;            // inside p5_backward (as opposed to stage-by-stage calls of a data-parallel caller)
  bool stage_pairs = true;                // staged backward: the encoder's weight gradients still leave in two-layer groups (the benchmarked launch)
  int64_t fin_b = 0, fin_e = 0;           // staged backward: gradient range completed by the stages run since the last p5_backward_final_range
  int64_t rep_b = 0, rep_e = 0;           // ... and the range that call reports
  void* dy_next = nullptr;
  void *kv_all = nullptr, *dkv_all = nullptr;   // cross-attention K/V (and their gradients) of all decoder layers, [M, n_dec*2*inner]
  float* dw_scratch = nullptr;   // [norm slots][<=1024 workgroups][d] partial norm-weight gradients
  // deterministic embedding gradients (p5_embed.h): set 0 = tied table E over the encoder ids followed by the decoder ids, set 1 = whole-word table
  int* emb_idx[P5_EMB_MAXSETS] = {};     // perm | skey | sstart | slen, n ints each
  unsigned long long* emb_csort[P5_EMB_MAXSETS] = {};
  float* emb_part[P5_EMB_MAXSETS] = {};
  float* dres_dec0 = nullptr;            // [Md, d] gradient of the decoder's embedding rows (kept until the last stage)
  float* dres_out_override = nullptr;    // the next swap_norm_bwd writes its residual gradient here
  float* nb_dotbuf = nullptr;            // [rows, max(d_ff / 64, n_heads)] partial sums of <dOut, Out> per row: producer = the kernel that writes dOut, consumer = the P5_EPI_NORM_BWD GEMM that follows it
  bool nb_done = false;                  // the sub-layer's norm backward ran in the epilogue of its data-gradient GEMM: the caller skips swap_norm_bwd
  int norm_slot = 0;
  int sub = -1;
  bool d_enc_started = false;
  bool grads_zeroed = false;      // the gradient arena holds zeros (p5_engine_clear_grads): the next backward need not clear it
  bool grads_keep = false;        // p5_engine_grads_zeroed(): the next backward ADDS to the arena (2nd.. micro-batch of an accumulation group)
  int wg_epi = P5_EPI_ACCUM;      // how this backward's grouped weight-gradient GEMMs write: P5_EPI_STORE on a first micro-batch
  // decode-step weights with the following RMSNorm weight folded in (W[out,in] * ln[in]): per decoder layer qkv / cross-q / wi,
  // and the tied head E * final_ln; caller-owned buffer in the compute dtype (p5_engine_bind_decode_fold)
  void* fold = nullptr;
  std::vector<int64_t> fold_qkv, fold_q, fold_wi;
  int64_t fold_E = 0, fold_count = 0;
  GenCtx gen;
  int* gen_hist_next = nullptr;    // p5_generate_draft: history buffer of the NEXT search (one-shot)
  const float* enc_ext_next = nullptr;   // p5_generate_set_encoder_output: fp32 encoder output the NEXT search starts from (one-shot)
  P5Forced ff_next = {0, {0}, {0}};  // p5_generate_set_forced_prefix: forced prefix of the NEXT search (one-shot)
  struct VerifyCtx* ver = nullptr;  // state of a verification pass between p5_verify_plan and p5_verify_run (p5_verify.h)
  // transposed bf16 copies of the 2-D layer weights (same arena offsets): the data gradients dx = dy W then read W^T as a
  // K-contiguous operand, i.e. run on the forward kernel (p5_engine_bind_transposed; optional)
  void* St = nullptr;
  struct TrDesc { int64_t off; int rows, cols, tile0; int64_t ln_off; };
  std::vector<TrDesc> tr_list;
  int tr_tiles = 0;
  int adam_tiles = 0;             // 64 x 256 tiles of p5_adamw_tiles_kernel over the same blocks
  bool tr_pending = false;
  bool zg_pending = false;        // p5_engine_clear_grads(): the clear runs on the side stream; the next backward waits for zg_ev
#ifndef P5_EMU
  hipEvent_t zg_ev = nullptr;
  hipEvent_t tr_ev = nullptr;
  hipEvent_t gen_ev[3] = {nullptr, nullptr, nullptr};   // p5_generate_timing: before the encoder pass / before the first / after the last decode step
  bool gen_timing = false;
  hipGraphExec_t gen_graph_exec = nullptr;
  bool gen_graph_failed = false;
  GraphKey gen_graph_key;
#endif
  bool st_side[P5_MAX_STAGED] = {};
  int64_t st_range[2 * P5_MAX_STAGED] = {};
  int st_n = 0;                   // p5_backward_staged: final ranges of the last call (events st_ev[0 .. st_n))
#ifndef P5_EMU
  hipEvent_t st_ev[P5_MAX_STAGED] = {};
  hipEvent_t st_ev_side[P5_MAX_STAGED] = {};      // the same point of the engine's side stream (weight gradients of the range may run there)
#endif
  // optional second stream for the weight-gradient GEMMs (off the critical dgrad chain)
  hipStream_t side = nullptr;
#ifndef P5_EMU
  bool side_events = false;               // the events below exist (created with the first side stream, destroyed with the engine)
  hipEvent_t ev_pool[32];
  hipEvent_t set_ev[P5_NSETS];            // recorded behind the last weight-gradient launch that reads set p
  hipEvent_t head_wg_ev;                  // ... behind the tied head's weight gradient (plain "+=" into shared.weight's gradient)
  hipEvent_t kv_ev[64];
  bool set_ev_valid[P5_NSETS] = {};
  bool head_wg_valid = false;
  int ev_next = 0;
#endif
};

static void begin_sublayer(P5Engine* e) {
  e->sub++;
  const int p = e->sub % P5_NSETS, pn = (e->sub + 1) % P5_NSETS;
  e->dy = e->dy2[p]; e->dy_next = e->dy2[pn];
  e->dh = e->dh2[p]; e->du = e->du2[p]; e->dqkv = e->dqkv2[p]; e->dkv = e->dkv2[p];
}
// side stream waits for everything enqueued on `main` so far
static void fork_to_side(P5Engine* e, hipStream_t main) {
#ifndef P5_EMU
  if (!e->side) return;
  hipEvent_t ev = e->ev_pool[e->ev_next++ & 31];
  hipEventRecord(ev, main);
  hipStreamWaitEvent(e->side, ev, 0);
#endif
}
static void join_side(P5Engine* e, hipStream_t main) {
#ifndef P5_EMU
  if (!e->side) return;
  hipEvent_t ev = e->ev_pool[e->ev_next++ & 31];
  hipEventRecord(ev, e->side);
  hipStreamWaitEvent(main, ev, 0);
#endif
}
static int wgrad_flush(P5Engine* e, hipStream_t main, bool is_head, bool on_side);
// called right before the norm backward that ends sub-layer `sub`: it writes dy of the NEXT set, and the next sub-layer's data
// gradients write the rest of that set -- whatever weight-gradient launch still reads it (P5_NSETS sub-layers ago) must be done
static void end_sublayer_sync(P5Engine* e, hipStream_t main) {
  if ((e->wg_sets >> ((e->sub + 1) % P5_NSETS)) & 1) wgrad_flush(e, main, false, false);   // (cannot happen with one flush per layer; kept for safety)
#ifndef P5_EMU
  if (!e->side) return;
  const int p = e->sub % P5_NSETS, pn = (e->sub + 1) % P5_NSETS;
  if (!((e->wg_sets >> p) & 1)) {                      // this sub-layer's weight gradients were launched one by one: they are all enqueued
    hipEventRecord(e->set_ev[p], e->side);
    e->set_ev_valid[p] = true;
  }
  if (e->set_ev_valid[pn]) { hipStreamWaitEvent(main, e->set_ev[pn], 0); e->set_ev_valid[pn] = false; }
#endif
}
static hipStream_t wgrad_stream(P5Engine* e, hipStream_t main) {
  if (!e->side) return main;
  fork_to_side(e, main);
  return e->side;
}

static void add_param(P5Engine* e, const std::string& name, int rows, int cols, int64_t& off_out) {
  // every tensor starts on a 64-element boundary so 16-byte vector accesses stay aligned in both dtypes
  e->n_params = (e->n_params + 63) & ~(int64_t)63;
  off_out = e->n_params;
  e->table.push_back({name, e->n_params, rows, cols});
  e->n_params += (int64_t)rows * cols;
}

static void add_attn(P5Engine* e, const std::string& p, AttnOff& a) {
  const int d = e->c.d_model, in = e->inner;
  add_param(e, p + ".q.weight", in, d, a.q);
  // q,k,v must be back-to-back ([3*inner, d] fused projection): inner*d is a multiple of 64, so no padding appears
  add_param(e, p + ".k.weight", in, d, a.k);
  add_param(e, p + ".v.weight", in, d, a.v);
  add_param(e, p + ".o.weight", d, in, a.o);
}

static void build_layout(P5Engine* e) {
  const P5Config& c = e->c;
  const int d = c.d_model, F = c.d_ff;
  e->n_params = 0;
  add_param(e, "shared.weight", c.vocab_size, d, e->off_E);
  add_param(e, "encoder.whole_word_embeddings.weight", c.whole_word_size, d, e->off_WW);
  add_param(e, "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", c.rel_buckets, c.n_heads, e->off_enc_rel);
  add_param(e, "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", c.rel_buckets, c.n_heads, e->off_dec_rel);
  e->n_params = (e->n_params + 63) & ~(int64_t)63;
  e->off_small_end = e->n_params;
  auto add_ff = [&](const std::string& p, LayerOff& l) {
    if (c.gated_gelu) {
      int64_t t;
      add_param(e, p + ".DenseReluDense.wi_0.weight", F, d, l.wi);
      add_param(e, p + ".DenseReluDense.wi_1.weight", F, d, t);
    } else {
      add_param(e, p + ".DenseReluDense.wi.weight", F, d, l.wi);
    }
    add_param(e, p + ".DenseReluDense.wo.weight", d, F, l.wo);
  };
  e->enc.resize(c.n_enc_layers);
  for (int i = 0; i < c.n_enc_layers; ++i) {
    LayerOff& l = e->enc[i];
    const std::string p = "encoder.block." + std::to_string(i);
    e->n_params = (e->n_params + 63) & ~(int64_t)63;
    l.begin = e->n_params;
    add_attn(e, p + ".layer.0.SelfAttention", l.sa);
    add_param(e, p + ".layer.0.layer_norm.weight", 1, d, l.sa.ln);
    add_ff(p + ".layer.1", l);
    add_param(e, p + ".layer.1.layer_norm.weight", 1, d, l.ff_ln);
    l.end = e->n_params;
  }
  add_param(e, "encoder.final_layer_norm.weight", 1, d, e->off_enc_fln);
  e->dec.resize(c.n_dec_layers);
  // the cross-attention K/V projections of ALL decoder layers sit back to back ([n_dec * 2 * inner, d]): they all read the same
  // encoder output, so forward, weight gradient and d(enc_out) are one GEMM each instead of one per layer.  The block lies
  // between the encoder's final norm and the first decoder layer, i.e. inside the data-parallel bucket of the stage that
  // follows the decoder (p5_backward_stage_range), which is when its gradient is complete.
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const std::string p = "decoder.block." + std::to_string(i) + ".layer.1.EncDecAttention";
    add_param(e, p + ".k.weight", e->inner, d, e->dec[i].ca.k);     // inner*d is a multiple of 64: no padding in between
    add_param(e, p + ".v.weight", e->inner, d, e->dec[i].ca.v);
  }
  for (int i = 0; i < c.n_dec_layers; ++i) {
    LayerOff& l = e->dec[i];
    const std::string p = "decoder.block." + std::to_string(i);
    e->n_params = (e->n_params + 63) & ~(int64_t)63;
    l.begin = e->n_params;
    add_attn(e, p + ".layer.0.SelfAttention", l.sa);
    add_param(e, p + ".layer.0.layer_norm.weight", 1, d, l.sa.ln);
    add_param(e, p + ".layer.1.EncDecAttention.q.weight", e->inner, d, l.ca.q);
    add_param(e, p + ".layer.1.EncDecAttention.o.weight", d, e->inner, l.ca.o);
    add_param(e, p + ".layer.1.layer_norm.weight", 1, d, l.ca.ln);
    add_ff(p + ".layer.2", l);
    add_param(e, p + ".layer.2.layer_norm.weight", 1, d, l.ff_ln);
    l.end = e->n_params;
  }
  add_param(e, "decoder.final_layer_norm.weight", 1, d, e->off_dec_fln);
  e->n_params = (e->n_params + 63) & ~(int64_t)63;
  // weights whose data gradient runs on the transposed copy: (offset, rows, cols) of each [rows, cols] block
  {
    e->tr_list.clear();
    int tiles = 0;
    // (ln: the T5LayerNorm weight whose output the projection consumes -- the blocks p5_refresh_transposed folds -- or -1)
    auto add = [&](int64_t off, int rows, int cols, int64_t ln = -1) {
      e->tr_list.push_back({off, rows, cols, tiles, ln});
      tiles += ((rows + 63) / 64) * ((cols + 63) / 64);
    };
    const int wi_rows = (c.gated_gelu ? 2 : 1) * F;
    for (int i = 0; i < c.n_enc_layers; ++i) {
      add(e->enc[i].sa.q, 3 * e->inner, d, e->enc[i].sa.ln); add(e->enc[i].sa.o, d, e->inner); add(e->enc[i].wi, wi_rows, d, e->enc[i].ff_ln); add(e->enc[i].wo, d, F);
    }
    add(e->dec[0].ca.k, c.n_dec_layers * 2 * e->inner, d);
    for (int i = 0; i < c.n_dec_layers; ++i) {
      add(e->dec[i].sa.q, 3 * e->inner, d, e->dec[i].sa.ln); add(e->dec[i].sa.o, d, e->inner); add(e->dec[i].ca.q, e->inner, d, e->dec[i].ca.ln);
      add(e->dec[i].ca.o, d, e->inner);
      add(e->dec[i].wi, wi_rows, d, e->dec[i].ff_ln); add(e->dec[i].wo, d, F);
    }
    e->tr_tiles = tiles;
  }
  // folded decode-step weights (element offsets into the fold buffer, 64-element aligned)
  {
    const int in = c.n_heads * c.d_kv;
    int64_t off = 0;
    auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) & ~(int64_t)63; return o; };
    e->fold_qkv.clear(); e->fold_q.clear(); e->fold_wi.clear();
    for (int i = 0; i < c.n_dec_layers; ++i) {
      e->fold_qkv.push_back(take((int64_t)3 * in * d));
      e->fold_q.push_back(take((int64_t)in * d));
      e->fold_wi.push_back(take((int64_t)(c.gated_gelu ? 2 : 1) * F * d));
    }
    e->fold_E = take((int64_t)c.vocab_size * d);
    e->fold_count = off;
  }
}

template <class T> static const T* Wc(const P5Engine* e, int64_t off) {
  if (sizeof(T) == 4) return (const T*)(e->P + off);
  return (const T*)((const bf16*)e->S + off);
}

static P5Drop mk_drop(const P5Engine* e, int stack, int layer, int which) {
  P5Drop d;
  d.state = nullptr; d.site_key = 0; d.thr = 0; d.scale = 1.f;
  if (e->training && e->c.dropout > 0.f) {
    d.state = e->rng;
    d.site_key = p5_site_key(p5_site_id(stack, layer, which));
    d.thr = p5_drop_thr(e->c.dropout);
    d.scale = 1.f / (1.f - e->c.dropout);
  }
  return d;
}
static P5Drop no_drop() { P5Drop d; d.state = nullptr; d.site_key = 0; d.thr = 0; d.scale = 1.f; return d; }

// fp32 engine: products of the K-contiguous forward GEMMs on the f16 matrix cores from a two-term split (p5_gemm.h); switched on by the
// verification pass for the duration of its calls (option "verify_split"), off everywhere else: the fp32 parity engine stays exact
static thread_local int g_f32_split = 0;
static int g_opt_verify_split = getenv("P5_VERIFY_SPLIT") ? atoi(getenv("P5_VERIFY_SPLIT")) : 1;
struct SplitScope { int prev; explicit SplitScope(int on) : prev(g_f32_split) { g_f32_split = on; } ~SplitScope() { g_f32_split = prev; } };

template <class T>
static int gemm(hipStream_t s, const void* A, int lda, int aks, const void* Bm, int ldb, int bks, void* C, int ldc, int M, int N,
                int K, int epi, const void* aux, int ldaux, float alpha, int c_f32, P5Drop drop, const float* rowss = nullptr,
                float rowss_eps = 0.f, float* ssq_out = nullptr) {
  P5GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = Bm; g.C = C; g.aux = aux; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.a_ks = aks; g.b_ks = bks; g.epi = epi; g.c_f32 = c_f32; g.splitk = 0; g.ring = 0; g.alpha = alpha; g.drop = drop;
  g.rowss = rowss; g.rowss_invd = 1.0f / (float)K; g.rowss_eps = rowss_eps; g.ssq_out = ssq_out;
  g.rowss_nt = 0; g.ssq_nt = 0; g.g4_tiles_n = 0; g.g4_nk = 0; g.xcd_bm = g.xcd_bn = 0; g.c_split_stride = 0;
  g.mm_split = (sizeof(T) == 4 && !aks && !bks && epi != P5_EPI_ATOMIC && epi != P5_EPI_ACCUM) ? g_f32_split : 0;
  return launch_gemm<T>(g, s);
}
// TRAINING forward with T5LayerNorm folded in (bf16 engine): y = rstd(x) * (x Wf^T), Wf = W diag(ln) from the folded
// weight copy, rstd from the d/64 partial sums of squares per row that the PRODUCER of x left behind (`rowss`); `ssq_out`: this
// GEMM's own output is a residual-stream row whose partial sums the next folded GEMM needs.
template <class T>
static int gemm_nf(hipStream_t s, const void* A, int lda, const void* Bm, int ldb, void* C, int ldc, int M, int N, int K, int epi, const void* aux,
                   int ldaux, P5Drop drop, const float* rowss, float eps, float* ssq_out, int d_model) {
  P5GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = Bm; g.C = C; g.aux = aux; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.epi = epi; g.alpha = 1.f; g.drop = drop;
  g.rowss = rowss; g.rowss_invd = 1.0f / (float)K; g.rowss_eps = eps; g.rowss_nt = rowss ? d_model / 64 : 0;
  g.ssq_out = ssq_out; g.ssq_nt = ssq_out ? d_model / 64 : 0;
  return launch_gemm<T>(g, s);
}
// decode step: y = rmsnorm(x) W^T with the norm weight folded into Wf and the row statistic taken from `rowss` (sum of squares
// of the rows of x, K = d_model); optional ssq_out collects the sum of squares of the rows of y for the NEXT norm
template <class T>
static int linear_fwd_fused(hipStream_t s, const void* x, int ldx, const T* Wf, void* y, int ldy, int M, int N, int K, int epi,
                            const void* aux, int ldaux, float alpha, int c_f32, const float* rowss, float eps, float* ssq_out) {
  return gemm<T>(s, x, ldx, 0, Wf, K, 0, y, ldy, M, N, K, epi, aux, ldaux, alpha, c_f32, no_drop(), rowss, eps, ssq_out);
}
// y = x W^T
template <class T>
static int linear_fwd(hipStream_t s, const void* x, int ldx, const T* W, void* y, int ldy, int M, int N, int K, int epi = P5_EPI_STORE,
                      const void* aux = nullptr, int ldaux = 0, float alpha = 1.f, int c_f32 = 0, P5Drop drop = no_drop()) {
  return gemm<T>(s, x, ldx, 0, W, K, 0, y, ldy, M, N, K, epi, aux, ldaux, alpha, c_f32, drop);
}
// dx = dy W     (W is [N_out, K_in] row-major; reduction over N_out)
template <class T>
static int linear_dgrad(hipStream_t s, const void* dy, int lddy, const T* W, void* dx, int lddx, int M, int N_out, int K_in,
                        int epi = P5_EPI_STORE, const void* aux = nullptr, int ldaux = 0, float alpha = 1.f, int c_f32 = 0) {
  return gemm<T>(s, dy, lddy, 0, W, K_in, 1, dx, lddx, M, K_in, N_out, epi, aux, ldaux, alpha, c_f32, no_drop());
}
// dx = dy W with W given by its arena offset: when a transposed copy is bound (bf16), W^T [K_in, N_out] is a K-contiguous operand
// and the product runs on the forward kernel (direct-to-LDS copies of both operands) instead of the K-strided-B variant
template <class T>
static int dgrad_w(P5Engine* e, hipStream_t s, const void* dy, int lddy, int64_t w_off, void* dx, int lddx, int M, int N_out, int K_in,
                   int epi = P5_EPI_STORE, const void* aux = nullptr, int ldaux = 0, float alpha = 1.f, int c_f32 = 0) {
  if (sizeof(T) == 2 && e->St && g_opt_dgrad_t && (N_out % 64) == 0)
    return gemm<T>(s, dy, lddy, 0, (const T*)e->St + w_off, N_out, 0, dx, lddx, M, K_in, N_out, epi, aux, ldaux, alpha, c_f32, no_drop());
  return linear_dgrad<T>(s, dy, lddy, Wc<T>(e, w_off), dx, lddx, M, N_out, K_in, epi, aux, ldaux, alpha, c_f32);
}
static int g_opt_wgrad_split_atomic = getenv("P5_WGRAD_SPLIT_ATOMIC") ? atoi(getenv("P5_WGRAD_SPLIT_ATOMIC")) : 0;    // 1 = split-K atomics (not reproducible)
// dW += dy^T x   (into the grad arena)
template <class T>
static int linear_wgrad_on(hipStream_t s, const void* dy, int lddy, const void* x, int ldx, float* dW, int M, int N_out, int K_in,
                        float alpha = 1.f) {
  // ONE split: every element of dW then receives exactly one add per backward -- the same bits on every run.  (Split-K with fp32
  // atomics -- rounds 1-3 -- made the last bits depend on the order the splits landed in; the grouped bf16 path never splits.)
  P5GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dy; g.B = x; g.C = dW; g.M = N_out; g.N = K_in; g.K = M; g.lda = lddy; g.ldb = ldx; g.ldc = K_in;
  g.a_ks = 1; g.b_ks = 1; g.epi = P5_EPI_ATOMIC; g.c_f32 = 1; g.splitk = g_opt_wgrad_split_atomic ? 0 : 1; g.alpha = alpha; g.drop = no_drop();
  g.rowss_invd = 1.0f / (float)M;
  return launch_gemm<T>(g, s);
}

// Weight gradients of the bf16 engine are DEFERRED: the problem is queued and the whole layer's queue goes out as one launch of the
// persistent ring kernel (p5_gemm4.h) -- every 128x128 tile of every weight of the layer reduces over ALL tokens (no split-K, no
// atomics, plain "dW += acc"; each weight has exactly one writer).  wgrad_flush is called at the end of every backward stage.
// which grouped weight-gradient launches go to the side stream: bit 0 = tied head + decoder layers (they run beside the decoder's chain of
// small, latency-bound kernels, which leaves most CUs idle), bit 1 = cross-attention K/V block + encoder layers (every main-stream
// kernel of the encoder backward fills the GPU by itself: beside it a 90 us weight-gradient workgroup only blocks CUs -- measured
// 5.07 ms per step with everything on the side stream, 4.74 ms with nothing on it)
static int g_opt_wgrad_side = getenv("P5_WGRAD_SIDE") ? atoi(getenv("P5_WGRAD_SIDE")) : 1;
static int g_opt_wgrad_wide = getenv("P5_WGRAD_WIDE") ? atoi(getenv("P5_WGRAD_WIDE")) : 1;
static int g_opt_wgrad_wide_min = getenv("P5_WGRAD_WIDE_MIN") ? atoi(getenv("P5_WGRAD_WIDE_MIN")) : 160;
static int g_opt_wgrad_wgs = getenv("P5_WGRAD_WGS") ? atoi(getenv("P5_WGRAD_WGS")) : 0;          // workgroups of a grouped weight-gradient launch (0 = one per unit, <= 256)
static int g_opt_wgrad_layers = getenv("P5_WGRAD_LAYERS") ? atoi(getenv("P5_WGRAD_LAYERS")) : 2;  // encoder layers per grouped launch (p5_backward only; staged backward: 1)
static int wgrad_flush(P5Engine* e, hipStream_t main, bool is_head, bool on_side = true) {
  if (e->wg_pending.empty()) return 0;
  hipStream_t s = on_side ? wgrad_stream(e, main) : main;        // (side: it now waits for everything the main stream has been given)
  size_t i = 0;
  while (i < e->wg_pending.size()) {
    P5GemmGroup grp;
    memset(&grp, 0, sizeof(grp));
    long u256 = 0;
    while (i < e->wg_pending.size() && grp.nprob < P5_MAX_GROUP) {
      const P5GemmArgs& q = e->wg_pending[i];
      u256 += (long)((q.M + 255) / 256) * ((q.N + 127) / 128);
      grp.p[grp.nprob++] = e->wg_pending[i++];
    }
    const int keep = g_opt_g4_wgs;
    if (g_opt_wgrad_wgs > 0) g_opt_g4_wgs = g_opt_wgrad_wgs;
    // 256x128 tiles (eight waves) once the group has enough of them to occupy most CUs: 3/4 of the operand bytes per MAC copied
    // into LDS -- which is what bounds these launches (tools/lab: two encoder layers 130 us vs 163 us on 128x128 tiles; one
    // layer alone has only 96 such tiles and stays on 128x128: 85 vs 102 us)
    const int rc = launch_gemm4(g_opt_wgrad_wide && u256 >= g_opt_wgrad_wide_min ? P5_G4_256x128 : P5_G4_128x128, true, grp, s);
    g_opt_g4_wgs = keep;
    P5_TRY(rc);
  }
#ifndef P5_EMU
  if (e->side) {
    for (int p = 0; p < P5_NSETS; ++p)
      if ((e->wg_sets >> p) & 1) { hipEventRecord(e->set_ev[p], s); e->set_ev_valid[p] = true; }
    if (is_head) { hipEventRecord(e->head_wg_ev, s); e->head_wg_valid = true; }
  }
#endif
  e->wg_pending.clear();
  e->wg_sets = 0;
  return 0;
}
// ONE predicate for "every weight gradient of this step's backward takes the deferred, layer-grouped path" (token counts that are
// multiples of 64; the other conditions of linear_wgrad -- leading dimensions, 16-byte alignment -- hold by construction of the layout):
// the folded-norm forward, the storing backward and linear_wgrad itself all depend on it and must not drift apart.
template <class T> static bool wg_all_deferred(const P5Engine* e) {
  return sizeof(T) == 2 && g_opt_wgrad_group != 0 && (e->M % 64) == 0 && (e->Md % 64) == 0;
}
template <class T>
static int linear_wgrad(P5Engine* e, hipStream_t main, const void* dy, int lddy, const void* x, int ldx, float* dW, int M, int N_out, int K_in,
                        float alpha = 1.f) {
  if (sizeof(T) == 2 && g_opt_wgrad_group && (M % 64) == 0 && (lddy % 8) == 0 && (ldx % 8) == 0 && ((uintptr_t)dy % 16) == 0 &&
      ((uintptr_t)x % 16) == 0 && (K_in % 4) == 0) {
    P5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dy; g.B = x; g.C = dW; g.M = N_out; g.N = K_in; g.K = M; g.lda = lddy; g.ldb = ldx; g.ldc = K_in;
    g.a_ks = 1; g.b_ks = 1; g.epi = e->wg_epi; g.c_f32 = 1; g.splitk = 1; g.alpha = alpha; g.drop = no_drop();
    e->wg_pending.push_back(g);
    if (e->sub >= 0) e->wg_sets |= 1u << (e->sub % P5_NSETS);
    return 0;
  }
  // an immediate weight gradient: under the folded-norm forward it would read the normalised rows before the norm backward has written
  // them, under a storing backward its target has not been cleared -- both modes require wg_all_deferred, which this problem contradicts
  P5_REQUIRE(!(e->Md > 0 && wg_all_deferred<T>(e)), "linear_wgrad: a weight gradient fell off the grouped path although wg_all_deferred holds (leading dimension / alignment)");
  if (e->wg_epi == P5_EPI_STORE) hipMemsetAsync(dW, 0, (size_t)N_out * K_in * 4, wgrad_stream(e, main));
  return linear_wgrad_on<T>(wgrad_stream(e, main), dy, lddy, x, ldx, dW, M, N_out, K_in, alpha);
}

template <class T>
static int rmsnorm_fwd(hipStream_t s, void* y, float* rstd, const void* x, const float* w, int rows, int d, float eps, P5Drop drop) {
  P5_REQUIRE(d % TT<T>::EPF == 0 && d <= 1024, "rmsnorm: d_model must be <= 1024 and a multiple of 8");
  P5_LAUNCH((p5_rmsnorm_fwd_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, (T*)y, rstd, (const T*)x, w, rows, d, eps, drop);
  return P5_KCHECK();
}
// workgroups of the norm backward (4 rows per workgroup and pass): with fewer than rows / 8 of them a wave makes several passes and its
// stores of one pass overlap its loads of the next
static int g_opt_norm_bwd_blocks = getenv("P5_NORM_BWD_BLOCKS") ? atoi(getenv("P5_NORM_BWD_BLOCKS")) : 1024;
template <class T>
static int rmsnorm_bwd(hipStream_t s, float* dres_out, void* dy_next, float* dw, const void* dy, const void* x, const float* w,
                       const float* rstd, const float* dres_in, int rows, int d, P5Drop din, P5Drop dnext, float* dw_partial = nullptr,
                       int* nblocks_out = nullptr, const float* ssq_part = nullptr, void* n_out = nullptr, float eps = 0.f) {
  P5_REQUIRE(d % TT<T>::EPF == 0 && d <= 1024, "rmsnorm: d_model must be <= 1024 and a multiple of 8");
  int blocks = (rows + 3) / 4;
  const int cap = g_opt_norm_bwd_blocks < 64 ? 64 : (g_opt_norm_bwd_blocks > 1024 ? 1024 : g_opt_norm_bwd_blocks);     // (the partial-sum scratch holds 1024 rows per norm)
  if (blocks > cap) blocks = cap;
  if (nblocks_out) *nblocks_out = blocks;
  const int nch = (d / TT<T>::EPF + 63) / 64;          // 16-byte pieces per lane
#define P5_NBWD(N) P5_LAUNCH((p5_rmsnorm_bwd_kernel<T, N>), dim3(blocks), dim3(256), 0, s, dres_out, (T*)dy_next, dw, (const T*)dy, (const T*)x, w, rstd, \
                             dres_in, rows, d, din, dnext, dw_partial, ssq_part, (T*)n_out, eps)
  if (nch <= 1) P5_NBWD(1);
  else if (nch == 2) P5_NBWD(2);
  else P5_NBWD(4);
#undef P5_NBWD
  return P5_KCHECK();
}

// relative-bias gradient: one slot of [rel_buckets, H] partial sums per attention-backward workgroup (layer x batch item x 64-query block),
// stored by that workgroup only and summed in slot order (p5_attn.h rel_bias_grad_flush)
static int rel_slots(int B, int Lq) { return B * ((Lq + 63) / 64); }
static constexpr int P5_HEAD_SPLITS = 32;      // most K-splits of the tied head's input-gradient GEMM (slices of its partial-product buffer)

// A backward that starts a new accumulation group does not clear the 4 B x n_params gradient arena and then add into it: every
// Linear weight's gradient is written by exactly one GEMM tile pass, which STORES on that first micro-batch (and `+=` on later ones);
// the head's gradient of the tied embedding is stored before the embedding scatter-adds land on it.  Only the parameters whose
// gradients are sums of atomics / partial reductions (whole-word embedding, relative-bias tables, T5LayerNorm weights: ~1 MB of
// T5-small's 242 MB) are cleared, by one table-driven launch.  Saves the fill and the read of the arena per step.
static int g_opt_grad_store_first = getenv("P5_GRAD_STORE_FIRST") ? atoi(getenv("P5_GRAD_STORE_FIRST")) : 1;
// embedding gradients as fixed-order segmented sums (p5_embed.h) instead of fp32 atomic scatters: bit-reproducible training (0 = the
// atomic scatter of rounds 1-3, kept for A/B timing)
static int g_opt_embed_det = getenv("P5_EMBED_DET") ? atoi(getenv("P5_EMBED_DET")) : 1;
struct P5ZeroTab {
  int n;
  struct D { long long off; int count, blk0; } e[200];
};
__global__ __launch_bounds__(256) void p5_zero_segments_kernel(float* __restrict__ G, P5ZeroTab tab) {
  const int b = blockIdx.x;
  int lo = 0, hi = tab.n - 1;             // last segment with blk0 <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.e[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const int i0 = (b - tab.e[lo].blk0) * 8192, n = tab.e[lo].count;
  float* __restrict__ dst = G + tab.e[lo].off;
  for (int i = i0 + threadIdx.x; i < i0 + 8192 && i < n; i += 256) dst[i] = 0.f;
}
static int zero_small_grads(P5Engine* e, hipStream_t s) {
  const P5Config& c = e->c;
  P5ZeroTab tab;
  tab.n = 0;
  int blocks = 0;
  auto add = [&](int64_t off, int64_t count) {
    if (count <= 0 || tab.n >= 200) return;
    P5ZeroTab::D& q = tab.e[tab.n++];
    q.off = off; q.count = (int)count; q.blk0 = blocks;
    blocks += (int)((count + 8191) / 8192);
  };
  add(e->off_WW, (int64_t)c.whole_word_size * c.d_model);
  add(e->off_enc_rel, (int64_t)c.rel_buckets * c.n_heads);
  add(e->off_dec_rel, (int64_t)c.rel_buckets * c.n_heads);
  for (const LayerOff& l : e->enc) { add(l.sa.ln, c.d_model); add(l.ff_ln, c.d_model); }
  for (const LayerOff& l : e->dec) { add(l.sa.ln, c.d_model); add(l.ca.ln, c.d_model); add(l.ff_ln, c.d_model); }
  add(e->off_enc_fln, c.d_model);
  add(e->off_dec_fln, c.d_model);
  P5_REQUIRE(tab.n < 200, "zero_small_grads: too many segments");
  P5_LAUNCH(p5_zero_segments_kernel, dim3(blocks), dim3(256), 0, s, e->G, tab);
  return P5_KCHECK();
}

static int64_t layout_ws(P5Engine* e, char* base, int B, int L, int T, bool with_bwd) {
  const P5Config& c = e->c;
  const size_t sz = c.dtype == 1 ? 2 : 4;
  const int d = c.d_model, in = e->inner, F = c.d_ff, H = c.n_heads;
  const size_t M = (size_t)B * L, Md = (size_t)B * T;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  Bump b{base, 0};
  e->es.assign(c.n_enc_layers, LayerSave());
  e->ds.assign(c.n_dec_layers, LayerSave());
  e->enc_x0 = b.take(M * d * sz);
  void* xprev = e->enc_x0;
  for (auto& l : e->es) {
    l.x_sa = xprev;
    l.n_sa = b.take(M * d * sz); l.rstd_sa = (float*)b.take(M * 4);
    l.ssq_sa = (float*)b.take(M * (d / 64) * 4); l.ssq_ff = (float*)b.take(M * (d / 64) * 4); l.ssq_ca = nullptr;
    l.qkv = b.take(M * 3 * in * sz); l.o_sa = b.take(M * in * sz); l.lse_sa = (float*)b.take((size_t)B * H * L * 4);
    // long sequences (the head-resident attention kernels): the forward's dropout decisions as bit masks for the two backward passes
    l.keep_sa = (with_bwd && c.dtype == 1 && L > 128 && g_opt_attn_keep_bits) ? (uint32_t*)b.take((size_t)B * H * ((L + 15) / 16) * 1024) : nullptr;
    l.x_ff = b.take(M * d * sz);
    l.n_ff = b.take(M * d * sz); l.rstd_ff = (float*)b.take(M * 4);
    l.u_ff = c.gated_gelu ? b.take(M * 2 * F * sz) : nullptr;
    l.h_ff = b.take(M * F * sz);
    xprev = b.take(M * d * sz);
  }
  e->enc_xf = xprev;
  e->ssq_scratch = (float*)b.take((M > Md ? M : Md) * (d / 64) * 4);     // statistics of rows nobody normalises through a GEMM (stack outputs)
  e->enc_rstd_f = (float*)b.take(M * 4);
  e->enc_out = b.take(M * d * sz);
  if (T > 0) {
    e->dec_ids = (int64_t*)b.take(Md * 8);
    e->dec_x0 = b.take(Md * d * sz);
    xprev = e->dec_x0;
    for (auto& l : e->ds) {
      l.x_sa = xprev;
      l.n_sa = b.take(Md * d * sz); l.rstd_sa = (float*)b.take(Md * 4);
      l.ssq_sa = (float*)b.take(Md * (d / 64) * 4); l.ssq_ca = (float*)b.take(Md * (d / 64) * 4); l.ssq_ff = (float*)b.take(Md * (d / 64) * 4);
      l.qkv = b.take(Md * 3 * in * sz); l.o_sa = b.take(Md * in * sz); l.lse_sa = (float*)b.take((size_t)B * H * T * 4);
      l.keep_sa = nullptr;
      l.x_ca = b.take(Md * d * sz);
      l.n_ca = b.take(Md * d * sz); l.rstd_ca = (float*)b.take(Md * 4);
      l.q_ca = b.take(Md * in * sz); l.o_ca = b.take(Md * in * sz);
      l.lse_ca = (float*)b.take((size_t)B * H * T * 4);
      l.x_ff = b.take(Md * d * sz);
      l.n_ff = b.take(Md * d * sz); l.rstd_ff = (float*)b.take(Md * 4);
      l.u_ff = c.gated_gelu ? b.take(Md * 2 * F * sz) : nullptr;
      l.h_ff = b.take(Md * F * sz);
      xprev = b.take(Md * d * sz);
    }
    e->dec_xf = xprev;
    // cross-attention K/V of all layers: [M, n_dec * 2 * inner]; layer i owns columns [i * 2 * inner, (i + 1) * 2 * inner)
    e->kv_all = b.take(M * (size_t)c.n_dec_layers * 2 * in * sz);
    for (size_t i = 0; i < e->ds.size(); ++i) e->ds[i].kv_ca = e->kv_all ? (char*)e->kv_all + i * 2 * in * sz : nullptr;
    e->dec_rstd_f = (float*)b.take(Md * 4);
    e->dec_hn = b.take(Md * d * sz);
    e->logits = (float*)b.take(Md * Vp * 4);
    e->lse_tok = (float*)b.take(Md * 4);
    e->ce_part = (float*)b.take(Md * (size_t)(Vp / 64) * 2 * 4);
    e->ce_lab = (float*)b.take(Md * 4);
    e->ce_g = (float*)b.take(Md * 4);
  }
  if (with_bwd) {
    const size_t Mx = M > Md ? M : Md;
    e->dres_a = (float*)b.take(Mx * d * 4);
    {   // (also the partial products of the tied head's deterministic split-K input gradient: up to P5_HEAD_SPLITS slices of [Md, d])
      const size_t need = Md * d * 4 * P5_HEAD_SPLITS;
      e->dres_b = (float*)b.take(need > Mx * d * 4 ? need : Mx * d * 4);
    }
    e->d_enc = (float*)b.take(M * d * 4);
    e->Dvec = (float*)b.take((size_t)B * H * (L > T ? L : T) * 4);
    e->rel_partial = (float*)b.take(((size_t)c.n_enc_layers * rel_slots(B, L) + (size_t)c.n_dec_layers * rel_slots(B, T > 0 ? T : 1)) * c.rel_buckets * H * 4);
    e->dw_scratch = (float*)b.take((size_t)(2 * c.n_enc_layers + 3 * c.n_dec_layers + 2) * 1024 * d * 4);
    e->nb_dotbuf = (float*)b.take(Mx * (size_t)((F / 64 > H ? F / 64 : H) + 4) * 4);
    {
      const size_t n0 = M + Md, n1 = M;
      e->emb_idx[0] = (int*)b.take(4 * n0 * 4); e->emb_idx[1] = (int*)b.take(4 * n1 * 4);
      e->emb_csort[0] = (unsigned long long*)b.take(((n0 + P5_EMB_CHUNK - 1) / P5_EMB_CHUNK) * P5_EMB_CHUNK * 8);
      e->emb_csort[1] = (unsigned long long*)b.take(((n1 + P5_EMB_CHUNK - 1) / P5_EMB_CHUNK) * P5_EMB_CHUNK * 8);
      e->emb_part[0] = (float*)b.take(((n0 + P5_EMB_SEG - 1) / P5_EMB_SEG) * 2 * d * 4);
      e->emb_part[1] = (float*)b.take(((n1 + P5_EMB_SEG - 1) / P5_EMB_SEG) * 2 * d * 4);
      e->dres_dec0 = (float*)b.take((Md > 0 ? Md : 1) * d * 4);
    }
    e->dn = b.take(Mx * d * sz);
    e->dO = b.take(Mx * in * sz);
    for (int p = 0; p < P5_NSETS; ++p) {
      e->dy2[p] = b.take(Mx * d * sz);
      e->dqkv2[p] = b.take(Mx * 3 * in * sz);
      e->dkv2[p] = nullptr;
      e->dh2[p] = b.take(Mx * F * sz);
      e->du2[p] = c.gated_gelu ? b.take(Mx * 2 * F * sz) : nullptr;
    }
    e->dy = e->dy2[0]; e->dqkv = e->dqkv2[0]; e->dkv = e->dkv2[0]; e->dh = e->dh2[0]; e->du = e->du2[0];
    e->dkv_all = T > 0 ? b.take(M * (size_t)c.n_dec_layers * 2 * in * sz) : nullptr;     // d(K/V) of all layers, same column layout
    e->dlogits = b.take(Md * Vp * sz);
  }
  return (int64_t)((b.off + 255) & ~(size_t)255);
}

// ---- shared sub-layer forward helpers -----------------------------------------------------------------
// T5LayerNorm of the TRAINING forward folded into the GEMMs around it (bf16 engine with the folded weight copy bound): the producer of a
// residual-stream row leaves d/64 partial sums of squares behind, the consuming projection multiplies the raw row with W diag(ln) and
// scales its accumulator rows by rstd.  The normalised rows the weight gradients need are written by the norm BACKWARD kernel.
static int g_opt_norm_fuse = getenv("P5_NORM_FUSE") ? atoi(getenv("P5_NORM_FUSE")) : 1;
// (the weight gradients must be the deferred, layer-grouped ones: they run after the sub-layer's norm backward has written n)
template <class T> static bool norm_fused(const P5Engine* e) {
  return sizeof(T) == 2 && e->Sf != nullptr && g_opt_norm_fuse != 0 && (e->Md == 0 || wg_all_deferred<T>(e));
}
template <class T> static const T* Wnf(const P5Engine* e, int64_t off) { return (const T*)((const bf16*)e->Sf + off); }

// xout = x_ff + drop(wo(act(wi(norm(x_ff)))));  ssq_next: where to leave the statistics of xout (fused mode)
// gated-GELU FFN (T5 v1.1; HF modeling_t5.py:97-123): the gate runs in the epilogue of the wi GEMM (forward) / of the wo data-gradient GEMM
// (backward) wherever those GEMMs take the wide whole-tile path (p5_gemm5.h: bf16, rows % 256 == 0, >= 160 tiles); elsewhere -- the
// decoder's 512 rows, fp32 parity mode, ragged batches -- the stand-alone p5_gated_gelu_fwd / _bwd kernels.  Option "gate_fuse" 0 = never.
static int g_opt_gate_fuse = getenv("P5_GATE_FUSE") ? atoi(getenv("P5_GATE_FUSE")) : 1;
template <class T> static bool gate_fused(int rows, int N, int K) {
  return sizeof(T) == 2 && g_opt_gate_fuse != 0 && p5l_gemm_gate_ok(rows, N, K, K, K);
}
template <class T>
static int ffn_fwd(P5Engine* e, hipStream_t s, const LayerOff& lo, LayerSave& l, void* xout, int rows, int stack, int li, float* ssq_next) {
  const P5Config& c = e->c;
  const int d = c.d_model, F = c.d_ff;
  const bool nf = norm_fused<T>(e);
  if (!nf) P5_TRY(rmsnorm_fwd<T>(s, l.n_ff, l.rstd_ff, l.x_ff, e->P + lo.ff_ln, rows, d, c.eps, no_drop()));
  if (c.gated_gelu && gate_fused<T>(rows, 2 * F, d)) {
    // h = drop(gelu_new(u0) * u1) in the epilogue of ONE GEMM over [wi_0; wi_1] (rows read gate-interleaved), u kept for the backward
    P5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = nf ? l.x_ff : l.n_ff; g.B = nf ? Wnf<T>(e, lo.wi) : Wc<T>(e, lo.wi); g.C = l.h_ff; g.C2 = l.u_ff;
    g.M = rows; g.N = 2 * F; g.K = d; g.lda = d; g.ldb = d; g.ldc = F; g.ldc2 = 2 * F; g.gate_F = F;
    g.epi = P5_EPI_GELU_GATE; g.alpha = 1.f; g.drop = mk_drop(e, stack, li, 5);
    if (nf) { g.rowss = l.ssq_ff; g.rowss_invd = 1.0f / (float)d; g.rowss_eps = c.eps; g.rowss_nt = d / 64; }
    P5_TRY(launch_gemm<T>(g, s));
  } else if (c.gated_gelu) {
    if (nf) P5_TRY(gemm_nf<T>(s, l.x_ff, d, Wnf<T>(e, lo.wi), d, l.u_ff, 2 * F, rows, 2 * F, d, P5_EPI_STORE, nullptr, 0, no_drop(), l.ssq_ff, c.eps, nullptr, d));
    else P5_TRY(linear_fwd<T>(s, l.n_ff, d, Wc<T>(e, lo.wi), l.u_ff, 2 * F, rows, 2 * F, d));
    const size_t n = (size_t)rows * F;
    P5_LAUNCH((p5_gated_gelu_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s,
              (T*)l.h_ff, (const T*)l.u_ff, rows, F, mk_drop(e, stack, li, 5));
    P5_TRY(P5_KCHECK());
  } else if (nf) {
    P5_TRY(gemm_nf<T>(s, l.x_ff, d, Wnf<T>(e, lo.wi), d, l.h_ff, F, rows, F, d, P5_EPI_RELU_DROP, nullptr, 0, mk_drop(e, stack, li, 5), l.ssq_ff, c.eps, nullptr, d));
  } else {
    P5_TRY(linear_fwd<T>(s, l.n_ff, d, Wc<T>(e, lo.wi), l.h_ff, F, rows, F, d, P5_EPI_RELU_DROP, nullptr, 0, 1.f, 0, mk_drop(e, stack, li, 5)));
  }
  if (nf) return gemm_nf<T>(s, l.h_ff, F, Wc<T>(e, lo.wo), F, xout, d, rows, d, F, P5_EPI_RESID_DROP, l.x_ff, d, mk_drop(e, stack, li, 6), nullptr, 0.f, ssq_next, d);
  return linear_fwd<T>(s, l.h_ff, F, Wc<T>(e, lo.wo), xout, d, rows, d, F, P5_EPI_RESID_DROP, l.x_ff, d, 1.f, 0, mk_drop(e, stack, li, 6));
}

template <class T>
static int encoder_fwd(P5Engine* e, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads, M = e->M;
  const bool nf = norm_fused<T>(e);
  P5_LAUNCH((p5_embed_fwd_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, s, (T*)e->enc_x0, Wc<T>(e, e->off_E), Wc<T>(e, e->off_WW), e->ids,
            e->ww, M, d, mk_drop(e, 0, 0, 0), nf ? e->es[0].ssq_sa : (float*)nullptr);
  P5_TRY(P5_KCHECK());
  for (int i = 0; i < c.n_enc_layers; ++i) {
    const LayerOff& lo = e->enc[i];
    LayerSave& l = e->es[i];
    void* xnext = (i + 1 < c.n_enc_layers) ? e->es[i + 1].x_sa : e->enc_xf;
    float* ssq_next = (i + 1 < c.n_enc_layers) ? e->es[i + 1].ssq_sa : e->ssq_scratch;
    if (nf) {
      P5_TRY(gemm_nf<T>(s, l.x_sa, d, Wnf<T>(e, lo.sa.q), d, l.qkv, 3 * in, M, 3 * in, d, P5_EPI_STORE, nullptr, 0, no_drop(), l.ssq_sa, c.eps, nullptr, d));
    } else {
      P5_TRY(rmsnorm_fwd<T>(s, l.n_sa, l.rstd_sa, l.x_sa, e->P + lo.sa.ln, M, d, c.eps, no_drop()));
      P5_TRY(linear_fwd<T>(s, l.n_sa, d, Wc<T>(e, lo.sa.q), l.qkv, 3 * in, M, 3 * in, d));
    }
    P5AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = l.qkv; a.K = (const T*)l.qkv + in; a.V = (const T*)l.qkv + 2 * in; a.O = l.o_sa; a.lse = l.lse_sa;
    a.rel_table = e->P + e->off_enc_rel; a.bucket_lut = e->lut_enc; a.lut_half = e->lut_half; a.kmask = e->mask;
    a.B = e->B; a.H = H; a.Lq = e->L; a.Lk = e->L; a.ldq = a.ldk = a.ldv = 3 * in; a.ldo = in; a.causal = 0;
    a.drop = mk_drop(e, 0, i, 1);
    a.keep_bits = (g_opt_attn_fwd_head && g_opt_attn_bwd_head && g_opt_attn_keep_bits) ? l.keep_sa : nullptr;      // (both passes on the head-resident kernels, or neither reads / writes the masks)
    P5_TRY(launch_attn_fwd<T>(a, s));
    if (nf) P5_TRY(gemm_nf<T>(s, l.o_sa, in, Wc<T>(e, lo.sa.o), in, l.x_ff, d, M, d, in, P5_EPI_RESID_DROP, l.x_sa, d, mk_drop(e, 0, i, 2), nullptr, 0.f, l.ssq_ff, d));
    else P5_TRY(linear_fwd<T>(s, l.o_sa, in, Wc<T>(e, lo.sa.o), l.x_ff, d, M, d, in, P5_EPI_RESID_DROP, l.x_sa, d, 1.f, 0, mk_drop(e, 0, i, 2)));
    P5_TRY(ffn_fwd<T>(e, s, lo, l, xnext, M, 0, i, ssq_next));
  }
  return rmsnorm_fwd<T>(s, e->enc_out, e->enc_rstd_f, e->enc_xf, e->P + e->off_enc_fln, M, d, c.eps, mk_drop(e, 0, 0, 7));
}

// Logit-free cross-entropy: in bf16 training the tied head's [B*T, V] logits (66 MB fp32 at C2) are never written -- the head
// GEMM's epilogue reduces every 64 columns of a row to (max, sum exp) and picks the label's logit, p5_ce_finish_kernel merges them into the
// row's log-sum-exp and NLL; the backward recomputes the same GEMM and its epilogue writes dlogits directly.  Option "ce_free" 0 = the
// materialised path (fp32 parity mode and shapes the wide kernel does not take always use it).
static int g_opt_ce_free = getenv("P5_CE_FREE") ? atoi(getenv("P5_CE_FREE")) : 1;
template <class T> static bool ce_free(const P5Engine* e) {
  return sizeof(T) == 2 && g_opt_ce_free != 0 && p5l_gemm_ce_ok(e->Md, e->c.vocab_size, e->c.d_model, e->c.d_model, e->c.d_model);
}
template <class T>
static int decoder_fwd(P5Engine* e, hipStream_t s, float* nll_out = nullptr) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads, M = e->M, Md = e->Md;
  const int nd_ = c.n_dec_layers, ldkv = nd_ * 2 * in;
  {
    // K/V projections of the encoder output for every layer: layer 0 on its own (the decoder needs it first), the rest as
    // ONE GEMM over the contiguous weight block; on the side stream when there is one (they only depend on enc_out)
    hipStream_t ks = e->side ? e->side : s;
    if (e->side) fork_to_side(e, s);      // enc_out is ready
    P5_TRY(linear_fwd<T>(ks, e->enc_out, d, Wc<T>(e, e->dec[0].ca.k), e->ds[0].kv_ca, ldkv, M, 2 * in, d));
#ifndef P5_EMU
    if (e->side) hipEventRecord(e->kv_ev[0], e->side);
#endif
    if (nd_ > 1) {
      P5_TRY(linear_fwd<T>(ks, e->enc_out, d, Wc<T>(e, e->dec[1].ca.k), e->ds[1].kv_ca, ldkv, M, (nd_ - 1) * 2 * in, d));
#ifndef P5_EMU
      if (e->side) hipEventRecord(e->kv_ev[1], e->side);
#endif
    }
  }
  P5_LAUNCH(p5_shift_right_kernel, dim3((Md + 255) / 256), dim3(256), 0, s, e->dec_ids, e->labels, e->B, e->T, (int64_t)c.pad_id);
  P5_TRY(P5_KCHECK());
  const bool nf = norm_fused<T>(e);
  P5_LAUNCH((p5_embed_fwd_kernel<T>), dim3((Md + 3) / 4), dim3(256), 0, s, (T*)e->dec_x0, Wc<T>(e, e->off_E), (const T*)nullptr,
            (const int64_t*)e->dec_ids, (const int64_t*)nullptr, Md, d, mk_drop(e, 1, 0, 0), nf ? e->ds[0].ssq_sa : (float*)nullptr);
  P5_TRY(P5_KCHECK());
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const LayerOff& lo = e->dec[i];
    LayerSave& l = e->ds[i];
    void* xnext = (i + 1 < c.n_dec_layers) ? e->ds[i + 1].x_sa : e->dec_xf;
    float* ssq_next = (i + 1 < c.n_dec_layers) ? e->ds[i + 1].ssq_sa : e->ssq_scratch;
    // self attention (causal, unidirectional buckets)
    if (nf) {
      P5_TRY(gemm_nf<T>(s, l.x_sa, d, Wnf<T>(e, lo.sa.q), d, l.qkv, 3 * in, Md, 3 * in, d, P5_EPI_STORE, nullptr, 0, no_drop(), l.ssq_sa, c.eps, nullptr, d));
    } else {
      P5_TRY(rmsnorm_fwd<T>(s, l.n_sa, l.rstd_sa, l.x_sa, e->P + lo.sa.ln, Md, d, c.eps, no_drop()));
      P5_TRY(linear_fwd<T>(s, l.n_sa, d, Wc<T>(e, lo.sa.q), l.qkv, 3 * in, Md, 3 * in, d));
    }
    P5AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = l.qkv; a.K = (const T*)l.qkv + in; a.V = (const T*)l.qkv + 2 * in; a.O = l.o_sa; a.lse = l.lse_sa;
    a.rel_table = e->P + e->off_dec_rel; a.bucket_lut = e->lut_dec; a.lut_half = e->lut_half; a.kmask = nullptr;
    a.B = e->B; a.H = H; a.Lq = e->T; a.Lk = e->T; a.ldq = a.ldk = a.ldv = 3 * in; a.ldo = in; a.causal = 1;
    a.drop = mk_drop(e, 1, i, 1);
    P5_TRY(launch_attn_fwd<T>(a, s));
    if (nf) {
      P5_TRY(gemm_nf<T>(s, l.o_sa, in, Wc<T>(e, lo.sa.o), in, l.x_ca, d, Md, d, in, P5_EPI_RESID_DROP, l.x_sa, d, mk_drop(e, 1, i, 2), nullptr, 0.f, l.ssq_ca, d));
      // cross attention (zero position bias + encoder padding mask)
      P5_TRY(gemm_nf<T>(s, l.x_ca, d, Wnf<T>(e, lo.ca.q), d, l.q_ca, in, Md, in, d, P5_EPI_STORE, nullptr, 0, no_drop(), l.ssq_ca, c.eps, nullptr, d));
    } else {
      P5_TRY(linear_fwd<T>(s, l.o_sa, in, Wc<T>(e, lo.sa.o), l.x_ca, d, Md, d, in, P5_EPI_RESID_DROP, l.x_sa, d, 1.f, 0, mk_drop(e, 1, i, 2)));
      P5_TRY(rmsnorm_fwd<T>(s, l.n_ca, l.rstd_ca, l.x_ca, e->P + lo.ca.ln, Md, d, c.eps, no_drop()));
      P5_TRY(linear_fwd<T>(s, l.n_ca, d, Wc<T>(e, lo.ca.q), l.q_ca, in, Md, in, d));
    }
#ifndef P5_EMU
    if (e->side && (i == 0 || (i == 1 && nd_ > 1))) hipStreamWaitEvent(s, e->kv_ev[i], 0);   // K/V projections were issued on the side stream up front
#endif
    memset(&a, 0, sizeof(a));
    a.Q = l.q_ca; a.K = l.kv_ca; a.V = (const T*)l.kv_ca + in; a.O = l.o_ca; a.lse = l.lse_ca;
    a.rel_table = nullptr; a.bucket_lut = nullptr; a.kmask = e->mask;
    a.B = e->B; a.H = H; a.Lq = e->T; a.Lk = e->L; a.ldq = in; a.ldk = a.ldv = ldkv; a.ldo = in; a.causal = 0;
    a.drop = mk_drop(e, 1, i, 3);
    P5_TRY(launch_attn_fwd<T>(a, s));
    if (nf) P5_TRY(gemm_nf<T>(s, l.o_ca, in, Wc<T>(e, lo.ca.o), in, l.x_ff, d, Md, d, in, P5_EPI_RESID_DROP, l.x_ca, d, mk_drop(e, 1, i, 4), nullptr, 0.f, l.ssq_ff, d));
    else P5_TRY(linear_fwd<T>(s, l.o_ca, in, Wc<T>(e, lo.ca.o), l.x_ff, d, Md, d, in, P5_EPI_RESID_DROP, l.x_ca, d, 1.f, 0, mk_drop(e, 1, i, 4)));
    P5_TRY(ffn_fwd<T>(e, s, lo, l, xnext, Md, 1, i, ssq_next));
  }
  P5_TRY(rmsnorm_fwd<T>(s, e->dec_hn, e->dec_rstd_f, e->dec_xf, e->P + e->off_dec_fln, Md, d, c.eps, mk_drop(e, 1, 0, 7)));
  // tied head, d^-0.5 rescale folded into alpha (P5_T5.py:352-361)
  const float alpha = 1.0f / sqrtf((float)d);
  e->ce_free_fwd = false;
  if (nll_out && ce_free<T>(e)) {
    P5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = e->dec_hn; g.B = Wc<T>(e, e->off_E); g.M = Md; g.N = c.vocab_size; g.K = d; g.lda = d; g.ldb = d;
    g.epi = P5_EPI_CE_STATS; g.alpha = alpha; g.drop = no_drop();
    g.ce_labels = e->labels; g.ce_part = e->ce_part; g.ce_lab = e->ce_lab; g.ce_np = (c.vocab_size + 63) / 64;
    P5_TRY(launch_gemm<T>(g, s));
    P5_LAUNCH((p5_ce_finish_kernel<T>), dim3((Md + 3) / 4), dim3(256), 0, s, nll_out, e->lse_tok, (const float*)e->ce_part, (const float*)e->ce_lab, e->labels, Md, g.ce_np);
    P5_TRY(P5_KCHECK());
    e->ce_free_fwd = true;
    return 0;
  }
  P5_TRY(linear_fwd<T>(s, e->dec_hn, d, Wc<T>(e, e->off_E), e->logits, e->Vp, Md, c.vocab_size, d, P5_EPI_STORE, nullptr, 0, alpha, 1));
  return 0;
}

template <class T>
static int forward_impl(P5Engine* e, float* nll_out, hipStream_t s) {
  P5_TRY(encoder_fwd<T>(e, s));
  P5_TRY(decoder_fwd<T>(e, s, nll_out));
  if (e->ce_free_fwd) return 0;       // (the head's epilogue + p5_ce_finish_kernel have written nll_out and the rows' log-sum-exp)
  P5_LAUNCH((p5_ce_fwd_kernel<T>), dim3(e->Md), dim3(256), 0, s, nll_out, e->lse_tok, (const float*)e->logits, e->labels, e->c.vocab_size, e->Vp);
  return P5_KCHECK();
}

// ---- backward ------------------------------------------------------------------------------------------
static int norm_flush(P5Engine* e, hipStream_t main);
// T5LayerNorm backward in the epilogue of the data-gradient GEMM that produces the norm's input gradient (P5_EPI_NORM_BWD, p5_gemm.h;
// round 6): no `dn` round trip, no stand-alone launch.  Needs the folded-norm forward (statistics as partial sums, n written by the
// backward), the transposed weight copy (K-contiguous operands), whole 128-row tiles, and a producer of dOut that leaves the row sums of
// <dOut, Out> in e->nb_dotbuf: the wide wo data-gradient GEMM (ReLU FFN) or the fused attention backward (encoder self-attention).
static int g_opt_norm_bwd_fuse = getenv("P5_NORM_BWD_FUSE") ? atoi(getenv("P5_NORM_BWD_FUSE")) : 1;
// (option 1: only where the N = d_model data gradient runs on 128-row tiles anyway -- at most one tile per CU, T5-small at the benchmark
//  batch; beyond that the 256x128 kernel's K loop is worth more than the saved pass: a T5-large FFN data gradient is 312 us there against 423.
//  2: wherever the shapes allow)
template <class T> static bool nb_fused(const P5Engine* e, int rows, int K) {
  const int d = e->c.d_model;
  return sizeof(T) == 2 && g_opt_norm_bwd_fuse != 0 && norm_fused<T>(e) && e->St != nullptr && g_opt_dgrad_t && e->nb_dotbuf != nullptr && (K % 64) == 0 &&
         p5l_gemm_normbwd_ok(rows, d, K, K, K) && (g_opt_norm_bwd_fuse >= 2 || (long)(rows / 128) * (d / 128) <= 256);
}
// dOut [rows, K] (T) x W^T copy -> residual gradient out (fp32), dy_next (T, dropout of the preceding sub-layer re-applied), n (T), norm-weight
// gradient partials; everything swap_norm_bwd does around its kernel, around the GEMM instead
template <class T>
static int norm_bwd_gemm(P5Engine* e, hipStream_t s, const void* dOut, int K, int64_t w_off, int rows, const void* x, const float* ssq, void* n_out,
                         int64_t ln_off, P5Drop dnext, int dot_nt) {
  const int d = e->c.d_model;
  float* out = (e->dres_cur == e->dres_a) ? e->dres_b : e->dres_a;
  if (e->dres_out_override) { out = e->dres_out_override; e->dres_out_override = nullptr; }
  end_sublayer_sync(e, s);             // (the epilogue writes dy_next and n: the set they live in must have been read out)
  float* part = e->dw_scratch + (size_t)(e->norm_slot++) * 1024 * d;
  P5_REQUIRE(rows / 64 <= 1024, "norm backward epilogue: more than 1024 partial rows of the norm-weight gradient");
  P5GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dOut; g.B = (const T*)e->St + w_off; g.C = e->dy_next; g.C2 = n_out; g.aux = x;
  g.M = rows; g.N = d; g.K = K; g.lda = K; g.ldb = K; g.ldc = d; g.ldc2 = d; g.ldaux = d;
  g.epi = P5_EPI_NORM_BWD; g.alpha = 1.f; g.drop = dnext; g.splitk = 1;
  g.rowss = ssq; g.rowss_nt = d / 64; g.rowss_invd = 1.0f / (float)d; g.rowss_eps = e->c.eps;
  g.nb_dot = e->nb_dotbuf; g.nb_dot_nt = dot_nt; g.nb_rin = e->dres_cur; g.nb_rout = out; g.nb_w = e->P + ln_off; g.nb_dw = part;
  P5_TRY(launch_gemm<T>(g, s));
  {
    P5ReduceMulti& r = e->nr_pending;
    if (r.n == P5_REDUCE_MULTI_MAX) P5_TRY(norm_flush(e, s));
    r.d = d;
    r.nrows[r.n] = rows / 64;
    r.dst_off[r.n] = ln_off;
    r.part_off[r.n] = part - e->dw_scratch;
    r.n++;
  }
  e->dres_cur = out;
  e->nb_done = true;
  return 0;
}

// nb_*: the T5LayerNorm in front of the sub-layer (its input rows, their statistics, where n goes, the norm weight, the dropout of the
// sub-layer BEFORE it) -- used when the norm backward runs inside the wi data-gradient GEMM; e->nb_done then tells the caller
template <class T>
static int ffn_bwd(P5Engine* e, hipStream_t s, const LayerOff& lo, LayerSave& l, int rows, int stack, int li, P5Drop nb_dnext = no_drop()) {
  // in: e->dy = masked grad wrt the wo output (T), e->dres_cur = grad wrt the sub-layer output (fp32)
  const P5Config& c = e->c;
  const int d = c.d_model, F = c.d_ff;
  e->nb_done = false;
  const float hscale = (e->training && c.dropout > 0.f) ? 1.f / (1.f - c.dropout) : 1.f;
  P5_TRY(linear_wgrad<T>(e, s, e->dy, d, l.h_ff, F, e->G + lo.wo, rows, d, F));
  if (c.gated_gelu) {
    if (sizeof(T) == 2 && e->St && g_opt_dgrad_t && gate_fused<T>(rows, F, d)) {
      // dh = dy Wo never reaches memory: the epilogue of the data-gradient GEMM (on the transposed weight copy) writes du
      P5GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->dy; g.B = (const T*)e->St + lo.wo; g.C = e->du; g.aux = l.u_ff;
      g.M = rows; g.N = F; g.K = d; g.lda = d; g.ldb = d; g.ldc = 2 * F; g.ldaux = 2 * F;
      g.epi = P5_EPI_GELU_GATE_BWD; g.alpha = 1.f; g.drop = mk_drop(e, stack, li, 5);
      P5_TRY(launch_gemm<T>(g, s));
    } else {
      P5_TRY(dgrad_w<T>(e, s, e->dy, d, lo.wo, e->dh, F, rows, d, F));
      const size_t n = (size_t)rows * F;
      P5_LAUNCH((p5_gated_gelu_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s,
                (T*)e->du, (const T*)e->dh, (const T*)l.u_ff, rows, F, mk_drop(e, stack, li, 5));
      P5_TRY(P5_KCHECK());
    }
    P5_TRY(linear_wgrad<T>(e, s, e->du, 2 * F, l.n_ff, d, e->G + lo.wi, rows, 2 * F, d));
    P5_TRY(dgrad_w<T>(e, s, e->du, 2 * F, lo.wi, e->dn, d, rows, 2 * F, d));
  } else if (nb_fused<T>(e, rows, F) && p5l_gemm_gate_ok(rows, F, d, d, d) && l.ssq_ff && l.n_ff) {
    // dh = mask(dy Wo) leaves the row sums of <dh, pre> behind (32 partial sums per row at d_ff = 2048) ...
    P5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = e->dy; g.B = (const T*)e->St + lo.wo; g.C = e->dh; g.aux = l.h_ff;
    g.M = rows; g.N = F; g.K = d; g.lda = d; g.ldb = d; g.ldc = F; g.ldaux = F;
    g.epi = P5_EPI_MASK_POS; g.alpha = hscale; g.drop = no_drop();
    g.ssq_out = e->nb_dotbuf; g.ssq_nt = F / 64;
    P5_TRY(launch_gemm<T>(g, s));
    P5_TRY(linear_wgrad<T>(e, s, e->dh, F, l.n_ff, d, e->G + lo.wi, rows, F, d));
    // ... and the wi data-gradient GEMM finishes the sub-layer: norm backward, residual add, next dy, n for the deferred weight gradient
    P5_TRY(norm_bwd_gemm<T>(e, s, e->dh, F, lo.wi, rows, l.x_ff, l.ssq_ff, l.n_ff, lo.ff_ln, nb_dnext, F / 64));
  } else {
    P5_TRY(dgrad_w<T>(e, s, e->dy, d, lo.wo, e->dh, F, rows, d, F, P5_EPI_MASK_POS, l.h_ff, F, hscale));
    P5_TRY(linear_wgrad<T>(e, s, e->dh, F, l.n_ff, d, e->G + lo.wi, rows, F, d));
    P5_TRY(dgrad_w<T>(e, s, e->dh, F, lo.wi, e->dn, d, rows, F, d));
  }
  return 0;
}

int norm_flush(P5Engine* e, hipStream_t main) {
  P5ReduceMulti& r = e->nr_pending;
  if (r.n == 0) return 0;
  P5_LAUNCH(p5_reduce_rows_multi_kernel, dim3((r.d + 15) / 16, r.n), dim3(256), 0, wgrad_stream(e, main), r, e->G, (const float*)e->dw_scratch);
  r.n = 0;
  return P5_KCHECK();
}

template <class T>
static int swap_norm_bwd(P5Engine* e, hipStream_t s, const void* x, int64_t ln_off, const float* rstd, int rows, P5Drop din, P5Drop dnext,
                         bool has_res_in = true, const float* ssq = nullptr, void* n_out = nullptr) {
  float* out = (e->dres_cur == e->dres_a) ? e->dres_b : e->dres_a;
  if (e->dres_out_override) { out = e->dres_out_override; e->dres_out_override = nullptr; }
  end_sublayer_sync(e, s);
  const int d = e->c.d_model;
  float* part = e->dw_scratch + (size_t)(e->norm_slot++) * 1024 * d;
  int nblk = 0;
  P5_TRY(rmsnorm_bwd<T>(s, out, e->dy_next, e->G + ln_off, e->dn, x, e->P + ln_off, rstd, has_res_in ? e->dres_cur : nullptr, rows, d, din,
                        dnext, part, &nblk, ssq, n_out, e->c.eps));
  // the per-workgroup partials are summed off the critical path, all norms of the stage in one launch (norm_flush)
  {
    P5ReduceMulti& r = e->nr_pending;
    if (r.n == P5_REDUCE_MULTI_MAX) P5_TRY(norm_flush(e, s));
    r.d = d;
    r.nrows[r.n] = nblk;
    r.dst_off[r.n] = ln_off;
    r.part_off[r.n] = part - e->dw_scratch;
    r.n++;
  }
  e->dres_cur = out;
  return 0;
}

template <class T>
static int self_attn_bwd(P5Engine* e, hipStream_t s, const LayerOff& lo, LayerSave& l, int rows, int Lq, bool is_dec, int li, P5Drop nb_dnext = no_drop()) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads;
  e->nb_done = false;
  P5_TRY(linear_wgrad<T>(e, s, e->dy, d, l.o_sa, in, e->G + lo.sa.o, rows, d, in));
  P5_TRY(dgrad_w<T>(e, s, e->dy, d, lo.sa.o, e->dO, in, rows, d, in));
  P5AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = l.qkv; a.K = (const T*)l.qkv + in; a.V = (const T*)l.qkv + 2 * in; a.O = l.o_sa; a.lse = l.lse_sa; a.dO = e->dO;
  a.dQ = e->dqkv; a.dK = (T*)e->dqkv + in; a.dV = (T*)e->dqkv + 2 * in; a.Dvec = e->Dvec;
  a.rel_table = e->P + (is_dec ? e->off_dec_rel : e->off_enc_rel);
  a.rel_stride = c.rel_buckets * H;
  a.rel_copies = c.rel_buckets;
  // every layer's launch stores into its own block of slots (encoder layers first, then decoder layers): nothing to clear, nothing to add to
  // (blocks of exactly as many slots as the launch writes -- p5l_attn_bwd_slots -- so that a stack's blocks are contiguous rows for the reducer)
  const int sl_enc = p5l_attn_bwd_slots(sizeof(T) == 2, e->B, e->L, e->L), sl_dec = p5l_attn_bwd_slots(sizeof(T) == 2, e->B, e->T, e->T);
  a.d_rel_table = e->rel_partial + (is_dec ? ((size_t)c.n_enc_layers * rel_slots(e->B, e->L) + (size_t)li * sl_dec) * a.rel_stride
                                           : (size_t)li * sl_enc * a.rel_stride);
  a.bucket_lut = is_dec ? e->lut_dec : e->lut_enc; a.lut_half = e->lut_half; a.kmask = is_dec ? nullptr : e->mask;
  a.B = e->B; a.H = H; a.Lq = Lq; a.Lk = Lq; a.ldq = a.ldk = a.ldv = 3 * in; a.ldo = in; a.lddo = in;
  a.lddq = a.lddk = a.lddv = 3 * in; a.causal = is_dec ? 1 : 0;
  a.drop = mk_drop(e, is_dec ? 1 : 0, li, 1);
  a.keep_bits = (!is_dec && g_opt_attn_fwd_head && g_opt_attn_bwd_head && g_opt_attn_keep_bits) ? l.keep_sa : nullptr;
  // the attention backward leaves <dq, q> + <dk, k> + <dv, v> per token and head behind, and the qkv data-gradient GEMM runs the T5LayerNorm
  // backward of the sub-layer's input in its epilogue (norm_bwd_gemm above)
  const bool fuse = nb_fused<T>(e, rows, 3 * in) && p5l_attn_bwd_dot_ok(sizeof(T) == 2, a) && l.ssq_sa && l.n_sa;
  if (fuse) a.dot_out = e->nb_dotbuf;
  P5_TRY(launch_attn_bwd<T>(a, s));
  P5_TRY(linear_wgrad<T>(e, s, e->dqkv, 3 * in, l.n_sa, d, e->G + lo.sa.q, rows, 3 * in, d));
  if (fuse) return norm_bwd_gemm<T>(e, s, e->dqkv, 3 * in, lo.sa.q, rows, l.x_sa, l.ssq_sa, l.n_sa, lo.sa.ln, nb_dnext, H);
  P5_TRY(dgrad_w<T>(e, s, e->dqkv, 3 * in, lo.sa.q, e->dn, d, rows, 3 * in, d));
  return 0;
}

template <class T>
static int backward_stage_impl(P5Engine* e, const float* dnll, int stage, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads, M = e->M, Md = e->Md;
  const int nd = c.n_dec_layers, ne = c.n_enc_layers;
  if (stage == 0) {
    if (e->grads_keep) {
      e->wg_epi = P5_EPI_ACCUM;                 // a later micro-batch of an accumulation group: everything adds
    } else if (g_opt_grad_store_first && wg_all_deferred<T>(e)) {
      // (token counts that are multiples of 64 -- every batch of 64 sequences -- put ALL weight gradients on the grouped, storing
      //  path; ragged counts keep the clear-then-add form below: their split-K atomics need a cleared target anyway)
      if (!e->grads_zeroed) P5_TRY(zero_small_grads(e, s));
      e->wg_epi = P5_EPI_STORE;
    } else {
      if (!e->grads_zeroed) hipMemsetAsync(e->G, 0, (size_t)e->n_params * 4, s);     // (the caller may just have done it: zero_grad())
      e->wg_epi = P5_EPI_ACCUM;
    }
    e->grads_zeroed = false;
    e->grads_keep = false;
    e->d_enc_started = false;
    e->sub = -1;
    e->norm_slot = 0;
    e->nr_pending.n = 0;
#ifndef P5_EMU
    for (int i = 0; i < P5_NSETS; ++i) e->set_ev_valid[i] = false;
    e->head_wg_valid = false;
#endif
    begin_sublayer(e);
#ifndef P5_EMU
    if (e->tr_pending && e->tr_ev) { hipStreamWaitEvent(s, e->tr_ev, 0); hipStreamWaitEvent(e->side ? e->side : s, e->tr_ev, 0); }
#endif
    e->tr_pending = false;
#ifndef P5_EMU
    if (e->zg_pending && e->zg_ev) { hipStreamWaitEvent(s, e->zg_ev, 0); if (e->side) hipStreamWaitEvent(e->side, e->zg_ev, 0); }
#endif
    e->zg_pending = false;
    if (!dnll) P5_REQUIRE(e->out_attn, "backward without dnll needs p5_forward_loss (output_attention mask)");
    const float alpha = 1.0f / sqrtf((float)d);
    if (e->ce_free_fwd) {
      // dlogits = (softmax - onehot) * g straight out of the recomputed head GEMM (same operands, same accumulation order: the same logits)
      P5_LAUNCH(p5_ce_gscale_kernel, dim3((Md + 255) / 256), dim3(256), 0, s, e->ce_g, e->labels, dnll, e->out_attn, e->T, 1.0f / (float)e->B, Md);
      P5_TRY(P5_KCHECK());
      P5GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->dec_hn; g.B = Wc<T>(e, e->off_E); g.C = e->dlogits; g.M = Md; g.N = c.vocab_size; g.K = d; g.lda = d; g.ldb = d; g.ldc = e->Vp;
      g.epi = P5_EPI_CE_GRAD; g.alpha = alpha; g.drop = no_drop();
      g.ce_labels = e->labels; g.ce_lse = e->lse_tok; g.ce_g = e->ce_g; g.ce_np = (c.vocab_size + 63) / 64;
      P5_TRY(launch_gemm<T>(g, s));
    } else {
      P5_LAUNCH((p5_ce_bwd_kernel<T>), dim3(Md, Md >= 2048 ? 1 : (Md >= 512 ? 4 : 8)), dim3(256), 0, s, (T*)e->dlogits, (const float*)e->logits, (const float*)e->lse_tok,
                e->labels, dnll, c.vocab_size, e->Vp, e->Vp, e->out_attn, e->T, 1.0f / (float)e->B);
      P5_TRY(P5_KCHECK());
    }
    // dE += alpha * dlogits^T hn ;  dhn = alpha * dlogits E
    P5_TRY(linear_wgrad<T>(e, s, e->dlogits, e->Vp, e->dec_hn, d, e->G + e->off_E, Md, c.vocab_size, d, alpha));
    P5_TRY(wgrad_flush(e, s, true, (g_opt_wgrad_side & 1) != 0));
    {
      // K = vocab is long and M*N small: split-K.  The splits STORE their partial products side by side and one pass sums them in index
      // order and casts -- not fp32 atomics into a cleared buffer: the order atomics land in changes from run to run, the bf16 rounding
      // of dhn then flips in a few elements, and that is the root of the whole backward (measured: two runs of the same step
      // differed by 1e-3 relative in every gradient tensor; with this, only the tensors that are themselves sums of atomics differ,
      // in their last bits).
      // (bf16: reduce over the padded vocabulary Vp = multiple of 64 so that both operands qualify for direct-to-LDS copies;
      //  dlogits columns V..Vp are exact zeros and the rows "E[V..Vp)" are the finite first rows of the next tensor in the arena)
      const int Kv = sizeof(T) == 2 ? e->Vp : c.vocab_size;
      const bool big = Kv >= 16384 || (long)((Md + 127) / 128) * ((d + 127) / 128) >= 512;      // (launch_gemm's tile choice for this problem)
      const long tiles = big ? (long)((Md + 127) / 128) * ((d + 127) / 128) : (long)((Md + 63) / 64) * ((d + 63) / 64);
      const int nst = (Kv + 2 * TT<T>::KCH - 1) / (2 * TT<T>::KCH);         // K-steps of the 128x128 / 64x64 kernels (two K-chunks each)
      int want = (int)(((Kv >= 16384 ? 384 : 768) + tiles - 1) / tiles);
      want = want > P5_HEAD_SPLITS ? P5_HEAD_SPLITS : want;
      want = want > nst / 4 ? (nst / 4 > 0 ? nst / 4 : 1) : want;
      const int per = (nst + want - 1) / want, active = (nst + per - 1) / per;       // splits that have K-steps (the kernel's own arithmetic)
      {
        P5GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = e->dlogits; g.B = Wc<T>(e, e->off_E); g.C = e->dres_b; g.M = Md; g.N = d; g.K = Kv; g.lda = e->Vp; g.ldb = d; g.ldc = d;
        g.a_ks = 0; g.b_ks = 1; g.epi = P5_EPI_ATOMIC; g.c_f32 = 1; g.splitk = want; g.alpha = alpha; g.drop = no_drop();
        g.rowss_invd = 1.0f / (float)Kv; g.c_split_stride = (long long)Md * d;
        P5_TRY(launch_gemm<T>(g, s));
      }
      const size_t n = (size_t)Md * d;
      P5_LAUNCH((p5_reduce_splits_kernel<T>), dim3((unsigned)((n / 8 + 255) / 256 > 4096 ? 4096 : (n / 8 + 255) / 256)), dim3(256), 0, s, (T*)e->dn,
                (const float*)e->dres_b, active, n, n);
      P5_TRY(P5_KCHECK());
    }
    e->dres_cur = e->dres_a;
    P5_TRY(swap_norm_bwd<T>(e, s, e->dec_xf, e->off_dec_fln, e->dec_rstd_f, Md, mk_drop(e, 1, 0, 7), mk_drop(e, 1, nd - 1, 6), false));
    return 0;
  }
  if (stage >= 1 && stage <= nd) {
    const int i = nd - stage;
    const LayerOff& lo = e->dec[i];
    LayerSave& l = e->ds[i];
    begin_sublayer(e);
    P5_TRY(ffn_bwd<T>(e, s, lo, l, Md, 1, i, mk_drop(e, 1, i, 4)));
    const bool nf = norm_fused<T>(e);
    if (!e->nb_done) P5_TRY(swap_norm_bwd<T>(e, s, l.x_ff, lo.ff_ln, l.rstd_ff, Md, no_drop(), mk_drop(e, 1, i, 4), true, nf ? l.ssq_ff : nullptr, nf ? l.n_ff : nullptr));
    e->nb_done = false;
    // cross attention
    begin_sublayer(e);
    P5_TRY(linear_wgrad<T>(e, s, e->dy, d, l.o_ca, in, e->G + lo.ca.o, Md, d, in));
    P5_TRY(dgrad_w<T>(e, s, e->dy, d, lo.ca.o, e->dO, in, Md, d, in));
    P5AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = l.q_ca; a.K = l.kv_ca; a.V = (const T*)l.kv_ca + in; a.O = l.o_ca; a.lse = l.lse_ca; a.dO = e->dO;
    const int ldkv = nd * 2 * in;
    T* dkv_i = (T*)e->dkv_all + (size_t)i * 2 * in;      // this layer's columns of d(K/V); consumed once, after the last layer
    a.dQ = e->dqkv; a.dK = dkv_i; a.dV = dkv_i + in; a.Dvec = e->Dvec;
    a.kmask = e->mask; a.B = e->B; a.H = H; a.Lq = e->T; a.Lk = e->L; a.ldq = in; a.ldk = a.ldv = ldkv; a.ldo = in; a.lddo = in;
    a.lddq = in; a.lddk = a.lddv = ldkv; a.causal = 0; a.drop = mk_drop(e, 1, i, 3);
    P5_TRY(launch_attn_bwd<T>(a, s));
    P5_TRY(linear_wgrad<T>(e, s, e->dqkv, in, l.n_ca, d, e->G + lo.ca.q, Md, in, d));
    P5_TRY(dgrad_w<T>(e, s, e->dqkv, in, lo.ca.q, e->dn, d, Md, in, d));
    P5_TRY(swap_norm_bwd<T>(e, s, l.x_ca, lo.ca.ln, l.rstd_ca, Md, no_drop(), mk_drop(e, 1, i, 2), true, nf ? l.ssq_ca : nullptr, nf ? l.n_ca : nullptr));
    // self attention
    begin_sublayer(e);
    if (i == 0 && g_opt_embed_det) e->dres_out_override = e->dres_dec0;      // gradient of the decoder's embedding rows: consumed by the last stage
    P5_TRY(self_attn_bwd<T>(e, s, lo, l, Md, e->T, true, i, i > 0 ? mk_drop(e, 1, i - 1, 6) : no_drop()));
    if (!e->nb_done) P5_TRY(swap_norm_bwd<T>(e, s, l.x_sa, lo.sa.ln, l.rstd_sa, Md, no_drop(), i > 0 ? mk_drop(e, 1, i - 1, 6) : no_drop(), true, nf ? l.ssq_sa : nullptr,
                            nf ? l.n_sa : nullptr));
    e->nb_done = false;
    return wgrad_flush(e, s, false, (g_opt_wgrad_side & 1) != 0);     // the six weight gradients of the layer: one launch
  }
  if (stage == nd + 1) {
    {
      // every layer's d(K/V) is in place: ONE weight-gradient GEMM for the contiguous K/V block and ONE dgrad for d(enc_out)
      // (K = n_dec * 2 * inner), both off the critical path on the side stream; the encoder backward joins it (stage nd + 2)
      const int ldkv = nd * 2 * in;
      // weight gradient of the K/V block: with the side stream it goes out now; otherwise it stays queued and leaves with the top encoder
      // layer's group (one launch).  d(enc_out) is on the critical path of the encoder backward either way.
      P5_TRY(linear_wgrad<T>(e, s, e->dkv_all, ldkv, e->enc_out, d, e->G + e->dec[0].ca.k, M, ldkv, d));
      const bool side2 = e->side && (g_opt_wgrad_side & 2) != 0;
      if (side2 || !(e->whole_backward || e->stage_pairs)) P5_TRY(wgrad_flush(e, s, false, side2));
      P5_TRY(dgrad_w<T>(e, side2 ? e->side : s, e->dkv_all, ldkv, e->dec[0].ca.k, e->d_enc, d, M, ldkv, d, P5_EPI_STORE,
                             nullptr, 0, 1.f, 1));
    }
    P5_LAUNCH(p5_reduce_rows_kernel, dim3((c.rel_buckets * H + 15) / 16), dim3(256), 0, s, e->G + e->off_dec_rel,
              (const float*)(e->rel_partial + (size_t)c.n_enc_layers * rel_slots(e->B, e->L) * c.rel_buckets * H),
              c.n_dec_layers * p5l_attn_bwd_slots(sizeof(T) == 2, e->B, e->T, e->T), c.rel_buckets * H);
    P5_TRY(P5_KCHECK());
#ifndef P5_EMU
    // shared.weight's gradient: the tied head's weight gradient adds with plain read-modify-writes (side stream, stage 0); the
    // embedding scatters (atomics, from here on) must not run beside it
    if (e->side && e->head_wg_valid) { hipStreamWaitEvent(s, e->head_wg_ev, 0); e->head_wg_valid = false; }
#endif
    if (g_opt_embed_det) return 0;      // (the decoder's lookup rows join the encoder's in the last stage: one owner per row of the tied table)
    P5_LAUNCH((p5_embed_bwd_kernel<T>), dim3((Md + 3) / 4), dim3(256), 0, s, e->G + e->off_E, (float*)nullptr, (const float*)e->dres_cur,
              (const int64_t*)e->dec_ids, (const int64_t*)nullptr, Md, d, mk_drop(e, 1, 0, 0));
    return P5_KCHECK();
  }
  if (stage == nd + 2) {
    join_side(e, s);      // d_enc is accumulated on the side stream
    const size_t n = (size_t)M * d;
    P5_LAUNCH((p5_cast_mask_kernel<T>), dim3((unsigned)((n / 8 + 255) / 256 > 4096 ? 4096 : (n / 8 + 255) / 256)), dim3(256), 0, s, (T*)e->dn,
              (const float*)e->d_enc, n, no_drop());
    P5_TRY(P5_KCHECK());
    e->dres_cur = e->dres_a;
    begin_sublayer(e);
    P5_TRY(swap_norm_bwd<T>(e, s, e->enc_xf, e->off_enc_fln, e->enc_rstd_f, M, mk_drop(e, 0, 0, 7), mk_drop(e, 0, ne - 1, 6), false));
    return 0;
  }
  if (stage >= nd + 3 && stage <= nd + 2 + ne) {
    const int i = ne - (stage - (nd + 2));
    const LayerOff& lo = e->enc[i];
    LayerSave& l = e->es[i];
    begin_sublayer(e);
    P5_TRY(ffn_bwd<T>(e, s, lo, l, M, 0, i, mk_drop(e, 0, i, 2)));
    const bool nf = norm_fused<T>(e);
    if (!e->nb_done) P5_TRY(swap_norm_bwd<T>(e, s, l.x_ff, lo.ff_ln, l.rstd_ff, M, no_drop(), mk_drop(e, 0, i, 2), true, nf ? l.ssq_ff : nullptr, nf ? l.n_ff : nullptr));
    e->nb_done = false;
    begin_sublayer(e);
    P5_TRY(self_attn_bwd<T>(e, s, lo, l, M, e->L, false, i, i > 0 ? mk_drop(e, 0, i - 1, 6) : no_drop()));
    if (!e->nb_done) P5_TRY(swap_norm_bwd<T>(e, s, l.x_sa, lo.sa.ln, l.rstd_sa, M, no_drop(), i > 0 ? mk_drop(e, 0, i - 1, 6) : no_drop(), true, nf ? l.ssq_sa : nullptr,
                            nf ? l.n_sa : nullptr));
    e->nb_done = false;
    // the four weight gradients of the layer: one launch of 192 tiles over all 8192 tokens (or of 2 layers = 384 tiles when the
    // whole backward runs in one call and nobody waits for per-layer gradient ranges)
    // (the top layer's group also carries the cross-attention K/V block queued in the previous stage: 5 problems; then pairs of layers)
    const bool pairs = (e->whole_backward || e->stage_pairs) && g_opt_wgrad_layers > 1;
    if (!pairs || i == 0 || e->wg_pending.size() >= 5) return wgrad_flush(e, s, false, (g_opt_wgrad_side & 2) != 0);
    return 0;
  }
  if (stage == nd + ne + 3) {
    // the tail of the backward: nothing is left to overlap these with except each other -- the whole-word scatter and the
    // relative-bias reduction go to the side stream (p5_backward_stage joins it after this stage), the token scatter stays here
    hipStream_t s2 = s;
#ifndef P5_EMU
    if (e->side) { fork_to_side(e, s); s2 = e->side; }
#endif
    P5_LAUNCH(p5_reduce_rows_kernel, dim3((c.rel_buckets * H + 15) / 16), dim3(256), 0, s2, e->G + e->off_enc_rel,
              (const float*)e->rel_partial, c.n_enc_layers * p5l_attn_bwd_slots(sizeof(T) == 2, e->B, e->L, e->L), c.rel_buckets * H);
    P5_TRY(P5_KCHECK());
    if (g_opt_embed_det) {
      // embedding gradients without atomics (p5_embed.h): tied table over (encoder ids ++ decoder ids), whole-word table over the encoder's
      P5EmbArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.nsets = 2; ea.d = d;
      const int n[2] = {M + Md, M};
      for (int k = 0; k < 2; ++k) {
        P5EmbSet& q = ea.s[k];
        q.key0 = k == 0 ? e->ids : e->ww; q.dres0 = e->dres_cur; q.drop0 = mk_drop(e, 0, 0, 0); q.n0 = M;
        if (k == 0) { q.key1 = e->dec_ids; q.dres1 = e->dres_dec0; q.drop1 = mk_drop(e, 1, 0, 0); q.n1 = Md; }
        q.table = e->G + (k == 0 ? e->off_E : e->off_WW);
        q.perm = e->emb_idx[k]; q.skey = q.perm + n[k]; q.sstart = q.skey + n[k]; q.slen = q.sstart + n[k];
        q.part = e->emb_part[k];
        q.csort = e->emb_csort[k];
      }
      P5_LAUNCH(p5_embed_sortchunk_kernel, dim3((n[0] + P5_EMB_CHUNK - 1) / P5_EMB_CHUNK, 2), dim3(256), 0, s, ea);
      P5_TRY(P5_KCHECK());
      P5_LAUNCH(p5_embed_rank_kernel, dim3((n[0] + 63) / 64, 2), dim3(256), 0, s, ea);
      P5_TRY(P5_KCHECK());
      const int nblk = (n[0] + P5_EMB_SEG - 1) / P5_EMB_SEG;
      P5_LAUNCH(p5_embed_seg_kernel, dim3(nblk, 2), dim3(256), 0, s, ea);
      P5_TRY(P5_KCHECK());
      P5_LAUNCH(p5_embed_fix_kernel, dim3(nblk, 2), dim3(256), 0, s, ea);
      return P5_KCHECK();
    }
    // (a gather of the whole-word table's gradient -- one workgroup per (table row, 256-row slice), matching rows summed in
    //  registers, one atomic per column -- was measured: 4.545 vs 4.493 ms per step; index 0 (every pad) makes a few workgroups long)
    if (s2 != s) {
      P5_LAUNCH((p5_embed_bwd_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, s2, (float*)nullptr, e->G + e->off_WW, (const float*)e->dres_cur,
                e->ids, e->ww, M, d, mk_drop(e, 0, 0, 0));
      P5_TRY(P5_KCHECK());
    }
    P5_LAUNCH((p5_embed_bwd_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, s, e->G + e->off_E, s2 != s ? (float*)nullptr : e->G + e->off_WW,
              (const float*)e->dres_cur, e->ids, e->ww, M, d, mk_drop(e, 0, 0, 0));
    return P5_KCHECK();
  }
  return fail("backward: bad stage");
}

// ---- generation ------------------------------------------------------------------------------------------

static int64_t layout_gen(P5Engine* e, char* base, int B, int L, int K, int max_len, int max_c, int excl_words, GenWs* g) {
  const P5Config& c = e->c;
  const size_t sz = c.dtype == 1 ? 2 : 4;
  const int d = c.d_model, in = e->inner, F = c.d_ff;
  const size_t R = (size_t)B * K;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  // (the encoder's buffers, and the training-layout decoder buffers of the forced-prefix pass: at most P5_FF_MAX positions per user)
  const int ffcap = g_opt_gen_ff && g_opt_decode_v2 ? (max_len - 2 < P5_FF_MAX ? (max_len - 2 > 0 ? max_len - 2 : 0) : P5_FF_MAX) : 0;
  const int64_t enc_bytes = layout_ws(e, base, B, L, ffcap, false);
  Bump b{base, (size_t)enc_bytes};
  GenWs tmp;
  GenWs& w = g ? *g : tmp;
  w.ff_labels = (int64_t*)b.take((size_t)B * (ffcap > 0 ? ffcap : 1) * 8);
  w.ff_nll = (float*)b.take((size_t)B * (ffcap > 0 ? ffcap : 1) * 4);
  {
    char* kv_all = g_opt_decode_v2 ? (char*)b.take((size_t)B * L * c.n_dec_layers * 2 * in * sz) : nullptr;
    w.ldkv = g_opt_decode_v2 ? c.n_dec_layers * 2 * in : 2 * in;
    for (int i = 0; i < c.n_dec_layers; ++i) {
      w.kv_cross[i] = g_opt_decode_v2 ? (base ? (void*)(kv_all + (size_t)i * 2 * in * sz) : nullptr) : b.take((size_t)B * L * 2 * in * sz);
      w.cache[i] = b.take((size_t)max_len * R * 2 * in * sz);
    }
  }
  w.xa = b.take(R * d * sz); w.xb = b.take(R * d * sz); w.n = b.take(R * d * sz);
  w.qkv = b.take(R * 3 * in * sz); w.q = b.take(R * in * sz); w.o = b.take(R * in * sz);
  w.h = b.take(R * (c.gated_gelu ? 3 : 1) * F * sz); w.hn = b.take(R * d * sz);
  w.x32 = (float*)b.take(R * d * 4);
  w.logits = (float*)b.take(R * Vp * 4);
  w.ssq = (float*)b.take((size_t)(3 * c.n_dec_layers + 1) * R * 4);
  w.cand = (float*)b.take(R * (size_t)max_c * 4);
  w.cand_key = (int*)b.take(R * (size_t)max_c * 4);
  w.part_m = (float*)b.take(R * (size_t)((c.vocab_size + 15) / 16) * 4);
  w.part_s = (float*)b.take(R * (size_t)((c.vocab_size + 15) / 16) * 4);
  w.n_cand = (int*)b.take(R * 4);
  w.row_top_score = (float*)b.take(R * (size_t)(2 * K) * 4);
  w.row_top_c = (int*)b.take(R * (size_t)(2 * K) * 4);
  w.mask_copy = (int64_t*)b.take((size_t)B * L * 8);
  w.excluded = (uint32_t*)b.take((size_t)B * (excl_words > 0 ? excl_words : 0) * 4 + 16);
  P5BeamState& st = w.st;
  st.run_seq = (int*)b.take(R * max_len * 4); st.run_seq_next = (int*)b.take(R * max_len * 4);
  st.fin_seq = (int*)b.take(R * max_len * 4); st.fin_seq_next = (int*)b.take(R * max_len * 4);
  st.anc = (int*)b.take(R * max_len * 4); st.anc_next = (int*)b.take(R * max_len * 4);
  st.run_score = (float*)b.take(R * 4); st.run_node = (int*)b.take(R * 4);
  st.fin_score = (float*)b.take(R * 4); st.fin_flag = (int*)b.take(R * 4); st.fin_len = (int*)b.take(R * 4);
  st.unsat = (int*)b.take((size_t)B * 4);
  st.last_tok = (int64_t*)b.take(R * 8);
  st.flags = (int*)b.take(64);
  // latency-shaped decode step: the beam step itself writes the next step's input embeddings into the fp32 residual stream
  st.x32 = g_opt_decode_v2 ? w.x32 : nullptr;
  st.E32 = e->P ? e->P + e->off_E : nullptr;
  st.d = d;
  st.hist = nullptr;
  return (int64_t)((b.off + 255) & ~(size_t)255);
}

template <class T>
static const T* Wf(const P5Engine* e, int64_t off) { return (const T*)e->fold + off; }

// One decoder step over R = B*K rows.  With a fold buffer bound (p5_engine_bind_decode_fold) every RMSNorm except the
// first disappears as a kernel: its weight is folded into the consuming projection (W * ln), its row statistic
// sum(x^2) is accumulated by the residual epilogue that PRODUCES x (one atomic per 64-column tile row) and applied as a
// row scale in the epilogue of the consuming GEMM -- 18 of 73 launches fewer for T5-small.
template <class T>
static int decode_step(P5Engine* e, GenWs& w, int B, int L, int K, int max_len, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads, F = c.d_ff, R = B * K;
  const bool fused = e->fold != nullptr && g_opt_decode_fused;
  P5_LAUNCH((p5_embed_fwd_kernel<T>), dim3((R + 3) / 4), dim3(256), 0, s, (T*)w.xa, Wc<T>(e, e->off_E), (const T*)nullptr,
            (const int64_t*)w.st.last_tok, (const int64_t*)nullptr, R, d, no_drop());
  P5_TRY(P5_KCHECK());
  void* x = w.xa; void* y = w.xb;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  int site = 0;                                   // index of the ssq row that holds sum(x^2) of the current x
  auto ssq_row = [&](int k) { return w.ssq + (size_t)k * R; };
  if (fused) hipMemsetAsync(w.ssq, 0, (size_t)(3 * c.n_dec_layers + 1) * R * 4, s);
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const LayerOff& lo = e->dec[i];
    // ---- self-attention ----
    if (fused && i > 0) {
      P5_TRY(linear_fwd_fused<T>(s, x, d, Wf<T>(e, e->fold_qkv[i]), w.qkv, 3 * in, R, 3 * in, d, P5_EPI_STORE, nullptr, 0, 1.f, 0,
                                 ssq_row(site), c.eps, nullptr));
    } else {
      P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.sa.ln, R, d, c.eps, no_drop()));
      P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.sa.q), w.qkv, 3 * in, R, 3 * in, d));
    }
    P5_LAUNCH((p5_dec_self_attn_kernel<T>), dim3((R * H + 3) / 4), dim3(256), 0, s, (T*)w.o, (const T*)w.qkv, (T*)w.cache[i],
              (const int*)w.st.anc, (const int*)w.st.anc_next, (const float*)(e->P + e->off_dec_rel), e->lut_dec, e->lut_half, R, H,
              (const int*)(w.st.flags + 2), max_len);
    P5_TRY(P5_KCHECK());
    ++site;
    P5_TRY(gemm<T>(s, w.o, in, 0, Wc<T>(e, lo.sa.o), in, 0, y, d, R, d, in, P5_EPI_RESID_DROP, x, d, 1.f, 0, no_drop(), nullptr, 0.f,
                   fused ? ssq_row(site) : nullptr));
    std::swap(x, y);
    // ---- cross-attention ----
    if (fused) {
      P5_TRY(linear_fwd_fused<T>(s, x, d, Wf<T>(e, e->fold_q[i]), w.q, in, R, in, d, P5_EPI_STORE, nullptr, 0, 1.f, 0, ssq_row(site), c.eps,
                                 nullptr));
    } else {
      P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.ca.ln, R, d, c.eps, no_drop()));
      P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.ca.q), w.q, in, R, in, d));
    }
    P5_LAUNCH((p5_dec_cross_attn_kernel<T>), dim3((R * H + 3) / 4), dim3(256), 0, s, (T*)w.o, (const T*)w.q, (const T*)w.kv_cross[i],
              (const int64_t*)w.mask_copy, R, H, K, L);
    P5_TRY(P5_KCHECK());
    ++site;
    P5_TRY(gemm<T>(s, w.o, in, 0, Wc<T>(e, lo.ca.o), in, 0, y, d, R, d, in, P5_EPI_RESID_DROP, x, d, 1.f, 0, no_drop(), nullptr, 0.f,
                   fused ? ssq_row(site) : nullptr));
    std::swap(x, y);
    // ---- feed-forward ----
    if (!fused) P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.ff_ln, R, d, c.eps, no_drop()));
    const void* ffn_in = fused ? x : w.n;
    const T* Wi = fused ? Wf<T>(e, e->fold_wi[i]) : Wc<T>(e, lo.wi);
    const float* rs = fused ? ssq_row(site) : nullptr;
    if (c.gated_gelu) {
      T* u = (T*)w.h + (size_t)R * F;
      P5_TRY(linear_fwd_fused<T>(s, ffn_in, d, Wi, u, 2 * F, R, 2 * F, d, P5_EPI_STORE, nullptr, 0, 1.f, 0, rs, c.eps, nullptr));
      const size_t n = (size_t)R * F;
      P5_LAUNCH((p5_gated_gelu_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (T*)w.h, (const T*)u, R, F, no_drop());
      P5_TRY(P5_KCHECK());
    } else {
      P5_TRY(linear_fwd_fused<T>(s, ffn_in, d, Wi, w.h, F, R, F, d, P5_EPI_RELU_DROP, nullptr, 0, 1.f, 0, rs, c.eps, nullptr));
    }
    ++site;
    P5_TRY(gemm<T>(s, w.h, F, 0, Wc<T>(e, lo.wo), F, 0, y, d, R, d, F, P5_EPI_RESID_DROP, x, d, 1.f, 0, no_drop(), nullptr, 0.f,
                   fused ? ssq_row(site) : nullptr));
    std::swap(x, y);
  }
  if (fused)      // logits = (norm(x) * d^-0.5) E^T  with  E * final_ln folded
    return linear_fwd_fused<T>(s, x, d, Wf<T>(e, e->fold_E), w.logits, Vp, R, c.vocab_size, d, P5_EPI_STORE, nullptr, 0,
                               1.0f / sqrtf((float)d), 1, ssq_row(site), c.eps, nullptr);
  P5_TRY(rmsnorm_fwd<T>(s, w.hn, nullptr, x, e->P + e->off_dec_fln, R, d, c.eps, no_drop()));
  return linear_fwd<T>(s, w.hn, d, Wc<T>(e, e->off_E), w.logits, Vp, R, c.vocab_size, d, P5_EPI_STORE, nullptr, 0, 1.0f / sqrtf((float)d), 1);
}

// ---- latency-shaped decode step (p5_decode2.h) ------------------------------------------------------------
template <class T, int NB, int AMODE>
static int launch_skinny_nb(const P5SkinnyArgs& g, int splits, int need_bytes, hipStream_t s) {
  dim3 grid((g.N + NB - 1) / NB, (g.M + 15) / 16, splits), block(256);
  // (the static LDS size is what bounds the workgroups per CU: 52 KiB -> 3, 80 KiB -> 2 of the 160 KiB)
  if (need_bytes <= 44 * 1024) P5_LAUNCH((p5_skinny_gemm_kernel<T, NB, AMODE, 44>), grid, block, 0, s, g);
  else if (need_bytes <= 52 * 1024) P5_LAUNCH((p5_skinny_gemm_kernel<T, NB, AMODE, 52>), grid, block, 0, s, g);
  else if (need_bytes <= 80 * 1024) P5_LAUNCH((p5_skinny_gemm_kernel<T, NB, AMODE, 80>), grid, block, 0, s, g);
  else if (need_bytes <= 100 * 1024) P5_LAUNCH((p5_skinny_gemm_kernel<T, NB, AMODE, 100>), grid, block, 0, s, g);
  else if (need_bytes <= 140 * 1024) P5_LAUNCH((p5_skinny_gemm_kernel<T, NB, AMODE, 140>), grid, block, 0, s, g);
  else return fail("skinny gemm: tile does not fit the LDS");
  return P5_KCHECK();
}
// C = A W^T over M <= a few hundred rows.  amode 1: A is the fp32 residual stream, normalised (T5LayerNorm, weight `ln`) by the
// consumer itself (K = d_model, no split); amode 0: A is a T matrix, K may be split over workgroups (atomic epilogue only).
template <class T>
static int skinny(hipStream_t s, int amode, const void* A, int lda, const float* ln, const T* W, int ldw, void* C, int ldc, int M, int N, int K,
                  int epi, float alpha, float eps, const int* done) {
  constexpr int EPS = SkT<T>::EPS;
  P5_REQUIRE(K % EPS == 0 && ldw % TT<T>::EPF == 0 && (amode == 1 || lda % TT<T>::EPF == 0), "skinny gemm: K / leading dims");
  P5SkinnyArgs g;
  g.A = A; g.ln = ln; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.epi = epi; g.alpha = alpha; g.eps = eps;
  g.done = done;
  auto need = [&](int nb, int kw) { return (kw / EPS) * (2048 + nb * 128) + (nb < 64 ? 3072 : 0); };   // (+ cross-wave reduction buffer when waves split K)
  int nb = g_opt_dec_nb, kw = K, splits = 1;
  g.kpasses = 1;
  if (amode == 1) {
    if (!nb) nb = need(64, K) <= 80 * 1024 ? 64 : (need(32, K) <= 100 * 1024 ? 32 : 16);
  } else if (epi == P5_SK_RESID) {
    // one writer per output element (bit-reproducible residual update): no K split over workgroups -- narrow column tiles make the
    // workgroups (>= ~200 for 200 rows x 512 columns), and a long reduction is walked in passes of at most 80 KiB of operands
    if (!nb) nb = need(32, K) <= 52 * 1024 ? 32 : 16;
    kw = K;
    while (need(nb, kw) > 80 * 1024 && kw > 2 * EPS) kw = ((kw / 2 + EPS - 1) / EPS) * EPS;
    g.kpasses = (K + kw - 1) / kw;
  } else {
    if (!nb) nb = 64;
    // K range per workgroup: at most 512 (bf16) / 256 (fp32) elements -- one 80 KiB burst -- and enough splits for >= ~200 workgroups
    int cap = g_opt_dec_kw ? g_opt_dec_kw : (sizeof(T) == 2 ? 512 : 256);
    const int tiles = ((N + nb - 1) / nb) * ((M + 15) / 16);
    if (!g_opt_dec_kw && epi == P5_SK_ATOMIC)
      while (cap > 2 * EPS && tiles * ((K + cap - 1) / cap) < 192) cap /= 2;
    if (epi != P5_SK_ATOMIC) cap = K;
    kw = cap < K ? (cap / EPS) * EPS : K;
    splits = (K + kw - 1) / kw;
  }
  g.kw = kw;
  const int nbytes = need(nb, kw);
  if (amode == 1) {
    if (nb == 64) return launch_skinny_nb<T, 64, 1>(g, 1, nbytes, s);
    if (nb == 32) return launch_skinny_nb<T, 32, 1>(g, 1, nbytes, s);
    return launch_skinny_nb<T, 16, 1>(g, 1, nbytes, s);
  }
  if (nb == 64) return launch_skinny_nb<T, 64, 0>(g, splits, nbytes, s);
  if (nb == 32) return launch_skinny_nb<T, 32, 0>(g, splits, nbytes, s);
  return launch_skinny_nb<T, 16, 0>(g, splits, nbytes, s);
}

// streaming head: rows of E kept in LDS per workgroup (0 = materialised-logits path): the largest of 128/64/32/16 whose tile
// fits 128 KiB (bf16 d_model 512 -> 128, 768/1024 -> 64; fp32 512 -> 64, 768/1024 -> 32)
static int head_nv(const P5Engine* e) {
  if (!g_opt_dec_head || !g_opt_decode_v2) return 0;
  // the kernel walks K in units of eight 64-byte chunks (p5_decode2.h): d_model % 256 (bf16) / % 128 (fp32) -- every T5 size;
  // other widths (toy models) take the materialised-logits head
  if (e->c.d_model % (e->c.dtype == 1 ? 256 : 128) != 0) return 0;
  const size_t row = (size_t)e->c.d_model * (e->c.dtype == 1 ? 2 : 4);
  if (g_opt_dec_head_nv) return (size_t)g_opt_dec_head_nv * row <= 128 * 1024 ? g_opt_dec_head_nv : 0;
  for (int nv = 128; nv >= 16; nv >>= 1)
    if ((size_t)nv * row <= 128 * 1024) return nv;
  return 0;
}

// streaming tied head over R rows: per vocabulary tile (max, sum exp) only; the logits the search needs are recomputed by p5_dec_score2_kernel
template <class T>
static int launch_head_lse(P5Engine* e, float* part_m, float* part_s, const void* hn, int R, const int* done, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model;
  const float alpha = 1.0f / sqrtf((float)d);
  const int nv = head_nv(e), V = c.vocab_size, nt = (V + nv - 1) / nv;
  const size_t bytes = (size_t)nv * d * sizeof(T);
#define P5_HEAD(NV, KB) P5_LAUNCH((p5_head_lse_kernel<T, NV, KB>), dim3(nt), dim3(256), 0, s, part_m, part_s, (const T*)hn, Wc<T>(e, e->off_E), R, d, V, alpha, done)
  if (nv == 128) P5_HEAD(128, 128);
  else if (nv == 64 && bytes <= 64 * 1024) P5_HEAD(64, 64);
  else if (nv == 64) P5_HEAD(64, 128);
  else if (nv == 32 && bytes <= 64 * 1024) P5_HEAD(32, 64);
  else if (nv == 32) P5_HEAD(32, 128);
  else P5_HEAD(16, 64);
#undef P5_HEAD
  return P5_KCHECK();
}

template <class T>
static int decode_step2(P5Engine* e, GenWs& w, int B, int L, int K, int max_len, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, H = c.n_heads, F = c.d_ff, R = B * K;
  const int* done = w.st.flags + 4;
  float* x = w.x32;       // holds E32[last token] of every row: written by the previous beam step (or the initial state)
  constexpr int EPS = SkT<T>::EPS;
  // LDS of the fused kernel: q image 2 KiB + scores 8.25 + probabilities 2 x 4.25 + stats + K/V chunk images 2 x 16 KiB + normalised
  // rows + the head's Wq slice + reduction buffer
  const int sk_resid = g_opt_dec_atomic ? P5_SK_ATOMIC : P5_SK_RESID;      // how the o / wo projections update the fp32 residual stream
  const bool fuseq = g_opt_dec_fuseq && sizeof(T) == 2 && (d / EPS) * (2048 + 64 * 128) + 3072 + 19712 + 2 * 16384 <= 136 * 1024;
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const LayerOff& lo = e->dec[i];
    // ---- self-attention: qkv = norm(x) Wqkv^T ; attention over the ancestry-indexed cache ; x += o Wo^T ----
    P5_TRY(skinny<T>(s, 1, x, d, e->P + lo.sa.ln, Wc<T>(e, lo.sa.q), d, w.qkv, 3 * in, R, 3 * in, d, P5_SK_STORE, 1.f, c.eps, done));
    if (max_len <= 64)
      P5_LAUNCH((p5_dec_self_attn2_kernel<T, 8>), dim3((R * H + 3) / 4), dim3(256), 0, s, (T*)w.o, (const T*)w.qkv, (T*)w.cache[i], (const int*)w.st.anc,
                (const int*)w.st.anc_next, (const float*)(e->P + e->off_dec_rel), e->lut_dec, e->lut_half, R, H, (const int*)(w.st.flags + 2), max_len, done);
    else
      P5_LAUNCH((p5_dec_self_attn2_kernel<T, P5_MAX_LEN / 8>), dim3((R * H + 3) / 4), dim3(256), 0, s, (T*)w.o, (const T*)w.qkv, (T*)w.cache[i], (const int*)w.st.anc,
                (const int*)w.st.anc_next, (const float*)(e->P + e->off_dec_rel), e->lut_dec, e->lut_half, R, H, (const int*)(w.st.flags + 2), max_len, done);
    P5_TRY(P5_KCHECK());
    P5_TRY(skinny<T>(s, 0, w.o, in, nullptr, Wc<T>(e, lo.sa.o), in, x, d, R, d, in, sk_resid, 1.f, 0.f, done));
    // ---- cross-attention ----
    P5CrossArgs a;
    a.out = w.o; a.q = w.q; a.x = x; a.ln = e->P + lo.ca.ln; a.Wq = Wc<T>(e, lo.ca.q); a.kv = w.kv_cross[i]; a.mask = w.mask_copy;
    a.R = R; a.H = H; a.Kb = K; a.L = L; a.d = d; a.eps = c.eps; a.done = done; a.ldkv = w.ldkv;
    const dim3 cgrid(B * ((K + 15) / 16), H);
    if (fuseq) {
      if constexpr (sizeof(T) == 2) {
        if (g_opt_dec_cross == 3) P5_LAUNCH((p5_dec_cross_attn3_kernel<T, true, 136>), cgrid, dim3(256), 0, s, a);
        else P5_LAUNCH((p5_dec_cross_attn2_kernel<T, true, 128>), cgrid, dim3(256), 0, s, a);
      }
    } else {
      P5_TRY(skinny<T>(s, 1, x, d, e->P + lo.ca.ln, Wc<T>(e, lo.ca.q), d, w.q, in, R, in, d, P5_SK_STORE, 1.f, c.eps, done));
      if (g_opt_dec_cross == 3) P5_LAUNCH((p5_dec_cross_attn3_kernel<T, false, sizeof(T) == 2 ? 52 : 96>), cgrid, dim3(256), 0, s, a);
      else P5_LAUNCH((p5_dec_cross_attn2_kernel<T, false, sizeof(T) == 2 ? 48 : 80>), cgrid, dim3(256), 0, s, a);
    }
    P5_TRY(P5_KCHECK());
    P5_TRY(skinny<T>(s, 0, w.o, in, nullptr, Wc<T>(e, lo.ca.o), in, x, d, R, d, in, sk_resid, 1.f, 0.f, done));
    // ---- feed-forward ----
    if (c.gated_gelu) {
      T* u = (T*)w.h + (size_t)R * F;
      P5_TRY(skinny<T>(s, 1, x, d, e->P + lo.ff_ln, Wc<T>(e, lo.wi), d, u, 2 * F, R, 2 * F, d, P5_SK_STORE, 1.f, c.eps, done));
      const size_t n = (size_t)R * F;
      P5_LAUNCH((p5_gated_gelu_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (T*)w.h, (const T*)u, R, F, no_drop());
      P5_TRY(P5_KCHECK());
    } else {
      P5_TRY(skinny<T>(s, 1, x, d, e->P + lo.ff_ln, Wc<T>(e, lo.wi), d, w.h, F, R, F, d, P5_SK_RELU, 1.f, c.eps, done));
    }
    P5_TRY(skinny<T>(s, 0, w.h, F, nullptr, Wc<T>(e, lo.wo), F, x, d, R, d, F, sk_resid, 1.f, 0.f, done));
  }
  // logits = (norm(x) * d^-0.5) E^T   (P5_T5.py:352-361)
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  P5_LAUNCH((p5_rmsnorm_f32in_kernel<T>), dim3((R + 3) / 4), dim3(256), 0, s, (T*)w.hn, (const float*)x, (const float*)(e->P + e->off_dec_fln), R, d, c.eps, done);
  P5_TRY(P5_KCHECK());
  const float alpha = 1.0f / sqrtf((float)d);
  if (head_nv(e) > 0) return launch_head_lse<T>(e, w.part_m, w.part_s, w.hn, R, done, s);
  return linear_fwd<T>(s, w.hn, d, Wc<T>(e, e->off_E), w.logits, Vp, R, c.vocab_size, d, P5_EPI_STORE, nullptr, 0, alpha, 1);
}

// ---- the search as three calls (p5_decode_begin / p5_decode_step / p5_decode_finish); p5_generate strings them together ----
template <class T>
static int decode_begin_impl(P5Engine* e, int B, int L, int K, int max_len, const int* child_off, const int* child_tok, const int* child_node,
                             const int* roots, const uint32_t* excluded, int excl_words, int max_c, char* ws, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner, R = B * K;
  GenCtx& g = e->gen;
  g.active = false;
  layout_gen(e, ws, B, L, K, max_len, max_c, excl_words, &g.w);
  GenWs& w = g.w;
  w.st.hist = e->gen_hist_next;       // (p5_generate_draft) one-shot
  e->gen_hist_next = nullptr;
  if (!excluded) excl_words = 0;
  e->B = B; e->L = L; e->T = 0; e->M = B * L; e->Md = 0; e->training = 0;
  if (e->enc_ext_next) {
    // the encoder output of the verification pass (fp32, same weights), rounded once to this engine's dtype: one encoder pass per batch
    // instead of two, and a draft that starts from the better numbers
    const size_t n = (size_t)B * L * d;
    if constexpr (sizeof(T) == 2) {
      P5_LAUNCH((p5_cast_kernel<T>), dim3((unsigned)((n / 8 + 255) / 256 > 2048 ? 2048 : (n / 8 + 255) / 256)), dim3(256), 0, s, (T*)e->enc_out, e->enc_ext_next, n);
      P5_TRY(P5_KCHECK());
    } else {
      hipMemcpyAsync(e->enc_out, e->enc_ext_next, n * 4, hipMemcpyDeviceToDevice, s);
    }
    e->enc_ext_next = nullptr;
  } else {
    P5_TRY(encoder_fwd<T>(e, s));
  }
  // forced-prefix fast-forward (p5_decode.h): the first F steps as one teacher-forced pass of the training-layout decoder over B x F rows
  P5Forced ff = e->ff_next;
  e->ff_next.n = 0;
  const int ffcap = g_opt_gen_ff && g_opt_decode_v2 ? (max_len - 2 < P5_FF_MAX ? max_len - 2 : P5_FF_MAX) : 0;
  if (ff.n > ffcap) ff.n = ffcap > 0 ? ffcap : 0;
  if (ff.n < 2 || roots != nullptr) ff.n = 0;          // (one forced step is what a decode step costs; per-user roots: not a shared prefix)
  const int F = ff.n;
  if (F > 0) {
    P5_LAUNCH(p5_ff_labels_kernel, dim3((B * F + 255) / 256), dim3(256), 0, s, w.ff_labels, ff, B);
    P5_TRY(P5_KCHECK());
    e->T = F; e->Md = B * F; e->labels = w.ff_labels; e->Vp = (c.vocab_size + 63) / 64 * 64;
    P5_TRY(decoder_fwd<T>(e, s));      // (also projects the encoder output to the cross-attention K/V of every layer: e->kv_all)
    P5_LAUNCH((p5_ce_fwd_kernel<T>), dim3(e->Md), dim3(256), 0, s, w.ff_nll, e->lse_tok, (const float*)e->logits, (const int64_t*)w.ff_labels, c.vocab_size, e->Vp);
    P5_TRY(P5_KCHECK());
    for (int i = 0; i < c.n_dec_layers; ++i) {
      w.kv_cross[i] = e->ds[i].kv_ca;                  // same [B*L, n_dec * 2 * inner] block layout as the decode step's own projection
      P5_LAUNCH((p5_ff_cache_kernel<T>), dim3(B * F), dim3(256), 0, s, (T*)w.cache[i], (const T*)e->ds[i].qkv, F, K, R, in);
      P5_TRY(P5_KCHECK());
    }
    e->T = 0; e->Md = 0; e->labels = nullptr;
  } else if (g_opt_decode_v2) {     // K/V projections of every decoder layer in ONE GEMM over the contiguous weight block (build_layout)
    P5_TRY(linear_fwd<T>(s, e->enc_out, d, Wc<T>(e, e->dec[0].ca.k), w.kv_cross[0], w.ldkv, B * L, c.n_dec_layers * 2 * in, d));
  } else {
    for (int i = 0; i < c.n_dec_layers; ++i)
      P5_TRY(linear_fwd<T>(s, e->enc_out, d, Wc<T>(e, e->dec[i].ca.k), w.kv_cross[i], 2 * in, B * L, 2 * in, d));
  }
  P5_LAUNCH(p5_beam_init_kernel, dim3((R * max_len + 255) / 256), dim3(256), 0, s, w.st, child_off, child_tok, child_node, roots, B, K, max_len,
            c.pad_id);
  P5_TRY(P5_KCHECK());
  if (F > 0) {
    P5_LAUNCH(p5_beam_forced_kernel, dim3((R * max_len + 255) / 256), dim3(256), 0, s, w.st, ff, (const float*)w.ff_nll, B, K, max_len, c.pad_id);
    P5_TRY(P5_KCHECK());
  }
  hipMemcpyAsync(w.mask_copy, e->mask, (size_t)B * L * 8, hipMemcpyDeviceToDevice, s);
  if (excl_words > 0) hipMemcpyAsync(w.excluded, excluded, (size_t)B * excl_words * 4, hipMemcpyDeviceToDevice, s);
  g.B = B; g.L = L; g.K = K; g.max_len = max_len; g.max_c = max_c; g.excl_words = excl_words; g.ws = ws;
  g.child_off = child_off; g.child_tok = child_tok; g.child_node = child_node; g.roots = roots;
  g.steps = F; g.steps0 = F;
  g.active = true;
  return 0;
}

template <class T>
static int decode_step_body(P5Engine* e, hipStream_t s) {
  const P5Config& c = e->c;
  GenCtx& g = e->gen;
  GenWs& w = g.w;
  const int d = c.d_model, B = g.B, L = g.L, K = g.K, R = B * K, max_len = g.max_len, max_c = g.max_c, excl_words = g.excl_words;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  const uint32_t* excl = excl_words > 0 ? w.excluded : nullptr;
  const int* done = w.st.flags + 4;
  if (g_opt_decode_v2) P5_TRY(decode_step2<T>(e, w, B, L, K, max_len, s));
  else P5_TRY(decode_step<T>(e, w, B, L, K, max_len, s));
  if (head_nv(e) > 0) {
    const int nv = head_nv(e), nt = (c.vocab_size + nv - 1) / nv;
    P5_LAUNCH((p5_dec_score2_kernel<T>), dim3(R), dim3(256), 0, s, w.row_top_score, w.row_top_c, w.n_cand, w.cand, (const float*)w.part_m,
              (const float*)w.part_s, nt, (const T*)w.hn, Wc<T>(e, e->off_E), d, 1.0f / sqrtf((float)d), (const int*)w.st.run_node,
              (const float*)w.st.run_score, g.child_off, g.child_tok, g.child_node, excl, excl_words, K, max_c, 2 * K, done);
  } else {
    P5_LAUNCH(p5_dec_score_kernel, dim3(R), dim3(256), 0, s, w.cand, w.row_top_score, w.row_top_c, w.n_cand, (const float*)w.logits, Vp,
              c.vocab_size, (const int*)w.st.run_node, (const float*)w.st.run_score, g.child_off, g.child_tok, g.child_node, excl, excl_words, K, max_c,
              2 * K, done);
  }
  P5_TRY(P5_KCHECK());
  P5_LAUNCH(p5_beam_step_kernel, dim3(B), dim3(256), 0, s, w.st, (const float*)w.row_top_score, (const int*)w.row_top_c,
            (const int*)w.n_cand, g.child_off, g.child_tok, g.child_node, max_c, K, max_len, c.eos_id, R);
  return P5_KCHECK();
}

// One beam-search step: the whole decoder over R = B*K rows, the tied head, HF's candidate selection and bookkeeping.  Nothing
// is read back: HF's global stop test (utils.py:3055-3075) is taken on the device by the beam step, which raises flags[4]; a
// step enqueued after that returns at once in every kernel.  From the second step of a call on, the step is replayed from a
// hipGraph captured once per (shape, workspace, trie, option) key.
template <class T>
static int decode_step_impl(P5Engine* e, hipStream_t s) {
  GenCtx& g = e->gen;
  P5_REQUIRE(g.active, "p5_decode_step without p5_decode_begin");
  if (g.steps >= g.max_len - 1) return 0;          // no hypothesis can grow beyond max_len tokens
#ifndef P5_EMU
  static const bool use_graph = !(getenv("P5_NO_GRAPH") && atoi(getenv("P5_NO_GRAPH")));
  GraphKey key;
  memset(&key, 0, sizeof(key));
  key.B = g.B; key.L = g.L; key.K = g.K; key.max_len = g.max_len; key.max_c = g.max_c; key.excl_words = g.excl_words; key.ws = g.ws;
  key.trie = g.child_off; key.trie_tok = g.child_tok; key.trie_node = g.child_node; key.roots = g.roots;
  key.P = e->P; key.S = e->S; key.sz = (int)sizeof(T); key.fold = e->fold; key.hist = g.w.st.hist;
  key.fused = g_opt_decode_fused + 2 * g_opt_decode_v2 + 4 * g_opt_dec_fuseq + 8 * g_opt_dec_nb + 4096 * g_opt_dec_kw + (g_opt_dec_cross << 20) +
              (g_opt_dec_head << 23) + (g_opt_dec_head_nv << 24) + ((g.steps0 > 0 ? 1 : 0) << 30) + (g_opt_dec_atomic << 29);     // (forced-prefix pass: the cross K/V live elsewhere)
  bool have_graph = use_graph && e->gen_graph_exec && memcmp(&key, &e->gen_graph_key, sizeof(key)) == 0;
  if (use_graph && !have_graph && g.steps > g.steps0 && !e->gen_graph_failed) {
    // (never during the very first step: the first launch of a kernel loads its code object, which is not allowed
    // while a stream is capturing)
    if (e->gen_graph_exec) { hipGraphExecDestroy(e->gen_graph_exec); e->gen_graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
      const int rc = decode_step_body<T>(e, s);
      const hipError_t ec = hipStreamEndCapture(s, &graph);
      if (rc == 0 && ec == hipSuccess && graph && hipGraphInstantiate(&e->gen_graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        e->gen_graph_key = key;
        have_graph = true;
      } else {
        e->gen_graph_exec = nullptr;
      }
      if (graph) hipGraphDestroy(graph);
    }
    (void)hipGetLastError();
    if (!have_graph) e->gen_graph_failed = true;
  }
  if (have_graph) {
    if (hipGraphLaunch(e->gen_graph_exec, s) != hipSuccess) return fail("hipGraphLaunch failed");
    g.steps++;
    return 0;
  }
#endif
  P5_TRY(decode_step_body<T>(e, s));
  g.steps++;
  return 0;
}

static int decode_finish_impl(P5Engine* e, int* out_seq, float* out_score, int* out_len, hipStream_t s) {
  GenCtx& g = e->gen;
  P5_REQUIRE(g.active, "p5_decode_finish without p5_decode_begin");
  const int R = g.B * g.K;
  P5_LAUNCH(p5_beam_finalize_kernel, dim3((R * g.max_len + 255) / 256), dim3(256), 0, s, out_seq, out_score, out_len, g.w.st, R, g.max_len);
  P5_TRY(P5_KCHECK());
  g.active = false;
  return 0;
}

// ---- verified generation (p5_verify.h): plan -> encode -> run on an fp32 engine bound to the same master parameters ----
struct VerifyWs {
  P5VerifyPlan pl;
  int64_t* ids; int *node_flat, *depth_flat;
  void *x, *y, *n, *qkv, *q, *o, *h, *hn, *kv_all;
  float *lse, *part_m, *part_s, *logits, *cand, *row_top_score, *zeros;
  int *row_top_c, *n_top, *vrow_a, *vrow_b, *missing;
  P5BeamState st;
};
struct VerifyCtx {
  bool begun = false, encoded = false, planned = false;
  P5Forced ff = {0, {0}, {0}};     // forced prefix of the search being verified (p5_generate_set_forced_prefix on this engine before p5_verify_begin)
  VerifyWs w;
  int B = 0, L = 0, K = 0, Kw = 0, max_len = 0, max_c = 0, excl_words = 0;
  const int *child_off = nullptr, *child_tok = nullptr, *child_node = nullptr, *roots = nullptr;
  char* ws = nullptr;
};
static int verify_cap(int Kw, int max_len) { return (Kw * (max_len - 1) + 1 + 15) / 16 * 16; }

static int64_t layout_verify(P5Engine* e, char* base, int B, int L, int K, int Kw, int max_len, int max_c, int excl_words, VerifyWs* out) {
  const P5Config& c = e->c;
  const size_t sz = c.dtype == 1 ? 2 : 4;
  const int d = c.d_model, in = e->inner, F = c.d_ff, H = c.n_heads;
  const int cap = verify_cap(Kw, max_len);
  const size_t rows = (size_t)B * cap, R = (size_t)B * K;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  const int64_t enc_bytes = layout_ws(e, base, B, L, 0, false);
  Bump b{base, (size_t)enc_bytes};
  VerifyWs tmp;
  VerifyWs& w = out ? *out : tmp;
  w.kv_all = b.take((size_t)B * L * c.n_dec_layers * 2 * in * sz);
  P5VerifyPlan& pl = w.pl;
  pl.cap = cap; pl.max_len = max_len;
  pl.hdr = (int*)b.take(64);
  pl.n_rows = (int*)b.take((size_t)B * 4);
  pl.row_tok = (int*)b.take(rows * 4); pl.row_parent = (int*)b.take(rows * 4); pl.row_depth = (int*)b.take(rows * 4); pl.row_node = (int*)b.take(rows * 4);
  pl.first = (int*)b.take((size_t)B * (max_len + 1) * 4);
  pl.anc = (int*)b.take(rows * max_len * 4);
  w.ids = (int64_t*)b.take(rows * 8); w.node_flat = (int*)b.take(rows * 4); w.depth_flat = (int*)b.take(rows * 4);
  w.x = b.take(rows * d * sz); w.y = b.take(rows * d * sz); w.n = b.take(rows * d * sz);
  w.qkv = b.take(rows * 3 * in * sz); w.q = b.take(rows * in * sz); w.o = b.take(rows * in * sz);
  w.h = b.take(rows * (c.gated_gelu ? 3 : 1) * F * sz); w.hn = b.take(rows * d * sz);
  w.lse = (float*)b.take((size_t)B * H * cap * 4);
  const int nv = head_nv(e);
  const size_t nt = nv > 0 ? (size_t)(c.vocab_size + nv - 1) / nv : 1;
  w.part_m = (float*)b.take(rows * nt * 4); w.part_s = (float*)b.take(rows * nt * 4);
  w.logits = (float*)b.take(rows * Vp * 4);        // (materialised logits: toy widths, and the split-product head -- one throughput GEMM)
  w.cand = (float*)b.take(rows * (size_t)max_c * 4);
  w.row_top_score = (float*)b.take(rows * (size_t)(2 * K) * 4); w.row_top_c = (int*)b.take(rows * (size_t)(2 * K) * 4);
  w.n_top = (int*)b.take(rows * 4); w.zeros = (float*)b.take(rows * 4);
  w.vrow_a = (int*)b.take(R * 4); w.vrow_b = (int*)b.take(R * 4); w.missing = (int*)b.take((size_t)B * 4);
  P5BeamState& st = w.st;
  st.run_seq = (int*)b.take(R * max_len * 4); st.run_seq_next = (int*)b.take(R * max_len * 4);
  st.fin_seq = (int*)b.take(R * max_len * 4); st.fin_seq_next = (int*)b.take(R * max_len * 4);
  st.anc = (int*)b.take(R * max_len * 4); st.anc_next = (int*)b.take(R * max_len * 4);
  st.run_score = (float*)b.take(R * 4); st.run_node = (int*)b.take(R * 4);
  st.fin_score = (float*)b.take(R * 4); st.fin_flag = (int*)b.take(R * 4); st.fin_len = (int*)b.take(R * 4);
  st.unsat = (int*)b.take((size_t)B * 4);
  st.last_tok = (int64_t*)b.take(R * 8);
  st.flags = (int*)b.take(64);
  st.x32 = nullptr; st.E32 = nullptr; st.d = d; st.hist = nullptr;
  (void)excl_words;
  return (int64_t)((b.off + 255) & ~(size_t)255);
}

static int verify_begin_impl(P5Engine* e, int B, int L, int K, int Kw, int max_len, const int* child_off, const int* child_tok,
                             const int* child_node, const int* roots, int max_c, int excl_words, char* ws) {
  if (!e->ver) e->ver = new VerifyCtx();
  VerifyCtx& v = *e->ver;
  v.begun = v.encoded = v.planned = false;
  layout_verify(e, ws, B, L, K, Kw, max_len, max_c, excl_words, &v.w);
  v.B = B; v.L = L; v.K = K; v.Kw = Kw; v.max_len = max_len; v.max_c = max_c; v.excl_words = excl_words; v.ws = ws;
  v.child_off = child_off; v.child_tok = child_tok; v.child_node = child_node; v.roots = roots;
  v.ff = e->ff_next;
  e->ff_next.n = 0;
  if (v.ff.n < 2 || roots != nullptr || !g_opt_gen_ff) v.ff.n = 0;
  if (v.ff.n > max_len - 2) v.ff.n = max_len - 2 > 0 ? max_len - 2 : 0;
  v.begun = true;
  return 0;
}
static int verify_plan_impl(P5Engine* e, const int* hist, hipStream_t s) {
  VerifyCtx& v = *e->ver;
  hipMemsetAsync(v.w.pl.hdr, 0, 64, s);
  P5_LAUNCH(p5_verify_plan_kernel, dim3(v.B), dim3(256), 0, s, v.w.pl, hist, v.child_off, v.child_tok, v.child_node, v.roots, v.B, v.Kw, e->c.pad_id);
  P5_TRY(P5_KCHECK());
  v.planned = true;
  return 0;
}

// the fp32 encoder pass + the cross-attention K/V of every decoder layer (one GEMM over the contiguous weight block); independent of the plan
template <class T>
static int verify_encode_impl(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, hipStream_t s) {
  VerifyCtx& v = *e->ver;
  const P5Config& c = e->c;
  SplitScope split(sizeof(T) == 4 ? g_opt_verify_split : 0);
  layout_ws(e, v.ws, v.B, v.L, 0, false);          // (the encoder's buffers: the head of the verification workspace)
  e->B = v.B; e->L = v.L; e->T = 0; e->M = v.B * v.L; e->Md = 0; e->training = 0;
  e->ids = input_ids; e->ww = whole_word_ids; e->mask = attention_mask; e->labels = nullptr;
  P5_TRY(encoder_fwd<T>(e, s));
  P5_TRY(linear_fwd<T>(s, e->enc_out, c.d_model, Wc<T>(e, e->dec[0].ca.k), v.w.kv_all, c.n_dec_layers * 2 * e->inner, v.B * v.L, c.n_dec_layers * 2 * e->inner, c.d_model));
  v.encoded = true;
  return 0;
}

// SCORE + REPLAY.  PU: rows per user of this pass (>= the plan's largest row count, a multiple of 16, <= cap): the decoder runs on
// [B x PU] rows -- self-attention sees them as B*PU single positions with ancestor lists, cross-attention as B sequences of PU queries.
template <class T>
static int verify_run_impl(P5Engine* e, int PU, const uint32_t* excluded, int* out_seq, float* out_score, int* out_len, int* out_missing, hipStream_t s) {
  VerifyCtx& v = *e->ver;
  VerifyWs& w = v.w;
  const P5Config& c = e->c;
  SplitScope split(sizeof(T) == 4 ? g_opt_verify_split : 0);
  const int d = c.d_model, in = e->inner, H = c.n_heads, F = c.d_ff, B = v.B, K = v.K, max_len = v.max_len;
  const int rows = B * PU, R = B * K, ldkv = c.n_dec_layers * 2 * in;
  const int Vp = (c.vocab_size + 63) / 64 * 64;
  P5_LAUNCH(p5_verify_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, w.ids, w.node_flat, w.depth_flat, w.pl, B, PU, c.pad_id);
  P5_TRY(P5_KCHECK());
  void* x = w.x; void* y = w.y;
  P5_LAUNCH((p5_embed_fwd_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, (T*)x, Wc<T>(e, e->off_E), (const T*)nullptr, (const int64_t*)w.ids,
            (const int64_t*)nullptr, rows, d, no_drop(), (float*)nullptr);
  P5_TRY(P5_KCHECK());
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const LayerOff& lo = e->dec[i];
    // self-attention over the row's own prefix
    P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.sa.ln, rows, d, c.eps, no_drop()));
    P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.sa.q), w.qkv, 3 * in, rows, 3 * in, d));
    P5_LAUNCH((p5_tree_attn_kernel<T>), dim3((rows * H + 3) / 4), dim3(256), 0, s, (T*)w.o, (const T*)w.qkv, w.pl, (const int*)w.depth_flat,
              (const float*)(e->P + e->off_dec_rel), e->lut_dec, e->lut_half, B, PU, H);
    P5_TRY(P5_KCHECK());
    P5_TRY(linear_fwd<T>(s, w.o, in, Wc<T>(e, lo.sa.o), y, d, rows, d, in, P5_EPI_RESID_DROP, x, d));
    std::swap(x, y);
    // cross-attention: the user's PU rows are PU queries against the user's encoder keys (zero position bias, padding mask)
    P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.ca.ln, rows, d, c.eps, no_drop()));
    P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.ca.q), w.q, in, rows, in, d));
    P5AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = w.q; a.K = (const T*)w.kv_all + (size_t)i * 2 * in; a.V = (const T*)w.kv_all + (size_t)i * 2 * in + in; a.O = w.o; a.lse = w.lse;
    a.rel_table = nullptr; a.bucket_lut = nullptr; a.kmask = e->mask;
    a.B = B; a.H = H; a.Lq = PU; a.Lk = v.L; a.ldq = in; a.ldk = a.ldv = ldkv; a.ldo = in; a.causal = 0;
    a.drop = no_drop();
    P5_TRY(launch_attn_fwd<T>(a, s));
    P5_TRY(linear_fwd<T>(s, w.o, in, Wc<T>(e, lo.ca.o), y, d, rows, d, in, P5_EPI_RESID_DROP, x, d));
    std::swap(x, y);
    // feed-forward
    P5_TRY(rmsnorm_fwd<T>(s, w.n, nullptr, x, e->P + lo.ff_ln, rows, d, c.eps, no_drop()));
    if (c.gated_gelu) {
      T* u = (T*)w.h + (size_t)rows * F;
      P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.wi), u, 2 * F, rows, 2 * F, d));
      const size_t n = (size_t)rows * F;
      P5_LAUNCH((p5_gated_gelu_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, (T*)w.h, (const T*)u, rows, F, no_drop());
      P5_TRY(P5_KCHECK());
    } else {
      P5_TRY(linear_fwd<T>(s, w.n, d, Wc<T>(e, lo.wi), w.h, F, rows, F, d, P5_EPI_RELU_DROP));
    }
    P5_TRY(linear_fwd<T>(s, w.h, F, Wc<T>(e, lo.wo), y, d, rows, d, F, P5_EPI_RESID_DROP, x, d));
    std::swap(x, y);
  }
  P5_TRY(rmsnorm_fwd<T>(s, w.hn, nullptr, x, e->P + e->off_dec_fln, rows, d, c.eps, no_drop()));
  // log-sum-exp over the full vocabulary per row + the log-probabilities of the row's trie children (running score 0)
  hipMemsetAsync(w.zeros, 0, (size_t)rows * 4, s);
  const float alpha = 1.0f / sqrtf((float)d);
  const uint32_t* excl = v.excl_words > 0 ? excluded : nullptr;
  // (with split products the tied head is one throughput GEMM into materialised logits: the streaming head multiplies on exact-fp32 MFMAs)
  if (head_nv(e) > 0 && !(sizeof(T) == 4 && g_opt_verify_split)) {
    P5_TRY(launch_head_lse<T>(e, w.part_m, w.part_s, w.hn, rows, nullptr, s));
    const int nv = head_nv(e), nt = (c.vocab_size + nv - 1) / nv;
    P5_LAUNCH((p5_dec_score2_kernel<T>), dim3(rows), dim3(256), 0, s, w.row_top_score, w.row_top_c, w.n_top, w.cand, (const float*)w.part_m,
              (const float*)w.part_s, nt, (const T*)w.hn, Wc<T>(e, e->off_E), d, alpha, (const int*)w.node_flat, (const float*)w.zeros, v.child_off,
              v.child_tok, v.child_node, excl, v.excl_words, PU, v.max_c, 2 * K, (const int*)nullptr);
  } else {
    P5_TRY(linear_fwd<T>(s, w.hn, d, Wc<T>(e, e->off_E), w.logits, Vp, rows, c.vocab_size, d, P5_EPI_STORE, nullptr, 0, alpha, 1));
    P5_LAUNCH(p5_dec_score_kernel, dim3(rows), dim3(256), 0, s, w.cand, w.row_top_score, w.row_top_c, w.n_top, (const float*)w.logits, Vp, c.vocab_size,
              (const int*)w.node_flat, (const float*)w.zeros, v.child_off, v.child_tok, v.child_node, excl, v.excl_words, PU, v.max_c, 2 * K,
              (const int*)nullptr);
  }
  P5_TRY(P5_KCHECK());
  // REPLAY with the real beam width
  P5_LAUNCH(p5_beam_init_kernel, dim3((R * max_len + 255) / 256), dim3(256), 0, s, w.st, v.child_off, v.child_tok, v.child_node, v.roots, B, K, max_len,
            c.pad_id);
  P5_TRY(P5_KCHECK());
  hipMemsetAsync(w.vrow_a, 0, (size_t)R * 4, s);        // every beam starts on row 0 (the start prefix)
  hipMemsetAsync(w.vrow_b, 0, (size_t)R * 4, s);
  hipMemsetAsync(w.missing, 0, (size_t)B * 4, s);
  P5_LAUNCH((p5_verify_range_kernel<T>), dim3(B), dim3(256), 0, s, w.missing, (const T*)w.hn, PU, d);      // (ordered behind the clear on this stream)
  P5_TRY(P5_KCHECK());
  int first = 1;
  if (v.ff.n > 0) {
    // the forced steps of the replay (p5_decode.h): rows 0 .. F-1 are the forced chain (one live prefix per depth), each with ONE child
    P5_LAUNCH(p5_verify_forced_kernel, dim3((R + B * v.ff.n + 255) / 256), dim3(256), 0, s, w.zeros, w.vrow_a, w.vrow_b, (const float*)w.row_top_score, PU, 2 * K,
              B, K, v.ff.n);
    P5_TRY(P5_KCHECK());
    P5_LAUNCH(p5_beam_forced_kernel, dim3((R * max_len + 255) / 256), dim3(256), 0, s, w.st, v.ff, (const float*)w.zeros, B, K, max_len, c.pad_id);
    P5_TRY(P5_KCHECK());
    first = v.ff.n + 1;
  }
  for (int cur_len = first; cur_len < max_len; ++cur_len) {
    P5_LAUNCH(p5_verify_step_kernel, dim3(B), dim3(256), 0, s, w.st, w.pl, (const float*)w.row_top_score, (const int*)w.row_top_c, (const int*)w.n_top, PU,
              v.child_off, v.child_tok, v.child_node, excl, v.excl_words, v.max_c, K, max_len, c.eos_id, R, w.vrow_a, w.vrow_b, w.missing);
    P5_TRY(P5_KCHECK());
  }
  P5_LAUNCH(p5_beam_finalize_kernel, dim3((R * max_len + 255) / 256), dim3(256), 0, s, out_seq, out_score, out_len, w.st, R, max_len);
  P5_TRY(P5_KCHECK());
  hipMemcpyAsync(out_missing, w.missing, (size_t)B * 4, hipMemcpyDeviceToDevice, s);
  v.begun = v.encoded = v.planned = false;
  return 0;
}

// =====================================================================================================
// C ABI
// =====================================================================================================
// out[r, c] = W[r, c] * ln[c]  (fp32 product of the master weights, rounded once to the compute dtype)
template <class T>
__global__ __launch_bounds__(256) void p5_fold_norm_kernel(T* __restrict__ out, const float* __restrict__ W, const float* __restrict__ ln,
                                                          size_t n, int cols) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = from_f<T>(W[i] * ln[i % cols]);
}
template <class T>
static int refresh_fold(P5Engine* e, hipStream_t s) {
  const P5Config& c = e->c;
  const int d = c.d_model, in = e->inner;
  auto fold = [&](int64_t dst, int64_t w_off, int64_t ln_off, int64_t rows) -> int {
    const size_t n = (size_t)rows * d;
    P5_LAUNCH((p5_fold_norm_kernel<T>), dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, s, (T*)e->fold + dst,
              (const float*)(e->P + w_off), (const float*)(e->P + ln_off), n, d);
    return P5_KCHECK();
  };
  for (int i = 0; i < c.n_dec_layers; ++i) {
    const LayerOff& lo = e->dec[i];
    P5_TRY(fold(e->fold_qkv[i], lo.sa.q, lo.sa.ln, 3 * in));       // q, k, v are contiguous in the arena
    P5_TRY(fold(e->fold_q[i], lo.ca.q, lo.ca.ln, in));
    P5_TRY(fold(e->fold_wi[i], lo.wi, lo.ff_ln, (int64_t)(c.gated_gelu ? 2 : 1) * c.d_ff));
  }
  return fold(e->fold_E, e->off_E, e->off_dec_fln, c.vocab_size);
}

// out[w_off + r*d + c] = bf16(P[w_off + r*d + c] * P[ln_off + c]) for every listed weight: 8 rows per workgroup, one launch for the model
struct P5FoldTab {
  int n, d;
  struct D { long long w_off, ln_off; int rows, blk0; } e[160];
};
__global__ __launch_bounds__(256) void p5_fold_rows_kernel(bf16* __restrict__ out, const float* __restrict__ P, P5FoldTab tab) {
  const int b = blockIdx.x;
  int lo = 0, hi = tab.n - 1;             // last descriptor with blk0 <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.e[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const long long w_off = tab.e[lo].w_off, ln_off = tab.e[lo].ln_off;
  const int rows = tab.e[lo].rows, r0 = (b - tab.e[lo].blk0) * 8, d = tab.d;
  for (int i = threadIdx.x; i < 8 * (d / 8); i += 256) {
    const int r = r0 + i / (d / 8), c = (i % (d / 8)) * 8;
    if (r >= rows) continue;
    float w[8], l[8], o[8];
    ldf<8>(P + w_off + (size_t)r * d + c, w);
    ldf<8>(P + ln_off + c, l);
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = w[q] * l[q];
    st16(out + w_off + (size_t)r * d + c, pack16<bf16>(o));
  }
}

// out[c, r] = in[r, c] for every [rows, cols] block of the descriptor table (64 x 64 tiles through LDS, 16-byte row accesses)
struct P5TrDesc { int64_t off; int rows, cols, tile0; int64_t ln_off; int atile0, pad_; };      // ln_off >= 0: the T5LayerNorm weight folded into this block's copy (W diag(ln)), else -1; atile0: first 64 x 256 tile of p5_adamw_tiles_kernel
__global__ __launch_bounds__(256) void p5_transpose_blocks_kernel(bf16* __restrict__ out, const bf16* __restrict__ in, const P5TrDesc* __restrict__ tab,
                                                                 int ndesc) {
  __shared__ unsigned short tile[64][66];
  const int t = blockIdx.x;
  int lo = 0, hi = ndesc - 1;             // last descriptor with tile0 <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const P5TrDesc dsc = tab[lo];
  const int tc = (dsc.cols + 63) / 64;
  const int lt = t - dsc.tile0, r0 = (lt / tc) * 64, c0 = (lt % tc) * 64;
  const unsigned short* src = (const unsigned short*)in + dsc.off;
  unsigned short* dst = (unsigned short*)out + dsc.off;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {        // 64 rows x 8 pieces of 8 elements
    const int r = i >> 3, p = (i & 7) * 8;
    if (r0 + r < dsc.rows && c0 + p < dsc.cols) {           // (cols is a multiple of 8)
      const u32x4 v = ld16(src + (size_t)(r0 + r) * dsc.cols + c0 + p);
#pragma unroll
      for (int q = 0; q < 4; ++q) { tile[r][p + 2 * q] = (unsigned short)(v[q] & 0xFFFF); tile[r][p + 2 * q + 1] = (unsigned short)(v[q] >> 16); }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {        // 64 output rows (= input columns) x 8 pieces
    const int c = i >> 3, p = (i & 7) * 8;
    if (c0 + c < dsc.cols && r0 + p < dsc.rows) {           // (rows is a multiple of 8)
      u32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (unsigned)tile[p + 2 * q][c] | ((unsigned)tile[p + 2 * q + 1][c] << 16);
      st16(dst + (size_t)(c0 + c) * dsc.rows + r0 + p, v);
    }
  }
}

// ---- clip + HF-AdamW over the arena that ALSO writes every bf16 copy the next step reads (round 6) --------------------------------------
// p5_adamw_kernel streams the flat arena and writes the bf16 shadow; the transposed copy W^T (data gradients) and the norm-folded copy
// W diag(ln) (training forward) were then rebuilt by two more launches that re-read what the update had just held in registers
// (p5_transpose_blocks_kernel + p5_fold_rows_kernel: 0.27 GB, ~80 us of the T5-small step).  Here the 2-D layer weights are updated tile by
// tile (the 64 x 64 tiles of the transpose table): a workgroup updates p / m / v of its tile, writes the shadow rows, transposes the bf16
// tile through LDS into W^T, and -- for a projection behind a T5LayerNorm -- multiplies by the NEW norm weight, which it derives itself from
// that weight's old state (64 columns: the same update formula on values nobody has overwritten yet; the norm weights themselves are
// updated by a second, tiny launch ordered behind this one).  Blocks past the tiles update the leading range of the arena (tied embedding,
// whole-word table, relative-bias tables) flat.  Same arithmetic per element as p5_adamw_kernel: bit-identical parameters and copies.
struct P5AdamTileArgs {
  P5AdamArgs a;
  const P5TrDesc* tab;
  int ndesc, ntiles;
  bf16* St; bf16* Sf;           // transposed / norm-folded copies (arena offsets), Sf may be null
  size_t flat_n;                // [0, flat_n): flat update by the blocks behind the tiles
};
__device__ static __forceinline__ float p5_adam_clip_coef(const P5AdamArgs& a, float* sp) {
  float coef = a.grad_scale;
  if (a.sumsq) {
    float t = 0.f;
    for (int i = threadIdx.x; i < P5_SUMSQ_PARTS; i += 256) t += a.sumsq[i];      // (the association of p5_adamw_kernel: same bits)
    sp[threadIdx.x] = t;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
      if ((int)threadIdx.x < st) sp[threadIdx.x] += sp[threadIdx.x + st];
      __syncthreads();
    }
    const float norm = sqrtf(sp[0]) * a.grad_scale;
    const float c = a.max_norm / (norm + 1e-6f);
    coef *= (c < 1.f ? c : 1.f);
  }
  return coef;
}
__device__ static __forceinline__ float p5_adam_update(const P5AdamArgs& a, float coef, float p, float g0, float& m, float& v) {
  const float g = g0 * coef;
  m = a.beta1 * m + a.omb1 * g;
  v = a.beta2 * v + a.omb2 * g * g;
  p = p - a.step_size * (m / (sqrtf(v) + a.eps));
  p = p - a.decay * p;
  return p;
}
#define P5_ADAM_TW 256        // columns of an AdamW tile: 64 rows x 256 columns = 1 KiB contiguous per row and fp32 array (64 x 64 tiles ran at
                              // 0.6 of the flat kernel's bandwidth: 256-byte pieces on a 2-8 KiB stride, profiles/r06_call2_*.txt)
__global__ __launch_bounds__(256) void p5_adamw_tiles_kernel(P5AdamTileArgs t) {
  __shared__ float sp[256];
  __shared__ unsigned short tile[64][P5_ADAM_TW + 2];
  const P5AdamArgs& a = t.a;
  const float coef = p5_adam_clip_coef(a, sp);
  const int b = blockIdx.x;
  if (b >= t.ntiles) {                       // ---- flat leading range
    const size_t nb = gridDim.x - t.ntiles;
    for (size_t i = (size_t)(b - t.ntiles) * 256 + threadIdx.x; i < t.flat_n; i += nb * 256) {
      float m = a.m[i], v = a.v[i];
      const float p = p5_adam_update(a, coef, a.p[i], a.g[i], m, v);
      a.m[i] = m; a.v[i] = v; a.p[i] = p;
      if (a.shadow) ((bf16*)a.shadow)[i] = from_f<bf16>(p);
    }
    return;
  }
  int lo = 0, hi = t.ndesc - 1;              // last descriptor with atile0 <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.tab[mid].atile0 <= b) lo = mid; else hi = mid - 1;
  }
  const P5TrDesc dsc = t.tab[lo];
  const int tc = (dsc.cols + P5_ADAM_TW - 1) / P5_ADAM_TW;
  const int lt = b - dsc.atile0, r0 = (lt / tc) * 64, c0 = (lt % tc) * P5_ADAM_TW;
  const int rq = threadIdx.x >> 5, cl = (threadIdx.x & 31) * 8;      // 32 threads x 8 columns = one 1-KiB row piece; 8 rows per pass
  const int c = c0 + cl;
  const bool col_ok = c < dsc.cols;                                    // (cols % 8 == 0)
  const bool fold = t.Sf != nullptr && dsc.ln_off >= 0 && col_ok;
  float ln_new[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ln_new[e] = 0.f;
  if (fold) {
    // the norm weight AFTER this step, from its state before it (the launch that updates it runs behind this one)
    const size_t l0 = (size_t)dsc.ln_off + c;
    float lp[8], lg[8], lm[8], lv[8];
    ldf<8>(a.p + l0, lp); ldf<8>(a.g + l0, lg); ldf<8>(a.m + l0, lm); ldf<8>(a.v + l0, lv);
#pragma unroll
    for (int e = 0; e < 8; ++e) ln_new[e] = p5_adam_update(a, coef, lp[e], lg[e], lm[e], lv[e]);
  }
#pragma unroll 2
  for (int it = 0; it < 8; ++it) {
    const int r = it * 8 + rq;
    if (col_ok && r0 + r < dsc.rows) {
      const size_t i0 = (size_t)dsc.off + (size_t)(r0 + r) * dsc.cols + c;
      float p[8], g[8], m[8], v[8], o[8];
      ldf<8>(a.p + i0, p); ldf<8>(a.g + i0, g); ldf<8>(a.m + i0, m); ldf<8>(a.v + i0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = p5_adam_update(a, coef, p[e], g[e], m[e], v[e]);
      stf<8>(a.p + i0, o); stf<8>(a.m + i0, m); stf<8>(a.v + i0, v);
      const u32x4 pk = pack16<bf16>(o);
      if (a.shadow) st16((bf16*)a.shadow + i0, pk);
      if (fold) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = o[e] * ln_new[e];
        st16(t.Sf + i0, pack16<bf16>(f));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { tile[r][cl + 2 * q] = (unsigned short)(pk[q] & 0xFFFF); tile[r][cl + 2 * q + 1] = (unsigned short)(pk[q] >> 16); }
    }
  }
  __syncthreads();
  unsigned short* dst = (unsigned short*)t.St + dsc.off;
  for (int i = threadIdx.x; i < P5_ADAM_TW * 8; i += 256) {        // TW output rows (= input columns) x 8 pieces of 8 input rows
    const int cc = i >> 3, p8 = (i & 7) * 8;
    if (c0 + cc < dsc.cols && r0 + p8 < dsc.rows) {                  // (rows % 8 == 0)
      u32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (unsigned)tile[p8 + 2 * q][cc] | ((unsigned)tile[p8 + 2 * q + 1][cc] << 16);
      st16(dst + (size_t)(c0 + cc) * dsc.rows + r0 + p8, v);
    }
  }
}
// the parameters the tiles and the leading range do not cover: the T5LayerNorm weights (one segment each), updated LAST
__global__ __launch_bounds__(256) void p5_adamw_segments_kernel(P5AdamArgs a, P5ZeroTab tab) {
  __shared__ float sp[256];
  const float coef = p5_adam_clip_coef(a, sp);
  const int b = blockIdx.x;
  int lo = 0, hi = tab.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.e[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const int i0 = (b - tab.e[lo].blk0) * 8192, n = tab.e[lo].count;
  for (int j = i0 + threadIdx.x; j < i0 + 8192 && j < n; j += 256) {
    const size_t i = (size_t)tab.e[lo].off + j;
    float m = a.m[i], v = a.v[i];
    const float p = p5_adam_update(a, coef, a.p[i], a.g[i], m, v);
    a.m[i] = m; a.v[i] = v; a.p[i] = p;
    if (a.shadow) ((bf16*)a.shadow)[i] = from_f<bf16>(p);
  }
}

extern "C" {

const char* p5_last_error(void) { return g_p5_err.c_str(); }
int p5_set_option(const char* name, int value) {
  if (!strcmp(name, "gemm_v2")) g_opt_gemm_v2 = value;
  else if (!strcmp(name, "gemm_tile")) g_opt_gemm_tile = value;
  else if (!strcmp(name, "gemm_ksdma")) g_opt_gemm_ksdma = value;
  else if (!strcmp(name, "gemm_ring")) g_opt_gemm_ring = value;
  else if (!strcmp(name, "decode_fused")) g_opt_decode_fused = value;
  else if (!strcmp(name, "gemm_small_ring")) g_opt_gemm_small_ring = value;
  else if (!strcmp(name, "gemm_small_ring_tiles")) g_opt_gemm_small_ring_tiles = value;
  else if (!strcmp(name, "attn_fused")) g_opt_attn_fused = value;
  else if (!strcmp(name, "attn_fwd_wg")) g_opt_attn_fwd_wg = value;
  else if (!strcmp(name, "attn_fwd_head")) g_opt_attn_fwd_head = value;
  else if (!strcmp(name, "attn_bwd_head")) g_opt_attn_bwd_head = value;
  else if (!strcmp(name, "attn_keep_bits")) g_opt_attn_keep_bits = value;
  else if (!strcmp(name, "attn_op_keep_bits")) g_opt_attn_op_keep_bits = value;
  else if (!strcmp(name, "attn_small")) g_opt_attn_small = value;
  else if (!strcmp(name, "gemm_ring32")) g_opt_gemm_ring32 = value;
  else if (!strcmp(name, "gemm_xcd_rect")) g_opt_gemm_xcd_rect = value;
  else if (!strcmp(name, "decode_v2")) g_opt_decode_v2 = value;
  else if (!strcmp(name, "dec_nb")) g_opt_dec_nb = value;
  else if (!strcmp(name, "dec_kw")) g_opt_dec_kw = value;
  else if (!strcmp(name, "dec_fuseq")) g_opt_dec_fuseq = value;
  else if (!strcmp(name, "dgrad_t")) g_opt_dgrad_t = value;
  else if (!strcmp(name, "dec_cross")) g_opt_dec_cross = value;
  else if (!strcmp(name, "dec_head")) g_opt_dec_head = value;
  else if (!strcmp(name, "dec_head_nv")) g_opt_dec_head_nv = value;
  else if (!strcmp(name, "verify_split")) g_opt_verify_split = value;
  else if (!strcmp(name, "gen_ff")) g_opt_gen_ff = value;
  else if (!strcmp(name, "dec_atomic")) g_opt_dec_atomic = value;
  else if (!strcmp(name, "wgrad_group")) g_opt_wgrad_group = value;
  else if (!strcmp(name, "norm_fuse")) g_opt_norm_fuse = value;
  else if (!strcmp(name, "wgrad_wgs")) g_opt_wgrad_wgs = value;
  else if (!strcmp(name, "wgrad_wide")) g_opt_wgrad_wide = value;
  else if (!strcmp(name, "wgrad_side")) g_opt_wgrad_side = value;
  else if (!strcmp(name, "wgrad_layers")) g_opt_wgrad_layers = value;
  else if (!strcmp(name, "gemm_ring_n512")) g_opt_gemm_ring_n512 = value;
  else if (!strcmp(name, "gemm_wide")) g_opt_gemm_wide = value;
  else if (!strcmp(name, "gemm_ws")) g_opt_gemm_ws = value;
  else if (!strcmp(name, "gemm_ws128")) g_opt_gemm_ws128 = value;
  else if (!strcmp(name, "gemm_rect")) g_opt_gemm_rect = value;
  else if (!strcmp(name, "gemm_ws128_min_k")) g_opt_gemm_ws128_min_k = value;
  else if (!strcmp(name, "gate_fuse")) g_opt_gate_fuse = value;
  else if (!strcmp(name, "norm_bwd_blocks")) g_opt_norm_bwd_blocks = value;
  else if (!strcmp(name, "norm_bwd_fuse")) g_opt_norm_bwd_fuse = value;
  else if (!strcmp(name, "adam_tiles")) g_opt_adam_tiles = value;
  else if (!strcmp(name, "ce_free")) g_opt_ce_free = value;
  else if (!strcmp(name, "gemm_wide_min_tiles")) g_opt_gemm_wide_min_tiles = value;
  else if (!strcmp(name, "gemm_ring128_min_k")) g_opt_gemm_ring128_min_k = value;
  else if (!strcmp(name, "gemm_ring128_min_tiles")) g_opt_gemm_ring128_min_tiles = value;
  else if (!strcmp(name, "wgrad_wide_min")) g_opt_wgrad_wide_min = value;
  else if (!strcmp(name, "grad_store_first")) g_opt_grad_store_first = value;
  else if (!strcmp(name, "embed_det")) g_opt_embed_det = value;
  else if (!strcmp(name, "g4_nst")) g_opt_g4_nst = value;
  else if (!strcmp(name, "g4_wgs")) g_opt_g4_wgs = value;
  else return fail("p5_set_option: unknown option");
  return 0;
}
int p5_abi_version(void) { return 5; }    // 2: p5_op_attn_bwd gained d_rel_scratch / rel_buckets (round 4); 3: p5_generate_verified, p5_backward_staged (round 5); 4: p5_allreduce_range / _sum, p5_verify_row_capacity, range flag in p5_verify_run (round 6); 5: P5GemmProblem gained the T5LayerNorm-backward epilogue fields, p5_op_attn_bwd gained dot_out (round 6)
// ---- in-run kernel profiler (p5_device.h P5Prof) ----
int p5_profile_begin(void) {
#ifndef P5_EMU
  P5Prof& p = p5_prof();
  p.recs.clear(); p.used = 0; p.pending_flops = 0.0; p.on = 1;
#endif
  return 0;
}
int p5_profile_end(char* report, int cap) {
  if (report && cap > 0) report[0] = 0;
#ifndef P5_EMU
  P5Prof& p = p5_prof();
  p.on = 0;
  if (p.recs.empty()) return 0;
  // records are made on every stream the library launches on (main, side, weight-gradient): wait for all of them, and refuse to report
  // a table with holes in it
  if (hipDeviceSynchronize() != hipSuccess) return fail("profile: device sync failed");
  int dropped = 0;
  struct Agg { std::string key; int n; double us, flops; };
  std::vector<Agg> agg;
  for (const P5Prof::Rec& r : p.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { ++dropped; continue; }
    char k[512];
    char shp[64] = "";
    if (r.m > 0) snprintf(shp, sizeof(shp), " %dx%dx%d", r.m, r.n, r.k);
    snprintf(k, sizeof(k), "%s%s%s%s grid=(%u,%u,%u) block=%u%s", r.name, r.tag[0] ? " [" : "", r.tag, r.tag[0] ? "]" : "", r.gx, r.gy, r.gz, r.bx, shp);
    size_t i = 0;
    for (; i < agg.size(); ++i) if (agg[i].key == k) break;
    if (i == agg.size()) agg.push_back({k, 0, 0.0, 0.0});
    agg[i].n++; agg[i].us += ms * 1e3; agg[i].flops += r.flops;
  }
  std::string out = "[";
  for (size_t i = 0; i < agg.size(); ++i) {
    std::string key;
    for (char c : agg[i].key) { if (c == '"' || c == '\\') key += '\\'; key += c; }
    char b[768];
    snprintf(b, sizeof(b), "%s{\"kernel\": \"%s\", \"launches\": %d, \"total_us\": %.3f, \"flops\": %.6e}", i ? ", " : "", key.c_str(), agg[i].n, agg[i].us, agg[i].flops);
    out += b;
  }
  out += "]";
  if (dropped) { p.recs.clear(); (void)hipGetLastError(); return fail("profile: " + std::to_string(dropped) + " launch records could not be read"); }
  if (report && cap > 0) {
    if ((int)out.size() + 1 > cap) return fail("profile: report buffer too small");
    memcpy(report, out.c_str(), out.size() + 1);
  }
  p.recs.clear();
#endif
  return 0;
}
int p5_is_emulator(void) {
#ifdef P5_EMU
  return 1;
#else
  return 0;
#endif
}

int p5_engine_create(const P5Config* cfg, P5Engine** out) {
  P5_REQUIRE(cfg && out, "null argument");
  P5_REQUIRE(cfg->d_kv == 64, "d_kv must be 64 (every T5 checkpoint)");
  P5_REQUIRE(cfg->d_model % 64 == 0 && cfg->d_model <= 1024, "d_model must be a multiple of 64, <= 1024");
  P5_REQUIRE(cfg->d_ff % 64 == 0, "d_ff must be a multiple of 64");
  P5_REQUIRE(cfg->n_enc_layers >= 1 && cfg->n_enc_layers <= 64 && cfg->n_dec_layers >= 1 && cfg->n_dec_layers <= 64, "layer count");
  P5_REQUIRE(cfg->dtype == 0 || cfg->dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
  P5_REQUIRE(cfg->rel_buckets >= 2 && cfg->rel_buckets <= 64, "relative_attention_num_buckets must be <= 64");
  P5Engine* e = new P5Engine();
  e->c = *cfg;
  e->inner = cfg->n_heads * cfg->d_kv;
  build_layout(e);
  *out = e;
  return 0;
}
int p5_engine_destroy(P5Engine* e) {
  if (!e) return 0;
#ifndef P5_EMU
  for (int i = 0; i < P5_MAX_STAGED; ++i) {
    if (e->st_ev[i]) hipEventDestroy(e->st_ev[i]);
    if (e->st_ev_side[i]) hipEventDestroy(e->st_ev_side[i]);
  }
  if (e->zg_ev) hipEventDestroy(e->zg_ev);
  if (e->tr_ev) hipEventDestroy(e->tr_ev);
  for (int i = 0; i < 3; ++i) if (e->gen_ev[i]) hipEventDestroy(e->gen_ev[i]);
  if (e->side_events) {
    for (int i = 0; i < 32; ++i) hipEventDestroy(e->ev_pool[i]);
    for (int i = 0; i < P5_NSETS; ++i) hipEventDestroy(e->set_ev[i]);
    hipEventDestroy(e->head_wg_ev);
    for (int i = 0; i < 64; ++i) hipEventDestroy(e->kv_ev[i]);
  }
#endif
  delete e->ver;
  delete e;
  return 0;
}
int64_t p5_param_count(const P5Engine* e) { return e->n_params; }
int p5_param_table(const P5Engine* e, int idx, char* name, int name_cap, int64_t* offset, int* rows, int* cols) {
  if (idx < 0 || idx >= (int)e->table.size()) return 1;
  const ParamInfo& p = e->table[idx];
  snprintf(name, name_cap, "%s", p.name.c_str());
  *offset = p.off; *rows = p.rows; *cols = p.cols;
  return 0;
}
int p5_engine_bind(P5Engine* e, float* params, float* grads, void* shadow, const int* lut_enc, const int* lut_dec, int lut_half,
                   uint32_t* rng_state) {
  P5_REQUIRE(params && lut_enc && lut_dec, "null argument");
  P5_REQUIRE(e->c.dtype == 0 || shadow, "bf16 mode needs a shadow arena");
  P5_REQUIRE(lut_half >= 511, "bucket LUT must cover |rel| <= 511");
  e->P = params; e->G = grads; e->S = shadow; e->lut_enc = lut_enc; e->lut_dec = lut_dec; e->lut_half = lut_half; e->rng = rng_state;
  return 0;
}
int64_t p5_decode_fold_count(const P5Engine* e) { return e->fold_count; }
int p5_engine_bind_decode_fold(P5Engine* e, void* buf) { e->fold = buf; return 0; }
int p5_refresh_decode_fold(P5Engine* e, void* stream) {
  P5_REQUIRE(e->P && e->fold, "engine / fold buffer not bound");
  return e->c.dtype == 1 ? refresh_fold<bf16>(e, (hipStream_t)stream) : refresh_fold<float>(e, (hipStream_t)stream);
}

// buffer layout: [W^T copy: n_params bf16][256-byte aligned descriptor table][256-byte aligned folded copy W diag(ln): n_params bf16]
static size_t tr_table_off(const P5Engine* e) { return ((size_t)e->n_params * 2 + 255) & ~(size_t)255; }
static size_t tr_fold_off(const P5Engine* e) { return (tr_table_off(e) + e->tr_list.size() * sizeof(P5TrDesc) + 255) & ~(size_t)255; }
int64_t p5_transposed_bytes(const P5Engine* e) { return (int64_t)(tr_fold_off(e) + (size_t)e->n_params * 2 + 256); }
int p5_engine_bind_transposed(P5Engine* e, void* buf, void* stream) {
  e->St = buf;
  e->Sf = nullptr;
  if (!buf) return 0;
  P5_REQUIRE(e->c.dtype == 1, "the transposed weight copy serves the bf16 mode only");
  // the descriptor table lives behind the copy itself (the library allocates nothing)
  std::vector<P5TrDesc> tab;
  int at0 = 0;
  for (auto& t : e->tr_list) {
    P5TrDesc q;
    memset(&q, 0, sizeof(q));
    q.off = t.off; q.rows = t.rows; q.cols = t.cols; q.tile0 = t.tile0; q.ln_off = t.ln_off; q.atile0 = at0;
    at0 += ((t.rows + 63) / 64) * ((t.cols + 255) / 256);
    tab.push_back(q);
  }
  e->adam_tiles = at0;
  char* at = (char*)buf + tr_table_off(e);
  e->Sf = (char*)buf + tr_fold_off(e);
#ifndef P5_EMU
  P5_REQUIRE(hipMemcpyAsync(at, tab.data(), tab.size() * sizeof(P5TrDesc), hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess, "descriptor upload");
  hipStreamSynchronize((hipStream_t)stream);      // (tab is a host temporary; one-time set-up call)
  if (!e->tr_ev) hipEventCreateWithFlags(&e->tr_ev, hipEventDisableTiming);
#else
  memcpy(at, tab.data(), tab.size() * sizeof(P5TrDesc));
#endif
  return 0;
}
// clip + AdamW over the ENGINE's arenas; with the transposed / folded copies bound (bf16 training) the update writes them too and the
// caller need not call p5_refresh_transposed.  *copies_fresh = 1 when it did.  Otherwise identical to p5_adamw_step(P, G, m, v, S, n, ...).
int p5_engine_adamw_step(P5Engine* e, float* m, float* v, const float* sumsq, double max_norm, double grad_scale, double lr, double beta1,
                         double beta2, double eps, double weight_decay, int step_t, int* copies_fresh, void* stream) {
  P5_REQUIRE(e->P && e->G && m && v, "p5_engine_adamw_step: arenas not bound");
  if (copies_fresh) *copies_fresh = 0;
  const bool tiles = g_opt_adam_tiles != 0 && e->c.dtype == 1 && e->St && e->S && !e->side;
  if (!tiles) return p5_adamw_step(e->P, e->G, m, v, e->S, e->n_params, sumsq, max_norm, grad_scale, lr, beta1, beta2, eps, weight_decay, step_t, stream);
  P5AdamTileArgs t;
  P5AdamArgs& a = t.a;
  a.p = e->P; a.g = e->G; a.m = m; a.v = v; a.shadow = e->S; a.sumsq = sumsq; a.n = (size_t)e->n_params;
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps; a.max_norm = (float)max_norm; a.grad_scale = (float)grad_scale;
  const double bc1 = 1.0 - pow(beta1, (double)step_t), bc2 = 1.0 - pow(beta2, (double)step_t);
  a.step_size = (float)(lr * sqrt(bc2) / bc1);
  a.decay = (float)(lr * weight_decay);
  t.tab = (const P5TrDesc*)((char*)e->St + tr_table_off(e));
  t.ndesc = (int)e->tr_list.size(); t.ntiles = e->adam_tiles;
  t.St = (bf16*)e->St; t.Sf = (bf16*)e->Sf;
  t.flat_n = (size_t)e->off_small_end;
  hipStream_t s = (hipStream_t)stream;
  P5_LAUNCH(p5_adamw_tiles_kernel, dim3(e->adam_tiles + 1024), dim3(256), 0, s, t);
  P5_TRY(P5_KCHECK());
  {
    const P5Config& c = e->c;
    P5ZeroTab tab;
    tab.n = 0;
    int blocks = 0;
    auto add = [&](int64_t off, int64_t count) {
      P5ZeroTab::D& q = tab.e[tab.n++];
      q.off = off; q.count = (int)count; q.blk0 = blocks;
      blocks += (int)((count + 8191) / 8192);
    };
    P5_REQUIRE(2 * c.n_enc_layers + 3 * c.n_dec_layers + 2 < 200, "p5_engine_adamw_step: too many norm segments");
    for (const LayerOff& l : e->enc) { add(l.sa.ln, c.d_model); add(l.ff_ln, c.d_model); }
    for (const LayerOff& l : e->dec) { add(l.sa.ln, c.d_model); add(l.ca.ln, c.d_model); add(l.ff_ln, c.d_model); }
    add(e->off_enc_fln, c.d_model);
    add(e->off_dec_fln, c.d_model);
    P5_LAUNCH(p5_adamw_segments_kernel, dim3(blocks), dim3(256), 0, s, a, tab);
    P5_TRY(P5_KCHECK());
  }
  e->tr_pending = false;       // (same stream as the next forward / backward: nothing to wait for)
  if (copies_fresh) *copies_fresh = 1;
  return 0;
}
// Rebuild the transposed copy from the bf16 shadow (call after every parameter update, e.g. right after p5_adamw_step).  With a
// side stream bound it runs there, behind everything enqueued on `stream` so far; the next backward waits for it -- the
// forward in between does not, so the ~0.2 GB of copies overlap it.
int p5_refresh_transposed(P5Engine* e, void* stream) {
  P5_REQUIRE(e->St && e->S, "transposed copy / shadow not bound");
  hipStream_t main = (hipStream_t)stream;
  hipStream_t s = main;
#ifndef P5_EMU
  if (e->side) { fork_to_side(e, main); s = e->side; }
#endif
  const P5TrDesc* tab = (const P5TrDesc*)((char*)e->St + tr_table_off(e));
  P5_LAUNCH(p5_transpose_blocks_kernel, dim3(e->tr_tiles), dim3(256), 0, s, (bf16*)e->St, (const bf16*)e->S, tab, (int)e->tr_list.size());
  P5_TRY(P5_KCHECK());
#ifndef P5_EMU
  if (e->tr_ev) hipEventRecord(e->tr_ev, s);       // (W^T is first read by the backward; the folded copy below by the very next forward)
#endif
  s = main;          // the folded copy is read by the very next forward: always on the caller's stream
  if (e->Sf) {
    // W diag(ln) of every projection that consumes a T5LayerNorm output (q/k/v, wi, cross-attention q), from the fp32 masters, at the
    // weights' own arena offsets -- what the training forward multiplies the raw residual stream with (norm_fused)
    const P5Config& c = e->c;
    const int d = c.d_model, in = e->inner;
    const int wi_rows = (c.gated_gelu ? 2 : 1) * c.d_ff;
    P5FoldTab tab;
    tab.n = 0; tab.d = d;
    int blocks = 0;
    auto add = [&](int64_t w_off, int64_t ln_off, int rows) {
      P5FoldTab::D& q = tab.e[tab.n++];
      q.w_off = w_off; q.ln_off = ln_off; q.rows = rows; q.blk0 = blocks;
      blocks += (rows + 7) / 8;
    };
    P5_REQUIRE(2 * c.n_enc_layers + 3 * c.n_dec_layers <= 160, "fold table");
    for (int i = 0; i < c.n_enc_layers; ++i) { add(e->enc[i].sa.q, e->enc[i].sa.ln, 3 * in); add(e->enc[i].wi, e->enc[i].ff_ln, wi_rows); }
    for (int i = 0; i < c.n_dec_layers; ++i) {
      add(e->dec[i].sa.q, e->dec[i].sa.ln, 3 * in); add(e->dec[i].ca.q, e->dec[i].ca.ln, in); add(e->dec[i].wi, e->dec[i].ff_ln, wi_rows);
    }
    P5_LAUNCH(p5_fold_rows_kernel, dim3(blocks), dim3(256), 0, s, (bf16*)e->Sf, (const float*)e->P, tab);
    P5_TRY(P5_KCHECK());
  }
  e->tr_pending = true;
  return 0;
}

int p5_refresh_shadow(P5Engine* e, void* stream) {
  if (e->c.dtype != 1) return 0;
  const size_t n = (size_t)e->n_params;
  P5_LAUNCH((p5_cast_kernel<bf16>), dim3(2048), dim3(256), 0, (hipStream_t)stream, (bf16*)e->S, (const float*)e->P, n);
  return P5_KCHECK();
}

int64_t p5_train_workspace_bytes(const P5Engine* e, int B, int L, int T) {
  P5Engine tmp = *e;
  return layout_ws(&tmp, nullptr, B, L, T, true);
}

int p5_forward(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, const int64_t* labels,
               int B, int L, int T, int training, float* nll_out, void* ws, int64_t ws_bytes, void* stream) {
  P5_REQUIRE(e->P, "engine not bound");
  P5_REQUIRE(B >= 1 && L >= 1 && L <= 512 && T >= 1 && T <= 512, "shape limits: 1 <= L,T <= 512");
  const int64_t need = layout_ws(e, (char*)ws, B, L, T, true);
  P5_REQUIRE(ws_bytes >= need, "workspace too small");
  P5_REQUIRE(((uintptr_t)ws % 256) == 0, "workspace must be 256-byte aligned");
  e->B = B; e->L = L; e->T = T; e->M = B * L; e->Md = B * T; e->training = training;
  e->Vp = (e->c.vocab_size + 63) / 64 * 64;
  e->ids = input_ids; e->ww = whole_word_ids; e->mask = attention_mask; e->labels = labels; e->out_attn = nullptr;
  if (training && e->c.dropout > 0.f) P5_REQUIRE(e->rng, "training with dropout needs rng_state");
  return e->c.dtype == 1 ? forward_impl<bf16>(e, nll_out, (hipStream_t)stream) : forward_impl<float>(e, nll_out, (hipStream_t)stream);
}
int p5_forward_loss(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, const int64_t* labels,
                    const int64_t* output_attention, int B, int L, int T, int training, float* nll_out, float* loss_out, void* ws, int64_t ws_bytes,
                    void* stream) {
  P5_REQUIRE(output_attention && loss_out, "null argument");
  P5_TRY(p5_forward(e, input_ids, whole_word_ids, attention_mask, labels, B, L, T, training, nll_out, ws, ws_bytes, stream));
  e->out_attn = output_attention;
  P5_LAUNCH(p5_masked_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_out, (const float*)nll_out, output_attention, B, T);
  return P5_KCHECK();
}
int p5_backward_num_stages(const P5Engine* e) { return e->c.n_dec_layers + e->c.n_enc_layers + 4; }
int p5_backward_stage(P5Engine* e, const float* dnll, int stage, void* stream) {
  P5_REQUIRE(e->G, "no gradient arena bound");
  P5_REQUIRE(e->Md > 0, "p5_forward must run first");
  P5_TRY(e->c.dtype == 1 ? backward_stage_impl<bf16>(e, dnll, stage, (hipStream_t)stream)
                         : backward_stage_impl<float>(e, dnll, stage, (hipStream_t)stream));
  const bool last_stage = stage == p5_backward_num_stages(e) - 1;
  if (!e->whole_backward || last_stage) {
    // stage-by-stage callers: norm partials never stay pending; weight gradients only while a two-layer encoder group (or the
    // cross-attention K/V block that leaves with the top group) is still filling up -- p5_backward_final_range reports a range only
    // once everything inside it has been launched
    if (!e->stage_pairs || last_stage) P5_TRY(wgrad_flush(e, (hipStream_t)stream, false, (g_opt_wgrad_side & 2) != 0));
    P5_TRY(norm_flush(e, (hipStream_t)stream));
  }
  {
    int64_t b = 0, en = 0;
    p5_backward_stage_range(e, stage, &b, &en);
    if (stage == 0) { e->fin_b = e->fin_e = 0; }
    if (en > b) {
      if (e->fin_e <= e->fin_b) { e->fin_b = b; e->fin_e = en; }
      else { e->fin_b = b < e->fin_b ? b : e->fin_b; e->fin_e = en > e->fin_e ? en : e->fin_e; }
    }
    e->rep_b = e->rep_e = 0;
    if (e->wg_pending.empty() && e->fin_e > e->fin_b) { e->rep_b = e->fin_b; e->rep_e = e->fin_e; e->fin_b = e->fin_e = 0; }
  }
  // the side stream is now ordered after this stage's main-stream work (a bucket all-reduce enqueued behind the side
  // stream sees every gradient of the stage); after the last stage the main stream waits for the side stream
  fork_to_side(e, (hipStream_t)stream);
  if (stage == p5_backward_num_stages(e) - 1) join_side(e, (hipStream_t)stream);
  return 0;
}
// the gradient range that became FINAL with the most recent p5_backward_stage call (empty while a grouped weight-gradient launch is
// still collecting problems): contiguous, because the stages walk the arena from the back.  What a data-parallel caller all-reduces.
int p5_backward_final_range(const P5Engine* e, int64_t* begin, int64_t* end) { *begin = e->rep_b; *end = e->rep_e; return 0; }
// The whole staged backward in ONE call (round 5): every stage is enqueued back to back; whenever a gradient range becomes final an event is
// recorded behind it on `stream` and the range is noted.  A data-parallel caller then makes its communication stream wait for event k
// (p5_backward_staged_wait) and enqueues the all-reduce of range k -- the exchange still overlaps the rest of the backward on the device,
// and the host pays one library call per step instead of one per stage plus a range query each (16 + 16 at T5-small).
int p5_backward_staged(P5Engine* e, const float* dnll, void* stream, int64_t* ranges, int max_ranges, int* n_ranges) {
  P5_REQUIRE(ranges && n_ranges && max_ranges >= 1, "p5_backward_staged: ranges / n_ranges");
  const int nst = p5_backward_num_stages(e);
  int n = 0;
  for (int st = 0; st < nst; ++st) {
    P5_TRY(p5_backward_stage(e, dnll, st, stream));
    if (e->rep_e > e->rep_b) {
      P5_REQUIRE(n < max_ranges && n < P5_MAX_STAGED, "p5_backward_staged: more final ranges than the caller has room for");
#ifndef P5_EMU
      if (!e->st_ev[n]) P5_REQUIRE(hipEventCreateWithFlags(&e->st_ev[n], hipEventDisableTiming) == hipSuccess, "hipEventCreate");
      P5_REQUIRE(hipEventRecord(e->st_ev[n], (hipStream_t)stream) == hipSuccess, "hipEventRecord");
      if (e->side) {      // weight gradients of this range may have been launched on the side stream: the range is final behind BOTH events
        if (!e->st_ev_side[n]) P5_REQUIRE(hipEventCreateWithFlags(&e->st_ev_side[n], hipEventDisableTiming) == hipSuccess, "hipEventCreate");
        P5_REQUIRE(hipEventRecord(e->st_ev_side[n], e->side) == hipSuccess, "hipEventRecord");
      }
      e->st_side[n] = e->side != nullptr;
#endif
      ranges[2 * n] = e->rep_b; ranges[2 * n + 1] = e->rep_e;
      e->st_range[2 * n] = e->rep_b; e->st_range[2 * n + 1] = e->rep_e;
      ++n;
    }
  }
  e->st_n = n;
  *n_ranges = n;
  return 0;
}
int p5_backward_staged_wait(P5Engine* e, int k, void* comm_stream) {
  P5_REQUIRE(k >= 0 && k < e->st_n, "p5_backward_staged_wait: range index");
#ifndef P5_EMU
  P5_REQUIRE(hipStreamWaitEvent((hipStream_t)comm_stream, e->st_ev[k], 0) == hipSuccess, "hipStreamWaitEvent");
  if (e->st_side[k] && (hipStream_t)comm_stream != e->side)
    P5_REQUIRE(hipStreamWaitEvent((hipStream_t)comm_stream, e->st_ev_side[k], 0) == hipSuccess, "hipStreamWaitEvent");
#else
  (void)comm_stream;
#endif
  return 0;
}
// ---- the exchange itself for a caller without torch.distributed: RCCL through the ncclComm_t the caller created ----------------------
// No link-time dependency on a particular librccl: the symbols are taken from whatever RCCL the process has loaded (the one that made the
// communicator -- torch ships its own copy next to its HIP runtime, a C host links /opt/rocm/lib/librccl.so), else from librccl.so.1.
#ifndef P5_EMU
typedef int (*p5_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*p5_nccl_errstr_fn)(int);
static p5_nccl_allreduce_fn g_nccl_allreduce = nullptr;
static p5_nccl_errstr_fn g_nccl_errstr = nullptr;
static int nccl_resolve() {
  if (g_nccl_allreduce) return 0;
  void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
  void* h = nullptr;
  if (!sym) {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) sym = dlsym(h, "ncclAllReduce");
  }
  if (!sym) return fail("p5_allreduce: no RCCL in this process (ncclAllReduce not found; load librccl before creating the communicator)");
  g_nccl_errstr = (p5_nccl_errstr_fn)(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
  g_nccl_allreduce = (p5_nccl_allreduce_fn)sym;
  return 0;
}
#endif
// rccl.h: ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9, ncclInt64 = 4; ncclSum = 0
int p5_allreduce_sum(void* buf, int64_t count, int dtype, void* nccl_comm, void* stream) {
  P5_REQUIRE(buf && nccl_comm && count >= 0 && dtype >= 0 && dtype <= 3, "p5_allreduce_sum: arguments");
#ifndef P5_EMU
  P5_TRY(nccl_resolve());
  static const int kType[4] = {7, 9, 8, 4};
  const int rc = g_nccl_allreduce(buf, buf, (size_t)count, kType[dtype], 0, nccl_comm, (hipStream_t)stream);
  if (rc != 0) return fail(std::string("ncclAllReduce: ") + (g_nccl_errstr ? g_nccl_errstr(rc) : "error ") + " (" + std::to_string(rc) + ")");
  return 0;
#else
  (void)stream;
  return fail("p5_allreduce_sum: the host emulation has no RCCL");
#endif
}
// bf16 -> fp32 of an exchanged bucket
__global__ __launch_bounds__(256) void p5_widen_kernel(float* __restrict__ out, const bf16* __restrict__ in, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = to_f<bf16>(in[i]);
}
int p5_allreduce_range(P5Engine* e, int k, void* nccl_comm, void* bf16_scratch, void* comm_stream) {
  P5_REQUIRE(k >= 0 && k < e->st_n, "p5_allreduce_range: range index (call p5_backward_staged first)");
  P5_REQUIRE(e->G != nullptr, "p5_allreduce_range: no gradient arena bound");
  P5_TRY(p5_backward_staged_wait(e, k, comm_stream));
  const int64_t b = e->st_range[2 * k], n = e->st_range[2 * k + 1] - b;
  if (n <= 0) return 0;
  hipStream_t cs = (hipStream_t)comm_stream;
  if (!bf16_scratch) return p5_allreduce_sum(e->G + b, n, 0, nccl_comm, comm_stream);
  const unsigned blocks = (unsigned)((n / 8 + 255) / 256 > 4096 ? 4096 : (n / 8 + 255) / 256 + 1);
  P5_LAUNCH((p5_cast_kernel<bf16>), dim3(blocks), dim3(256), 0, cs, (bf16*)bf16_scratch, (const float*)(e->G + b), (size_t)n);
  P5_TRY(P5_KCHECK());
  P5_TRY(p5_allreduce_sum(bf16_scratch, n, 1, nccl_comm, comm_stream));
  P5_LAUNCH(p5_widen_kernel, dim3(blocks), dim3(256), 0, cs, e->G + b, (const bf16*)bf16_scratch, (size_t)n);
  return P5_KCHECK();
}
int p5_backward_stage_pairs(P5Engine* e, int on) { e->stage_pairs = on != 0; return 0; }
int p5_engine_grads_zeroed(P5Engine* e) { e->grads_keep = true; return 0; }
// zero_grad(set_to_none=True) of the reference loop: the gradients are dead until the next backward rewrites them -- nothing to do
// on the device (the next backward stores / clears what it needs); p5_engine_clear_grads is the eager form (set_to_none=False)
int p5_engine_discard_grads(P5Engine* e) { e->grads_zeroed = false; e->grads_keep = false; return 0; }

// zero_grad() of the reference loop (DistributedRunner.py:93): the 4 B x n_params fill is HBM-bound and nothing before the next
// backward needs its result, so it goes to the side stream (ordered after everything `stream` has been given so far, i.e. after
// the optimizer read the gradients) and overlaps the next forward; backward stage 0 waits for it.
int p5_engine_clear_grads(P5Engine* e, void* stream) {
  P5_REQUIRE(e->G, "engine not bound");
  hipStream_t main = (hipStream_t)stream, s = main;
#ifndef P5_EMU
  if (e->side) { fork_to_side(e, main); s = e->side; }
#endif
  if (hipMemsetAsync(e->G, 0, (size_t)e->n_params * 4, s) != 0) return fail("clear_grads: hipMemsetAsync failed");
#ifndef P5_EMU
  if (s != main) {
    if (!e->zg_ev) hipEventCreateWithFlags(&e->zg_ev, hipEventDisableTiming);
    hipEventRecord(e->zg_ev, s);
    e->zg_pending = true;
  }
#endif
  e->grads_zeroed = true;
  return 0;
}
int p5_engine_set_side_stream(P5Engine* e, void* side_stream) {
#ifndef P5_EMU
  if (side_stream && !e->side_events) {
    e->side_events = true;
    for (int i = 0; i < 32; ++i) hipEventCreateWithFlags(&e->ev_pool[i], hipEventDisableTiming);
    for (int i = 0; i < P5_NSETS; ++i) hipEventCreateWithFlags(&e->set_ev[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&e->head_wg_ev, hipEventDisableTiming);
    for (int i = 0; i < 64; ++i) hipEventCreateWithFlags(&e->kv_ev[i], hipEventDisableTiming);
  }
  e->side = (hipStream_t)side_stream;
#else
  (void)e; (void)side_stream;
#endif
  return 0;
}
int p5_backward(P5Engine* e, const float* dnll, void* stream) {
  const int n = p5_backward_num_stages(e);
  e->whole_backward = true;       // nobody consumes per-stage gradient ranges: weight gradients may be grouped across stages
  int rc = 0;
  for (int s = 0; s < n && rc == 0; ++s) rc = p5_backward_stage(e, dnll, s, stream);
  e->whole_backward = false;
  return rc;
}
int p5_backward_stage_range(const P5Engine* e, int stage, int64_t* begin, int64_t* end) {
  const int nd = e->c.n_dec_layers, ne = e->c.n_enc_layers;
  *begin = *end = 0;
  if (stage == 0) { *begin = e->off_dec_fln; *end = e->n_params; }
  else if (stage >= 1 && stage <= nd) { *begin = e->dec[nd - stage].begin; *end = e->dec[nd - stage].end; }
  else if (stage == nd + 2) { *begin = e->off_enc_fln; *end = e->dec[0].begin; }
  else if (stage >= nd + 3 && stage <= nd + 2 + ne) { const int i = ne - (stage - (nd + 2)); *begin = e->enc[i].begin; *end = e->enc[i].end; }
  else if (stage == nd + ne + 3) { *begin = 0; *end = e->off_small_end; }
  return 0;
}

int p5_grad_sumsq(const float* grads, int64_t n, float* out_scalar, void* stream) {
  P5_LAUNCH(p5_sumsq_kernel, dim3(P5_SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, out_scalar, grads, (size_t)n);
  return P5_KCHECK();
}
int p5_adamw_step(float* params, const float* grads, float* m, float* v, void* shadow_bf16, int64_t n, const float* sumsq, double max_norm,
                  double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay, int step_t, void* stream) {
  // Hyper-parameters arrive as the doubles the reference holds them in (Python floats) and every derived scalar is formed in double and
  // rounded ONCE to fp32 -- as torch does with the Python scalars of transformers' AdamW.step (`alpha=1.0 - beta1`, `value=1.0 - beta2`,
  // `value=-step_size`, `alpha=-lr * weight_decay`): 1.f - 0.999f is 1.3e-5 off 0.001, which tests/golden/adamw_426.json sees in `v`.
  P5AdamArgs a;
  a.p = params; a.g = grads; a.m = m; a.v = v; a.shadow = shadow_bf16; a.sumsq = sumsq; a.n = (size_t)n;
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps; a.max_norm = (float)max_norm; a.grad_scale = (float)grad_scale;
  const double bc1 = 1.0 - pow(beta1, (double)step_t), bc2 = 1.0 - pow(beta2, (double)step_t);
  a.step_size = (float)(lr * sqrt(bc2) / bc1);
  a.decay = (float)(lr * weight_decay);
  P5_LAUNCH(p5_adamw_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, a);
  return P5_KCHECK();
}

int64_t p5_generate_workspace_bytes(const P5Engine* e, int B, int L, int K, int max_len, int max_children, int excluded_words) {
  P5Engine tmp = *e;
  return layout_gen(&tmp, nullptr, B, L, K, max_len, max_children, excluded_words, nullptr);
}
static void gen_mark_begin(P5Engine* e, void* stream) {
#ifndef P5_EMU
  if (e->gen_timing) {
    if (!e->gen_ev[0]) for (int i = 0; i < 3; ++i) hipEventCreate(&e->gen_ev[i]);
    hipEventRecord(e->gen_ev[0], (hipStream_t)stream);
  }
#endif
}
int p5_decode_begin(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, int B, int L, int K,
                    int max_len, const int* child_off, const int* child_tok, const int* child_node, const int* roots,
                    const uint32_t* excluded_nodes, int excluded_words, int max_children, void* ws, int64_t ws_bytes, void* stream) {
  P5_REQUIRE(e->P, "engine not bound");
  P5_REQUIRE(K >= 1 && K <= 64, "1 <= num_beams <= 64");
  P5_REQUIRE(max_len >= 2 && max_len <= P5_MAX_LEN, "2 <= max_length <= 128 (P5_MAX_LEN)");
  P5_REQUIRE(max_len <= 64 || g_opt_decode_v2, "max_length > 64 needs the decode_v2 step (the first-generation self-attention kernel keeps one score per lane)");
  P5_REQUIRE(L >= 1 && L <= 512, "1 <= L <= 512");
  P5_REQUIRE(max_children >= 1, "max_children");
  P5_REQUIRE(e->lut_half >= max_len, "bucket LUT too short");
  P5_REQUIRE(excluded_words >= 0 && (excluded_nodes || excluded_words == 0), "excluded_nodes / excluded_words");
  const int64_t need = layout_gen(e, nullptr, B, L, K, max_len, max_children, excluded_words, nullptr);
  P5_REQUIRE(ws_bytes >= need, "workspace too small");
  gen_mark_begin(e, stream);
  e->ids = input_ids; e->ww = whole_word_ids; e->mask = attention_mask; e->labels = nullptr;
  return e->c.dtype == 1
             ? decode_begin_impl<bf16>(e, B, L, K, max_len, child_off, child_tok, child_node, roots, excluded_nodes, excluded_words, max_children,
                                       (char*)ws, (hipStream_t)stream)
             : decode_begin_impl<float>(e, B, L, K, max_len, child_off, child_tok, child_node, roots, excluded_nodes, excluded_words, max_children,
                                        (char*)ws, (hipStream_t)stream);
}
int p5_decode_step(P5Engine* e, void* stream) {
  return e->c.dtype == 1 ? decode_step_impl<bf16>(e, (hipStream_t)stream) : decode_step_impl<float>(e, (hipStream_t)stream);
}
const int* p5_decode_done_flag(const P5Engine* e) { return e->gen.active ? e->gen.w.st.flags + 4 : nullptr; }
int p5_decode_finish(P5Engine* e, int* out_seq, float* out_score, int* out_len, void* stream) {
  return decode_finish_impl(e, out_seq, out_score, out_len, (hipStream_t)stream);
}
int p5_generate(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, int B, int L, int K,
                int max_len, const int* child_off, const int* child_tok, const int* child_node, const int* roots,
                const uint32_t* excluded_nodes, int excluded_words, int max_children, int* out_seq, float* out_score, int* out_len, void* ws, int64_t ws_bytes, void* stream) {
  P5_TRY(p5_decode_begin(e, input_ids, whole_word_ids, attention_mask, B, L, K, max_len, child_off, child_tok, child_node, roots, excluded_nodes,
                         excluded_words, max_children, ws, ws_bytes, stream));
#ifndef P5_EMU
  if (e->gen_timing) {        // device-time brackets of the decode loop (p5_generate_timing): two event records, nothing is waited for
    if (!e->gen_ev[0]) for (int i = 0; i < 3; ++i) hipEventCreate(&e->gen_ev[i]);
    hipEventRecord(e->gen_ev[1], (hipStream_t)stream);
  }
#endif
  // every step is enqueued without reading anything back (the search stops on the device); the caller bounds max_len by the
  // depth of the trie, so at most a step or two are no-ops
  for (int cur_len = 1; cur_len < max_len; ++cur_len) P5_TRY(p5_decode_step(e, stream));
#ifndef P5_EMU
  if (e->gen_timing) hipEventRecord(e->gen_ev[2], (hipStream_t)stream);
#endif
  return p5_decode_finish(e, out_seq, out_score, out_len, stream);
}
int p5_generate_set_forced_prefix(P5Engine* e, const int* tokens, const int* nodes, int n) {
  P5_REQUIRE(n >= 0 && (n == 0 || (tokens && nodes)), "forced prefix: tokens / nodes");
  e->ff_next.n = n < P5_FF_MAX ? n : P5_FF_MAX;
  for (int i = 0; i < e->ff_next.n; ++i) { e->ff_next.tok[i] = tokens[i]; e->ff_next.node[i] = nodes[i]; }
  return 0;
}
int64_t p5_generate_history_count(int B, int K, int max_len) { return 4 + (int64_t)max_len * P5_HIST_FIELDS * B * K; }
int p5_generate_draft(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, int B, int L, int K,
                      int max_len, const int* child_off, const int* child_tok, const int* child_node, const int* roots,
                      const uint32_t* excluded_nodes, int excluded_words, int max_children, int* out_seq, float* out_score, int* out_len, int* hist,
                      void* ws, int64_t ws_bytes, void* stream) {
  P5_REQUIRE(hist, "p5_generate_draft: history buffer");
  e->gen_hist_next = hist;
  const int rc = p5_generate(e, input_ids, whole_word_ids, attention_mask, B, L, K, max_len, child_off, child_tok, child_node, roots, excluded_nodes,
                             excluded_words, max_children, out_seq, out_score, out_len, ws, ws_bytes, stream);
  e->gen_hist_next = nullptr;
  return rc;
}
int64_t p5_verify_workspace_bytes(const P5Engine* e, int B, int L, int K, int Kw, int max_len, int max_children, int excluded_words) {
  return layout_verify(const_cast<P5Engine*>(e), nullptr, B, L, K, Kw, max_len, max_children, excluded_words, nullptr);
}
int p5_verify_begin(P5Engine* e, int B, int L, int K, int Kw, int max_len, const int* child_off, const int* child_tok, const int* child_node,
                    const int* roots, int max_children, int excluded_words, void* ws, int64_t ws_bytes) {
  P5_REQUIRE(e->P, "engine not bound");
  P5_REQUIRE(K >= 1 && 2 * K * K <= P5_VERIFY_POOL, "verified generation: 1 <= num_beams <= 22");
  P5_REQUIRE(Kw >= K && Kw <= P5_MAX_K, "draft beam width: num_beams <= Kw <= 64");
  P5_REQUIRE(max_len >= 2 && max_len <= P5_MAX_LEN, "2 <= max_length <= 128 (P5_MAX_LEN)");
  P5_REQUIRE(L >= 1 && L <= 512, "1 <= L <= 512");
  P5_REQUIRE(max_children >= 1 && excluded_words >= 0, "max_children / excluded_words");
  P5_REQUIRE(e->lut_half >= max_len, "bucket LUT too short");
  const int64_t need = layout_verify(e, nullptr, B, L, K, Kw, max_len, max_children, excluded_words, nullptr);
  P5_REQUIRE(ws_bytes >= need, "workspace too small");
  return verify_begin_impl(e, B, L, K, Kw, max_len, child_off, child_tok, child_node, roots, max_children, excluded_words, (char*)ws);
}
int p5_verify_encode(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, void* stream) {
  P5_REQUIRE(e->ver && e->ver->begun, "p5_verify_encode without p5_verify_begin");
  return e->c.dtype == 1 ? verify_encode_impl<bf16>(e, input_ids, whole_word_ids, attention_mask, (hipStream_t)stream)
                         : verify_encode_impl<float>(e, input_ids, whole_word_ids, attention_mask, (hipStream_t)stream);
}
const void* p5_verify_encoder_output(const P5Engine* e) { return (e->ver && e->ver->encoded) ? e->enc_out : nullptr; }
int p5_verify_plan(P5Engine* e, const int* hist, void* stream) {
  P5_REQUIRE(e->ver && e->ver->begun, "p5_verify_plan without p5_verify_begin");
  P5_REQUIRE(hist, "p5_verify_plan: history of the draft search");
  return verify_plan_impl(e, hist, (hipStream_t)stream);
}
int p5_verify_row_capacity(int Kw, int max_len) { return verify_cap(Kw, max_len); }
const int* p5_verify_plan_header(const P5Engine* e) { return (e->ver && e->ver->begun) ? e->ver->w.pl.hdr : nullptr; }
int p5_verify_run(P5Engine* e, int rows_per_user, const uint32_t* excluded_nodes, int* out_seq, float* out_score, int* out_len, int* out_missing,
                  void* stream) {
  P5_REQUIRE(e->ver && e->ver->begun && e->ver->encoded && e->ver->planned, "p5_verify_run needs p5_verify_begin + p5_verify_encode + p5_verify_plan");
  P5_REQUIRE(rows_per_user >= 1 && rows_per_user <= e->ver->w.pl.cap, "rows_per_user: 1 .. capacity of the plan");
  P5_REQUIRE(e->ver->excl_words == 0 || excluded_nodes, "excluded_nodes");
  return e->c.dtype == 1 ? verify_run_impl<bf16>(e, rows_per_user, excluded_nodes, out_seq, out_score, out_len, out_missing, (hipStream_t)stream)
                         : verify_run_impl<float>(e, rows_per_user, excluded_nodes, out_seq, out_score, out_len, out_missing, (hipStream_t)stream);
}
int p5_generate_set_encoder_output(P5Engine* e, const float* enc_out_f32) { e->enc_ext_next = enc_out_f32; return 0; }
int p5_generate_timing(P5Engine* e, int enable, float* encode_ms, float* decode_ms) {
#ifndef P5_EMU
  if (encode_ms || decode_ms) {
    P5_REQUIRE(e->gen_timing && e->gen_ev[0], "generate_timing: enable it before the p5_generate call to be measured");
    P5_REQUIRE(hipEventSynchronize(e->gen_ev[2]) == hipSuccess, "generate_timing: event sync failed");
    float a = 0.f, b = 0.f;
    hipEventElapsedTime(&a, e->gen_ev[0], e->gen_ev[1]);
    hipEventElapsedTime(&b, e->gen_ev[1], e->gen_ev[2]);
    if (encode_ms) *encode_ms = a;
    if (decode_ms) *decode_ms = b;
  }
  e->gen_timing = enable != 0;
#else
  if (encode_ms) *encode_ms = 0.f;
  if (decode_ms) *decode_ms = 0.f;
  (void)enable;
#endif
  return 0;
}
int p5_encode(P5Engine* e, const int64_t* input_ids, const int64_t* whole_word_ids, const int64_t* attention_mask, int B, int L,
              void* enc_out, void* ws, int64_t ws_bytes, void* stream) {
  P5_REQUIRE(e->P, "engine not bound");
  const int64_t need = layout_ws(e, (char*)ws, B, L, 0, false);
  P5_REQUIRE(ws_bytes >= need, "workspace too small");
  e->B = B; e->L = L; e->T = 0; e->M = B * L; e->Md = 0; e->training = 0;
  e->ids = input_ids; e->ww = whole_word_ids; e->mask = attention_mask;
  P5_TRY(e->c.dtype == 1 ? encoder_fwd<bf16>(e, (hipStream_t)stream) : encoder_fwd<float>(e, (hipStream_t)stream));
  const size_t bytes = (size_t)B * L * e->c.d_model * (e->c.dtype == 1 ? 2 : 4);
  hipMemcpyAsync(enc_out, e->enc_out, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return 0;
}

// ---- per-kernel entry points ---------------------------------------------------------------------------
static P5Drop op_drop(const uint32_t* state, uint32_t site, float p) {
  P5Drop d = no_drop();
  if (state && p > 0.f) { d.state = state; d.site_key = p5_site_key(site); d.thr = p5_drop_thr(p); d.scale = 1.f / (1.f - p); }
  return d;
}
int p5_op_gemm(int dtype, const void* A, const void* Bm, void* C, const void* aux, int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
               int a_ks, int b_ks, int epi, int c_f32, int splitk, float alpha, const uint32_t* rng_state, uint32_t site, float drop_p,
               void* stream) {
  P5GemmArgs g;
  g.A = A; g.B = Bm; g.C = C; g.aux = aux; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.a_ks = a_ks; g.b_ks = b_ks; g.epi = epi; g.c_f32 = c_f32; g.splitk = splitk; g.ring = 0; g.alpha = alpha; g.drop = op_drop(rng_state, site, drop_p);
  g.mm_split = (dtype == 2) ? 1 : 0;      // dtype 2 (tests): fp32 operands, split-f16 products
  g.rowss = nullptr; g.rowss_invd = 0.f; g.rowss_eps = 0.f; g.ssq_out = nullptr; g.rowss_nt = 0; g.ssq_nt = 0; g.c_split_stride = 0;
  g.g4_tiles_n = 0; g.g4_nk = 0; g.xcd_bm = g.xcd_bn = 0;
  return dtype == 1 ? launch_gemm<bf16>(g, (hipStream_t)stream) : launch_gemm<float>(g, (hipStream_t)stream);
}
int p5_op_gemm_group(int tile_cfg, int ks, int nprob, const P5GemmProblem* probs, const uint32_t* rng_state, uint32_t site, float drop_p,
                     void* stream) {
  P5_REQUIRE(nprob >= 1 && nprob <= P5_MAX_GROUP && probs, "gemm_group: 1..8 problems");
  P5GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.nprob = nprob;
  for (int i = 0; i < nprob; ++i) {
    const P5GemmProblem& q = probs[i];
    P5GemmArgs& g = grp.p[i];
    g.A = q.A; g.B = q.B; g.C = q.C; g.aux = q.aux; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc; g.ldaux = q.ldaux;
    g.a_ks = ks; g.b_ks = ks; g.epi = q.epi; g.c_f32 = q.c_f32; g.splitk = q.splitk; g.alpha = q.alpha; g.drop = op_drop(rng_state, site, drop_p);
    g.rowss = q.rowss; g.rowss_invd = q.rowss ? 1.0f / (float)q.K : 0.f; g.rowss_eps = q.rowss_eps; g.ssq_out = q.ssq_out;
    g.rowss_nt = q.rowss_nt; g.ssq_nt = q.ssq_nt;
    g.C2 = q.C2; g.ldc2 = q.ldc2; g.gate_F = q.gate_F;
    g.nb_dot = q.nb_dot; g.nb_dot_nt = q.nb_dot_nt; g.nb_rin = q.nb_rin; g.nb_rout = q.nb_rout; g.nb_w = q.nb_w; g.nb_dw = q.nb_dw;
    if (q.epi == P5_EPI_NORM_BWD) {
      P5_REQUIRE(tile_cfg == P5_G5_128x128 && !ks, "gemm_group: the T5LayerNorm-backward epilogue runs on tile_cfg 4, ks 0");
      g.rowss_invd = 1.0f / (float)q.N;      // (the statistics are those of the d_model-wide rows of x = aux, not of the reduction dimension)
    }
    if (q.epi == P5_EPI_MASK_POS && q.ssq_out)
      P5_REQUIRE((tile_cfg == P5_G4_256x128 || tile_cfg == P5_G5_256x128 || tile_cfg == P5_G5_128x128) && !ks && (g_opt_gemm_ws & 1) && q.ssq_nt > 0 && !q.c_f32,
                 "gemm_group: row sums of <d pre, pre> (epi 3 + ssq_out) exist in the wave-specialised kernel only (tile_cfg 1, 3 or 4)");
    if (q.epi == P5_EPI_GELU_GATE || q.epi == P5_EPI_GELU_GATE_BWD)
      P5_REQUIRE(tile_cfg == P5_G4_256x128 && !ks && (g_opt_gemm_ws & 1) && (q.M % 256) == 0 && (q.N % 128) == 0 && !q.c_f32 &&
                 (q.epi == P5_EPI_GELU_GATE ? (q.C2 && q.gate_F * 2 == q.N) : (q.aux && q.gate_F == 0)),
                 "gemm_group: the gated-GELU epilogues run on whole 256x128 tiles of the wave-specialised kernel (tile_cfg 1, ks 0)");
  }
  return launch_gemm4(tile_cfg, ks != 0, grp, (hipStream_t)stream);
}
int p5_op_rmsnorm_fwd(int dtype, void* y, float* rstd, const void* x, const float* w, int rows, int d, float eps, void* stream) {
  return dtype == 1 ? rmsnorm_fwd<bf16>((hipStream_t)stream, y, rstd, x, w, rows, d, eps, no_drop())
                    : rmsnorm_fwd<float>((hipStream_t)stream, y, rstd, x, w, rows, d, eps, no_drop());
}
int p5_op_rmsnorm_bwd(int dtype, float* dres_out, void* dy_next, float* dw, const void* dy, const void* x, const float* w, const float* rstd,
                      const float* dres_in, int rows, int d, float* dw_partial, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int nblk = 0;
  P5_TRY(dtype == 1 ? rmsnorm_bwd<bf16>(s, dres_out, dy_next, dw, dy, x, w, rstd, dres_in, rows, d, no_drop(), no_drop(), dw_partial, &nblk)
                    : rmsnorm_bwd<float>(s, dres_out, dy_next, dw, dy, x, w, rstd, dres_in, rows, d, no_drop(), no_drop(), dw_partial, &nblk));
  if (dw_partial) {    // the engine's mode: per-workgroup partials + a separate reduction
    P5_LAUNCH(p5_reduce_rows_kernel, dim3((d + 15) / 16), dim3(256), 0, s, dw, (const float*)dw_partial, nblk, d);
    P5_TRY(P5_KCHECK());
  }
  return 0;
}
// test hook (p5_set_option "attn_op_keep_bits"): the standalone attention ops hand the long-sequence kernels a keep-mask buffer of their
// own (P5AttnArgs::keep_bits: written by p5_op_attn_fwd, read by the p5_op_attn_bwd that follows it), as the engine does per encoder layer
static uint32_t* op_keep_bits(int dtype, int B, int H, int Lq, int Lk, const P5Drop& d) {
  static uint32_t* buf = nullptr;
  static size_t cap = 0;
  if (!g_opt_attn_op_keep_bits || dtype != 1 || Lk <= 128 || d.state == nullptr || d.thr == 0 || !g_opt_attn_fwd_head || !g_opt_attn_bwd_head) return nullptr;
  const size_t need = (size_t)B * H * ((Lq + 15) / 16) * 1024;
  if (need > cap) {
#ifdef P5_EMU
    free(buf);
    buf = (uint32_t*)malloc(need);
    if (!buf) { cap = 0; return nullptr; }
#else
    if (buf) (void)hipFree(buf);
    buf = nullptr; cap = 0;
    if (hipMalloc((void**)&buf, need) != hipSuccess) return nullptr;      // (the only allocation of the library: a test hook, off by default)
#endif
    cap = need;
  }
  return buf;
}
int p5_op_attn_fwd(int dtype, const void* Q, const void* K, const void* V, void* O, float* lse, const float* rel_table, const int* lut,
                   int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int causal,
                   const uint32_t* rng_state, uint32_t site, float drop_p, void* stream) {
  P5AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.lse = lse; a.rel_table = rel_table; a.bucket_lut = lut; a.lut_half = lut_half; a.kmask = kmask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.causal = causal;
  a.drop = op_drop(rng_state, site, drop_p);
  a.keep_bits = op_keep_bits(dtype, B, H, Lq, Lk, a.drop);
  return dtype == 1 ? launch_attn_fwd<bf16>(a, (hipStream_t)stream) : launch_attn_fwd<float>(a, (hipStream_t)stream);
}
int p5_op_attn_bwd(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse, float* Dvec,
                   void* dQ, void* dK, void* dV, const float* rel_table, float* d_rel_table, float* d_rel_scratch, int rel_buckets, const int* lut,
                   int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                   int causal, const uint32_t* rng_state, uint32_t site, float drop_p, void* stream) {
  return p5_op_attn_bwd_dot(dtype, Q, K, V, O, dO, lse, Dvec, dQ, dK, dV, rel_table, d_rel_table, d_rel_scratch, rel_buckets, lut, lut_half, kmask, B, H, Lq, Lk,
                            ldq, ldk, ldv, ldo, lddq, lddk, lddv, causal, rng_state, site, drop_p, nullptr, stream);
}
int p5_op_attn_bwd_dot(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse, float* Dvec,
                       void* dQ, void* dK, void* dV, const float* rel_table, float* d_rel_table, float* d_rel_scratch, int rel_buckets, const int* lut,
                       int lut_half, const int64_t* kmask, int B, int H, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                       int causal, const uint32_t* rng_state, uint32_t site, float drop_p, float* dot_out, void* stream) {
  P5AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.dot_out = dot_out;
  a.Q = Q; a.K = K; a.V = V; a.O = (void*)O; a.dO = dO; a.lse = (float*)lse; a.Dvec = Dvec; a.dQ = dQ; a.dK = dK; a.dV = dV;
  a.rel_table = rel_table; a.d_rel_table = nullptr; a.bucket_lut = lut; a.lut_half = lut_half; a.kmask = kmask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = ldo; a.lddq = lddq; a.lddk = lddk;
  a.lddv = lddv; a.causal = causal; a.rel_copies = rel_buckets; a.rel_stride = rel_buckets * H; a.drop = op_drop(rng_state, site, drop_p);
  a.keep_bits = op_keep_bits(dtype, B, H, Lq, Lk, a.drop);
  hipStream_t s = (hipStream_t)stream;
  if (dot_out) P5_REQUIRE(p5l_attn_bwd_dot_ok(dtype == 1, a), "attn_bwd: dot_out is written by the fused bf16 self-attention backward only (Lq == Lk in 17..128)");
  if (d_rel_table) {
    P5_REQUIRE(d_rel_scratch && rel_buckets >= 1 && rel_buckets <= 64, "attn_bwd: d_rel_table needs d_rel_scratch [B * ceil(Lq / 64)][rel_buckets * H] and rel_buckets <= 64");
    a.d_rel_table = d_rel_scratch;
  }
  P5_TRY(dtype == 1 ? launch_attn_bwd<bf16>(a, s) : launch_attn_bwd<float>(a, s));
  if (d_rel_table) {
    P5_LAUNCH(p5_reduce_rows_kernel, dim3((rel_buckets * H + 15) / 16), dim3(256), 0, s, d_rel_table, (const float*)d_rel_scratch,
              p5l_attn_bwd_slots(dtype == 1, B, Lq, Lk), rel_buckets * H);
    P5_TRY(P5_KCHECK());
  }
  return 0;
}
int p5_op_ce_fwd(float* nll, float* lse, const float* logits, const int64_t* labels, int rows, int V, int ldl, void* stream) {
  P5_LAUNCH((p5_ce_fwd_kernel<float>), dim3(rows), dim3(256), 0, (hipStream_t)stream, nll, lse, logits, labels, V, ldl);
  return P5_KCHECK();
}
int p5_op_skinny_gemm(int dtype, int amode, const void* A, int lda, const float* ln, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                      int epi, float alpha, float eps, void* stream) {
  return dtype == 1 ? skinny<bf16>((hipStream_t)stream, amode, A, lda, ln, (const bf16*)W, ldw, C, ldc, M, N, K, epi, alpha, eps, nullptr)
                    : skinny<float>((hipStream_t)stream, amode, A, lda, ln, (const float*)W, ldw, C, ldc, M, N, K, epi, alpha, eps, nullptr);
}
// single-token cross-attention of R = B * Kb beam rows (q given, not fused): variant 3 = MFMA kernel, 2 = scalar kernel
int p5_op_dec_cross_attn(int dtype, int variant, void* out, const void* q, const void* kv, const int64_t* mask, int B, int H, int Kb, int L,
                         void* stream) {
  P5CrossArgs a;
  memset(&a, 0, sizeof(a));
  a.out = out; a.q = q; a.kv = kv; a.mask = mask; a.R = B * Kb; a.H = H; a.Kb = Kb; a.L = L; a.d = 0; a.eps = 0.f; a.done = nullptr;
  a.ldkv = 2 * H * 64;
  const dim3 grid(B * ((Kb + 15) / 16), H), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 1) {
    if (variant == 3) P5_LAUNCH((p5_dec_cross_attn3_kernel<bf16, false, 52>), grid, block, 0, s, a);
    else P5_LAUNCH((p5_dec_cross_attn2_kernel<bf16, false, 48>), grid, block, 0, s, a);
  } else {
    if (variant == 3) P5_LAUNCH((p5_dec_cross_attn3_kernel<float, false, 96>), grid, block, 0, s, a);
    else P5_LAUNCH((p5_dec_cross_attn2_kernel<float, false, 80>), grid, block, 0, s, a);
  }
  return P5_KCHECK();
}
int p5_op_tr_probe(void* out, const void* in, void* stream) {
  P5_LAUNCH(p5_tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned short*)out, (const unsigned short*)in);
  return P5_KCHECK();
}

}  // extern "C"
