// p5_decode.h -- device-resident trie-constrained beam search (the `generate()` half of the hot path).
//
// Restates HF beam search as called by DistributedRunner.py:361-371 (generate(num_beams=K, num_return_sequences=K,
// prefix_allowed_tokens_fn=trie)), following transformers 5.15.0 generation/utils.py:3008-3560 (vectorised
// _beam_search) and PrefixConstrainedLogitsProcessor (logits_process.py:1536-1553):
//   log_softmax over the FULL vocab, candidates outside the trie children of the beam's prefix are -inf (not
//   renormalised), + running beam score, top-2K per batch item, EOS candidates ranked < K finish with
//   score / generated_len, best K non-finished continue, early_stopping=False heuristic.
// The reference walks a Python dict trie per (batch x beam) row per step with a D2H sync each
// (generation_trie.py:47-70,91-97); here every beam carries its trie NODE id and the children come from a CSR
// copy of the trie in HBM, so only the <= fan-out allowed logits are ever gathered.
// KV-cache: self-attention K/V are stored per (step, row) and never permuted -- an ancestry table maps
// (step, current beam) -> row that produced it; cross-attention K/V are kept per batch item and
// shared by its K beams (the reference expands the encoder states xK, P5_T5.py:571-576).
#pragma once
#include "p5_device.h"

// ---- single-token self-attention over the ancestry-indexed cache: one wave per (row, head) ----
template <class T>
__global__ __launch_bounds__(256) void p5_dec_self_attn_kernel(T* __restrict__ out, const T* __restrict__ qkv, T* __restrict__ cache,
                                                              const int* __restrict__ anc_odd, const int* __restrict__ anc_even,
                                                              const float* __restrict__ rel_table, const int* __restrict__ lut,
                                                              int lut_half, int R, int H, const int* __restrict__ step, int max_len) {
  // the step counter lives in device memory (cur_len = tokens so far, incl. the start token) so that ONE captured
  // hipGraph of the decode step can be replayed for every step; double-buffered state is selected by its parity
  const int cur_len = *step;
  const int pos = cur_len - 1;
  const int* __restrict__ anc = (cur_len & 1) ? anc_odd : anc_even;
  const int lane = threadIdx.x & 63;
  const int rh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rh >= R * H) return;
  const int r = rh / H, h = rh % H;
  const int inner = H * 64;
  const float q = to_f<T>(qkv[(size_t)r * 3 * inner + h * 64 + lane]);
  const T kcur = qkv[(size_t)r * 3 * inner + inner + h * 64 + lane];
  const T vcur = qkv[(size_t)r * 3 * inner + 2 * inner + h * 64 + lane];
  // cache layout: [max_len][R][2*inner]  (K then V)
  cache[((size_t)pos * R + r) * 2 * inner + h * 64 + lane] = kcur;
  cache[((size_t)pos * R + r) * 2 * inner + inner + h * 64 + lane] = vcur;
  float mys = P5_NEG_INF;
  for (int t = 0; t <= pos; ++t) {
    float kv;
    if (t == pos) kv = to_f<T>(kcur);
    else kv = to_f<T>(cache[((size_t)t * R + anc[(size_t)t * R + r]) * 2 * inner + h * 64 + lane]);
    float s = wave_sum(q * kv);
    s += rel_table[lut[(t - pos) + lut_half] * H + h];
    if (lane == t) mys = s;
  }
  const float m = wave_max(mys);
  const float p = (lane <= pos) ? expf(mys - m) : 0.f;
  const float l = wave_sum(p);
  float o = 0.f;
  for (int t = 0; t <= pos; ++t) {
    const float pt = __shfl(p, t);
    float vv;
    if (t == pos) vv = to_f<T>(vcur);
    else vv = to_f<T>(cache[((size_t)t * R + anc[(size_t)t * R + r]) * 2 * inner + inner + h * 64 + lane]);
    o += pt * vv;
  }
  out[(size_t)r * inner + h * 64 + lane] = from_f<T>(o / l);
}

// ---- single-token cross-attention; K/V of batch item r / Kb are shared by its beams ----
template <class T>
__global__ __launch_bounds__(256) void p5_dec_cross_attn_kernel(T* __restrict__ out, const T* __restrict__ q, const T* __restrict__ kv,
                                                               const int64_t* __restrict__ mask, int R, int H, int Kb, int L) {
  __shared__ float sq[4][64];
  __shared__ float sp[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rh = blockIdx.x * 4 + wave;
  const bool active = rh < R * H;
  const int r = active ? rh / H : 0, h = active ? rh % H : 0;
  const int b = r / Kb;
  const int inner = H * 64;
  if (active) sq[wave][lane] = to_f<T>(q[(size_t)r * inner + h * 64 + lane]);
  __syncthreads();
  float s[8];
  float m = P5_NEG_INF;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    s[i] = P5_NEG_INF;
    if (active && j < L && mask[(size_t)b * L + j] != 0) {
      const T* kr = kv + ((size_t)b * L + j) * 2 * inner + h * 64;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 64 / TT<T>::EPF; ++c) {
        float x[8];
        unpack16<T>(ld16(kr + c * TT<T>::EPF), x);
#pragma unroll
        for (int e = 0; e < TT<T>::EPF; ++e) acc += x[e] * sq[wave][c * TT<T>::EPF + e];
      }
      s[i] = acc;
    }
    m = fmaxf(m, s[i]);
  }
  m = wave_max(m);
  if (m == P5_NEG_INF) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    const float p = expf(s[i] - m);
    if (j < 512) sp[wave][j] = p;
    l += p;
  }
  l = wave_sum(l);
  __syncthreads();
  if (!active) return;
  // P V: lane = (key group kg = lane / NDG, dim group dg = lane % NDG); each lane accumulates EPF dims over the keys
  // j == kg (mod NKG) with one 16-byte load per key, then the key groups are summed with shuffles
  constexpr int EPF = TT<T>::EPF, NDG = 64 / EPF, NKG = 64 / NDG;
  const int kg = lane / NDG, dg = lane % NDG;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j = kg; j < L; j += NKG) {
    float x[8];
    unpack16<T>(ld16(kv + ((size_t)b * L + j) * 2 * inner + inner + h * 64 + dg * EPF), x);
    const float pj = sp[wave][j];
#pragma unroll
    for (int e = 0; e < EPF; ++e) o[e] += pj * x[e];
  }
#pragma unroll
  for (int e = 0; e < EPF; ++e) {
#pragma unroll
    for (int msk = NDG; msk < 64; msk <<= 1) o[e] += __shfl_xor(o[e], msk);
  }
  if (kg == 0) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int e = 0; e < EPF; ++e) o[e] *= inv;
    st16(out + (size_t)r * inner + h * 64 + dg * EPF, pack16<T>(o));
  }
}

// block-wide arg-max with deterministic tie-break (lowest index); every thread gets the winner
__device__ static __forceinline__ void block_argmax(float& bv, int& bi, float* s_val, int* s_idx) {
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
    const float ov = __shfl_xor(bv, msk);
    const int oi = __shfl_xor(bi, msk);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = bv; s_idx[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = s_val[0]; bi = s_idx[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
}

// ---- per row: full-vocab log-sum-exp in ONE pass (online max/sum, 16-byte loads), then gather only the trie
// children's log-probs (+ running score) and keep the row's best K2 = 2K of them, sorted (score desc, child asc).
// The global top-2K of a batch item is contained in the union of its rows' top-2K lists.
#define P5_ROW_LDS_CAND 2048
__global__ __launch_bounds__(256) void p5_dec_score_kernel(float* __restrict__ cand_score, float* __restrict__ top_score,
                                                          int* __restrict__ top_c, int* __restrict__ n_top,
                                                          const float* __restrict__ logits, int ldl, int V,
                                                          const int* __restrict__ node, const float* __restrict__ run_score,
                                                          const int* __restrict__ child_off, const int* __restrict__ child_tok,
                                                          const int* __restrict__ child_node, const uint32_t* __restrict__ excluded,
                                                          int excl_words, int Kb, int max_c, int K2, const int* __restrict__ done) {
  __shared__ float sm[4], ss[4];
  if (done && *done) return;
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float sc[P5_ROW_LDS_CAND];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int nd = node[r];
  if (nd < 0) {  // dead beam: no candidates (uniform per block)
    if (tid == 0) n_top[r] = 0;
    return;
  }
  const float* lr = logits + (size_t)r * ldl;
  float m = P5_NEG_INF, sum = 0.f;
  const int V4 = V >> 2;
  for (int j = tid; j < V4; j += 256) {
    const f32x4 v = *(const f32x4*)(lr + 4 * j);
    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (mx > m) { sum *= expf(m - mx); m = mx; }
    sum += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
  }
  for (int j = (V4 << 2) + tid; j < V; j += 256) {
    const float v = lr[j];
    if (v > m) { sum *= expf(m - v); m = v; }
    sum += expf(v - m);
  }
  {
    const float wm_ = wave_max(m);
    sum = wave_sum(m == P5_NEG_INF ? 0.f : sum * expf(m - wm_));
    if ((tid & 63) == 0) { sm[tid >> 6] = wm_; ss[tid >> 6] = sum; }
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    sum = 0.f;
    for (int w = 0; w < 4; ++w) sum += ss[w] * expf(sm[w] - m);
  }
  const float lse = m + logf(sum);
  const int c0 = child_off[nd];
  int nc = child_off[nd + 1] - c0;
  if (nc > max_c) nc = max_c;
  const float rs = run_score[r];
  const bool in_lds = nc <= P5_ROW_LDS_CAND;
  float* cs = in_lds ? sc : cand_score + (size_t)r * max_c;
  // per-item excluded-node bitmap (filtered evaluation, DistributedRunner.py:286-297): a child whose subtree holds only
  // items the user has already interacted with does not exist in that user's trie
  const uint32_t* ex = excluded ? excluded + (size_t)(r / Kb) * excl_words : nullptr;
  for (int c = tid; c < nc; c += 256) {
    float v = (lr[child_tok[c0 + c]] - lse) + rs;
    if (ex) {
      const int cn = child_node[c0 + c];
      if ((ex[cn >> 5] >> (cn & 31)) & 1u) v = P5_NEG_INF;
    }
    cs[c] = v;
  }
  __syncthreads();
  const int want = K2 < nc ? K2 : nc;
  for (int it = 0; it < want; ++it) {
    float bv = P5_NEG_INF;
    int bi = 0x7fffffff;
    for (int c = tid; c < nc; c += 256) {
      const float v = cs[c];
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    block_argmax(bv, bi, s_val, s_idx);
    if (bi == 0x7fffffff) {            // only excluded children left; stop early
      if (tid == 0) n_top[r] = it;
      return;
    }
    if (tid == 0) {
      top_score[(size_t)r * K2 + it] = bv;
      top_c[(size_t)r * K2 + it] = bi;
      cs[bi] = P5_NEG_INF;   // taken (a genuine -inf candidate is never selected above)
    }
    __syncthreads();
  }
  if (tid == 0) n_top[r] = want;
}

#define P5_MAX_K 64
#define P5_MAX_K2 128
#define P5_RANK_POOL 2048        // candidates (beams x 2 beams) ranked in parallel: beam widths up to 32
// (score desc, key asc) as ONE unsigned 64-bit order: larger = better.  -inf candidates hold 0 and are never counted.
__device__ static __forceinline__ unsigned long long p5_rank_key(float v, int key) {
  union { float f; unsigned u; } c; c.f = v;
  const unsigned o = (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
  return v == P5_NEG_INF ? 0ull : (((unsigned long long)o << 32) | (unsigned long long)(~(unsigned)key));
}
// rank of `mine` among the n keys of k64 (n padded to a multiple of 2 with zeros): two candidates per 16-byte LDS read, one 64-bit compare each
// (the (score, key) pair compare of round 4 took ~6 VALU operations per candidate: 13 us of the beam step at 16 beams)
__device__ static __forceinline__ int p5_rank_of(const unsigned long long* k64, int n, unsigned long long mine) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  int rank = 0;
  const int n2 = (n + 1) >> 1;
#pragma unroll 8
  for (int u = 0; u < n2; ++u) {
    const u64x2 k = ((const u64x2*)k64)[u];
    rank += (k[0] > mine) ? 1 : 0;
    rank += (k[1] > mine) ? 1 : 0;
  }
  return rank;
}
struct P5BeamState {
  int* run_seq; int* run_seq_next;   // [B,K,max_len]
  float* run_score;                  // [B,K]
  int* run_node;                     // [B,K] trie node after the beam's prefix, -1 = dead
  int* fin_seq; int* fin_seq_next;   // [B,K,max_len]
  float* fin_score;                  // [B,K]
  int* fin_flag;                     // [B,K]
  int* fin_len;                      // [B,K]
  int* unsat;                        // [B]
  int* anc; int* anc_next;           // [max_len, R]
  int64_t* last_tok;                 // [R] decoder input for the next step
  int* flags;                        // [0] any_unsat, [1] not_all_hits (reset by the last workgroup of each beam step), [2] cur_len, [3] arrivals, [4] done
  float* x32;                        // optional [R, d] fp32 residual stream of the latency-shaped decode step: the beam step writes
  const float* E32;                  //   the NEXT step's input embeddings E32[token] into it (one launch per step fewer)
  int d;
  int* hist;                         // optional (p5_generate_draft): per executed step and beam (parent beam, token, trie node, live) -- what the
                                     //   fp32 verification pass (p5_verify.h) needs to know about the search; [0] = steps executed
};

// ---- shared tail of the beam step: HF steps d-g (utils.py:3131-3204, 3008-3075) from the item's top-2K candidate list in
// `sh`, materialisation of the new finished / running sets, step counter + stop flag.  `st` is already parity-swapped. ----
// longest hypothesis the decode kernels hold (beam bookkeeping below in LDS: 3 x P5_MAX_K x P5_MAX_LEN ints = 96 KiB; ancestry gather of
// p5_dec_self_attn2_kernel unrolled over P5_MAX_LEN / 8 passes); p5_generate / p5_decode_begin refuse anything longer (p5_lib.hip) -- one
// constant for both sides.  OpenP5 decodes item ids of <= 30 tokens (DistributedRunner.py:361-371, max_length=30)
#ifndef P5_MAX_LEN
#define P5_MAX_LEN 128
#endif
struct P5BeamSh {
  float top_lp[P5_MAX_K2], run_lp[P5_MAX_K2], msc[P5_MAX_K + P5_MAX_K2];
  int top_beam[P5_MAX_K2], top_tok[P5_MAX_K2], top_node[P5_MAX_K2], hit[P5_MAX_K2];
  int sel_run[P5_MAX_K], fin_src[P5_MAX_K], fin_fl[P5_MAX_K], fin_ln[P5_MAX_K];
  float fin_sc[P5_MAX_K], run_sc[P5_MAX_K];
  int s_nothit;
  // the item's OLD state, fetched at kernel entry together with the candidate lists (one round trip instead of a chain of
  // dependent ones further down: the kernel is one workgroup per item and spent 77 % of its cycles waiting on such loads)
  int old_unsat, old_node[P5_MAX_K], old_coff[P5_MAX_K];
  float old_fin_score[P5_MAX_K];
  int old_fin_flag[P5_MAX_K], old_fin_len[P5_MAX_K];
  int old_run_seq[P5_MAX_K * P5_MAX_LEN], old_fin_seq[P5_MAX_K * P5_MAX_LEN], old_anc[P5_MAX_K * P5_MAX_LEN];   // [Kb][max_len], [Kb][max_len], [pos+1][Kb]
};

// issue every load of the item's old state (see P5BeamSh); the caller's next __syncthreads() publishes it
__device__ static __forceinline__ void p5_beam_prefetch(const P5BeamState& st, P5BeamSh& sh, int b, int tid, int Kb, int max_len, int R, int cur_len,
                                                        const int* __restrict__ child_off) {
  if (tid == 0) sh.old_unsat = st.unsat[b];
  if (tid < Kb) {
    const int nd = st.run_node[b * Kb + tid];
    sh.old_node[tid] = nd;
    sh.old_coff[tid] = nd >= 0 ? child_off[nd] : 0;
    sh.old_fin_score[tid] = st.fin_score[b * Kb + tid];
    sh.old_fin_flag[tid] = st.fin_flag[b * Kb + tid];
    sh.old_fin_len[tid] = st.fin_len[b * Kb + tid];
  }
  for (int t = tid; t < Kb * max_len; t += 256) {
    sh.old_run_seq[t] = st.run_seq[(size_t)b * Kb * max_len + t];
    sh.old_fin_seq[t] = st.fin_seq[(size_t)b * Kb * max_len + t];
  }
  const int pos = cur_len - 1;
  for (int t = tid; t < Kb * pos; t += 256) {                // ancestry rows 0 .. pos-1 of the item's beams
    const int p = t / Kb, j = t - p * Kb;
    sh.old_anc[t] = st.anc[(size_t)p * R + b * Kb + j];
  }
}

__device__ static __forceinline__ void p5_beam_tail(P5BeamState& st, P5BeamSh& sh, int b, int tid, int Kb, int K2, int max_len, int eos_id, int R,
                                                    int cur_len) {
  float* top_lp = sh.top_lp; float* run_lp = sh.run_lp; float* msc = sh.msc;
  int* top_beam = sh.top_beam; int* top_tok = sh.top_tok; int* top_node = sh.top_node; int* hit = sh.hit;
  int* sel_run = sh.sel_run; int* fin_src = sh.fin_src; int* fin_fl = sh.fin_fl; int* fin_ln = sh.fin_ln;
  float* fin_sc = sh.fin_sc; float* run_sc = sh.run_sc;
  int& s_nothit = sh.s_nothit;
  // ---- d/e: hits, running beams = stable top-K of run_lp ----
  const bool at_max = (cur_len + 1 >= max_len);
  if (tid < K2) {
    const int h = (top_tok[tid] == eos_id) || at_max;
    hit[tid] = h;
    run_lp[tid] = top_lp[tid] + (h ? -1.0e9f : 0.f);
    if (!h) atomicAdd(&s_nothit, 1);
  }
  __syncthreads();
  if (tid < K2) {
    int rank = 0;
    const float v = run_lp[tid];
    for (int j = 0; j < K2; ++j) rank += (run_lp[j] > v || (run_lp[j] == v && j < tid)) ? 1 : 0;
    if (rank < Kb) { sel_run[rank] = tid; run_sc[rank] = v; }
  }
  // ---- f: finished beams = stable top-K over [old finished ; new candidates] ----
  const bool uns = sh.old_unsat != 0;
  if (tid < Kb) msc[tid] = sh.old_fin_score[tid];
  else if (tid < Kb + K2) {
    const int i = tid - Kb;
    float v = top_lp[i] / (float)cur_len;
    if (!uns) v += -1.0e9f;
    if (!(hit[i] && i < Kb)) v += -1.0e9f;
    msc[tid] = v;
  }
  __syncthreads();
  if (tid < Kb + K2) {
    int rank = 0;
    const float v = msc[tid];
    for (int j = 0; j < Kb + K2; ++j) rank += (msc[j] > v || (msc[j] == v && j < tid)) ? 1 : 0;
    if (rank < Kb) {
      fin_sc[rank] = v;
      if (tid < Kb) { fin_src[rank] = tid; fin_fl[rank] = sh.old_fin_flag[tid]; fin_ln[rank] = sh.old_fin_len[tid]; }
      else { const int i = tid - Kb; fin_src[rank] = -(i + 1); fin_fl[rank] = (hit[i] && i < Kb) ? 1 : 0; fin_ln[rank] = cur_len; }
    }
  }
  __syncthreads();
  // ---- g: early-stop heuristic with the NEW running / finished sets ----
  if (tid == 0) {
    const float best_possible = run_sc[0] / (float)cur_len;   // (cur_len+1) - prompt_len(1)
    float mn = fin_sc[0];
    for (int j = 1; j < Kb; ++j) mn = fminf(mn, fin_sc[j]);
    bool any = false;
    for (int j = 0; j < Kb; ++j) any = any || (best_possible > (fin_fl[j] ? mn : -1.0e9f));
    const int new_unsat = (uns && any) ? 1 : 0;
    st.unsat[b] = new_unsat;
    if (new_unsat) atomicAdd(&st.flags[0], 1);
    if (s_nothit > 0) atomicAdd(&st.flags[1], 1);
  }
  __syncthreads();
  // ---- materialise the new finished set (reads OLD fin_seq / run_seq, writes fin_seq_next) ----
  for (int t = tid; t < Kb * max_len; t += 256) {
    const int j = t / max_len, p = t % max_len;
    int v;
    if (fin_src[j] >= 0) v = sh.old_fin_seq[fin_src[j] * max_len + p];
    else {
      const int i = -fin_src[j] - 1;
      v = (p == cur_len) ? top_tok[i] : sh.old_run_seq[top_beam[i] * max_len + p];
    }
    st.fin_seq_next[((size_t)b * Kb + j) * max_len + p] = v;
  }
  if (tid < Kb) {
    st.fin_len[b * Kb + tid] = fin_ln[tid];
    st.fin_score[b * Kb + tid] = fin_sc[tid];
    st.fin_flag[b * Kb + tid] = fin_fl[tid];
  }
  // ---- new running sequences / nodes / ancestry (reads OLD run_seq / anc, writes *_next) ----
  for (int t = tid; t < Kb * max_len; t += 256) {
    const int j = t / max_len, p = t % max_len;
    const int i = sel_run[j];
    st.run_seq_next[((size_t)b * Kb + j) * max_len + p] =
        (p == cur_len) ? top_tok[i] : sh.old_run_seq[top_beam[i] * max_len + p];
  }
  const int pos = cur_len - 1;   // K/V of this step were stored at `pos` by row (b*Kb + old beam)
  for (int t = tid; t < Kb * (pos + 1); t += 256) {
    const int j = t / (pos + 1), p = t % (pos + 1);
    const int parent = top_beam[sel_run[j]];
    st.anc_next[(size_t)p * R + b * Kb + j] = (p == pos) ? b * Kb + parent : sh.old_anc[p * Kb + parent];
  }
  if (tid < Kb) {
    const int i = sel_run[tid];
    st.run_node[b * Kb + tid] = top_node[i];
    st.last_tok[b * Kb + tid] = (int64_t)top_tok[i];
    st.run_score[b * Kb + tid] = run_sc[tid];
    if (st.hist) {     // history of the search for the verification pass (p5_verify.h): step `cur_len` produced the token at position cur_len
      int* hrec = st.hist + 4 + (size_t)cur_len * 4 * R + b * Kb + tid;
      hrec[0] = top_beam[i]; hrec[R] = top_tok[i]; hrec[2 * (size_t)R] = top_node[i]; hrec[3 * (size_t)R] = run_sc[tid] > -1.0e8f ? 1 : 0;
    }
  }
  if (st.x32) {      // decoder input of the next step: x32[row, :] = E32[token, :]  (fp32 master table, P5_T5.py:94-100 for the decoder)
    // four independent 16-byte loads per thread before the first store (one load -> store round trip per iteration cost ~2 us each: 8
    // iterations at 16 beams)
    const int d4 = st.d >> 2, n4 = Kb * d4;
    for (int t0 = tid; t0 < n4; t0 += 4 * 256) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * 256;
        if (t < n4) { const int j = t / d4, c4 = t - j * d4; v[u] = *(const f32x4*)(st.E32 + (size_t)top_tok[sel_run[j]] * st.d + c4 * 4); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * 256;
        if (t < n4) { const int j = t / d4, c4 = t - j * d4; *(f32x4*)(st.x32 + ((size_t)(b * Kb + j)) * st.d + c4 * 4) = v[u]; }
      }
    }
  }
  // the step counter advances once every workgroup of this launch is done with it (they all read it on entry): the last one
  // to arrive bumps it -- this used to be a launch of its own
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(&st.flags[3], 1) == (int)gridDim.x - 1) {
      st.flags[3] = 0;
      st.flags[2] = cur_len + 1;
      // HF's global stop condition (utils.py:3055-3075), decided on the device: every later launch of this generate call
      // sees flags[4] and returns at once, so the host never has to read anything back between steps
      const int any_unsat = atomicExch(&st.flags[0], 0), not_all_hits = atomicExch(&st.flags[1], 0);     // (reset for the next step)
      if (!(any_unsat > 0 && not_all_hits > 0)) st.flags[4] = 1;
    }
  }
}

// ---- one workgroup per batch item: merge the rows' sorted top lists into the item's top-2K, then HF steps d-g
// (utils.py:3131-3204, 3008-3075) with rank-based stable selections done in parallel ----
__global__ __launch_bounds__(256) void p5_beam_step_kernel(P5BeamState st, const float* __restrict__ row_top_score,
                                                          const int* __restrict__ row_top_c, const int* __restrict__ row_n_top,
                                                          const int* __restrict__ child_off, const int* __restrict__ child_tok,
                                                          const int* __restrict__ child_node, int max_c, int Kb, int max_len,
                                                          int eos_id, int R) {
  if (st.flags[4]) return;       // the search stopped in an earlier step
  const int cur_len = st.flags[2];
  if ((cur_len & 1) == 0) {      // even step: the "next" buffers of the previous step are the current ones
    int* t;
    t = st.run_seq; st.run_seq = st.run_seq_next; st.run_seq_next = t;
    t = st.fin_seq; st.fin_seq = st.fin_seq_next; st.fin_seq_next = t;
    t = st.anc; st.anc = st.anc_next; st.anc_next = t;
  }
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ __attribute__((aligned(16))) float cs[P5_MAX_K * P5_MAX_K2];
  __shared__ P5BeamSh sh;
  float* top_lp = sh.top_lp;
  int* top_beam = sh.top_beam; int* top_tok = sh.top_tok; int* top_node = sh.top_node;
  int& s_nothit = sh.s_nothit;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int K2 = 2 * Kb;
  // candidate pool: Kb rows x (<= K2) entries, flat index j*K2 + i.  The lists are read unconditionally (the buffers are fully
  // allocated; entries beyond a row's count are ignored) so that counts, scores and children travel in one round trip
  p5_beam_prefetch(st, sh, b, tid, Kb, max_len, R, cur_len, child_off);
  for (int t = tid; t < Kb * K2; t += 256) {
    const int j = t / K2, i = t % K2;
    const float v = row_top_score[(size_t)(b * Kb + j) * K2 + i];
    cs[t] = i < row_n_top[b * Kb + j] ? v : P5_NEG_INF;
  }
  if (tid == 0) s_nothit = 0;
  __syncthreads();
  __shared__ __attribute__((aligned(16))) unsigned long long k64[P5_RANK_POOL + 2];
  const bool by_rank = Kb * K2 <= P5_RANK_POOL;
  if (by_rank) {
    // every candidate computes its own rank in the (score desc, beam*max_c + child asc) order -- all in parallel instead of
    // 2K rounds of block-wide arg-max; the 2K best land at their rank
    for (int t = tid; t < Kb * K2; t += 256)
      k64[t] = p5_rank_key(cs[t], (t / K2) * max_c + row_top_c[(size_t)(b * Kb + t / K2) * K2 + t % K2]);
    if (tid < 2) k64[Kb * K2 + tid] = 0ull;
    if (tid < K2) { top_lp[tid] = P5_NEG_INF; top_beam[tid] = 0; top_tok[tid] = 0; top_node[tid] = -1; }   // fewer than 2K candidates
    __syncthreads();
    for (int t = tid; t < Kb * K2; t += 256) {
      const unsigned long long mine = k64[t];
      if (mine == 0ull) continue;
      const int rank = p5_rank_of(k64, Kb * K2, mine);
      if (rank < K2) {
        const int j = t / K2, c = (int)(~(unsigned)mine) - j * max_c;
        top_lp[rank] = cs[t]; top_beam[rank] = j;
        top_tok[rank] = child_tok[sh.old_coff[j] + c];
        top_node[rank] = child_node[sh.old_coff[j] + c];
      }
    }
    __syncthreads();
  }
  for (int it = 0; it < (by_rank ? 0 : K2); ++it) {
    float bv = P5_NEG_INF;
    int bi = 0x7fffffff;          // tie-break key = beam * max_c + child  (== HF's flat beam*V + token order)
    for (int t = tid; t < Kb * K2; t += 256) {
      const float v = cs[t];
      if (v == P5_NEG_INF) continue;
      const int j = t / K2, i = t % K2;
      const int key = j * max_c + row_top_c[(size_t)(b * Kb + j) * K2 + i];
      if (v > bv || (v == bv && key < bi)) { bv = v; bi = key; }
    }
    block_argmax(bv, bi, s_val, s_idx);
    if (tid == 0) {
      if (bi != 0x7fffffff) {
        const int j = bi / max_c, c = bi % max_c;
        top_lp[it] = bv; top_beam[it] = j;
        top_tok[it] = child_tok[sh.old_coff[j] + c];
        top_node[it] = child_node[sh.old_coff[j] + c];
        for (int i = 0; i < K2; ++i)        // mark taken (rows are short: <= K2 entries)
          if (cs[j * K2 + i] != P5_NEG_INF && row_top_c[(size_t)(b * Kb + j) * K2 + i] == c) { cs[j * K2 + i] = P5_NEG_INF; break; }
      } else {   // fewer than 2K allowed continuations: HF would pick arbitrary -inf entries
        top_lp[it] = P5_NEG_INF; top_beam[it] = 0; top_tok[it] = 0; top_node[it] = -1;
      }
    }
    __syncthreads();
  }
  p5_beam_tail(st, sh, b, tid, Kb, K2, max_len, eos_id, R, cur_len);
}


// initial state: every beam at the trie node reached by the decoder start token, scores [0, -1e9, ...]
__global__ __launch_bounds__(256) void p5_beam_init_kernel(P5BeamState st, const int* __restrict__ child_off, const int* __restrict__ child_tok,
                                                          const int* __restrict__ child_node, const int* __restrict__ roots, int B, int Kb, int max_len,
                                                          int start_id) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int R = B * Kb;
  if (i < R * max_len) {
    const int v0 = ((i % max_len) == 0) ? start_id : 0;
    st.run_seq[i] = v0; st.run_seq_next[i] = v0; st.fin_seq[i] = v0; st.fin_seq_next[i] = v0;
    st.anc[i] = 0; st.anc_next[i] = 0;
  }
  if (i < R) {
    int nd = -1;
    const int root = roots ? roots[i / Kb] : 0;
    for (int c = child_off[root]; c < child_off[root + 1]; ++c)
      if (child_tok[c] == start_id) nd = child_node[c];
    st.run_node[i] = nd;
    st.run_score[i] = (i % Kb == 0) ? 0.f : -1.0e9f;
    st.fin_score[i] = -1.0e9f;
    st.fin_flag[i] = 0;
    st.fin_len[i] = 0;
    st.last_tok[i] = start_id;
  }
  if (st.x32) {      // decoder input of the first step: every row = E32[decoder start token]
    const int d4 = st.d >> 2;
    for (int t = i; t < R * d4; t += gridDim.x * 256)
      *(f32x4*)(st.x32 + (size_t)t * 4) = *(const f32x4*)(st.E32 + (size_t)start_id * st.d + (t % d4) * 4);
  }
  if (i < B) st.unsat[i] = 1;
  if (i == 0) { st.flags[0] = 0; st.flags[1] = 0; st.flags[2] = 1; st.flags[3] = 0; st.flags[4] = 0; }
}


// =====================================================================================================================
// Candidate scores for the streaming head (p5_head_lse_kernel), one workgroup per decode row:
//   1. log-sum-exp of the row from the per-tile (max, sum exp) partials;
//   2. the logits the search needs -- the trie children of the row's beam -- recomputed as dot products hn[row] . E[token]
//      (a handful, except at a high-fan-out trie level), minus lse, plus the running score; children whose subtree is excluded
//      for this user (filtered evaluation, DistributedRunner.py:286-297) are dropped;
//   3. the row's best K2 = 2K of them in (score desc, child asc) order: rank counting for small fan-outs, a 4-pass radix
//      select of the K2-th score first for large ones (the global top-2K of an item is contained in the union of its rows'
//      top-2K lists; p5_beam_step_kernel merges them).
// This is "log_softmax over the full vocabulary, then mask to the allowed tokens" (HF generation/utils.py:3388-3389) without
// ever holding a [R, V] tensor.
// =====================================================================================================================
#define P5_POOL 2048
template <class T>
__global__ __launch_bounds__(256) void p5_dec_score2_kernel(float* __restrict__ top_score, int* __restrict__ top_c, int* __restrict__ n_top,
                                                           float* __restrict__ cand_scratch, const float* __restrict__ part_m,
                                                           const float* __restrict__ part_s, int ntiles, const T* __restrict__ hn,
                                                           const T* __restrict__ E, int d, float alpha, const int* __restrict__ node,
                                                           const float* __restrict__ run_score, const int* __restrict__ child_off,
                                                           const int* __restrict__ child_tok, const int* __restrict__ child_node,
                                                           const uint32_t* __restrict__ excluded, int excl_words, int Kb, int max_c, int K2,
                                                           const int* __restrict__ done) {
  constexpr int EPF = TT<T>::EPF;
  __shared__ float sm[4], ss[4];
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float pool_v[P5_POOL];
  __shared__ float cmp_v[P5_MAX_K2];
  __shared__ int cmp_k[P5_MAX_K2];
  __shared__ int hist[256];
  __shared__ unsigned s_sel[4];      // [0] selected bin, [1] count above it / compact count, [2] "take everything" flag, [3] placed
  if (done && *done) return;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = node[r];
  if (nd < 0) {  // dead beam: no candidates (uniform per block)
    if (tid == 0) n_top[r] = 0;
    return;
  }
  // the trie lookups do not depend on the reduction below: issue them first
  const int c0 = child_off[nd];
  int nc = child_off[nd + 1] - c0;
  nc = nc < max_c ? nc : max_c;
  const float rs = run_score[r];
  // ---- 1. log-sum-exp over the vocabulary tiles ----
  float m = P5_NEG_INF, sum = 0.f;
  for (int t = tid; t < ntiles; t += 256) {
    const float pm = part_m[(size_t)r * ntiles + t], ps = part_s[(size_t)r * ntiles + t];
    if (pm > m) { sum = sum * expf(m - pm) + ps; m = pm; }
    else if (pm != P5_NEG_INF) sum += ps * expf(pm - m);
  }
  {
    const float wm_ = wave_max(m);
    sum = wave_sum(m == P5_NEG_INF ? 0.f : sum * expf(m - wm_));
    if (lane == 0) { sm[wave] = wm_; ss[wave] = sum; }
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    sum = 0.f;
    for (int w = 0; w < 4; ++w) sum += (sm[w] == P5_NEG_INF) ? 0.f : ss[w] * expf(sm[w] - m);
  }
  const float lse = m + logf(sum);
  // ---- 2. children's scores: 8 lanes per candidate, 32 candidates per pass ----
  const bool in_lds = nc <= P5_POOL;
  float* pv = in_lds ? pool_v : cand_scratch + (size_t)r * max_c;
  const uint32_t* ex = excluded ? excluded + (size_t)(r / Kb) * excl_words : nullptr;
  {
    // the row's hn stays in registers (this lane's 16-byte pieces: element offsets sub*EPF + k*8*EPF), every load of a
    // candidate's E row is issued before the first FMA, and the next pass's token id is fetched a pass ahead
    constexpr int MAXP = 1024 / (8 * EPF);          // d_model <= 1024
    const int grp = tid >> 3, sub = tid & 7;
    const T* hp = hn + (size_t)r * d;
    const int np = d / (8 * EPF);
    u32x4 hx[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) hx[k] = k < np ? ld16(hp + sub * EPF + k * 8 * EPF) : zero16();
    int tok_next = grp < nc ? child_tok[c0 + grp] : 0;
    for (int i0 = 0; i0 < nc; i0 += 32) {
      const int i = i0 + grp;
      const int tok = tok_next;
      if (i + 32 < nc) tok_next = child_tok[c0 + i + 32];
      float acc = 0.f;
      if (i < nc) {
        const T* ep = E + (size_t)tok * d + sub * EPF;
#pragma unroll
        for (int k0 = 0; k0 < MAXP; k0 += 8) {          // 8 x 16 bytes of the E row in flight per lane
          if (k0 < np) {
            u32x4 wr[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wr[k] = k0 + k < np ? ld16(ep + (k0 + k) * 8 * EPF) : zero16();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float x[8], w[8];
              unpack16<T>(hx[k0 + k], x);
              unpack16<T>(wr[k], w);
#pragma unroll
              for (int e = 0; e < EPF; ++e) acc += x[e] * w[e];
            }
          }
        }
      }
      acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
      if (i < nc && sub == 0) {
        float v = (acc * alpha - lse) + rs;
        if (ex) {
          const int cn = child_node[c0 + i];
          if ((ex[cn >> 5] >> (cn & 31)) & 1u) v = P5_NEG_INF;
        }
        pv[i] = v;
      }
    }
  }
  __syncthreads();
  // ---- 3. the row's top-K2 ----
  auto okey = [](float v) -> unsigned {           // order-preserving float -> uint
    union { float f; unsigned u; } c; c.f = v;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
  };
  const int want = K2 < nc ? K2 : nc;
  if (in_lds) {
    const float* rv = pool_v;
    const int* rk = nullptr;            // nullptr: the candidate's child index is its position
    int n = nc;
    if (nc > 256) {
      // radix select of the K2-th largest finite score: after the four passes `prefix` is its key and `remaining` the number of
      // candidates needed from those TIED with it (dead beams carry -1e9 + log-prob, which fp32 rounds to the same value for
      // every child: a thousand-way tie is the normal case there, and HF's order among ties is the child order)
      unsigned prefix = 0, mask = 0;
      int remaining = K2;
      bool all = false;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += 256) {
          const float v = pool_v[i];
          if (v == P5_NEG_INF) continue;
          const unsigned k = okey(v);
          if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
        }
        __syncthreads();
        if (wave == 0) {
          // lane l owns bins 255-4l .. 252-4l (descending); `before` = candidates in strictly higher bins than this lane's
          int h4[4], own = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) { h4[q] = hist[255 - (lane * 4 + q)]; own += h4[q]; }
          int before = 0, tot = 0;
          for (int l2 = 0; l2 < 64; ++l2) {
            const int o = __shfl(own, l2);
            before += (l2 < lane) ? o : 0;
            tot += o;
          }
          if (pass == 0 && lane == 0) s_sel[2] = (tot <= remaining) ? 1u : 0u;      // no more finite candidates than K2: take them all
          if (before < remaining && before + own >= remaining) {
            int cum = before;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (cum < remaining && cum + h4[q] >= remaining) { s_sel[0] = (unsigned)(255 - (lane * 4 + q)); s_sel[1] = (unsigned)cum; }
              cum += h4[q];
            }
          }
        }
        __syncthreads();
        if (s_sel[2]) { all = true; break; }
        prefix |= s_sel[0] << shift;
        mask |= 255u << shift;
        remaining -= (int)s_sel[1];
        __syncthreads();
      }
      // ordered compaction: everything above the threshold, plus the first `remaining` (in child order) of the ties.  Each
      // thread owns a contiguous run of candidates so that an exclusive scan of the per-thread tie counts yields tie ranks.
      const int per = (nc + 255) / 256, i_lo = tid * per, i_hi = (i_lo + per < nc) ? i_lo + per : nc;
      int ties = 0;
      if (!all)
        for (int i = i_lo; i < i_hi; ++i) ties += (pool_v[i] != P5_NEG_INF && okey(pool_v[i]) == prefix) ? 1 : 0;
      __syncthreads();
      hist[tid] = ties;
      if (tid == 0) s_sel[1] = 0u;
      __syncthreads();
      if (wave == 0) {
        int h4[4], own = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { h4[q] = hist[lane * 4 + q]; own += h4[q]; }
        int before = 0;
        for (int l2 = 0; l2 < 64; ++l2) {
          const int o = __shfl(own, l2);
          before += (l2 < lane) ? o : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { hist[lane * 4 + q] = before; before += h4[q]; }
      }
      __syncthreads();
      int tie_rank = hist[tid];
      for (int i = i_lo; i < i_hi; ++i) {
        const float v = pool_v[i];
        if (v == P5_NEG_INF) continue;
        const unsigned k = okey(v);
        bool keep = all || k > prefix;
        if (!all && k == prefix) { keep = tie_rank < remaining; ++tie_rank; }
        if (keep) {
          const unsigned at = atomicAdd(&s_sel[1], 1u);
          if (at < (unsigned)P5_MAX_K2) { cmp_v[at] = v; cmp_k[at] = i; }
        }
      }
      __syncthreads();
      n = (int)s_sel[1];               // == min(K2, finite candidates)
      rv = cmp_v; rk = cmp_k;
    }
    if (tid == 0) s_sel[3] = 0u;
    __syncthreads();
    for (int t = tid; t < n; t += 256) {
      const float v = rv[t];
      if (v == P5_NEG_INF) continue;
      const int key = rk ? rk[t] : t;
      int rank = 0;
      for (int u = 0; u < n; ++u) {
        const float vu = rv[u];
        rank += (vu > v || (vu == v && (rk ? rk[u] : u) < key)) ? 1 : 0;
      }
      if (rank < K2) {
        top_score[(size_t)r * K2 + rank] = v;
        top_c[(size_t)r * K2 + rank] = key;
        atomicAdd(&s_sel[3], 1u);
      }
    }
    __syncthreads();
    if (tid == 0) n_top[r] = (int)s_sel[3];       // finite candidates placed (ranks 0 .. count-1 are exactly the ones written)
    return;
  }
  // ---- fan-outs beyond the LDS pool: K2 rounds of block-wide arg-max over the global scratch ----
  for (int it = 0; it < want; ++it) {
    float bv = P5_NEG_INF;
    int bi = 0x7fffffff;
    for (int c = tid; c < nc; c += 256) {
      const float v = pv[c];
      if (v == P5_NEG_INF) continue;
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    block_argmax(bv, bi, s_val, s_idx);
    if (bi == 0x7fffffff) {            // only excluded children left; stop early
      if (tid == 0) n_top[r] = it;
      return;
    }
    if (tid == 0) {
      top_score[(size_t)r * K2 + it] = bv;
      top_c[(size_t)r * K2 + it] = bi;
      pv[bi] = P5_NEG_INF;   // taken (a genuine -inf candidate is never selected above)
    }
    __syncthreads();
  }
  if (tid == 0) n_top[r] = want;
}

// =====================================================================================================================
// Forced-prefix fast-forward (round 5).  Every OpenP5 item id starts with the same tokens -- "<dataset> item _" (data/.../indexing: the
// target template `{dataset} {target}`) -- so for the first F steps of HF's beam search every beam of every user has
// exactly ONE allowed token: beam 0 carries the real score, beams 1..K-1 the -1e9 of HF's initial state, nothing can finish, and step s
// only adds log p(f_s | start, f_1 .. f_{s-1}) to beam 0.  Those F steps are therefore ONE teacher-forced decoder pass over F positions per
// user (B x F rows through the throughput kernels, causal self-attention) instead of F latency-bound decode steps over B x K rows: same
// numbers (a decoder position never sees later tokens), F - 1 steps saved -- 4 of the 8 steps of the ML-1M-shaped benchmark trie.
// The pass leaves per position the self-attention K/V of every layer (scattered into the step cache at row b*K, which the ancestry
// table maps every beam of the user to) and the per-token NLL; p5_beam_forced_kernel builds the beam state HF would have after F steps.
// =====================================================================================================================
#define P5_FF_MAX 16
struct P5Forced { int n; int tok[P5_FF_MAX]; int node[P5_FF_MAX]; };      // f_1 .. f_n and the trie node each one leads to

__global__ __launch_bounds__(256) void p5_ff_labels_kernel(int64_t* __restrict__ labels, P5Forced ff, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * ff.n) labels[i] = (int64_t)ff.tok[i % ff.n];
}
// K/V of the forced positions: qkv [B*F, 3*inner] of one layer -> cache [max_len][R][2*inner] at (position p, row b*Kb)
template <class T>
__global__ __launch_bounds__(256) void p5_ff_cache_kernel(T* __restrict__ cache, const T* __restrict__ qkv, int F, int Kb, int R, int inner) {
  constexpr int EPF = TT<T>::EPF;
  const int bp = blockIdx.x, b = bp / F, p = bp % F;
  const T* src = qkv + (size_t)bp * 3 * inner + inner;
  T* dst = cache + ((size_t)p * R + (size_t)b * Kb) * 2 * inner;
  for (int c = threadIdx.x; c < 2 * inner / EPF; c += 256) st16(dst + c * EPF, ld16(src + c * EPF));
}
// the beam state after F forced steps (p5_beam_init_kernel has run): see the header above
__global__ __launch_bounds__(256) void p5_beam_forced_kernel(P5BeamState st, P5Forced ff, const float* __restrict__ nll, int B, int Kb, int max_len,
                                                            int start_id) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int R = B * Kb, F = ff.n;
  if (i < R * max_len) {
    const int p = i % max_len;
    const int v = (p == 0) ? start_id : (p <= F ? ff.tok[p - 1] : 0);
    st.run_seq[i] = v; st.run_seq_next[i] = v;
    // ancestry tables are [position][row]: K/V of position p < F live in the row of the user's beam 0
    const int pp = i / R, r = i % R;
    if (pp < F) { st.anc[i] = (r / Kb) * Kb; st.anc_next[i] = (r / Kb) * Kb; }
  }
  if (i < R) {
    const int b = i / Kb, j = i % Kb;
    float sc = 0.f;
    for (int p = 0; p < F; ++p) sc += -nll[b * F + p];        // HF adds one log-probability per step, in step order
    st.run_score[i] = (j == 0) ? sc : -1.0e9f;
    st.run_node[i] = ff.node[F - 1];
    st.last_tok[i] = (int64_t)ff.tok[F - 1];
    if (st.hist)
      for (int s = 1; s <= F; ++s) {
        int* hrec = st.hist + 4 + (size_t)s * 4 * R + i;
        hrec[0] = j; hrec[R] = ff.tok[s - 1]; hrec[2 * (size_t)R] = ff.node[s - 1]; hrec[3 * (size_t)R] = (j == 0) ? 1 : 0;
      }
  }
  if (st.x32) {
    const int d4 = st.d >> 2;
    for (int t = i; t < R * d4; t += gridDim.x * 256)
      *(f32x4*)(st.x32 + (size_t)t * 4) = *(const f32x4*)(st.E32 + (size_t)ff.tok[F - 1] * st.d + (t % d4) * 4);
  }
  if (i == 0) st.flags[2] = F + 1;
}

// results of the search: the finished set written by the last EXECUTED step lives in the "next" buffer of that step's parity
__global__ __launch_bounds__(256) void p5_beam_finalize_kernel(int* __restrict__ out_seq, float* __restrict__ out_score, int* __restrict__ out_len,
                                                              P5BeamState st, int R, int max_len) {
  const int steps = st.flags[2] - 1;                       // beam steps executed (cur_len starts at 1)
  const int* src = (steps & 1) ? st.fin_seq_next : st.fin_seq;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < R * max_len) out_seq[i] = src[i];
  if (i < R) { out_score[i] = st.fin_score[i]; out_len[i] = st.fin_len[i]; }
  if (st.hist && i == 0) st.hist[0] = steps;
}
