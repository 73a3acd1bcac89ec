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
                                                          int excl_words, int Kb, int max_c, int K2) {
  __shared__ float sm[4], ss[4];
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
  int* flags;                        // [0] any_unsat, [1] not_all_hits (zeroed at the start of each step), [2] cur_len, [3] arrivals, [4] done
};

// ---- one workgroup per batch item: merge the rows' sorted top lists into the item's top-2K, then HF steps d-g
// (utils.py:3131-3204, 3008-3075) with rank-based stable selections done in parallel ----
#define P5_MAX_K 64
#define P5_MAX_K2 128
__global__ __launch_bounds__(256) void p5_beam_step_kernel(P5BeamState st, const float* __restrict__ row_top_score,
                                                          const int* __restrict__ row_top_c, const int* __restrict__ row_n_top,
                                                          const int* __restrict__ child_off, const int* __restrict__ child_tok,
                                                          const int* __restrict__ child_node, int max_c, int Kb, int max_len,
                                                          int eos_id, int R) {
  const int cur_len = st.flags[2];
  if ((cur_len & 1) == 0) {      // even step: the "next" buffers of the previous step are the current ones
    int* t;
    t = st.run_seq; st.run_seq = st.run_seq_next; st.run_seq_next = t;
    t = st.fin_seq; st.fin_seq = st.fin_seq_next; st.fin_seq_next = t;
    t = st.anc; st.anc = st.anc_next; st.anc_next = t;
  }
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float cs[P5_MAX_K * P5_MAX_K2];
  __shared__ float top_lp[P5_MAX_K2], run_lp[P5_MAX_K2], msc[P5_MAX_K + P5_MAX_K2];
  __shared__ int top_beam[P5_MAX_K2], top_tok[P5_MAX_K2], top_node[P5_MAX_K2], hit[P5_MAX_K2];
  __shared__ int sel_run[P5_MAX_K], fin_src[P5_MAX_K], fin_fl[P5_MAX_K], fin_ln[P5_MAX_K];
  __shared__ float fin_sc[P5_MAX_K], run_sc[P5_MAX_K];
  __shared__ int s_nothit;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int K2 = 2 * Kb;
  // candidate pool: Kb rows x (<= K2) entries, flat index j*K2 + i
  for (int t = tid; t < Kb * K2; t += 256) {
    const int j = t / K2, i = t % K2;
    cs[t] = i < row_n_top[b * Kb + j] ? row_top_score[(size_t)(b * Kb + j) * K2 + i] : P5_NEG_INF;
  }
  if (tid == 0) s_nothit = 0;
  __syncthreads();
  __shared__ int ckey[1024];
  const bool by_rank = Kb * K2 <= 1024;
  if (by_rank) {
    // every candidate computes its own rank in the (score desc, beam*max_c + child asc) order -- all in parallel instead of
    // 2K rounds of block-wide arg-max; the 2K best land at their rank
    for (int t = tid; t < Kb * K2; t += 256)
      ckey[t] = (cs[t] == P5_NEG_INF) ? 0x7fffffff : (t / K2) * max_c + row_top_c[(size_t)(b * Kb + t / K2) * K2 + t % K2];
    if (tid < K2) { top_lp[tid] = P5_NEG_INF; top_beam[tid] = 0; top_tok[tid] = 0; top_node[tid] = -1; }   // fewer than 2K candidates
    __syncthreads();
    for (int t = tid; t < Kb * K2; t += 256) {
      const float v = cs[t];
      if (v == P5_NEG_INF) continue;
      const int key = ckey[t];
      int rank = 0;
      for (int u = 0; u < Kb * K2; ++u) {
        const float vu = cs[u];
        rank += (vu > v || (vu == v && ckey[u] < key)) ? 1 : 0;
      }
      if (rank < K2) {
        const int j = t / K2, c = key - j * max_c;
        const int nd = st.run_node[b * Kb + j];
        top_lp[rank] = v; top_beam[rank] = j;
        top_tok[rank] = child_tok[child_off[nd] + c];
        top_node[rank] = child_node[child_off[nd] + c];
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
        const int nd = st.run_node[b * Kb + j];
        top_lp[it] = bv; top_beam[it] = j;
        top_tok[it] = child_tok[child_off[nd] + c];
        top_node[it] = child_node[child_off[nd] + c];
        for (int i = 0; i < K2; ++i)        // mark taken (rows are short: <= K2 entries)
          if (cs[j * K2 + i] != P5_NEG_INF && row_top_c[(size_t)(b * Kb + j) * K2 + i] == c) { cs[j * K2 + i] = P5_NEG_INF; break; }
      } else {   // fewer than 2K allowed continuations: HF would pick arbitrary -inf entries
        top_lp[it] = P5_NEG_INF; top_beam[it] = 0; top_tok[it] = 0; top_node[it] = -1;
      }
    }
    __syncthreads();
  }
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
  const bool uns = st.unsat[b] != 0;
  if (tid < Kb) msc[tid] = st.fin_score[b * Kb + tid];
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
      if (tid < Kb) { fin_src[rank] = tid; fin_fl[rank] = st.fin_flag[b * Kb + tid]; fin_ln[rank] = st.fin_len[b * Kb + tid]; }
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
    if (fin_src[j] >= 0) v = st.fin_seq[((size_t)b * Kb + fin_src[j]) * max_len + p];
    else {
      const int i = -fin_src[j] - 1;
      v = (p == cur_len) ? top_tok[i] : st.run_seq[((size_t)b * Kb + top_beam[i]) * max_len + p];
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
        (p == cur_len) ? top_tok[i] : st.run_seq[((size_t)b * Kb + top_beam[i]) * max_len + p];
  }
  const int pos = cur_len - 1;   // K/V of this step were stored at `pos` by row (b*Kb + old beam)
  for (int t = tid; t < Kb * (pos + 1); t += 256) {
    const int j = t / (pos + 1), p = t % (pos + 1);
    const int parent_row = b * Kb + top_beam[sel_run[j]];
    st.anc_next[(size_t)p * R + b * Kb + j] = (p == pos) ? parent_row : st.anc[(size_t)p * R + parent_row];
  }
  if (tid < Kb) {
    const int i = sel_run[tid];
    st.run_node[b * Kb + tid] = top_node[i];
    st.last_tok[b * Kb + tid] = (int64_t)top_tok[i];
    st.run_score[b * Kb + tid] = run_sc[tid];
  }
  // the step counter advances once every workgroup of this launch is done with it (they all read it on entry): the last one
  // to arrive bumps it -- this used to be a launch of its own
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(&st.flags[3], 1) == (int)gridDim.x - 1) {
      st.flags[3] = 0;
      st.flags[2] = cur_len + 1;
    }
  }
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
  if (i < B) st.unsat[i] = 1;
  if (i == 0) { st.flags[0] = 0; st.flags[1] = 0; st.flags[2] = 1; st.flags[3] = 0; st.flags[4] = 0; }
}
