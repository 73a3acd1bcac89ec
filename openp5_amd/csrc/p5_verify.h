// p5_verify.h -- "bf16 drafts, fp32 decides": the generation mode whose ranked lists are the fp32 search's (round 5).
//
// The reference ranks items by the fp32 scores of HF beam search (DistributedRunner.py:361-387, utils/evaluate.py:37-58).  A bf16 search
// reproduces those lists only up to near-ties; an fp32 search does, at less than half the speed.  This file is the
// bridge, in the shape of speculative decoding:
//   1. DRAFT   -- the bf16 engine runs the ordinary device beam search with a WIDER beam (K' = K + extra) and records, per step and beam,
//                 (parent beam, token, trie node, live) -- `hist`, written by p5_beam_tail (p5_decode.h).
//   2. PLAN    -- p5_verify_plan_kernel turns the history into a forest of DISTINCT live prefixes per user (rows), keyed by
//                 (parent row, token) -- path identity, so it is also right for a grafted (DAG) trie -- with parent / depth / node and an
//                 ancestor table.  Everything the draft ever kept alive is a row; typically ~50 rows per user where the search touched
//                 K' x steps = 128 beam slots.
//   3. SCORE   -- ONE teacher-forced fp32 decoder pass over all rows at once (throughput GEMMs on [B x rows, d]; self-attention over
//                 each row's ancestors: p5_tree_attn_kernel; cross-attention per user against the fp32 encoder output), then the
//                 streaming tied head (log-sum-exp over the full vocabulary) and, per row, the log-probabilities of the row's trie
//                 children -- the kernels of the fp32 decode step (p5_decode2.h, p5_decode.h), once instead of once per step.
//   4. REPLAY  -- p5_verify_step_kernel re-runs HF's beam search with the REAL beam width K, step by step, on those fp32
//                 numbers: candidates of a beam = its row's children, bookkeeping = p5_beam_tail (the same code the search itself uses).
//                 The replay never looks at a bf16 score.  If it ever needs a live prefix the draft did not keep (it was dropped from a
//                 K'-wide bf16 beam although the fp32 search ranks it among its K), the user is FLAGGED and the host re-runs that user
//                 through the plain fp32 search -- so a returned list is always the fp32 search's list, never "close to" it.
// Dead (-1e9) beams need no model values: a candidate of a dead beam scores fl(-1e9 + log p) = -1e9 for every |log p| < 32 (the ulp of
// 1e9 is 64), and HF breaks those ties in (beam, token) order, i.e. child order here.
#pragma once
#include "p5_decode.h"

struct P5VerifyPlan {
  int* hdr;          // [0] max rows of any user, [1] beam steps the draft executed, [2] rows of all users, [3] overflow flag
  int* n_rows;       // [B]
  int* row_tok;      // [B][cap]  last token of the prefix (the decoder input at this position)
  int* row_parent;   // [B][cap]  row of the prefix without its last token (-1: the start prefix)
  int* row_depth;    // [B][cap]  position of the row = number of generated tokens in the prefix
  int* row_node;     // [B][cap]  trie node the prefix leads to (its children are the row's candidates)
  int* first;        // [B][max_len + 1]  rows of depth s are [first[s], first[s + 1])
  int* anc;          // [B][cap][max_len]  row of the ancestor at depth t (t <= depth)
  int cap, max_len;
};

// history record of the draft search: hist[4 + ((step * 4 + field) * R + row)], step = 1 .. max_len - 1; hist[0] = steps executed
#define P5_HIST_FIELDS 4
__device__ static __forceinline__ int p5_hist_at(const int* hist, int R, int step, int field, int row) {
  return hist[4 + ((size_t)step * P5_HIST_FIELDS + field) * R + row];
}

// ---- 2. PLAN: one workgroup per user ----
__global__ __launch_bounds__(256) void p5_verify_plan_kernel(P5VerifyPlan pl, const int* __restrict__ hist, const int* __restrict__ child_off,
                                                            const int* __restrict__ child_tok, const int* __restrict__ child_node,
                                                            const int* __restrict__ roots, int B, int Kw, int start_id) {
  __shared__ int prow[P5_MAX_K], crow[P5_MAX_K], kpar[P5_MAX_K], ktok[P5_MAX_K], knode[P5_MAX_K], kvalid[P5_MAX_K], isnew[P5_MAX_K], s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int R = B * Kw, cap = pl.cap, ML = pl.max_len;
  const int steps = hist[0];
  int* rtok = pl.row_tok + (size_t)b * cap;
  int* rpar = pl.row_parent + (size_t)b * cap;
  int* rdep = pl.row_depth + (size_t)b * cap;
  int* rnode = pl.row_node + (size_t)b * cap;
  int* first = pl.first + (size_t)b * (ML + 1);
  int* anc = pl.anc + (size_t)b * cap * ML;
  if (tid == 0) {
    int nd = -1;
    const int root = roots ? roots[b] : 0;
    for (int c = child_off[root]; c < child_off[root + 1]; ++c)
      if (child_tok[c] == start_id) nd = child_node[c];
    rtok[0] = start_id; rpar[0] = -1; rdep[0] = 0; rnode[0] = nd; anc[0] = 0;
    first[0] = 0; first[1] = 1;
    s_n = 1;
  }
  if (tid < Kw) prow[tid] = 0;
  __syncthreads();
  for (int s = 1; s <= steps && s < ML; ++s) {
    if (tid < Kw) {
      const int r = b * Kw + tid;
      const int par = p5_hist_at(hist, R, s, 0, r), tok = p5_hist_at(hist, R, s, 1, r), nd = p5_hist_at(hist, R, s, 2, r);
      const int live = p5_hist_at(hist, R, s, 3, r);
      const int pr = (par >= 0 && par < Kw) ? prow[par] : -1;
      kpar[tid] = pr; ktok[tid] = tok; knode[tid] = nd;
      kvalid[tid] = (live && nd >= 0 && pr >= 0) ? 1 : 0;
    }
    __syncthreads();
    if (tid < Kw) {
      int dup_of = -1;
      if (kvalid[tid])
        for (int j = 0; j < tid; ++j)
          if (kvalid[j] && kpar[j] == kpar[tid] && ktok[j] == ktok[tid]) { dup_of = j; break; }
      isnew[tid] = (kvalid[tid] && dup_of < 0) ? 1 : 0;
      crow[tid] = dup_of;          // (temporarily: the beam this one duplicates)
    }
    __syncthreads();
    const int n0 = s_n;
    int myrow = -1;
    bool mynew = false;
    if (tid < Kw) {
      int me = tid;
      if (kvalid[tid] && crow[tid] >= 0) me = crow[tid];
      int pos = 0;
      for (int j = 0; j < me; ++j) pos += isnew[j];
      const int row = n0 + pos;
      myrow = (kvalid[tid] && row < cap) ? row : -1;
      mynew = isnew[tid] && row < cap;
    }
    __syncthreads();            // every beam has read whom it duplicates before crow becomes the row table
    if (tid < Kw) {
      crow[tid] = myrow;
      if (mynew) {
        rtok[myrow] = ktok[tid]; rpar[myrow] = kpar[tid]; rdep[myrow] = s; rnode[myrow] = knode[tid];
        for (int t = 0; t < s; ++t) anc[(size_t)myrow * ML + t] = anc[(size_t)kpar[tid] * ML + t];
        anc[(size_t)myrow * ML + s] = myrow;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int j = 0; j < Kw; ++j) add += isnew[j];
      int n1 = n0 + add;
      if (n1 > cap) { n1 = cap; pl.hdr[3] = 1; }
      s_n = n1;
      first[s + 1] = n1;
    }
    if (tid < Kw) prow[tid] = crow[tid];
    __syncthreads();
  }
  if (tid == 0) {
    const int n = s_n;
    for (int s = (steps < ML - 1 ? steps : ML - 1) + 1; s < ML; ++s) first[s + 1] = n;
    pl.n_rows[b] = n;
    atomicMax(&pl.hdr[0], n);
    atomicAdd(&pl.hdr[2], n);
    if (b == 0) pl.hdr[1] = steps;
  }
}

// rows of the pass, padded to PU per user: decoder input ids, trie node (-1 = padding row: no candidates), depth
__global__ __launch_bounds__(256) void p5_verify_rows_kernel(int64_t* __restrict__ ids, int* __restrict__ node_flat, int* __restrict__ depth_flat,
                                                            P5VerifyPlan pl, int B, int PU, int pad_id) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * PU) return;
  const int b = i / PU, r = i % PU;
  const bool ok = r < pl.n_rows[b] && r < pl.cap;
  ids[i] = ok ? (int64_t)pl.row_tok[(size_t)b * pl.cap + r] : (int64_t)pad_id;
  node_flat[i] = ok ? pl.row_node[(size_t)b * pl.cap + r] : -1;
  depth_flat[i] = ok ? pl.row_depth[(size_t)b * pl.cap + r] : 0;
}

// ---- 3. self-attention of a row over its ancestors (the causal decoder self-attention of HF modeling_t5.py:217-279 restricted to the
// row's own prefix; unidirectional relative-position bias by depth difference).  One wave per (row, head); lane = (key slot lane / 8,
// dims (lane % 8) * 8 ..): eight ancestors per pass, 16-byte loads. ----
template <class T>
__global__ __launch_bounds__(256) void p5_tree_attn_kernel(T* __restrict__ out, const T* __restrict__ qkv, P5VerifyPlan pl,
                                                          const int* __restrict__ depth_flat, const float* __restrict__ rel_table,
                                                          const int* __restrict__ lut, int lut_half, int B, int PU, int H) {
  const int lane = threadIdx.x & 63;
  const int rh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rh >= B * PU * H) return;
  const int r = rh / H, h = rh % H;
  const int b = r / PU, ri = r % PU;
  const int inner = H * 64;
  const int ts = lane >> 3, dc = lane & 7;
  const int depth = depth_flat[r];
  const int* __restrict__ anc = pl.anc + ((size_t)b * pl.cap + (ri < pl.cap ? ri : 0)) * pl.max_len;
  auto ld8 = [](const T* p, float* o) {
    if constexpr (sizeof(T) == 2) unpack16<T>(ld16(p), o);
    else { unpack16<T>(ld16(p), o); unpack16<T>(ld16(p + 4), o + 4); }
  };
  auto st8 = [](T* p, const float* o) {
    if constexpr (sizeof(T) == 2) st16(p, pack16<T>(o));
    else { st16(p, pack16<T>(o)); st16(p + 4, pack16<T>(o + 4)); }
  };
  float q[8];
  ld8(qkv + (size_t)r * 3 * inner + h * 64 + dc * 8, q);
  float m = P5_NEG_INF, l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int t0 = 0; t0 <= depth; t0 += 8) {
    const int t = t0 + ts;
    float kk[8], vv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { kk[e] = 0.f; vv[e] = 0.f; }
    float s = P5_NEG_INF;
    if (t <= depth) {
      const int ar = (t == depth) ? r : b * PU + anc[t];
      const T* base = qkv + (size_t)ar * 3 * inner + inner + h * 64 + dc * 8;
      ld8(base, kk);
      ld8(base + inner, vv);
    }
    float d8 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d8 += q[e] * kk[e];
    d8 += __shfl_xor(d8, 1); d8 += __shfl_xor(d8, 2); d8 += __shfl_xor(d8, 4);
    if (t <= depth) s = d8 + rel_table[lut[(t - depth) + lut_half] * H + h];
    float cm = s;
    cm = fmaxf(cm, __shfl_xor(cm, 8)); cm = fmaxf(cm, __shfl_xor(cm, 16)); cm = fmaxf(cm, __shfl_xor(cm, 32));
    const float mn = fmaxf(m, cm);
    const float sc = (m == P5_NEG_INF) ? 0.f : expf(m - mn);
    const float p = (s == P5_NEG_INF) ? 0.f : expf(s - mn);
    float ps = p;
    ps += __shfl_xor(ps, 8); ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
    l = l * sc + ps;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float pv = p * vv[e];
      pv += __shfl_xor(pv, 8); pv += __shfl_xor(pv, 16); pv += __shfl_xor(pv, 32);
      o[e] = o[e] * sc + pv;
    }
    m = mn;
  }
  if (ts == 0) {
    const float inv = 1.f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    st8(out + (size_t)r * inner + h * 64 + dc * 8, o);
  }
}

// ---- 4. REPLAY: one beam-search step of the real width Kb on the fp32 row scores; one workgroup per user ----
// row_top_*: the per-ROW candidate lists written by p5_dec_score2_kernel / p5_dec_score_kernel with a zero running score: entry i of row
// r is (log p of the row's i-th best child, child index), sorted (score desc, child asc).  vrow_*: verification row of every running
// beam (double-buffered by the parity of cur_len like the other beam state), -1 = none (dead beam, or a live prefix the draft never kept).
// forced steps of the replay (p5_decode.h, forced-prefix fast-forward): rows 0 .. F-1 are the chain, each with exactly one candidate -- its
// log-probability is what the step adds to beam 0; every beam then stands on row F
__global__ __launch_bounds__(256) void p5_verify_forced_kernel(float* __restrict__ nll, int* __restrict__ vrow_a, int* __restrict__ vrow_b,
                                                              const float* __restrict__ row_top_score, int PU, int K2, int B, int Kb, int F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * F) nll[i] = -row_top_score[((size_t)(i / F) * PU + (i % F)) * K2];
  const int j = i - B * F;
  if (j >= 0 && j < B * Kb) { vrow_a[j] = F; vrow_b[j] = F; }
}

// Range guard of the split-product passes (p5_gemm.h::p5_gemm_split_kernel): an fp32 operand x is multiplied as fp16 hi + lo / 4096, which
// covers |x| < 2^15 (beyond it lo = fp16((x - hi) * 4096) overflows; beyond 65504 hi does).  An operand out of range turns into inf / NaN
// and reaches the final normalised hidden rows through the residual stream, so ONE look at those rows per user decides: a user with a
// non-finite or out-of-range value there is flagged like a user with a missing prefix and the caller re-runs that user on the exact-fp32
// search (p5hip.h, p5_verify_run).  T5's ReLU / gated-GELU hidden units are where such magnitudes are known to occur (bf16 / fp16 T5).
template <class T>
__global__ __launch_bounds__(256) void p5_verify_range_kernel(int* __restrict__ missing, const T* __restrict__ hn, int PU, int d) {
  const int b = blockIdx.x;
  const T* __restrict__ p = hn + (size_t)b * PU * d;
  bool bad = false;
  for (int i = threadIdx.x; i < PU * d; i += 256) bad |= !(fabsf(to_f<T>(p[i])) < 32768.f);
  if (bad) missing[b] = 1;        // (every writer stores the same value)
}

#define P5_VERIFY_POOL 1024          // Kb x 2 Kb candidates of the replay: real beam widths up to 22 (the DRAFT may be as wide as P5_MAX_K)
__global__ __launch_bounds__(256) void p5_verify_step_kernel(P5BeamState st, P5VerifyPlan pl, const float* __restrict__ row_top_score,
                                                            const int* __restrict__ row_top_c, const int* __restrict__ row_n_top, int PU,
                                                            const int* __restrict__ child_off, const int* __restrict__ child_tok,
                                                            const int* __restrict__ child_node, const uint32_t* __restrict__ excluded,
                                                            int excl_words, int max_c, int Kb, int max_len, int eos_id, int R,
                                                            int* __restrict__ vrow_odd, int* __restrict__ vrow_even, int* __restrict__ missing) {
  if (st.flags[4]) return;
  const int cur_len = st.flags[2];
  if ((cur_len & 1) == 0) {
    int* t;
    t = st.run_seq; st.run_seq = st.run_seq_next; st.run_seq_next = t;
    t = st.fin_seq; st.fin_seq = st.fin_seq_next; st.fin_seq_next = t;
    t = st.anc; st.anc = st.anc_next; st.anc_next = t;
  }
  const int* __restrict__ vcur = (cur_len & 1) ? vrow_odd : vrow_even;
  int* __restrict__ vnext = (cur_len & 1) ? vrow_even : vrow_odd;
  __shared__ float cs[P5_VERIFY_POOL];
  __shared__ int ckey[P5_VERIFY_POOL];
  __shared__ __attribute__((aligned(16))) unsigned long long k64[P5_VERIFY_POOL + 2];
  __shared__ P5BeamSh sh;
  __shared__ float s_rs[P5_MAX_K];
  __shared__ int s_vr[P5_MAX_K], s_miss;
  float* top_lp = sh.top_lp;
  int* top_beam = sh.top_beam; int* top_tok = sh.top_tok; int* top_node = sh.top_node;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int K2 = 2 * Kb;
  p5_beam_prefetch(st, sh, b, tid, Kb, max_len, R, cur_len, child_off);
  if (tid < Kb) { s_rs[tid] = st.run_score[b * Kb + tid]; s_vr[tid] = vcur[b * Kb + tid]; }
  if (tid == 0) { sh.s_nothit = 0; s_miss = 0; }
  __syncthreads();
  const uint32_t* ex = excluded ? excluded + (size_t)b * excl_words : nullptr;
  for (int t = tid; t < Kb * K2; t += 256) {
    const int j = t / K2, i = t % K2;
    const int nd = sh.old_node[j];
    const float rs = s_rs[j];
    float v = P5_NEG_INF;
    int key = 0x7fffffff;
    if (nd >= 0) {
      if (rs > -1.0e8f) {
        const int vr = s_vr[j];
        if (vr >= 0) {
          const size_t rr = (size_t)b * PU + vr;
          if (i < row_n_top[rr]) { v = row_top_score[rr * K2 + i] + rs; key = j * max_c + row_top_c[rr * K2 + i]; }
        } else if (i == 0) {
          s_miss = 1;               // a live beam without verified scores: this user's replay cannot be trusted from here on
        }
      } else {
        // dead beam: every child scores fl(rs + log p) = rs; HF's tie order is the child order -> the i-th allowed child
        const int c0 = sh.old_coff[j];
        int nc = child_off[nd + 1] - c0;
        nc = nc < max_c ? nc : max_c;
        int seen = 0, c = 0;
        for (; c < nc; ++c) {
          bool ok = true;
          if (ex) { const int cn = child_node[c0 + c]; ok = !((ex[cn >> 5] >> (cn & 31)) & 1u); }
          if (ok) { if (seen == i) break; ++seen; }
        }
        if (c < nc) { v = rs; key = j * max_c + c; }
      }
    }
    cs[t] = v; ckey[t] = key; k64[t] = p5_rank_key(v, key);
  }
  if (tid < 2) k64[Kb * K2 + tid] = 0ull;
  if (tid < K2) { top_lp[tid] = P5_NEG_INF; top_beam[tid] = 0; top_tok[tid] = 0; top_node[tid] = -1; }
  __syncthreads();
  // every candidate computes its own rank in the (score desc, beam * max_c + child asc) order (as p5_beam_step_kernel); the 2K best land at
  // their rank
  for (int t = tid; t < Kb * K2; t += 256) {
    const float v = cs[t];
    if (v == P5_NEG_INF) continue;
    const int key = ckey[t];
    const int rank = p5_rank_of(k64, Kb * K2, k64[t]);
    if (rank < K2) {
      const int j = t / K2, c = key - j * max_c;
      top_lp[rank] = v; top_beam[rank] = j;
      top_tok[rank] = child_tok[sh.old_coff[j] + c];
      top_node[rank] = child_node[sh.old_coff[j] + c];
    }
  }
  __syncthreads();
  // rows of the NEW running beams (the tail below overwrites nothing this needs, but ends with the grid-wide step counter)
  const int was_unsat = sh.old_unsat;
  p5_beam_tail(st, sh, b, tid, Kb, K2, max_len, eos_id, R, cur_len);
  if (tid < Kb) {
    const int i = sh.sel_run[tid];
    const int parent = top_beam[i], tok = top_tok[i], nd = top_node[i];
    int row = -1;
    if (sh.run_sc[tid] > -1.0e8f && nd >= 0) {
      const int pr = s_vr[parent];
      if (pr >= 0 && cur_len < pl.max_len) {
        const int* first = pl.first + (size_t)b * (pl.max_len + 1);
        const int* rpar = pl.row_parent + (size_t)b * pl.cap;
        const int* rtok = pl.row_tok + (size_t)b * pl.cap;
        for (int r = first[cur_len]; r < first[cur_len + 1]; ++r)
          if (rpar[r] == pr && rtok[r] == tok) { row = r; break; }
      }
      // (a missing row only matters if this beam is expanded again: the next step raises the flag when it reads row == -1)
    }
    vnext[b * Kb + tid] = row;
  }
  if (tid == 0 && s_miss && was_unsat) missing[b] = 1;
}
