// p5_gemm.h -- LDS-tiled MFMA GEMM for the bias-free Linear layers of T5 (forward, dgrad, wgrad).
//
// Replaces the cuBLAS calls torch issues for nn.Linear in HF T5Attention / T5LayerFF
// (HF modeling_t5.py:205-208,304,325-326,367 and :83-94) and their autograd transposes.
//
//   C[M,N] (+)= sum_k A(m,k) * B(n,k)
//
// Each operand is addressed either "KC" (reduction dim contiguous: element (r,k) at p[r*ld + k]) or
// "KS" (reduction dim strided: element (r,k) at p[k*ld + r]).  That covers, without ever materialising a
// transpose in HBM:
//   forward  y  = x W^T      : A = x  KC,  B = W  KC   ([out,in] row-major nn.Linear weight)
//   dgrad    dx = dy W       : A = dy KC,  B = W  KS
//   wgrad    dW = dy^T x     : A = dy KS,  B = x  KS   (split-K over the token dim, fp32 atomics)
// KS fragments come out of LDS through ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32).
//
// Tile: BM x BN per 256-thread workgroup (4 waves as 2x2), NCK 64-byte K-chunks per step, double-buffered in
// LDS with the next step's global loads in flight during the MFMAs.  bf16 results leave through an LDS-staged
// epilogue so that residual reads and C writes are 16 bytes per lane.
#pragma once
#include "p5_device.h"
#include "p5_rng.h"

enum P5Epi : int {
  P5_EPI_STORE = 0,       // C = acc * alpha
  P5_EPI_RELU_DROP = 1,   // C = drop(relu(acc))
  P5_EPI_RESID_DROP = 2,  // C = aux + drop(acc)            (aux: residual stream, same dtype as C)
  P5_EPI_MASK_POS = 3,    // C = aux > 0 ? acc * alpha : 0  (relu/dropout backward through saved hidden)
                          // with ssq_out (p5_gemm5.h only): ssq_out[row][64-column group] = sum of stored C * aux / alpha = the row's share of
                          // <d pre, pre> -- what the T5LayerNorm backward of the sub-layer's input needs (P5_EPI_NORM_BWD below)
  P5_EPI_ATOMIC = 4,      // C += acc * alpha  (fp32 atomics; split-K wgrad)
  P5_EPI_GELU_GATE = 5,   // gated-GELU forward (T5 v1.1 FFN, HF modeling_t5.py:97-123) fused into the wi GEMM: B = [wi_0; wi_1] read gate-interleaved
                          // (P5GemmArgs::gate_F), C = h = drop(gelu_new(u0) * u1) [M, F], C2 = u = [u0 | u1] [M, 2F] kept for the backward
                          // (p5_gemm5.h whole-tile epilogue only: bf16, M % 256 == 0, N = 2F, N % 128 == 0)
  P5_EPI_ACCUM = 6,       // C += acc * alpha  (fp32, exclusive ownership: no split-K)
  P5_EPI_CE_STATS = 8,    // logit-free cross-entropy, forward (P5_T5.py:361-369): nothing is stored but, per row and 64-column group,
                          // (max, sum of exp) of acc * alpha, and the logit at the row's label -> P5GemmArgs::ce_part / ce_lab (p5_gemm5.h)
  P5_EPI_CE_GRAD = 9,     // ... backward: the same GEMM recomputed, C = dlogits = (exp(acc * alpha - lse[row]) - [col == label]) * g[row]
  P5_EPI_NORM_BWD = 10,   // T5LayerNorm backward (HF modeling_t5.py:59-72 under autograd) in the epilogue of the data-gradient GEMM that produces its
                          // input gradient (round 6; p5_gemm5.h, 128-row tiles, whole tiles only): acc = dn = dOut W (unfolded W^T copy).  With x = aux
                          // (the sub-layer's input rows), rstd from `rowss`, w = nb_w (the norm weight) and m = sum(nb_dot[row][:]) / N:
                          //   xh = x rstd;  v = rstd (acc w - xh m) + nb_rin;  nb_rout = v (fp32);  C = dropout_next(v);  C2 = w * round(xh) (the forward
                          //   norm's output, which the folded forward never wrote);  nb_dw[row / 64][col] = sum over the 64 rows of acc xh (norm-weight
                          //   gradient partials).  m needs no pass over the row: sum_j (acc w)_j xh_j = <dOut, Out> over the projection's OUTPUT columns,
                          //   which the producer of dOut leaves as partial sums (MASK_POS + ssq_out above; P5AttnArgs::dot_out)
  P5_EPI_GELU_GATE_BWD = 7,   // gated-GELU backward fused into the wo data-gradient GEMM: acc = dh [M, F], aux = u [M, 2F] (ldaux), C = du =
                              // [dh u1 gelu'(u0) | dh gelu(u0)] [M, 2F] (ldc); the dropout mask of h is re-hashed (same conditions as above, N = F)
};

struct P5GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const void* aux;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int a_ks, b_ks;   // 0 = KC, 1 = KS
  int epi;
  int c_f32;        // 1: C is fp32 regardless of T
  int splitk;       // >=1
  int ring;         // launcher-internal: use the multi-stage ring kernel
  int xcd_bm, xcd_bn;   // launcher-internal: tiles per XCD rectangle (0 = contiguous runs)
  // RMSNorm folded into the GEMMs of the decode step: the norm weight lives in the B operand, the row
  // statistic arrives / leaves as a sum of squares:
  const float* rowss;   // [M] sum_j x[row,j]^2 of the A rows, or nullptr: acc *= rsqrt(rowss[row] * rowss_invd + rowss_eps)
  float rowss_invd, rowss_eps;
  float* ssq_out;       // [M] or nullptr: += sum of squares of the stored C row (as rounded to the C dtype)
  float alpha;
  P5Drop drop;
  int g4_tiles_n, g4_nk;   // p5_gemm4.h launcher-internal: tiles along N, K-steps (of 64) per work unit
  int g4_cb;               // p5_gemm5.h launcher-internal: > 0 = the 32 workgroups of an XCD work on a (32 / g4_cb) x g4_cb block of tiles at a time
  // T5LayerNorm folded into the TRAINING GEMMs (bf16 engine): the row statistic travels as `nt` partial sums of squares per row, one
  // per 64-column group of the residual stream, each written by exactly one wave with a plain store (no atomics, no clearing, same
  // bits every run) and summed in a fixed order by the consumer.  0 = the decode step's scalar-per-row form (atomic accumulate).
  int rowss_nt, ssq_nt;
  // Deterministic split-K (fp32 C, epilogue kind P5_EPI_ATOMIC): when > 0, split z STORES its partial product at C + z * stride
  // (elements) instead of adding to C with atomics; the caller sums the active splits in index order (p5_reduce_splits_kernel).
  long long c_split_stride;
  int mm_split;          // fp32 operands, K-contiguous: products on the f16 matrix cores from a two-term split of each operand (below); 0 = exact fp32 MFMAs
  // gated-GELU epilogues (P5_EPI_GELU_GATE / _BWD): second output of the forward, and F = d_ff when the rows of B are read gate-interleaved
  // -- GEMM column n = 64 t + j is row 32 t + j of wi_0 for j < 32 and row F + 32 t + (j - 32) (= wi_1) otherwise, so that a lane's two
  // 8-column groups of a 64-column wave tile are u0 and u1 of the SAME eight hidden units
  void* C2;
  int ldc2, gate_F;
  // logit-free cross-entropy (P5_EPI_CE_STATS / _GRAD): labels [M]; forward outputs ce_part [M, 2 * ce_np] ((max, sum exp) per 64-column group,
  // ce_np = ceil(N / 64)) and ce_lab [M] (logit at the label); backward inputs ce_lse [M], ce_g [M] (gradient of the row's NLL)
  const int64_t* ce_labels;
  float* ce_part; float* ce_lab;
  const float* ce_lse; const float* ce_g;
  int ce_np;
  // T5LayerNorm backward epilogue (P5_EPI_NORM_BWD): nb_dot [M, nb_dot_nt] partial sums of <dOut, Out> per row, nb_rin / nb_rout the fp32
  // residual-stream gradient in / out [M, N] (leading dimension N), nb_w [N] the norm weight, nb_dw [M / 64, N] its gradient's partial rows
  const float* nb_dot; int nb_dot_nt;
  const float* nb_rin; float* nb_rout;
  const float* nb_w; float* nb_dw;
};

// XCD-aware tile order (MI355X: workgroup b runs on XCD b % 8, each XCD has a private 4 MiB L2).  Default: every XCD gets
// a contiguous run of tiles, n fastest inside an m row, so the A panel of a row and the B panels are fetched from HBM once
// per XCD and then hit in its L2 instead of every workgroup streaming its own 2 x (tile x K) bytes.  When the launcher
// finds an exact cover of the tile grid by 8 rectangles of xcd_bm x xcd_bn tiles it passes that shape instead: an XCD then
// fetches xcd_bm A panels + xcd_bn B panels, least for the squarest rectangle (a 4 x 16 tile grid of a weight gradient:
// 4 x 2 blocks move 6 panels per XCD and K-split, 1 x 8 runs move 9).
template <int BM, int BN>
__device__ static __forceinline__ void gemm_tile_origin(const P5GemmArgs& g, int& m0, int& n0) {
  const int gx = gridDim.x, nb = gridDim.x * gridDim.y;
  const int bid = blockIdx.x + blockIdx.y * gx;
  const int xcd = bid & 7, idx = bid >> 3;
  if (g.xcd_bn > 0) {
    const int bpr = gx / g.xcd_bn;                 // rectangles per row of rectangles
    m0 = ((xcd / bpr) * g.xcd_bm + idx / g.xcd_bn) * BM;
    n0 = ((xcd % bpr) * g.xcd_bn + idx % g.xcd_bn) * BN;
    return;
  }
  const int q = nb >> 3, r = nb & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  m0 = (lid / gx) * BM;
  n0 = (lid % gx) * BN;
}

// byte size of ONE 64-byte-K chunk of an operand tile of R rows
template <class T, int R, bool KS> struct LdsChunk {
  static constexpr int KS_STRIDE = R * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
  static constexpr int BYTES = KS ? TT<T>::KCH * KS_STRIDE : R * 64;
};

__device__ static __forceinline__ int kc_off(int row, int kc) {
  // XOR swizzle that makes every ds_read_b128 lane group hit 16 distinct 16-byte slots
  const int h = (0x1230 >> (((row >> 2) & 3) * 4)) & 3;  // h = [0,3,2,1]
  return row * 64 + ((kc ^ h) << 4);
}

// loads ONE chunk (KCH elements of K starting at k0) of an R-row operand tile into registers
template <class T, int R, bool KS>
__device__ static __forceinline__ void stage_load(u32x4* regs, const T* __restrict__ p, int ld, int r0, int k0, int nrows,
                                                  int K, int tid) {
  constexpr int EPF = TT<T>::EPF;
  if constexpr (!KS) {
    constexpr int NCH = R * 4 / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 2, kc = c & 3;
      const int gr = r0 + row, gk = k0 + kc * EPF;
      regs[i] = (gr < nrows && gk < K) ? ld16(p + (size_t)gr * ld + gk) : zero16();
    }
  } else {
    constexpr int CPR = R / EPF;  // chunks per k-row
    constexpr int NCH = TT<T>::KCH * CPR / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      const int krow = c / CPR, mc = c % CPR;
      const int gk = k0 + krow, gr = r0 + mc * EPF;
      regs[i] = (gk < K && gr < nrows) ? ld16(p + (size_t)gk * ld + gr) : zero16();
    }
  }
}

template <class T, int R, bool KS>
__device__ static __forceinline__ void stage_store(const u32x4* regs, char* lds, int tid) {
  constexpr int EPF = TT<T>::EPF;
  if constexpr (!KS) {
    constexpr int NCH = R * 4 / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      st16(lds + kc_off(c >> 2, c & 3), regs[i]);
    }
  } else {
    constexpr int CPR = R / EPF;
    constexpr int NCH = TT<T>::KCH * CPR / 256;
    constexpr int STRIDE = LdsChunk<T, R, true>::KS_STRIDE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      st16(lds + (c / CPR) * STRIDE + (c % CPR) * 16, regs[i]);
    }
  }
}

// K-contiguous operand straight from HBM into LDS (no VGPR round trip, no ds_write).  One K-step = 128 bytes per row
// (two 64-byte chunks), so every wave instruction copies 8 rows x 128 B = whole cache lines; with 64-byte row
// segments each 128-byte line would be requested twice through the per-CU vector-memory path, which is the limiter of
// these GEMMs.  LDS image = [R][128 B]; the 16-byte slot index is XOR-ed with (row & 7) -- applied on the
// SOURCE address, since a wave instruction's LDS image is linear -- which makes every ds_read_b128 lane group
// conflict-free.  Rows past the matrix edge are clamped (they only feed C rows/cols that are never stored).
// Needs K % (2*KCH) == 0.
template <class T, int R>
__device__ static __forceinline__ void stage_dma128(char* lds, const T* __restrict__ p, int ld, int r0, int k0, int nrows, int tid) {
  constexpr int NI = R / 32;                       // wave instructions per wave: R*128 B / (4 waves * 1 KiB)
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rbase = (wave * NI + i) * 8;
    const int row = rbase + (lane >> 3), slot = lane & 7;
    int gr = r0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    glds16(p + (size_t)gr * ld + k0 + ((slot ^ (row & 7)) * TT<T>::EPF), lds + rbase * 128);
  }
}
template <class T>
__device__ static __forceinline__ u32x4 frag_load_kc128(const char* lds, int t0, int c, int lane) {
  const int row = t0 + (lane & 15);
  return ld16(lds + row * 128 + ((((c << 2) | (lane >> 4)) ^ (row & 7)) << 4));
}

// K-STRIDED operand (element (r,k) at p[k*ld + r]: dgrad weights, both wgrad operands) straight from HBM into LDS.
// LDS image of one K-step = [64 k-rows][R elements], rows back to back (a wave instruction writes 1 KiB linearly, so
// there is no room for row padding).  ds_read_b64_tr_b16 serves 32 lanes per LDS cycle and those touch 8 different
// k-rows at the same column range, i.e. the same banks; the 16-byte chunk index inside a row is therefore XOR-ed with
// 2*s(row) -- s picks a different 32-byte bank slot for each of the 8 rows -- again on the SOURCE address.
template <int R> __device__ static __forceinline__ int ksd_swz(int krow) {
  if constexpr (R >= 128) return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1;
  else return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1;
}
// Needs K % 64 == 0.  Chunks past the matrix edge feed C rows/cols that are never stored; they are clamped to the last whole
// chunk of the k-row in MEMORY (ld, a multiple of 8), not of the matrix: a ragged row count (vocabulary 32100) still gets
// its last partial chunk, and whatever lies between nrows and ld is readable padding.
template <class T, int R>
__device__ static __forceinline__ void stage_dma_ks(char* lds, const T* __restrict__ p, int ld, int r0, int k0, int nrows, int tid) {
  (void)nrows;
  static_assert(sizeof(T) == 2, "direct-to-LDS staging is a bf16 path");
  constexpr int RB = R * 2, CPR = RB / 16, RPI = 1024 / RB, NI = 64 / RPI / 4;   // NI wave instructions per wave (4 waves)
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave * NI + i;
    const int krow = q * RPI + lane / CPR;
    int cg = (lane % CPR) ^ ksd_swz<R>(krow);
    const int cmax = (ld - r0) / 8 - 1;
    cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
    glds16(p + (size_t)(k0 + krow) * ld + r0 + cg * 8, lds + q * 1024);
  }
}
template <class T, int R>
__device__ static __forceinline__ u32x4 frag_load_ksd(const char* lds, int t0, int c, int lane) {
  constexpr int RB = R * 2;
  const int g = lane >> 4, i = lane & 15;
  const int row = c * 32 + g * 8 + (i >> 2);                 // rows row and row + 4 share ksd_swz (bits 0,1,3 only)
  const int cg = (t0 >> 3) + ((i & 3) >> 1);
  const char* base = lds + row * RB + ((cg ^ ksd_swz<R>(row)) << 4) + ((i & 1) << 3);
  const u32x2 lo = lds_tr16_b64(base);
  const u32x2 hi = lds_tr16_b64(base + 4 * RB);
  u32x4 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
  return r;
}

// byte offset of this lane's transpose reads for fragment t0 inside a K-chunk of the image above (K-chunk c adds c*32 rows,
// the second read 4 rows: both immediates) -- loop-invariant, so a software-pipelined loop keeps it in a register
template <int R> __device__ static __forceinline__ int ksd_lane_off(int t0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int row = g * 8 + (i >> 2);
  const int cg = (t0 >> 3) + ((i & 3) >> 1);
  return row * (R * 2) + ((cg ^ ksd_swz<R>(row)) << 4) + ((i & 1) << 3);
}

// one 16-row fragment (rows t0..t0+15 of the tile) for this lane
template <class T, int R, bool KS>
__device__ static __forceinline__ u32x4 frag_load(const char* lds, int t0, int lane) {
  if constexpr (!KS) {
    return ld16(lds + kc_off(t0 + (lane & 15), lane >> 4));
  } else {
    constexpr int STRIDE = LdsChunk<T, R, true>::KS_STRIDE;
    const int g = lane >> 4, i = lane & 15;
    u32x4 r;
    if constexpr (sizeof(T) == 2) {
#ifndef P5_NO_TR
      const char* base = lds + (g * 8 + (i >> 2)) * STRIDE + (t0 + (i & 3) * 4) * 2;
      u32x2 lo = lds_tr16_b64(base);
      u32x2 hi = lds_tr16_b64(base + 4 * STRIDE);
      r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
#else
      const unsigned short* b = (const unsigned short*)(lds + (g * 8) * STRIDE + (t0 + i) * 2);
      unsigned short v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = b[j * (STRIDE / 2)];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
#endif
    } else {
      const unsigned* b = (const unsigned*)(lds + (g * 4) * STRIDE + (t0 + i) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = b[j * (STRIDE / 4)];
    }
    return r;
  }
}

// acc scale of an output row whose A row is an un-normalised residual-stream row: rsqrt(mean(x^2) + eps) from the carried statistic
__device__ static __forceinline__ float gemm_row_rstd(const P5GemmArgs& g, int row) {
  float ss;
  if (g.rowss_nt > 0) {
    const float* p = g.rowss + (size_t)row * g.rowss_nt;
    if ((g.rowss_nt & 3) == 0) {          // d_model a multiple of 256: 16-byte loads, summed in index order like the scalar loop
      ss = 0.f;
      for (int t = 0; t < g.rowss_nt; t += 4) {
        const f32x4 v = *(const f32x4*)(p + t);
        ss = (((ss + v[0]) + v[1]) + v[2]) + v[3];
      }
    } else {
      ss = 0.f;
      for (int t = 0; t < g.rowss_nt; ++t) ss += p[t];
    }
  } else {
    ss = g.rowss[row];
  }
  return rsqrtf(ss * g.rowss_invd + g.rowss_eps);
}
__device__ static __forceinline__ float gemm_epi_apply(const P5GemmArgs& g, float v, float auxv, uint32_t seed, bool do_drop, int row,
                                                       int col) {
  if (g.epi == P5_EPI_RELU_DROP) {
    v = v > 0.f ? v : 0.f;
    if (do_drop) v = p5_keep(seed, g.drop.site_key, (uint32_t)(row * g.N + col), g.drop.thr) ? v * g.drop.scale : 0.f;
  } else if (g.epi == P5_EPI_RESID_DROP) {
    if (do_drop) v = p5_keep(seed, g.drop.site_key, (uint32_t)(row * g.N + col), g.drop.thr) ? v * g.drop.scale : 0.f;
    v += auxv;
  } else if (g.epi == P5_EPI_MASK_POS) {
    v = auxv > 0.f ? v : 0.f;
  }
  return v;
}

template <int V> struct P5EpiTag { static constexpr int value = V; };

template <class T, int BM, int BN, int LDSB, int NT = 256, int WNW = 2>
__device__ static __forceinline__ void gemm_epilogue(const P5GemmArgs& g, f32x4 (&acc)[BM / (16 * (NT / 64 / WNW))][BN / (16 * WNW)], char* lds,
                                                     int m0, int n0, int tid) {
  constexpr int WMW = NT / 64 / WNW;                  // waves along M x waves along N; wave tile = (BM/WMW) x (BN/WNW)
  constexpr int TM = BM / (16 * WMW), TN = BN / (16 * WNW);
  constexpr int CST = BN * 2 + 16;                    // LDS row stride of the staged bf16 C tile
  static_assert(LDSB >= BM * CST, "LDS buffer too small for the staged C tile");
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const uint32_t seed = p5_seed(g.drop);
  const bool do_drop = g.drop.state != nullptr && g.drop.thr != 0;

  if constexpr (sizeof(T) == 2) {
    if (!g.c_f32 && (g.N % 8) == 0 && (g.ldc % 8) == 0 && ((uintptr_t)g.C % 16) == 0 &&
        (g.aux == nullptr || ((g.ldaux % 8) == 0 && ((uintptr_t)g.aux % 16) == 0))) {
      // ---- LDS-staged epilogue: accumulators -> bf16 tile in LDS -> 16-byte rows (aux reads + C writes) ----
      // (the K loop ended with a barrier, so the staging buffers are free)
      constexpr int PPR = BN / 8;                 // 16-byte pieces per tile row
      constexpr int NPIECE = BM * PPR / NT;
      // the aux pieces (residual / saved hidden) are requested BEFORE the accumulators are staged: inside the store loop below
      // hipcc cannot move a load above the preceding C store (possible alias), which serialises NPIECE HBM round trips
      constexpr bool PRE = NPIECE <= 8;
      u32x4 auxr[PRE ? NPIECE : 1];
      if constexpr (PRE) {
        if (g.aux && g.epi != P5_EPI_STORE) {
#pragma unroll
          for (int i = 0; i < NPIECE; ++i) {
            const int p = tid + i * NT;
            const int row = m0 + p / PPR, col = n0 + (p % PPR) * 8;
            auxr[i] = (row < g.M && col < g.N) ? ld16((const bf16*)g.aux + (size_t)row * g.ldaux + col) : zero16();
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = wm * (BM / WMW) + i * 16 + (lane >> 4) * 4 + r;
          float sc = g.alpha;
          if (g.rowss) sc *= gemm_row_rstd(g, (m0 + lr) < g.M ? (m0 + lr) : (g.M - 1));
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int lc = wn * (BN / WNW) + j * 16 + (lane & 15);
            *(bf16*)(lds + lr * CST + lc * 2) = from_f<bf16>(acc[i][j][r] * sc);
          }
        }
      __syncthreads();
      // The epilogue kind is a compile-time constant of the store loop (EK: 0 store, 1 ReLU, 2 ReLU + dropout, 3 + residual,
      // 4 dropout + residual, 5 ReLU' mask), chosen once per tile: a per-element `if (g.epi == ...)` chain compiles to scalar
      // branches -- the dropout hash keeps hipcc from if-converting it -- about five per element (p5_gemm5.h, round 3), and the
      // descriptor fields it uses are read once here instead of where they are used.
      const int eM = g.M, eN = g.N, eldc = g.ldc, eldaux = g.ldaux, essq_nt = g.ssq_nt, eepi = g.epi;
      float* const essq = g.ssq_out;
      const bf16* const eaux = (const bf16*)g.aux;
      bf16* const eC = (bf16*)g.C;
      const uint32_t ethr = g.drop.thr;
      const float escale = g.drop.scale;
      const uint32_t hseed = p5_mix32(seed + g.drop.site_key);
      auto pieces = [&](auto ek) {
        constexpr int EK = decltype(ek)::value;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
          const int p = tid + i * NT;
          const int lr = p / PPR, pc = p % PPR;
          const int row = m0 + lr, col = n0 + pc * 8;
          const bool ok = row < eM && col < eN;
          float ss = 0.f;
          if (ok) {
            float v[8];
            unpack16<bf16>(ld16(lds + lr * CST + pc * 16), v);
            if constexpr (EK != 0) {
              float av[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) av[e] = 0.f;
              if constexpr (EK >= 3) {
                if (eaux) unpack16<bf16>(PRE ? auxr[PRE ? i : 0] : ld16(eaux + (size_t)row * eldaux + col), av);
              }
              const uint32_t idx0 = (uint32_t)(row * eN + col);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float x = v[e];
                if constexpr (EK == 1 || EK == 2) x = x > 0.f ? x : 0.f;
                if constexpr (EK == 2 || EK == 4) x = (p5_mix32((idx0 + e) ^ hseed) >> 8) >= ethr ? x * escale : 0.f;
                if constexpr (EK == 3 || EK == 4) x += av[e];
                if constexpr (EK == 5) x = av[e] > 0.f ? x : 0.f;
                v[e] = x;
              }
            }
            const u32x4 packed = pack16<bf16>(v);
            st16(eC + (size_t)row * eldc + col, packed);
            if (essq) {    // sum of squares of the row as stored
              float w[8];
              unpack16<bf16>(packed, w);
#pragma unroll
              for (int e = 0; e < 8; ++e) ss += w[e] * w[e];
            }
          }
          if (essq) {      // (uniform branch) the PPR lanes of a tile row are adjacent: one atomic per tile row
            static_assert(PPR <= 64 && (PPR & (PPR - 1)) == 0, "pieces per row: power of two within a wave");
            if (essq_nt > 0) {   // one partial per 64-column group (8 adjacent lanes), plain store: exactly one writer
              static_assert(PPR >= 8 || BN < 64, "a 64-column group is 8 pieces");
#pragma unroll
              for (int m = (PPR < 8 ? PPR : 8) / 2; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
              if ((pc & 7) == 0 && ok) essq[(size_t)row * essq_nt + (col >> 6)] = ss;
            } else {
#pragma unroll
              for (int m = PPR / 2; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
              if (pc == 0 && row < eM) atomicAdd(essq + row, ss);
            }
          }
        }
      };
      if (eepi == P5_EPI_RELU_DROP) { if (do_drop) pieces(P5EpiTag<2>{}); else pieces(P5EpiTag<1>{}); }
      else if (eepi == P5_EPI_RESID_DROP) { if (do_drop) pieces(P5EpiTag<4>{}); else pieces(P5EpiTag<3>{}); }
      else if (eepi == P5_EPI_MASK_POS) pieces(P5EpiTag<5>{});
      else pieces(P5EpiTag<0>{});
      return;
    }
  }

  // ---- direct epilogue: lane owns C[row = (lane>>4)*4 + r][col = lane & 15] of every 16x16 tile ----
  // fp32 outputs without auxiliary operand (weight gradients: store / atomic add / accumulate): the kind is a compile-time
  // constant of the loop nest and the descriptor is read once -- the general nest below decides everything per element, about
  // ten scalar branches each.
  if (g.c_f32 && g.aux == nullptr && g.ssq_out == nullptr && g.rowss == nullptr &&
      (g.epi == P5_EPI_STORE || g.epi == P5_EPI_ATOMIC || g.epi == P5_EPI_ACCUM)) {
    const int eM = g.M, eN = g.N, eldc = g.ldc;
    const float ealpha = g.alpha;
    float* const eC = (float*)g.C + (g.c_split_stride > 0 ? (size_t)blockIdx.z * (size_t)g.c_split_stride : (size_t)0);
    auto nest = [&](auto ek) {
      constexpr int EK = decltype(ek)::value;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * (BM / WMW) + i * 16 + (lane >> 4) * 4 + r;
          float* const crow = eC + (size_t)row * eldc;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WNW) + j * 16 + (lane & 15);
            if (row < eM && col < eN) {
              const float v = acc[i][j][r] * ealpha;
              if constexpr (EK == P5_EPI_ATOMIC) atomicAdd(crow + col, v);
              else if constexpr (EK == P5_EPI_ACCUM) crow[col] += v;
              else crow[col] = v;
            }
          }
        }
    };
    if (g.epi == P5_EPI_ATOMIC && g.c_split_stride > 0) nest(P5EpiTag<P5_EPI_STORE>{});
    else if (g.epi == P5_EPI_ATOMIC) nest(P5EpiTag<P5_EPI_ATOMIC>{});
    else if (g.epi == P5_EPI_ACCUM) nest(P5EpiTag<P5_EPI_ACCUM>{});
    else nest(P5EpiTag<P5_EPI_STORE>{});
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WNW) + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (BM / WMW) + i * 16 + (lane >> 4) * 4 + r;
        if (row >= g.M || col >= g.N) continue;
        const size_t ci = (size_t)row * g.ldc + col;
        float auxv = 0.f;
        if (g.aux) auxv = to_f<T>(((const T*)g.aux)[(size_t)row * g.ldaux + col]);
        float sc = g.alpha;
        if (g.rowss) sc *= gemm_row_rstd(g, row);
        const float v = gemm_epi_apply(g, acc[i][j][r] * sc, auxv, seed, do_drop, row, col);
        if (g.ssq_out) {
          const float w = g.c_f32 ? v : to_f<T>(from_f<T>(v));
          atomicAdd(g.ssq_out + row, w * w);
        }
        if (g.epi == P5_EPI_ATOMIC) {
          atomicAdd(((float*)g.C) + ci, v);
        } else if (g.epi == P5_EPI_ACCUM) {
          ((float*)g.C)[ci] += v;
        } else if (g.c_f32) {
          ((float*)g.C)[ci] = v;
        } else {
          ((T*)g.C)[ci] = from_f<T>(v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 operands on the f16 matrix cores (round 5; the verification pass of p5_verify.h).  v_mfma_f32_16x16x4_f32 runs at 1/16 of the
// f16 / bf16 MFMA rate.  Each operand element is split exactly as  x = hi + lo / 4096  with hi = fp16(x) (11 significant bits) and
// lo = fp16((x - hi) * 4096) (the next 11 bits; the scaling keeps lo a normal fp16 number for |x| > 2^-14 * 2^-1, far below anything a
// T5 weight or activation needs), and  a.b = hi_a hi_b + (hi_a lo_b + lo_a hi_b) / 4096 [+ lo_a lo_b / 4096^2]  is accumulated in two
// fp32 accumulators by three (P5_SPLIT_TERMS = 4: four) v_mfma_f32_16x16x32_f16 per 32 values of K -- 51 (68) matrix-pipe cycles where
// the fp32 instruction needs 256.  Every product of two fp16 numbers is exact in fp32, so what is lost is the dropped lo.lo term: 2^-22
// relative per product (with four terms: nothing down to 2^-33), against fp32's own 2^-24 rounding of each product.
// The split is done ONCE per staged element on its way into LDS (not per fragment use), into the SAME LDS images: the two 64-byte
// chunks of a K-step hold the hi halves and the lo halves -- a 16-byte piece = 8 fp16 = [4 elements of chunk 0, 4 of chunk 1] of
// that row and piece index, the same K-permutation for A and B -- so the fragment loads are the fp32 kernel's own.
// Limits: |x| < 65504 (fp16 range; T5 activations behind a T5LayerNorm and its weights are O(1)).
// ---------------------------------------------------------------------------------------------------------
#ifndef P5_SPLIT_TERMS
#define P5_SPLIT_TERMS 3
#endif
__host__ __device__ static __forceinline__ unsigned short p5_f2h_bits(float f) {      // round-to-nearest-even, subnormals kept, overflow -> inf
  union { float f; unsigned u; } c; c.f = f;
  const unsigned sign = (c.u >> 16) & 0x8000u, ex = (c.u >> 23) & 0xFFu, man = c.u & 0x7FFFFFu;
  if (ex == 0xFFu) return (unsigned short)(sign | 0x7C00u | (man ? 0x200u : 0u));
  const int e = (int)ex - 127 + 15;
  if (e >= 31) return (unsigned short)(sign | 0x7C00u);
  if (e <= 0) {
    if (e < -10) return (unsigned short)sign;
    const unsigned m = man | 0x800000u;
    const int sh = 14 - e;                       // 24-bit significand -> 10-bit field of a subnormal
    unsigned r = m >> sh;
    const unsigned rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (unsigned short)(sign | r);
  }
  unsigned r = ((unsigned)e << 10) | (man >> 13);
  const unsigned rem = man & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;      // (a carry into the exponent is the right answer, up to inf)
  return (unsigned short)(sign | r);
}
__host__ __device__ static __forceinline__ float p5_h2f_bits(unsigned short h) {
  const unsigned sign = ((unsigned)h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
  union { float f; unsigned u; } c;
  if (ex == 0) {
    if (man == 0) { c.u = sign; return c.f; }
    c.f = (float)man * 5.9604644775390625e-8f;    // 2^-24
    c.u |= sign;
    return c.f;
  }
  if (ex == 31) { c.u = sign | 0x7F800000u | (man << 13); return c.f; }
  c.u = sign | ((ex - 15 + 127) << 23) | (man << 13);
  return c.f;
}
// 4 + 4 fp32 elements (one piece of chunk 0 and of chunk 1 of a K-step) -> 8 hi halves, 8 lo halves
__device__ static __forceinline__ void p5_split8(const u32x4& x0, const u32x4& x1, u32x4& hi, u32x4& lo) {
  float f[8];
  unpack16<float>(x0, f);
  unpack16<float>(x1, f + 4);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P5_EMU)
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const _Float16 h0 = (_Float16)f[2 * q], h1 = (_Float16)f[2 * q + 1];
    const _Float16 l0 = (_Float16)((f[2 * q] - (float)h0) * 4096.f), l1 = (_Float16)((f[2 * q + 1] - (float)h1) * 4096.f);
    hi[q] = __builtin_bit_cast(unsigned, (h2){h0, h1});
    lo[q] = __builtin_bit_cast(unsigned, (h2){l0, l1});
  }
#else
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned short h0 = p5_f2h_bits(f[2 * q]), h1 = p5_f2h_bits(f[2 * q + 1]);
    const unsigned short l0 = p5_f2h_bits((f[2 * q] - p5_h2f_bits(h0)) * 4096.f), l1 = p5_f2h_bits((f[2 * q + 1] - p5_h2f_bits(h1)) * 4096.f);
    hi[q] = (unsigned)h0 | ((unsigned)h1 << 16);
    lo[q] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
#endif
}
// acc += A B^T over the 32 K-values of a K-step; a / b: 8 fp16 per lane (lane group g supplies K-values 8g .. 8g+7 of its row)
#ifdef P5_EMU
static inline void mma32_f16(f32x4& acc, const u32x4& a, const u32x4& b) {
  emu::Wave& w = emu::wave();
  const int l = (int)emu::lane();
  memcpy(w.slot[l], &a, 16);
  memcpy(w.slot[l] + 16, &b, 16);
  emu::wave_barrier();
  const int col = l & 15, rg = l >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = rg * 4 + r;
    float s = acc[r];
    for (int g = 0; g < 4; ++g) {
      const unsigned short* pa = (const unsigned short*)(w.slot[g * 16 + row]);
      const unsigned short* pb = (const unsigned short*)(w.slot[g * 16 + col] + 16);
      for (int j = 0; j < 8; ++j) s += p5_h2f_bits(pa[j]) * p5_h2f_bits(pb[j]);
    }
    acc[r] = s;
  }
  emu::wave_barrier();
}
#else
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ static __forceinline__ void mma32_f16(f32x4& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
#endif
// both chunks of a K-step of a K-contiguous fp32 operand tile, split on the way into LDS (chunk 0 image <- hi, chunk 1 image <- lo)
template <int R>
__device__ static __forceinline__ void stage_store_split(const u32x4* r0, const u32x4* r1, char* lds0, char* lds1, int tid) {
  constexpr int NCH = R * 4 / 256;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    u32x4 hi, lo;
    p5_split8(r0[i], r1[i], hi, lo);
    st16(lds0 + kc_off(c >> 2, c & 3), hi);
    st16(lds1 + kc_off(c >> 2, c & 3), lo);
  }
}

template <class T, int BM, int BN, bool AKS, bool BKS, int NCK, bool ADMA, bool BDMA, int MM = 0>
__global__ __launch_bounds__(256) void p5_gemm_kernel(P5GemmArgs g) {
  static_assert(MM == 0 || (sizeof(T) == 4 && !AKS && !BKS && NCK == 2 && !ADMA && !BDMA), "split products: fp32, K-contiguous, register-staged");
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int KCH = TT<T>::KCH;
  // bytes of one 32-deep K-chunk of an operand tile in LDS (direct-to-LDS images are unpadded)
  constexpr int ACH = (ADMA && AKS) ? 32 * BM * 2 : LdsChunk<T, BM, AKS>::BYTES;
  constexpr int BCH = (BDMA && BKS) ? 32 * BN * 2 : LdsChunk<T, BN, BKS>::BYTES;
  constexpr int STAGE = NCK * (ACH + BCH);
  constexpr int NA = AKS ? (KCH * (BM / TT<T>::EPF) / 256) : (BM * 4 / 256);
  constexpr int NB = BKS ? (KCH * (BN / TT<T>::EPF) / 256) : (BN * 4 / 256);
  constexpr int CST = BN * 2 + 16;                    // LDS row stride of the staged bf16 C tile
  constexpr int LDS_BYTES = (2 * STAGE > BM * CST) ? 2 * STAGE : BM * CST;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (MI355X: workgroup b runs on XCD b % 8, each XCD has a private 4 MiB L2): give every XCD a
  // contiguous run of tiles -- n fastest inside an m row -- so the A panel of a row and the B panels are fetched from
  // HBM once per XCD and then hit in its L2, instead of every workgroup streaming its own 2 x (tile x K) bytes.
  int m0, n0;
  gemm_tile_origin<BM, BN>(g, m0, n0);
  // split-K range (in steps of NCK chunks)
  const int nst = (g.K + KCH * NCK - 1) / (KCH * NCK);
  const int per = (nst + g.splitk - 1) / g.splitk;
  const int st_begin = blockIdx.z * per;
  const int st_end = (st_begin + per < nst) ? st_begin + per : nst;
  if (st_begin >= st_end) return;

  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bp = (const T*)g.B;

  f32x4 acc[TM][TN];
  f32x4 acc2[MM ? TM : 1][MM ? TN : 1];       // split products: the cross terms (scaled by 4096)
#if P5_SPLIT_TERMS >= 4
  f32x4 acc3[MM ? TM : 1][MM ? TN : 1];       // ... and lo.lo (scaled by 4096^2)
#endif
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (MM != 0) {
        acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if P5_SPLIT_TERMS >= 4
        acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
      }
    }

  static_assert(!(ADMA || BDMA) || NCK == 2, "direct-to-LDS staging copies whole K-steps = two K-chunks");
  u32x4 ra[NCK][ADMA ? 1 : NA], rb[NCK][BDMA ? 1 : NB];
#pragma unroll
  for (int c = 0; c < NCK; ++c) {
    const int k0 = (st_begin * NCK + c) * KCH;
    if constexpr (!ADMA) stage_load<T, BM, AKS>(ra[c], A, g.lda, m0, k0, g.M, g.K, tid);
    if constexpr (!BDMA) stage_load<T, BN, BKS>(rb[c], Bp, g.ldb, n0, k0, g.N, g.K, tid);
  }
  if constexpr (ADMA && !AKS) stage_dma128<T, BM>(lds, A, g.lda, m0, st_begin * NCK * KCH, g.M, tid);
  if constexpr (ADMA && AKS) stage_dma_ks<T, BM>(lds, A, g.lda, m0, st_begin * NCK * KCH, g.M, tid);
  if constexpr (BDMA && !BKS) stage_dma128<T, BN>(lds + NCK * ACH, Bp, g.ldb, n0, st_begin * NCK * KCH, g.N, tid);
  if constexpr (BDMA && BKS) stage_dma_ks<T, BN>(lds + NCK * ACH, Bp, g.ldb, n0, st_begin * NCK * KCH, g.N, tid);
  if constexpr (MM != 0) {
    stage_store_split<BM>(ra[0], ra[1], lds, lds + ACH, tid);
    stage_store_split<BN>(rb[0], rb[1], lds + NCK * ACH, lds + NCK * ACH + BCH, tid);
  } else {
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      if constexpr (!ADMA) stage_store<T, BM, AKS>(ra[c], lds + c * ACH, tid);
      if constexpr (!BDMA) stage_store<T, BN, BKS>(rb[c], lds + NCK * ACH + c * BCH, tid);
    }
  }
  __syncthreads();

  for (int st = st_begin; st < st_end; ++st) {
    const int cur = (st - st_begin) & 1;
    const char* base = lds + cur * STAGE;
    const bool more = (st + 1 < st_end);
    if (more) {
      char* nb = lds + (cur ^ 1) * STAGE;   // last read in step st-1, which ended with a barrier
#pragma unroll
      for (int c = 0; c < NCK; ++c) {
        const int k0 = ((st + 1) * NCK + c) * KCH;
        if constexpr (!ADMA) stage_load<T, BM, AKS>(ra[c], A, g.lda, m0, k0, g.M, g.K, tid);
        if constexpr (!BDMA) stage_load<T, BN, BKS>(rb[c], Bp, g.ldb, n0, k0, g.N, g.K, tid);
      }
      if constexpr (ADMA && !AKS) stage_dma128<T, BM>(nb, A, g.lda, m0, (st + 1) * NCK * KCH, g.M, tid);
      if constexpr (ADMA && AKS) stage_dma_ks<T, BM>(nb, A, g.lda, m0, (st + 1) * NCK * KCH, g.M, tid);
      if constexpr (BDMA && !BKS) stage_dma128<T, BN>(nb + NCK * ACH, Bp, g.ldb, n0, (st + 1) * NCK * KCH, g.N, tid);
      if constexpr (BDMA && BKS) stage_dma_ks<T, BN>(nb + NCK * ACH, Bp, g.ldb, n0, (st + 1) * NCK * KCH, g.N, tid);
    }
    if constexpr (MM != 0) {
      u32x4 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fah[i] = frag_load<T, BM, false>(base, wm * (BM / 2) + i * 16, lane);
        fal[i] = frag_load<T, BM, false>(base + ACH, wm * (BM / 2) + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fbh[j] = frag_load<T, BN, false>(base + NCK * ACH, wn * (BN / 2) + j * 16, lane);
        fbl[j] = frag_load<T, BN, false>(base + NCK * ACH + BCH, wn * (BN / 2) + j * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          mma32_f16(acc[i][j], fah[i], fbh[j]);
          mma32_f16(acc2[i][j], fah[i], fbl[j]);
          mma32_f16(acc2[i][j], fal[i], fbh[j]);
#if P5_SPLIT_TERMS >= 4
          mma32_f16(acc3[i][j], fal[i], fbl[j]);
#endif
        }
    } else {
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const char* la = base + c * ACH;
      const char* lb = base + NCK * ACH + c * BCH;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = (ADMA && AKS)   ? frag_load_ksd<T, BM>(base, wm * (BM / 2) + i * 16, c, lane)
                : ADMA          ? frag_load_kc128<T>(base, wm * (BM / 2) + i * 16, c, lane)
                                : frag_load<T, BM, AKS>(la, wm * (BM / 2) + i * 16, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = (BDMA && BKS)   ? frag_load_ksd<T, BN>(base + NCK * ACH, wn * (BN / 2) + j * 16, c, lane)
                : BDMA          ? frag_load_kc128<T>(base + NCK * ACH, wn * (BN / 2) + j * 16, c, lane)
                                : frag_load<T, BN, BKS>(lb, wn * (BN / 2) + j * 16, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
    }
    }
    if (more) {
      char* nb = lds + (cur ^ 1) * STAGE;
      if constexpr (MM != 0) {
        stage_store_split<BM>(ra[0], ra[1], nb, nb + ACH, tid);
        stage_store_split<BN>(rb[0], rb[1], nb + NCK * ACH, nb + NCK * ACH + BCH, tid);
      } else {
#pragma unroll
        for (int c = 0; c < NCK; ++c) {
          if constexpr (!ADMA) stage_store<T, BM, AKS>(ra[c], nb + c * ACH, tid);
          if constexpr (!BDMA) stage_store<T, BN, BKS>(rb[c], nb + NCK * ACH + c * BCH, tid);
        }
      }
    }
    P5_SCHED_FENCE();     // the copies issued at the top of the step stay in flight under the MFMAs above
    __syncthreads();
  }

  if constexpr (MM != 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#if P5_SPLIT_TERMS >= 4
          acc[i][j][r] += (acc2[i][j][r] + acc3[i][j][r] * (1.f / 4096.f)) * (1.f / 4096.f);
#else
          acc[i][j][r] += acc2[i][j][r] * (1.f / 4096.f);
#endif
        }
  }
  gemm_epilogue<T, BM, BN, LDS_BYTES>(g, acc, lds, m0, n0, tid);
}


// ---------------------------------------------------------------------------------------------------------
// fp32 x fp32 K-contiguous GEMM with split products (see p5_split8 above), pipelined three K-steps deep.  p5_gemm_kernel<.., MM = 1> is
// correct but exposes a global-load latency per 32-wide K-step (one step of lookahead: 2 us per step on the verification pass's
// 2560 x 1536 x 512 -- 127 TF/s where the three f16 MFMAs per step take 0.1 us).  Here the loads of K-step s+3 are issued before step s
// is multiplied (three register stages: at 64 x 64 tiles that is 12 registers), the split + LDS store of step s+1 follows the MFMAs of
// step s, one barrier per step.  Loads past K are predicated off and read as zeros, so the loop body has no tail variants.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void p5_gemm_split_kernel(P5GemmArgs g) {
  using T = float;
  constexpr int TM = BM / 32, TN = BN / 32, KCH = TT<T>::KCH;
  constexpr int ACH = LdsChunk<T, BM, false>::BYTES, BCH = LdsChunk<T, BN, false>::BYTES;
  constexpr int STAGE = 2 * (ACH + BCH);
  constexpr int NA = BM * 4 / 256, NB = BN * 4 / 256;
  constexpr int CST = BN * 2 + 16;
  constexpr int LDS_BYTES = (2 * STAGE > BM * CST) ? 2 * STAGE : BM * CST;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  gemm_tile_origin<BM, BN>(g, m0, n0);
  const int nst = (g.K + 2 * KCH - 1) / (2 * KCH);
  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bp = (const T*)g.B;
  f32x4 acc[TM][TN], acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  u32x4 ra0[2][NA], ra1[2][NA], ra2[2][NA], rb0[2][NB], rb1[2][NB], rb2[2][NB];
  // The loads are RAW (inline asm, p5_device.h gload16_raw): with compiler-tracked loads -- predicated per lane for the matrix edges --
  // hipcc waits for EVERY load in flight at the loop header (`s_waitcnt vmcnt(0)`, ISA of the first version), i.e. the three-deep
  // prefetch bought 10 %.  Rows past the edge are clamped (they only feed C rows / columns that are never stored), K is a multiple of
  // 32 (launcher), steps past the last one re-fetch it (constant count of loads in flight), and the waits are counted by hand:
  // LPS loads per step and thread, two later steps may still be in flight when a step is split into LDS.
  constexpr int LPS = 2 * (NA + NB);
  const T* pa[NA];
  const T* pb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = tid + i * 256;
    int gr = m0 + (c >> 2);
    gr = gr < g.M ? gr : g.M - 1;
    pa[i] = A + (size_t)gr * g.lda + (c & 3) * TT<T>::EPF;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int c = tid + i * 256;
    int gr = n0 + (c >> 2);
    gr = gr < g.N ? gr : g.N - 1;
    pb[i] = Bp + (size_t)gr * g.ldb + (c & 3) * TT<T>::EPF;
  }
  auto load = [&](u32x4(&ra)[2][NA], u32x4(&rb)[2][NB], int st) {
    const int k0 = (st < nst ? st : nst - 1) * 2 * KCH;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < NA; ++i) gload16_raw(ra[c][i], pa[i] + k0 + c * KCH);
#pragma unroll
      for (int i = 0; i < NB; ++i) gload16_raw(rb[c][i], pb[i] + k0 + c * KCH);
    }
  };
  auto store = [&](const u32x4(&ra)[2][NA], const u32x4(&rb)[2][NB], char* stage) {
    stage_store_split<BM>(ra[0], ra[1], stage, stage + ACH, tid);
    stage_store_split<BN>(rb[0], rb[1], stage + 2 * ACH, stage + 2 * ACH + BCH, tid);
  };
  auto compute = [&](const char* base) {
    u32x4 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      fah[i] = frag_load<T, BM, false>(base, wm * (BM / 2) + i * 16, lane);
      fal[i] = frag_load<T, BM, false>(base + ACH, wm * (BM / 2) + i * 16, lane);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      fbh[j] = frag_load<T, BN, false>(base + 2 * ACH, wn * (BN / 2) + j * 16, lane);
      fbl[j] = frag_load<T, BN, false>(base + 2 * ACH + BCH, wn * (BN / 2) + j * 16, lane);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        mma32_f16(acc[i][j], fah[i], fbh[j]);
        mma32_f16(acc2[i][j], fah[i], fbl[j]);
        mma32_f16(acc2[i][j], fal[i], fbh[j]);
      }
  };
  load(ra0, rb0, 0);
  load(ra1, rb1, 1);
  load(ra2, rb2, 2);
  P5_WAIT_VM(2 * LPS);
  P5_SCHED_FENCE();
  store(ra0, rb0, lds);
  P5_BARRIER_LDS();
  // step s lives in register stage s % 3 until it is stored; LDS stage s & 1
#define P5_SPLIT_STEP(RA_CUR, RB_CUR, RA_NEXT, RB_NEXT, OFF)                                  \
  if (st + (OFF) < nst) {                                                                       \
    load(RA_CUR, RB_CUR, st + (OFF) + 3);              /* (its own step was stored a step ago) */ \
    compute(lds + ((st + (OFF)) & 1) * STAGE);                                                  \
    if (st + (OFF) + 1 < nst) {                                                                 \
      P5_WAIT_VM(2 * LPS);                             /* step s+1 has landed; s+2, s+3 may fly */ \
      P5_SCHED_FENCE();                                                                         \
      store(RA_NEXT, RB_NEXT, lds + ((st + (OFF) + 1) & 1) * STAGE);                              \
    }                                                                                           \
    P5_SCHED_FENCE();                                                                           \
    P5_BARRIER_LDS();                                                                           \
  }
  for (int st = 0; st < nst; st += 3) {
    P5_SPLIT_STEP(ra0, rb0, ra1, rb1, 0)
    P5_SPLIT_STEP(ra1, rb1, ra2, rb2, 1)
    P5_SPLIT_STEP(ra2, rb2, ra0, rb0, 2)
  }
  P5_WAIT_VM(0);            // (the re-fetches past the last step) before the epilogue's own memory traffic
#undef P5_SPLIT_STEP
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += acc2[i][j][r] * (1.f / 4096.f);
  gemm_epilogue<T, BM, BN, LDS_BYTES>(g, acc, lds, m0, n0, tid);
}



// ---------------------------------------------------------------------------------------------------------
// v2 main loop for the bf16 K-contiguous x K-contiguous case (every forward Linear): same tile, same LDS image
// (128-byte rows copied straight into LDS, XOR-swizzled), same epilogue as p5_gemm_kernel, but software-pipelined by
// hand.  The compiler's schedule of the loop above reads a fragment group, waits lgkmcnt(0), issues 8 MFMAs, reads the
// next group ... and ends every K-step with vmcnt(0) + barrier, so the matrix pipe idles during every LDS round trip and
// a step lasts at least one global-load latency.  Here
//   * fragments are double-buffered in registers: the 8 ds_read_b128 of the next 32-wide K-half are issued under the 16
//     MFMAs of the current one (sched_group_barrier pins the 1 read : 2 MFMA interleave);
//   * NST = 3: a three-deep LDS ring.  The copy of stage s+2 is issued at the top of step s, the single barrier sits in
//     the MIDDLE of the step (after a counted vmcnt that only waits for stage s+1, which has had a whole step to land),
//     and the second half prefetches the first fragments of stage s+1 -- no LDS or HBM latency is exposed in steady state;
//   * NST = 2: two stages (two workgroups per CU instead of one); fragments of both K-halves are requested at the top.
// ---------------------------------------------------------------------------------------------------------
template <bool B> struct P5Bool { static constexpr bool value = B; };

#ifdef P5_EMU
#define P5_WAVES_PER_SIMD(lo, hi)
#else
#define P5_WAVES_PER_SIMD(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
// (three and more stages fill the LDS of a CU with one workgroup = one wave per SIMD: let it have the whole register file)
template <int BM, int BN, int NST, bool AKS = false, bool BKS = false>
__global__ __launch_bounds__(256) P5_WAVES_PER_SIMD(1, NST >= 3 ? 1 : 2) void p5_gemm2_kernel(P5GemmArgs g) {
  using T = bf16;
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int ASZ = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int CST = BN * 2 + 16;
  constexpr int LDS_BYTES = (NST * STAGE > BM * CST) ? NST * STAGE : BM * CST;
  constexpr int NDMA = (BM + BN) / 32;                // direct-to-LDS wave instructions per wave per stage
  constexpr int NFR = TM + TN, NMM = TM * TN;
  static_assert(NST >= 2 && NST <= 8 && (NST - 2) * ((BM + BN) / 32) <= 60, "LDS stages / vmcnt range");
  constexpr int PFD = NST - 1;                        // stages in flight ahead of the one being multiplied
  // fragment reads / stage copies issued behind each MFMA of a K-half (1 whenever the wave tile has at least as many MFMAs as
  // fragments, i.e. from 64x64 up; 2 for the 32x64 tile of the sub-CU-count decoder problems)
  constexpr int LPM = (NFR + NMM - 1) / NMM, DPM = (NDMA + NMM - 1) / NMM;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS destinations of the copies stay in SGPRs
#endif
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  gemm_tile_origin<BM, BN>(g, m0, n0);
  const int nst = g.K / 64;
  const int per = (nst + g.splitk - 1) / g.splitk;
  const int st_begin = blockIdx.z * per;
  const int st_end = (st_begin + per < nst) ? st_begin + per : nst;
  if (st_begin >= st_end) return;
  const int n = st_end - st_begin;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane source pointers of this wave's copies (row = 8 rows per wave instruction, 16-byte slot XOR-ed with row & 7 on
  // the source side: a wave instruction's LDS image is linear); they advance by 128 bytes per stage
  // K-strided operands: [64 k-rows][R] image with the chunk swizzle of stage_dma_ks; they advance by 64 rows per stage
  const T* srcA[BM / 32];
  const T* srcB[BN / 32];
  size_t incA = AKS ? (size_t)64 * g.lda : 64, incB = BKS ? (size_t)64 * g.ldb : 64;
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    if constexpr (AKS) {
      constexpr int CPR = BM / 8, RPI = 512 / BM;
      const int krow = (wave * (BM / 32) + i) * RPI + lane / CPR;
      int cg = (lane % CPR) ^ ksd_swz<BM>(krow);
      const int cmax = (g.lda - m0) / 8 - 1;      // (row capacity of the k-row in memory, see stage_dma_ks)
      cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
      srcA[i] = (const T*)g.A + ((size_t)st_begin * 64 + krow) * g.lda + m0 + cg * 8;
    } else {
      const int row = (wave * (BM / 32) + i) * 8 + (lane >> 3);
      int gr = m0 + row;
      gr = gr < g.M ? gr : g.M - 1;
      srcA[i] = (const T*)g.A + (size_t)gr * g.lda + (size_t)st_begin * 64 + (((lane & 7) ^ (row & 7)) * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    if constexpr (BKS) {
      constexpr int CPR = BN / 8, RPI = 512 / BN;
      const int krow = (wave * (BN / 32) + i) * RPI + lane / CPR;
      int cg = (lane % CPR) ^ ksd_swz<BN>(krow);
      const int cmax = (g.ldb - n0) / 8 - 1;
      cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
      srcB[i] = (const T*)g.B + ((size_t)st_begin * 64 + krow) * g.ldb + n0 + cg * 8;
    } else {
      const int row = (wave * (BN / 32) + i) * 8 + (lane >> 3);
      int gr = n0 + row;
      gr = gr < g.N ? gr : g.N - 1;
      srcB[i] = (const T*)g.B + (size_t)gr * g.ldb + (size_t)st_begin * 64 + (((lane & 7) ^ (row & 7)) * 8);
    }
  }
  auto copy_one = [&](int buf, int idx) {   // idx-th wave instruction of the copy of the next not-yet-copied stage
    char* b = lds + buf * STAGE;
    if (idx < BM / 32) {
      glds16_raw(srcA[idx], b + (wave * (BM / 32) + idx) * 1024);
      srcA[idx] += incA;
    } else {
      const int i = idx - BM / 32;
      glds16_raw(srcB[i], b + ASZ + (wave * (BN / 32) + i) * 1024);
      srcB[i] += incB;
    }
  };
  auto copy_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) copy_one(buf, i);
  };
  // fragment idx of a K-half, in the order the MFMAs below first need them: A row-block 0, all B column-blocks, other A
  // K-strided operands: lane offsets of the transpose reads, fixed for the whole kernel
  int koffA[AKS ? TM : 1], koffB[BKS ? TN : 1];
  if constexpr (AKS) {
#pragma unroll
    for (int i = 0; i < TM; ++i) koffA[i] = ksd_lane_off<BM>(wm * (BM / 2) + i * 16, lane);
  }
  if constexpr (BKS) {
#pragma unroll
    for (int j = 0; j < TN; ++j) koffB[j] = ksd_lane_off<BN>(wn * (BN / 2) + j * 16, lane) + ASZ;
  }
  auto tr_frag = [](const char* p, int hi_off) {
    const u32x2 lo = lds_tr16_b64(p), hi = lds_tr16_b64(p + hi_off);
    u32x4 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return r;
  };
  auto load_one = [&](u32x4(&fa)[TM], u32x4(&fb)[TN], int buf, auto c_c, int idx) {
    constexpr int c = decltype(c_c)::value ? 1 : 0;
    const char* b = lds + buf * STAGE;
    auto fra = [&](int i) {
      if constexpr (AKS) return tr_frag(b + koffA[i] + c * 32 * BM * 2, 4 * BM * 2);
      else return frag_load_kc128<T>(b, wm * (BM / 2) + i * 16, c, lane);
    };
    auto frb = [&](int j) {
      if constexpr (BKS) return tr_frag(b + koffB[j] + c * 32 * BN * 2, 4 * BN * 2);
      else return frag_load_kc128<T>(b + ASZ, wn * (BN / 2) + j * 16, c, lane);
    };
    if (idx == 0) fa[0] = fra(0);
    else if (idx <= TN) fb[idx - 1] = frb(idx - 1);
    else fa[idx - TN] = fra(idx - TN);
  };
  auto load_frags = [&](u32x4(&fa)[TM], u32x4(&fb)[TN], int buf, auto c_c) {
#pragma unroll
    for (int i = 0; i < NFR; ++i) load_one(fa, fb, buf, c_c, i);
  };
  u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];

  if constexpr (NST >= 3) {
    // ONE loop body for every K-step (peeled prologue / tail variants make hipcc shuffle the 64 accumulators through
    // ~100 register copies per step): every step issues a stage copy and prefetches "the next stage's" first fragments.
    // Past the end of the K range the copy re-fetches the last stage into a free ring slot (`inc` drops to 0) and the
    // prefetched fragments are never multiplied -- a few hundred wasted bytes per workgroup for a loop without branches,
    // with a constant vmcnt: at the mid-step barrier exactly PFD-1 younger stage copies may still be in flight.
    // The instruction order below IS the schedule: a fence after every MFMA keeps hipcc from regrouping it.
    auto step = [&](int buf, int nb1, int nb2) {
      P5_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        mma16<T>(acc[t / TN][t % TN], fa0[t / TN], fb0[t % TN]);
        P5_SCHED_FENCE();
#pragma unroll
        for (int q = t * LPM; q < (t + 1) * LPM; ++q)
          if (q < NFR) load_one(fa1, fb1, buf, P5Bool<true>(), q);
#pragma unroll
        for (int q = t * DPM; q < (t + 1) * DPM; ++q)
          if (q < NDMA) copy_one(nb2, q);   // stage s+PFD -> ring slot of stage s-1 (read out before that step's barrier)
        P5_SCHED_FENCE();
      }
      P5_WAIT_VM((PFD - 1) * NDMA);         // this wave's share of stage s+1 has landed
      P5_BARRIER_LDS();                     // ... everyone's; and slot `buf` is fully read
      P5_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        mma16<T>(acc[t / TN][t % TN], fa1[t / TN], fb1[t % TN]);
        P5_SCHED_FENCE();
#pragma unroll
        for (int q = t * LPM; q < (t + 1) * LPM; ++q)
          if (q < NFR) load_one(fa0, fb0, nb1, P5Bool<false>(), q);
        P5_SCHED_FENCE();
      }
    };
    // a copy post-increments its source pointers: the increment must already be 0 when the LAST stage is copied
    auto before_copy_of = [&](int stage) {
      if (stage >= n - 1) { incA = 0; incB = 0; }
    };
#pragma unroll
    for (int q = 0; q < PFD; ++q) {
      before_copy_of(q);
      copy_stage(q);
    }
    P5_WAIT_VM((PFD - 1) * NDMA);
    P5_BARRIER_LDS();
    load_frags(fa0, fb0, 0, P5Bool<false>());
    int buf = 0, s = 0;
    do {    // n >= 1; (a for loop's zero-trip guard makes hipcc read all 64 accumulators back to VGPRs in every iteration)
      const int nb1 = buf == NST - 1 ? 0 : buf + 1, nb2 = buf == 0 ? NST - 1 : buf - 1;
      before_copy_of(s + PFD);
      step(buf, nb1, nb2);
      buf = nb1;
    } while (++s < n);
  } else {
    // two-slot ring: the copy of stage s+1 is issued at the top of step s and has the whole step to land; the barrier sits
    // at the END of the step, so only the first fragment reads of the next stage are exposed (16 reads against 128 MFMAs
    // for a 256x256 tile)
    auto step = [&](auto copy_c, int buf) {
      constexpr bool COPY = decltype(copy_c)::value;
      P5_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        mma16<T>(acc[t / TN][t % TN], fa0[t / TN], fb0[t % TN]);
        P5_SCHED_FENCE();
#pragma unroll
        for (int q = t * LPM; q < (t + 1) * LPM; ++q)
          if (q < NFR) load_one(fa1, fb1, buf, P5Bool<true>(), q);
        if constexpr (COPY) {
#pragma unroll
          for (int q = t * DPM; q < (t + 1) * DPM; ++q)
            if (q < NDMA) copy_one(buf ^ 1, q);
        }
        P5_SCHED_FENCE();
      }
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        mma16<T>(acc[t / TN][t % TN], fa1[t / TN], fb1[t % TN]);
        P5_SCHED_FENCE();
      }
      if constexpr (COPY) {
        P5_WAIT_VM(0);
        P5_BARRIER_LDS();
        P5_SCHED_FENCE();
        load_frags(fa0, fb0, buf ^ 1, P5Bool<false>());
        P5_SCHED_FENCE();
      }
    };
    copy_stage(0);
    P5_WAIT_VM(0);
    P5_BARRIER_LDS();
    load_frags(fa0, fb0, 0, P5Bool<false>());
    int s = 0;
    for (; s + 1 < n; ++s) step(P5Bool<true>(), s & 1);
    step(P5Bool<false>(), s & 1);
  }
  // The trailing copies of the ring (re-fetches of the last K-step into free slots, issued to keep vmcnt a constant) may still be
  // in flight here -- up to PFD - 1 K-steps of them -- and the LDS-staged epilogue writes the C tile over the first ring slots: a
  // copy that lands after that corrupts the tile.  Which slots they target depends on the number of K-steps modulo the ring
  // depth (K = 768 on the eight-slot ring hits slot 0), so the race showed only at some shapes and only under memory load (round 3:
  // bf16 gradients of shared.weight / the first decoder norm at T5-base depth 12+12).  Drain them first.
  P5_WAIT_VM(0);
  P5_BARRIER_LDS();     // all fragment reads retired before the epilogue reuses the ring
  gemm_epilogue<T, BM, BN, LDS_BYTES>(g, acc, lds, m0, n0, tid);
}


// ---------------------------------------------------------------------------------------------------------
// 256x256 tiles, eight waves: the 128x128 kernels copy (BM+BN)/(BM*BN) = 1/64 byte from L2 into LDS per MAC and stall at
// ~0.9 PFLOP/s on that copy rate whatever the loop looks like; a 256x256 tile halves it.  One
// workgroup of 512 threads per CU (2 waves per SIMD, <= 256 registers each): waves as WMW x WNW, wave tile
// (BM/WMW) x (BN/WNW) = 128 x 64 -> 128 accumulator registers.  Fragments cannot be double-buffered per K-half at this
// size, so A fragments run through a 4-deep register ring three MFMA groups ahead of their use and the B fragments of
// the other K-half are fetched during the first groups of the current one.  Two-slot LDS ring (2 x 64 KiB): the copy of
// stage s+1 is issued during step s, "vmcnt(0) + barrier" closes the step, then 7 fragment reads restart the pipeline.
// bf16, both operands K-contiguous, K % 64 == 0; bf16 C through the LDS-staged epilogue (135 KiB) or fp32/atomic direct.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WMW, int WNW>
__global__ __launch_bounds__(WMW* WNW * 64) void p5_gemm3_kernel(P5GemmArgs g) {
  using T = bf16;
  constexpr int NW = WMW * WNW, NT = NW * 64;
  constexpr int TM = BM / (16 * WMW), TN = BN / (16 * WNW);
  constexpr int ASZ = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int CST = BN * 2 + 16;
  constexpr int LDS_BYTES = (2 * STAGE > BM * CST) ? 2 * STAGE : BM * CST;
  constexpr int NA = BM / (8 * NW), NB = BN / (8 * NW), NDMA = NA + NB;   // direct-to-LDS wave instructions per wave per stage
  constexpr int RA = 4, PF = 3;                                           // A-fragment ring / prefetch distance (MFMA groups)
  constexpr int NG = 2 * TM;                                              // MFMA groups per K-step: (K-half, A row-block)
  static_assert(TN <= TM && NDMA <= NG && PF < RA && PF <= TM, "static schedule below");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave / WNW, wn = wave % WNW;
  int m0, n0;
  gemm_tile_origin<BM, BN>(g, m0, n0);
  const int nst = g.K / 64;
  const int per = (nst + g.splitk - 1) / g.splitk;
  const int st_begin = blockIdx.z * per;
  const int st_end = (st_begin + per < nst) ? st_begin + per : nst;
  if (st_begin >= st_end) return;
  const int n = st_end - st_begin;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* srcA[NA];
  const T* srcB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave * NA + i) * 8 + (lane >> 3);
    int gr = m0 + row;
    gr = gr < g.M ? gr : g.M - 1;
    srcA[i] = (const T*)g.A + (size_t)gr * g.lda + (size_t)st_begin * 64 + (((lane & 7) ^ (row & 7)) * 8);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + (lane >> 3);
    int gr = n0 + row;
    gr = gr < g.N ? gr : g.N - 1;
    srcB[i] = (const T*)g.B + (size_t)gr * g.ldb + (size_t)st_begin * 64 + (((lane & 7) ^ (row & 7)) * 8);
  }
  auto copy_one = [&](int buf, int idx) {
    char* b = lds + buf * STAGE;
    if (idx < NA) {
      glds16(srcA[idx], b + (wave * NA + idx) * 1024);
      srcA[idx] += 64;
    } else {
      const int i = idx - NA;
      glds16(srcB[i], b + ASZ + (wave * NB + i) * 1024);
      srcB[i] += 64;
    }
  };
  u32x4 ar[RA], bc[2][TN];
  auto load_a = [&](int buf, int gi) {        // A fragment of MFMA group gi (K-half gi / TM, row-block gi % TM) into its ring slot
    ar[gi % RA] = frag_load_kc128<T>(lds + buf * STAGE, wm * (BM / WMW) + (gi % TM) * 16, gi / TM, lane);
  };
  auto load_b = [&](int buf, int c, int j) {
    bc[c][j] = frag_load_kc128<T>(lds + buf * STAGE + ASZ, wn * (BN / WNW) + j * 16, c, lane);
  };
  auto restart = [&](int buf) {               // the reads that must precede the first MFMA of a stage
#pragma unroll
    for (int j = 0; j < TN; ++j) load_b(buf, 0, j);
#pragma unroll
    for (int gi = 0; gi < PF; ++gi) load_a(buf, gi);
  };
  auto step = [&](auto copy_c, int buf) {
    constexpr bool COPY = decltype(copy_c)::value;
    P5_SCHED_FENCE();
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        mma16<T>(acc[gi % TM][j], ar[gi % RA], bc[gi / TM][j]);
        P5_SCHED_FENCE();
        if (j == 0 && gi + PF < NG) load_a(buf, gi + PF);
        if (j == 1 && gi < TN) load_b(buf, 1, gi);
        if constexpr (COPY)
          if (j == 2 % TN && gi < NDMA) copy_one(buf ^ 1, gi);
        P5_SCHED_FENCE();
      }
    }
    if constexpr (COPY) {
      P5_WAIT_VM(0);
      P5_BARRIER_LDS();
      P5_SCHED_FENCE();
      restart(buf ^ 1);
      P5_SCHED_FENCE();
    }
  };
#pragma unroll
  for (int i = 0; i < NDMA; ++i) copy_one(0, i);
  P5_WAIT_VM(0);
  P5_BARRIER_LDS();
  restart(0);
  int s = 0;
  for (; s + 1 < n; ++s) step(P5Bool<true>(), s & 1);
  step(P5Bool<false>(), s & 1);
  P5_BARRIER_LDS();     // all fragment reads retired before the epilogue reuses the ring
  gemm_epilogue<T, BM, BN, LDS_BYTES, NT, WNW>(g, acc, lds, m0, n0, tid);
}
