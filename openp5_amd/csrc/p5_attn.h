// p5_attn.h -- T5 attention (training / encoder path): unscaled QK^T + relative-position bias + key-padding /
// causal mask -> softmax -> (dropout) -> PV, and its backward.  d_kv is fixed at 64 (every T5 checkpoint).
//
// Restates HF T5Attention.forward + eager_attention_forward (HF modeling_t5.py:144-173, 281-369) as used by
// JointEncoder / the decoder T5Stack (P5_T5.py:136-171, 338-350):
//   * scores are NOT scaled by 1/sqrt(d_kv) (modeling_t5.py:197);
//   * position bias = RelEmb[bucket(k - q), h] (modeling_t5.py:217-279), never materialised as [B,H,L,L]:
//     the host passes the bucket LUT (exact torch fp32 bucket semantics) and each workgroup expands the
//     per-head bias over relative positions into LDS;
//   * masked keys get -inf (the reference adds finfo.min; identical whenever a row has one unmasked key).
//
// Layout: Q/K/V are column slices of the fused projection output ([rows, 3*inner] etc.), addressed with a row
// stride; head h occupies columns h*64..h*64+63.  No head transpose is ever written to HBM.
//
// Forward: one wave owns 16 query rows and keeps the full score row block (<= 512 keys) in accumulators, so
// the softmax is the exact two-pass max/sum of the reference; K and V stream through LDS in 64-key tiles
// shared by the 4 waves of the workgroup.  Backward: dQ kernel (per 16 queries, streams keys) and dK/dV
// kernel (per 16 keys, streams queries) recompute P from the saved log-sum-exp.
#pragma once
#include "p5_device.h"
#include "p5_rng.h"

struct P5AttnArgs {
  const void* Q; const void* K; const void* V;
  void* O;
  float* lse;            // [B,H,Lq]
  const void* dO;
  void* dQ; void* dK; void* dV;
  float* Dvec;           // [B,H,Lq]  rowsum(dO * O)
  const float* rel_table;   // [num_buckets, H] fp32 or nullptr (cross-attention: zero bias)
  float* d_rel_table;       // grad of rel_table (atomic accumulate) or nullptr
  const int* bucket_lut;    // bucket of rel = k - q at [rel + lut_half]
  int lut_half;
  const int64_t* kmask;     // [B, Lk], nonzero = attend; nullptr = all
  int B, H, Lq, Lk;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int causal;
  int rel_copies;        // number of buckets of the relative-bias table: the workgroup STORES that many sums per head into its slot of d_rel_table
  int rel_stride;
  P5Drop drop;
  // dropout keep decisions of the probabilities, written by the head-resident forward and read by the head-resident backward (bf16,
  // 128 < L <= 512) instead of being re-hashed twice: per (batch, head, 16-query block) 256 words -- word [half * 128 + n], n = t * 4 + r,
  // is the lane mask (lanes 32 half .. 32 half + 31; lane = 16 g + li) of keep(query 16 block + li, key 16 t + 4 g + r).  nullptr: hash.
  uint32_t* keep_bits;
  // self-attention backward (fused kernel, bf16, Lq == Lk <= 128): dot_out[(b * Lq + token) * H + h] = <dq, q> + <dk, k> + <dv, v> of the
  // token's 64 head columns, from the values as stored (bf16) -- the H partial sums of <d qkv, qkv> per row that the T5LayerNorm backward in
  // the epilogue of the qkv data-gradient GEMM needs (P5_EPI_NORM_BWD, p5_gemm.h).  nullptr: not written.
  float* dot_out;
};


template <bool B> struct AttnBool { static constexpr bool value = B; };
template <int V> struct P5EpiTagA { static constexpr int value = V; };
template <class T> struct AttnC {
  static constexpr int SZ = (int)sizeof(T);
  static constexpr int KCH = TT<T>::KCH;
  static constexpr int EPF = TT<T>::EPF;
  static constexpr int NCK = 64 / KCH;        // 64-byte K-chunks that cover 64 elements
  static constexpr int TS = 64 * SZ + 16;     // LDS row stride of a [64][64] tile (bytes)
  static constexpr int PPR = 64 * SZ / 16;    // 16-byte pieces per tile row
};

// A wave's [16][64] fp32 result (MFMA C layout: lane owns rows g*4+r, column li of every 16-column block) -> global rows of
// 64 T elements.  Staged through the wave's own [16][TS] LDS scratch so that each lane issues 16-byte stores of whole
// 128-/256-byte row segments instead of sixteen 2-byte stores (no barrier: a wave's LDS operations execute in order).
template <class T>
__device__ static __forceinline__ void wave_store_16x64(T* __restrict__ out, size_t ld, int row0, int row_end, const f32x4 (&acc)[4],
                                                        const float (&rs)[4], char* pw, int lane) {
  using C = AttnC<T>;
  const int g = lane >> 4, li = lane & 15;
  if ((ld % C::EPF) != 0 || ((uintptr_t)out % 16) != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + g * 4 + r;
      if (row >= row_end) continue;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) out[(size_t)row * ld + dt * 16 + li] = from_f<T>(acc[dt][r] * rs[r]);
    }
    return;
  }
  P5_WAVE_SYNC();      // the scratch may still be read by slower lanes of this wave (emulator)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(T*)(pw + (g * 4 + r) * C::TS + (dt * 16 + li) * C::SZ) = from_f<T>(acc[dt][r] * rs[r]);
  P5_WAVE_SYNC();
  constexpr int RPP = 64 / C::PPR;       // rows covered by one wave instruction
#pragma unroll
  for (int p = 0; p < 16 / RPP; ++p) {
    const int lr = p * RPP + lane / C::PPR, piece = lane % C::PPR;
    if (row0 + lr < row_end) st16(out + (size_t)(row0 + lr) * ld + piece * C::EPF, ld16(pw + lr * C::TS + piece * 16));
  }
}

// The same store for a bf16 [16][64] result whose row-wise inner product with the FORWARD values of the same rows is wanted as well
// (P5AttnArgs::dot_out): fw[p] = this lane's 16-byte piece (row p * 8 + lane / 8, piece lane % 8) of the forward tile; dot[p] accumulates
// the lane's eight products of the values AS STORED.  Vector path only (the caller checks ld % 8 == 0 and the alignment).
template <class T>
__device__ static __forceinline__ void wave_store_16x64_dot(T* __restrict__ out, size_t ld, int row0, int row_end, const f32x4 (&acc)[4], char* pw,
                                                            int lane, const u32x4 (&fw)[2], float (&dot)[2]) {
  using C = AttnC<T>;
  static_assert(sizeof(T) == 2, "bf16 tiles: eight 16-byte pieces per row");
  const int g = lane >> 4, li = lane & 15;
  P5_WAVE_SYNC();
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(T*)(pw + (g * 4 + r) * C::TS + (dt * 16 + li) * C::SZ) = from_f<T>(acc[dt][r]);
  P5_WAVE_SYNC();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int lr = p * 8 + (lane >> 3), piece = lane & 7;
    const u32x4 v = ld16(pw + lr * C::TS + piece * 16);
    if (row0 + lr < row_end) {
      st16(out + (size_t)(row0 + lr) * ld + piece * 8, v);
      float x[8], y[8];
      unpack16<T>(v, x);
      unpack16<T>(fw[p], y);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot[p] += x[e] * y[e];
    }
  }
}

template <class T>
__device__ static __forceinline__ void stage_tile64(char* lds, const T* base, int ld, int valid_rows, int tid) {
  using C = AttnC<T>;
  constexpr int NP = 64 * C::PPR / 256;
  u32x4 v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + i * 256, row = p / C::PPR, pc = p % C::PPR;
    v[i] = row < valid_rows ? ld16(base + (size_t)row * ld + pc * C::EPF) : zero16();
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + i * 256, row = p / C::PPR, pc = p % C::PPR;
    st16(lds + row * C::TS + pc * 16, v[i]);
  }
}

// fragment with the reduction along the tile's 64 contiguous elements (rows r0..r0+15, K-chunk c)
template <class T>
__device__ static __forceinline__ u32x4 tile_frag_kc(const char* lds, int r0, int c, int lane) {
  using C = AttnC<T>;
  return ld16(lds + (r0 + (lane & 15)) * C::TS + c * 64 + (lane >> 4) * 16);
}
// fragment with the reduction along the tile's ROWS (operand rows = elements e0..e0+15, K-chunk kc of rows)
template <class T>
__device__ static __forceinline__ u32x4 tile_frag_ks(const char* lds, int e0, int kc, int lane) {
  using C = AttnC<T>;
  const int g = lane >> 4, i = lane & 15;
  u32x4 r;
  if constexpr (sizeof(T) == 2) {
#ifndef P5_NO_TR
    const char* base = lds + (kc * 32 + g * 8 + (i >> 2)) * C::TS + (e0 + (i & 3) * 4) * 2;
    u32x2 lo = lds_tr16_b64(base);
    u32x2 hi = lds_tr16_b64(base + 4 * C::TS);
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
#else
    const unsigned short* b = (const unsigned short*)(lds + (kc * 32 + g * 8) * C::TS + (e0 + i) * 2);
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b[j * (C::TS / 2)];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
#endif
  } else {
    const unsigned* b = (const unsigned*)(lds + (kc * 16 + g * 4) * C::TS + (e0 + i) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = b[j * (C::TS / 4)];
  }
  return r;
}


// stage per-head relative bias (index = k - q + Lq - 1) and the additive key mask into LDS
// (scale != 1, nfill > 0: the backward kernels that fold log2(e) into the exponent's terms -- every one of the first `nfill` entries is
//  written, zeros where there is no table or no such relative position, so that a score needs no "is there a bias" test)
__device__ static __forceinline__ void stage_bias_mask(const P5AttnArgs& a, int b, int h, float* sbias, float* skneg,
                                                       int nkeys_padded, int tid, float scale = 1.f, int nfill = 0) {
  const int nrel = a.Lq + a.Lk - 1;
  if (nfill > 0) {
    for (int i = tid; i < nfill; i += 256)
      sbias[i] = (a.rel_table && i < nrel) ? scale * a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h] : 0.f;
  } else if (a.rel_table) {
    for (int i = tid; i < nrel; i += 256) sbias[i] = a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h];
  }
  for (int j = tid; j < nkeys_padded; j += 256)
    skneg[j] = (j < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + j] != 0)) ? 0.f : P5_NEG_INF;
}


// d(relative-bias table), deterministic: `sv[i]` holds this workgroup's sum of dS over relative position i - (Lq - 1) (already in a
// fixed association).  Positions -> buckets: thread t < 64 owns bucket t and adds the positions that map to it in increasing order
// (it scans only [first, last] position of its bucket -- T5's bucket function is monotone on either side of 0, so that interval holds
// nothing else; integer LDS min/max give the same interval on every run), then STORES its sum into THIS workgroup's slot of the partial
// table (`rel_copies` buckets per head, zeros for buckets without positions): a slot belongs to exactly one workgroup of one launch --
// the caller gives every layer's launch its own block of slots -- so nothing is read, cleared or added on the way, and the store is in
// flight while the workgroup carries on.  The fixed-association reducer (p5_elem.h) sums the slots in index order.  No fp32 atomics:
// rounds 1-3 used LDS and global float atomics here, whose arrival order changed the last bits of the table's gradient from run to run.
// `scratch`: LDS that is dead by now -- 128 ints + nrel bytes (one wave) / 4 KiB (a whole workgroup: + NT / 64 partial sums per bucket).  Called by the NT threads of a workgroup (NT > 64: contains
// workgroup barriers) or by ONE wave (NT == 64: wave-level ordering only, the other waves of the workgroup carry on).
template <int NT>
__device__ static __forceinline__ void rel_bias_grad_flush(const P5AttnArgs& a, int h, int slot, const float* sv, void* scratch, int tid) {
#define P5_RB_SYNC() do { if constexpr (NT == 64) { P5_WAVE_SYNC(); } else { __syncthreads(); } } while (0)
  const int nrel = a.Lq + a.Lk - 1;
  int* slo = (int*)scratch;
  int* shi = slo + 64;
  unsigned char* sid = (unsigned char*)(shi + 64);
  if (tid < 64) { slo[tid] = 0x7fffffff; shi[tid] = -1; }
  P5_RB_SYNC();
  for (int i = tid; i < nrel; i += NT) {
    const int bk = a.bucket_lut[i - (a.Lq - 1) + a.lut_half] & 63;
    sid[i] = (unsigned char)bk;
    atomicMin(&slo[bk], i);
    atomicMax(&shi[bk], i);
  }
  P5_RB_SYNC();
  if constexpr (NT == 64) {
    if (tid < a.rel_copies) {          // (a bucket without positions stores 0: the slot is written in full, nobody clears it)
      float acc = 0.f;
      for (int i = slo[tid]; i <= shi[tid]; ++i) acc += (sid[i] == tid) ? sv[i] : 0.f;
      a.d_rel_table[(size_t)slot * a.rel_stride + tid * a.H + h] = acc;
    }
  } else {
    // NT / 64 threads per bucket: each adds a contiguous part of the bucket's interval in order, the parts are added in order
    // (at L = 512 the outermost bucket holds ~400 relative positions: one thread walking them alone took ~17 us per workgroup)
    static_assert(NT % 64 == 0 && NT / 64 <= 8, "rel_bias_grad_flush: at most eight threads per bucket");
    constexpr int NP = NT / 64;
    float* spart = (float*)scratch + 512;          // [NP][64] behind the interval bounds and the nrel <= 1023 bucket ids (scratch: >= 4 KiB = 2 KiB + 8 x 64 sums)
    const int bk = tid & 63, part = tid >> 6;
    float acc = 0.f;
    if (shi[bk] >= 0) {
      const int lo = slo[bk], len = shi[bk] - lo + 1, per = (len + NP - 1) / NP;
      const int i0 = lo + part * per, i1 = i0 + per < lo + len ? i0 + per : lo + len;
      for (int i = i0; i < i1; ++i) acc += (sid[i] == bk) ? sv[i] : 0.f;
    }
    spart[part * 64 + bk] = acc;
    __syncthreads();
    if (tid < a.rel_copies) {
      float t = spart[tid];
#pragma unroll
      for (int q = 1; q < NP; ++q) t += spart[q * 64 + tid];
      a.d_rel_table[(size_t)slot * a.rel_stride + tid * a.H + h] = t;
    }
  }
#undef P5_RB_SYNC
}

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
template <class T, int NKT>
__global__ __launch_bounds__(256) void p5_attn_fwd_kernel(P5AttnArgs a) {
  using C = AttnC<T>;
  __shared__ __attribute__((aligned(16))) char tile[64 * C::TS];
  __shared__ __attribute__((aligned(16))) char pbuf[4 * 16 * C::TS];
  __shared__ float sbias[1024];
  __shared__ __attribute__((aligned(16))) float skneg[NKT * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const T* Q = (const T*)a.Q; const T* K = (const T*)a.K; const T* V = (const T*)a.V;

  stage_bias_mask(a, b, h, sbias, skneg, NKT * 16, tid);

  u32x4 qf[C::NCK];
  {
    const int qrow = q0 + li;
#pragma unroll
    for (int c = 0; c < C::NCK; ++c)
      qf[c] = qrow < a.Lq ? ld16(Q + ((size_t)b * a.Lq + qrow) * a.ldq + h * 64 + c * C::KCH + g * C::EPF) : zero16();
  }
  // Scores are formed TRANSPOSED (keys along the accumulator rows): s[t][r] = score of key t*16 + g*4 + r for the lane's ONE
  // query q0 + li.  The softmax statistics are then per lane (reduced over r, t in registers and over the four g groups with
  // two shuffles), the mask is one 16-byte LDS read per 16 keys, and the four probabilities a lane owns are consecutive keys of
  // one row of P: one 8-byte LDS store instead of four 2-byte ones.  Everything below is select-based -- no per-lane branches
  // (the first version's `if (valid) { LDS read; ... }` per element serialised ~2 LDS round trips per score).
  f32x4 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int ch = 0; ch < NKT / 4; ++ch) {
    if (ch * 64 < a.Lk) {
      __syncthreads();
      stage_tile64<T>(tile, K + ((size_t)b * a.Lk + ch * 64) * a.ldk + h * 64, a.ldk, a.Lk - ch * 64, tid);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < C::NCK; ++c) mma16<T>(s[ch * 4 + t], tile_frag_kc<T>(tile, t * 16, c, lane), qf[c]);
    }
  }
  __syncthreads();  // sbias/skneg visible even when Lk == 0 chunks were skipped

  // ---- exact softmax over the register-resident score rows ----
  const int qi = q0 + li;
  const bool qok = qi < a.Lq;
  const int qic = qok ? qi : a.Lq - 1;          // (keeps the bias index of a padding row inside the table)
  const bool causal = a.causal != 0;
  float m = P5_NEG_INF;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kb = t * 16 + g * 4;
    const f32x4 kn = *(const f32x4*)(skneg + kb);
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.rel_table) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[r] = sbias[kb + r - qic + a.Lq - 1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kj = kb + r;
      const bool ok = (kj < a.Lk) & qok & !(causal & (kj > qi));
      const float v = ok ? (s[t][r] + kn[r]) + bias[r] : P5_NEG_INF;
      s[t][r] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  if (m == P5_NEG_INF) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = p5_exp<T>(s[t][r] - m);
      s[t][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (g == 0 && qok && a.lse) a.lse[((size_t)b * a.H + h) * a.Lq + qi] = m + logf(l);
  if (a.drop.state != nullptr && a.drop.thr != 0) {
    const uint32_t seed = p5_seed(a.drop);
    const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t idx = rowbase + (uint32_t)(t * 16 + g * 4 + r);
        s[t][r] = p5_keep(seed, a.drop.site_key, idx, a.drop.thr) ? s[t][r] * a.drop.scale : 0.f;
      }
  }

  // ---- O = P V ----
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  char* pw = pbuf + wave * 16 * C::TS;
#pragma unroll
  for (int ch = 0; ch < NKT / 4; ++ch) {
    if (ch * 64 < a.Lk) {
      __syncthreads();
      stage_tile64<T>(tile, V + ((size_t)b * a.Lk + ch * 64) * a.ldv + h * 64, a.ldv, a.Lk - ch * 64, tid);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v4[4] = {s[ch * 4 + t][0], s[ch * 4 + t][1], s[ch * 4 + t][2], s[ch * 4 + t][3]};
        st4<T>(pw + li * C::TS + (t * 16 + g * 4) * C::SZ, v4);
      }
      __syncthreads();
#pragma unroll
      for (int kc = 0; kc < C::NCK; ++kc) {
        const u32x4 pa = ld16(pw + li * C::TS + kc * 64 + g * 16);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma16<T>(o[dt], pa, tile_frag_ks<T>(tile, dt * 16, kc, lane));
      }
    }
  }
  // the output tile is in the usual layout (lane owns queries g*4 + r): fetch those rows' 1/l from the lanes that hold them
  const float inv_q = l > 0.f ? 1.f / l : 0.f;
  float inv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv[r] = __shfl(inv_q, g * 4 + r);
  wave_store_16x64<T>((T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64, a.ldo, q0, a.Lq, o, inv, pw, lane);
}

// ------------------------------------------------------------------------------------------------------------
// forward, one workgroup per (batch, head) (bf16, Lq and Lk <= 128).  The kernel above is ~60 % operand fetch at the benchmark
// shape (its loads alone take 10-14 us of its 19 us) and reads K and V once per 64-query block; the decoder's
// 8-query calls are chains of round trips (bias table, K tile, V tile, each behind a barrier).  Here NW waves (8 for
// 64 < Lq <= 128, 4 below) share ONE copy of the whole K and V of the head, every global load of the workgroup is issued before
// the first use, and after the single barrier each wave runs on its own: scores for its 16 queries against all keys, the exact
// softmax, P through a wave-private tile, O = P V.  Same arithmetic, same order per row as p5_attn_fwd_kernel<T, 8>.
// ------------------------------------------------------------------------------------------------------------
template <class T, int NW>
__global__ __launch_bounds__(NW * 64) void p5_attn_fwd_wg_kernel(P5AttnArgs a) {
  static_assert(sizeof(T) == 2, "whole-head attention forward: bf16 only");
  using C = AttnC<T>;
  constexpr int NT = NW * 64, NKT = 8, NP = 128 * C::PPR / NT;
  __shared__ __attribute__((aligned(16))) char tK[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char tV[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char pbuf[NW * 16 * C::TS];
  __shared__ float sbias[256];
  __shared__ __attribute__((aligned(16))) float skneg[NKT * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int q0 = wave * 16, qi = q0 + li;
  const bool qok = qi < a.Lq;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;

  // ---- every global load of the workgroup, then one barrier ----
  u32x4 qf[C::NCK], rk[NP], rv[NP];
#pragma unroll
  for (int c = 0; c < C::NCK; ++c) qf[c] = qok ? ld16(Q + (size_t)qi * a.ldq + c * C::KCH + g * C::EPF) : zero16();
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + i * NT, row = p / C::PPR, pc = p % C::PPR;
    rk[i] = row < a.Lk ? ld16(K + (size_t)row * a.ldk + pc * C::EPF) : zero16();
    rv[i] = row < a.Lk ? ld16(V + (size_t)row * a.ldv + pc * C::EPF) : zero16();
  }
  stage_bias_mask(a, b, h, sbias, skneg, NKT * 16, tid);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + i * NT, row = p / C::PPR, pc = p % C::PPR;
    st16(tK + row * C::TS + pc * 16, rk[i]);
    st16(tV + row * C::TS + pc * 16, rv[i]);
  }
  __syncthreads();
  if (q0 >= a.Lq) return;            // (no barrier below: a wave without queries is done once the tiles are staged)

  // ---- scores (transposed: keys along the accumulator rows, see p5_attn_fwd_kernel) ----
  const int nkt = (a.Lk + 15) / 16;
  f32x4 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < nkt) {
#pragma unroll
      for (int c = 0; c < C::NCK; ++c) mma16<T>(s[t], tile_frag_kc<T>(tK, t * 16, c, lane), qf[c]);
    }
  }
  const int qic = qok ? qi : a.Lq - 1;
  const bool causal = a.causal != 0;
  float m = P5_NEG_INF;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kb = t * 16 + g * 4;
    const f32x4 kn = *(const f32x4*)(skneg + kb);
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.rel_table) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[r] = sbias[(kb + r - qic + a.Lq - 1) & 255];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kj = kb + r;
      const bool ok = (kj < a.Lk) & qok & !(causal & (kj > qi));
      const float v = ok ? (s[t][r] + kn[r]) + bias[r] : P5_NEG_INF;
      s[t][r] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  if (m == P5_NEG_INF) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = p5_exp<T>(s[t][r] - m);
      s[t][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (g == 0 && qok && a.lse) a.lse[((size_t)b * a.H + h) * a.Lq + qi] = m + logf(l);
  if (a.drop.state != nullptr && a.drop.thr != 0) {
    const uint32_t seed = p5_seed(a.drop);
    const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t idx = rowbase + (uint32_t)(t * 16 + g * 4 + r);
        s[t][r] = p5_keep(seed, a.drop.site_key, idx, a.drop.thr) ? s[t][r] * a.drop.scale : 0.f;
      }
  }

  // ---- O = P V: P goes through the wave's own tile, 64 keys at a time; V is already resident ----
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  char* pw = pbuf + wave * 16 * C::TS;
#pragma unroll
  for (int ch = 0; ch < NKT / 4; ++ch) {
    if (ch * 64 < a.Lk) {
      P5_WAVE_SYNC();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v4[4] = {s[ch * 4 + t][0], s[ch * 4 + t][1], s[ch * 4 + t][2], s[ch * 4 + t][3]};
        st4<T>(pw + li * C::TS + (t * 16 + g * 4) * C::SZ, v4);
      }
      P5_WAVE_SYNC();
#pragma unroll
      for (int kc = 0; kc < C::NCK; ++kc) {
        const u32x4 pa = ld16(pw + li * C::TS + kc * 64 + g * 16);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma16<T>(o[dt], pa, tile_frag_ks<T>(tV + ch * 64 * C::TS, dt * 16, kc, lane));
      }
    }
  }
  const float inv_q = l > 0.f ? 1.f / l : 0.f;
  float inv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv[r] = __shfl(inv_q, g * 4 + r);
  wave_store_16x64<T>((T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64, a.ldo, q0, a.Lq, o, inv, pw, lane);
}

// ------------------------------------------------------------------------------------------------------------
// forward, one workgroup per (batch, head) for 128 < Lk <= 512 (bf16; C5: T5-large at L = 512).  p5_attn_fwd_kernel<T, 32> streams
// K and V in 64-key tiles per 64-query block, every tile a global round trip behind a barrier with 8 MFMAs of work per wave in
// between: 76 TF/s at B=64, H=16, L=512 (profiles/r04_c5_t5large_l512_step_kernels.md).  Here the whole K and V of the head are
// resident in LDS (2 x 64 KiB at Lk = 512: rows of 128 bytes, no padding, 16-byte pieces XOR-swizzled by the row so that the
// fragment reads below stay conflict-free), fetched ONCE per head with every load in flight before the single barrier; the 8 waves
// then take the 16-query blocks round-robin with no further workgroup barrier: scores for all keys in registers, the exact two-pass
// softmax of p5_attn_fwd_kernel (same operations per score), and O = P V with P handed to the MFMA straight from the score
// registers -- a lane's eight probabilities of a 32-key chunk (keys 32u + 4g + r and 32u + 16 + 4g + r) become the k-slots 8g..8g+7
// of the A operand, and the V fragment is read (ds_read_b64_tr_b16) from the rows of exactly those keys, so P never goes through LDS.
//   K image: piece' = piece ^ ((row >> 1) & 7)      (ds_read_b128 of 16 consecutive rows, one piece: 16 distinct 16-byte slots)
//   V image: piece' = piece ^ (((row >> 1) & 3) << 1)   (transposed 8-byte reads of 8 consecutive rows x 32 bytes)
// ------------------------------------------------------------------------------------------------------------
template <int NKT, bool BITS>          // BITS: the dropout keep decisions also go out as lane masks (P5AttnArgs::keep_bits != nullptr)
__global__ __launch_bounds__(512) void p5_attn_fwd_head_kernel(P5AttnArgs a) {
  using T = bf16;
  using C = AttnC<T>;
  constexpr int LK = NKT * 16, NT = 512, NP = LK * 8 / NT;
  static_assert(NKT == 16 || NKT == 32, "whole-head attention forward: 256 or 512 key slots");
  __shared__ __attribute__((aligned(16))) char tK[LK * 128];
  __shared__ __attribute__((aligned(16))) char tV[LK * 128];
  __shared__ __attribute__((aligned(16))) char pbuf[8 * 16 * C::TS];        // wave_store_16x64's staging rows, one block per wave
  __shared__ float sbias[1024];
  __shared__ __attribute__((aligned(16))) float skneg[LK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;

  // ---- every global load of the workgroup, then one barrier ----
  {
    u32x4 rk[NP], rv[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rk[i] = row < a.Lk ? ld16(K + (size_t)row * a.ldk + pc * 8) : zero16();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rv[i] = row < a.Lk ? ld16(V + (size_t)row * a.ldv + pc * 8) : zero16();      // (rows past Lk: zeros, 0 x garbage would be NaN)
    }
    const int nrel = a.Lq + a.Lk - 1;
    // (no table -- cross-attention: zeros, the softmax pass below reads the bias unconditionally, also for the key slots past Lk of the
    // last block: positions up to LK + Lq - 2 are read, (-inf) + an uninitialised NaN would poison the row maximum)
    for (int i = tid; i < LK + a.Lq - 1; i += NT)
      sbias[i] = (a.rel_table && i < nrel) ? a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h] : 0.f;
    for (int j = tid; j < LK; j += NT) skneg[j] = (j < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + j] != 0)) ? 0.f : P5_NEG_INF;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      st16(tK + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4), rk[i]);
      st16(tV + row * 128 + ((pc ^ (((row >> 1) & 3) << 1)) << 4), rv[i]);
    }
  }
  u32x4 qn[2];
  {
    const int qrow = wave * 16 + li;
#pragma unroll
    for (int c = 0; c < 2; ++c) qn[c] = qrow < a.Lq ? ld16(Q + (size_t)qrow * a.ldq + c * 32 + g * 8) : zero16();
  }
  __syncthreads();

  const int nkt = (a.Lk + 15) / 16;
  const bool causal = a.causal != 0;
  const int koff0 = li * 128 + (((0 + g) ^ ((li >> 1) & 7)) << 4), koff1 = li * 128 + (((4 + g) ^ ((li >> 1) & 7)) << 4);
  const int vrow = g * 4 + (li >> 2), vsw = ((g * 2 + (li >> 3)) & 3) << 1;
  char* pw = pbuf + wave * 16 * C::TS;
  // BITS: a block's masks are stored one block later, after the next block's score MFMAs (and after the loop for the last one), so that
  // a long vector-only section follows the stores.  HARDWARE NOTE (profiles/r05_call21_keep_mask_bisect.txt): the first versions of this
  // variant returned 1e38 / NaN in the first 16 output columns of some query blocks -- the first P V MFMA of a block had multiplied by
  // the registers' PREVIOUS contents (ballot words) instead of the transposed V reads issued just before it, behind a partial
  // `s_waitcnt lgkmcnt(2)`; different from run to run, only on the hardware, only in this variant (whose dropout pass leaves SGPR-sourced
  // selects right in front of the reads).  Neither moving the stores nor collecting the masks differently changed it; waiting for ALL
  // eight transposed reads of a chunk before its four MFMAs (P5_WAIT_LGKM0 between scheduling fences, below) did: 0 differing elements
  // over repeated runs at every grid size (tests: attn_keep_masks_forward_case).
  uint32_t keep_lo[NKT / 16] = {}, keep_hi[NKT / 16] = {};
  uint32_t* keep_dst = nullptr;
  for (int q0 = wave * 16; q0 < a.Lq; q0 += 128) {
    const u32x4 qf0 = qn[0], qf1 = qn[1];
    if (q0 + 128 < a.Lq) {           // the next block's Q fragment is in flight under this block's work
      const int qrow = q0 + 128 + li;
#pragma unroll
      for (int c = 0; c < 2; ++c) qn[c] = qrow < a.Lq ? ld16(Q + (size_t)qrow * a.ldq + c * 32 + g * 8) : zero16();
    }
    const int qi = q0 + li;
    const bool qok = qi < a.Lq;
    const int qic = qok ? qi : a.Lq - 1;

    // ---- scores (transposed: keys along the accumulator rows, see p5_attn_fwd_kernel) ----
    f32x4 s[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t < nkt) {
        mma16<T>(s[t], ld16(tK + t * 2048 + koff0), qf0);
        mma16<T>(s[t], ld16(tK + t * 2048 + koff1), qf1);
      }
      if ((t & 3) == 3) P5_SCHED_FENCE();       // (keeps the fragment reads of at most four key blocks in flight: registers)
    }
    if constexpr (BITS) {
      if (keep_dst) {
#pragma unroll
        for (int i = 0; i < NKT / 16; ++i) { keep_dst[i * 64 + lane] = keep_lo[i]; keep_dst[128 + i * 64 + lane] = keep_hi[i]; }
      }
    }
    if (causal) {
#pragma unroll
      for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + g * 4 + r > qi) ? P5_NEG_INF : s[t][r];
    }
    // (key slots past Lk and masked keys carry -inf in skneg, rows past Lq are never stored: no per-element validity select;
    // a valid score goes through exactly the operations of p5_attn_fwd_kernel: (s + mask) + bias)
    float m = P5_NEG_INF;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int kb = t * 16 + g * 4;
      const f32x4 kn = *(const f32x4*)(skneg + kb);
      const float* pb = sbias + (kb - qic + a.Lq - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (s[t][r] + kn[r]) + pb[r];
        s[t][r] = v;
        m = fmaxf(m, v);
      }
      if (t & 1) P5_SCHED_FENCE();             // (bias / mask reads of two key blocks at a time, not of all of them)
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    if (m == P5_NEG_INF) m = 0.f;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = p5_exp<T>(s[t][r] - m);
        s[t][r] = p;
        l += p;
      }
      if ((t & 3) == 3) P5_SCHED_FENCE();
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (g == 0 && qok && a.lse) a.lse[((size_t)b * a.H + h) * a.Lq + qi] = m + logf(l);
    if (a.drop.state != nullptr && a.drop.thr != 0) {
      const uint32_t seed = p5_seed(a.drop);
      const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);
      if constexpr (BITS) {
        // the keep decisions also go out as lane masks for the backward (P5AttnArgs::keep_bits): lane n keeps the mask of (t, r) = n
        uint32_t mlo[NKT / 16], mhi[NKT / 16];
#pragma unroll
        for (int i = 0; i < NKT / 16; ++i) { mlo[i] = 0u; mhi[i] = 0u; }
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t idx = rowbase + (uint32_t)(t * 16 + g * 4 + r);
            const bool keep = p5_keep(seed, a.drop.site_key, idx, a.drop.thr);
            s[t][r] = keep ? s[t][r] * a.drop.scale : 0.f;
            const unsigned long long bal = __ballot(keep);
            const int n = t * 4 + r;
            if (lane == (n & 63)) { mlo[n >> 6] = (uint32_t)bal; mhi[n >> 6] = (uint32_t)(bal >> 32); }
          }
          P5_SCHED_FENCE();
        }
#pragma unroll
        for (int i = 0; i < NKT / 16; ++i) { keep_lo[i] = mlo[i]; keep_hi[i] = mhi[i]; }
        keep_dst = a.keep_bits + (((size_t)b * a.H + h) * ((a.Lq + 15) / 16) + (q0 >> 4)) * 256;
      } else {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t idx = rowbase + (uint32_t)(t * 16 + g * 4 + r);
          s[t][r] = p5_keep(seed, a.drop.site_key, idx, a.drop.thr) ? s[t][r] * a.drop.scale : 0.f;
        }
        P5_SCHED_FENCE();      // (four hash chains interleaved, not all 128: registers)
      }
      }
    }

    // ---- O = P V, P from the score registers (32 keys per MFMA: blocks 2u and 2u + 1) ----
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      if (u * 32 < a.Lk) {
        const float p8[8] = {s[2 * u][0], s[2 * u][1], s[2 * u][2], s[2 * u][3], s[2 * u + 1][0], s[2 * u + 1][1], s[2 * u + 1][2], s[2 * u + 1][3]};
        const u32x4 pa = pack16<T>(p8);
        u32x4 vb[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const char* base = tV + (u * 32 + vrow) * 128 + (((dt * 2 + ((li >> 1) & 1)) ^ vsw) << 4) + (li & 1) * 8;
          const u32x2 lo = lds_tr16_b64(base);
          const u32x2 hi = lds_tr16_b64(base + 16 * 128);
          vb[dt][0] = lo[0]; vb[dt][1] = lo[1]; vb[dt][2] = hi[0]; vb[dt][3] = hi[1];
        }
        P5_SCHED_FENCE();
        P5_WAIT_LGKM0();          // (all eight transposed reads have landed before the first MFMA: HARDWARE NOTE above)
        P5_SCHED_FENCE();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma16<T>(o[dt], pa, vb[dt]);
      }
      P5_SCHED_FENCE();
    }
    const float inv_q = l > 0.f ? 1.f / l : 0.f;
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) inv[r] = __shfl(inv_q, g * 4 + r);
    wave_store_16x64<T>((T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64, a.ldo, q0, a.Lq, o, inv, pw, lane);
  }
  if constexpr (BITS) {
    if (keep_dst) {
#pragma unroll
      for (int i = 0; i < NKT / 16; ++i) { keep_dst[i * 64 + lane] = keep_lo[i]; keep_dst[128 + i * 64 + lane] = keep_hi[i]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, part 1: dQ (+ d rel-bias table, + D = rowsum(dO*O) for part 2).  One wave = 16 queries.
// ------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void p5_attn_bwd_dq_kernel(P5AttnArgs a) {
  using C = AttnC<T>;
  __shared__ __attribute__((aligned(16))) char tileK[64 * C::TS];
  __shared__ __attribute__((aligned(16))) char tileV[64 * C::TS];
  __shared__ __attribute__((aligned(16))) char pbuf[4 * 16 * C::TS];
  __shared__ __attribute__((aligned(16))) float sbias[1024];
  __shared__ float sdb[4][1024];          // per-wave sums of dS along the diagonals (relative positions)
  __shared__ __attribute__((aligned(16))) float skneg[512];
  // static LDS: 3 tiles of [64][TS] + 4 KiB bias + 16 KiB diagonal sums + 2 KiB mask = 49.0 KiB in bf16 (3 workgroups per CU), 73.0 KiB
  // in fp32 (2 per CU; above the 64 KiB of older parts: gfx950 only).  The 16 KiB of sums are sized for L = 512; bf16 calls with L > 128
  // take the head-resident kernels below (sums in registers), so this kernel's long-sequence caller is the fp32 parity engine.
  static_assert(3 * 64 * C::TS + 4096 + 16384 + 2048 <= 80 * 1024, "p5_attn_bwd_dq_kernel: two workgroups per CU in fp32");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const T* Q = (const T*)a.Q; const T* K = (const T*)a.K; const T* V = (const T*)a.V;
  const T* dO = (const T*)a.dO; const T* O = (const T*)a.O;
  const int nrel = a.Lq + a.Lk - 1;
  const int nch = (a.Lk + 63) / 64;

  stage_bias_mask(a, b, h, sbias, skneg, nch * 64, tid);
  if (a.d_rel_table)
    for (int i = tid; i < 4 * 1024; i += 256) (&sdb[0][0])[i] = 0.f;

  u32x4 qf[C::NCK], dof[C::NCK];
  float Drow = 0.f;
  {
    const int qrow = q0 + li;
#pragma unroll
    for (int c = 0; c < C::NCK; ++c) {
      const size_t col = h * 64 + c * C::KCH + g * C::EPF;
      if (qrow < a.Lq) {
        qf[c] = ld16(Q + ((size_t)b * a.Lq + qrow) * a.ldq + col);
        dof[c] = ld16(dO + ((size_t)b * a.Lq + qrow) * a.lddo + col);
        const u32x4 of = ld16(O + ((size_t)b * a.Lq + qrow) * a.ldo + col);
        float x[8], y[8];
        unpack16<T>(dof[c], x);
        unpack16<T>(of, y);
#pragma unroll
        for (int e = 0; e < C::EPF; ++e) Drow += x[e] * y[e];
      } else {
        qf[c] = zero16();
        dof[c] = zero16();
      }
    }
    Drow += __shfl_xor(Drow, 16);
    Drow += __shfl_xor(Drow, 32);
    if (g == 0 && qrow < a.Lq) a.Dvec[((size_t)b * a.H + h) * a.Lq + qrow] = Drow;
  }
  // scores transposed as in the forward kernel: the lane owns ONE query (q0 + li) and keys t*16 + g*4 + r
  const int qi = q0 + li;
  const bool qok = qi < a.Lq;
  const int qic = qok ? qi : a.Lq - 1;
  const float lse_q = qok ? a.lse[((size_t)b * a.H + h) * a.Lq + qi] : 0.f;
  const float D_q = Drow;          // (all four g lanes of a query hold the reduced sum)
  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);
  const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);

  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  char* pw = pbuf + wave * 16 * C::TS;

  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();
    stage_tile64<T>(tileK, K + ((size_t)b * a.Lk + ch * 64) * a.ldk + h * 64, a.ldk, a.Lk - ch * 64, tid);
    stage_tile64<T>(tileV, V + ((size_t)b * a.Lk + ch * 64) * a.ldv + h * 64, a.ldv, a.Lk - ch * 64, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C::NCK; ++c) {
        mma16<T>(sacc, tile_frag_kc<T>(tileK, t * 16, c, lane), qf[c]);
        mma16<T>(dpacc, tile_frag_kc<T>(tileV, t * 16, c, lane), dof[c]);
      }
      const int kb = ch * 64 + t * 16 + g * 4;
      const f32x4 kn = *(const f32x4*)(skneg + kb);
      float bias[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.rel_table) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = sbias[kb + r - qic + a.Lq - 1];
      }
      float mk[4] = {1.f, 1.f, 1.f, 1.f};
      if (do_drop) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mk[r] = p5_keep(seed, a.drop.site_key, rowbase + (uint32_t)(kb + r), a.drop.thr) ? a.drop.scale : 0.f;
      }
      float dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = kb + r;
        const bool ok = (kj < a.Lk) & qok & !(causal & (kj > qi)) & (kn[r] == 0.f);     // (bitwise: no short-circuit branches)
        const float p = p5_exp<T>((sacc[r] + bias[r]) - lse_q);
        dsv[r] = ok ? p * (dpacc[r] * mk[r] - D_q) : 0.f;
      }
      st4<T>(pw + li * C::TS + (t * 16 + g * 4) * C::SZ, dsv);
    }
    __syncthreads();
    if (a.d_rel_table) {
      // d(rel-bias): sum dS along the diagonals (constant key - query) of this wave's [16 q][64 keys] tile held in LDS, into
      // the wave's OWN row of sums (plain read-modify-write: the lanes of a wave hold different diagonals, its key chunks come
      // one after the other) -- no LDS atomics, same association every run
      for (int dd = lane; dd < 79; dd += 64) {
        float sum = 0.f;
#pragma unroll
        for (int qr = 0; qr < 16; ++qr) {
          const int kcol = dd - 15 + qr;
          if (kcol >= 0 && kcol < 64) sum += to_f<T>(*(const T*)(pw + qr * C::TS + kcol * C::SZ));
        }
        const int idx = ch * 64 + dd - 15 - q0 + a.Lq - 1;
        if (idx >= 0 && idx < nrel) sdb[wave][idx] += sum;
      }
    }
#pragma unroll
    for (int kc = 0; kc < C::NCK; ++kc) {
      const u32x4 dsa = ld16(pw + li * C::TS + kc * 64 + g * 16);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma16<T>(dq[dt], dsa, tile_frag_ks<T>(tileK, dt * 16, kc, lane));
    }
  }
  {
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    wave_store_16x64<T>((T*)a.dQ + (size_t)b * a.Lq * a.lddq + h * 64, a.lddq, q0, a.Lq, dq, one, pw, lane);
  }
  if (a.d_rel_table) {
    __syncthreads();
    for (int i = tid; i < nrel; i += 256) sdb[0][i] = ((sdb[0][i] + sdb[1][i]) + sdb[2][i]) + sdb[3][i];      // (waves in index order)
    __syncthreads();
    rel_bias_grad_flush<256>(a, h, b * (int)gridDim.x + (int)blockIdx.x, &sdb[0][0], sbias, tid);      // (sbias is dead from here on)
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, part 2: dK, dV.  One wave = 16 keys, streams 64-query tiles.  Needs lse and Dvec.
// ------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void p5_attn_bwd_dkv_kernel(P5AttnArgs a) {
  using C = AttnC<T>;
  __shared__ __attribute__((aligned(16))) char tileQ[64 * C::TS];
  __shared__ __attribute__((aligned(16))) char tileDO[64 * C::TS];
  __shared__ __attribute__((aligned(16))) char pbufP[4 * 16 * C::TS];
  __shared__ __attribute__((aligned(16))) char pbufS[4 * 16 * C::TS];
  __shared__ float sbias[1024];
  __shared__ __attribute__((aligned(16))) float slse[64];
  __shared__ __attribute__((aligned(16))) float sD[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const T* Q = (const T*)a.Q; const T* K = (const T*)a.K; const T* V = (const T*)a.V;
  const T* dO = (const T*)a.dO;
  const int nrel = a.Lq + a.Lk - 1;
  if (a.rel_table)
    for (int i = tid; i < nrel; i += 256) sbias[i] = a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h];

  u32x4 kf[C::NCK], vf[C::NCK];
  {
    const int krow = k0 + li;
#pragma unroll
    for (int c = 0; c < C::NCK; ++c) {
      const size_t col = h * 64 + c * C::KCH + g * C::EPF;
      kf[c] = krow < a.Lk ? ld16(K + ((size_t)b * a.Lk + krow) * a.ldk + col) : zero16();
      vf[c] = krow < a.Lk ? ld16(V + ((size_t)b * a.Lk + krow) * a.ldv + col) : zero16();
    }
  }
  // scores transposed (queries along the accumulator rows): the lane owns ONE key (k0 + li) and queries t*16 + g*4 + r of the
  // staged tile, so its four P / dS values are consecutive queries of one key row: 8-byte LDS stores, per-query statistics in
  // one 16-byte read each, no per-lane branches
  const int kj = k0 + li;
  const bool kok = kj < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + kj] != 0);
  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);
  const uint32_t headbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq) * a.Lk) + (uint32_t)kj;

  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  char* pP = pbufP + wave * 16 * C::TS;
  char* pS = pbufS + wave * 16 * C::TS;
  const int nqc = (a.Lq + 63) / 64;

  for (int qc = 0; qc < nqc; ++qc) {
    __syncthreads();
    stage_tile64<T>(tileQ, Q + ((size_t)b * a.Lq + qc * 64) * a.ldq + h * 64, a.ldq, a.Lq - qc * 64, tid);
    stage_tile64<T>(tileDO, dO + ((size_t)b * a.Lq + qc * 64) * a.lddo + h * 64, a.lddo, a.Lq - qc * 64, tid);
    if (tid < 64) {
      const int qi = qc * 64 + tid;
      slse[tid] = qi < a.Lq ? a.lse[((size_t)b * a.H + h) * a.Lq + qi] : 0.f;
      sD[tid] = qi < a.Lq ? a.Dvec[((size_t)b * a.H + h) * a.Lq + qi] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C::NCK; ++c) {
        mma16<T>(sacc, tile_frag_kc<T>(tileQ, t * 16, c, lane), kf[c]);
        mma16<T>(dpacc, tile_frag_kc<T>(tileDO, t * 16, c, lane), vf[c]);
      }
      const int qb = t * 16 + g * 4;
      const f32x4 ls = *(const f32x4*)(slse + qb), dd = *(const f32x4*)(sD + qb);
      float bias[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.rel_table) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qc * 64 + qb + r;
          bias[r] = sbias[kj - (qi < a.Lq ? qi : a.Lq - 1) + a.Lq - 1];
        }
      }
      const uint32_t tbase = headbase + (uint32_t)(qc * 64 + qb) * (uint32_t)a.Lk;
      float mk[4] = {1.f, 1.f, 1.f, 1.f};
      if (do_drop) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mk[r] = p5_keep(seed, a.drop.site_key, tbase + (uint32_t)r * (uint32_t)a.Lk, a.drop.thr) ? a.drop.scale : 0.f;
      }
      float pv[4], dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qc * 64 + qb + r;
        const bool ok = kok & (qi < a.Lq) & !(causal & (kj > qi));
        const float p = p5_exp<T>((sacc[r] + bias[r]) - ls[r]);
        pv[r] = ok ? p * mk[r] : 0.f;
        dsv[r] = ok ? p * (dpacc[r] * mk[r] - dd[r]) : 0.f;
      }
      st4<T>(pP + li * C::TS + (t * 16 + g * 4) * C::SZ, pv);
      st4<T>(pS + li * C::TS + (t * 16 + g * 4) * C::SZ, dsv);
    }
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < C::NCK; ++kc) {
      const u32x4 pa = ld16(pP + li * C::TS + kc * 64 + g * 16);
      const u32x4 sa = ld16(pS + li * C::TS + kc * 64 + g * 16);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        mma16<T>(dv[dt], pa, tile_frag_ks<T>(tileDO, dt * 16, kc, lane));
        mma16<T>(dk[dt], sa, tile_frag_ks<T>(tileQ, dt * 16, kc, lane));
      }
    }
  }
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  wave_store_16x64<T>((T*)a.dK + (size_t)b * a.Lk * a.lddk + h * 64, a.lddk, k0, a.Lk, dk, one, pP, lane);
  wave_store_16x64<T>((T*)a.dV + (size_t)b * a.Lk * a.lddv + h * 64, a.lddv, k0, a.Lk, dv, one, pS, lane);
}

// ------------------------------------------------------------------------------------------------------------
// backward for 128 < L <= 512 (bf16; C5), the operands that every block re-reads resident in LDS as in p5_attn_fwd_head_kernel (rows of
// 128 bytes, pieces XOR-swizzled by (row >> 1) & 7: the 16-row fragment reads are conflict-free, the transposed reads two-way).  The two
// kernels above fetch their K / V (Q / dO) tiles per 64-row block behind two barriers each: 28 ms of the 144 ms C5 step
// (profiles/r04_c5_t5large_l512_step_kernels.md).  Same element arithmetic as those kernels (P recomputed from the saved log-sum-exp);
// dS (and P) reach the MFMA from the registers that computed them -- keys (queries) 32u + 4g + r and 32u + 16 + 4g + r as the k-slots
// 8g .. 8g+7, the other operand's rows read in that order -- instead of through an LDS tile.
//   part 1, dQ (+ D, + d rel-bias): K and V of the head resident; eight waves take the 16-query blocks round-robin.  The sums of dS
//   along the diagonals (relative positions) stay in REGISTERS: relative position + 16 = 64 J + lane, and a 64-key chunk `ch` of the
//   wave's i-th query block touches J = ch - 2 i + c, c wave-constant, J + 1 and J + 2 -- with the chunk loop unrolled and the 18
//   accumulators rotated by two per query block those are the fixed registers R[ch], R[ch + 1], R[ch + 2] (four 4-KiB rows of sums per
//   workgroup, as p5_attn_bwd_dq_kernel keeps them, would not fit beside K and V; with them the kernel was limited to four waves and ran
//   at 1.27 ms where the blocked kernel takes 0.74: profiles/r05_call14_attention_head_resident.txt).  The per-wave sums meet in LDS
//   after the last block (K and V are dead by then) and are added in wave order: ONE relative-bias slot per (batch, head).
// ------------------------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(512) void p5_attn_bwd_dq_head_kernel(P5AttnArgs a) {
  using T = bf16;
  using C = AttnC<T>;
  constexpr int LK = NKT * 16, NT = 512, NP = LK * 8 / NT, NCH = NKT / 4;
  static_assert(NKT == 16 || NKT == 32, "whole-head attention backward: 256 or 512 key slots");
  static_assert(2 * LK * 128 >= 8 * 1024 * 4, "the per-wave rows of diagonal sums reuse the K and V images");
  __shared__ __attribute__((aligned(16))) char tKV[2 * LK * 128];
  // the wave's [16 queries][64 keys] tile of dS, for the diagonal sums of the relative-bias gradient: rows of DSTR bytes with 16 zero
  // columns in front of the 64 keys and 15 behind them, so that a diagonal's 16 elements are 16 reads at constant offsets from one lane
  // address with no clamp and no select (round 6: the clamped / selected form cost 14 of this pass's 23 vector instructions per score).
  // The same bytes are the staging tile of the dQ store, which overwrites the zero columns: they are cleared again per query block.
  constexpr int DSTR = 192, DPAD = 16;
  __shared__ __attribute__((aligned(16))) char pbuf[8 * 16 * DSTR];
  __shared__ __attribute__((aligned(16))) float sbias[1024];
  __shared__ __attribute__((aligned(16))) float skneg[LK];
  char* tK = tKV;
  char* tV = tKV + LK * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;
  const T* dO = (const T*)a.dO + (size_t)b * a.Lq * a.lddo + h * 64;
  const T* O = (const T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64;
  const int nrel = a.Lq + a.Lk - 1;
  const int nch = (a.Lk + 63) / 64;
  {
    u32x4 rk[NP], rv[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rk[i] = row < a.Lk ? ld16(K + (size_t)row * a.ldk + pc * 8) : zero16();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rv[i] = row < a.Lk ? ld16(V + (size_t)row * a.ldv + pc * 8) : zero16();
    }
    // (round 6, as in the dK / dV pass below: log2(e) folded into the bias table and the row's log-sum-exp, the key mask ADDED to the
    //  exponent as 0 / -inf -- a probability is two adds, one fma and one v_exp_f32, no compare, no branch)
    for (int i = tid; i < LK + a.Lq - 1; i += NT)       // (every position the blocks read, key slots past Lk included)
      sbias[i] = (a.rel_table && i < nrel) ? P5_LOG2E * a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h] : 0.f;
    for (int j = tid; j < LK; j += NT) skneg[j] = (j < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + j] != 0)) ? 0.f : P5_NEG_INF;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      st16(tK + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4), rk[i]);
      st16(tV + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4), rv[i]);
    }
  }
  // the first block's row operands; the next block's are fetched under the current block's work
  u32x4 qn[2], don[2], on[2];
  float lse_n = 0.f;
  {
    const int qrow = wave * 16 + li;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      qn[c] = qrow < a.Lq ? ld16(Q + (size_t)qrow * a.ldq + c * 32 + g * 8) : zero16();
      don[c] = qrow < a.Lq ? ld16(dO + (size_t)qrow * a.lddo + c * 32 + g * 8) : zero16();
      on[c] = qrow < a.Lq ? ld16(O + (size_t)qrow * a.ldo + c * 32 + g * 8) : zero16();
    }
    lse_n = qrow < a.Lq ? a.lse[((size_t)b * a.H + h) * a.Lq + qrow] : 0.f;
  }
  __syncthreads();

  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const bool do_rel = a.d_rel_table != nullptr;
  const bool use_bits = do_drop && a.keep_bits != nullptr;
  const uint32_t seed = p5_seed(a.drop);
  const int ksw = (li >> 1) & 7;
  const int koff0 = li * 128 + (((0 + g) ^ ksw) << 4), koff1 = li * 128 + (((4 + g) ^ ksw) << 4);
  const int trow = g * 4 + (li >> 2), tsw = (g * 2 + (li >> 3)) & 7;
  char* pw = pbuf + wave * 16 * DSTR;
  static_assert(16 * DSTR >= 16 * C::TS && (DPAD + 64 + 15) * 2 <= DSTR, "diagonal tile: store staging fits, zero columns fit");
  auto clear_diag_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 16 * DSTR / (64 * 16); ++i) st16(pw + (i * 64 + lane) * 16, zero16());
  };
  if (a.d_rel_table != nullptr) clear_diag_tile();
  // this lane's diagonal dd = lane (and 64 + lane for lanes 0 .. 14) starts at column DPAD + dd - 15 of row 0 and moves one row down, one
  // column right per element: byte offset qr * (DSTR + 2)
  const char* dg1 = pw + (DPAD + lane - 15) * 2;
  const char* dg2 = pw + (DPAD + (lane < 15 ? lane : 14) + 49) * 2;
  // diagonal sums: relative position (key - query + Lq - 1) + 16 = 64 (ch - 2 i + cw) + rot + dd for the dd-th diagonal (0 .. 78) of
  // chunk ch of this wave's i-th block (q0 = 16 wave + 128 i)
  const int cwf = a.Lq - 16 * wave;                  // > 0 for a wave that has a block
  const int rot = cwf & 63, cw = cwf >> 6;
  const int src = (lane - rot) & 63;
  const bool up = lane >= rot;
  float R[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) R[k] = 0.f;
  int nblk = 0;
  for (int q0 = wave * 16; q0 < a.Lq; q0 += 128, ++nblk) {
    const u32x4 qf0 = qn[0], qf1 = qn[1], dof0 = don[0], dof1 = don[1];
    const float lse_q = lse_n * P5_LOG2E;
    const int qi = q0 + li;
    const bool qok = qi < a.Lq;
    const int qic = qok ? qi : a.Lq - 1;
    float Drow = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float x[8], y[8];
      unpack16<T>(don[c], x);
      unpack16<T>(on[c], y);
#pragma unroll
      for (int e = 0; e < 8; ++e) Drow += x[e] * y[e];
    }
    Drow += __shfl_xor(Drow, 16);
    Drow += __shfl_xor(Drow, 32);
    if (g == 0 && qok) a.Dvec[((size_t)b * a.H + h) * a.Lq + qi] = Drow;
    const float D_q = Drow;
    if (q0 + 128 < a.Lq) {
      const int qrow = q0 + 128 + li;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        qn[c] = qrow < a.Lq ? ld16(Q + (size_t)qrow * a.ldq + c * 32 + g * 8) : zero16();
        don[c] = qrow < a.Lq ? ld16(dO + (size_t)qrow * a.lddo + c * 32 + g * 8) : zero16();
        on[c] = qrow < a.Lq ? ld16(O + (size_t)qrow * a.ldo + c * 32 + g * 8) : zero16();
      }
      lse_n = qrow < a.Lq ? a.lse[((size_t)b * a.H + h) * a.Lq + qrow] : 0.f;
    }
    const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);
    const uint32_t* kp = use_bits ? a.keep_bits + (((size_t)b * a.H + h) * ((a.Lq + 15) / 16) + (q0 >> 4)) * 256 : nullptr;

    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch < nch) {
        const char* cK = tK + ch * 64 * 128;
        const char* cV = tV + ch * 64 * 128;
        float dsv[4][4];
        u32x4 kw[4];          // the forward's keep masks of this chunk's 16 (t, r): the half of each that holds this lane
        if (use_bits) {
#pragma unroll
          for (int t = 0; t < 4; ++t) kw[t] = ld16(kp + (lane >> 5) * 128 + ch * 16 + t * 4);
        }
        // a block whose 16 queries all exist, without a causal mask (every block of the encoder at L % 16 == 0): the key mask is the only
        // condition left, and it is part of the exponent
        const bool plain = !causal && q0 + 16 <= a.Lq;
        const uint32_t lanebit = 1u << (lane & 31);
        auto tiles = [&](auto plain_c, auto mode_c) {
          constexpr bool PLAIN = decltype(plain_c)::value;
          constexpr int MODE = decltype(mode_c)::value;        // dropout of the probabilities: 0 none, 1 the forward's stored keep masks, 2 re-hashed
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
            mma16<T>(sacc, ld16(cK + t * 2048 + koff0), qf0);
            mma16<T>(sacc, ld16(cK + t * 2048 + koff1), qf1);
            mma16<T>(dpacc, ld16(cV + t * 2048 + koff0), dof0);
            mma16<T>(dpacc, ld16(cV + t * 2048 + koff1), dof1);
            const int kb = ch * 64 + t * 16 + g * 4;
            const f32x4 kn = *(const f32x4*)(skneg + kb);
            const float* pb = sbias + (kb - qic + a.Lq - 1);
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (MODE == 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) mk[r] = (kw[t][r] & lanebit) ? a.drop.scale : 0.f;
            } else if constexpr (MODE == 2) {
#pragma unroll
              for (int r = 0; r < 4; ++r) mk[r] = p5_keep(seed, a.drop.site_key, rowbase + (uint32_t)(kb + r), a.drop.thr) ? a.drop.scale : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kj = kb + r;
              float ex = fmaf(sacc[r], P5_LOG2E, pb[r] + (kn[r] - lse_q));     // (key slots past Lk and masked keys carry -inf in skneg)
              if constexpr (!PLAIN) ex = (qok & !(causal & (kj > qi))) ? ex : P5_NEG_INF;
              const float p = p5_exp2(ex);
              dsv[t][r] = p * (dpacc[r] * mk[r] - D_q);
            }
            if (do_rel) st4<T>(pw + li * DSTR + (DPAD + t * 16 + g * 4) * C::SZ, dsv[t]);
          }
        };
        if (use_bits) { if (plain) tiles(AttnBool<true>(), P5EpiTagA<1>()); else tiles(AttnBool<false>(), P5EpiTagA<1>()); }
        else if (do_drop) { if (plain) tiles(AttnBool<true>(), P5EpiTagA<2>()); else tiles(AttnBool<false>(), P5EpiTagA<2>()); }
        else { if (plain) tiles(AttnBool<true>(), P5EpiTagA<0>()); else tiles(AttnBool<false>(), P5EpiTagA<0>()); }
        if (do_rel) {
          // sums of dS (as the MFMA sees it: bf16) along the 79 diagonals of this wave's [16 q][64 keys] tile: lane = diagonal for the
          // first 64, lanes 0..14 the rest; every read unconditional (clamped column, the value selected away) so that the sixteen of them
          // are in flight together
          P5_WAVE_SYNC();
          float x1 = 0.f, x2 = 0.f;
#pragma unroll
          for (int qr = 0; qr < 16; ++qr) {
            x1 += to_f<T>(*(const T*)(dg1 + qr * (DSTR + 2)));
            x2 += to_f<T>(*(const T*)(dg2 + qr * (DSTR + 2)));
          }
          x2 = lane < 15 ? x2 : 0.f;
          P5_WAVE_SYNC();
          const float y1 = __shfl(x1, src), y2 = __shfl(x2, src);
          R[ch] += up ? y1 : 0.f;
          R[ch + 1] += up ? y2 : y1;
          R[ch + 2] += up ? 0.f : y2;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float d8[8] = {dsv[2 * u][0], dsv[2 * u][1], dsv[2 * u][2], dsv[2 * u][3], dsv[2 * u + 1][0], dsv[2 * u + 1][1], dsv[2 * u + 1][2], dsv[2 * u + 1][3]};
          const u32x4 da = pack16<T>(d8);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const char* base = cK + (u * 32 + trow) * 128 + (((dt * 2 + ((li >> 1) & 1)) ^ tsw) << 4) + (li & 1) * 8;
            const u32x2 lo = lds_tr16_b64(base);
            const u32x2 hi = lds_tr16_b64(base + 16 * 128);
            u32x4 kb4;
            kb4[0] = lo[0]; kb4[1] = lo[1]; kb4[2] = hi[0]; kb4[3] = hi[1];
            mma16<T>(dq[dt], da, kb4);
          }
        }
      }
    }
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    wave_store_16x64<T>((T*)a.dQ + (size_t)b * a.Lq * a.lddq + h * 64, a.lddq, q0, a.Lq, dq, one, pw, lane);
    if (do_rel && q0 + 128 < a.Lq) {      // (the store staged dQ over the tile's zero columns)
      P5_WAVE_SYNC();
      clear_diag_tile();
    }
    // the next block's diagonals sit 128 positions lower: J' = J + 2, the accumulators move up by two (the top two wrap to the bottom)
    {
      const float t16 = R[16], t17 = R[17];
#pragma unroll
      for (int k = 17; k >= 2; --k) R[k] = R[k - 2];
      R[0] = t16; R[1] = t17;
    }
  }
  if (do_rel) {
    // R[k] holds J = (cw - 2 nblk + k) mod 18 ... after nblk rotations; relative position = 64 J + lane - 16.  Rows of 1024 sums per wave
    // over the dead K / V images, added in wave order
    __syncthreads();
    float* rows = (float*)tKV;
    for (int i = lane; i < 1024; i += 64) rows[wave * 1024 + i] = 0.f;
    P5_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      int J = (cw - 2 * nblk + k) % 18;
      if (J < 0) J += 18;
      const int idx = 64 * J + lane - 16;
      if (idx >= 0 && idx < nrel) rows[wave * 1024 + idx] = R[k];
    }
    __syncthreads();
    for (int i = tid; i < nrel; i += NT) {
      float t = rows[i];
#pragma unroll
      for (int w = 1; w < 8; ++w) t += rows[w * 1024 + i];
      rows[i] = t;
    }
    __syncthreads();
    rel_bias_grad_flush<512>(a, h, b, rows, sbias, tid);      // (sbias is dead from here on)
  }
}

//   part 2, dK and dV: Q and dO of the head resident (+ the per-query log-sum-exp and D); eight waves take the 16-key blocks round-robin.
template <int NQT>
__global__ __launch_bounds__(512) void p5_attn_bwd_dkv_head_kernel(P5AttnArgs a) {
  using T = bf16;
  using C = AttnC<T>;
  constexpr int LQ = NQT * 16, NT = 512, NP = LQ * 8 / NT;
  static_assert(NQT == 16 || NQT == 32, "whole-head attention backward: 256 or 512 query slots");
  __shared__ __attribute__((aligned(16))) char tQ[LQ * 128];
  __shared__ __attribute__((aligned(16))) char tDO[LQ * 128];
  __shared__ __attribute__((aligned(16))) char pbuf[8 * 16 * C::TS];
  // Element arithmetic of this pass (round 6; 36 vector instructions per score before, counted with SQ_INSTS_VALU): log2(e) is folded into
  // the bias table and the log-sum-exp rows, so a probability is one subtract, one fma and one v_exp_f32; the four bias entries of a lane's
  // (key, four consecutive queries) are four reads at constant offsets from one address -- the table carries PADB entries of slack in front
  // instead of a clamp per score (query slots past Lq are masked anyway); a masked score is a select on the exponent (-inf), never a
  // branch around the exp: hipcc sank the exp into `ok ? ... : 0` and emitted a compare + exec-mask branch per score.
  constexpr int PADB = 64;
  __shared__ float sbias[PADB + 1024];
  __shared__ __attribute__((aligned(16))) float slse[LQ];
  __shared__ __attribute__((aligned(16))) float sD[LQ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;
  const T* dO = (const T*)a.dO + (size_t)b * a.Lq * a.lddo + h * 64;
  const int nrel = a.Lq + a.Lk - 1;
  {
    u32x4 rq[NP], rd[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rq[i] = row < a.Lq ? ld16(Q + (size_t)row * a.ldq + pc * 8) : zero16();
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      rd[i] = row < a.Lq ? ld16(dO + (size_t)row * a.lddo + pc * 8) : zero16();
    }
    for (int i = tid; i < PADB + 1023; i += NT) {       // (every position the blocks read, key slots past Lk and query slots past Lq included)
      const int j = i - PADB;
      sbias[i] = (a.rel_table && j >= 0 && j < nrel) ? P5_LOG2E * a.rel_table[a.bucket_lut[j - (a.Lq - 1) + a.lut_half] * a.H + h] : 0.f;
    }
    for (int i = tid; i < LQ; i += NT) {
      slse[i] = i < a.Lq ? P5_LOG2E * a.lse[((size_t)b * a.H + h) * a.Lq + i] : 0.f;
      sD[i] = i < a.Lq ? a.Dvec[((size_t)b * a.H + h) * a.Lq + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 3, pc = p & 7;
      st16(tQ + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4), rq[i]);
      st16(tDO + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4), rd[i]);
    }
  }
  u32x4 kn[2], vn[2];
  bool kok_n;
  {
    const int krow = wave * 16 + li;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      kn[c] = krow < a.Lk ? ld16(K + (size_t)krow * a.ldk + c * 32 + g * 8) : zero16();
      vn[c] = krow < a.Lk ? ld16(V + (size_t)krow * a.ldv + c * 32 + g * 8) : zero16();
    }
    kok_n = krow < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + krow] != 0);
  }
  __syncthreads();

  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const bool use_bits = do_drop && a.keep_bits != nullptr;
  const uint32_t seed = p5_seed(a.drop);
  const int ksw = (li >> 1) & 7;
  const int koff0 = li * 128 + (((0 + g) ^ ksw) << 4), koff1 = li * 128 + (((4 + g) ^ ksw) << 4);
  const int trow = g * 4 + (li >> 2), tsw = (g * 2 + (li >> 3)) & 7;
  char* pw = pbuf + wave * 16 * C::TS;
  const int nqc = (a.Lq + 63) / 64;
  for (int k0 = wave * 16; k0 < a.Lk; k0 += 128) {
    const u32x4 kf0 = kn[0], kf1 = kn[1], vf0 = vn[0], vf1 = vn[1];
    const bool kok = kok_n;
    const int kj = k0 + li;
    if (k0 + 128 < a.Lk) {
      const int krow = k0 + 128 + li;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        kn[c] = krow < a.Lk ? ld16(K + (size_t)krow * a.ldk + c * 32 + g * 8) : zero16();
        vn[c] = krow < a.Lk ? ld16(V + (size_t)krow * a.ldv + c * 32 + g * 8) : zero16();
      }
      kok_n = krow < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + krow] != 0);
    }
    const uint32_t headbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq) * a.Lk) + (uint32_t)kj;
    // the forward's keep masks (P5AttnArgs::keep_bits): this lane's key is (t, r) = (kj / 16, kj % 4) in lanes 16 g' .. 16 g' + 15, g' = kj / 4 % 4
    const char* kpk = use_bits ? (const char*)(a.keep_bits + ((size_t)b * a.H + h) * ((a.Lq + 15) / 16) * 256) +
                                     ((((kj >> 3) & 1) * 128 + (kj >> 4) * 4 + (kj & 3)) * 4 + ((kj >> 2) & 1) * 2) : nullptr;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    unsigned pcn[4] = {0u, 0u, 0u, 0u};          // this key's keep bits for the four 16-query blocks of the NEXT 64-query chunk
    if (use_bits) {
#pragma unroll
      for (int t = 0; t < 4; ++t) pcn[t] = t * 16 < a.Lq ? *(const unsigned short*)(kpk + (size_t)t * 1024) : 0u;
    }
    for (int qc = 0; qc < nqc; ++qc) {
      const char* cQ = tQ + qc * 64 * 128;
      const char* cD = tDO + qc * 64 * 128;
      float pv[4][4], dsv[4][4];
      const unsigned pcc[4] = {pcn[0], pcn[1], pcn[2], pcn[3]};
      if (use_bits && (qc + 1) * 64 < a.Lq) {      // fetched under this chunk's work
#pragma unroll
        for (int t = 0; t < 4; ++t) pcn[t] = ((qc + 1) * 4 + t) * 16 < a.Lq ? *(const unsigned short*)(kpk + (size_t)((qc + 1) * 4 + t) * 1024) : 0u;
      }
      // a chunk whose 64 query slots all exist, without a causal mask (every chunk of the encoder at L % 64 == 0): the only per-score
      // condition left is the lane's own key
      const bool plain = !causal && (qc + 1) * 64 <= a.Lq;
      auto tiles = [&](auto plain_c, auto mode_c) {
        constexpr bool PLAIN = decltype(plain_c)::value;
        constexpr int MODE = decltype(mode_c)::value;        // dropout of the probabilities: 0 none, 1 the forward's stored keep masks, 2 re-hashed
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
          mma16<T>(sacc, ld16(cQ + t * 2048 + koff0), kf0);
          mma16<T>(sacc, ld16(cQ + t * 2048 + koff1), kf1);
          mma16<T>(dpacc, ld16(cD + t * 2048 + koff0), vf0);
          mma16<T>(dpacc, ld16(cD + t * 2048 + koff1), vf1);
          const int qb = qc * 64 + t * 16 + g * 4;
          const f32x4 ls = *(const f32x4*)(slse + qb), dd = *(const f32x4*)(sD + qb);
          const float* pbq = sbias + (PADB + kj - qb + a.Lq - 4);      // entries of queries qb + 3 .. qb (relative position key - query, descending)
          const float bias[4] = {pbq[3], pbq[2], pbq[1], pbq[0]};
          const uint32_t tbase = headbase + (uint32_t)qb * (uint32_t)a.Lk;
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if constexpr (MODE == 1) {
            const unsigned piece = pcc[t] >> (g * 4);       // 16 queries of block qc * 4 + t: this lane group's four
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[r] = ((piece >> r) & 1u) ? a.drop.scale : 0.f;
          } else if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[r] = p5_keep(seed, a.drop.site_key, tbase + (uint32_t)r * (uint32_t)a.Lk, a.drop.thr) ? a.drop.scale : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qb + r;
            const bool ok = PLAIN ? kok : (kok & (qi < a.Lq) & !(causal & (kj > qi)));
            float ex = fmaf(sacc[r], P5_LOG2E, bias[r] - ls[r]);
            ex = ok ? ex : P5_NEG_INF;
            const float p = p5_exp2(ex);
            pv[t][r] = p * mk[r];
            dsv[t][r] = p * (dpacc[r] * mk[r] - dd[r]);
          }
        }
      };
      // (every condition that is uniform over the launch decided here, once per chunk -- not per tile)
      if (use_bits) { if (plain) tiles(AttnBool<true>(), P5EpiTagA<1>()); else tiles(AttnBool<false>(), P5EpiTagA<1>()); }
      else if (do_drop) { if (plain) tiles(AttnBool<true>(), P5EpiTagA<2>()); else tiles(AttnBool<false>(), P5EpiTagA<2>()); }
      else { if (plain) tiles(AttnBool<true>(), P5EpiTagA<0>()); else tiles(AttnBool<false>(), P5EpiTagA<0>()); }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float p8[8] = {pv[2 * u][0], pv[2 * u][1], pv[2 * u][2], pv[2 * u][3], pv[2 * u + 1][0], pv[2 * u + 1][1], pv[2 * u + 1][2], pv[2 * u + 1][3]};
        const float d8[8] = {dsv[2 * u][0], dsv[2 * u][1], dsv[2 * u][2], dsv[2 * u][3], dsv[2 * u + 1][0], dsv[2 * u + 1][1], dsv[2 * u + 1][2], dsv[2 * u + 1][3]};
        const u32x4 pa = pack16<T>(p8), sa = pack16<T>(d8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int off = (u * 32 + trow) * 128 + (((dt * 2 + ((li >> 1) & 1)) ^ tsw) << 4) + (li & 1) * 8;
          u32x2 lo = lds_tr16_b64(cD + off), hi = lds_tr16_b64(cD + off + 16 * 128);
          u32x4 f;
          f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
          mma16<T>(dv[dt], pa, f);
          lo = lds_tr16_b64(cQ + off);
          hi = lds_tr16_b64(cQ + off + 16 * 128);
          f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
          mma16<T>(dk[dt], sa, f);
        }
      }
    }
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    wave_store_16x64<T>((T*)a.dK + (size_t)b * a.Lk * a.lddk + h * 64, a.lddk, k0, a.Lk, dk, one, pw, lane);
    wave_store_16x64<T>((T*)a.dV + (size_t)b * a.Lk * a.lddv + h * 64, a.lddv, k0, a.Lk, dv, one, pw, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward for SHORT query blocks (Lq <= 16: the decoder's self-attention over T target tokens and its cross-attention, T <= 16 queries
// against the encoder's L keys) -- dQ, dK, dV and the relative-bias gradient in ONE launch, one workgroup per (batch, head).
// The two-kernel path above runs these as a dQ kernel in which only one of four waves has queries (the wave = 16 queries split leaves
// three idle) followed by a dK/dV kernel: 24 latency-bound launches per T5-small step.  Here the four waves split the KEYS of every
// 64-key chunk (wave w: keys 16 w .. 16 w + 15): each forms its [16 keys x 16 queries] block of P and dS once and uses it three times:
//   dV += P^T dO, dK += dS^T Q (its own 16 keys: stored per chunk), dQ += dS K (accumulated over all chunks, the four waves' partial
//   sums added in wave order at the end).  Same element arithmetic as the other kernels (recomputed P from the saved log-sum-exp).
// ------------------------------------------------------------------------------------------------------------
#ifdef P5_EMU
#define P5_ATTN_SMALL_OCC
#else
#define P5_ATTN_SMALL_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))      // two workgroups per CU (<= 256 registers, <= 80 KiB of LDS in bf16)
#endif
template <class T>
__global__ __launch_bounds__(256) P5_ATTN_SMALL_OCC void p5_attn_bwd_small_kernel(P5AttnArgs a) {
  using C = AttnC<T>;
  constexpr int KR = C::KCH;                 // reduction elements of one mma16 (32 bf16 / 16 f32) = rows of a K-strided fragment chunk
  constexpr int RS = 64 + 16;                // row stride (bytes) of the small A-operand images: KR elements + pad
    __shared__ __attribute__((aligned(16))) char tileKV[2 * 64 * C::TS];
  char* const tileK = tileKV;
  char* const tileV = tileKV + 64 * C::TS; __shared__ __attribute__((aligned(16))) char tileQ[32 * C::TS];       // rows >= Lq are zero
  __shared__ __attribute__((aligned(16))) char tileDO[32 * C::TS];
  __shared__ __attribute__((aligned(16))) char aP[4][16 * RS];          // per wave: P^T  [key][query]   (A operand of dV)
  __shared__ __attribute__((aligned(16))) char aS[4][16 * RS];          //           dS^T [key][query]   (A operand of dK)
  __shared__ __attribute__((aligned(16))) char bS[4][16 * RS];          //           dS   [query][key of the wave's KR-row chunk]  (A operand of dQ)
  __shared__ __attribute__((aligned(16))) char wscr[4][16 * C::TS];     // wave scratch of the row stores
  static_assert(2 * 64 * C::TS >= 4 * 16 * 68 * 4, "the dQ partial sums reuse the K and V tiles");
  float (*sdq)[16][68] = (float (*)[16][68])tileKV;                     // [4][16][68] fp32 partial dQ, after the last chunk
  __shared__ __attribute__((aligned(16))) float sbias[1024];
  __shared__ float sdb[4][528];
  __shared__ __attribute__((aligned(16))) float slse[16];
  __shared__ __attribute__((aligned(16))) float sD[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;
  const T* dO = (const T*)a.dO + (size_t)b * a.Lq * a.lddo + h * 64;
  const T* O = (const T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64;
  const int nrel = a.Lq + a.Lk - 1;
  const int nch = (a.Lk + 63) / 64;

  // ---- stage Q, dO (zero rows beyond Lq), the per-head bias over relative positions, lse, D = rowsum(dO * O) ----
  for (int p = tid; p < 32 * C::PPR; p += 256) {
    const int row = p / C::PPR, pc = p % C::PPR;
    st16(tileQ + row * C::TS + pc * 16, row < a.Lq ? ld16(Q + (size_t)row * a.ldq + pc * C::EPF) : zero16());
    st16(tileDO + row * C::TS + pc * 16, row < a.Lq ? ld16(dO + (size_t)row * a.lddo + pc * C::EPF) : zero16());
  }
  if (a.rel_table) {
    for (int i = tid; i < nrel; i += 256) sbias[i] = a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h];
    if (a.d_rel_table)
      for (int i = tid; i < 4 * 528; i += 256) (&sdb[0][0])[i] = 0.f;
  }
  if (tid < 64) {        // 4 lanes per query, 16 columns each
    const int q = tid >> 2, part = tid & 3;
    float d = 0.f;
    if (q < a.Lq) {
      for (int c = 0; c < 16; c += C::EPF) {
        float x[8], y[8];
        unpack16<T>(ld16(dO + (size_t)q * a.lddo + part * 16 + c), x);
        unpack16<T>(ld16(O + (size_t)q * a.ldo + part * 16 + c), y);
#pragma unroll
        for (int e = 0; e < C::EPF; ++e) d += x[e] * y[e];
      }
    }
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    if (part == 0) {
      sD[q] = d;
      slse[q] = q < a.Lq ? a.lse[((size_t)b * a.H + h) * a.Lq + q] : 0.f;
      if (a.Dvec && q < a.Lq) a.Dvec[((size_t)b * a.H + h) * a.Lq + q] = d;
    }
  }
  for (int i = lane; i < 16 * RS / 4; i += 64) {        // zero the wave's A-operand images once (their padding columns stay zero)
    ((unsigned*)aP[wave])[i] = 0u; ((unsigned*)aS[wave])[i] = 0u; ((unsigned*)bS[wave])[i] = 0u;
  }
  __syncthreads();

  u32x4 qf[C::NCK], dof[C::NCK];
#pragma unroll
  for (int c = 0; c < C::NCK; ++c) { qf[c] = tile_frag_kc<T>(tileQ, 0, c, lane); dof[c] = tile_frag_kc<T>(tileDO, 0, c, lane); }
  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);
  const int qb = g * 4;                                 // the lane's four queries qb .. qb + 3; its key is li of the wave's 16
  const f32x4 ls = *(const f32x4*)(slse + qb), dd = *(const f32x4*)(sD + qb);
  const int kc_w = (wave * 16) / KR, koff = (wave * 16) % KR;      // where the wave's 16 keys sit inside tileK's K-strided fragment chunks

  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();
    stage_tile64<T>(tileK, K + (size_t)ch * 64 * a.ldk, a.ldk, a.Lk - ch * 64, tid);
    stage_tile64<T>(tileV, V + (size_t)ch * 64 * a.ldv, a.ldv, a.Lk - ch * 64, tid);
    __syncthreads();
    const int k0 = ch * 64 + wave * 16, kj = k0 + li;
    if (k0 < a.Lk) {                                    // (wave-uniform; a wave without keys in this chunk only takes part in the barriers)
      const bool kok = kj < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + kj] != 0);
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C::NCK; ++c) {
        // C[row = query g*4 + r][col = key li]
        mma16<T>(sacc, qf[c], tile_frag_kc<T>(tileK, wave * 16, c, lane));
        mma16<T>(dpacc, dof[c], tile_frag_kc<T>(tileV, wave * 16, c, lane));
      }
      float pv[4], dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qb + r;
        const int qic = qi < a.Lq ? qi : a.Lq - 1;
        const float bias = a.rel_table ? sbias[(kj < a.Lk ? kj : a.Lk - 1) - qic + a.Lq - 1] : 0.f;
        const float mk = do_drop ? (p5_keep(seed, a.drop.site_key, (uint32_t)((((size_t)b * a.H + h) * a.Lq + qic) * a.Lk) + (uint32_t)kj, a.drop.thr) ? a.drop.scale : 0.f) : 1.f;
        const bool ok = kok & (qi < a.Lq) & !(causal & (kj > qi));
        const float p = p5_exp<T>((sacc[r] + bias) - ls[r]);
        pv[r] = ok ? p * mk : 0.f;
        dsv[r] = ok ? p * (dpacc[r] * mk - dd[r]) : 0.f;
      }
      P5_WAVE_SYNC();
      st4<T>(aP[wave] + li * RS + qb * C::SZ, pv);                 // [key li][queries qb..qb+3]
      st4<T>(aS[wave] + li * RS + qb * C::SZ, dsv);
#pragma unroll
      for (int r = 0; r < 4; ++r) *(T*)(bS[wave] + (qb + r) * RS + (koff + li) * C::SZ) = from_f<T>(dsv[r]);      // [query][key]
      P5_WAVE_SYNC();
      if (a.d_rel_table) {
        // sums of dS along the 31 diagonals (constant key - query) of the wave's [16 q][16 keys] block, into the wave's own row
        for (int dg = lane; dg < 31; dg += 64) {
          float sum = 0.f;
#pragma unroll
          for (int qr = 0; qr < 16; ++qr) {
            const int kcol = dg - 15 + qr;
            if (kcol >= 0 && kcol < 16) sum += to_f<T>(*(const T*)(bS[wave] + qr * RS + (koff + kcol) * C::SZ));
          }
          const int idx = k0 + dg - 15 + a.Lq - 1;
          if (idx >= 0 && idx < nrel) sdb[wave][idx] += sum;
        }
      }
      f32x4 dk[4], dv[4];
      const u32x4 pa = ld16(aP[wave] + li * RS + g * 16), sa = ld16(aS[wave] + li * RS + g * 16);
      const u32x4 dsa = ld16(bS[wave] + li * RS + g * 16);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma16<T>(dv[dt], pa, tile_frag_ks<T>(tileDO, dt * 16, 0, lane));
        mma16<T>(dk[dt], sa, tile_frag_ks<T>(tileQ, dt * 16, 0, lane));
        mma16<T>(dq[dt], dsa, tile_frag_ks<T>(tileK, dt * 16, kc_w, lane));
      }
      const float one[4] = {1.f, 1.f, 1.f, 1.f};
      wave_store_16x64<T>((T*)a.dK + (size_t)b * a.Lk * a.lddk + h * 64, a.lddk, k0, a.Lk, dk, one, wscr[wave], lane);
      wave_store_16x64<T>((T*)a.dV + (size_t)b * a.Lk * a.lddv + h * 64, a.lddv, k0, a.Lk, dv, one, wscr[wave], lane);
    }
  }
  // ---- dQ: the four waves' partial sums (each over its keys of every chunk), added in wave order ----
  __syncthreads();                 // (every wave is done with the K / V tiles: their storage now holds the partial sums)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sdq[wave][g * 4 + r][dt * 16 + li] = dq[dt][r];
  __syncthreads();
  if (wave == 0) {
    f32x4 t[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        t[dt][r] = ((sdq[0][g * 4 + r][dt * 16 + li] + sdq[1][g * 4 + r][dt * 16 + li]) + sdq[2][g * 4 + r][dt * 16 + li]) + sdq[3][g * 4 + r][dt * 16 + li];
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    wave_store_16x64<T>((T*)a.dQ + (size_t)b * a.Lq * a.lddq + h * 64, a.lddq, 0, a.Lq, t, one, wscr[0], lane);
  }
  if (a.d_rel_table) {
    for (int i = tid; i < nrel; i += 256) sdb[0][i] = ((sdb[0][i] + sdb[1][i]) + sdb[2][i]) + sdb[3][i];
    __syncthreads();
    rel_bias_grad_flush<256>(a, h, b, &sdb[0][0], sbias, tid);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, fused (bf16, Lq and Lk <= 128: the encoder's self-attention at the benchmark shape).  The two-kernel backward
// above reads Q, K, V, dO twice and recomputes P twice; measured on MI355X both kernels (and the forward) take ~0.4 us per
// MB they move whatever their arithmetic looks like (rewriting the per-element code branch-free changed nothing), i.e. the
// strided 128-byte head-slice rows of the fused projection output bound them, not MFMA or VALU.  Here ONE 8-wave workgroup
// per (batch, head) loads Q, K, V, dO once, keeps P and dS for the whole [Lq, Lk] block in LDS, and produces dQ, dK and dV:
//   phase A  wave w = queries 16w..16w+15: S^T and dP^T by MFMA, P and dS element-wise (transposed layout as above),
//            written to the [query][key] LDS matrices; d(rel-bias) diagonal sums; dQ = dS K;
//   phase B  wave w = keys 16w..16w+15: dV = P^T dO, dK = dS^T Q, the A fragments read TRANSPOSED out of the same matrices
//            (ds_read_b64_tr_b16), Q / dO as K-strided operands.
// LDS: four [128][64] tiles + two [128][128] matrices = 143 KiB -> one workgroup per CU.
// ------------------------------------------------------------------------------------------------------------
__device__ static __forceinline__ u32x4 frag_ks_stride(const char* lds, int stride, int e0, int kc, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* base = lds + (kc * 32 + g * 8 + (i >> 2)) * stride + (e0 + (i & 3) * 4) * 2;
  const u32x2 lo = lds_tr16_b64(base), hi = lds_tr16_b64(base + 4 * stride);
  u32x4 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
  return r;
}

template <class T>
__global__ __launch_bounds__(512) void p5_attn_bwd_fused_kernel(P5AttnArgs a) {
  static_assert(sizeof(T) == 2, "fused attention backward: bf16 only");
  using C = AttnC<T>;
  constexpr int TP = 128 * 2 + 16;      // row stride of the [128][128] bf16 matrices
  __shared__ __attribute__((aligned(16))) char tK[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char tV[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char tQ[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char tDO[128 * C::TS];
  __shared__ __attribute__((aligned(16))) char mP[128 * TP];
  __shared__ __attribute__((aligned(16))) char mS[128 * TP];
  __shared__ float sbias[256];
  __shared__ float sdiag[8 * 144];        // per-wave diagonal sums of dS
  __shared__ __attribute__((aligned(16))) float skneg[128];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* Q = (const T*)a.Q + (size_t)b * a.Lq * a.ldq + h * 64;
  const T* K = (const T*)a.K + (size_t)b * a.Lk * a.ldk + h * 64;
  const T* V = (const T*)a.V + (size_t)b * a.Lk * a.ldv + h * 64;
  const T* dO = (const T*)a.dO + (size_t)b * a.Lq * a.lddo + h * 64;
  const T* O = (const T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64;
  const int nrel = a.Lq + a.Lk - 1;

  // ---- every global load of the workgroup is issued before the first use ----
  u32x4 rq[2], rk[2], rv[2], rdo[2], of[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + i * 512, row = p >> 3, pc = p & 7;
    rq[i] = row < a.Lq ? ld16(Q + (size_t)row * a.ldq + pc * 8) : zero16();
    rdo[i] = row < a.Lq ? ld16(dO + (size_t)row * a.lddo + pc * 8) : zero16();
    rk[i] = row < a.Lk ? ld16(K + (size_t)row * a.ldk + pc * 8) : zero16();
    rv[i] = row < a.Lk ? ld16(V + (size_t)row * a.ldv + pc * 8) : zero16();
  }
  const int q0 = wave * 16, qi = q0 + li;
  const bool qok = qi < a.Lq;
  const int qic = qok ? qi : a.Lq - 1;
#pragma unroll
  for (int c = 0; c < 2; ++c) of[c] = qok ? ld16(O + (size_t)qi * a.ldo + c * 32 + g * 8) : zero16();
  const float lse_q = qok ? P5_LOG2E * a.lse[((size_t)b * a.H + h) * a.Lq + qi] : 0.f;
  // (round 6, as the long-sequence passes: log2(e) folded into the bias table and the row's log-sum-exp, the key mask added to the
  //  exponent, a masked score a select on the exponent -- hipcc had turned `ok ? p ... : 0` into a compare + exec-mask branch per score)
  stage_bias_mask(a, b, h, sbias, skneg, 128, tid, P5_LOG2E, 256);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + i * 512, row = p >> 3, pc = p & 7;
    st16(tQ + row * C::TS + pc * 16, rq[i]);
    st16(tDO + row * C::TS + pc * 16, rdo[i]);
    st16(tK + row * C::TS + pc * 16, rk[i]);
    st16(tV + row * C::TS + pc * 16, rv[i]);
  }
  __syncthreads();

  // ---- phase A ----
  const bool causal = a.causal != 0;
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);
  const uint32_t rowbase = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk);
  u32x4 qf[2], dof[2];
  float D_q = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    qf[c] = tile_frag_kc<T>(tQ, q0, c, lane);
    dof[c] = tile_frag_kc<T>(tDO, q0, c, lane);
    float x[8], y[8];
    unpack16<T>(dof[c], x);
    unpack16<T>(of[c], y);
#pragma unroll
    for (int e = 0; e < 8; ++e) D_q += x[e] * y[e];
  }
  D_q += __shfl_xor(D_q, 16);
  D_q += __shfl_xor(D_q, 32);
  char* myP = mP + qi * TP;
  char* myS = mS + qi * TP;
  const bool plain = !causal && q0 + 16 <= a.Lq;      // every query of this wave's block exists, no causal mask
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      mma16<T>(sacc, tile_frag_kc<T>(tK, t * 16, c, lane), qf[c]);
      mma16<T>(dpacc, tile_frag_kc<T>(tV, t * 16, c, lane), dof[c]);
    }
    const int kb = t * 16 + g * 4;
    const f32x4 kn = *(const f32x4*)(skneg + kb);          // (0, or -inf for a masked key / a key slot past Lk)
    const float* pb = sbias + (kb - qic + a.Lq - 1);        // (< 256: Lq, Lk <= 128)
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (do_drop) {
#pragma unroll
      for (int r = 0; r < 4; ++r) mk[r] = p5_keep(seed, a.drop.site_key, rowbase + (uint32_t)(kb + r), a.drop.thr) ? a.drop.scale : 0.f;
    }
    float pv[4], dsv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kj = kb + r;
      float ex = fmaf(sacc[r], P5_LOG2E, pb[r] + (kn[r] - lse_q));
      if (!plain) ex = (qok & !(causal & (kj > qi))) ? ex : P5_NEG_INF;        // (wave-uniform test: the common case has no select at all)
      const float p = p5_exp2(ex);
      pv[r] = p * mk[r];
      dsv[r] = p * (dpacc[r] * mk[r] - D_q);
    }
    st4<T>(myP + kb * 2, pv);
    st4<T>(myS + kb * 2, dsv);
  }
  P5_WAVE_SYNC();
  if (a.d_rel_table) {
    // d(rel-bias): sums of dS along the 143 diagonals (constant key - query) of this wave's [16 q][128 keys] rows, one diagonal
    // per lane, 16 independent 2-byte reads each, kept PER WAVE: LDS float atomics retire at ~2.5 cycles per lane (one
    // ds_add_f32 per score element cost 20 us per workgroup), so nothing is accumulated atomically per element or per diagonal
    for (int dd = lane; dd < 128 + 15; dd += 64) {
      float sum = 0.f;
#pragma unroll
      for (int qr = 0; qr < 16; ++qr) {
        const int kcol = dd - 15 + qr;
        const bool in = (kcol >= 0) & (kcol < 128);
        const float v = to_f<T>(*(const T*)(mS + (q0 + qr) * TP + (in ? kcol : 0) * 2));
        sum += in ? v : 0.f;
      }
      sdiag[wave * 144 + dd] = sum;
    }
  }
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const u32x4 dsa = ld16(myS + kc * 64 + g * 16);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma16<T>(dq[dt], dsa, tile_frag_ks<T>(tK, dt * 16, kc, lane));
  }
  __syncthreads();      // P and dS complete; K and V tiles dead from here on

  if (a.d_rel_table) {
    // d(relative-bias table), every wave its share and no barrier: wave w owns the buckets w, w + 8, ...  A lane looks at the relative
    // positions i = lane + 64 j (nrel <= 255: four of them), forms their sums over the eight waves' diagonal sums once (relative position
    // i - (Lq - 1) is diagonal dd = i + 15 + 16 w' - (Lq - 1) of wave w' rows, added in wave order) and, per bucket, adds the positions
    // that map to it; a butterfly over the lanes (fixed association) gives the bucket's sum, which is STORED into this workgroup's slot
    // (buckets without positions store 0).  Same bits every run; ~0.5 us per wave instead of ~4 us on one wave that then finished last.
    float pv[4];
    int pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = lane + 64 * j;
      float v = 0.f;
      pb[j] = -1;
      if (i < nrel) {
        pb[j] = a.bucket_lut[i - (a.Lq - 1) + a.lut_half] & 63;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const int dd = i + 15 + 16 * w - (a.Lq - 1);
          const bool in = (dd >= 0) & (dd < 128 + 15);
          const float x = sdiag[w * 144 + (in ? dd : 0)];
          v += in ? x : 0.f;
        }
      }
      pv[j] = v;
    }
    for (int bk = wave; bk < a.rel_copies; bk += 8) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += pb[j] == bk ? pv[j] : 0.f;
      acc = wave_sum(acc);
      if (lane == 0) a.d_rel_table[(size_t)b * a.rel_stride + bk * a.H + h] = acc;
    }
  }
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  char* scratch = tK + wave * 16 * C::TS;     // (8 waves x 16 rows = exactly the K tile)
  // row sums of <d qkv, qkv> (P5AttnArgs::dot_out): a wave's queries and keys are the same 16 tokens, so the three inner products of a token
  // meet in one lane group.  Its K rows are this wave's store scratch: read them first.
  const bool want_dot = a.dot_out != nullptr;
  float dot[2] = {0.f, 0.f};
  u32x4 fwk[2], fwq[2], fwv[2];
  if (want_dot) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int off = (wave * 16 + p * 8 + (lane >> 3)) * C::TS + (lane & 7) * 16;
      fwk[p] = ld16(tK + off);
      fwq[p] = ld16(tQ + off);
      fwv[p] = ld16(tV + off);
    }
    wave_store_16x64_dot<T>((T*)a.dQ + (size_t)b * a.Lq * a.lddq + h * 64, a.lddq, q0, a.Lq, dq, scratch, lane, fwq, dot);
  } else {
    wave_store_16x64<T>((T*)a.dQ + (size_t)b * a.Lq * a.lddq + h * 64, a.lddq, q0, a.Lq, dq, one, scratch, lane);
  }

  // ---- phase B ----
  const int k0 = wave * 16;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const u32x4 pa = frag_ks_stride(mP, TP, k0, kc, lane);
    const u32x4 sa = frag_ks_stride(mS, TP, k0, kc, lane);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      mma16<T>(dv[dt], pa, tile_frag_ks<T>(tDO, dt * 16, kc, lane));
      mma16<T>(dk[dt], sa, tile_frag_ks<T>(tQ, dt * 16, kc, lane));
    }
  }
  if (want_dot) {
    wave_store_16x64_dot<T>((T*)a.dK + (size_t)b * a.Lk * a.lddk + h * 64, a.lddk, k0, a.Lk, dk, scratch, lane, fwk, dot);
    wave_store_16x64_dot<T>((T*)a.dV + (size_t)b * a.Lk * a.lddv + h * 64, a.lddv, k0, a.Lk, dv, scratch, lane, fwv, dot);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float t = dot[p];
      t += __shfl_xor(t, 1);
      t += __shfl_xor(t, 2);
      t += __shfl_xor(t, 4);
      const int tok = wave * 16 + p * 8 + (lane >> 3);
      if ((lane & 7) == 0 && tok < a.Lq) a.dot_out[((size_t)b * a.Lq + tok) * a.H + h] = t;
    }
    return;
  }
  wave_store_16x64<T>((T*)a.dK + (size_t)b * a.Lk * a.lddk + h * 64, a.lddk, k0, a.Lk, dk, one, scratch, lane);
  wave_store_16x64<T>((T*)a.dV + (size_t)b * a.Lk * a.lddv + h * 64, a.lddv, k0, a.Lk, dv, one, scratch, lane);
}
