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
  int rel_copies;        // d_rel_table holds this many partial copies (stride rel_stride floats) to spread atomics; 0/1 = one
  int rel_stride;
  P5Drop drop;
};


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

template <class T> __device__ static __forceinline__ float p5_exp(float x);
template <> __device__ __forceinline__ float p5_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float p5_exp<bf16>(float x) { return __expf(x); }

// stage per-head relative bias (index = k - q + Lq - 1) and the additive key mask into LDS
__device__ static __forceinline__ void stage_bias_mask(const P5AttnArgs& a, int b, int h, float* sbias, float* skneg,
                                                       int nkeys_padded, int tid) {
  const int nrel = a.Lq + a.Lk - 1;
  if (a.rel_table) {
    for (int i = tid; i < nrel; i += 256) sbias[i] = a.rel_table[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] * a.H + h];
  }
  for (int j = tid; j < nkeys_padded; j += 256)
    skneg[j] = (j < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + j] != 0)) ? 0.f : P5_NEG_INF;
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
  __shared__ float skneg[NKT * 16];

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
        for (int c = 0; c < C::NCK; ++c) mma16<T>(s[ch * 4 + t], qf[c], tile_frag_kc<T>(tile, t * 16, c, lane));
    }
  }
  __syncthreads();  // sbias/skneg visible even when Lk == 0 chunks were skipped

  // ---- exact softmax over the register-resident score rows ----
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = P5_NEG_INF;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kj = t * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + g * 4 + r;
      float v = P5_NEG_INF;
      if (kj < a.Lk && qi < a.Lq && !(a.causal && kj > qi)) {
        v = s[t][r] + skneg[kj];
        if (a.rel_table) v += sbias[kj - qi + a.Lq - 1];
      }
      s[t][r] = v;
      m[r] = fmaxf(m[r], v);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = row16_max(m[r]);
    if (m[r] == P5_NEG_INF) m[r] = 0.f;
    l[r] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = p5_exp<T>(s[t][r] - m[r]);
      s[t][r] = p;
      l[r] += p;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    l[r] = row16_sum(l[r]);
    const int qi = q0 + g * 4 + r;
    if (li == 0 && qi < a.Lq && a.lse) a.lse[((size_t)b * a.H + h) * a.Lq + qi] = m[r] + logf(l[r]);
  }
  if (a.drop.state != nullptr && a.drop.thr != 0) {
    const uint32_t seed = p5_seed(a.drop);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + g * 4 + r, kj = t * 16 + li;
        const uint32_t idx = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk + kj);
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
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(T*)(pw + (g * 4 + r) * C::TS + (t * 16 + li) * C::SZ) = from_f<T>(s[ch * 4 + t][r]);
      __syncthreads();
#pragma unroll
      for (int kc = 0; kc < C::NCK; ++kc) {
        const u32x4 pa = ld16(pw + li * C::TS + kc * 64 + g * 16);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma16<T>(o[dt], pa, tile_frag_ks<T>(tile, dt * 16, kc, lane));
      }
    }
  }
  float inv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv[r] = l[r] > 0.f ? 1.f / l[r] : 0.f;
  wave_store_16x64<T>((T*)a.O + (size_t)b * a.Lq * a.ldo + h * 64, a.ldo, q0, a.Lq, o, inv, pw, lane);
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
  __shared__ float sbias[1024];
  __shared__ float sdb[1024];
  __shared__ float skneg[512];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const T* Q = (const T*)a.Q; const T* K = (const T*)a.K; const T* V = (const T*)a.V;
  const T* dO = (const T*)a.dO; const T* O = (const T*)a.O;
  const int nrel = a.Lq + a.Lk - 1;
  const int nch = (a.Lk + 63) / 64;

  stage_bias_mask(a, b, h, sbias, skneg, nch * 64, tid);
  if (a.d_rel_table)
    for (int i = tid; i < nrel; i += 256) sdb[i] = 0.f;

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
  float lse_r[4], D_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + g * 4 + r;
    lse_r[r] = qi < a.Lq ? a.lse[((size_t)b * a.H + h) * a.Lq + qi] : 0.f;
    D_r[r] = __shfl(Drow, g * 4 + r);
  }
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);

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
        mma16<T>(sacc, qf[c], tile_frag_kc<T>(tileK, t * 16, c, lane));
        mma16<T>(dpacc, dof[c], tile_frag_kc<T>(tileV, t * 16, c, lane));
      }
      const int kj = ch * 64 + t * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + g * 4 + r;
        float ds = 0.f;
        if (kj < a.Lk && qi < a.Lq && !(a.causal && kj > qi) && skneg[kj] == 0.f) {
          float sv = sacc[r];
          if (a.rel_table) sv += sbias[kj - qi + a.Lq - 1];
          const float p = p5_exp<T>(sv - lse_r[r]);
          float dp = dpacc[r];
          if (do_drop) {
            const uint32_t idx = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk + kj);
            dp = p5_keep(seed, a.drop.site_key, idx, a.drop.thr) ? dp * a.drop.scale : 0.f;
          }
          ds = p * (dp - D_r[r]);
        }
        *(T*)(pw + (g * 4 + r) * C::TS + (t * 16 + li) * C::SZ) = from_f<T>(ds);
      }
    }
    __syncthreads();
    if (a.d_rel_table) {
      // d(rel-bias): sum dS along the diagonals (constant key - query) of this wave's [16 q][64 keys] tile held in
      // LDS -- one LDS atomic per diagonal per wave instead of one per score element
      for (int dd = lane; dd < 79; dd += 64) {
        float sum = 0.f;
#pragma unroll
        for (int qr = 0; qr < 16; ++qr) {
          const int kcol = dd - 15 + qr;
          if (kcol >= 0 && kcol < 64) sum += to_f<T>(*(const T*)(pw + qr * C::TS + kcol * C::SZ));
        }
        const int idx = ch * 64 + dd - 15 - q0 + a.Lq - 1;
        if (sum != 0.f && idx >= 0 && idx < nrel) atomicAdd(&sdb[idx], sum);
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
    // relative positions -> buckets inside the workgroup, then one global atomic per (bucket, head) into one of
    // `rel_copies` partial tables (same-address atomics from ~B*Lq/64 workgroups serialise at L2 otherwise)
    __syncthreads();
    float* sbk = sbias;   // sbias is dead from here on
    if (tid < 64) sbk[tid] = 0.f;
    __syncthreads();
    for (int i = tid; i < nrel; i += 256) {
      const float v = sdb[i];
      if (v != 0.f) atomicAdd(&sbk[a.bucket_lut[i - (a.Lq - 1) + a.lut_half] & 63], v);
    }
    __syncthreads();
    if (tid < 64 && sbk[tid] != 0.f) {
      const int copy = a.rel_copies > 1 ? (b % a.rel_copies) : 0;
      atomicAdd(&a.d_rel_table[(size_t)copy * a.rel_stride + tid * a.H + h], sbk[tid]);
    }
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
  __shared__ float slse[64];
  __shared__ float sD[64];

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
  bool kvalid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = k0 + g * 4 + r;
    kvalid[r] = kj < a.Lk && (!a.kmask || a.kmask[(size_t)b * a.Lk + kj] != 0);
  }
  const bool do_drop = a.drop.state != nullptr && a.drop.thr != 0;
  const uint32_t seed = p5_seed(a.drop);

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
        mma16<T>(sacc, kf[c], tile_frag_kc<T>(tileQ, t * 16, c, lane));
        mma16<T>(dpacc, vf[c], tile_frag_kc<T>(tileDO, t * 16, c, lane));
      }
      const int qi = qc * 64 + t * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k0 + g * 4 + r;
        float pd = 0.f, ds = 0.f;
        if (kvalid[r] && qi < a.Lq && !(a.causal && kj > qi)) {
          float sv = sacc[r];
          if (a.rel_table) sv += sbias[kj - qi + a.Lq - 1];
          const float p = p5_exp<T>(sv - slse[t * 16 + li]);
          float mk = 1.f;
          if (do_drop) {
            const uint32_t idx = (uint32_t)((((size_t)b * a.H + h) * a.Lq + qi) * a.Lk + kj);
            mk = p5_keep(seed, a.drop.site_key, idx, a.drop.thr) ? a.drop.scale : 0.f;
          }
          pd = p * mk;
          ds = p * (dpacc[r] * mk - sD[t * 16 + li]);
        }
        *(T*)(pP + (g * 4 + r) * C::TS + (t * 16 + li) * C::SZ) = from_f<T>(pd);
        *(T*)(pS + (g * 4 + r) * C::TS + (t * 16 + li) * C::SZ) = from_f<T>(ds);
      }
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
