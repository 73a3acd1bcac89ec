// p5_decode2.h -- latency-shaped kernels of the decode step (R = batch x beams rows, a few hundred at most).
//
// One decode step of HF beam search (DistributedRunner.py:361-371 -> transformers generation/utils.py:3384-3420) is a chain
// of ~45 dependent launches over R rows; none of them has enough work for a throughput-shaped tile, so what
// bounds a launch is the bytes ONE workgroup has to pull in before it can finish, and how many round trips it needs to do so.
// These kernels are therefore built the other way round from p5_gemm.h:
//   * small output tiles (16 rows x 16..64 columns) so that 100-800 workgroups share the operand bytes of a launch;
//   * every operand byte a workgroup needs is requested UP FRONT with direct-to-LDS copies (one HBM/L2 round trip), then one
//     barrier, then a few MFMAs -- no K loop pipeline to fill and drain;
//   * the residual stream stays in fp32 and is updated in place with atomics by the output / FFN-down projections (split-K
//     over workgroups needs no second pass), and the consuming projection normalises the rows itself (it owns whole rows
//     because K = d_model): T5LayerNorm (HF modeling_t5.py:59-72) costs no launch, no folded weight copy, no statistics buffer;
//   * cross-attention: one workgroup per (batch item, head) stages that item's K/V once for all its beams (the reference
//     expands encoder states x num_beams, P5_T5.py:571-576) and also computes the beams' q projection.
#pragma once
#include "p5_device.h"
#include "p5_gemm.h"

enum P5SkinnyEpi : int {
  P5_SK_STORE = 0,      // C(T)   = acc * alpha
  P5_SK_RELU = 1,       // C(T)   = relu(acc)
  P5_SK_ATOMIC = 2,     // C(f32) += acc            (residual stream update, split-K over workgroups)
  P5_SK_STORE_F32 = 3,  // C(f32) = acc * alpha
  P5_SK_RESID = 4,      // C(f32) = C + acc         (residual stream update by the element's ONLY writer: no K split, bit-reproducible)
};

struct P5SkinnyArgs {
  const void* A;        // AMODE 0: T [M, lda] (K-contiguous);  AMODE 1: fp32 residual stream [M, K]
  const float* ln;      // AMODE 1: norm weight [K]
  const void* W;        // T [N, ldw] (nn.Linear layout, K-contiguous)
  void* C;
  int M, N, K, lda, ldw, ldc;
  int kw;               // K range per pass (AMODE 0); = K for AMODE 1
  int kpasses;          // AMODE 0: passes of kw a workgroup makes one after the other over its K range (1 = the whole range up front)
  int epi;
  float alpha, eps;
  const int* done;      // optional device flag: != 0 -> the whole launch is a no-op (search finished in an earlier step)
};

template <class T> struct SkT {
  static constexpr int EPS = 128 / (int)sizeof(T);   // K elements per 128-byte step
};

// 16 rows of the fp32 residual stream -> T5LayerNorm -> swizzled A image [step][16][128 B]
template <class T>
__device__ static __forceinline__ void sk_norm_rows(char* aimg, const float* __restrict__ x, const float* __restrict__ ln, int m0, int M, int d,
                                                    float eps, int tid) {
  constexpr int EPS = SkT<T>::EPS;
  const int lane = tid & 63, wave = tid >> 6;
  // all four rows of this wave are requested before the first reduction (four dependent round trips otherwise)
  float xv[4][2][8];
  float wv[2][8];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int gr = m0 + wave * 4 + rr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + i * 64;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[rr][i][e] = 0.f;
      if (c * 8 < d && gr < M) ldf<8>(x + (size_t)gr * d + c * 8, xv[rr][i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[i][e] = 0.f;
    if (c * 8 < d) ldf<8>(ln + c * 8, wv[i]);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = wave * 4 + rr;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += xv[rr][i][e] * xv[rr][i][e];
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < d) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wv[i][e] * to_f<T>(from_f<T>(xv[rr][i][e] * rstd));   // reference rounding order (p5_rmsnorm_fwd_kernel)
        const int k0 = c * 8, step = k0 / EPS;
        if constexpr (sizeof(T) == 2) {
          const int slot = (k0 % EPS) >> 3;
          st16(aimg + step * 2048 + row * 128 + ((slot ^ (row & 7)) << 4), pack16<T>(o));
        } else {
          const int slot = (k0 % EPS) >> 2;
          st16(aimg + step * 2048 + row * 128 + ((slot ^ (row & 7)) << 4), pack16<T>(o));
          st16(aimg + step * 2048 + row * 128 + (((slot + 1) ^ (row & 7)) << 4), pack16<T>(o + 4));
        }
      }
    }
  }
}

// direct-to-LDS copy of `nrows` (multiple of 8) rows x nsteps x 128 B of a K-contiguous operand into a swizzled image
// [step][nrows][128 B]; the wave instructions are dealt round-robin to the four waves.  Rows past `rmax` are clamped.
template <class T>
__device__ static __forceinline__ void sk_dma_rows(char* img, const T* __restrict__ p, int ld, int r0, int rmax, int k0, int nrows, int nsteps,
                                                   int tid) {
  constexpr int EPS = SkT<T>::EPS, EPF = TT<T>::EPF;
  const int lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int per_step = nrows >> 3;
  const int total = per_step * nsteps;
  for (int q = wave; q < total; q += 4) {
    const int step = q / per_step, rb = (q % per_step) * 8;
    const int row = rb + (lane >> 3), slot = lane & 7;
    int gr = r0 + row;
    gr = gr < rmax ? gr : rmax - 1;
    glds16(p + (size_t)gr * ld + k0 + step * EPS + ((slot ^ (row & 7)) * EPF), img + (size_t)step * nrows * 128 + rb * 128);
  }
}

// C tile 16 x NB over K = nsteps * EPS from the two images; the 4 waves own NB/16 column tiles x 4/(NB/16) K parts.
// Returns this wave's accumulator (complete only on waves with K part 0 after the cross-wave reduction).
template <class T, int NB>
__device__ static __forceinline__ f32x4 sk_mma(const char* aimg, const char* bimg, int nsteps, float* red, int tid) {
  constexpr int NT = NB / 16, KP = 4 / NT;
  const int lane = tid & 63, wave = tid >> 6;
  const int nt = wave % NT, kp = wave / NT;
  const int per = (nsteps + KP - 1) / KP;
  const int s0 = kp * per, s1 = (s0 + per < nsteps) ? s0 + per : nsteps;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  // four steps (eight 64-byte chunks) at a time: all sixteen fragment reads are issued before the first MFMA -- with one or two
  // waves per SIMD nothing else hides the LDS latency of a read-then-multiply sequence
  for (int sb = s0; sb < s1; sb += 4) {
    u32x4 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int s = sb + (i >> 1);
      s = s < s1 ? s : s1 - 1;                                   // (steps past the range re-read the last one; masked below)
      fa[i] = frag_load_kc128<T>(aimg + s * 2048, 0, i & 1, lane);
      fb[i] = frag_load_kc128<T>(bimg + (size_t)s * NB * 128, nt * 16, i & 1, lane);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (sb + (i >> 1) < s1) mma16<T>(acc, fa[i], fb[i]);
  }
  if constexpr (KP > 1) {
    f32x4* r4 = (f32x4*)red;
    if (kp > 0) r4[(wave - NT) * 64 + lane] = acc;
    __syncthreads();
    if (kp == 0) {
#pragma unroll
      for (int j = 1; j < KP; ++j) {
        const f32x4 o = r4[((j * NT + nt) - NT) * 64 + lane];
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
      }
    }
  }
  return acc;
}

// y[16 x NB tile] of  A W^T  (see P5SkinnyArgs).  grid = (ceil(N / NB), ceil(M / 16), K splits), 256 threads.
template <class T, int NB, int AMODE, int LDSKB>
__global__ __launch_bounds__(256) void p5_skinny_gemm_kernel(P5SkinnyArgs g) {
  constexpr int EPS = SkT<T>::EPS, NT = NB / 16, KP = 4 / NT;
  __shared__ __attribute__((aligned(16))) char lds[LDSKB * 1024];
  if (g.done && *g.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NB, m0 = blockIdx.y * 16;
  const int npass = AMODE == 1 ? 1 : (g.kpasses > 0 ? g.kpasses : 1);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int ps = 0; ps < npass; ++ps) {
    const int k0 = AMODE == 1 ? 0 : (blockIdx.z * npass + ps) * g.kw;
    int kw = AMODE == 1 ? g.K : ((k0 + g.kw <= g.K) ? g.kw : g.K - k0);
    if (kw <= 0) break;
    const int nsteps = kw / EPS;
    char* aimg = lds;
    char* bimg = lds + (size_t)nsteps * 2048;
    float* red = (float*)(bimg + (size_t)nsteps * NB * 128);
    if (ps > 0) __syncthreads();          // the previous pass's images (and its cross-wave reduction buffer) are read out
    sk_dma_rows<T>(bimg, (const T*)g.W, g.ldw, n0, g.N, k0, NB, nsteps, tid);
    if constexpr (AMODE == 1) sk_norm_rows<T>(aimg, (const float*)g.A, g.ln, m0, g.M, g.K, g.eps, tid);
    else sk_dma_rows<T>(aimg, (const T*)g.A, g.lda, m0, g.M, k0, 16, nsteps, tid);
    __syncthreads();          // (drains the direct-to-LDS copies: cdna_hip_programming.md section 5)
    const f32x4 part = sk_mma<T, NB>(aimg, bimg, nsteps, red, tid);
    acc[0] += part[0]; acc[1] += part[1]; acc[2] += part[2]; acc[3] += part[3];      // (complete on the waves with K part 0 only)
  }
  if (KP > 1 && wave >= NT) return;
  const int col = n0 + (wave % NT) * 16 + (lane & 15);
  if (col >= g.N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + (lane >> 4) * 4 + r;
    if (row >= g.M) continue;
    const size_t ci = (size_t)row * g.ldc + col;
    float v = acc[r];
    if (g.epi == P5_SK_ATOMIC) atomicAdd((float*)g.C + ci, v);
    else if (g.epi == P5_SK_RESID) ((float*)g.C)[ci] += v;
    else if (g.epi == P5_SK_STORE_F32) ((float*)g.C)[ci] = v * g.alpha;
    else if (g.epi == P5_SK_RELU) ((T*)g.C)[ci] = from_f<T>(v > 0.f ? v : 0.f);
    else ((T*)g.C)[ci] = from_f<T>(v * g.alpha);
  }
}

// ---- hn(T) = T5LayerNorm(x32)  (input of the tied head) ----
template <class T>
__global__ __launch_bounds__(256) void p5_rmsnorm_f32in_kernel(T* __restrict__ y, const float* __restrict__ x, const float* __restrict__ w, int rows, int d,
                                                              float eps, const int* __restrict__ done) {
  if (done && *done) return;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float xv[2][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[i][e] = 0.f;
    if (c * 8 < d) {
      ldf<8>(x + (size_t)row * d + c * 8, xv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)d + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + i * 64;
    if (c * 8 < d) {
      float wv[8], o[8];
      ldf<8>(w + c * 8, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = wv[e] * to_f<T>(from_f<T>(xv[i][e] * rstd));
      if constexpr (sizeof(T) == 2) {
        st16(y + (size_t)row * d + c * 8, pack16<T>(o));
      } else {
        st16(y + (size_t)row * d + c * 8, pack16<T>(o));
        st16(y + (size_t)row * d + c * 8 + 4, pack16<T>(o + 4));
      }
    }
  }
}

// ---- single-token self-attention over the ancestry-indexed cache, one wave per (row, head) ----
// lane = (key slot ts = lane / 8, dim chunk dc = lane % 8 -> 8 dims): 8 cached positions are scored per pass with 16-byte
// loads, all passes' loads are independent (NP passes, unrolled: 8 for max_len <= 64, 16 up to P5_MAX_LEN = 128), so a wave has one
// ancestry gather and one K/V gather in flight instead of a dependent chain of `pos` of each.
template <class T, int NP = 8>
__global__ __launch_bounds__(256) void p5_dec_self_attn2_kernel(T* __restrict__ out, const T* __restrict__ qkv, T* __restrict__ cache,
                                                               const int* __restrict__ anc_odd, const int* __restrict__ anc_even,
                                                               const float* __restrict__ rel_table, const int* __restrict__ lut, int lut_half, int R,
                                                               int H, const int* __restrict__ step, int max_len, const int* __restrict__ done) {
  if (done && *done) return;
  const int cur_len = *step;
  const int pos = cur_len - 1;
  const int* __restrict__ anc = (cur_len & 1) ? anc_odd : anc_even;
  const int lane = threadIdx.x & 63;
  const int rh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rh >= R * H) return;
  const int r = rh / H, h = rh % H;
  const int inner = H * 64;
  const int ts = lane >> 3, dc = lane & 7;
  const T* qp = qkv + (size_t)r * 3 * inner + h * 64 + dc * 8;
  float q[8], kc[8], vc[8];
  auto ld8 = [](const T* p, float* o) {
    if constexpr (sizeof(T) == 2) unpack16<T>(ld16(p), o);
    else { unpack16<T>(ld16(p), o); unpack16<T>(ld16(p + 4), o + 4); }
  };
  auto st8 = [](T* p, const float* o) {
    if constexpr (sizeof(T) == 2) st16(p, pack16<T>(o));
    else { st16(p, pack16<T>(o)); st16(p + 4, pack16<T>(o + 4)); }
  };
  ld8(qp, q);
  ld8(qp + inner, kc);
  ld8(qp + 2 * inner, vc);
  if (ts == 0) {     // this step's K/V enter the cache at `pos`: layout [max_len][R][2*inner] (K then V)
    st8(cache + ((size_t)pos * R + r) * 2 * inner + h * 64 + dc * 8, kc);
    st8(cache + ((size_t)pos * R + r) * 2 * inner + inner + h * 64 + dc * 8, vc);
  }
  float s[NP], vv[NP][8];
  float m = P5_NEG_INF;
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int t = it * 8 + ts;
    s[it] = P5_NEG_INF;
#pragma unroll
    for (int e = 0; e < 8; ++e) vv[it][e] = 0.f;
    float kk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) kk[e] = 0.f;
    float bias = 0.f;
    if (t == pos) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { kk[e] = kc[e]; vv[it][e] = vc[e]; }
    } else if (t < pos) {
      const T* base = cache + ((size_t)t * R + anc[(size_t)t * R + r]) * 2 * inner + h * 64 + dc * 8;
      ld8(base, kk);
      ld8(base + inner, vv[it]);
    }
    if (t <= pos) bias = rel_table[lut[(t - pos) + lut_half] * H + h];
    float d8 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d8 += q[e] * kk[e];
    d8 += __shfl_xor(d8, 1); d8 += __shfl_xor(d8, 2); d8 += __shfl_xor(d8, 4);     // (all lanes: the 8 lanes of a key slot agree on t)
    if (t <= pos) {
      s[it] = d8 + bias;
      m = fmaxf(m, s[it]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const float p = (s[it] == P5_NEG_INF) ? 0.f : expf(s[it] - m);
    l += p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += p * vv[it][e];
  }
  l += __shfl_xor(l, 8); l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o[e] += __shfl_xor(o[e], 8); o[e] += __shfl_xor(o[e], 16); o[e] += __shfl_xor(o[e], 32);
  }
  if (ts == 0) {
    const float inv = 1.f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    st8(out + (size_t)r * inner + h * 64 + dc * 8, o);
  }
}

// ---- single-token cross-attention of the (<= 16) beams of one batch item for one head, optionally with their q projection ----
// grid = (B * ceil(Kb / 16), H).  K/V of (item, head) are staged ONCE per workgroup in 128-key chunks (direct-to-LDS) and
// shared by the item's beams; FUSEQ additionally normalises the beams' residual rows and multiplies them with the head's 64
// rows of Wq (all of it requested up front together with the first K/V chunk).  Zero position bias, encoder padding mask
// (HF modeling_t5.py:336-343,404-432); softmax in fp32, streamed over the chunks.
struct P5CrossArgs {
  void* out;               // T [R, inner]
  const void* q;           // !FUSEQ: T [R, inner]
  const float* x;          // FUSEQ: fp32 residual stream [R, d]
  const float* ln;         // FUSEQ: cross-attention norm weight [d]
  const void* Wq;          // FUSEQ: T [inner, d]
  const void* kv;          // T [B*L, ldkv]: K (inner columns) then V of this layer
  int ldkv;                // row stride of kv in elements (2*inner, or n_layers*2*inner when all layers share one projection GEMM)
  const int64_t* mask;     // [B, L]
  int R, H, Kb, L, d;
  float eps;
  const int* done;
};

template <class T, bool FUSEQ, int LDSKB>
__global__ __launch_bounds__(256) void p5_dec_cross_attn2_kernel(P5CrossArgs a) {
  constexpr int KC = 128;                              // keys per chunk
  constexpr int NSD = 64 * (int)sizeof(T) / 128;       // 128-byte steps per 64-dim K/V row (1 bf16, 2 fp32)
  constexpr int KVB = KC * 128 * NSD;                  // bytes of one K (or V) chunk image
  __shared__ __attribute__((aligned(16))) char lds[LDSKB * 1024];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt_per = (a.Kb + 15) / 16;
  const int b = blockIdx.x / mt_per, mt = blockIdx.x % mt_per, h = blockIdx.y;
  const int r0 = b * a.Kb + mt * 16;
  const int nrow = (a.Kb - mt * 16) < 16 ? (a.Kb - mt * 16) : 16;
  const int inner = a.H * 64;
  // LDS carve-up
  float* sq = (float*)lds;                              // [16][64] q rows (fp32)
  float* sS = sq + 16 * 64;                             // [16][KC] scores -> probabilities
  float* sM = sS + 16 * KC;                             // [16] running max, [16] running sum, [16] rescale
  char* kimg = (char*)(sM + 64);
  char* vimg = kimg + KVB;
  char* aimg = vimg + KVB;
  const T* kvp = (const T*)a.kv + (size_t)b * a.L * a.ldkv + h * 64;
  auto stage_kv = [&](int j0) {
    // rows j0.. of K and V: key row stride = 2*inner elements; NSD steps of 128 B each
    sk_dma_rows<T>(kimg, kvp, a.ldkv, j0, a.L, 0, KC, NSD, tid);
    sk_dma_rows<T>(vimg, kvp + inner, a.ldkv, j0, a.L, 0, KC, NSD, tid);
  };
  stage_kv(0);
  if constexpr (FUSEQ) {
    constexpr int EPS = SkT<T>::EPS;
    const int nsteps = a.d / EPS;
    char* bimg = aimg + (size_t)nsteps * 2048;
    float* red = (float*)(bimg + (size_t)nsteps * 64 * 128);
    sk_dma_rows<T>(bimg, (const T*)a.Wq, a.d, h * 64, inner, 0, 64, nsteps, tid);
    sk_norm_rows<T>(aimg, a.x, a.ln, r0, r0 + nrow, a.d, a.eps, tid);
    __syncthreads();
    const f32x4 acc = sk_mma<T, 64>(aimg, bimg, nsteps, red, tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) sq[((lane >> 4) * 4 + r) * 64 + wave * 16 + (lane & 15)] = to_f<T>(from_f<T>(acc[r]));   // q as the activation dtype stores it
  } else {
    for (int i = tid; i < 16 * 64; i += 256) {
      const int row = i >> 6;
      sq[i] = row < nrow ? to_f<T>(((const T*)a.q)[(size_t)(r0 + row) * inner + h * 64 + (i & 63)]) : 0.f;
    }
  }
  if (tid < 16) { sM[tid] = P5_NEG_INF; sM[16 + tid] = 0.f; }
  float o[4] = {0.f, 0.f, 0.f, 0.f};                     // thread (row = tid / 16, dims (tid % 16) * 4 ..)
  const int orow = tid >> 4, od = (tid & 15) * 4;
  for (int j0 = 0; j0 < a.L; j0 += KC) {
    if (j0 > 0) { __syncthreads(); stage_kv(j0); }
    __syncthreads();                                     // K/V chunk landed (and sq / sM written)
    {
      // scores: thread = (key j = tid % 128, row group g = tid / 128 -> rows g*8 .. g*8+7)
      const int j = tid & (KC - 1), g = tid >> 7;
      float acc8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc8[i] = 0.f;
      constexpr int EPF = TT<T>::EPF;
#pragma unroll
      for (int c = 0; c < 64 / EPF; ++c) {
        float kk[8];
        const int stepc = (c * EPF * (int)sizeof(T)) / 128, slot = ((c * EPF * (int)sizeof(T)) % 128) >> 4;
        unpack16<T>(ld16(kimg + stepc * KC * 128 + j * 128 + ((slot ^ (j & 7)) << 4)), kk);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float* qr = sq + (g * 8 + i) * 64 + c * EPF;
#pragma unroll
          for (int e = 0; e < EPF; ++e) acc8[i] += qr[e] * kk[e];
        }
      }
      const bool ok = (j0 + j) < a.L && a.mask[(size_t)b * a.L + j0 + j] != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) sS[(g * 8 + i) * KC + j] = ok ? acc8[i] : P5_NEG_INF;
    }
    __syncthreads();
    {
      // online softmax: 16 threads per row, 8 keys each
      const int row = tid >> 4, sub = tid & 15;
      float cm = P5_NEG_INF;
#pragma unroll
      for (int i = 0; i < KC / 16; ++i) cm = fmaxf(cm, sS[row * KC + sub + i * 16]);
      cm = row16_max(cm);
      const float mo = sM[row];
      const float mn = fmaxf(mo, cm);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < KC / 16; ++i) {
        const float sv = sS[row * KC + sub + i * 16];
        const float p = (sv == P5_NEG_INF) ? 0.f : expf(sv - mn);
        sS[row * KC + sub + i * 16] = p;
        sum += p;
      }
      sum = row16_sum(sum);
      const float sc = (mo == P5_NEG_INF) ? 0.f : expf(mo - mn);
      __syncthreads();       // every thread of the row has read sM[row]
      if (sub == 0) { sM[row] = mn; sM[16 + row] = sM[16 + row] * sc + sum; sM[32 + row] = sc; }
    }
    __syncthreads();
    {
      const float sc = sM[32 + orow];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] *= sc;
      const int nk = (a.L - j0) < KC ? (a.L - j0) : KC;
      const int byte0 = od * (int)sizeof(T);
      const int stepc = byte0 / 128, slot = (byte0 % 128) >> 4, within = byte0 & 15;
      for (int j = 0; j < nk; ++j) {
        const float p = sS[orow * KC + j];
        const char* vp = vimg + stepc * KC * 128 + j * 128 + ((slot ^ (j & 7)) << 4) + within;
        float v4[4];
        if constexpr (sizeof(T) == 2) {
          const u32x2 raw = *(const u32x2*)vp;
          union { unsigned u; float f; } c0, c1, c2, c3;
          c0.u = raw[0] << 16; c1.u = raw[0] & 0xFFFF0000u; c2.u = raw[1] << 16; c3.u = raw[1] & 0xFFFF0000u;
          v4[0] = c0.f; v4[1] = c1.f; v4[2] = c2.f; v4[3] = c3.f;
        } else {
          const f32x4 raw = *(const f32x4*)vp;
          v4[0] = raw[0]; v4[1] = raw[1]; v4[2] = raw[2]; v4[3] = raw[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += p * v4[e];
      }
    }
  }
  __syncthreads();
  if (orow < nrow) {
    const float l = sM[16 + orow];
    const float inv = l > 0.f ? 1.f / l : 0.f;
    T* op = (T*)a.out + (size_t)(r0 + orow) * inner + h * 64 + od;
#pragma unroll
    for (int e = 0; e < 4; ++e) op[e] = from_f<T>(o[e] * inv);
  }
}

// =====================================================================================================================
// Cross-attention of the beams of one batch item for one head on the matrix cores (replaces the scalar score / PV loops of
// p5_dec_cross_attn2_kernel): S = q K^T and O = P V are MFMA tiles -- the <= 16 beams of an item are exactly one 16-row
// tile, which is why the item's K/V are staged once and shared (the reference expands encoder states x num_beams,
// P5_T5.py:571-576).  Zero position bias, encoder padding mask, fp32 online softmax over 128-key chunks
// (HF modeling_t5.py:336-343,404-432).  FUSEQ: the beams' q projection (T5LayerNorm of the fp32 residual rows, then the head's
// 64 rows of Wq) is computed here too, all operand bytes requested up front.
//   LDS: q image [16][64] T (kc128 layout) | S fp32 [16][KC+4] | P (T) [16][KC*sizeof(T)+16 B] (x2 in bf16) | stats | K image | V image
//        | (FUSEQ) normalised rows + Wq slice
// =====================================================================================================================
template <class T, bool FUSEQ, int LDSKB>
__global__ __launch_bounds__(256) void p5_dec_cross_attn3_kernel(P5CrossArgs a) {
  constexpr int KC = 128;                              // keys per chunk
  constexpr int SZ = (int)sizeof(T), EPF = TT<T>::EPF, KCH = TT<T>::KCH;
  constexpr int NSD = 64 * SZ / 128;                   // 128-byte steps per 64-dim K/V/q row (1 bf16, 2 fp32)
  constexpr int KVB = KC * 128 * NSD;                  // bytes of one K (or V) chunk image
  constexpr int PROW = KC * SZ + 16;                   // padded row pitch of the P image (bytes)
  __shared__ __attribute__((aligned(16))) char lds[LDSKB * 1024];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt_per = (a.Kb + 15) / 16;
  const int b = blockIdx.x / mt_per, mt = blockIdx.x % mt_per, h = blockIdx.y;
  const int r0 = b * a.Kb + mt * 16;
  const int nrow = (a.Kb - mt * 16) < 16 ? (a.Kb - mt * 16) : 16;
  const int inner = a.H * 64;
  char* qimg = lds;                                     // [NSD][16][128 B]
  float* sS = (float*)(qimg + NSD * 2048);              // [16][KC + 4]
  char* pimg = (char*)(sS + 16 * (KC + 4));             // [16][PROW]; bf16: a second image holds the rounding residual of P
  constexpr int NPI = SZ == 2 ? 2 : 1;
  float* sM = (float*)(pimg + NPI * 16 * PROW);         // [16] max, [16] sum, [16] rescale, [16] spare
  char* kimg = (char*)(sM + 64);
  char* vimg = kimg + KVB;
  char* aimg = vimg + KVB;
  const T* kvp = (const T*)a.kv + (size_t)b * a.L * a.ldkv + h * 64;
  auto stage_kv = [&](int j0) {
    sk_dma_rows<T>(kimg, kvp, a.ldkv, j0, a.L, 0, KC, NSD, tid);
    sk_dma_rows<T>(vimg, kvp + inner, a.ldkv, j0, a.L, 0, KC, NSD, tid);
  };
  stage_kv(0);
  // ---- q rows -> qimg (as the activation dtype stores them) ----
  auto q_store = [&](int row, int col, float v) {
    const int byte = col * SZ, step = byte / 128, slot = (byte % 128) >> 4, within = byte & 15;
    *(T*)(qimg + step * 2048 + row * 128 + ((slot ^ (row & 7)) << 4) + within) = from_f<T>(v);
  };
  if constexpr (FUSEQ) {
    constexpr int EPS = SkT<T>::EPS;
    const int nsteps = a.d / EPS;
    char* bimg = aimg + (size_t)nsteps * 2048;
    float* red = (float*)(bimg + (size_t)nsteps * 64 * 128);
    sk_dma_rows<T>(bimg, (const T*)a.Wq, a.d, h * 64, inner, 0, 64, nsteps, tid);
    sk_norm_rows<T>(aimg, a.x, a.ln, r0, r0 + nrow, a.d, a.eps, tid);
    __syncthreads();
    const f32x4 acc = sk_mma<T, 64>(aimg, bimg, nsteps, red, tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) q_store((lane >> 4) * 4 + r, wave * 16 + (lane & 15), acc[r]);
  } else {
    for (int i = tid; i < 16 * 64; i += 256) {
      const int row = i >> 6;
      q_store(row, i & 63, row < nrow ? to_f<T>(((const T*)a.q)[(size_t)(r0 + row) * inner + h * 64 + (i & 63)]) : 0.f);
    }
  }
  if (tid < 16) { sM[tid] = P5_NEG_INF; sM[16 + tid] = 0.f; }
  f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};               // wave w owns dims [16w, 16w+16): rows (lane>>4)*4 + r, dim 16w + (lane&15)
  for (int j0 = 0; j0 < a.L; j0 += KC) {
    if (j0 > 0) { __syncthreads(); stage_kv(j0); }
    __syncthreads();                                     // K/V chunk landed (and q image / stats written)
    // ---- S = q K^T : 8 key tiles of 16, two per wave ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kt = wave * 2 + t;
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NSD; ++st) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 fa = frag_load_kc128<T>(qimg + st * 2048, 0, c, lane);
          const u32x4 fb = frag_load_kc128<T>(kimg + st * KC * 128, kt * 16, c, lane);
          mma16<T>(s, fa, fb);
        }
      }
      const int key = kt * 16 + (lane & 15);
      const bool ok = (j0 + key) < a.L && a.mask[(size_t)b * a.L + j0 + key] != 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) sS[((lane >> 4) * 4 + r) * (KC + 4) + key] = ok ? s[r] : P5_NEG_INF;
    }
    __syncthreads();
    {
      // online softmax: 16 threads per row, 8 keys each; probabilities go to the P image in the compute dtype
      const int row = tid >> 4, sub = tid & 15;
      float sv[KC / 16];
      float cm = P5_NEG_INF;
#pragma unroll
      for (int i = 0; i < KC / 16; ++i) { sv[i] = sS[row * (KC + 4) + sub + i * 16]; cm = fmaxf(cm, sv[i]); }
      cm = row16_max(cm);
      const float mo = sM[row];
      const float mn = fmaxf(mo, cm);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < KC / 16; ++i) {
        const float p = (sv[i] == P5_NEG_INF) ? 0.f : expf(sv[i] - mn);
        const T pt = from_f<T>(p);
        *(T*)(pimg + row * PROW + (sub + i * 16) * SZ) = pt;
        // bf16: P = hi + lo keeps 16 mantissa bits of the probabilities (the scalar kernel multiplied fp32 P with V; a
        // single bf16 P alone costs 2e-3 of score accuracy, as much as everything else in the bf16 decode step together)
        if constexpr (SZ == 2) *(T*)(pimg + 16 * PROW + row * PROW + (sub + i * 16) * SZ) = from_f<T>(p - to_f<T>(pt));
        sum += p;
      }
      sum = row16_sum(sum);
      const float sc = (mo == P5_NEG_INF) ? 0.f : expf(mo - mn);
      __syncthreads();       // every thread of the row has read sM[row]
      if (sub == 0) { sM[row] = mn; sM[16 + row] = sM[16 + row] * sc + sum; sM[32 + row] = sc; }
    }
    __syncthreads();
    // ---- O = O * rescale + P V : wave w -> dims [16w, 16w+16); V fragments gathered from the key-major image ----
    {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] *= sM[32 + (lane >> 4) * 4 + r];
      const int dim = wave * 16 + (lane & 15);
      const int byte = dim * SZ, vstep = byte / 128, vslot = (byte % 128) >> 4, vwithin = byte & 15;
#pragma unroll
      for (int kc = 0; kc < KC / KCH; ++kc) {
        const u32x4 fa = ld16(pimg + (lane & 15) * PROW + (kc * KCH + (lane >> 4) * EPF) * SZ);
        T vb[EPF];
#pragma unroll
        for (int e = 0; e < EPF; ++e) {
          const int key = kc * KCH + (lane >> 4) * EPF + e;
          vb[e] = *(const T*)(vimg + vstep * KC * 128 + key * 128 + ((vslot ^ (key & 7)) << 4) + vwithin);
        }
        u32x4 fb;
        if constexpr (SZ == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) fb[q] = (unsigned)((const unsigned short*)vb)[2 * q] | ((unsigned)((const unsigned short*)vb)[2 * q + 1] << 16);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) fb[q] = ((const unsigned*)vb)[q];
        }
        mma16<T>(o, fa, fb);
        if constexpr (SZ == 2) {
          const u32x4 fl = ld16(pimg + 16 * PROW + (lane & 15) * PROW + (kc * KCH + (lane >> 4) * EPF) * SZ);
          mma16<T>(o, fl, fb);
        }
      }
    }
  }
  // masked keys carry P = 0 and clamped rows finite V, so nothing needs fixing up; rows >= nrow are not stored
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) * 4 + r;
    if (row < nrow) {
      const float l = sM[16 + row];
      const float inv = l > 0.f ? 1.f / l : 0.f;
      ((T*)a.out)[(size_t)(r0 + row) * inner + h * 64 + wave * 16 + (lane & 15)] = from_f<T>(o[r] * inv);
    }
  }
}

// =====================================================================================================================
// Tied head as a streaming reduction (HF generation/utils.py:3388-3389 takes log_softmax over the FULL
// vocabulary and only then masks to the trie children): every workgroup keeps NV rows of E in LDS (fetched once, direct to
// LDS), multiplies ALL decode rows against them and reduces each row's NV logits to (max, sum exp) on the spot -- the
// [R, V] logits are never written.  The few logits the search needs (the trie children of each beam) are recomputed as dot
// products by p5_beam_step2_kernel.  Waves split the rows (m-tiles of 16), A fragments come straight from global memory
// (hn is R x d, L2/L1 resident), so there is no barrier after the one that publishes the E tile.
//   grid = ceil(V / NV), 256 threads.  part_m / part_s: [R][gridDim.x].
// =====================================================================================================================
template <class T, int NV, int LDSKB>
__global__ __launch_bounds__(256) void p5_head_lse_kernel(float* __restrict__ part_m, float* __restrict__ part_s, const T* __restrict__ hn,
                                                         const T* __restrict__ E, int R, int d, int V, float alpha,
                                                         const int* __restrict__ done) {
  constexpr int EPS = SkT<T>::EPS, EPF = TT<T>::EPF, KCH = TT<T>::KCH, NT = NV / 16;
  __shared__ __attribute__((aligned(16))) char lds[LDSKB * 1024];
  if (done && *done) return;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the unit loop below branches on scalars
#endif
  const int v0 = blockIdx.x * NV;
  const int nsteps = d / EPS;                                        // 128-byte steps per row
  sk_dma_rows<T>(lds, E, d, v0, V, 0, NV, nsteps, tid);              // rows past V are clamped (masked below)
  __syncthreads();
  const int nmt = (R + 15) / 16;
  const int nkc = d / KCH;                                           // 64-byte chunks per row; two per 128-byte step
  // B fragment addresses: row (lane & 15) of an n-tile, 16-byte slot ((chunk << 2 | lane >> 4) ^ (row & 7)); the swizzle only
  // involves lane bits, so the two chunks of a step sit at two per-lane constants and everything else is an immediate.
  const int off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
  const int off1 = (lane & 15) * 128 + (((4 | (lane >> 4)) ^ (lane & 7)) << 4);
  // A fragments come straight from global memory (hn is L2-resident), one UNIT = 8 chunks (4 steps) of one m-tile at a time,
  // through a ring of four register buffers: a unit's loads are issued three units (~190 MFMAs) before its first use -- a
  // load's latency (1-2 us under load) is ~10 x the MFMA time of a step.  The loads are raw (gload16_raw) and waited for by
  // count: with compiler-tracked loads hipcc waits at every loop iteration for ALL loads in flight, including the ones just
  // issued for later units (ISA: s_waitcnt vmcnt(7) right behind eight new loads), i.e. no lookahead at all.
  // d_model % (8 * KCH) == 0 for every T5 size, so units are never partial.
  constexpr int UC = 8;
  const int upm = nkc / UC;                                          // units per m-tile
  const int my_mt = wave < nmt ? (nmt - wave + 3) / 4 : 0;           // m-tiles of this wave: wave, wave + 4, ...
  const int nunits = my_mt * upm;
  auto load_unit = [&](u32x4 (&f)[UC], int u) {
    const int mt_ = wave + 4 * (u / upm), hf = u % upm;
    int ar = mt_ * 16 + (lane & 15);
    ar = ar < R ? ar : R - 1;
    const T* ap = hn + (size_t)ar * d + (lane >> 4) * EPF + (size_t)hf * UC * KCH;
#pragma unroll
    for (int i = 0; i < UC; ++i) gload16_raw(f[i], ap + (size_t)i * KCH);
  };
  f32x4 acc[NT];
  auto epilogue = [&](int mt) {
    // row-wise (max, sum exp) over this tile's NV columns: element (row (lane>>4)*4 + r, col n*16 + (lane&15))
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = P5_NEG_INF;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bool ok = v0 + n * 16 + (lane & 15) < V;
        acc[n][r] = ok ? acc[n][r] * alpha : P5_NEG_INF;
        m = fmaxf(m, acc[n][r]);
      }
      m = row16_max(m);
      float s = 0.f;
#pragma unroll
      // (fast mode: exp2-based __expf, 2 ulp -- the precise expf is ~30 instructions and, at 32 calls per lane per m-tile, was most
      //  of this kernel: 53 % of its wave cycles were instruction issue with the matrix pipe 8 % busy)
      for (int n = 0; n < NT; ++n) s += (acc[n][r] == P5_NEG_INF) ? 0.f : p5_exp<T>(acc[n][r] - m);
      s = row16_sum(s);
      const int row = mt * 16 + (lane >> 4) * 4 + r;
      if ((lane & 15) == 0 && row < R) {
        part_m[(size_t)row * gridDim.x + blockIdx.x] = m;
        part_s[(size_t)row * gridDim.x + blockIdx.x] = s;
      }
    }
  };
  auto compute = [&](const u32x4 (&f)[UC], int u) {
    const int mt = wave + 4 * (u / upm), hf = u % upm;
    // this unit's eight loads are followed by those of up to three later units (and possibly by an epilogue's stores, which
    // only makes the wait stricter than necessary)
    const int ahead = nunits - 1 - u;
    if (ahead >= 3) P5_WAIT_VM(24);
    else if (ahead == 2) P5_WAIT_VM(16);
    else if (ahead == 1) P5_WAIT_VM(8);
    else P5_WAIT_VM(0);
    P5_SCHED_FENCE();
    if (hf == 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // B fragments of chunk i+1 are read from LDS while the MFMAs of chunk i issue (one wave per SIMD: nobody else hides the
    // ~100-cycle LDS latency; the compiler's own order was read, wait, MFMA, read, wait, MFMA)
    const char* bs0 = lds + (size_t)(hf * (UC / 2)) * NV * 128;
    u32x4 bq[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bq[0][n] = ld16(bs0 + n * 2048 + off0);
#pragma unroll
    for (int i = 0; i < UC; ++i) {
      if (i + 1 < UC) {
        const char* bn = bs0 + (size_t)((i + 1) >> 1) * NV * 128 + (((i + 1) & 1) ? off1 : off0);
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[(i + 1) & 1][n] = ld16(bn + n * 2048);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) mma16<T>(acc[n], f[i], bq[i & 1][n]);
    }
    if (hf == upm - 1) epilogue(mt);
  };
  u32x4 f0[UC], f1[UC], f2[UC], f3[UC];
  if (nunits > 0) load_unit(f0, 0);
  if (nunits > 1) load_unit(f1, 1);
  if (nunits > 2) load_unit(f2, 2);
  for (int u = 0; u < nunits; u += 4) {
    if (u + 3 < nunits) load_unit(f3, u + 3);
    compute(f0, u);
    if (u + 1 >= nunits) break;
    if (u + 4 < nunits) load_unit(f0, u + 4);
    compute(f1, u + 1);
    if (u + 2 >= nunits) break;
    if (u + 5 < nunits) load_unit(f1, u + 5);
    compute(f2, u + 2);
    if (u + 3 >= nunits) break;
    if (u + 6 < nunits) load_unit(f2, u + 6);
    compute(f3, u + 3);
  }
}
