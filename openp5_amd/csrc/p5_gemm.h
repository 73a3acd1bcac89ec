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
// Tile: BM x BN per 256-thread workgroup (4 waves as 2x2), one 64-byte K-chunk per step, double-buffered
// in LDS with the next chunk's global loads in flight during the MFMAs.
#pragma once
#include "p5_device.h"
#include "p5_rng.h"

enum P5Epi : int {
  P5_EPI_STORE = 0,       // C = acc * alpha
  P5_EPI_RELU_DROP = 1,   // C = drop(relu(acc))
  P5_EPI_RESID_DROP = 2,  // C = aux + drop(acc)            (aux: residual stream, same dtype as C)
  P5_EPI_MASK_POS = 3,    // C = aux > 0 ? acc * alpha : 0  (relu/dropout backward through saved hidden)
  P5_EPI_ATOMIC = 4,      // C += acc * alpha  (fp32 atomics; split-K wgrad)
  P5_EPI_GELU_GATE = 5,   // reserved (gated-gelu epilogue handled by an elementwise kernel in v1)
  P5_EPI_ACCUM = 6,       // C += acc * alpha  (fp32, exclusive ownership: no split-K)
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
  float alpha;
  P5Drop drop;
};

template <class T, bool KS> struct LdsTile {
  // byte layout of one operand tile of R rows x 64B of K
  template <int R> static constexpr int bytes() {
    return KS ? TT<T>::KCH * (R * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16)) : R * 64;
  }
};

__device__ static __forceinline__ int kc_off(int row, int kc) {
  // XOR swizzle that makes every ds_read_b128 lane group hit 16 distinct 16-byte slots
  const int h = (0x1230 >> (((row >> 2) & 3) * 4)) & 3;  // h = [0,3,2,1]
  return row * 64 + ((kc ^ h) << 4);
}

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
    constexpr int STRIDE = R * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      st16(lds + (c / CPR) * STRIDE + (c % CPR) * 16, regs[i]);
    }
  }
}

// one 16-row fragment (rows t0..t0+15 of the tile) for this lane
template <class T, int R, bool KS>
__device__ static __forceinline__ u32x4 frag_load(const char* lds, int t0, int lane) {
  if constexpr (!KS) {
    return ld16(lds + kc_off(t0 + (lane & 15), lane >> 4));
  } else {
    constexpr int STRIDE = R * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
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

template <class T, int BM, int BN, bool AKS, bool BKS>
__global__ __launch_bounds__(256) void p5_gemm_kernel(P5GemmArgs g) {
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int KCH = TT<T>::KCH;
  constexpr int ABYTES = LdsTile<T, AKS>::template bytes<BM>();
  constexpr int BBYTES = LdsTile<T, BKS>::template bytes<BN>();
  constexpr int NA = AKS ? (KCH * (BM / TT<T>::EPF) / 256) : (BM * 4 / 256);
  constexpr int NB = BKS ? (KCH * (BN / TT<T>::EPF) / 256) : (BN * 4 / 256);
  __shared__ __attribute__((aligned(16))) char lds[2 * (ABYTES + BBYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // split-K range (in K-chunks)
  const int nkc = (g.K + KCH - 1) / KCH;
  const int per = (nkc + g.splitk - 1) / g.splitk;
  const int kc_begin = blockIdx.z * per;
  const int kc_end = (kc_begin + per < nkc) ? kc_begin + per : nkc;
  if (kc_begin >= kc_end) return;

  const T* __restrict__ A = (const T*)g.A;
  const T* __restrict__ Bp = (const T*)g.B;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[NA], rb[NB];
  stage_load<T, BM, AKS>(ra, A, g.lda, m0, kc_begin * KCH, g.M, g.K, tid);
  stage_load<T, BN, BKS>(rb, Bp, g.ldb, n0, kc_begin * KCH, g.N, g.K, tid);
  stage_store<T, BM, AKS>(ra, lds, tid);
  stage_store<T, BN, BKS>(rb, lds + ABYTES, tid);
  __syncthreads();

  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int cur = (kc - kc_begin) & 1;
    char* la = lds + cur * (ABYTES + BBYTES);
    char* lb = la + ABYTES;
    const bool more = (kc + 1 < kc_end);
    if (more) {
      stage_load<T, BM, AKS>(ra, A, g.lda, m0, (kc + 1) * KCH, g.M, g.K, tid);
      stage_load<T, BN, BKS>(rb, Bp, g.ldb, n0, (kc + 1) * KCH, g.N, g.K, tid);
    }
    u32x4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = frag_load<T, BM, AKS>(la, wm * (BM / 2) + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = frag_load<T, BN, BKS>(lb, wn * (BN / 2) + j * 16, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
    if (more) {
      char* na = lds + (cur ^ 1) * (ABYTES + BBYTES);
      stage_store<T, BM, AKS>(ra, na, tid);
      stage_store<T, BN, BKS>(rb, na + ABYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns C[row = (lane>>4)*4 + r][col = lane & 15] of every 16x16 tile ----
  const uint32_t seed = p5_seed(g.drop);
  const bool do_drop = g.drop.state != nullptr && g.drop.thr != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
        if (row >= g.M || col >= g.N) continue;
        float v = acc[i][j][r] * g.alpha;
        const size_t ci = (size_t)row * g.ldc + col;
        if (g.epi == P5_EPI_RELU_DROP) {
          v = v > 0.f ? v : 0.f;
          if (do_drop) v = p5_keep(seed, g.drop.site_key, (uint32_t)(row * g.N + col), g.drop.thr) ? v * g.drop.scale : 0.f;
        } else if (g.epi == P5_EPI_RESID_DROP) {
          if (do_drop) v = p5_keep(seed, g.drop.site_key, (uint32_t)(row * g.N + col), g.drop.thr) ? v * g.drop.scale : 0.f;
          v += to_f<T>(((const T*)g.aux)[(size_t)row * g.ldaux + col]);
        } else if (g.epi == P5_EPI_MASK_POS) {
          v = to_f<T>(((const T*)g.aux)[(size_t)row * g.ldaux + col]) > 0.f ? v : 0.f;
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
