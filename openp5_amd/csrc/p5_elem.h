// p5_elem.h -- HBM-bound row kernels of the T5 path: embedding gather(+whole-word add), T5 RMSNorm
// forward/backward, fp32->T mask/cast, embedding-gradient scatter, token cross-entropy on materialised logits,
// fused clip + HF-AdamW over the flat parameter arena.  All of them move 16 bytes per lane per access and use
// one 64-lane wave per row (d_model <= 1024 -> <= 2 pieces per lane in bf16, 4 in f32).
#pragma once
#include "p5_device.h"
#include "p5_rng.h"

// ---------------------------------------------------------------------------------------------------------
// K1: x[row,:] = E[ids[row],:] (+ WW[ww[row],:]), dropout.   P5_T5.py:94-100,125 (JointEncoder), decoder: E only
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void p5_embed_fwd_kernel(T* __restrict__ out, const T* __restrict__ E, const T* __restrict__ WW,
                                                          const int64_t* __restrict__ ids, const int64_t* __restrict__ ww,
                                                          int rows, int d, P5Drop drop, float* __restrict__ ssq_part = nullptr) {
  constexpr int EPF = TT<T>::EPF;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t id = ids[row];
  const int64_t w = WW ? ww[row] : 0;
  const bool do_drop = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
  for (int c = lane; c < d / EPF; c += 64) {
    float x[8], y[8];
    unpack16<T>(ld16(E + (size_t)id * d + c * EPF), x);
    if (WW) {
      unpack16<T>(ld16(WW + (size_t)w * d + c * EPF), y);
#pragma unroll
      for (int e = 0; e < EPF; ++e) x[e] += y[e];
    }
    if (do_drop) {
#pragma unroll
      for (int e = 0; e < EPF; ++e)
        x[e] = p5_keep(seed, drop.site_key, (uint32_t)(row * d + c * EPF + e), drop.thr) ? x[e] * drop.scale : 0.f;
    }
    const u32x4 packed = pack16<T>(x);
    st16(out + (size_t)row * d + c * EPF, packed);
    if (ssq_part) {      // (uniform) T5LayerNorm statistic of the row as stored, per 64-column group (p5_gemm.h `rowss_nt`): 64 / EPF adjacent lanes
      float w[8], ss = 0.f;
      unpack16<T>(packed, w);
#pragma unroll
      for (int e = 0; e < EPF; ++e) ss += w[e] * w[e];
#pragma unroll
      for (int m = 64 / EPF / 2; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
      if ((lane % (64 / EPF)) == 0) ssq_part[(size_t)row * (d / 64) + (c * EPF) / 64] = ss;
    }
  }
}

// dE[ids[row],:] += mask(dres[row,:]);  dWW[ww[row],:] += same   (SURVEY.md App. C "Embedding lookups"); either table may be
// nullptr (the two scatters of the encoder run on two streams at the end of the backward, where nothing else is left to overlap)
template <class T>
__global__ __launch_bounds__(256) void p5_embed_bwd_kernel(float* __restrict__ dE, float* __restrict__ dWW,
                                                          const float* __restrict__ dres, const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ ww, int rows, int d, P5Drop drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t id = dE ? ids[row] : 0;
  const int64_t w = dWW ? ww[row] : 0;
  const bool do_drop = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
  // the row's values are all requested before the first atomic (an atomic per dependent load ran at 40 % of the atomic rate of the
  // part): 16 columns per lane per pass
  for (int c0 = 0; c0 < d; c0 += 64 * 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = c0 + k * 64 + lane;
      v[k] = c < d ? dres[(size_t)row * d + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = c0 + k * 64 + lane;
      if (c >= d) continue;
      float x = v[k];
      if (do_drop) x = p5_keep(seed, drop.site_key, (uint32_t)(row * d + c), drop.thr) ? x * drop.scale : 0.f;
      if (dE) atomicAdd(dE + (size_t)id * d + c, x);
      if (dWW) atomicAdd(dWW + (size_t)w * d + c, x);
    }
  }
}

// EPF (4 or 8) consecutive fp32 values as 16-byte accesses (the compiler otherwise emits one dword access per element,
// each touching every cache line of the row again)
template <int EPF> __device__ static __forceinline__ void ldf(const float* __restrict__ p, float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < EPF / 4; ++q) {
    const f32x4 t = *(const f32x4*)(p + 4 * q);
    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
  }
}
template <int EPF> __device__ static __forceinline__ void stf(float* __restrict__ p, const float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < EPF / 4; ++q) *(f32x4*)(p + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

// ---------------------------------------------------------------------------------------------------------
// K2: T5LayerNorm (HF modeling_t5.py:59-72): y = w * x * rsqrt(mean(x^2) + eps), fp32 statistics.
// Optional dropout on y (final norms, P5_T5.py:179-180).  Saves rstd for the backward.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void p5_rmsnorm_fwd_kernel(T* __restrict__ y, float* __restrict__ rstd_out, const T* __restrict__ x,
                                                            const float* __restrict__ w, int rows, int d, float eps, P5Drop drop) {
  constexpr int EPF = TT<T>::EPF;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int npc = d / EPF;
  float xv[4][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < npc) {
      unpack16<T>(ld16(x + (size_t)row * d + c * EPF), xv[i]);
#pragma unroll
      for (int e = 0; e < EPF; ++e) ss += xv[i][e] * xv[i][e];
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)d + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  const bool do_drop = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < npc) {
      float o[8], wv[8];
      ldf<EPF>(w + c * EPF, wv);
#pragma unroll
      for (int e = 0; e < EPF; ++e) {
        // reference order: (x * rstd) rounded to the activation dtype, then * weight
        const float n = to_f<T>(from_f<T>(xv[i][e] * rstd));
        o[e] = wv[e] * n;
        if (do_drop) o[e] = p5_keep(seed, drop.site_key, (uint32_t)(row * d + c * EPF + e), drop.thr) ? o[e] * drop.scale : 0.f;
      }
      st16(y + (size_t)row * d + c * EPF, pack16<T>(o));
    }
  }
}

// Backward:  xh = x*rstd;  dw += sum_rows dy*xh;  dx = rstd*(dy*w - xh*mean_j(dy_j w_j xh_j))
//   dres_out = (dres_in ? dres_in : 0) + dx                         (fp32 residual-stream gradient)
//   dy_next  = dropmask_next(dres_out) cast to T                     (input to the PRECEDING sub-layer's dgrad)
// drop_in masks the incoming dy (only for the final norms whose output is dropped).
// Each workgroup walks rows with a grid stride and flushes its dw partials with one atomicAdd per column.
// NCH = 16-byte pieces per lane (d_model <= NCH * 64 * EPF): a compile-time bound keeps the per-row state in exactly as many
// registers as the model needs (bf16 d_model 512: one piece), which is what lets a wave keep TWO rows in flight -- the kernel
// is a chain of dependent HBM round trips per row, so rows in flight per wave is its throughput.
template <class T, int NCH>
__global__ __launch_bounds__(256) void p5_rmsnorm_bwd_kernel(float* __restrict__ dres_out, T* __restrict__ dy_next,
                                                            float* __restrict__ dw, const T* __restrict__ dy,
                                                            const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ rstd_in, const float* __restrict__ dres_in,
                                                            int rows, int d, P5Drop drop_in, P5Drop drop_next, float* __restrict__ dw_partial,
                                                            const float* __restrict__ ssq_part = nullptr, T* __restrict__ n_out = nullptr, float eps = 0.f) {
  constexpr int EPF = TT<T>::EPF;
  constexpr int RU = NCH <= 2 ? 2 : 1;             // rows in flight per wave
  __shared__ float sdw[4][NCH * 64 * EPF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npc = d / EPF;
  float dwacc[NCH][8], wv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwacc[i][e] = 0.f; wv[i][e] = 0.f; }
    if (c < npc) ldf<EPF>(w + c * EPF, wv[i]);
  }
  const bool din = drop_in.state != nullptr && drop_in.thr != 0;
  const bool dnx = drop_next.state != nullptr && drop_next.thr != 0;
  const uint32_t seed_in = p5_seed(drop_in), seed_nx = p5_seed(drop_next);
  const int stride = gridDim.x * 4;

  for (int row0 = blockIdx.x * 4 + wave; row0 < rows; row0 += stride * RU) {
    float dyv[RU][NCH][8], xh[RU][NCH][8], rin[RU][NCH][8], rstd[RU];
    // every load of the rows is issued before the first reduction
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * stride;
      if (ssq_part) {      // (uniform) the forward carried the statistic as d/64 partial sums of squares per row (norm folded into the GEMMs)
        float ss = 0.f;
        if (row < rows) {
          const float* sp = ssq_part + (size_t)row * (d / 64);
          if (((d / 64) & 3) == 0) {
            for (int t = 0; t < d / 64; t += 4) {
              const f32x4 v = *(const f32x4*)(sp + t);
              ss = (((ss + v[0]) + v[1]) + v[2]) + v[3];
            }
          } else {
            for (int t = 0; t < d / 64; ++t) ss += sp[t];
          }
        }
        rstd[u] = row < rows ? rsqrtf(ss / (float)d + eps) : 0.f;
      } else {
        rstd[u] = row < rows ? rstd_in[row] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < npc && row < rows) {
          unpack16<T>(ld16(dy + (size_t)row * d + c * EPF), dyv[u][i]);
          unpack16<T>(ld16(x + (size_t)row * d + c * EPF), xh[u][i]);
          if (dres_in) ldf<EPF>(dres_in + (size_t)row * d + c * EPF, rin[u][i]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * stride;
      if (row >= rows) continue;                    // (wave-uniform)
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < npc) {
          float nrm[8];
#pragma unroll
          for (int e = 0; e < EPF; ++e) {
            if (din) dyv[u][i][e] = p5_keep(seed_in, drop_in.site_key, (uint32_t)(row * d + c * EPF + e), drop_in.thr) ? dyv[u][i][e] * drop_in.scale : 0.f;
            xh[u][i][e] *= rstd[u];
            if (n_out) nrm[e] = wv[i][e] * to_f<T>(from_f<T>(xh[u][i][e]));     // the forward norm's output, which the folded forward never wrote
            dwacc[i][e] += dyv[u][i][e] * xh[u][i][e];
            dyv[u][i][e] *= wv[i][e];
            dot += dyv[u][i][e] * xh[u][i][e];
          }
          if (n_out) st16(n_out + (size_t)row * d + c * EPF, pack16<T>(nrm));
        }
      }
      dot = wave_sum(dot) / (float)d;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < npc) {
          float o[8], rout[8];
          const size_t g0 = (size_t)row * d + c * EPF;
#pragma unroll
          for (int e = 0; e < EPF; ++e) {
            float v = rstd[u] * (dyv[u][i][e] - xh[u][i][e] * dot);
            if (dres_in) v += rin[u][i][e];
            rout[e] = v;
            o[e] = dnx ? (p5_keep(seed_nx, drop_next.site_key, (uint32_t)(g0 + e), drop_next.thr) ? v * drop_next.scale : 0.f) : v;
          }
          stf<EPF>(dres_out + g0, rout);
          if (dy_next) st16(dy_next + (size_t)row * d + c * EPF, pack16<T>(o));
        }
      }
    }
  }
  // flush dw
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < npc) {
#pragma unroll
      for (int e = 0; e < EPF; ++e) sdw[wave][c * EPF + e] = dwacc[i][e];
    }
  }
  __syncthreads();
  if (dw_partial) {   // one row of partial sums per workgroup, reduced later by p5_reduce_rows_kernel (no atomics)
    for (int j = threadIdx.x; j < d; j += 256) dw_partial[(size_t)blockIdx.x * d + j] = sdw[0][j] + sdw[1][j] + sdw[2][j] + sdw[3][j];
    return;
  }
  // every workgroup starts its flush at a different column so that concurrent workgroups hit different addresses
  for (int j0 = threadIdx.x; j0 < d; j0 += 256) {
    const int j = (j0 + (int)(blockIdx.x % 8u) * 64) % d;
    const float v = sdw[0][j] + sdw[1][j] + sdw[2][j] + sdw[3][j];
    if (v != 0.f) atomicAdd(dw + j, v);
  }
}

// out(T) = dropmask(in fp32)  -- top of the encoder backward (grad of drop(final_norm(x)) arrives in fp32)
template <class T>
__global__ __launch_bounds__(256) void p5_cast_mask_kernel(T* __restrict__ out, const float* __restrict__ in, size_t n, P5Drop drop) {
  constexpr int EPF = TT<T>::EPF;      // n is a multiple of d_model, hence of EPF
  const bool dd = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPF; i < n; i += (size_t)gridDim.x * 256 * EPF) {
    float v[8];
    ldf<EPF>(in + i, v);
    if (dd) {
#pragma unroll
      for (int e = 0; e < EPF; ++e) v[e] = p5_keep(seed, drop.site_key, (uint32_t)(i + e), drop.thr) ? v[e] * drop.scale : 0.f;
    }
    st16(out + i, pack16<T>(v));
  }
}

// out[i] = T(sum_{z < nsplit} part[z * stride + i]), summed in index order: the second half of a deterministic split-K GEMM
// (P5GemmArgs::c_split_stride) fused with the cast to the compute dtype
template <class T>
__global__ __launch_bounds__(256) void p5_reduce_splits_kernel(T* __restrict__ out, const float* __restrict__ part, int nsplit, size_t stride, size_t n) {
  constexpr int EPF = TT<T>::EPF;      // n is a multiple of d_model, hence of EPF
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPF; i < n; i += (size_t)gridDim.x * 256 * EPF) {
    float acc[8], v[8];
    ldf<EPF>(part + i, acc);
    for (int z = 1; z < nsplit; ++z) {
      ldf<EPF>(part + (size_t)z * stride + i, v);
#pragma unroll
      for (int e = 0; e < EPF; ++e) acc[e] += v[e];
    }
    st16(out + i, pack16<T>(acc));
  }
}

// dst[j] += sum_b partial[b][j]  (per-workgroup norm-weight gradient partials).  DETERMINISTIC: one workgroup owns 16 columns and
// sums ALL partial rows in a fixed association -- 16 row classes (b mod 16) summed sequentially in registers, then the classes in index
// order -- and is the only writer of its columns (plain read-modify-write, no atomics): the same bits on every run.  (Rounds 1-3 split
// the rows over 16 workgroups that added with fp32 atomics; their arrival order made the last bits of every T5LayerNorm gradient vary
// from run to run, and AdamW turns a last-bit difference of a near-zero gradient into an O(lr) difference of the parameter.)
// grid = (d/16); 256 threads = 16 row classes x 16 columns.
__device__ static __forceinline__ void reduce_rows_16col(float* __restrict__ dst, const float* __restrict__ partial, int nrows, int d, int cg) {
  const int col = cg * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
  __shared__ float sred[16][17];
  float s = 0.f;
  if (col < d) {
    int b = part;
    for (; b + 48 < nrows; b += 64) {      // four independent loads in flight, added in row order
      const float v0 = partial[(size_t)b * d + col], v1 = partial[(size_t)(b + 16) * d + col];
      const float v2 = partial[(size_t)(b + 32) * d + col], v3 = partial[(size_t)(b + 48) * d + col];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; b < nrows; b += 16) s += partial[(size_t)b * d + col];
  }
  sred[part][threadIdx.x & 15] = s;
  __syncthreads();
  if (part == 0 && col < d) {
    float v = sred[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 16; ++q) v += sred[q][threadIdx.x];
    dst[col] += v;
  }
}
__global__ __launch_bounds__(256) void p5_reduce_rows_kernel(float* __restrict__ dst, const float* __restrict__ partial, int nrows, int d) {
  reduce_rows_16col(dst, partial, nrows, d, blockIdx.x);
}

// the same for up to 32 T5LayerNorms in ONE launch: slot z = blockIdx.y reduces its partial rows into its weight's gradient (the whole
// backward of T5-small: 32 norms -> one launch at its end; a staged, data-parallel backward flushes at the end of every stage)
#define P5_REDUCE_MULTI_MAX 32
struct P5ReduceMulti {
  int n, d;
  int nrows[P5_REDUCE_MULTI_MAX];
  long long dst_off[P5_REDUCE_MULTI_MAX], part_off[P5_REDUCE_MULTI_MAX];    // element offsets into the gradient arena / the partial-sum scratch
};
__global__ __launch_bounds__(256) void p5_reduce_rows_multi_kernel(P5ReduceMulti a, float* __restrict__ G, const float* __restrict__ scratch) {
  const int z = blockIdx.y;
  reduce_rows_16col(G + a.dst_off[z], scratch + a.part_off[z], a.nrows[z], a.d, blockIdx.x);
}

// dst[i] += sum_c partial[c][i]   (partial copies of the relative-bias gradient)
__global__ __launch_bounds__(256) void p5_reduce_copies_kernel(float* __restrict__ dst, const float* __restrict__ partial, int n, int copies) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < copies; ++c) s += partial[(size_t)c * n + i];
  dst[i] += s;
}

// fp32 master -> compute-dtype shadow (bf16 fast mode)
template <class T>
__global__ __launch_bounds__(256) void p5_cast_kernel(T* __restrict__ out, const float* __restrict__ in, size_t n) {
  constexpr int EPF = TT<T>::EPF;
  const size_t nv = n / EPF * EPF;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPF; i < nv; i += (size_t)gridDim.x * 256 * EPF) {
    float v[8];
    ldf<EPF>(in + i, v);
    st16(out + i, pack16<T>(v));
  }
  for (size_t i = nv + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = from_f<T>(in[i]);
}

// gated-gelu epilogue (T5 v1.1 / Flan: HF modeling_t5.py:97-123): h = gelu_new(u0) * u1, u = [u0 | u1] per row
template <class T>
__global__ __launch_bounds__(256) void p5_gated_gelu_fwd_kernel(T* __restrict__ h, const T* __restrict__ u, int rows, int F, P5Drop drop) {
  const bool dd = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
  const size_t n = (size_t)rows * F;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / F, c = i % F;
    const float a = to_f<T>(u[r * 2 * F + c]), b = to_f<T>(u[r * 2 * F + F + c]);
    const float t = tanhf(0.7978845608028654f * (a + 0.044715f * a * a * a));
    float v = 0.5f * a * (1.f + t) * b;
    if (dd) v = p5_keep(seed, drop.site_key, (uint32_t)i, drop.thr) ? v * drop.scale : 0.f;
    h[i] = from_f<T>(v);
  }
}
// du = d(gelu_new(u0)*u1): dh arrives already multiplied by nothing; recomputes the dropout mask.
template <class T>
__global__ __launch_bounds__(256) void p5_gated_gelu_bwd_kernel(T* __restrict__ du, const T* __restrict__ dh, const T* __restrict__ u,
                                                               int rows, int F, P5Drop drop) {
  const bool dd = drop.state != nullptr && drop.thr != 0;
  const uint32_t seed = p5_seed(drop);
  const size_t n = (size_t)rows * F;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / F, c = i % F;
    const float a = to_f<T>(u[r * 2 * F + c]), b = to_f<T>(u[r * 2 * F + F + c]);
    float g = to_f<T>(dh[i]);
    if (dd) g = p5_keep(seed, drop.site_key, (uint32_t)i, drop.thr) ? g * drop.scale : 0.f;
    const float k = 0.7978845608028654f;
    const float inner = k * (a + 0.044715f * a * a * a);
    const float t = tanhf(inner);
    const float gel = 0.5f * a * (1.f + t);
    const float dgel = 0.5f * (1.f + t) + 0.5f * a * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * a * a);
    du[r * 2 * F + c] = from_f<T>(g * b * dgel);
    du[r * 2 * F + F + c] = from_f<T>(g * gel);
  }
}

// ---------------------------------------------------------------------------------------------------------
// K9/K10: token cross-entropy on fp32 logits [R, ldl] (P5_T5.py:368-369, reduction="none", ignore_index=-100)
// ---------------------------------------------------------------------------------------------------------
__device__ static __forceinline__ float block_reduce(float v, bool is_max, float* sred) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sred[wave] = v;
  __syncthreads();
  float r = sred[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sred[i]) : r + sred[i];
  return r;
}

// (T selects the exponential: exact expf in the fp32 parity mode, the 2-ulp exp2-based one in the bf16 mode -- 16 M calls per step)
template <class T>
__global__ __launch_bounds__(256) void p5_ce_fwd_kernel(float* __restrict__ nll, float* __restrict__ lse_out,
                                                       const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       int V, int ldl) {
  __shared__ float sm[4], ss[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* lr = logits + (size_t)row * ldl;
  // one pass: running (max, sum of exp) per thread with 16-byte loads, merged across the workgroup
  float m = P5_NEG_INF, s = 0.f;
  const int V4 = V >> 2;
  for (int j = tid; j < V4; j += 256) {
    const f32x4 v = *(const f32x4*)(lr + 4 * j);
    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (mx > m) { s *= p5_exp<T>(m - mx); m = mx; }
    s += p5_exp<T>(v[0] - m) + p5_exp<T>(v[1] - m) + p5_exp<T>(v[2] - m) + p5_exp<T>(v[3] - m);
  }
  for (int j = (V4 << 2) + tid; j < V; j += 256) {
    const float v = lr[j];
    if (v > m) { s *= p5_exp<T>(m - v); m = v; }
    s += p5_exp<T>(v - m);
  }
  {
    const float wm_ = wave_max(m);
    s = wave_sum(m == P5_NEG_INF ? 0.f : s * p5_exp<T>(m - wm_));
    if ((tid & 63) == 0) { sm[tid >> 6] = wm_; ss[tid >> 6] = s; }
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    s = 0.f;
    for (int w = 0; w < 4; ++w) s += ss[w] * p5_exp<T>(sm[w] - m);
  }
  if (threadIdx.x == 0) {
    const float lse = m + logf(s);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    nll[row] = (lab == -100) ? 0.f : lse - lr[lab];
  }
}

// Logit-free cross-entropy, second half of the forward: the head GEMM's epilogue (p5_gemm5.h, P5_EPI_CE_STATS) left, per row,
// `np` pairs (max, sum of exp) -- one per 64 vocabulary columns -- and the logit at the label; one wave per row merges them in index order:
// lse = M + log(sum_k s_k exp(m_k - M)), nll = lse - logit[label]  (P5_T5.py:368-369, reduction "none"; ignore_index -100 -> 0).
template <class T>
__global__ __launch_bounds__(256) void p5_ce_finish_kernel(float* __restrict__ nll, float* __restrict__ lse_out, const float* __restrict__ part,
                                                          const float* __restrict__ lab_logit, const int64_t* __restrict__ labels, int rows, int np) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = part + (size_t)row * np * 2;
  float m = P5_NEG_INF;
  for (int k = lane; k < np; k += 64) m = fmaxf(m, pr[2 * k]);
  m = wave_max(m);
  float s = 0.f;
  for (int k = lane; k < np; k += 64) s += pr[2 * k + 1] * p5_exp<T>(pr[2 * k] - m);
  s = wave_sum(s);
  if (lane == 0) {
    const float lse = m + logf(s);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    nll[row] = (lab == -100) ? 0.f : lse - lab_logit[row];
  }
}
// g[row] = gradient of the row's NLL: dnll[row] (autograd), or the runner's masked-mean loss (p5_ce_bwd_kernel's rule), 0 for ignored labels
__global__ __launch_bounds__(256) void p5_ce_gscale_kernel(float* __restrict__ g_out, const int64_t* __restrict__ labels, const float* __restrict__ dnll,
                                                          const int64_t* __restrict__ out_attn, int Tlen, float gscale, int rows) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float g;
  if (dnll) {
    g = dnll[row];
  } else {
    const int b = row / Tlen;
    float cnt = 0.f;
    for (int t = 0; t < Tlen; ++t) cnt += out_attn[(size_t)b * Tlen + t] != 0 ? 1.f : 0.f;
    g = out_attn[row] != 0 ? gscale / fmaxf(cnt, 1.f) : 0.f;
  }
  if (labels[row] == -100) g = 0.f;
  g_out[row] = g;
}

// Runner loss behind the CE kernel (DistributedRunner.py:72-77):
//   loss = mean_b( sum_t nll[b,t] * m[b,t] / max(sum_t m[b,t], 1) ),  m = (output_attention != 0)
// one workgroup, fixed summation order (deterministic): thread i owns batch rows i, i+256, ...
__global__ __launch_bounds__(256) void p5_masked_mean_kernel(float* __restrict__ loss, const float* __restrict__ nll,
                                                            const int64_t* __restrict__ out_attn, int B, int T) {
  __shared__ float sred[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float num = 0.f, cnt = 0.f;
    for (int t = 0; t < T; ++t) {
      const float m = out_attn[(size_t)b * T + t] != 0 ? 1.f : 0.f;
      num += nll[(size_t)b * T + t] * m;
      cnt += m;
    }
    s += num / fmaxf(cnt, 1.f);
  }
  s = block_reduce(s, false, sred);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// dlogits[row, j] = (softmax_j - [j == label]) * g[row]   -> T, padded columns (V..ldd) zeroed.
// g = dnll[row] (gradient of the per-token NLL handed in by autograd), or -- when dnll is null -- the gradient of the runner's
// masked-mean loss itself: g[b,t] = m[b,t] / (max(sum_t m[b,:], 1) * B) * gscale, recomputed from the label mask (a8).
template <class T>
__global__ __launch_bounds__(256) void p5_ce_bwd_kernel(T* __restrict__ dlogits, const float* __restrict__ logits,
                                                       const float* __restrict__ lse, const int64_t* __restrict__ labels,
                                                       const float* __restrict__ dnll, int V, int ldl, int ldd,
                                                       const int64_t* __restrict__ out_attn, int Tlen, float gscale) {
  // grid (rows, column slices); one 16-byte store per lane (ldl, ldd are multiples of 64: the lm_head GEMM pads V)
  constexpr int EPF = TT<T>::EPF;
  const int row = blockIdx.x;
  const float* lr = logits + (size_t)row * ldl;
  const int64_t lab = labels[row];
  float g;
  if (dnll) {
    g = dnll[row];
  } else {
    const int b = row / Tlen;
    float cnt = 0.f;
    for (int t = 0; t < Tlen; ++t) cnt += out_attn[(size_t)b * Tlen + t] != 0 ? 1.f : 0.f;
    g = out_attn[row] != 0 ? gscale / fmaxf(cnt, 1.f) : 0.f;
  }
  if (lab == -100) g = 0.f;
  const float l = lse[row];
  for (int j = (blockIdx.y * 256 + threadIdx.x) * EPF; j < ldd; j += gridDim.y * 256 * EPF) {
    float v[8], o[8];
    ldf<EPF>(lr + j, v);
#pragma unroll
    for (int e = 0; e < EPF; ++e) o[e] = (j + e < V) ? (p5_exp<T>(v[e] - l) - (j + e == lab ? 1.f : 0.f)) * g : 0.f;
    st16(dlogits + (size_t)row * ldd + j, pack16<T>(o));
  }
}

// ---------------------------------------------------------------------------------------------------------
// K11/K12: global-norm clip + HF AdamW (SingleRunner.py:191-217, DistributedRunner.py:81)
// over the flat arena: one pass for sum(g^2), one pass for the update (also refreshes the bf16 shadow).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void p5_sumsq_kernel(float* __restrict__ out, const float* __restrict__ g, size_t n) {
  __shared__ float sred[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
  s = block_reduce(s, false, sred);
  if (threadIdx.x == 0) out[blockIdx.x] = s;     // one partial per workgroup: summed in a fixed order by the consumer, so
}                                                // every data-parallel rank derives bit-identical clip factors
#define P5_SUMSQ_PARTS 1024

struct P5AdamArgs {
  float* p; const float* g; float* m; float* v;
  void* shadow;            // bf16 compute copy or nullptr
  const float* sumsq;      // device float[P5_SUMSQ_PARTS] partial sums of squared grads; nullptr => no clipping
  size_t n;
  float beta1, beta2, omb1, omb2, eps, max_norm, grad_scale;      // omb = 1 - beta, formed in double by the host
  float step_size, decay;  // lr * sqrt(1 - beta2^t) / (1 - beta1^t);  lr * weight_decay  (doubles rounded once)
};

__global__ __launch_bounds__(256) void p5_adamw_kernel(P5AdamArgs a) {
  float coef = a.grad_scale;
  if (a.sumsq) {
    __shared__ float sp[256];
    float t = 0.f;
    for (int i = threadIdx.x; i < P5_SUMSQ_PARTS; i += 256) t += a.sumsq[i];      // fixed order -> deterministic
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
  const float step_size = a.step_size;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
    const float g = a.g[i] * coef;
    const float m = a.beta1 * a.m[i] + a.omb1 * g;
    const float v = a.beta2 * a.v[i] + a.omb2 * g * g;
    a.m[i] = m;
    a.v[i] = v;
    float p = a.p[i];
    p = p - step_size * (m / (sqrtf(v) + a.eps));
    p = p - a.decay * p;
    a.p[i] = p;
    if (a.shadow) ((bf16*)a.shadow)[i] = from_f<bf16>(p);
  }
}
