// p5_embed.h -- gradient of the embedding lookups (SURVEY.md App. C "Embedding lookups"; P5_T5.py:94-100 encoder: E[ids] + WW[ww],
// decoder: E[dec_ids]; the three contributions to the tied shared.weight are the head GEMM and these two lookups) WITHOUT fp32 atomics.
//
// Rounds 1-3 scattered every row of the residual-stream gradient into its table row with one fp32 atomic per element.  The order the
// atomics of a repeated token land in changes from run to run, the last bits of shared.weight's gradient with it, and AdamW turns a
// last-bit difference of a near-zero gradient into an O(lr) difference of the parameter (tools/diag_repro.py,
// profiles/r04_repro_before_fix.txt): two runs of the same training differed by up to 2 x lr.  Here the rows are summed in a FIXED order:
//   1. p5_embed_sortchunk_kernel + p5_embed_rank_kernel
//                               stable rank of every lookup row by (key, row): sorted position, start and length of its key's segment
//                               (chunks of 256 rows sorted in LDS, then one binary search per row and chunk -- integer comparisons only);
//   2. p5_embed_seg_kernel      one workgroup per block of 32 sorted positions: rows of one key are added in row order; a segment that
//                               lies inside the block is added to the table row by its only owner (plain read-modify-write), pieces of
//                               segments that cross block boundaries go to a partial buffer;
//   3. p5_embed_fix_kernel      the block in which a crossing segment starts adds its pieces in block order and updates the table row.
// The same association on every run -> bit-identical gradients.  A key set may be the concatenation of two index arrays (encoder ids
// followed by the decoder's shifted labels, both looking up the tied table): the tied table then has ONE owner per row.
#pragma once
#include "p5_device.h"
#include "p5_rng.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define P5_EMB_SEG 32          // sorted positions per workgroup of the segmented sum
#define P5_EMB_MAXSETS 2

struct P5EmbSet {
  const int64_t* key0; const int64_t* key1;     // virtual concatenation: row r < n0 looks up key0[r], else key1[r - n0]
  const float* dres0; const float* dres1;       // [n0, d] / [n1, d] fp32 gradients of the looked-up rows
  P5Drop drop0, drop1;                          // dropout applied to the looked-up rows in the forward (element index = local row * d + c)
  float* table;                                 // gradient of the table, += (row-major [*, d])
  int n0, n1;
  int* perm; int* sstart; int* slen;            // [n] sorted position -> row, start / length of the position's key segment
  int* skey;                                    // [n] sorted position -> key
  unsigned long long* csort;                    // [ceil(n / 256) * 256] scratch: (key << 32 | row), sorted inside each chunk of 256 rows
  float* part;                                  // [blocks][2][d] pieces of segments crossing block boundaries (0: continues from before, 1: continues after)
};
struct P5EmbArgs {
  P5EmbSet s[P5_EMB_MAXSETS];
  int nsets, d;
};

__device__ static __forceinline__ int emb_key(const P5EmbSet& s, int r) { return (int)(r < s.n0 ? s.key0[r] : s.key1[r - s.n0]); }

// Stable rank by (key, row) in two levels -- O(n (log c + (n / c) log c)) instead of the n^2 / 2 comparisons of a brute-force count
// (which took 98 us at n = 8704 and would take milliseconds at the 33k rows of a T5-large, L = 512 batch):
//   p5_embed_sortchunk_kernel   one workgroup per chunk of 256 rows: bitonic sort of the composite 64-bit values (key << 32 | row) in LDS
//                               -> `csort` (sorted chunks, back to back);
//   p5_embed_rank_kernel        one thread per row: its rank = sum over ALL chunks of the number of values below its own (a binary
//                               search per chunk, chunks staged through LDS), likewise the start and the end of its key's segment
//                               (values below key << 32 / below (key + 1) << 32).
// Integer comparisons only: the same permutation on every run.
#define P5_EMB_CHUNK 256
__global__ __launch_bounds__(256) void p5_embed_sortchunk_kernel(P5EmbArgs a) {
  __shared__ unsigned long long sv[P5_EMB_CHUNK];
  const P5EmbSet& s = a.s[blockIdx.y];
  const int n = s.n0 + s.n1;
  const int r = blockIdx.x * P5_EMB_CHUNK + threadIdx.x;
  if (blockIdx.x * P5_EMB_CHUNK >= n) return;
  sv[threadIdx.x] = r < n ? (((unsigned long long)(unsigned)emb_key(s, r)) << 32) | (unsigned)r : ~0ull;      // (padding sorts last)
  __syncthreads();
  for (int k = 2; k <= P5_EMB_CHUNK; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int i = threadIdx.x, l = i ^ j;
      if (l > i) {
        const unsigned long long x = sv[i], y = sv[l];
        const bool up = (i & k) == 0;
        if ((x > y) == up) { sv[i] = y; sv[l] = x; }
      }
      __syncthreads();
    }
  }
  s.csort[(size_t)blockIdx.x * P5_EMB_CHUNK + threadIdx.x] = sv[threadIdx.x];
}

// number of values of the sorted array v[0..m) that are < x
__device__ static __forceinline__ int emb_lower_bound(const unsigned long long* v, int m, unsigned long long x) {
  int lo = 0, hi = m;
#pragma unroll 1
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// grid (ceil(max n / 64), nsets), 256 threads: lane = one of 64 rows, wave w searches the chunks c = w (mod 4) of every LDS batch
// (16 sorted chunks = 32 KiB at a time); the four waves' counts are added through LDS
__global__ __launch_bounds__(256) void p5_embed_rank_kernel(P5EmbArgs a) {
  constexpr int CPB = 16;                 // chunks per LDS batch
  __shared__ unsigned long long sc[CPB * P5_EMB_CHUNK];
  __shared__ int scnt[3][4][64];
  const P5EmbSet& s = a.s[blockIdx.y];
  const int n = s.n0 + s.n1;
  if (blockIdx.x * 64 >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  const int k = r < n ? emb_key(s, r) : 0;
  const unsigned long long me = (((unsigned long long)(unsigned)k) << 32) | (unsigned)r;
  const unsigned long long klo = ((unsigned long long)(unsigned)k) << 32, khi = ((unsigned long long)(unsigned)k + 1ull) << 32;
  const int nchunks = (n + P5_EMB_CHUNK - 1) / P5_EMB_CHUNK;
  int pos = 0, start = 0, end = 0;
  for (int c0 = 0; c0 < nchunks; c0 += CPB) {
    const int nc = nchunks - c0 < CPB ? nchunks - c0 : CPB;
    __syncthreads();
    for (int i = threadIdx.x; i < nc * P5_EMB_CHUNK; i += 256) sc[i] = s.csort[(size_t)c0 * P5_EMB_CHUNK + i];
    __syncthreads();
    for (int c = wave; c < nc; c += 4) {
      const unsigned long long* v = sc + c * P5_EMB_CHUNK;
      const int lo = emb_lower_bound(v, P5_EMB_CHUNK, klo);      // (padding values ~0 are above every real value)
      start += lo;
      // the chunk's values of this key form the run [lo, hi): search the rest inside it
      const int hi = lo + emb_lower_bound(v + lo, P5_EMB_CHUNK - lo, khi);
      end += hi;
      pos += lo + emb_lower_bound(v + lo, hi - lo, me);
    }
  }
  scnt[0][wave][lane] = pos; scnt[1][wave][lane] = start; scnt[2][wave][lane] = end;
  __syncthreads();
  if (wave == 0 && r < n) {
    pos = scnt[0][0][lane] + scnt[0][1][lane] + scnt[0][2][lane] + scnt[0][3][lane];
    start = scnt[1][0][lane] + scnt[1][1][lane] + scnt[1][2][lane] + scnt[1][3][lane];
    end = scnt[2][0][lane] + scnt[2][1][lane] + scnt[2][2][lane] + scnt[2][3][lane];
    s.perm[pos] = r; s.skey[pos] = k; s.sstart[pos] = start; s.slen[pos] = end - start;
  }
}

// value of lookup row `r`, columns c..c+1 (dropout of the forward re-applied)
__device__ static __forceinline__ void emb_row2(const P5EmbSet& s, int r, int d, int c, uint32_t seed0, uint32_t seed1, float& x0, float& x1) {
  const bool first = r < s.n0;
  const int lr = first ? r : r - s.n0;
  const float* src = (first ? s.dres0 : s.dres1) + (size_t)lr * d + c;
  const P5Drop& dp = first ? s.drop0 : s.drop1;
  x0 = src[0]; x1 = src[1];
  if (dp.state != nullptr && dp.thr != 0) {
    const uint32_t seed = first ? seed0 : seed1;
    const uint32_t i0 = (uint32_t)(lr * d + c);
    x0 = p5_keep(seed, dp.site_key, i0, dp.thr) ? x0 * dp.scale : 0.f;
    x1 = p5_keep(seed, dp.site_key, i0 + 1, dp.thr) ? x1 * dp.scale : 0.f;
  }
}

// grid (ceil(max n / 32), nsets), 256 threads = 512 columns per pass
__global__ __launch_bounds__(256) void p5_embed_seg_kernel(P5EmbArgs a) {
  constexpr int S = P5_EMB_SEG;
  __shared__ int sp[S], sk[S], sst[S], sln[S];
  const P5EmbSet& s = a.s[blockIdx.y];
  const int n = s.n0 + s.n1, d = a.d;
  const int p0 = blockIdx.x * S;
  if (p0 >= n) return;
  const int cnt = n - p0 < S ? n - p0 : S;
  if (threadIdx.x < S) {
    const int p = p0 + threadIdx.x;
    const bool ok = threadIdx.x < cnt;
    sp[threadIdx.x] = ok ? s.perm[p] : 0; sk[threadIdx.x] = ok ? s.skey[p] : -1;
    sst[threadIdx.x] = ok ? s.sstart[p] : 0; sln[threadIdx.x] = ok ? s.slen[p] : 0;
  }
  __syncthreads();
  const uint32_t seed0 = p5_seed(s.drop0), seed1 = p5_seed(s.drop1);
  for (int c = threadIdx.x * 2; c < d; c += 512) {
    float v0[S], v1[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {                 // every row of the block is requested before the first add
      v0[j] = 0.f; v1[j] = 0.f;
      if (j < cnt) emb_row2(s, sp[j], d, c, seed0, seed1, v0[j], v1[j]);
    }
    // the table rows this block owns (segments that lie inside it: distinct keys, distinct rows, this workgroup their only writer) are
    // requested up front as well: `dst += a` row by row was one dependent HBM round trip per key (48 us per launch at the C2 shape for
    // 35 MB of reads; round 6)
    float o0[S], o1[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      o0[j] = 0.f; o1[j] = 0.f;
      if (j < cnt && (j == cnt - 1 || sk[j + 1] != sk[j])) {       // (uniform)
        const int ss = sst[j], se = ss + sln[j];
        if (ss >= p0 && se <= p0 + S) {
          const float* src = s.table + (size_t)sk[j] * d + c;
          o0[j] = src[0]; o1[j] = src[1];
        }
      }
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (j >= cnt) break;                        // (uniform)
      a0 += v0[j]; a1 += v1[j];
      const bool last = j == cnt - 1 || sk[j + 1] != sk[j];       // (uniform)
      if (last) {
        const int ss = sst[j], se = ss + sln[j];
        const bool head = ss < p0, tail = se > p0 + S;
        float* dst = (!head && !tail) ? s.table + (size_t)sk[j] * d + c : s.part + ((size_t)blockIdx.x * 2 + (head ? 0 : 1)) * d + c;
        if (!head && !tail) { dst[0] = o0[j] + a0; dst[1] = o1[j] + a1; } else { dst[0] = a0; dst[1] = a1; }
        a0 = 0.f; a1 = 0.f;
      }
    }
  }
}

// grid (ceil(max n / 32), nsets): the block in which a boundary-crossing segment STARTS owns it
__global__ __launch_bounds__(256) void p5_embed_fix_kernel(P5EmbArgs a) {
  constexpr int S = P5_EMB_SEG;
  const P5EmbSet& s = a.s[blockIdx.y];
  const int n = s.n0 + s.n1, d = a.d;
  const int p0 = blockIdx.x * S;
  if (p0 + S >= n) return;                        // (the last block has no successor to cross into)
  const int last = p0 + S - 1;
  const int ss = s.sstart[last], se = ss + s.slen[last];
  if (se <= p0 + S || ss < p0) return;            // does not cross, or started in an earlier block (uniform)
  const int key = s.skey[last];
  const int b1 = (se - 1) / S;                    // last block holding a piece (its piece is a "continues from before" piece: slot 0)
  for (int c = threadIdx.x * 2; c < d; c += 512) {
    const float* p = s.part + ((size_t)blockIdx.x * 2 + 1) * d + c;
    float a0 = p[0], a1 = p[1];
    int b = blockIdx.x + 1;
    for (; b + 3 <= b1; b += 4) {                 // four pieces in flight, added in block order
      const float* q0 = s.part + (size_t)b * 2 * d + c;
      const float x00 = q0[0], x01 = q0[1], x10 = q0[2 * d], x11 = q0[2 * d + 1], x20 = q0[4 * d], x21 = q0[4 * d + 1], x30 = q0[6 * d], x31 = q0[6 * d + 1];
      a0 = (((a0 + x00) + x10) + x20) + x30;
      a1 = (((a1 + x01) + x11) + x21) + x31;
    }
    for (; b <= b1; ++b) {
      const float* q = s.part + (size_t)b * 2 * d + c;
      a0 += q[0]; a1 += q[1];
    }
    float* dst = s.table + (size_t)key * d + c;
    dst[0] += a0; dst[1] += a1;
  }
}
