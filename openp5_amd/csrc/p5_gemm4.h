// p5_gemm4.h -- persistent ring GEMM for the T5-small-sized Linear layers (K = 512 ... 8192, M = B*L = 8192 rows).
//
// Same contract as p5_gemm.h (C[M,N] (+)= sum_k A(m,k) B(n,k), bias-free nn.Linear forward / dgrad / wgrad, HF
// modeling_t5.py:205-208,304,325-326,367 and :83-94 with their autograd transposes), different execution shape.  What
// bounds the 128x128 / 64x64 kernels of p5_gemm.h on these problems is not the matrix pipe and not bandwidth but the
// number of operand bytes a CU has IN FLIGHT: a K-step of a 128x128 tile needs 32 KiB, the L2 round trip under load
// is >= 1 us, a two-stage pipeline therefore moves <= 64 KiB/us/CU (measured 14 TB/s chip-wide = 55 GB/s/CU) where the
// MFMAs of that tile could consume 150.  And every tile pays a cold prologue (first operands arrive after a full
// round trip) plus an epilogue nothing overlaps.  Hence:
//   * the whole LDS of a CU is one ring of NST K-steps (5 x 32 KiB for a 128x128 tile: 128 KiB in flight);
//   * the workgroup is PERSISTENT: it walks a list of work units (problem, tile, K-split) and the ring never drains --
//     the copies of the next unit's first K-steps are issued while the current unit's last K-steps are multiplied;
//   * the epilogue goes from the accumulators straight to global memory, so it needs no LDS (the ring stays live) and
//     no barrier.  The MFMA operands are swapped (D = B_frag x A_frag^T), which makes a lane own 4 consecutive output
//     COLUMNS of one row instead of 4 rows of one column; for K-contiguous B tiles the 16 rows an MFMA reads are also
//     permuted (rows q*8 + (j&1)*4 + k of a 32-row block) so that the lane's values of two neighbouring MFMA tiles
//     are 8 consecutive columns = one 16-byte store, 64 contiguous bytes per row and wave instruction;
//   * a launch may carry several problems (P5GemmGroup): the weight gradients of one layer run as ONE launch of 192
//     128x128 tiles, each over the full 8192-token reduction -- no split-K, no atomics, plain "C += acc";
//   * units are dealt to workgroups so that the workgroups of one XCD (blockIdx % 8) work on a contiguous run of tiles
//     of one problem at any moment: their operand panels are fetched into that XCD's L2 once.
// Operand staging (direct-to-LDS copies through inline asm, counted vmcnt, one LDS-only barrier in the middle of a step,
// fragments double-buffered in registers) follows p5_gemm2_kernel, whose schedule was tuned on the hardware in round 1/2.
#pragma once
#include "p5_gemm.h"

#define P5_MAX_GROUP 8
struct P5GemmGroup {
  int nprob, total_units;
  int unit_begin[P5_MAX_GROUP + 1];   // first unit of problem p; units of a problem: tile-major (n fastest), K-split fastest
  P5GemmArgs p[P5_MAX_GROUP];         // .g4_tiles_n / .g4_nk / .splitk filled by the launcher
};

// slot swizzles of the K-contiguous LDS images ([rows][128 B], 16-byte slot index XOR sigma(row)): A rows are read 16 consecutive
// rows at a time, B rows in the permuted order described above -- each makes the 16 lanes of a ds_read_b128 service group hit 16
// different bank groups
__device__ static __forceinline__ int g4_sigma_a(int row) { return row & 7; }
__device__ static __forceinline__ int g4_sigma_b(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// Wave tiles: 64x64 (TM = TN = 4; 128x128 tiles with 4 waves, 256x128 / 128x256 with 8), or 128x128 (TM = TN = 8: a 256x256 tile
// with FOUR waves, 256 accumulator registers each -- the unified 512-entry register file of a SIMD holds them at one wave per
// SIMD).  The LDS pipe is what bounds the 64x64 wave tile (tools/lab, ablated builds: MFMA alone 11.9 us, MFMA + fragment reads
// 22.3 us, the direct-to-LDS copies alone 19 us on 8192x2048x512): per MFMA it reads 0.5 fragments and receives 1/64 of a
// copied byte per MAC; the 128x128 wave tile halves both.
// ablation switches (tools/lab): 1 = no MFMA, 2 = no copies after the prologue, 4 = no fragment reads, 8 = no epilogue
// OCC = workgroups per CU the register allocation must leave room for; FLAGS bit 0: every unit starts its K loop at a different
// K-step (workgroups that share an operand panel then pull different lines of it at any moment)
#ifdef P5_EMU
static inline uint64_t sgpr64(uint64_t v) { return v; }
#else
__device__ static __forceinline__ uint64_t sgpr64(uint64_t v) {     // a wave-uniform 64-bit value the compiler must keep in SGPRs
  return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
}
#endif
template <int BM, int BN, int WMW, int WNW, int NST, bool KS, int ABL = 0, int OCC = 1, int FLAGS = 0>
__global__ __launch_bounds__(WMW* WNW * 64) P5_WAVES_PER_SIMD(WMW* WNW / 4 * OCC, WMW* WNW / 4 * OCC) void p5_gemm4_kernel(P5GemmGroup grp) {
  using T = bf16;
  constexpr int NW = WMW * WNW;
  constexpr int TM = BM / (16 * WMW), TN = BN / (16 * WNW);
  constexpr int WTM = TM * 16, WTN = TN * 16;
  static_assert((TM == 4 && TN == 4) || (TM == 8 && TN == 8), "64x64 or 128x128 wave tiles");
  constexpr bool KROT = (FLAGS & 1) != 0;
  // 128x128 wave tiles: 256 accumulator registers leave no room for sixteen 64-bit copy pointers -- scalar tile base + one per-lane
  // offset register per operand instead (needs whole tiles: M % BM == 0, N % BN == 0; the launcher checks)
  constexpr bool SADDR = TM == 8;
  static_assert(!(SADDR && KS), "128x128 wave tiles: K-contiguous operands only");
  static_assert(!SADDR || NST == 2, "128x128 wave tiles: two-slot ring (128 KiB)");
  constexpr int ASZ = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int LDS_BYTES = NST * STAGE;
  constexpr int NDA = BM / (8 * NW), NDB = BN / (8 * NW), NDMA = NDA + NDB;   // copy instructions per wave per K-step
  constexpr int PFD = NST - 1;                                                // K-steps in flight ahead of the one being multiplied
  constexpr int NFR = TM + TN, NMM = TM * TN;
  // copies of K-step s+PFD are issued in the first half of step s (into the slot read out before the previous step's barrier);
  // a two-slot ring issues them in the SECOND half of step s-1 instead, right behind the barrier that frees the slot
  constexpr bool COPY2 = NST == 2;
  constexpr int RSP = NMM / NFR, CSP = NMM / NDMA;      // one fragment read every RSP MFMAs, one copy every CSP
  static_assert(NST >= 2 && PFD * NDMA <= 56 && NDMA <= NMM && LDS_BYTES <= 160 * 1024, "ring geometry");
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave / WNW, wn = wave % WNW;

  // ---- this workgroup's units: XCD x = blockIdx % 8 owns units [x*UPX, (x+1)*UPX), its workgroups take them round-robin ----
  const int nwg = (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3, gx = nwg >> 3;
  const int upx = (grp.total_units + 7) >> 3;
  int ulast = upx < grp.total_units - xcd * upx ? upx : grp.total_units - xcd * upx;   // units of this XCD
  const int nmy = ulast > jx ? (ulast - jx + gx - 1) / gx : 0;
  if (nmy <= 0) return;

  struct Unit { int pi, m0, n0, kb, nk, rot; };
  auto decode = [&](int it) {
    Unit u;
    const int id = xcd * upx + it * gx + jx;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < P5_MAX_GROUP; ++q)
      if (q < grp.nprob && id >= grp.unit_begin[q]) pi = q;
    const P5GemmArgs& g = grp.p[pi];
    const int local = id - grp.unit_begin[pi];
    const int sp = local % g.splitk, tile = local / g.splitk;
    u.pi = pi;
    u.m0 = (tile / g.g4_tiles_n) * BM;
    u.n0 = (tile % g.g4_tiles_n) * BN;
    u.nk = g.g4_nk;
    u.kb = sp * g.g4_nk * 64;
    u.rot = KROT ? (tile * 5 + sp) % g.g4_nk : 0;
    return u;
  };

  // ---- copy cursor: the K-step the next copy fetches.  Past the last unit it keeps re-fetching the last K-step into free ring
  // slots (inc = 0): every step issues the same number of copies, so the vmcnt bookkeeping is a constant ----
  const T* srcA[SADDR ? 1 : NDA];
  const T* srcB[SADDR ? 1 : NDB];
  size_t incA = 0, incB = 0;
  int c_it = 0, c_left = 0, c_wrap = 0, c_nk = 0;
  uint64_t tA = 0, tB = 0;                      // SADDR: (tile base + K offset) of the K-step the next copy fetches -- kept in SGPRs
  unsigned rsA = 0, rsB = 0, voffA = 0, voffB0 = 0, voffB1 = 0;   // bytes per 8-row group; per-lane byte offsets (B: even / odd group)
  auto copy_setup = [&](int it) {
    Unit u = decode(it);
    const P5GemmArgs& g = grp.p[u.pi];
    c_left = u.nk;
    c_nk = u.nk;
    c_wrap = u.nk - u.rot;         // K-steps until the rotated loop wraps around to the unit's first K-step
    u.kb += u.rot * 64;
    if constexpr (SADDR) {
      tA = sgpr64((uint64_t)(uintptr_t)g.A + ((size_t)u.m0 * g.lda + u.kb) * 2);
      tB = sgpr64((uint64_t)(uintptr_t)g.B + ((size_t)u.n0 * g.ldb + u.kb) * 2);
      rsA = 8u * g.lda * 2u; rsB = 8u * g.ldb * 2u;
      const int r8 = lane >> 3, sl = lane & 7;    // row inside the 8-row group of a copy instruction, 16-byte slot
      voffA = ((unsigned)r8 * g.lda + ((sl ^ r8) * 8)) * 2u;                       // sigma_a(row) = row & 7 = r8
      voffB0 = ((unsigned)r8 * g.ldb + ((sl ^ (r8 & 3)) * 8)) * 2u;                // sigma_b(row) = (row & 3) | (group parity << 2)
      voffB1 = ((unsigned)r8 * g.ldb + ((sl ^ ((r8 & 3) | 4)) * 8)) * 2u;
    } else if constexpr (KS) {
      incA = (size_t)64 * g.lda; incB = (size_t)64 * g.ldb;
#pragma unroll
      for (int i = 0; i < NDA; ++i) {
        constexpr int CPR = BM / 8, RPI = 512 / BM;
        const int krow = (wave * NDA + i) * RPI + lane / CPR;
        int cg = (lane % CPR) ^ ksd_swz<BM>(krow);
        const int cmax = (g.lda - u.m0) / 8 - 1;        // (row capacity of the k-row in memory, see stage_dma_ks)
        cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
        srcA[i] = (const T*)g.A + ((size_t)u.kb + krow) * g.lda + u.m0 + cg * 8;
      }
#pragma unroll
      for (int i = 0; i < NDB; ++i) {
        constexpr int CPR = BN / 8, RPI = 512 / BN;
        const int krow = (wave * NDB + i) * RPI + lane / CPR;
        int cg = (lane % CPR) ^ ksd_swz<BN>(krow);
        const int cmax = (g.ldb - u.n0) / 8 - 1;
        cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
        srcB[i] = (const T*)g.B + ((size_t)u.kb + krow) * g.ldb + u.n0 + cg * 8;
      }
    } else {
      incA = 64; incB = 64;
#pragma unroll
      for (int i = 0; i < NDA; ++i) {
        const int row = (wave * NDA + i) * 8 + (lane >> 3);
        int gr = u.m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        srcA[i] = (const T*)g.A + (size_t)gr * g.lda + u.kb + (((lane & 7) ^ g4_sigma_a(row)) * 8);
      }
#pragma unroll
      for (int i = 0; i < NDB; ++i) {
        const int row = (wave * NDB + i) * 8 + (lane >> 3);
        int gr = u.n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        srcB[i] = (const T*)g.B + (size_t)gr * g.ldb + u.kb + (((lane & 7) ^ g4_sigma_b(row)) * 8);
      }
    }
  };
  auto copy_one = [&](int buf, int idx, bool prologue = false) {
    if ((ABL & 2) != 0 && !prologue) return;      // (ablation: the prologue still fills the ring once)
    char* b = lds + buf * STAGE;
    if constexpr (SADDR) {
      static_assert(!SADDR || (NDB % 2) == 0, "group parity of a B copy must be a compile-time constant");
      if (idx < NDA) glds16_raw_s(tA + (uint64_t)(wave * NDA + idx) * rsA, voffA, b + (wave * NDA + idx) * 1024);
      else glds16_raw_s(tB + (uint64_t)(wave * NDB + (idx - NDA)) * rsB, ((idx - NDA) & 1) ? voffB1 : voffB0, b + ASZ + (wave * NDB + (idx - NDA)) * 1024);
    } else {
      if (idx < NDA) glds16_raw(srcA[idx], b + (wave * NDA + idx) * 1024);
      else glds16_raw(srcB[idx - NDA], b + ASZ + (wave * NDB + (idx - NDA)) * 1024);
    }
  };
  auto copy_advance = [&]() {     // after the NDMA copies of a K-step
    if (--c_left > 0) {
      if constexpr (SADDR) {
        if (KROT && --c_wrap == 0) { tA -= (uint64_t)128 * (c_nk - 1); tB -= (uint64_t)128 * (c_nk - 1); }
        else { tA += 128; tB += 128; }
      } else if (KROT && --c_wrap == 0) {
#pragma unroll
        for (int i = 0; i < NDA; ++i) srcA[i] -= incA * (size_t)(c_nk - 1);
#pragma unroll
        for (int i = 0; i < NDB; ++i) srcB[i] -= incB * (size_t)(c_nk - 1);
      } else {
#pragma unroll
        for (int i = 0; i < NDA; ++i) srcA[i] += incA;
#pragma unroll
        for (int i = 0; i < NDB; ++i) srcB[i] += incB;
      }
    } else if (c_it + 1 < nmy) {
      copy_setup(++c_it);
    } else {
      c_left = 1;                 // stay on the last K-step
    }
  };

  // ---- fragment addressing (loop-invariant lane offsets inside a ring slot) ----
  // K-contiguous images: ONE lane offset per operand; row block i of A sits i * 16 rows further (sigma_a unchanged), MFMA tile j of
  // B (j >> 1) * 32 + (j & 1) * 4 rows further (sigma_b unchanged) -- compile-time deltas that fold into the ds_read offset field
  int offA[KS ? TM : 1], offB[KS ? TN : 1];
  if constexpr (KS) {
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = ksd_lane_off<BM>(wm * WTM + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = ksd_lane_off<BN>(wn * WTN + j * 16, lane) + ASZ;
  } else {
    const int rowa = wm * WTM + (lane & 15);
    offA[0] = rowa * 128 + (((lane >> 4) ^ g4_sigma_a(rowa)) << 4);
    const int rowb = wn * WTN + ((lane & 15) >> 2) * 8 + (lane & 3);
    offB[0] = ASZ + rowb * 128 + (((lane >> 4) ^ g4_sigma_b(rowb)) << 4);
  }
  auto frag = [&](int buf, bool is_b, int t, int c) -> u32x4 {
    const char* p = lds + buf * STAGE;
    if constexpr ((ABL & 4) != 0) { u32x4 z = {(unsigned)(buf + t), 1u, 2u, 3u}; return z; }
    if constexpr (KS) {
      const int off = is_b ? offB[is_b ? t : 0] : offA[is_b ? 0 : t], rbytes = is_b ? BN * 2 : BM * 2;
      const u32x2 lo = lds_tr16_b64(p + off + c * 32 * rbytes), hi = lds_tr16_b64(p + off + c * 32 * rbytes + 4 * rbytes);
      u32x4 r;
      r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
      return r;
    } else {
      const int delta = is_b ? ((t >> 1) * 32 + (t & 1) * 4) * 128 : t * 16 * 128;
      return ld16(p + ((is_b ? offB[0] : offA[0]) ^ (c << 6)) + delta);     // K-chunk 1 = slot index ^ 4
    }
  };
  // fragment idx of a K-chunk in the order the MFMAs need them: A row-block 0, all B column-blocks, the other A row-blocks
  auto load_one = [&](auto& fa, u32x4(&fb)[TN], int buf, int c, int idx) {
    if (idx == 0) fa[0] = frag(buf, false, 0, c);
    else if (idx <= TN) fb[idx - 1] = frag(buf, true, idx - 1, c);
    else fa[idx - TN] = frag(buf, false, idx - TN, c);
  };

  f32x4 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mm = [&](f32x4& a, const u32x4& fa, const u32x4& fb) {
    if constexpr ((ABL & 1) != 0) { a[0] += __builtin_bit_cast(float, fa[0] ^ fb[0]); }
#ifndef P5_EMU
    // 128x128 wave tiles: the 256 accumulators must sit in the AGPR half of the register file and be updated in place.  Left to
    // itself hipcc keeps half of them in VGPRs and moves every tile through a scratch AGPR quad around its MFMA (8 moves per MFMA);
    // the "a" constraint pins them.  (No MFMA of a K-step touches a tile the previous 63 wrote, and the epilogue starts behind a
    // barrier-free run of scalar work plus an explicit s_nop: no software-visible MFMA hazard is left to the assembler.)
    else if constexpr (SADDR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(a) : "v"(fb), "v"(fa));
#endif
    else mma16<T>(a, fb, fa);     // operands swapped: lane <- C[m = 16 i + (lane & 15)][4 columns]
  };

  // One 16-row block of the wave tile at a time, called with a CONSTANT row-block index (hipcc will not fully unroll a loop over
  // all 8 row blocks of the 128x128 wave tile around this body, and a dynamically indexed accumulator array lives in scratch).
  auto epi_rows = [&](const f32x4(&a)[TN], int i, const Unit& u, const P5GemmArgs& g, uint32_t seed, bool do_drop, bool vec_ok, int le) {
    const int gl = le >> 4;
    const int row = u.m0 + wm * WTM + i * 16 + (le & 15);
    if constexpr (KS) {
      // lane owns C[row][col .. col+3] of every 16x16 tile: fp32 outputs (weight gradients)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = u.n0 + wn * WTN + j * 16 + gl * 4;
        if (row >= g.M || col >= g.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[j][r] * g.alpha;
        const size_t ci = (size_t)row * g.ldc + col;
        if (g.c_f32 && col + 4 <= g.N && (g.ldc & 3) == 0) {
          float* cp = (float*)g.C + ci;
          if (g.epi == P5_EPI_ATOMIC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
          } else if (g.epi == P5_EPI_ACCUM) {
            f32x4 c = *(const f32x4*)cp;
            *(f32x4*)cp = (f32x4){c[0] + v[0], c[1] + v[1], c[2] + v[2], c[3] + v[3]};
          } else {
            *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (col + r >= g.N) continue;
            if (g.epi == P5_EPI_ATOMIC) atomicAdd((float*)g.C + ci + r, v[r]);
            else if (g.epi == P5_EPI_ACCUM) ((float*)g.C)[ci + r] += v[r];
            else if (g.c_f32) ((float*)g.C)[ci + r] = v[r];
            else ((T*)g.C)[ci + r] = from_f<T>(v[r]);
          }
        }
      }
    } else {
      // lane owns 8 consecutive columns of a row per pair of MFMA tiles (2h, 2h+1): one 16-byte bf16 store (two for fp32)
      const bool row_ok = row < g.M;
      float sc = g.alpha;
      if (g.rowss) sc *= gemm_row_rstd(g, row_ok ? row : g.M - 1);
      float ss = 0.f;
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) {
        const int col = u.n0 + wn * WTN + h * 32 + gl * 8;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = a[2 * h][r] * sc; v[4 + r] = a[2 * h + 1][r] * sc; }
        if (row_ok && col < g.N) {
          const size_t ci = (size_t)row * g.ldc + col;
          const bool full = vec_ok && col + 8 <= g.N;
          if (g.epi != P5_EPI_STORE && g.epi != P5_EPI_ATOMIC && g.epi != P5_EPI_ACCUM) {
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = 0.f;
            if (g.aux) {
              const T* ap = (const T*)g.aux + (size_t)row * g.ldaux + col;
              if (full) unpack16<T>(ld16(ap), av);
              else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (col + e < g.N) av[e] = to_f<T>(ap[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gemm_epi_apply(g, v[e], av[e], seed, do_drop, row, col + e);
          }
          if (g.c_f32) {
            float* cp = (float*)g.C + ci;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (col + e >= g.N) continue;
              if (g.epi == P5_EPI_ATOMIC) atomicAdd(cp + e, v[e]);
              else if (g.epi == P5_EPI_ACCUM) cp[e] += v[e];
              else if (!full) cp[e] = v[e];
            }
            if (full && g.epi != P5_EPI_ATOMIC && g.epi != P5_EPI_ACCUM) {
              *(f32x4*)cp = (f32x4){v[0], v[1], v[2], v[3]};
              *(f32x4*)(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            }
            if (g.ssq_out) {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (col + e < g.N) ss += v[e] * v[e];
            }
          } else {
            const u32x4 packed = pack16<T>(v);
            if (full) st16((T*)g.C + ci, packed);
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (col + e < g.N) ((T*)g.C)[ci + e] = from_f<T>(v[e]);
            }
            if (g.ssq_out) {     // sum of squares of the row as stored
              float w[8];
              unpack16<T>(packed, w);
#pragma unroll
              for (int e = 0; e < 8; ++e) if (col + e < g.N) ss += w[e] * w[e];
            }
          }
        }
      }
      if (g.ssq_out) {          // (uniform branch) the four lane groups hold the row's columns of this wave
        if (g.ssq_nt > 0 && WTN == 64) {     // this wave IS the only writer of the row's partial for its 64-column group
          ss += __shfl_xor(ss, 16);
          ss += __shfl_xor(ss, 32);
          const int cg = (u.n0 + wn * WTN) >> 6;
          if (gl == 0 && row_ok && cg < g.ssq_nt) g.ssq_out[(size_t)row * g.ssq_nt + cg] = ss;
        } else {
          ss += __shfl_xor(ss, 16);
          ss += __shfl_xor(ss, 32);
          if (gl == 0 && row_ok) atomicAdd(g.ssq_out + row, ss);
        }
      }
    }
  };
  // ---- epilogue: accumulators -> global, no LDS ----
  auto epilogue = [&](const Unit& u) {
    if constexpr ((ABL & 8) != 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) ((float*)grp.p[0].C)[0] = s;
      return;
    }
    const P5GemmArgs& g = grp.p[u.pi];
    const uint32_t seed = p5_seed(g.drop);
    const bool do_drop = g.drop.state != nullptr && g.drop.thr != 0;
    const bool vec_ok = (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.aux == nullptr || ((g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0));
    // the lane id goes through an opaque asm: hipcc otherwise hoists every lane-dependent output offset of the epilogue out of the
    // unit loop and keeps dozens of registers alive across the K loop, where the 128x128 wave tile has none to spare
    int le = lane;
#ifndef P5_EMU
    asm volatile("" : "+v"(le));
    if constexpr (SADDR) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs (inline asm) retire before the accumulators are read
#endif
    epi_rows(acc[0], 0, u, g, seed, do_drop, vec_ok, le);
    epi_rows(acc[1], 1, u, g, seed, do_drop, vec_ok, le);
    epi_rows(acc[2], 2, u, g, seed, do_drop, vec_ok, le);
    epi_rows(acc[3], 3, u, g, seed, do_drop, vec_ok, le);
    if constexpr (TM > 4) {
      epi_rows(acc[4], 4, u, g, seed, do_drop, vec_ok, le);
      epi_rows(acc[5], 5, u, g, seed, do_drop, vec_ok, le);
      epi_rows(acc[6], 6, u, g, seed, do_drop, vec_ok, le);
      epi_rows(acc[7], 7, u, g, seed, do_drop, vec_ok, le);
    }
  };

  // ---- the ring ----
  // 64x64 wave tiles: A and B fragments of both K-chunks double-buffered (fa0/fb0, fa1/fb1: 128 registers).
  // 128x128 wave tiles (256 accumulators): ONE set of A fragments, refilled in place -- the MFMAs run row block by row block, and
  // when row block i of a K-chunk is done its A fragment is dead, so the next chunk's fragment of that row block is read into the
  // same registers (needed 56 MFMAs later); only the B fragments are double-buffered (fa0/fb0/fb1: 96 registers).
  u32x4 fa0[TM], fb0[TN], fa1[SADDR ? 1 : TM], fb1[TN];
  auto step = [&](int buf, int nb1, int nb2) {
    P5_SCHED_FENCE();
    if constexpr (SADDR) {
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        const int i = t / TN, j = t % TN;
        mm(acc[i][j], fa0[i], fb0[j]);
        P5_SCHED_FENCE();
        if (j == TN / 2 - 1) fb1[i] = frag(buf, true, i, 1);
        if (j == TN - 1) fa0[i] = frag(buf, false, i, 1);
        P5_SCHED_FENCE();
      }
      P5_WAIT_VM(0);                        // (two-slot ring) K-step s+1 has landed
      P5_BARRIER_LDS();
      P5_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NMM; ++t) {
        const int i = t / TN, j = t % TN;
        mm(acc[i][j], fa0[i], fb1[j]);
        P5_SCHED_FENCE();
        if (j == TN / 2 - 1) fb0[i] = frag(nb1, true, i, 0);
        if (j == TN - 1) fa0[i] = frag(nb1, false, i, 0);
        if (t % CSP == 1 && t / CSP < NDMA) copy_one(buf, t / CSP);     // K-step s+2 -> the slot just read out
        P5_SCHED_FENCE();
      }
      copy_advance();
      return;
    }
#pragma unroll
    for (int t = 0; t < NMM; ++t) {
      mm(acc[t / TN][t % TN], fa0[t / TN], fb0[t % TN]);
      P5_SCHED_FENCE();
      if (t % RSP == 0 && t / RSP < NFR) load_one(fa1, fb1, buf, 1, t / RSP);
      if constexpr (!COPY2)
        if (t % CSP == 0 && t / CSP < NDMA) copy_one(nb2, t / CSP);     // K-step s+PFD -> ring slot of K-step s-1 (read out before that step's barrier)
      P5_SCHED_FENCE();
    }
    if constexpr (!COPY2) copy_advance();
    P5_WAIT_VM(COPY2 ? 0 : (PFD - 1) * NDMA);   // this wave's share of K-step s+1 has landed
    P5_BARRIER_LDS();                     // ... everyone's; and slot `buf` is fully read
    P5_SCHED_FENCE();
#pragma unroll
    for (int t = 0; t < NMM; ++t) {
      mm(acc[t / TN][t % TN], fa1[t / TN], fb1[t % TN]);
      P5_SCHED_FENCE();
      if (t % RSP == 0 && t / RSP < NFR) load_one(fa0, fb0, nb1, 0, t / RSP);
      if constexpr (COPY2)
        if (t % CSP == 0 && t / CSP < NDMA) copy_one(buf, t / CSP);     // two-slot ring: K-step s+2 -> the slot just read out
      P5_SCHED_FENCE();
    }
    if constexpr (COPY2) copy_advance();
  };

  copy_setup(0);
#pragma unroll
  for (int q = 0; q < PFD; ++q) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      copy_one(q, i, true);
    }
    copy_advance();
  }
  if constexpr (COPY2) {               // the two-slot ring runs one K-step further ahead: K-step 1 goes out before the loop
#pragma unroll
    for (int i = 0; i < NDMA; ++i) copy_one(1, i);
    copy_advance();
    P5_WAIT_VM(NDMA);
  } else {
    P5_WAIT_VM((PFD - 1) * NDMA);
  }
  P5_BARRIER_LDS();
#pragma unroll
  for (int i = 0; i < NFR; ++i) load_one(fa0, fb0, 0, 0, i);
  zero_acc();
  int buf = 0;
  for (int it = 0; it < nmy; ++it) {
    const Unit u = decode(it);
    int k = 0;
    do {    // nk >= 1 (a for loop's zero-trip guard makes hipcc read the accumulators back every iteration, p5_gemm.h)
      const int nb1 = buf == NST - 1 ? 0 : buf + 1, nb2 = buf == 0 ? NST - 1 : buf - 1;
      step(buf, nb1, nb2);
      buf = nb1;
    } while (++k < u.nk);
    epilogue(u);
    zero_acc();
  }
  P5_WAIT_VM(0);      // the trailing re-fetches must land before the workgroup's LDS is handed to the next one
}
