// p5_gemm6_experimental.h -- LAB ONLY (tools/lab/gemm_lab.hip, mode lab4); not part of libp5hip.so.
// A wave-specialised persistent GEMM whose LOADER waves also write the finished tiles out.  Correct (it passed the grouped-problem
// parity cases on the MI355X) and SLOWER than p5_gemm5_kernel everywhere: 39.5 vs 24.1 us on 8192x2048x512, 284 vs 230 us on
// 8192x8192x2048 (gpurun_out/lab4c.txt, round 3).  The 64 KiB staging image only fits beside a ring of 32-wide K-steps, and those
// cost more than the overlapped epilogue saves: a 1 KiB copy then covers 16 rows x 64 B -- half cache lines, each line of the
// operands is requested twice -- (copies ablated: 23.0 us), and the compute waves meet at a barrier every 32 MFMAs instead of every
// 64 (MFMA + fragment reads alone 23.6 us against 12.1 us for the 64-wide K-steps of p5_gemm5.h).  Kept for the record of what
// was tried.
//
// Same contract, work-unit scheduling and wave roles as p5_gemm5.h (bias-free nn.Linear forward / dgrad of the T5 layers, HF
// modeling_t5.py:83-94,205-208,304,325-326,367), K-contiguous operands and bf16 outputs only.  What it changes is where a
// finished tile goes.  On the wide forward GEMMs of the benchmark step (8192 x 2048 x 512: K is short, the output is 33.5 MB)
// p5_gemm5_kernel spends 14.8 us in its K loops and 7.6 us in the epilogue (tools/lab lab4, round 3): the four compute waves of
// a CU issue the tile's 64 KiB of stores themselves, the chip takes ~3 TB/s from 1024 storing waves (a dedicated write kernel
// with four waves per CU: 11.6 us for the same 33.5 MB), all 256 workgroups reach their epilogue at the same moment, and while
// they store no MFMA runs.  Here:
//   * the ring holds K-steps of 32 (24 KiB each, four slots = 96 KiB) and the other 64 KiB of the CU's LDS is a STAGING image of
//     one bf16 output tile;
//   * at the end of a tile the compute waves only scale, round and drop their accumulators into the staging image (16
//     ds_write_b128 each) and start the next tile's K loop -- its first K-steps are already in the ring;
//   * the loader waves, which spend most of their time waiting for copies, drain the staging image during the next tile's K loop:
//     two 1 KiB pieces (4 rows x 256 B, full lines) per K-step and wave -- read back, apply the epilogue (T5LayerNorm row scale,
//     ReLU, dropout, residual add, ReLU' mask, per-64-column sums of squares), store.  A store that has to wait for the fabric
//     delays a loader wave that has a K-step of slack, not an MFMA; the output stream of a tile is spread over the next tile's
//     K loop instead of arriving as one burst from every CU at once.
// Requirements (the launcher checks; otherwise p5_gemm5_kernel runs): whole 256x128 tiles, K % 64 == 0 and K >= 320, bf16 C with
// 16-byte rows, epilogue kinds store / ReLU(+dropout) / (dropout+) residual / ReLU' mask.
//
// Order of the vector-memory queue of a loader wave in K-step s (gfx9: one in-order counter for copies, loads and stores):
//   R_s  (row statistics / residual loads for the pieces drained in step s+1)   C_s (the six copies of K-step s+3)
//   wait: K-step s+1 landed, R_{s-1} arrived          barrier_s           S_s (the stores of the pieces drained now)
// so the wait may leave outstanding exactly C_s, C_{s-1} and whatever of S_{s-1}, R_s lies between them (counted, not guessed).
#pragma once
#include "p5_gemm5.h"

__device__ static __forceinline__ int g6_sigma(int q) { return (0x78 >> (2 * (q & 3))) & 3; }   // [0, 2, 3, 1]
// 16-byte slot of K-piece g in a 64-byte row: g ^ sigma; A rows are read 16 consecutive rows per fragment (quad index = row >> 2),
// B rows in p5_gemm4.h's permuted order (rows q*8 + k of a 32-row block: quad index = row >> 3).  With sigma = [0,2,3,1] each of
// the four 16-lane service groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) hits 16 different 16-byte bank groups.
__device__ static __forceinline__ int g6_sigma_a(int row) { return g6_sigma(row >> 2); }
__device__ static __forceinline__ int g6_sigma_b(int row) { return g6_sigma(row >> 3); }

#ifdef P5_EMU
#define P5_LAMBDA_INLINE
#else
#define P5_LAMBDA_INLINE __attribute__((always_inline))     // (a lambda hipcc declines to inline takes its captures -- the kernel
                                                              //  argument block, the register arrays -- through memory: scratch)
#endif
#ifdef P5_EMU
#define P5_WAIT_VM_DYN(base, extra) ((void)0)
#else
// s_waitcnt takes an immediate: the few values the loader's bookkeeping can ask for (rounded DOWN to what is instantiated -- waiting
// for more than necessary is always safe)
#define P5_WAIT_VM_DYN(base, extra)                                   \
  do {                                                                \
    if ((extra) >= 8) P5_WAIT_VM((base) + 8);                         \
    else if ((extra) >= 6) P5_WAIT_VM((base) + 6);                    \
    else if ((extra) >= 4) P5_WAIT_VM((base) + 4);                    \
    else if ((extra) >= 2) P5_WAIT_VM((base) + 2);                    \
    else P5_WAIT_VM(base);                                            \
  } while (0)
#endif

template <int ABL = 0>
__global__ __launch_bounds__(512) void p5_gemm6_kernel(P5GemmGroup grp) {
  using T = bf16;
  constexpr int BM = 256, BN = 128, NST = 4;
  constexpr int NWC = 4, NWL = 4;
  constexpr int WTM = 128, WTN = 64, TM = 8, TN = 4;
  constexpr int ASZ = BM * 64, STAGE = (BM + BN) * 64;                      // K-steps of 32: 64-byte rows
  constexpr int NDA = BM / (16 * NWL), NDB = BN / (16 * NWL), NDMA = NDA + NDB;   // 1 KiB copy = 16 rows x 64 B; per loader wave and K-step
  constexpr int PFD = NST - 1;
  constexpr int NMM = TM * TN;
  constexpr int RING = NST * STAGE, STG = BM * BN * 2;                      // staging: [256 rows][16 x 16 B], piece index XOR (row & 15)
  constexpr int NPC = BM * BN * 2 / 1024 / NWL;                             // 1 KiB staging pieces per loader wave (16)
  constexpr int DPS = 2;                                                    // pieces drained per K-step
  constexpr int DSTEPS = NPC / DPS;                                         // K-steps a drain takes (8); needs nk >= DSTEPS + 2
  static_assert(RING + STG <= 160 * 1024 && (PFD - 1) * NDMA + 16 <= 56 && DPS == 2 && NPC == 16, "LDS / vmcnt budget");
  __shared__ __attribute__((aligned(16))) char lds[RING + STG];
  char* const stg = lds + RING;

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif

  const int nwg = (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3, gx = nwg >> 3;
  const int upx = (grp.total_units + 7) >> 3;
  const int ulast = upx < grp.total_units - xcd * upx ? upx : grp.total_units - xcd * upx;
  const int nmy = ulast > jx ? (ulast - jx + gx - 1) / gx : 0;
  if (nmy <= 0) return;

  struct Unit { int pi, m0, n0, nk; };
  auto decode = [&](int it) P5_LAMBDA_INLINE {
    Unit u;
    const int id = xcd * upx + it * gx + jx;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < P5_MAX_GROUP; ++q)
      if (q < grp.nprob && id >= grp.unit_begin[q]) pi = q;
    const P5GemmArgs& g = grp.p[pi];
    const int tile = id - grp.unit_begin[pi];
    u.pi = pi;
    u.m0 = (tile / g.g4_tiles_n) * BM;
    u.n0 = (tile % g.g4_tiles_n) * BN;
    u.nk = g.g4_nk;                     // K-steps of 32 (the launcher's unit of g4_nk for this kernel)
    return u;
  };

  if (wave >= NWC) {
    // =========================================== loader waves ===========================================
    const int lw = wave - NWC;
    // the dropout seed, read before the ring starts (a tracked load in the loop would make hipcc wait for
    // every copy in flight: it cannot see the inline-asm copies in the queue)
    // (the launcher requires one RNG state for all problems of a launch)
    uint32_t seed0 = p5_seed(grp.p[0].drop);
#ifndef P5_EMU
    seed0 = __builtin_amdgcn_readfirstlane(seed0);
#endif
    const T* srcA[NDA];
    const T* srcB[NDB];
    int c_it = 0, c_left = 0;
    auto copy_setup = [&](int it) P5_LAMBDA_INLINE {
      const Unit u = decode(it);
      const P5GemmArgs& g = grp.p[u.pi];
      c_left = u.nk;
#pragma unroll
      for (int i = 0; i < NDA; ++i) {
        const int row = (lw * NDA + i) * 16 + (lane >> 2);
        srcA[i] = (const T*)g.A + (size_t)(u.m0 + row) * g.lda + (((lane & 3) ^ g6_sigma_a(row)) * 8);
      }
#pragma unroll
      for (int i = 0; i < NDB; ++i) {
        const int row = (lw * NDB + i) * 16 + (lane >> 2);
        srcB[i] = (const T*)g.B + (size_t)(u.n0 + row) * g.ldb + (((lane & 3) ^ g6_sigma_b(row)) * 8);
      }
    };
    auto copy_stage = [&](int buf) P5_LAMBDA_INLINE {
      char* b = lds + buf * STAGE;
      if constexpr ((ABL & 2) == 0) {
#pragma unroll
        for (int i = 0; i < NDA; ++i) glds16_raw(srcA[i], b + (lw * NDA + i) * 1024);
#pragma unroll
        for (int i = 0; i < NDB; ++i) glds16_raw(srcB[i], b + ASZ + (lw * NDB + i) * 1024);
      }
      if (--c_left > 0) {
#pragma unroll
        for (int i = 0; i < NDA; ++i) srcA[i] += 32;
#pragma unroll
        for (int i = 0; i < NDB; ++i) srcB[i] += 32;
      } else if (c_it + 1 < nmy) {
        copy_setup(++c_it);
      } else {
        c_left = 1;                 // past the last unit: keep re-fetching its last K-step into free slots (constant vmcnt bookkeeping)
      }
    };

    // ---- the tile being drained: descriptor scalars + this lane's row statistic ----
    struct Drain {
      T* C; const T* aux; float* ssq; const float* rowss;
      int epi, N, ldc, ldaux, ssq_nt, rowss_nt, m0, n0;
      uint32_t thr, hseed;
      float dscale, invd, eps;
      bool do_drop;
    } d;
    d.C = nullptr; d.aux = nullptr; d.ssq = nullptr; d.rowss = nullptr;
    d.epi = 0; d.N = 0; d.ldc = 0; d.ldaux = 0; d.ssq_nt = 0; d.rowss_nt = 0; d.m0 = 0; d.n0 = 0; d.thr = 0; d.hseed = 0;
    d.dscale = 1.f; d.invd = 0.f; d.eps = 0.f; d.do_drop = false;
    auto drain_setup = [&](const Unit& u) P5_LAMBDA_INLINE {
      const P5GemmArgs& g = grp.p[u.pi];
      d.C = (T*)g.C; d.aux = (const T*)g.aux; d.ssq = g.ssq_out; d.rowss = g.rowss;
      d.epi = g.epi; d.N = g.N; d.ldc = g.ldc; d.ldaux = g.ldaux; d.ssq_nt = g.ssq_nt; d.rowss_nt = g.rowss_nt; d.m0 = u.m0; d.n0 = u.n0;
      d.thr = g.drop.thr; d.dscale = g.drop.scale; d.invd = g.rowss_invd; d.eps = g.rowss_eps;
      d.do_drop = g.drop.state != nullptr && g.drop.thr != 0;
      const uint32_t sd = seed0;
      d.hseed = p5_mix32(sd + g.drop.site_key);
    };
    // row statistics of this wave's 64 tile rows, lane l <-> row lw*64 + l: issue (returns the number of vector-memory loads)
    u32x4 rs_part[4];            // (untracked loads -- gload16_raw -- completed by the counted waits below)
    float rstd_lane = 1.f;
    auto rowss_issue = [&]() P5_LAMBDA_INLINE -> int {
      if (!d.rowss) return 0;
      const int row = d.m0 + lw * 64 + lane;
      if (d.rowss_nt > 0 && (d.rowss_nt & 3) == 0 && d.rowss_nt <= 16) {
        const float* p = d.rowss + (size_t)row * d.rowss_nt;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (t * 4 < d.rowss_nt) gload16_raw(rs_part[t], p + t * 4);
        return d.rowss_nt >> 2;
      }
      // (other layouts: summed here and now -- a full wait, once per tile)
      float ss = 0.f;
      if (d.rowss_nt > 0) {
        const float* p = d.rowss + (size_t)row * d.rowss_nt;
        for (int t = 0; t < d.rowss_nt; ++t) ss += p[t];
      } else {
        ss = d.rowss[row];
      }
#ifndef P5_EMU
      asm volatile("" : "+v"(ss));     // (tracked loads: hipcc waits for them -- and with them for everything in flight -- HERE, in this
                                       //  branch, not at the join where the other branch's untracked loads are consumed)
#endif
      rs_part[0] = (u32x4){__builtin_bit_cast(uint32_t, ss), 0u, 0u, 0u};
      return 0;
    };
    auto rowss_finish = [&]() P5_LAMBDA_INLINE {
      if (!d.rowss) { rstd_lane = 1.f; return; }
      float ss = 0.f;
      if (d.rowss_nt > 0 && (d.rowss_nt & 3) == 0 && d.rowss_nt <= 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (t * 4 < d.rowss_nt) {
            const f32x4 v = __builtin_bit_cast(f32x4, rs_part[t]);
            ss = (((ss + v[0]) + v[1]) + v[2]) + v[3];
          }
      } else {
        ss = __builtin_bit_cast(float, rs_part[0][0]);
      }
      rstd_lane = rsqrtf(ss * d.invd + d.eps);
    };
    // piece q of this wave: tile rows lw*64 + q*4 + (lane >> 4), 16-byte column piece lane & 15.  The residual / saved-hidden
    // pieces of a whole tile (16 per wave, 64 registers -- the loader waves have them to spare) are requested in the tile's first
    // K-step and used from the second on.
    u32x4 auxa[NPC];
    auto aux_issue = [&]() P5_LAMBDA_INLINE -> int {
      if (!d.aux || d.epi == P5_EPI_STORE || d.epi == P5_EPI_RELU_DROP) return 0;
#pragma unroll
      for (int q = 0; q < NPC; ++q) {
        const int row = d.m0 + lw * 64 + q * 4 + (lane >> 4);
        gload16_raw(auxa[q], d.aux + (size_t)row * d.ldaux + d.n0 + (lane & 15) * 8);
      }
      return NPC;
    };
    auto drain_piece = [&](int q, const u32x4& auxv) P5_LAMBDA_INLINE {
      int le = lane;
#ifndef P5_EMU
      asm volatile("" : "+v"(le));      // (recompute the piece's offsets here: hoisted out of the K loop they cost 80 registers and spill --
                                        //  and a scratch reload is a vector-memory operation the copy bookkeeping does not count)
#endif
      const int lr = lw * 64 + q * 4 + (le >> 4), pc = le & 15;
      const int row = d.m0 + lr, col = d.n0 + pc * 8;
      float v[8];
      unpack16<T>(ld16(stg + lr * 256 + ((pc ^ (lr & 15)) << 4)), v);
      const float sc = __shfl(rstd_lane, q * 4 + (le >> 4));
      const bool has_aux = d.aux != nullptr;
      auto apply = [&](auto ek) P5_LAMBDA_INLINE {
        constexpr int EK = decltype(ek)::value;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = 0.f;
        if constexpr (EK >= 3) {
          if (has_aux) unpack16<T>(auxv, av);
        }
        const uint32_t idx0 = (uint32_t)(row * d.N + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = v[e] * sc;
          if constexpr (EK == 1 || EK == 2) x = x > 0.f ? x : 0.f;
          if constexpr (EK == 2 || EK == 4) x = (p5_mix32((idx0 + e) ^ d.hseed) >> 8) >= d.thr ? x * d.dscale : 0.f;
          if constexpr (EK == 3 || EK == 4) x += av[e];
          if constexpr (EK == 5) x = av[e] > 0.f ? x : 0.f;
          v[e] = x;
        }
      };
      if (d.epi == P5_EPI_RELU_DROP) { if (d.do_drop) apply(P5EpiTag<2>{}); else apply(P5EpiTag<1>{}); }
      else if (d.epi == P5_EPI_RESID_DROP) { if (d.do_drop) apply(P5EpiTag<4>{}); else apply(P5EpiTag<3>{}); }
      else if (d.epi == P5_EPI_MASK_POS) apply(P5EpiTag<5>{});
      else apply(P5EpiTag<0>{});
      const u32x4 packed = pack16<T>(v);
      if constexpr ((ABL & 16) != 0) { if (packed[0] == 0x12345678u) st16(d.C + (size_t)row * d.ldc + col, packed); }
      else st16(d.C + (size_t)row * d.ldc + col, packed);
      if (d.ssq) {               // (uniform) one partial per 64-column group = 8 adjacent lanes, exactly one writer
        float w[8], ss = 0.f;
        unpack16<T>(packed, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += w[e] * w[e];
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        const int cg = col >> 6;
        if ((pc & 7) == 0 && cg < d.ssq_nt) d.ssq[(size_t)row * d.ssq_nt + cg] = ss;
      }
    };

    copy_setup(0);
#pragma unroll
    for (int q = 0; q < PFD; ++q) copy_stage(q);
    P5_WAIT_VM((PFD - 1) * NDMA);           // K-step 0 has landed
    P5_BARRIER_LDS();
    // One K-step of the loader: C_g (the copies of K-step g+3 into the slot of K-step g-1, read out before barrier_{g-1}), the
    // counted wait (K-step g+1 = C_{g-2} landed; `extra` = the drain's vector-memory operations queued behind C_{g-2} other than
    // C_{g-1} and C_g), barrier_g.  The caller drains pieces AFTER it: a store that waits for the fabric then costs the loader
    // slack it has, not the compute waves' barrier.
    int buf = 0;                            // slot of K-step g
    auto kstep = [&](int extra) P5_LAMBDA_INLINE {
      const int nb3 = buf == 0 ? NST - 1 : buf - 1;
      copy_stage(nb3);
      if (extra >= 16) P5_WAIT_VM((PFD - 1) * NDMA + 16);
      else P5_WAIT_VM_DYN((PFD - 1) * NDMA, extra);
      P5_BARRIER_LDS();
      buf = buf == NST - 1 ? 0 : buf + 1;
    };
    for (int it = 0; it < nmy; ++it) {
      const int nk = decode(it).nk;
      if (it == 0) {
        for (int k = 0; k < nk; ++k) kstep(0);
        continue;
      }
      // K-step 0 of tile `it`: request what the drain of tile it-1 needs (R), then K-steps 1..8 drain two pieces each
      drain_setup(decode(it - 1));
      const int nR = rowss_issue() + aux_issue();
      kstep(nR);                            // behind C_{g-2}: C_{g-1}, R, C_g
      const int nS = DPS + (d.ssq ? DPS : 0);
      // (written out, not a loop: the piece index must be a compile-time constant for auxa[] to stay in registers)
      auto dstep = [&](auto jt) P5_LAMBDA_INLINE {
        constexpr int j = decltype(jt)::value;
        // behind C_{g-2}: [S_{g-2}] C_{g-1} [S_{g-1}] C_g; R lies behind C_{g-2} only in the first of these steps and must be
        // complete there too -- it is older than C_{g-1}, so the same count covers it
        kstep(j == 1 ? 0 : j == 2 ? nS : 2 * nS);
#ifndef P5_EMU
        if constexpr (j == 1) {
          // (the untracked loads' destinations are usable from here on: every later use depends on this statement)
          asm volatile("" : "+v"(rs_part[0]), "+v"(rs_part[1]), "+v"(rs_part[2]), "+v"(rs_part[3]));
          asm volatile("" : "+v"(auxa[0]), "+v"(auxa[1]), "+v"(auxa[2]), "+v"(auxa[3]), "+v"(auxa[4]), "+v"(auxa[5]), "+v"(auxa[6]), "+v"(auxa[7]));
          asm volatile("" : "+v"(auxa[8]), "+v"(auxa[9]), "+v"(auxa[10]), "+v"(auxa[11]), "+v"(auxa[12]), "+v"(auxa[13]), "+v"(auxa[14]), "+v"(auxa[15]));
        }
#endif
        if constexpr (j == 1) rowss_finish();
        drain_piece((j - 1) * DPS, auxa[(j - 1) * DPS]);
        drain_piece((j - 1) * DPS + 1, auxa[(j - 1) * DPS + 1]);
      };
      static_assert(DSTEPS == 8, "eight drain steps");
      dstep(P5EpiTag<1>{}); dstep(P5EpiTag<2>{}); dstep(P5EpiTag<3>{}); dstep(P5EpiTag<4>{});
      dstep(P5EpiTag<5>{}); dstep(P5EpiTag<6>{}); dstep(P5EpiTag<7>{}); dstep(P5EpiTag<8>{});
      kstep(2 * nS);
      for (int k = DSTEPS + 2; k < nk; ++k) kstep(k == DSTEPS + 2 ? nS : 0);
    }
    // the last tile: published by the final barrier, drained in one go
    P5_BARRIER_LDS();
    drain_setup(decode(nmy - 1));
    (void)rowss_issue();
    (void)aux_issue();
    P5_WAIT_VM(0);
#ifndef P5_EMU
    asm volatile("" : "+v"(rs_part[0]), "+v"(rs_part[1]), "+v"(rs_part[2]), "+v"(rs_part[3]));
    asm volatile("" : "+v"(auxa[0]), "+v"(auxa[1]), "+v"(auxa[2]), "+v"(auxa[3]), "+v"(auxa[4]), "+v"(auxa[5]), "+v"(auxa[6]), "+v"(auxa[7]));
    asm volatile("" : "+v"(auxa[8]), "+v"(auxa[9]), "+v"(auxa[10]), "+v"(auxa[11]), "+v"(auxa[12]), "+v"(auxa[13]), "+v"(auxa[14]), "+v"(auxa[15]));
#endif
    rowss_finish();
    drain_piece(0, auxa[0]); drain_piece(1, auxa[1]); drain_piece(2, auxa[2]); drain_piece(3, auxa[3]);
    drain_piece(4, auxa[4]); drain_piece(5, auxa[5]); drain_piece(6, auxa[6]); drain_piece(7, auxa[7]);
    drain_piece(8, auxa[8]); drain_piece(9, auxa[9]); drain_piece(10, auxa[10]); drain_piece(11, auxa[11]);
    drain_piece(12, auxa[12]); drain_piece(13, auxa[13]); drain_piece(14, auxa[14]); drain_piece(15, auxa[15]);
    P5_WAIT_VM(0);
    return;
  }

  // =========================================== compute waves ===========================================
  const int wm = wave >> 1, wn = wave & 1;
  const int rowa = wm * WTM + (lane & 15);
  const int offA = rowa * 64 + (((lane >> 4) ^ g6_sigma_a(rowa)) << 4);
  const int rowb = wn * WTN + ((lane & 15) >> 2) * 8 + (lane & 3);
  const int offB = ASZ + rowb * 64 + (((lane >> 4) ^ g6_sigma_b(rowb)) << 4);
  auto frag = [&](int buf, bool is_b, int t) P5_LAMBDA_INLINE -> u32x4 {
    const char* p = lds + buf * STAGE;
    if constexpr ((ABL & 4) != 0) { u32x4 z = {(unsigned)(buf + t), 1u, 2u, 3u}; return z; }
    const int delta = is_b ? ((t >> 1) * 32 + (t & 1) * 4) * 64 : t * 16 * 64;
    return ld16(p + (is_b ? offB : offA) + delta);
  };
  f32x4 acc[TM][TN];
  auto zero_acc = [&]() P5_LAMBDA_INLINE {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mm = [&](f32x4& a, const u32x4& fa, const u32x4& fb) P5_LAMBDA_INLINE {
    if constexpr ((ABL & 1) != 0) { a[0] += __builtin_bit_cast(float, fa[0] ^ fb[0]); }
    else mma16<T>(a, fb, fa);     // operands swapped: lane <- C[m = 16 i + (lane & 15)][4 columns]  (p5_gemm4.h)
  };
  // accumulators -> staging image: lane holds, per 16-row block i and 32-column half h, 8 consecutive columns of one row
  auto stage_out = [&](const Unit& u) P5_LAMBDA_INLINE {
    if constexpr ((ABL & 8) != 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) ((float*)grp.p[0].C)[0] = s;
      return;
    }
    const float alpha = grp.p[u.pi].alpha;
    int le = lane;
#ifndef P5_EMU
    asm volatile("" : "+v"(le));
#endif
    const int gl = le >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int lr = wm * WTM + i * 16 + (le & 15);
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * h][r] * alpha; v[4 + r] = acc[i][2 * h + 1][r] * alpha; }
        const int pc = wn * 8 + h * 4 + gl;
        st16(stg + lr * 256 + ((pc ^ (lr & 15)) << 4), pack16<T>(v));
      }
    }
  };

  u32x4 fa0[TM], fa1[TM], fb0[TN], fb1[TN];
  auto body = [&](u32x4(&fa)[TM], u32x4(&fb)[TN], u32x4(&na)[TM], u32x4(&nb)[TN], int nbuf) {
    P5_SCHED_FENCE();
#pragma unroll
    for (int t = 0; t < NMM; ++t) {
      const int i = t / TN, j = t % TN;
      mm(acc[i][j], fa[i], fb[j]);
      P5_SCHED_FENCE();
      if ((t & 1) == 0 && t / 2 < TM + TN) {
        const int r = t / 2;          // read order: B0 A0 B1 B2 B3 A1 .. A7
        if (r == 0) nb[0] = frag(nbuf, true, 0);
        else if (r == 1) na[0] = frag(nbuf, false, 0);
        else if (r < 1 + TN) nb[r - 1] = frag(nbuf, true, r - 1);
        else na[r - TN] = frag(nbuf, false, r - TN);
        P5_SCHED_FENCE();
      }
    }
  };

  P5_BARRIER_LDS();                       // K-step 0 has landed
#pragma unroll
  for (int j = 0; j < TN; ++j) fb0[j] = frag(0, true, j);
#pragma unroll
  for (int i = 0; i < TM; ++i) fa0[i] = frag(0, false, i);
  zero_acc();
  int buf = 0;
  for (int it = 0; it < nmy; ++it) {
    const Unit u = decode(it);
    int k = 0;
    do {                                  // two K-steps per trip (the launcher requires an even count): the fragment sets alternate
      const int nb1 = buf == NST - 1 ? 0 : buf + 1;
      const int nb2 = nb1 == NST - 1 ? 0 : nb1 + 1;
      P5_BARRIER_LDS();                   // K-step g+1 has landed; every wave has read K-step g's fragments (slot g is free)
      body(fa0, fb0, fa1, fb1, nb1);
      P5_BARRIER_LDS();
      body(fa1, fb1, fa0, fb0, nb2);
      buf = nb2;
      k += 2;
    } while (k < u.nk);
    stage_out(u);
    zero_acc();
  }
  P5_BARRIER_LDS();                       // publishes the last tile's staging image
}
