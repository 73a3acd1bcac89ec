// p5_gemm5.h -- wave-specialised persistent GEMM: four LOADER waves feed a ring of K-steps in LDS, four COMPUTE waves multiply.
//
// Same contract and work-unit scheduling as p5_gemm4.h (bias-free nn.Linear forward / dgrad / wgrad of the T5 layers, HF
// modeling_t5.py:83-94,205-208,304,325-326,367 and their autograd transposes).  Why another kernel: the ablated builds of
// p5_gemm4_kernel (tools/lab, round 3) show that what limits every kernel here whose waves both copy and multiply is the ISSUE
// cost of the direct-to-LDS copy -- one `global_load_lds_dwordx4` wave instruction moves 1 KiB and occupies its wave's (in-order)
// issue for 60-185 cycles (MI355X_MICROARCH.md, per-instruction constants), the time of 4-11 MFMAs; with the copies in the MFMA
// stream the chip never moves more than ~14 TB/s from L2 into LDS, whatever the ring depth.  The two instruction kinds issue in the
// same cycle only from DIFFERENT waves of a SIMD.  Hence:
//   * waves 4-7 (one per SIMD) only copy: each issues its 12 pieces of a 256x128 K-step (48 KiB) and waits for its share of the
//     previous one; waves 0-3 (one per SIMD) only read fragments and issue MFMAs on 128x64 wave tiles -- 128 accumulator registers,
//     0.375 fragment reads per MFMA instead of 0.5, which also takes the compute waves off the LDS-bandwidth bound of the 64x64
//     wave tile;
//   * one LDS-only barrier per K-step joins the two groups (copies of K-step s+1 landed / slot of K-step s-1 free);
//   * the loaders run ahead of the compute waves by the depth of the ring, across work-unit boundaries: while the compute waves
//     write a finished tile out (straight from the accumulators, as in p5_gemm4.h) the next unit's first K-steps are landing,
//     and the epilogue's stores / residual loads share no vmcnt with any copy.
#pragma once
#include "p5_gemm4.h"

// VAR: 0 = the product instances; 1 = the instance that carries the gated-GELU epilogues (P5_EPI_GELU_GATE / _BWD); 2 = the logit-free
// cross-entropy epilogues (P5_EPI_CE_STATS / _GRAD); 3 = the T5LayerNorm-backward epilogue (P5_EPI_NORM_BWD, 128-row tiles) -- the product
// instances do not pay for their registers
// BMT: rows of a tile.  256 = the product shape above.  128 (round 6) = the N = d_model outputs of the encoder (8192 x 512 is 256 tiles of
// 128 x 128 but only 128 of 256 x 128 -- half the chip): 2 x 2 compute waves on 64 x 64 wave tiles, four-slot ring of 32 KiB K-steps, the
// same loader / compute split, so the copies' issue cost stays out of the MFMA stream (p5_gemm2_kernel<128,128,4>, whose four waves do both,
// moves 8 TB/s into LDS on these shapes).
template <bool KS, int ABL = 0, int VAR = 0, int BMT = 256>
__global__ __launch_bounds__(512) void p5_gemm5_kernel(P5GemmGroup grp) {
  constexpr bool GATE = VAR == 1;
  using T = bf16;
  static_assert(BMT == 256 || BMT == 128, "tile rows");
  constexpr int BM = BMT, BN = 128, NST = BMT == 256 ? 3 : 4;
  constexpr int NWC = 4, NWL = 4;                          // compute waves (2 x 2), loader waves
  constexpr int WTM = BM / 2, WTN = 64, TM = WTM / 16, TN = 4;
  constexpr int RPG = TM / 4;                              // row blocks whose statistics a lane group fetches in the whole-tile epilogue
  constexpr int ASZ = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NDA = BM / (8 * NWL), NDB = BN / (8 * NWL), NDMA = NDA + NDB;   // copy instructions per loader wave per K-step
  constexpr int PFD = NST - 1;
  constexpr int NMM = TM * TN;
  static_assert(NST * STAGE <= 160 * 1024 && PFD * NDMA <= 56, "ring geometry");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef P5_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif

  // ---- this workgroup's units (as p5_gemm4.h): XCD x = blockIdx % 8 owns units [x*UPX, (x+1)*UPX), its workgroups take them round-robin ----
  const int nwg = (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3, gx = nwg >> 3;
  const int upx = (grp.total_units + 7) >> 3;
  const int ulast = upx < grp.total_units - xcd * upx ? upx : grp.total_units - xcd * upx;
  const int nmy = ulast > jx ? (ulast - jx + gx - 1) / gx : 0;
  if (nmy <= 0) return;

  struct Unit { int pi, m0, n0, kb, nk; };
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
    int tm = tile / g.g4_tiles_n, tn = tile % g.g4_tiles_n;
    if (g.g4_cb > 0) {
      // Rectangular blocks per XCD (round 6; one problem, no split-K, 32 workgroups per XCD: the launcher checks).  With the tiles of an
      // XCD's range dealt n-fastest, the 32 workgroups of an XCD sit on 32 / tiles_n tile rows x ALL column tiles: at N = 4096 that is one
      // A panel and the whole of B (8 MB against 4 MB of L2) per round.  As a (32 / cb) x cb block they share 32 / cb A panels and cb B
      // panels: half the bytes through the fabric, and what the K = 4096, N = 1024 shapes -- whose 8 column tiles make that block by
      // themselves -- run at (1.11 against 0.78 PFLOP/s at T5-large dims, profiles/r06_call11_c3_c5_kernel_tables.txt).
      const int il = it * gx + jx, blk = il >> 5, w = il & 31;
      const int nbc = g.g4_tiles_n / g.g4_cb, rb = 32 / g.g4_cb;
      tm = xcd * (upx / g.g4_tiles_n) + (blk / nbc) * rb + w / g.g4_cb;
      tn = (blk % nbc) * g.g4_cb + w % g.g4_cb;
    }
    u.pi = pi;
    u.m0 = tm * BM;
    u.n0 = tn * BN;
    u.nk = g.g4_nk;
    u.kb = sp * g.g4_nk * 64;
    return u;
  };

  if (wave >= NWC) {
    // =========================================== loader waves ===========================================
    const int lw = wave - NWC;
    const T* srcA[NDA];
    const T* srcB[NDB];
    size_t incA = 0, incB = 0;
    int c_it = 0, c_left = 0;
    auto copy_setup = [&](int it) {
      const Unit u = decode(it);
      const P5GemmArgs& g = grp.p[u.pi];
      c_left = u.nk;
      if constexpr (KS) {
        incA = (size_t)64 * g.lda; incB = (size_t)64 * g.ldb;
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
          constexpr int CPR = BM / 8, RPI = 512 / BM;
          const int krow = (lw * NDA + i) * RPI + lane / CPR;
          int cg = (lane % CPR) ^ ksd_swz<BM>(krow);
          const int cmax = (g.lda - u.m0) / 8 - 1;        // (row capacity of the k-row in memory, see stage_dma_ks)
          cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
          srcA[i] = (const T*)g.A + ((size_t)u.kb + krow) * g.lda + u.m0 + cg * 8;
        }
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
          constexpr int CPR = BN / 8, RPI = 512 / BN;
          const int krow = (lw * NDB + i) * RPI + lane / CPR;
          int cg = (lane % CPR) ^ ksd_swz<BN>(krow);
          const int cmax = (g.ldb - u.n0) / 8 - 1;
          cg = cg < cmax ? cg : (cmax > 0 ? cmax : 0);
          srcB[i] = (const T*)g.B + ((size_t)u.kb + krow) * g.ldb + u.n0 + cg * 8;
        }
      } else {
        incA = 64; incB = 64;
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
          const int row = (lw * NDA + i) * 8 + (lane >> 3);
          int gr = u.m0 + row;
          gr = gr < g.M ? gr : g.M - 1;
          srcA[i] = (const T*)g.A + (size_t)gr * g.lda + u.kb + (((lane & 7) ^ g4_sigma_a(row)) * 8);
        }
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
          const int row = (lw * NDB + i) * 8 + (lane >> 3);
          int gr = u.n0 + row;
          gr = gr < g.N ? gr : g.N - 1;
          if constexpr (GATE)
            if (g.gate_F > 0) gr = ((gr & 32) ? g.gate_F : 0) + ((gr >> 6) << 5) + (gr & 31);      // [wi_0; wi_1] read gate-interleaved (p5_gemm.h)
          srcB[i] = (const T*)g.B + (size_t)gr * g.ldb + u.kb + (((lane & 7) ^ g4_sigma_b(row)) * 8);
        }
      }
    };
    auto copy_stage = [&](int buf) {
      char* b = lds + buf * STAGE;
      if constexpr ((ABL & 2) == 0) {
#pragma unroll
        for (int i = 0; i < NDA; ++i) glds16_raw(srcA[i], b + (lw * NDA + i) * 1024);
#pragma unroll
        for (int i = 0; i < NDB; ++i) glds16_raw(srcB[i], b + ASZ + (lw * NDB + i) * 1024);
      }
      if (--c_left > 0) {
#pragma unroll
        for (int i = 0; i < NDA; ++i) srcA[i] += incA;
#pragma unroll
        for (int i = 0; i < NDB; ++i) srcB[i] += incB;
      } else if (c_it + 1 < nmy) {
        copy_setup(++c_it);
      } else {
        c_left = 1;                 // past the last unit: keep re-fetching its last K-step into free slots (constant vmcnt bookkeeping)
      }
    };
    int total = 0;
    for (int it = 0; it < nmy; ++it) total += decode(it).nk;
    copy_setup(0);
#pragma unroll
    for (int q = 0; q < PFD; ++q) copy_stage(q);
    P5_WAIT_VM((PFD - 1) * NDMA);           // K-step 0 has landed
    P5_BARRIER_LDS();
    int buf = 0;
    for (int g = 0; g < total; ++g) {
      const int nb2 = buf == 0 ? NST - 1 : buf - 1;
      copy_stage(nb2);                      // K-step g+PFD -> the slot of K-step g-1 (read out before the barrier every wave has passed)
      P5_WAIT_VM((PFD - 1) * NDMA);         // this wave's share of K-step g+1 has landed
      P5_BARRIER_LDS();
      buf = buf == NST - 1 ? 0 : buf + 1;
    }
    P5_WAIT_VM(0);
    if constexpr (VAR == 3 && !KS) {
      // P5_EPI_NORM_BWD: n = w * round(x rstd) needs nothing from the accumulators -- the loader waves, idle from here on, write it (x read +
      // n written = 64 of the 224 KiB a tile's epilogue moves) while the compute waves run the rest.  A 128 x 128 tile is 2048 16-byte
      // pieces: eight per loader lane, all loads of a tile before its first store.
      static_assert(BM == 128 || VAR != 3, "n tiles of 128 rows");
      for (int it = 0; it < nmy; ++it) {
        const Unit u = decode(it);
        const P5GemmArgs& g = grp.p[u.pi];
        if (g.C2 == nullptr) continue;
        const int N = g.N, col = u.n0 + (lane & 15) * 8;
        const T* const xb = (const T*)g.aux;
        T* const nb = (T*)g.C2;
        const f32x4 w0 = *(const f32x4*)(g.nb_w + col), w1 = *(const f32x4*)(g.nb_w + col + 4);
        u32x4 xv[8];
        float rs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = u.m0 + j * 16 + lw * 4 + (lane >> 4);
          xv[j] = ld16(xb + ((uint32_t)row * (uint32_t)N + (uint32_t)col));
          rs[j] = gemm_row_rstd(g, row);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = u.m0 + j * 16 + lw * 4 + (lane >> 4);
          float x[8], nr[8];
          unpack16<T>(xv[j], x);
#pragma unroll
          for (int e = 0; e < 8; ++e) nr[e] = (e < 4 ? w0[e & 3] : w1[e & 3]) * to_f<T>(from_f<T>(x[e] * rs[j]));
          st16(nb + ((uint32_t)row * (uint32_t)N + (uint32_t)col), pack16<T>(nr));
        }
      }
    }
    return;
  }

  // =========================================== compute waves ===========================================
  const int wm = wave >> 1, wn = wave & 1;
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

  f32x4 acc[TM][TN];
#define P5_G5_ROWBLOCKS(F) do { F(0); F(1); F(2); F(3); if constexpr (TM > 4) { F(4); F(5); F(6); F(7); } } while (0)
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mm = [&](f32x4& a, const u32x4& fa, const u32x4& fb) {
    if constexpr ((ABL & 1) != 0) { a[0] += __builtin_bit_cast(float, fa[0] ^ fb[0]); }
    else mma16<T>(a, fb, fa);     // operands swapped: lane <- C[m = 16 i + (lane & 15)][4 columns]  (p5_gemm4.h)
  };

  // one 16-row block of the 128x64 wave tile, constant row-block index (p5_gemm4.h epi_rows)
  auto epi_rows = [&](const f32x4(&a)[TN], int i, const Unit& u, const P5GemmArgs& g, uint32_t seed, bool do_drop, bool vec_ok, int le) {
    const int gl = le >> 4;
    const int row = u.m0 + wm * WTM + i * 16 + (le & 15);
    if constexpr (KS) {
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
          float av[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) av[e] = 0.f;
          if (g.epi != P5_EPI_STORE && g.epi != P5_EPI_ATOMIC && g.epi != P5_EPI_ACCUM) {
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
            if (g.ssq_out) {
              float w[8];
              unpack16<T>(packed, w);
#pragma unroll
              for (int e = 0; e < 8; ++e) if (col + e < g.N) ss += g.epi == P5_EPI_MASK_POS ? w[e] * av[e] : w[e] * w[e];
            }
          }
        }
      }
      if (g.ssq_out) {          // (uniform) this wave is the only writer of the row's partial for its 64-column group
        if (g.epi == P5_EPI_MASK_POS) ss *= 1.f / g.alpha;      // <d pre, pre>: the saved hidden is pre * alpha where it is positive
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (g.ssq_nt > 0) {
          const int cg = (u.n0 + wn * WTN) >> 6;
          if (gl == 0 && row_ok && cg < g.ssq_nt) g.ssq_out[(size_t)row * g.ssq_nt + cg] = ss;
        } else if (gl == 0 && row_ok) {
          atomicAdd(g.ssq_out + row, ss);
        }
      }
    }
  };
  // what the whole-tile epilogue needs from the problem descriptor, as scalars
  struct EpiCtx {
    void* C; const void* aux; float* ssq; const float* rowss; void* C2;
    int epi, N, ldc, ldaux, ssq_nt, rowss_nt, ldc2, gate_F;
    uint32_t thr, hseed;
    float dscale, alpha, invd, eps;
    bool fast, fast32, do_drop;
  };
  auto load_ctx = [&](const Unit& u) {
    const P5GemmArgs& g = grp.p[u.pi];
    EpiCtx c;
    c.C = g.C; c.aux = g.aux; c.ssq = g.ssq_out; c.rowss = g.rowss; c.C2 = g.C2;
    c.epi = g.epi; c.N = g.N; c.ldc = g.ldc; c.ldaux = g.ldaux; c.ssq_nt = g.ssq_nt; c.rowss_nt = g.rowss_nt; c.ldc2 = g.ldc2; c.gate_F = g.gate_F;
    c.thr = g.drop.thr; c.dscale = g.drop.scale; c.alpha = g.alpha; c.invd = g.rowss_invd; c.eps = g.rowss_eps;
    c.do_drop = g.drop.state != nullptr && g.drop.thr != 0;
    c.hseed = p5_mix32(p5_seed(g.drop) + g.drop.site_key);
    const bool inside = u.m0 + BM <= g.M && u.n0 + BN <= g.N;
    if constexpr (KS) {
      c.fast = inside && g.c_f32 && (g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.epi == P5_EPI_ACCUM || g.epi == P5_EPI_STORE);
    } else {
      const bool vec_ok = (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.aux == nullptr || ((g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0));
      c.fast = inside && vec_ok && !g.c_f32 && g.epi != P5_EPI_ATOMIC && g.epi != P5_EPI_ACCUM && (g.ssq_out == nullptr || g.ssq_nt > 0);
      if (g.epi == P5_EPI_GELU_GATE) c.fast = c.fast && g.C2 != nullptr && (g.ldc2 & 7) == 0 && ((uintptr_t)g.C2 & 15) == 0 && g.gate_F * 2 == g.N && g.ssq_out == nullptr;
      if (g.epi == P5_EPI_GELU_GATE_BWD) c.fast = c.fast && g.aux != nullptr && g.rowss == nullptr && g.ssq_out == nullptr;
      // plain fp32 store of a whole tile (the tied head's logits: 66 MB per step): straight from the accumulators, two 16-byte stores
      // per lane and row block -- the general path below re-reads the descriptor per element (87 us for the 512 x 32100 x 512 head GEMM)
      c.fast32 = inside && g.c_f32 && g.epi == P5_EPI_STORE && (g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && g.ssq_out == nullptr && g.rowss == nullptr;
    }
    if constexpr (KS) c.fast32 = false;
    return c;
  };
  // P5_EPI_NORM_BWD: the row scalars of a tile, formed BEFORE its K loop -- the accumulators are dead there, so the partial sums may pass
  // through as many registers as they like, and the first K-step is still on its way.  Lane group gl holds the rows of row blocks
  // RPG gl .. RPG gl + RPG - 1 (the epilogue fetches them by shuffle): rstd from the d/64 partial sums of squares the producer of x left
  // behind, and the mean of <dn w, xh> from the partial sums of <dOut, Out> (index order, as everywhere: the same bits every run).
  float nb_rstd[2] = {0.f, 0.f}, nb_dotm[2] = {0.f, 0.f};
  auto nb_stats = [&](const Unit& u) {
    const P5GemmArgs& g = grp.p[u.pi];
    const int gl = lane >> 4, li = lane & 15;
#pragma unroll
    for (int q = 0; q < RPG; ++q) {
      const int row = u.m0 + wm * WTM + (gl * RPG + q) * 16 + li;
      nb_rstd[q] = gemm_row_rstd(g, row < g.M ? row : g.M - 1);
      const float* p = g.nb_dot + (size_t)(row < g.M ? row : g.M - 1) * g.nb_dot_nt;
      float sm = 0.f;
      int t = 0;
      if ((g.nb_dot_nt & 3) == 0) {
#pragma unroll 8
        for (; t < g.nb_dot_nt; t += 4) {
          const f32x4 v = *(const f32x4*)(p + t);
          sm = (((sm + v[0]) + v[1]) + v[2]) + v[3];
        }
      }
      for (; t < g.nb_dot_nt; ++t) sm += p[t];
      nb_dotm[q] = sm * g.rowss_invd;          // (rowss_invd = 1 / d_model = 1 / N)
    }
  };
  auto epilogue = [&](const Unit& u, const EpiCtx& cx) {
    if constexpr ((ABL & 8) != 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) ((float*)grp.p[0].C)[0] = s;
      return;
    }
    int le = lane;
#ifndef P5_EMU
    asm volatile("" : "+v"(le));      // (keeps hipcc from hoisting every lane-dependent output offset out of the unit loop)
#endif
    if constexpr (VAR == 2 && !KS) {
      // ---- logit-free cross-entropy (tied head, P5_T5.py:352-369): the [rows, V] logits never reach memory.  A lane holds 16 of the 64
      // columns of its wave tile for one row per row block; the row's (max, sum exp) over those 64 columns is two shuffles away.
      const P5GemmArgs& g = grp.p[u.pi];
      const int M = g.M, N = g.N, gl = le >> 4;
      const float alpha = g.alpha;
      const int cw = u.n0 + wn * WTN;                       // first column of the wave tile
      const int row0 = u.m0 + wm * WTM + (le & 15);
      if (cw >= N) return;
      const bool stats = g.epi == P5_EPI_CE_STATS;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = row0 + i * 16;
        const bool row_ok = row < M;
        const long long lab = row_ok ? (long long)g.ce_labels[row] : -1;
        float v[16];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[h * 8 + e] = acc[i][2 * h + (e >> 2)][e & 3] * alpha;
        if (stats) {
          float m = P5_NEG_INF;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int col = cw + (q >> 3) * 32 + gl * 8 + (q & 7);
            if (col < N) m = fmaxf(m, v[q]);
            if (row_ok && col == lab) g.ce_lab[row] = v[q];          // (exactly one lane of the launch per row)
          }
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          float sum = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int col = cw + (q >> 3) * 32 + gl * 8 + (q & 7);
            if (col < N) sum += p5_exp<T>(v[q] - m);
          }
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          if (gl == 0 && row_ok) {
            float* pp = g.ce_part + ((size_t)row * g.ce_np + (cw >> 6)) * 2;
            pp[0] = m; pp[1] = sum;
          }
        } else {
          const float lse = row_ok ? g.ce_lse[row] : 0.f, gg = row_ok ? g.ce_g[row] : 0.f;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int col = cw + h * 32 + gl * 8;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (col + e < N) ? (p5_exp<T>(v[h * 8 + e] - lse) - (col + e == lab ? 1.f : 0.f)) * gg : 0.f;
            if (row_ok && col < g.ldc) st16((T*)g.C + (size_t)row * g.ldc + col, pack16<T>(o));      // (ldc % 8 == 0: padded columns N .. ldc get exact zeros)
          }
        }
      }
      return;
    }
    if constexpr (VAR == 3 && !KS) {
      const P5GemmArgs& g = grp.p[u.pi];
      {   // (a launch of this instance carries P5_EPI_NORM_BWD problems only: the launcher checks)
        // ---- T5LayerNorm backward of the sub-layer's input (P5_EPI_NORM_BWD, p5_gemm.h): whole tiles only (the launcher checks).  The row
        // scalars (rstd, mean of <dn w, xh>) were formed before the K loop (nb_stats); a lane holds, per row block, 2 x 8 consecutive columns.
        // (n = w * round(x rstd) is written by the loader waves, above)
        // every [M, N] operand has leading dimension N (the launcher checks): ONE 32-bit element offset per lane, uniform bases and row-block
        // strides -- five 64-bit lane pointers x four row blocks cost the registers this epilogue does not have
        const int N = g.N;
        const int gl = le >> 4, li = le & 15;
        const int row = u.m0 + wm * WTM + li;
        const int col0 = u.n0 + wn * WTN + gl * 8;
        const uint32_t eo = (uint32_t)row * (uint32_t)N + (uint32_t)col0, bs = 16u * (uint32_t)N;
        const T* const xb = (const T*)g.aux;
        const float* const rib = g.nb_rin;
        float* const rob = g.nb_rout;
        T* const yb = (T*)g.C;
        const bool do_drop = g.drop.state != nullptr && g.drop.thr != 0;
        const uint32_t thr = g.drop.thr, hseed = p5_mix32(p5_seed(g.drop) + g.drop.site_key);
        const float dscale = g.drop.scale;
        float wv[2][8], dwv[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 w0 = *(const f32x4*)(g.nb_w + col0 + h * 32), w1 = *(const f32x4*)(g.nb_w + col0 + h * 32 + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { wv[h][r] = w0[r]; wv[h][4 + r] = w1[r]; dwv[h][r] = 0.f; dwv[h][4 + r] = 0.f; }
        }
        // loads of the tile: the x rows and the fp32 residual gradient run two row blocks ahead (the next unit's first fragments stay in their
        // registers across the epilogue: what is left holds two blocks of loads and one of results)
        constexpr int AD = 2, RD = 2;
        u32x4 xv[AD][2];
        f32x4 rv[RD][2][2];
        auto x_load = [&](int i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) xv[i % AD][h] = ld16(xb + (eo + (uint32_t)i * bs + (uint32_t)h * 32u));
        };
        auto r_load = [&](int i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            rv[i % RD][h][0] = *(const f32x4*)(rib + (eo + (uint32_t)i * bs + (uint32_t)h * 32u));
            rv[i % RD][h][1] = *(const f32x4*)(rib + (eo + (uint32_t)i * bs + (uint32_t)h * 32u + 4u));
          }
        };
#pragma unroll
        for (int i = 0; i < AD && i < TM; ++i) x_load(i);
#pragma unroll
        for (int i = 0; i < RD && i < TM; ++i) r_load(i);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float rs = __shfl(nb_rstd[i % RPG], li + 16 * (i / RPG)), dm = __shfl(nb_dotm[i % RPG], li + 16 * (i / RPG));
          u32x4 po[2];
          f32x4 ro[2][2];
          // the keep decisions of the block's 16 elements first, as two bit masks: the hash chains need a dozen temporaries each, and
          // left to itself the compiler interleaves them with the arithmetic below (46 spilled registers)
          uint32_t km[2] = {0xFFu, 0xFFu};
          if (do_drop) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t idx0 = (uint32_t)((row + i * 16) * N + col0 + h * 32);
              uint32_t m = 0;
#pragma unroll
              for (int e = 0; e < 8; ++e) m |= ((p5_mix32((idx0 + e) ^ hseed) >> 8) >= thr ? 1u : 0u) << e;
              km[h] = m;
            }
          }
#ifndef P5_EMU
          asm volatile("" : "+v"(km[0]), "+v"(km[1]));
#endif
          const float ds = do_drop ? dscale : 1.f;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float x[8], o[8];
            unpack16<T>(xv[i % AD][h], x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dy = acc[i][2 * h + (e >> 2)][e & 3];
              const float xh = x[e] * rs;
              dwv[h][e] += dy * xh;
              const float v = rs * (dy * wv[h][e] - xh * dm) + rv[i % RD][h][e >> 2][e & 3];
              ro[h][e >> 2][e & 3] = v;
              o[e] = ((km[h] >> e) & 1u) ? v * ds : 0.f;
            }
            po[h] = pack16<T>(o);
          }
          if (i + AD < TM) x_load(i + AD);
          if (i + RD < TM) r_load(i + RD);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t o = eo + (uint32_t)i * bs + (uint32_t)h * 32u;
            *(f32x4*)(rob + o) = ro[h][0];
            *(f32x4*)(rob + (o + 4u)) = ro[h][1];
            st16(yb + o, po[h]);
          }
        }
        // norm-weight gradient: this wave's 64 rows summed per column (16 lanes of a lane group hold the same columns), one partial row
        // per wave row block of 64, reduced later in row order (p5_reduce_rows_multi_kernel) -- no atomics
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = dwv[h][e];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            t += __shfl_xor(t, 8);
            dwv[h][e] = t;
          }
        if (li == 0) {
          float* const dp = g.nb_dw + (size_t)((u.m0 + wm * WTM) >> 6) * N + col0;
          static_assert(WTM == 64 || VAR != 3, "norm-backward epilogue: one partial row per 64-row wave tile");
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            *(f32x4*)(dp + h * 32) = (f32x4){dwv[h][0], dwv[h][1], dwv[h][2], dwv[h][3]};
            *(f32x4*)(dp + h * 32 + 4) = (f32x4){dwv[h][4], dwv[h][5], dwv[h][6], dwv[h][7]};
          }
        }
        return;
      }
    }
    // Fast path: a whole tile inside the output, 16-byte stores.  Everything the eight row blocks need from the problem descriptor
    // was read ONCE into scalars (EpiCtx): the general code below reads descriptor
    // fields where it uses them, and with the accumulators holding the scalar registers' spill space hipcc re-issues those kernarg
    // loads (s_load + lgkmcnt(0), ~100 ns each) in every row block -- measured 7-9 us per 256x128 tile, more than the tile's
    // MFMAs (tools/lab lab4, round 3).
    if constexpr (!KS) {
      if (cx.fast32) {
        const float alpha = cx.alpha;
        float* cp = (float*)cx.C + (size_t)(u.m0 + wm * WTM + (le & 15)) * cx.ldc + (u.n0 + wn * WTN + (le >> 4) * 8);
        const size_t step = (size_t)16 * cx.ldc;
        auto rows32 = [&](const f32x4(&a)[TN]) {
#pragma unroll
          for (int h = 0; h < TN / 2; ++h) {
            *(f32x4*)(cp + h * 32) = (f32x4){a[2 * h][0] * alpha, a[2 * h][1] * alpha, a[2 * h][2] * alpha, a[2 * h][3] * alpha};
            *(f32x4*)(cp + h * 32 + 4) = (f32x4){a[2 * h + 1][0] * alpha, a[2 * h + 1][1] * alpha, a[2 * h + 1][2] * alpha, a[2 * h + 1][3] * alpha};
          }
          cp += step;
        };
#define P5_G5_R32(i) rows32(acc[i])
        P5_G5_ROWBLOCKS(P5_G5_R32);
#undef P5_G5_R32
        return;
      }
    }
    if (cx.fast) {
      if constexpr (!KS) {
        const int epi = cx.epi, N = cx.N, ldc = cx.ldc, ldaux = cx.ldaux;
        float* const ssq = cx.ssq;
        const int ssq_nt = cx.ssq_nt;
        const uint32_t thr = cx.thr, hseed = cx.hseed;
        const float dscale = cx.dscale, alpha = cx.alpha, inv_alpha = 1.f / cx.alpha;
        const float* const rowss = cx.rowss;
        const int rowss_nt = cx.rowss_nt;
        const float invd = cx.invd, eps = cx.eps;
        const int gl = le >> 4;
        const int row = u.m0 + wm * WTM + (le & 15);
        const int col0 = u.n0 + wn * WTN + gl * 8;
        T* const cp = (T*)cx.C + (size_t)row * ldc + col0;
        const T* ap = cx.aux ? (const T*)cx.aux + (size_t)row * ldaux + col0 : nullptr;
        // EK: the epilogue kind as a compile-time constant (0 store, 1 ReLU, 2 ReLU + dropout, 3 + residual, 4 dropout + residual,
        // 5 ReLU' mask) -- chosen ONCE per tile below: a per-element `if (epi == ...)` chain compiles to scalar branches (the
        // dropout hash keeps hipcc from if-converting it), ~5 per element, 640 per tile.
        //
        // Round 5: EVERY load of the tile goes out before its first store.  The in-step ablation (profiles/r05_ablations_in_step.txt) put
        // 16 us of the 41 us of an 8192x2048x512 launch into "epilogue arithmetic": the eight row blocks each loaded their row statistics
        // and aux rows, waited, computed and stored -- and because vmcnt counts stores as well as loads on this part, the wait for row
        // block i+1's loads was also a wait for row block i's stores to be acknowledged: eight dependent memory round trips per tile.
        // Now: lane group gl fetches the statistics of row blocks RPG gl .. RPG gl + RPG - 1 (RPG = 2 on 256-row tiles; 16-byte loads, summed
        // in index order as everywhere else) and the scale reaches the other lane groups by a shuffle; the aux rows run through a
        // four-block ring (below).
        float sc2[2] = {alpha, alpha};
        if (rowss) {
          const int rb = row + gl * 16 * RPG;                 // row of block RPG gl for this lane
          if (rowss_nt > 0 && (rowss_nt & 3) == 0 && rowss_nt <= 16) {
            f32x4 pv[2][4];
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
              const float* p = rowss + (size_t)(rb + q * 16) * rowss_nt;
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (t * 4 < rowss_nt) pv[q][t] = *(const f32x4*)(p + t * 4);
            }
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
              float ss = 0.f;
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (t * 4 < rowss_nt) ss = (((ss + pv[q][t][0]) + pv[q][t][1]) + pv[q][t][2]) + pv[q][t][3];
              sc2[q] = alpha * rsqrtf(ss * invd + eps);
            }
          } else {
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
              float ss = 0.f;
              if (rowss_nt > 0) {
                const float* p = rowss + (size_t)(rb + q * 16) * rowss_nt;
                for (int t = 0; t < rowss_nt; ++t) ss += p[t];
              } else {
                ss = rowss[rb + q * 16];
              }
              sc2[q] = alpha * rsqrtf(ss * invd + eps);
            }
          }
        }
        auto all_rows = [&](auto ek) {
          constexpr int EK = decltype(ek)::value;
          // aux rows (residual / saved hidden): a ring of four row blocks' loads in flight.  Program order per block: use block i's aux ->
          // request block i+4's into the same registers -> store block i.  A later wait for block i+4's loads then never includes block
          // i's stores (they were issued after it); the stores it does include are four blocks old.
          constexpr int AD = 4;
          u32x4 auxv[AD][TN / 2];
          auto aux_load = [&](int i) {
#pragma unroll
            for (int h = 0; h < TN / 2; ++h)
              auxv[i % AD][h] = (ap && (ABL & 64) == 0) ? ld16(ap + (size_t)i * 16 * ldaux + h * 32) : (u32x4){0u, 0u, 0u, 0u};      // (ABL 64, lab: no residual / saved-hidden read)
          };
          if constexpr (EK >= 3) {
#pragma unroll
            for (int i = 0; i < AD; ++i) aux_load(i);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            // scale of row block i: held by lane group i / RPG (same row-in-block)
            const float sc = rowss ? __shfl(sc2[i % RPG], (le & 15) + 16 * (i / RPG)) : alpha;
            float sq = 0.f;
            u32x4 packed[TN / 2];
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
              float v[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * h][r] * sc; v[4 + r] = acc[i][2 * h + 1][r] * sc; }
              float av[8];
              if constexpr (EK != 0) {
                if constexpr (EK >= 3) unpack16<T>(auxv[i % AD][h], av);
                const uint32_t idx0 = (uint32_t)((row + i * 16) * N + col0 + h * 32);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  float x = v[e];
                  if constexpr (EK == 1 || EK == 2) x = x > 0.f ? x : 0.f;
                  if constexpr ((EK == 2 || EK == 4) && (ABL & 32) == 0) x = (p5_mix32((idx0 + e) ^ hseed) >> 8) >= thr ? x * dscale : 0.f;      // (ABL 32, lab: no dropout hash)
                  if constexpr (EK == 3 || EK == 4) x += av[e];
                  if constexpr (EK == 5) x = av[e] > 0.f ? x : 0.f;
                  v[e] = x;
                }
              }
              packed[h] = pack16<T>(v);
              if (ssq) {
                float w[8];
                unpack16<T>(packed[h], w);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  if constexpr (EK == 5) sq += w[e] * av[e];      // (MASK_POS + ssq_out: <d pre, pre> * alpha, see P5_EPI_MASK_POS)
                  else sq += w[e] * w[e];
                }
              }
            }
            if constexpr (EK >= 3) {
              if (i + AD < TM) aux_load(i + AD);
            }
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
              T* const cpi = cp + (size_t)i * 16 * ldc + h * 32;
              if constexpr ((ABL & 16) != 0) { if (packed[h][0] == 0x12345678u) st16(cpi, packed[h]); }      // (lab: epilogue math without the stores)
#if !defined(P5_GEMM5_NO_NT) && !defined(P5_EMU)
              // streaming stores: a 33 MB output passes through once and is next read by another kernel from the Infinity Cache / HBM, not from
              // this XCD's 4 MB L2 -- in the C2 step 4.237 vs 4.251 ms with plain stores (profiles/r06_call2_option_ab.txt)
              else __builtin_nontemporal_store(packed[h], (u32x4*)cpi);
#else
              else st16(cpi, packed[h]);
#endif
            }
            if (ssq) {
              if constexpr (EK == 5) sq *= inv_alpha;
              sq += __shfl_xor(sq, 16);
              sq += __shfl_xor(sq, 32);
              const int cg = (u.n0 + wn * WTN) >> 6;
              if (gl == 0 && cg < ssq_nt) ssq[(size_t)(row + i * 16) * ssq_nt + cg] = sq;
            }
          }
        };
        // ---- gated-GELU (T5 v1.1 FFN, HF modeling_t5.py:97-123: h = gelu_new(wi_0 x) * (wi_1 x)), forward: the two 8-column groups of a lane
        // are u0 and u1 of the same eight hidden units (gate-interleaved B rows); u is stored for the backward in its [u0 | u1] layout
        // and the gate is taken from the ROUNDED values, i.e. exactly what the stand-alone p5_gated_gelu_fwd_kernel reads back.
        auto gate_fwd = [&](auto dk) {
          constexpr bool DROP = decltype(dk)::value != 0;
          const int F = cx.gate_F, ldc2 = cx.ldc2;
          const int hc = ((u.n0 + wn * WTN) >> 1) + gl * 8;                // first of this lane's eight hidden units
          T* const hp = (T*)cx.C + (size_t)row * ldc + hc;
          T* const up = (T*)cx.C2 + (size_t)row * ldc2 + hc;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float sc = rowss ? __shfl(sc2[i % RPG], (le & 15) + 16 * (i / RPG)) : alpha;
            float a[8], b[8], hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = acc[i][0][r] * sc; a[4 + r] = acc[i][1][r] * sc; b[r] = acc[i][2][r] * sc; b[4 + r] = acc[i][3][r] * sc; }
            const u32x4 pa = pack16<T>(a), pb = pack16<T>(b);
            unpack16<T>(pa, a);
            unpack16<T>(pb, b);
            const uint32_t idx0 = (uint32_t)((row + i * 16) * F + hc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t = tanhf(0.7978845608028654f * (a[e] + 0.044715f * a[e] * a[e] * a[e]));
              float v = 0.5f * a[e] * (1.f + t) * b[e];
              if constexpr (DROP) v = (p5_mix32((idx0 + e) ^ hseed) >> 8) >= thr ? v * dscale : 0.f;
              hv[e] = v;
            }
            st16(up + (size_t)i * 16 * ldc2, pa);
            st16(up + (size_t)i * 16 * ldc2 + F, pb);
            st16(hp + (size_t)i * 16 * ldc, pack16<T>(hv));
          }
        };
        // ---- backward: acc = dh (grad of the dropped product), aux = the stored u; writes du = [dh u1 gelu'(u0) | dh gelu(u0)]
        auto gate_bwd = [&](auto dk) {
          constexpr bool DROP = decltype(dk)::value != 0;
          const int F = N;
          const T* const uq = (const T*)cx.aux + (size_t)row * ldaux + col0;
          T* const dq = (T*)cx.C + (size_t)row * ldc + col0;
          constexpr int AD = 2;
          u32x4 ua[AD][TN / 2], ub[AD][TN / 2];
          auto u_load = [&](int i) {
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
              ua[i % AD][h] = ld16(uq + (size_t)i * 16 * ldaux + h * 32);
              ub[i % AD][h] = ld16(uq + (size_t)i * 16 * ldaux + h * 32 + F);
            }
          };
          u_load(0);
          u_load(1);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            u32x4 o0[TN / 2], o1[TN / 2];
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
              float g8[8], a[8], b[8], d0[8], d1[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) { g8[r] = acc[i][2 * h][r] * alpha; g8[4 + r] = acc[i][2 * h + 1][r] * alpha; }
              {   // dh as the stand-alone path sees it: rounded to the compute dtype by the data-gradient GEMM's store
                const u32x4 pg = pack16<T>(g8);
                unpack16<T>(pg, g8);
              }
              unpack16<T>(ua[i % AD][h], a);
              unpack16<T>(ub[i % AD][h], b);
              const uint32_t idx0 = (uint32_t)((row + i * 16) * F + col0 + h * 32);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float g = g8[e];
                if constexpr (DROP) g = (p5_mix32((idx0 + e) ^ hseed) >> 8) >= thr ? g * dscale : 0.f;
                const float k = 0.7978845608028654f;
                const float t = tanhf(k * (a[e] + 0.044715f * a[e] * a[e] * a[e]));
                const float gel = 0.5f * a[e] * (1.f + t);
                const float dgel = 0.5f * (1.f + t) + 0.5f * a[e] * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * a[e] * a[e]);
                d0[e] = g * b[e] * dgel;
                d1[e] = g * gel;
              }
              o0[h] = pack16<T>(d0);
              o1[h] = pack16<T>(d1);
            }
            if (i + AD < TM) u_load(i + AD);
#pragma unroll
            for (int h = 0; h < TN / 2; ++h) {
              st16(dq + (size_t)i * 16 * ldc + h * 32, o0[h]);
              st16(dq + (size_t)i * 16 * ldc + h * 32 + F, o1[h]);
            }
          }
        };
        if constexpr (GATE) {
          if (epi == P5_EPI_GELU_GATE) { if (cx.do_drop) gate_fwd(P5EpiTag<1>{}); else gate_fwd(P5EpiTag<0>{}); return; }
          if (epi == P5_EPI_GELU_GATE_BWD) { if (cx.do_drop) gate_bwd(P5EpiTag<1>{}); else gate_bwd(P5EpiTag<0>{}); return; }
        }
        if (epi == P5_EPI_RELU_DROP) { if (cx.do_drop) all_rows(P5EpiTag<2>{}); else all_rows(P5EpiTag<1>{}); }
        else if (epi == P5_EPI_RESID_DROP) { if (cx.do_drop) all_rows(P5EpiTag<4>{}); else all_rows(P5EpiTag<3>{}); }
        else if (epi == P5_EPI_MASK_POS) all_rows(P5EpiTag<5>{});
        else all_rows(P5EpiTag<0>{});
      } else {
        const int epi = cx.epi, ldc = cx.ldc;
        const float alpha = cx.alpha;
        float* cp = (float*)cx.C + (size_t)(u.m0 + wm * WTM + (le & 15)) * ldc + (u.n0 + wn * WTN + (le >> 4) * 4);
        auto rows = [&](const f32x4(&a)[TN]) {
          f32x4 c[TN];
          if (epi == P5_EPI_ACCUM) {
#pragma unroll
            for (int j = 0; j < TN; ++j) c[j] = *(const f32x4*)(cp + j * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
              *(f32x4*)(cp + j * 16) = (f32x4){c[j][0] + a[j][0] * alpha, c[j][1] + a[j][1] * alpha, c[j][2] + a[j][2] * alpha, c[j][3] + a[j][3] * alpha};
          } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) *(f32x4*)(cp + j * 16) = (f32x4){a[j][0] * alpha, a[j][1] * alpha, a[j][2] * alpha, a[j][3] * alpha};
          }
          cp += (size_t)16 * ldc;
        };
#define P5_G5_R(i) rows(acc[i])
        P5_G5_ROWBLOCKS(P5_G5_R);
#undef P5_G5_R
      }
      return;
    }
    const P5GemmArgs& g = grp.p[u.pi];
    const uint32_t seed = p5_seed(g.drop);
    const bool do_drop = g.drop.state != nullptr && g.drop.thr != 0;
    const bool vec_ok = (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.aux == nullptr || ((g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0));
#define P5_G5_ER(i) epi_rows(acc[i], i, u, g, seed, do_drop, vec_ok, le)
    P5_G5_ROWBLOCKS(P5_G5_ER);
#undef P5_G5_ER
  };

  // Both operands' fragments double-buffered (96 registers): the 12 reads of the next K-chunk go out under the first 24 of a
  // chunk's 32 MFMAs, so the `lgkmcnt(0)` in front of the mid-step barrier (slot `buf` must be read out before a loader may
  // overwrite it) finds nothing outstanding.
  u32x4 fa0[TM], fa1[TM], fb0[TN], fb1[TN];
  auto half = [&](u32x4(&fa)[TM], u32x4(&fb)[TN], u32x4(&na)[TM], u32x4(&nb)[TN], int nbuf, int nc) {
    P5_SCHED_FENCE();
#pragma unroll
    for (int t = 0; t < NMM; ++t) {
      const int i = t / TN, j = t % TN;
      mm(acc[i][j], fa[i], fb[j]);
      P5_SCHED_FENCE();
      if ((t & 1) == 0 && t / 2 < TM + TN) {
        const int r = t / 2;          // read order: B0 A0 B1 B2 B3 A1 .. A7 (what the next chunk's first MFMAs need first)
        if (r == 0) nb[0] = frag(nbuf, true, 0, nc);
        else if (r == 1) na[0] = frag(nbuf, false, 0, nc);
        else if (r < 1 + TN) nb[r - 1] = frag(nbuf, true, r - 1, nc);
        else na[r - TN] = frag(nbuf, false, r - TN, nc);
        P5_SCHED_FENCE();
      }
    }
  };
  auto step = [&](int buf, int nb1) {
    half(fa0, fb0, fa1, fb1, buf, 1);
    P5_BARRIER_LDS();                     // K-step s+1 has landed (the loaders waited for their copies); slot `buf` is fully read
    half(fa1, fb1, fa0, fb0, nb1, 0);
  };

  P5_BARRIER_LDS();                       // K-step 0 has landed
#pragma unroll
  for (int j = 0; j < TN; ++j) fb0[j] = frag(0, true, j, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) fa0[i] = frag(0, false, i, 0);
  zero_acc();
  int buf = 0;
  for (int it = 0; it < nmy; ++it) {
    const Unit u = decode(it);
    if constexpr (VAR == 3 && !KS) nb_stats(u);
    int k = 0;
    do {
      const int nb1 = buf == NST - 1 ? 0 : buf + 1;
      step(buf, nb1);
      buf = nb1;
    } while (++k < u.nk);
    // (forcing the descriptor scalars into registers before the K loop -- an asm pin after the first K-step -- was measured:
    //  the 24 live scalars spill through a vector register's lanes and the main loop slows from 14.8 to 20 us on 8192x2048x512)
    epilogue(u, load_ctx(u));
    zero_acc();
    if constexpr (VAR == 3 && !KS) {
      // the norm-backward epilogue needs the registers the next unit's first fragments were prefetched into: read them again (their K-step
      // stays in slot `buf` until this wave has passed the next barrier)
      if (it + 1 < nmy) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fb0[j] = frag(buf, true, j, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa0[i] = frag(buf, false, i, 0);
      }
    }
  }
}
#undef P5_G5_ROWBLOCKS
