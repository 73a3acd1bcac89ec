// p5_gemm_tu.hip -- translation unit of the GEMM family: tile selection and launch of p5_gemm.h / p5_gemm4.h / p5_gemm5.h kernels
// (moved out of p5_lib.hip so that the units of libp5hip.so compile in parallel; declarations in p5_host.h).
#include <cstring>
#include <cstdio>
#include "p5_host.h"
#include "p5_gemm5.h"

// tuning knobs (p5_set_option / environment): gemm_v2 = LDS stages (2 or 3) of the hand-pipelined main loop, 0 = v1 loop
int g_opt_gemm_v2 = getenv("P5_GEMM_V2") ? atoi(getenv("P5_GEMM_V2")) : 0;
int g_opt_gemm_tile = getenv("P5_GEMM_TILE") ? atoi(getenv("P5_GEMM_TILE")) : 0;
int g_opt_gemm_ring = getenv("P5_GEMM_RING") ? atoi(getenv("P5_GEMM_RING")) : 1;      // ring kernel for weight gradients
int g_opt_gemm_xcd_rect = getenv("P5_GEMM_XCD_RECT") ? atoi(getenv("P5_GEMM_XCD_RECT")) : 1;   // rectangular per-XCD tile blocks
int g_opt_gemm_small_ring = getenv("P5_GEMM_SMALL_RING") ? atoi(getenv("P5_GEMM_SMALL_RING")) : 1;   // 8-slot ring for sub-CU-count problems
int g_opt_gemm_ring32 = getenv("P5_GEMM_RING32") ? atoi(getenv("P5_GEMM_RING32")) : 128;   // 32x64 ring tiles for problems of at most this many 64x64 tiles (0 = off)
int g_opt_gemm_small_ring_tiles = getenv("P5_GEMM_SMALL_RING_TILES") ? atoi(getenv("P5_GEMM_SMALL_RING_TILES")) : 256;   // ... up to this many 64x64 tiles
int g_opt_gemm_ring_stages = getenv("P5_GEMM_RING_STAGES") ? atoi(getenv("P5_GEMM_RING_STAGES")) : 4;
int g_opt_gemm_ring_wgs = getenv("P5_GEMM_RING_WGS") ? atoi(getenv("P5_GEMM_RING_WGS")) : 160;      // target tiles x splits (in-step sweep: 96..192 equal, 256 +1.5 %)
int g_opt_gemm_ksdma = getenv("P5_GEMM_KSDMA") ? atoi(getenv("P5_GEMM_KSDMA")) : 1;   // direct-to-LDS copies of K-strided operands
int g_opt_gemm_wide = getenv("P5_GEMM_WIDE") ? atoi(getenv("P5_GEMM_WIDE")) : 1;         // 256x128 persistent ring for wide outputs
int g_opt_gemm_wide_min_tiles = getenv("P5_GEMM_WIDE_MIN_TILES") ? atoi(getenv("P5_GEMM_WIDE_MIN_TILES")) : 160;
int g_opt_gemm_ring128_min_k = getenv("P5_GEMM_RING128_MIN_K") ? atoi(getenv("P5_GEMM_RING128_MIN_K")) : 1024;       // ... from this reduction length on
int g_opt_gemm_ring128_min_tiles = getenv("P5_GEMM_RING128_MIN_TILES") ? atoi(getenv("P5_GEMM_RING128_MIN_TILES")) : 128;   // ... and this many 128x128 tiles
int g_opt_gemm_ring_n512 = getenv("P5_GEMM_RING_N512") ? atoi(getenv("P5_GEMM_RING_N512")) : 1;   // ring kernel for N = d_model, K >= 1024
int g_opt_g4_nst = getenv("P5_G4_NST") ? atoi(getenv("P5_G4_NST")) : 3;          // ring depth of the 128x128 configuration
int g_opt_g4_wgs = getenv("P5_G4_WGS") ? atoi(getenv("P5_G4_WGS")) : 256;        // workgroups per launch (one per CU)
int g_opt_split_pipe = getenv("P5_SPLIT_PIPE") ? atoi(getenv("P5_SPLIT_PIPE")) : 1;    // split-f16 fp32 GEMMs: the three-deep pipelined kernel (0 = p5_gemm_kernel<MM = 1>)
int g_opt_split_big_tiles = getenv("P5_SPLIT_BIG_TILES") ? atoi(getenv("P5_SPLIT_BIG_TILES")) : 0;    // split-f16 fp32 GEMMs: 128x128 tiles from this many of them (0 = the fp32 rule: from 512; measured on the verification pass, 64x64 tiles win below that: 5.05 vs 5.30 ms per batch)
int g_opt_gemm_ws128 = getenv("P5_GEMM_WS128") ? atoi(getenv("P5_GEMM_WS128")) : 1;    // N = d_model outputs (128..256 tiles of 128x128) on the wave-specialised 128x128 instance, from K = gemm_ws128_min_k
int g_opt_gemm_ws128_min_k = getenv("P5_GEMM_WS128_MIN_K") ? atoi(getenv("P5_GEMM_WS128_MIN_K")) : 512;
int g_opt_gemm_rect = getenv("P5_GEMM_RECT") ? atoi(getenv("P5_GEMM_RECT")) : 1;     // wave-specialised kernel: (32 / cb) x cb tile blocks per XCD round instead of n-fastest runs
int g_opt_gemm_ws = getenv("P5_GEMM_WS") ? atoi(getenv("P5_GEMM_WS")) : 3;          // 256x128 tiles on the wave-specialised kernel (p5_gemm5.h): bit 0 K-contiguous (forward / dgrad), bit 1 K-strided (wgrad groups)

template <class T, int BM, int BN>
static int launch_gemm_tile(P5GemmArgs g, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk), block(256);
  P5_PROF_FLOPS(2.0 * g.M * g.N * g.K);
  P5_PROF_SHAPE(g.M, g.N, g.K);
  P5_PROF_TAG(sizeof(T) == 2 ? (BM == 256 ? "bf16 256x256" : BM == 128 ? (g.a_ks ? "bf16 128x128 KS" : "bf16 128x128 KC") : (g.a_ks ? "bf16 64x64 KS" : (g.b_ks ? "bf16 64x64 KC/KS" : "bf16 64x64 KC")))
                             : (BM == 128 ? "f32 128x128" : "f32 64x64"));
  g.xcd_bm = g.xcd_bn = 0;
  if (g_opt_gemm_xcd_rect) {
    // exact cover of the gx x gy tile grid by 8 equal rectangles; keep the one with the smallest half-perimeter, and only if
    // it beats the contiguous-run order (runs of q tiles: ~ceil(q / gx) rows x min(q, gx) columns)
    const int gx = (int)grid.x, gy = (int)grid.y;
    if ((gx * gy) % 8 == 0) {
      const int q = gx * gy / 8;
      int best = (q + gx - 1) / gx + (q < gx ? q : gx);
      for (int bn = 1; bn <= gx; ++bn) {
        if (gx % bn || q % bn) continue;
        const int bm = q / bn;
        if (bm > gy || gy % bm || (gx / bn) * (gy / bm) != 8) continue;
        if (bm + bn < best) { best = bm + bn; g.xcd_bm = bm; g.xcd_bn = bn; }
      }
    }
  }
  const int mode = g.a_ks * 2 + g.b_ks;
  // direct-to-LDS staging for K-contiguous operands whenever every K-step is full (fast-mode dtype only)
  const bool dma = sizeof(T) == 2 && (g.K % (TT<T>::KCH * 2)) == 0;
  const int v2 = g_opt_gemm_v2;
  if constexpr (sizeof(T) == 2 && BM == 256) {
    P5_REQUIRE(mode == 0 && dma, "gemm: 256x256 tiles need bf16 K-contiguous operands with K % 64 == 0");
    P5_LAUNCH((p5_gemm3_kernel<BM, BN, 2, 4>), grid, dim3(512), 0, s, g);
    return P5_KCHECK();
  }
  if constexpr (sizeof(T) == 2 && BM == 128) {
    if (mode == 0 && dma && g.ring && v2 == 0) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 4>), grid, block, 0, s, g); return P5_KCHECK(); }
    if (mode == 0 && dma && v2 == 3) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 3>), grid, block, 0, s, g); return P5_KCHECK(); }
    if (mode == 0 && dma && v2 == 2) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 2>), grid, block, 0, s, g); return P5_KCHECK(); }
    if (mode == 3 && dma && (v2 >= 3 || g.ring)) {
      if (v2 == 3 || (v2 == 0 && g_opt_gemm_ring_stages == 3)) P5_LAUNCH((p5_gemm2_kernel<BM, BN, 3, true, true>), grid, block, 0, s, g);
      else P5_LAUNCH((p5_gemm2_kernel<BM, BN, 4, true, true>), grid, block, 0, s, g);
      return P5_KCHECK();
    }
  }
  if constexpr (sizeof(T) == 2 && BM == 64) {
    // small problems (fewer tiles than CUs): the K loop of a lone workgroup is a chain of load latencies, ~1 us per step with
    // one stage of lookahead.  Eight 16 KiB ring slots keep seven K-steps in flight (K = 512 is fetched entirely up front):
    // 512x512x512 8.2 -> ~4 us.  This is what the decoder and the decode step are made of.
    if (g.ring && dma) {
      if (mode == 0) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 8, false, false>), grid, block, 0, s, g); return P5_KCHECK(); }
      if (mode == 1) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 8, false, true>), grid, block, 0, s, g); return P5_KCHECK(); }
      if (mode == 3) { P5_LAUNCH((p5_gemm2_kernel<BM, BN, 8, true, true>), grid, block, 0, s, g); return P5_KCHECK(); }
    }
  }
  if constexpr (sizeof(T) == 4 && BM <= 128) {
    if (mode == 0 && g.mm_split) {       // fp32 operands, products on the f16 matrix cores (p5_gemm.h, two-term split)
      P5_PROF_TAG(BM == 128 ? "f32 128x128 split-f16" : "f32 64x64 split-f16");
      if (g_opt_split_pipe && g.splitk <= 1 && (g.K % 32) == 0) P5_LAUNCH((p5_gemm_split_kernel<BM, BN>), grid, block, 0, s, g);      // three K-steps of lookahead
      else P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, false, 2, false, false, 1>), grid, block, 0, s, g);
      return P5_KCHECK();
    }
  }
  if (mode == 0 && dma) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, false, 2, sizeof(T) == 2, sizeof(T) == 2>), grid, block, 0, s, g);
  else if (mode == 0) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, false, 2, false, false>), grid, block, 0, s, g);
  else if (mode == 1 && dma && g_opt_gemm_ksdma) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, true, 2, sizeof(T) == 2, sizeof(T) == 2>), grid, block, 0, s, g);
  else if (mode == 3 && dma && g_opt_gemm_ksdma) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, true, true, 2, sizeof(T) == 2, sizeof(T) == 2>), grid, block, 0, s, g);
  else if (mode == 1 && dma) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, true, 2, sizeof(T) == 2, false>), grid, block, 0, s, g);
  else if (mode == 1) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, false, true, 2, false, false>), grid, block, 0, s, g);
  else if (mode == 3) P5_LAUNCH((p5_gemm_kernel<T, BM, BN, true, true, 2, false, false>), grid, block, 0, s, g);
  else return fail("gemm: (A strided, B contiguous) is not instantiated");
  return P5_KCHECK();
}

// ---- persistent ring GEMM (p5_gemm4.h): one launch over a group of problems, bf16 operands, K % (64 * splitk) == 0 ----
template <int BM, int BN, int WMW, int WNW, int NST, bool KS, int OCC = 1>
static int launch_gemm4_cfg(P5GemmGroup& grp, hipStream_t s) {
  int units = 0;
  for (int i = 0; i < grp.nprob; ++i) {
    P5GemmArgs& g = grp.p[i];
    if (g.splitk < 1) g.splitk = 1;
    P5_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 64 * g.splitk && g.K % (64 * g.splitk) == 0, "gemm4: K must be a multiple of 64 x split-K");
    P5_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm4: operand alignment");
    P5_REQUIRE(g.splitk == 1 || g.epi == P5_EPI_ATOMIC, "gemm4: split-K needs the atomic epilogue");
    if (g.epi == P5_EPI_ATOMIC || g.epi == P5_EPI_ACCUM) P5_REQUIRE(g.c_f32, "gemm4: accumulate epilogues need fp32 C");
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    g.g4_tiles_n = tn;
    g.g4_nk = g.K / 64 / g.splitk;
    grp.unit_begin[i] = units;
    units += tm * tn * g.splitk;
  }
  grp.unit_begin[grp.nprob] = units;
  grp.total_units = units;
  int nwg = ((units + 7) / 8) * 8;
  if (nwg > g_opt_g4_wgs * OCC) nwg = g_opt_g4_wgs * OCC;
  {
    double fl = 0.0;
    for (int i = 0; i < grp.nprob; ++i) fl += 2.0 * grp.p[i].M * grp.p[i].N * grp.p[i].K;
    P5_PROF_FLOPS(fl);
    P5_PROF_TAG(KS ? (BM == 256 ? "256x128 KS" : "128x128 KS") : (BM == 256 ? "256x128 KC" : "128xN KC"));
    if (grp.nprob == 1) P5_PROF_SHAPE(grp.p[0].M, grp.p[0].N, grp.p[0].K);
  }
  P5_LAUNCH((p5_gemm4_kernel<BM, BN, WMW, WNW, NST, KS, 0, OCC>), dim3(nwg), dim3(WMW * WNW * 64), 0, s, grp);
  return P5_KCHECK();
}
template <bool KS, int BM = 256>
static int launch_gemm5(P5GemmGroup& grp, hipStream_t s) {     // same unit bookkeeping as launch_gemm4_cfg<BM, 128, ...>, 4 loader + 4 compute waves
  int units = 0;
  for (int i = 0; i < grp.nprob; ++i) {
    P5GemmArgs& g = grp.p[i];
    if (g.splitk < 1) g.splitk = 1;
    P5_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 64 * g.splitk && g.K % (64 * g.splitk) == 0, "gemm5: K must be a multiple of 64 x split-K");
    P5_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm5: operand alignment");
    P5_REQUIRE(g.splitk == 1 || g.epi == P5_EPI_ATOMIC, "gemm5: split-K needs the atomic epilogue");
    if (g.epi == P5_EPI_ATOMIC || g.epi == P5_EPI_ACCUM) P5_REQUIRE(g.c_f32, "gemm5: accumulate epilogues need fp32 C");
    g.g4_tiles_n = (g.N + 127) / 128;
    g.g4_nk = g.K / 64 / g.splitk;
    grp.unit_begin[i] = units;
    units += ((g.M + BM - 1) / BM) * g.g4_tiles_n * g.splitk;
  }
  grp.unit_begin[grp.nprob] = units;
  grp.total_units = units;
  int nwg = ((units + 7) / 8) * 8;
  if (nwg > g_opt_g4_wgs) nwg = g_opt_g4_wgs;
  for (int i = 0; i < grp.nprob; ++i) grp.p[i].g4_cb = 0;
  // (option 1: only where the B operand does not fit an XCD's 4 MiB L2 beside the A panels -- measured in one call: T5-large 115.2 -> 112.8 ms
  //  per step, T5-base 12.43 -> 12.42, T5-small 4.145 -> 4.174 with it everywhere (profiles/r06_call15_rect_blocks.txt); 2: wherever whole blocks fit)
  const bool rect_pays = g_opt_gemm_rect >= 2 || (long long)grp.p[0].N * grp.p[0].K * 2 >= (6ll << 20);
  if (!KS && g_opt_gemm_rect && rect_pays && grp.nprob == 1 && grp.p[0].splitk == 1 && nwg == 256 && units % 8 == 0) {
    // rectangular per-XCD tile blocks (p5_gemm5.h decode): whole blocks must tile the XCD's range of tile rows
    const int tn = grp.p[0].g4_tiles_n, upx = units / 8;
    if (upx % tn == 0) {
      const int rows = upx / tn;
      for (int cb = 8; cb >= 4 && grp.p[0].g4_cb == 0; cb >>= 1)
        if (tn % cb == 0 && tn > cb && rows % (32 / cb) == 0) grp.p[0].g4_cb = cb;
    }
  }
  {
    double fl = 0.0;
    for (int i = 0; i < grp.nprob; ++i) fl += 2.0 * grp.p[i].M * grp.p[i].N * grp.p[i].K;
    P5_PROF_FLOPS(fl);
    P5_PROF_TAG(KS ? "KS: grouped weight gradients" : (BM == 128 ? "KC 128x128: N = d_model outputs" : "KC: forward / data-gradient GEMMs"));
    if (grp.nprob == 1) P5_PROF_SHAPE(grp.p[0].M, grp.p[0].N, grp.p[0].K);
  }
  if constexpr (BM == 128) {
    bool special = false;
    for (int i = 0; i < grp.nprob; ++i)
      special = special || grp.p[i].epi == P5_EPI_GELU_GATE || grp.p[i].epi == P5_EPI_GELU_GATE_BWD || grp.p[i].epi == P5_EPI_CE_STATS || grp.p[i].epi == P5_EPI_CE_GRAD;
    P5_REQUIRE(!KS && !special, "gemm5: the 128-row tile carries the plain K-contiguous epilogues only");
    bool nbw = false;
    for (int i = 0; i < grp.nprob; ++i) nbw = nbw || grp.p[i].epi == P5_EPI_NORM_BWD;
    for (int i = 0; i < grp.nprob && nbw; ++i) {
      const P5GemmArgs& g = grp.p[i];
      P5_REQUIRE(g.epi == P5_EPI_NORM_BWD, "gemm5: a T5LayerNorm-backward launch carries problems of that kind only");
      P5_REQUIRE((g.M % 128) == 0 && (g.N % 128) == 0 && !g.c_f32 && g.splitk == 1, "gemm5: the T5LayerNorm-backward epilogue writes whole 128x128 tiles");
      P5_REQUIRE(g.ldc == g.N && g.ldaux == g.N && (g.C2 == nullptr || g.ldc2 == g.N) && (long long)g.M * g.N < (1ll << 30), "gemm5: T5LayerNorm-backward operands are [M, N] with leading dimension N");
      P5_REQUIRE(g.C && g.aux && g.rowss && g.rowss_nt > 0 && g.nb_dot && g.nb_dot_nt > 0 && g.nb_rin && g.nb_rout && g.nb_w && g.nb_dw, "gemm5: T5LayerNorm-backward epilogue arguments");
      P5_REQUIRE((g.ldc % 8) == 0 && (g.ldaux % 8) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)g.aux % 16) == 0 && ((uintptr_t)g.nb_rin % 16) == 0 &&
                 ((uintptr_t)g.nb_rout % 16) == 0 && ((uintptr_t)g.nb_w % 16) == 0 && ((uintptr_t)g.nb_dw % 16) == 0 &&
                 (g.C2 == nullptr || ((g.ldc2 % 8) == 0 && ((uintptr_t)g.C2 % 16) == 0)), "gemm5: T5LayerNorm-backward epilogue alignment");
    }
    if (nbw) {
      P5_PROF_TAG("KC 128x128 + T5LayerNorm-backward epilogue");
      P5_LAUNCH((p5_gemm5_kernel<false, 0, 3, 128>), dim3(nwg), dim3(512), 0, s, grp);
    } else {
      P5_LAUNCH((p5_gemm5_kernel<false, 0, 0, 128>), dim3(nwg), dim3(512), 0, s, grp);
    }
    return P5_KCHECK();
  }
#ifdef P5_GEMM5_ABL      // lab builds only (tools/lab/build_ablations.sh): the forward / data-gradient instance with parts of it removed, timed INSIDE the step
  P5_LAUNCH((p5_gemm5_kernel<KS, KS ? 0 : P5_GEMM5_ABL>), dim3(nwg), dim3(512), 0, s, grp);
#else
  // (spelled out so that the in-run profiler's stringified kernel name says which instance ran, not "<KS>")
  bool gate = false, ce = false;
  for (int i = 0; i < grp.nprob; ++i) {
    gate = gate || grp.p[i].epi == P5_EPI_GELU_GATE || grp.p[i].epi == P5_EPI_GELU_GATE_BWD;
    ce = ce || grp.p[i].epi == P5_EPI_CE_STATS || grp.p[i].epi == P5_EPI_CE_GRAD;
  }
  if constexpr (KS) {
    P5_REQUIRE(!gate && !ce, "gemm5: the gated-GELU / cross-entropy epilogues belong to the K-contiguous instance");
    P5_LAUNCH((p5_gemm5_kernel<true>), dim3(nwg), dim3(512), 0, s, grp);
  } else if (ce) {
    P5_REQUIRE(grp.nprob == 1 && !gate, "gemm5: a cross-entropy launch carries one problem");
    P5_PROF_TAG("KC + logit-free cross-entropy epilogue");
    P5_LAUNCH((p5_gemm5_kernel<false, 0, 2>), dim3(nwg), dim3(512), 0, s, grp);
  } else if (gate) {
    P5_PROF_TAG("KC + gated-GELU epilogue");
    P5_LAUNCH((p5_gemm5_kernel<false, 0, 1>), dim3(nwg), dim3(512), 0, s, grp);
  } else {
    P5_LAUNCH((p5_gemm5_kernel<false>), dim3(nwg), dim3(512), 0, s, grp);
  }
#endif
  return P5_KCHECK();
}
int launch_gemm4(int cfg, bool ks, P5GemmGroup& grp, hipStream_t s) {
  P5_REQUIRE(grp.nprob >= 1 && grp.nprob <= P5_MAX_GROUP, "gemm4: 1..8 problems per launch");
  if (cfg == P5_G5_128x128) {
    P5_REQUIRE(!ks, "gemm5: the 128-row tile is K-contiguous only");
    return launch_gemm5<false, 128>(grp, s);
  }
  if (cfg == P5_G5_256x128 || (cfg == P5_G4_256x128 && (g_opt_gemm_ws & (ks ? 2 : 1)))) return ks ? launch_gemm5<true>(grp, s) : launch_gemm5<false>(grp, s);
  if (ks) {
    if (cfg == P5_G4_256x128) return launch_gemm4_cfg<256, 128, 4, 2, 3, true>(grp, s);
    P5_REQUIRE(cfg == P5_G4_128x128, "gemm4: K-strided operands run on 128x128 or 256x128 tiles");
    if (g_opt_g4_nst == 2) return launch_gemm4_cfg<128, 128, 2, 2, 2, true, 2>(grp, s);     // two-slot ring, two workgroups per CU
    if (g_opt_g4_nst == 3) return launch_gemm4_cfg<128, 128, 2, 2, 3, true>(grp, s);
    if (g_opt_g4_nst == 4) return launch_gemm4_cfg<128, 128, 2, 2, 4, true>(grp, s);
    return launch_gemm4_cfg<128, 128, 2, 2, 5, true>(grp, s);
  }
  if (cfg == P5_G4_256x128) return launch_gemm4_cfg<256, 128, 4, 2, 3, false>(grp, s);
  if (cfg == P5_G4_128x256) return launch_gemm4_cfg<128, 256, 2, 4, 3, false>(grp, s);
  if (g_opt_g4_nst == 2) return launch_gemm4_cfg<128, 128, 2, 2, 2, false, 2>(grp, s);
  if (g_opt_g4_nst == 3) return launch_gemm4_cfg<128, 128, 2, 2, 3, false>(grp, s);
  if (g_opt_g4_nst == 4) return launch_gemm4_cfg<128, 128, 2, 2, 4, false>(grp, s);
  return launch_gemm4_cfg<128, 128, 2, 2, 5, false>(grp, s);
}

// the fused gated-GELU epilogues exist in the whole-tile path of p5_gemm5.h only: bf16, both operands K-contiguous, every 256x128 tile
// inside the output, enough tiles for the wide route.  M rows, N GEMM columns (2F forward, F backward), K reduction length.
bool p5l_gemm_gate_ok(int M, int N, int K, int lda, int ldb) {
  return g_opt_gemm_wide && (g_opt_gemm_ws & 1) && !g_opt_gemm_tile && !g_opt_gemm_v2 && (M % 256) == 0 && (N % 128) == 0 && (K % 64) == 0 && (lda % 64) == 0 &&
         (ldb % 64) == 0 && (long)(M / 256) * (N / 128) >= g_opt_gemm_wide_min_tiles;
}

// the T5LayerNorm-backward epilogue (P5_EPI_NORM_BWD): whole 128x128 tiles of the wave-specialised kernel, any number of them
bool p5l_gemm_normbwd_ok(int M, int N, int K, int lda, int ldb) {
  return (g_opt_gemm_ws & 1) && !g_opt_gemm_tile && !g_opt_gemm_v2 && (M % 128) == 0 && (N % 128) == 0 && (K % 64) == 0 && (lda % 64) == 0 && (ldb % 64) == 0;
}

bool p5l_gemm_ce_ok(int M, int N, int K, int lda, int ldb) {
  return g_opt_gemm_wide && (g_opt_gemm_ws & 1) && !g_opt_gemm_tile && !g_opt_gemm_v2 && (K % 64) == 0 && (lda % 64) == 0 && (ldb % 64) == 0 &&
         (long)((M + 255) / 256) * ((N + 127) / 128) >= g_opt_gemm_wide_min_tiles;
}

template <class T>
static int launch_gemm_impl(P5GemmArgs g, hipStream_t s) {
  constexpr int EPF = TT<T>::EPF;
  if (g.epi == P5_EPI_CE_STATS || g.epi == P5_EPI_CE_GRAD) {
    // logit-free cross-entropy: the epilogue guards rows and columns itself, so any M, N go -- but only the wave-specialised kernel has it
    P5_REQUIRE(sizeof(T) == 2 && !g.a_ks && !g.b_ks && g.splitk <= 1 && p5l_gemm_ce_ok(g.M, g.N, g.K, g.lda, g.ldb), "gemm: the cross-entropy epilogues run on the wide bf16 kernel (caller: check p5l_gemm_ce_ok)");
    P5_REQUIRE(g.ce_labels && (g.epi == P5_EPI_CE_STATS ? (g.ce_part && g.ce_lab && g.ce_np == (g.N + 63) / 64) : (g.ce_lse && g.ce_g && g.C && !g.c_f32 && (g.ldc % 8) == 0 && ((uintptr_t)g.C % 16) == 0)),
               "gemm: cross-entropy epilogue arguments");
    g.C2 = nullptr; g.ldc2 = 0; g.gate_F = 0;
    P5GemmGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.nprob = 1;
    grp.p[0] = g;
    grp.p[0].splitk = 1;
    return launch_gemm4(P5_G4_256x128, false, grp, s);
  }
  g.ce_labels = nullptr; g.ce_part = nullptr; g.ce_lab = nullptr; g.ce_lse = nullptr; g.ce_g = nullptr; g.ce_np = 0;
  if (g.epi == P5_EPI_NORM_BWD) {
    P5_REQUIRE(sizeof(T) == 2 && !g.a_ks && !g.b_ks && g.splitk <= 1 && p5l_gemm_normbwd_ok(g.M, g.N, g.K, g.lda, g.ldb),
               "gemm: the T5LayerNorm-backward epilogue runs on 128-row tiles of the wave-specialised bf16 kernel (caller: check p5l_gemm_normbwd_ok)");
    P5GemmGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.nprob = 1;
    grp.p[0] = g;
    grp.p[0].splitk = 1;
    grp.p[0].gate_F = 0;
    return launch_gemm4(P5_G5_128x128, false, grp, s);
  }
  g.nb_dot = nullptr; g.nb_rin = nullptr; g.nb_rout = nullptr; g.nb_w = nullptr; g.nb_dw = nullptr; g.nb_dot_nt = 0;
  if (g.epi == P5_EPI_MASK_POS && g.ssq_out)
    P5_REQUIRE(sizeof(T) == 2 && !g.a_ks && !g.b_ks && g.splitk <= 1 && !g.c_f32 && g.ssq_nt > 0 && p5l_gemm_gate_ok(g.M, g.N, g.K, g.lda, g.ldb),
               "gemm: row sums of <d pre, pre> (MASK_POS + ssq_out) exist in the wide bf16 kernel only (caller: check p5l_gemm_gate_ok)");
  if (g.epi == P5_EPI_GELU_GATE || g.epi == P5_EPI_GELU_GATE_BWD) {
    P5_REQUIRE(sizeof(T) == 2 && !g.a_ks && !g.b_ks && g.splitk <= 1 && !g.c_f32 && p5l_gemm_gate_ok(g.M, g.N, g.K, g.lda, g.ldb),
               "gemm: the gated-GELU epilogues need the whole-tile path of the wide bf16 kernel (caller: check p5l_gemm_gate_ok)");
    P5_REQUIRE((g.ldc % 8) == 0 && ((uintptr_t)g.C % 16) == 0, "gemm: gated-GELU epilogue output alignment");
    if (g.epi == P5_EPI_GELU_GATE) P5_REQUIRE(g.C2 && g.gate_F * 2 == g.N && (g.ldc2 % 8) == 0 && ((uintptr_t)g.C2 % 16) == 0 && !g.ssq_out, "gemm: gated-GELU forward arguments");
    else P5_REQUIRE(g.aux && (g.ldaux % 8) == 0 && ((uintptr_t)g.aux % 16) == 0 && !g.rowss && !g.ssq_out && g.gate_F == 0, "gemm: gated-GELU backward arguments");
  } else {
    g.C2 = nullptr; g.ldc2 = 0; g.gate_F = 0;
  }
  P5_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem");
  P5_REQUIRE(g.lda % EPF == 0 && g.ldb % EPF == 0, "gemm: leading dims must be multiples of 16 bytes");
  P5_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm: operands must be 16-byte aligned");
  if (!g.a_ks) P5_REQUIRE(g.K % EPF == 0 || g.lda >= ((g.K + EPF - 1) / EPF) * EPF, "gemm: A K-extent");
  if (!g.b_ks) P5_REQUIRE(g.K % EPF == 0 || g.ldb >= ((g.K + EPF - 1) / EPF) * EPF, "gemm: B K-extent");
  if (g.a_ks) P5_REQUIRE(g.M % EPF == 0 || g.lda >= ((g.M + EPF - 1) / EPF) * EPF, "gemm: A M-extent (KS)");
  if (g.b_ks) P5_REQUIRE(g.N % EPF == 0 || g.ldb >= ((g.N + EPF - 1) / EPF) * EPF, "gemm: B N-extent (KS)");
  if (g.epi == P5_EPI_ATOMIC || g.epi == P5_EPI_ACCUM) P5_REQUIRE(g.c_f32, "gemm: accumulate epilogues need fp32 C");
  const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const int force_tile = g_opt_gemm_tile;
  if constexpr (sizeof(T) == 2) {
    const bool kc = !g.a_ks && !g.b_ks && (g.K % 64) == 0 && g.splitk <= 1 && (g.lda % 64) == 0 && (g.ldb % 64) == 0 && !force_tile && !g_opt_gemm_v2;
    // wide outputs: 256x128 tiles, eight waves, persistent three-slot ring (p5_gemm4.h) once there are enough of them to occupy most
    // CUs -- 0.75 of the L2->LDS bytes per MAC of a 128x128 tile and two waves per SIMD to overlap LDS reads with MFMAs
    // (tools/lab, round 3: 8192x2048x512 25.3 vs 27.6 us, 8192x3072x768 47 vs 54, 8192x4096x1024 73 vs 87 (128x128) / 125 (256x256
    // eight-wave two-slot kernel), 8192^2 x 2048 252 vs 382 us)
    if (kc && g_opt_gemm_wide && (long)((g.M + 255) / 256) * ((g.N + 127) / 128) >= g_opt_gemm_wide_min_tiles && g.epi != P5_EPI_ATOMIC && g.epi != P5_EPI_ACCUM) {
      P5GemmGroup grp;
      memset(&grp, 0, sizeof(grp));
      grp.nprob = 1;
      grp.p[0] = g;
      grp.p[0].splitk = 1;
      return launch_gemm4(P5_G4_256x128, false, grp, s);
    }
    // narrow outputs (N = d_model): one 128x128 tile per CU, loader waves + compute waves (round 6)
    if (kc && g_opt_gemm_ws128 && (g_opt_gemm_ws & 1) && t128 >= g_opt_gemm_ring128_min_tiles && t128 <= 256 && g.K >= g_opt_gemm_ws128_min_k && g.epi != P5_EPI_ATOMIC &&
        g.epi != P5_EPI_ACCUM) {
      P5GemmGroup grp;
      memset(&grp, 0, sizeof(grp));
      grp.nprob = 1;
      grp.p[0] = g;
      grp.p[0].splitk = 1;
      return launch_gemm4(P5_G5_128x128, false, grp, s);
    }
    // narrow outputs (N = d_model) with a long reduction: one 128x128 tile per CU on the four-slot ring instead of 64x64 tiles
    if (kc && g_opt_gemm_ring_n512 && t128 >= g_opt_gemm_ring128_min_tiles && t128 <= 256 && g.K >= g_opt_gemm_ring128_min_k && g.epi != P5_EPI_ATOMIC) {
      g.ring = 1;
      g.splitk = 1;
      return launch_gemm_tile<T, 128, 128>(g, s);
    }
  }
  static const int split_target = getenv("P5_GEMM_SPLIT_TARGET") ? atoi(getenv("P5_GEMM_SPLIT_TARGET")) : 768;
  // measured on MI355X (tools/gemm_bench2.py): 128x128 tiles win once there are >= 2 full rounds of them, 64x64 below
  bool big = force_tile ? force_tile == 128 : (t128 >= 512 || (g.epi == P5_EPI_ATOMIC && g.K >= 16384));
  if (sizeof(T) == 4 && g.mm_split && !force_tile && g_opt_split_big_tiles > 0 && t128 >= g_opt_split_big_tiles) big = true;   // (split products: the wider wave tile halves the conversions per MFMA)
  // weight gradients (both operands K-strided, long K, few tiles): the four-slot-ring kernel, one 128x128 workgroup per CU,
  // split-K so that tiles x splits ~ 160: in isolation ~256 (every CU) is fastest, inside the step fewer, longer workgroups leave
  // CUs to the main stream (tools/wgrad_bench.py: 8192-deep 512x2048 50.9 -> 34.4 us, 2048x512 39.6 -> 32.6 us;
  // below 48 tiles the 64x64 kernel still wins)
  if (sizeof(T) == 2 && !force_tile && g_opt_gemm_ring && g.a_ks && g.b_ks && g.epi == P5_EPI_ATOMIC && g.splitk <= 0 && t128 >= 48 &&
      t128 <= 256 && g.K >= 2048 && (g.K % 64) == 0) {
    g.ring = 1;
    big = true;
    int sk = (int)((g_opt_gemm_ring_wgs + t128 / 2) / t128);
    const int maxs = g.K / 64 / 8;
    g.splitk = sk < 1 ? 1 : (sk > maxs ? maxs : sk);
  }
  const long tiles = big ? t128 : (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
  if (sizeof(T) == 2 && !big && !force_tile && g_opt_gemm_small_ring && !g.ring && tiles <= g_opt_gemm_small_ring_tiles && g.K >= 256 && (g.K % 64) == 0 &&
      (g.a_ks == 0 || g.b_ks == 1) &&
      (g.epi != P5_EPI_ATOMIC || g.K <= 1024)) {
    g.ring = 1;                      // (long-K atomic problems keep the split-K path below)
    if (g.splitk <= 0) g.splitk = 1;
  }
  if (g.splitk <= 0) {
    g.splitk = 1;
    if (g.epi == P5_EPI_ATOMIC) {
      const int nkc = (g.K + TT<T>::KCH - 1) / TT<T>::KCH;
      int want = (int)(((g.K >= 16384 ? 384 : split_target) + tiles - 1) / tiles);
      int maxs = nkc / 8 > 0 ? nkc / 8 : 1;
      g.splitk = want < maxs ? want : maxs;
      if (g.splitk < 1) g.splitk = 1;
    }
  }
  if (g.splitk > 1) P5_REQUIRE(g.epi == P5_EPI_ATOMIC, "gemm: split-K needs the atomic epilogue");
  if (g.c_split_stride > 0) return big ? launch_gemm_tile<T, 128, 128>(g, s) : launch_gemm_tile<T, 64, 64>(g, s);    // (blockIdx.z = split index)
  if constexpr (sizeof(T) == 2) {
    // 256x256 tiles halve the L2->LDS bytes per MAC; they pay off once every CU gets a tile and the K loop is long enough to
    // amortise the un-overlapped prologue/epilogue of the single resident workgroup (tools/gemm_v2_bench.py: 8192x2048x2048
    // 80 -> 63 us, 4096^3 150 -> 110 us; at K = 512 it is a wash)
    const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
    const bool kc_dma = !g.a_ks && !g.b_ks && (g.K % 64) == 0;
    if (force_tile == 256 || (!force_tile && kc_dma && t256 >= 256 && g.K >= 1024 && g.splitk <= 1)) {
      if (g.splitk <= 0) g.splitk = 1;
      return launch_gemm_tile<T, 256, 256>(g, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // the decoder's 512-row problems: 64 tiles of 64x64 use a quarter of the CUs, each pulling 128 KiB through its ring; 32x64
    // tiles double the workgroups and halve the A rows each one waits for
    if (!big && g.ring && !g.a_ks && !g.b_ks && g.splitk <= 1 && tiles <= g_opt_gemm_ring32) {
      dim3 grid((g.N + 63) / 64, (g.M + 31) / 32, 1);
      g.xcd_bm = g.xcd_bn = 0;
      g.splitk = 1;
      P5_PROF_FLOPS(2.0 * g.M * g.N * g.K);
      P5_PROF_SHAPE(g.M, g.N, g.K);
      P5_LAUNCH((p5_gemm2_kernel<32, 64, 8, false, false>), grid, dim3(256), 0, s, g);
      return P5_KCHECK();
    }
  }
  return big ? launch_gemm_tile<T, 128, 128>(g, s) : launch_gemm_tile<T, 64, 64>(g, s);
}


int p5l_gemm_bf16(P5GemmArgs g, hipStream_t s) { return launch_gemm_impl<bf16>(g, s); }
int p5l_gemm_f32(P5GemmArgs g, hipStream_t s) { return launch_gemm_impl<float>(g, s); }
