// gemm_lab.hip -- standalone A/B harness for the GEMM kernels of openp5_amd/csrc (dev tool, not part of the product):
// times every variant on the exact shapes of the T5-small training step, checks each against a plain fp32 reference kernel,
// and runs the ablated builds of the persistent ring kernel (no MFMA / no copies / no fragment reads / no epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openp5_amd/csrc tools/lab/gemm_lab.hip -o tools/lab/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#include "p5_gemm4.h"
#include "p5_gemm5.h"
#include "p5_gemm6_experimental.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void ref_kernel(float* C, const bf16* A, const bf16* B, int M, int N, int K, int lda, int ldb, int aks, int bks) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = bf2f(aks ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]);
    const float b = bf2f(bks ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k]);
    s += a * b;
  }
  C[(size_t)m * N + n] = s;
}
__global__ void fill_kernel(bf16* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t h = p5_mix32((uint32_t)i * 2654435761u + seed);
    p[i] = f2bf(((float)(h >> 8) / 8388608.0f - 1.0f));       // uniform [-1, 1)
  }
}
__global__ void diff_kernel(float* out, const void* C, const float* ref, size_t n, int c_f32, float scale) {
  float m = 0.f, r = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = c_f32 ? ((const float*)C)[i] : bf2f(((const bf16*)C)[i]);
    m = fmaxf(m, fabsf(v * scale - ref[i]));
    r = fmaxf(r, fabsf(ref[i]));
  }
  atomicMax((int*)out, __float_as_int(m));
  atomicMax((int*)out + 1, __float_as_int(r));
}

struct Prob {
  int M, N, K, aks, bks, c_f32, epi;
  bf16 *A, *B; void* C; float* ref;
  int lda, ldb;
};
static Prob make_prob(int M, int N, int K, int ks, int c_f32, int epi, uint32_t seed) {
  Prob p;
  p.M = M; p.N = N; p.K = K; p.aks = ks; p.bks = ks; p.c_f32 = c_f32; p.epi = epi;
  p.lda = ks ? M : K; p.ldb = ks ? N : K;
  CK(hipMalloc(&p.A, (size_t)M * K * 2)); CK(hipMalloc(&p.B, (size_t)N * K * 2));
  CK(hipMalloc(&p.C, (size_t)M * N * 4)); CK(hipMalloc(&p.ref, (size_t)M * N * 4));
  fill_kernel<<<1024, 256>>>(p.A, (size_t)M * K, seed);
  fill_kernel<<<1024, 256>>>(p.B, (size_t)N * K, seed * 7 + 1);
  ref_kernel<<<dim3((N + 255) / 256, M), 256>>>(p.ref, p.A, p.B, M, N, K, p.lda, p.ldb, ks, ks);
  CK(hipDeviceSynchronize());
  return p;
}
static P5GemmArgs args_of(const Prob& p, int splitk = 1) {
  P5GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = p.A; g.B = p.B; g.C = p.C; g.aux = nullptr; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = p.lda; g.ldb = p.ldb; g.ldc = p.N; g.ldaux = 0;
  g.a_ks = p.aks; g.b_ks = p.bks; g.epi = p.epi; g.c_f32 = p.c_f32; g.splitk = splitk; g.alpha = 1.f;
  g.drop.state = nullptr; g.drop.thr = 0; g.drop.scale = 1.f;
  g.rowss = nullptr; g.ssq_out = nullptr;
  return g;
}
static float* d_diff = nullptr;
// runs `launch` (which may be several kernel launches), returns avg us; checks the result of ONE run against the reference.
// accum_runs: how many times C has been accumulated into when checked (atomic / accumulate epilogues start from zeroed C)
static void bench(const char* name, const std::vector<Prob>& ps, const std::function<void()>& launch, bool accumulates, int iters = 30) {
  double flops = 0;
  for (const Prob& p : ps) flops += 2.0 * p.M * p.N * p.K;
  for (const Prob& p : ps) CK(hipMemset(p.C, 0, (size_t)p.M * p.N * 4));
  launch();
  CK(hipDeviceSynchronize());
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { printf("  %-44s LAUNCH ERROR %s\n", name, hipGetErrorString(le)); return; }
  float worst = 0.f;
  for (const Prob& p : ps) {
    CK(hipMemset(d_diff, 0, 8));
    diff_kernel<<<512, 256>>>(d_diff, p.C, p.ref, (size_t)p.M * p.N, p.c_f32, 1.f);
    float h[2];
    CK(hipMemcpy(h, d_diff, 8, hipMemcpyDeviceToHost));
    worst = fmaxf(worst, h[0] / fmaxf(h[1], 1e-6f));
  }
  (void)accumulates;
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f / iters);
  }
  std::sort(ts.begin(), ts.end());
  printf("  %-44s %8.1f us (min %7.1f)  %7.1f TF/s  relerr %.1e%s\n", name, ts[1], ts[0], flops / ts[1] / 1e6, worst, worst > 2e-2f ? "  <-- MISMATCH" : "");
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

// ---- launch helpers -----------------------------------------------------------------------------------
template <int BM, int BN, int WMW, int WNW, int NST, bool KS, int ABL = 0, int OCC = 1, int FLAGS = 0>
static void launch_g4(const std::vector<P5GemmArgs>& gs, int max_wg = 256) {
  P5GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.nprob = (int)gs.size();
  int units = 0;
  for (int i = 0; i < grp.nprob; ++i) {
    P5GemmArgs g = gs[i];
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    g.g4_tiles_n = tn;
    g.g4_nk = g.K / 64 / g.splitk;
    grp.unit_begin[i] = units;
    units += tm * tn * g.splitk;
    grp.p[i] = g;
  }
  grp.unit_begin[grp.nprob] = units;
  grp.total_units = units;
  int nwg = ((units + 7) / 8) * 8;
  if (nwg > max_wg) nwg = max_wg;
  hipLaunchKernelGGL((p5_gemm4_kernel<BM, BN, WMW, WNW, NST, KS, ABL, OCC, FLAGS>), dim3(nwg), dim3(WMW * WNW * 64), 0, 0, grp);
}
template <bool KS, int ABL = 0>
static void launch_g5(const std::vector<P5GemmArgs>& gs, int max_wg = 256) {
  P5GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.nprob = (int)gs.size();
  int units = 0;
  for (int i = 0; i < grp.nprob; ++i) {
    P5GemmArgs g = gs[i];
    const int tm = (g.M + 255) / 256, tn = (g.N + 127) / 128;
    g.g4_tiles_n = tn;
    g.g4_nk = g.K / 64 / g.splitk;
    grp.unit_begin[i] = units;
    units += tm * tn * g.splitk;
    grp.p[i] = g;
  }
  grp.unit_begin[grp.nprob] = units;
  grp.total_units = units;
  int nwg = ((units + 7) / 8) * 8;
  if (nwg > max_wg) nwg = max_wg;
  hipLaunchKernelGGL((p5_gemm5_kernel<KS, ABL>), dim3(nwg), dim3(512), 0, 0, grp);
}
// write probe: `nwg` workgroups of 256 threads write `bytes` of bf16 output.  mode 0: every wave instruction one contiguous 1 KiB;
// mode 1: the register epilogue's pattern (a wave instruction = 16 rows x 64 bytes, row pitch `pitch` bytes); mode 2: 4 rows x 256 bytes
__global__ __launch_bounds__(256) void write_probe(char* out, size_t bytes, int mode, int pitch) {
  const size_t nchunk = bytes / 1024;                 // 1 KiB per wave instruction
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (size_t)gridDim.x * 4;
  const u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
  for (size_t c = wave; c < nchunk; c += nwave) {
    size_t off;
    if (mode == 0) off = c * 1024 + lane * 16;
    else if (mode == 1) {          // chunk c: rows (c / (pitch/64)) * 16 .. +15, 64-byte column segment c % (pitch/64)
      const size_t segs = pitch / 64, rb = c / segs, sg = c % segs;
      off = (rb * 16 + (lane & 15)) * (size_t)pitch + sg * 64 + (lane >> 4) * 16;
    } else {
      const size_t segs = pitch / 256, rb = c / segs, sg = c % segs;
      off = (rb * 4 + (lane >> 4)) * (size_t)pitch + sg * 256 + (lane & 15) * 16;
    }
    *(u32x4*)(out + off) = v;
  }
}
template <int ABL = 0>
static void launch_g6(const std::vector<P5GemmArgs>& gs, int max_wg = 256) {
  P5GemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.nprob = (int)gs.size();
  int units = 0;
  for (int i = 0; i < grp.nprob; ++i) {
    P5GemmArgs g = gs[i];
    g.g4_tiles_n = g.N / 128;
    g.g4_nk = g.K / 32;
    grp.unit_begin[i] = units;
    units += (g.M / 256) * g.g4_tiles_n;
    grp.p[i] = g;
  }
  grp.unit_begin[grp.nprob] = units;
  grp.total_units = units;
  int nwg = ((units + 7) / 8) * 8;
  if (nwg > max_wg) nwg = max_wg;
  hipLaunchKernelGGL((p5_gemm6_kernel<ABL>), dim3(nwg), dim3(512), 0, 0, grp);
}
template <int BM, int BN>
static void set_rect(P5GemmArgs& g) {     // the launcher's XCD rectangle choice (p5_lib.hip::launch_gemm_tile)
  const int gx = (g.N + BN - 1) / BN, gy = (g.M + BM - 1) / BM;
  g.xcd_bm = g.xcd_bn = 0;
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
template <int BM, int BN>
static void launch_v1(P5GemmArgs g) {     // p5_gemm_kernel, direct-to-LDS both operands
  set_rect<BM, BN>(g);
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk);
  if (g.a_ks) hipLaunchKernelGGL((p5_gemm_kernel<bf16, BM, BN, true, true, 2, true, true>), grid, dim3(256), 0, 0, g);
  else hipLaunchKernelGGL((p5_gemm_kernel<bf16, BM, BN, false, false, 2, true, true>), grid, dim3(256), 0, 0, g);
}
template <int BM, int BN, int NST, bool KS>
static void launch_ring(P5GemmArgs g) {   // p5_gemm2_kernel
  set_rect<BM, BN>(g);
  g.ring = 1;
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk);
  hipLaunchKernelGGL((p5_gemm2_kernel<BM, BN, NST, KS, KS>), grid, dim3(256), 0, 0, g);
}
static void launch_256(P5GemmArgs g) {
  set_rect<256, 256>(g);
  dim3 grid((g.N + 255) / 256, (g.M + 255) / 256, 1);
  hipLaunchKernelGGL((p5_gemm3_kernel<256, 256, 2, 4>), grid, dim3(512), 0, 0, g);
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "all";
  CK(hipMalloc(&d_diff, 8));
  const bool all = !strcmp(which, "all");
  if (all || !strcmp(which, "fwd")) {
    const int shapes[][3] = {{8192, 2048, 512}, {8192, 1536, 512}, {8192, 512, 512}, {8192, 512, 2048}, {8192, 512, 1536}, {512, 512, 512}, {512, 2048, 512}};
    for (auto& s : shapes) {
      Prob p = make_prob(s[0], s[1], s[2], 0, 0, P5_EPI_STORE, 11);
      printf("FWD  M=%d N=%d K=%d (bf16 C)\n", s[0], s[1], s[2]);
      P5GemmArgs g = args_of(p);
      bench("v1 128x128 2-stage", {p}, [&] { launch_v1<128, 128>(g); }, false);
      bench("v1 64x64 2-stage", {p}, [&] { launch_v1<64, 64>(g); }, false);
      bench("ring 128x128 NST=3 (gemm2)", {p}, [&] { launch_ring<128, 128, 3, false>(g); }, false);
      bench("ring 128x128 NST=4 (gemm2)", {p}, [&] { launch_ring<128, 128, 4, false>(g); }, false);
      if (s[0] >= 256 && s[1] >= 256) bench("256x256 8 waves 2-stage (gemm3)", {p}, [&] { launch_256(g); }, false);
      bench("g4 128x128 NST=5 persistent", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false>({g}); }, false);
      bench("g4 128x128 NST=4 persistent", {p}, [&] { launch_g4<128, 128, 2, 2, 4, false>({g}); }, false);
      bench("g4 128x128 NST=3 persistent", {p}, [&] { launch_g4<128, 128, 2, 2, 3, false>({g}); }, false);
      bench("g4 256x128 8w NST=3 persistent", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false>({g}); }, false);
      bench("g4 128x256 8w NST=3 persistent", {p}, [&] { launch_g4<128, 256, 2, 4, 3, false>({g}); }, false);
      bench("g4 128x128 NST=5, one unit per WG", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false>({g}, 1 << 20); }, false);
      if (s[0] == 8192 && (s[1] == 2048 || s[2] == 2048)) {
        bench("  abl: no MFMA", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 1>({g}); }, false);
        bench("  abl: no copies", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 2>({g}); }, false);
        bench("  abl: no frag reads", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 4>({g}); }, false);
        bench("  abl: no epilogue", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 8>({g}); }, false);
        bench("  abl: copies only (no MFMA, no reads)", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 5>({g}); }, false);
        bench("  abl: MFMA only", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 14>({g}); }, false);
        bench("  abl: MFMA + reads", {p}, [&] { launch_g4<128, 128, 2, 2, 5, false, 10>({g}); }, false);
      }
      CK(hipFree(p.A)); CK(hipFree(p.B)); CK(hipFree(p.C)); CK(hipFree(p.ref));
    }
  }
  if (all || !strcmp(which, "wgrad")) {
    const int shapes[][2] = {{2048, 512}, {512, 2048}, {1536, 512}, {512, 512}};
    std::vector<Prob> ps;
    for (auto& s : shapes) ps.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 23 + s[0]));
    double old_sum = 0;
    for (Prob& p : ps) {
      printf("WGRAD out %dx%d over 8192 tokens (fp32 C)\n", p.M, p.N);
      const int t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
      int sk = (160 + t128 / 2) / t128;
      sk = sk < 1 ? 1 : sk;
      P5GemmArgs ga = args_of(p, sk);
      char nm[96];
      snprintf(nm, sizeof nm, "ring NST=4 split-K %d atomics (today)", sk);
      if (t128 >= 48) bench(nm, {p}, [&] { launch_ring<128, 128, 4, true>(ga); }, true);
      else { P5GemmArgs g64 = args_of(p, 12); bench("v1 64x64 split-K 12 atomics (today)", {p}, [&] { launch_v1<64, 64>(g64); }, true); }
      P5GemmArgs g4s = args_of(p, 4);
      bench("ring NST=4 split-K 4 atomics", {p}, [&] { launch_ring<128, 128, 4, true>(g4s); }, true);
      P5GemmArgs g1 = args_of(p, 1);
      g1.epi = P5_EPI_ACCUM;
      bench("g4 KS NST=5 split 1, C += (one problem)", {p}, [&] { launch_g4<128, 128, 2, 2, 5, true>({g1}); }, true);
      P5GemmArgs g2 = args_of(p, 2);
      bench("g4 KS NST=5 split 2 atomics", {p}, [&] { launch_g4<128, 128, 2, 2, 5, true>({g2}); }, true);
      P5GemmArgs g4 = args_of(p, 4);
      bench("g4 KS NST=5 split 4 atomics", {p}, [&] { launch_g4<128, 128, 2, 2, 5, true>({g4}); }, true);
    }
    printf("WGRAD grouped: the four weight gradients of one encoder layer in ONE launch\n");
    {
      std::vector<P5GemmArgs> gs;
      for (Prob& p : ps) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; gs.push_back(g); }
      bench("g4 KS NST=5 grouped, split 1, C += (192 units)", ps, [&] { launch_g4<128, 128, 2, 2, 5, true>(gs); }, true);
      bench("g4 KS NST=4 grouped, split 1, C +=", ps, [&] { launch_g4<128, 128, 2, 2, 4, true>(gs); }, true);
      bench("g4 KS NST=3 grouped, split 1, C +=", ps, [&] { launch_g4<128, 128, 2, 2, 3, true>(gs); }, true);
      std::vector<P5GemmArgs> gs2;
      for (Prob& p : ps) gs2.push_back(args_of(p, 2));
      bench("g4 KS NST=5 grouped, split 2 atomics (384 units)", ps, [&] { launch_g4<128, 128, 2, 2, 5, true>(gs2); }, true);
      bench("  abl grouped: no MFMA", ps, [&] { launch_g4<128, 128, 2, 2, 5, true, 1>(gs); }, true);
      bench("  abl grouped: no copies", ps, [&] { launch_g4<128, 128, 2, 2, 5, true, 2>(gs); }, true);
      bench("  abl grouped: no frag reads", ps, [&] { launch_g4<128, 128, 2, 2, 5, true, 4>(gs); }, true);
      bench("  abl grouped: copies only", ps, [&] { launch_g4<128, 128, 2, 2, 5, true, 5>(gs); }, true);
      bench("  abl grouped: MFMA + reads", ps, [&] { launch_g4<128, 128, 2, 2, 5, true, 10>(gs); }, true);
      // today's four launches back to back
      std::vector<std::function<void()>> today;
      std::vector<P5GemmArgs> keep;
      for (Prob& p : ps) {
        const int t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        int sk = (160 + t128 / 2) / t128;
        keep.push_back(args_of(p, t128 >= 48 ? (sk < 1 ? 1 : sk) : 12));
      }
      bench("today: four launches (ring split-K / 64x64)", ps, [&] {
        for (size_t i = 0; i < ps.size(); ++i) {
          const int t128 = ((ps[i].M + 127) / 128) * ((ps[i].N + 127) / 128);
          if (t128 >= 48) launch_ring<128, 128, 4, true>(keep[i]); else launch_v1<64, 64>(keep[i]);
        }
      }, true);
    }
    (void)old_sum;
  }
  if (!strcmp(which, "lab2") || !strcmp(which, "fwd2")) {
    const int shapes[][3] = {{8192, 2048, 512}, {8192, 1536, 512}, {8192, 512, 2048}, {8192, 512, 512}, {8192, 3072, 768}, {8192, 768, 3072}, {8192, 4096, 1024},
                             {8192, 8192, 2048}};
    for (auto& s : shapes) {
      Prob p = make_prob(s[0], s[1], s[2], 0, 0, P5_EPI_STORE, 11);
      printf("FWD2 M=%d N=%d K=%d (bf16 C)\n", s[0], s[1], s[2]);
      P5GemmArgs g = args_of(p);
      bench("v1 128x128 2-stage", {p}, [&] { launch_v1<128, 128>(g); }, false);
      bench("ring 128x128 NST=4 (gemm2)", {p}, [&] { launch_ring<128, 128, 4, false>(g); }, false);
      if (s[1] >= 1024) bench("256x256 8 waves 2-stage (gemm3)", {p}, [&] { launch_256(g); }, false);
      bench("g4 256x128 8w NST=3", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false>({g}); }, false);
      bench("g4 256x128 8w NST=3 K-rotated", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false, 0, 1, 1>({g}); }, false);
      bench("g4 128x128 NST=3 K-rotated", {p}, [&] { launch_g4<128, 128, 2, 2, 3, false, 0, 1, 1>({g}); }, false);
      bench("g4 128x128 NST=2 two WGs/CU", {p}, [&] { launch_g4<128, 128, 2, 2, 2, false, 0, 2, 0>({g}, 512); }, false);
      bench("g4 128x128 NST=2 two WGs/CU, one unit each", {p}, [&] { launch_g4<128, 128, 2, 2, 2, false, 0, 2, 0>({g}, 1 << 20); }, false);
      bench("g5 256x256 4w (128x128 wave tiles) NST=2", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false>({g}); }, false);
      bench("g5 256x256 4w NST=2 K-rotated", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 0, 1, 1>({g}); }, false);
      if (s[1] == 2048 || s[1] == 8192) {
        bench("  g5 abl: no MFMA", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 1>({g}); }, false);
        bench("  g5 abl: no copies", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 2>({g}); }, false);
        bench("  g5 abl: no frag reads", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 4>({g}); }, false);
        bench("  g5 abl: no epilogue", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 8>({g}); }, false);
        bench("  g5 abl: MFMA + reads", {p}, [&] { launch_g4<256, 256, 2, 2, 2, false, 10>({g}); }, false);
      }
      CK(hipFree(p.A)); CK(hipFree(p.B)); CK(hipFree(p.C)); CK(hipFree(p.ref));
    }
  }
  if (!strcmp(which, "lab2") || !strcmp(which, "wgrad2")) {
    const int shapes[][2] = {{2048, 512}, {512, 2048}, {1536, 512}, {512, 512}};
    std::vector<Prob> ps;
    for (auto& s : shapes) ps.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 23 + s[0]));
    std::vector<P5GemmArgs> gs;
    for (Prob& p : ps) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; gs.push_back(g); }
    printf("WGRAD2 grouped (192 units of 128x128 over 8192 tokens)\n");
    bench("g4 KS NST=3", ps, [&] { launch_g4<128, 128, 2, 2, 3, true>(gs); }, true);
    bench("g4 KS NST=3 K-rotated", ps, [&] { launch_g4<128, 128, 2, 2, 3, true, 0, 1, 1>(gs); }, true);
    bench("g4 KS NST=2 two WGs/CU (192 WGs)", ps, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 0>(gs, 512); }, true);
    bench("g4 KS NST=2 two WGs/CU K-rotated", ps, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 1>(gs, 512); }, true);
    std::vector<P5GemmArgs> gs2;
    for (Prob& p : ps) gs2.push_back(args_of(p, 2));
    bench("g4 KS NST=2 two WGs/CU split 2 atomics (384 WGs)", ps, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 0>(gs2, 512); }, true);
    // two layers' worth in one launch (384 units: 1.5 rounds of 256 persistent workgroups)
    std::vector<Prob> ps2 = ps;
    for (auto& s : shapes) ps2.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 77 + s[0]));
    std::vector<P5GemmArgs> g8;
    for (Prob& p : ps2) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; g8.push_back(g); }
    bench("g4 KS NST=3, TWO layers in one launch (384 units)", ps2, [&] { launch_g4<128, 128, 2, 2, 3, true>(g8); }, true);
    bench("g4 KS NST=2 two WGs/CU, TWO layers (384 WGs)", ps2, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 0>(g8, 512); }, true);
    // the tied head's weight gradient: [32100 x 512] over 512 tokens
    Prob ph = make_prob(32100, 512, 512, 1, 1, P5_EPI_ATOMIC, 5);
    printf("WGRAD2 head: 32100x512 over 512 tokens\n");
    P5GemmArgs gh = args_of(ph, 1);
    bench("v1 128x128 atomics (today)", {ph}, [&] { launch_v1<128, 128>(gh); }, true);
    P5GemmArgs gha = args_of(ph, 1);
    gha.epi = P5_EPI_ACCUM;
    bench("g4 KS NST=3 C +=", {ph}, [&] { launch_g4<128, 128, 2, 2, 3, true>({gha}); }, true);
    bench("g4 KS NST=2 two WGs/CU C +=", {ph}, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 0>({gha}, 512); }, true);
  }
  if (!strcmp(which, "lab3")) {
    const int shapes[][2] = {{2048, 512}, {512, 2048}, {1536, 512}, {512, 512}};
    std::vector<Prob> ps;
    for (auto& s : shapes) ps.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 23 + s[0]));
    std::vector<Prob> ps2 = ps;
    for (auto& s : shapes) ps2.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 77 + s[0]));
    std::vector<P5GemmArgs> g4, g8;
    for (Prob& p : ps) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; g4.push_back(g); }
    for (Prob& p : ps2) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; g8.push_back(g); }
    printf("WGRAD3: one layer (4 problems) / two layers (8 problems) per launch\n");
    bench("128x128 4w NST=3, one layer (192 units)", ps, [&] { launch_g4<128, 128, 2, 2, 3, true>(g4); }, true);
    bench("256x128 8w NST=3, one layer (96 units)", ps, [&] { launch_g4<256, 128, 4, 2, 3, true>(g4); }, true);
    bench("128x256 8w NST=3, one layer (96 units)", ps, [&] { launch_g4<128, 256, 2, 4, 3, true>(g4); }, true);
    bench("128x128 4w NST=3, two layers (384 units)", ps2, [&] { launch_g4<128, 128, 2, 2, 3, true>(g8); }, true);
    bench("128x128 4w NST=2 2 WGs/CU, two layers", ps2, [&] { launch_g4<128, 128, 2, 2, 2, true, 0, 2, 0>(g8, 512); }, true);
    bench("256x128 8w NST=3, two layers (192 units)", ps2, [&] { launch_g4<256, 128, 4, 2, 3, true>(g8); }, true);
    bench("128x256 8w NST=3, two layers (192 units)", ps2, [&] { launch_g4<128, 256, 2, 4, 3, true>(g8); }, true);
    bench("  abl 256x128 two layers: copies only", ps2, [&] { launch_g4<256, 128, 4, 2, 3, true, 5>(g8); }, true);
    bench("  abl 256x128 two layers: MFMA + reads", ps2, [&] { launch_g4<256, 128, 4, 2, 3, true, 10>(g8); }, true);
  }
  if (!strcmp(which, "lab4") || !strcmp(which, "wprobe")) {
    char* buf;
    const size_t bytes = (size_t)8192 * 2048 * 2;
    CK(hipMalloc(&buf, bytes * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("WRITE PROBE: %.1f MB of output per launch (8192 x 2048 bf16), 30 launches back to back\n", bytes / 1e6);
    for (int rot = 0; rot < 2; ++rot)
      for (int mode = 0; mode < 3; ++mode)
        for (int nwg : {256, 1024, 4096}) {
          for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(write_probe, dim3(nwg), dim3(256), 0, 0, buf, bytes, mode, 4096);
          CK(hipEventRecord(e0));
          for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(write_probe, dim3(nwg), dim3(256), 0, 0, buf + (rot ? (size_t)(i % 8) * bytes : 0), bytes, mode, 4096);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          printf("  mode %d (%s) wgs %4d %s: %6.1f us  %.2f TB/s\n", mode, mode == 0 ? "1 KiB contiguous" : mode == 1 ? "16 rows x 64 B" : "4 rows x 256 B", nwg,
                 rot ? "rotating over 8 buffers" : "same buffer", ms * 1e3 / 30, bytes / (ms * 1e-3 / 30) / 1e12);
        }
    CK(hipFree(buf));
  }
  if (!strcmp(which, "lab4")) {
    const int fshapes[][3] = {{8192, 2048, 512}, {8192, 1536, 512}, {8192, 512, 2048}, {8192, 512, 512}, {8192, 3072, 768}, {8192, 768, 3072}, {8192, 4096, 1024},
                              {8192, 8192, 2048}, {8000, 2000, 512}};
    for (auto& s : fshapes) {
      Prob p = make_prob(s[0], s[1], s[2], 0, 0, P5_EPI_STORE, 11);
      printf("FWD4 M=%d N=%d K=%d (bf16 C)\n", s[0], s[1], s[2]);
      P5GemmArgs g = args_of(p);
      bench("g4 256x128 8w NST=3", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false>({g}); }, false);
      bench("g5 256x128 4 compute + 4 loader waves", {p}, [&] { launch_g5<false>({g}); }, false);
      const bool g6ok = s[0] % 256 == 0 && s[1] % 128 == 0;
      if (g6ok) bench("g6 256x128 loaders drain the tiles", {p}, [&] { launch_g6<0>({g}); }, false);
      {
        static uint32_t* d_state = nullptr;
        if (!d_state) { CK(hipMalloc(&d_state, 8)); uint32_t h[2] = {77, 3}; CK(hipMemcpy(d_state, h, 8, hipMemcpyHostToDevice)); }
        P5GemmArgs gd = g;
        gd.epi = P5_EPI_RELU_DROP; gd.drop.state = d_state; gd.drop.site_key = p5_site_key(5); gd.drop.thr = p5_drop_thr(0.1f); gd.drop.scale = 1.f / 0.9f;
        bench("g4 256x128, ReLU + dropout epilogue (timing only)", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false>({gd}); }, false);
        bench("g5 256x128, ReLU + dropout epilogue (timing only)", {p}, [&] { launch_g5<false>({gd}); }, false);
        if (g6ok) bench("g6 256x128, ReLU + dropout epilogue (timing only)", {p}, [&] { launch_g6<0>({gd}); }, false);
        P5GemmArgs gr = g;
        gr.epi = P5_EPI_RESID_DROP; gr.drop = gd.drop; gr.aux = p.C; gr.ldaux = p.N;
        bench("g4 256x128, dropout + residual epilogue (timing only)", {p}, [&] { launch_g4<256, 128, 4, 2, 3, false>({gr}); }, false);
        bench("g5 256x128, dropout + residual epilogue (timing only)", {p}, [&] { launch_g5<false>({gr}); }, false);
        if (g6ok) bench("g6 256x128, dropout + residual epilogue (timing only)", {p}, [&] { launch_g6<0>({gr}); }, false);
      }
      if (s[1] == 2048 || s[1] == 8192) {
        bench("  g6 abl: no MFMA", {p}, [&] { launch_g6<1>({g}); }, false);
        bench("  g6 abl: no copies", {p}, [&] { launch_g6<2>({g}); }, false);
        bench("  g6 abl: no frag reads", {p}, [&] { launch_g6<4>({g}); }, false);
        bench("  g6 abl: no staging / drain of real data (compute side)", {p}, [&] { launch_g6<8>({g}); }, false);
        bench("  g6 abl: drain without stores", {p}, [&] { launch_g6<16>({g}); }, false);
        bench("  g6 abl: MFMA + reads", {p}, [&] { launch_g6<10>({g}); }, false);
        bench("  g5 abl: no MFMA", {p}, [&] { launch_g5<false, 1>({g}); }, false);
        bench("  g5 abl: no copies", {p}, [&] { launch_g5<false, 2>({g}); }, false);
        bench("  g5 abl: no frag reads", {p}, [&] { launch_g5<false, 4>({g}); }, false);
        bench("  g5 abl: no epilogue", {p}, [&] { launch_g5<false, 8>({g}); }, false);
        bench("  g5 abl: epilogue math, no stores", {p}, [&] { launch_g5<false, 16>({g}); }, false);
        bench("  g5 abl: copies only", {p}, [&] { launch_g5<false, 5>({g}); }, false);
        bench("  g5 abl: MFMA + reads", {p}, [&] { launch_g5<false, 10>({g}); }, false);
        bench("  g5 abl: MFMA only", {p}, [&] { launch_g5<false, 14>({g}); }, false);
      }
      CK(hipFree(p.A)); CK(hipFree(p.B)); CK(hipFree(p.C)); CK(hipFree(p.ref));
    }
    const int shapes[][2] = {{2048, 512}, {512, 2048}, {1536, 512}, {512, 512}};
    std::vector<Prob> ps;
    for (auto& s : shapes) ps.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 23 + s[0]));
    std::vector<Prob> ps2 = ps;
    for (auto& s : shapes) ps2.push_back(make_prob(s[0], s[1], 8192, 1, 1, P5_EPI_ATOMIC, 77 + s[0]));
    std::vector<P5GemmArgs> g4, g8;
    for (Prob& p : ps) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; g4.push_back(g); }
    for (Prob& p : ps2) { P5GemmArgs g = args_of(p, 1); g.epi = P5_EPI_ACCUM; g8.push_back(g); }
    printf("WGRAD4: one layer (4 problems) / two layers (8 problems) per launch\n");
    bench("g4 256x128 8w NST=3, one layer (96 units)", ps, [&] { launch_g4<256, 128, 4, 2, 3, true>(g4); }, true);
    bench("g5 256x128, one layer (96 units)", ps, [&] { launch_g5<true>(g4); }, true);
    bench("g4 256x128 8w NST=3, two layers (192 units)", ps2, [&] { launch_g4<256, 128, 4, 2, 3, true>(g8); }, true);
    bench("g5 256x128, two layers (192 units)", ps2, [&] { launch_g5<true>(g8); }, true);
    bench("  g5 abl two layers: copies only", ps2, [&] { launch_g5<true, 5>(g8); }, true);
    bench("  g5 abl two layers: MFMA + reads", ps2, [&] { launch_g5<true, 10>(g8); }, true);
    bench("  g5 abl two layers: MFMA only", ps2, [&] { launch_g5<true, 14>(g8); }, true);
  }
  printf("done\n");
  return 0;
}
