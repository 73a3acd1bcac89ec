// p5_host.h -- what the translation units of libp5hip.so share on the host side: the error channel, launch-check macros, the tuning
// knobs that live next to the kernels they steer, and the launcher entry points of the GEMM and attention families (p5_gemm_tu.hip,
// p5_attn_tu.hip), so that p5_lib.hip (engine + C ABI) does not instantiate those kernel templates itself and the three units compile in
// parallel.
#pragma once
#include <string>
#include <cstdlib>
#include "p5_device.h"
#include "p5_rng.h"
#include "p5_gemm.h"
#include "p5_gemm4.h"
#include "p5_attn.h"

extern thread_local std::string g_p5_err;
static inline int fail(const std::string& m) { g_p5_err = m; return -1; }
#define P5_REQUIRE(cond, msg) do { if (!(cond)) return fail(std::string(msg) + " [" #cond "]"); } while (0)
#define P5_TRY(expr) do { int _rc = (expr); if (_rc != 0) return _rc; } while (0)
#ifdef P5_EMU
#define P5_KCHECK() 0
#define P5_PROF_FLOPS(x) ((void)0)
#define P5_PROF_TAG(x) ((void)0)
#define P5_PROF_SHAPE(m, n, k) ((void)0)
#else
static inline int kcheck(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(std::string(where) + ": " + hipGetErrorString(e));
  return 0;
}
#define P5_KCHECK() kcheck(__func__)
#define P5_PROF_FLOPS(x) (p5_prof().pending_flops = (x))
#define P5_PROF_TAG(x) (p5_prof().pending_tag = (x))
#define P5_PROF_SHAPE(m, n, k) (p5_prof().pm = (m), p5_prof().pn = (n), p5_prof().pk = (k))
#endif

// ---- tuning knobs defined in p5_gemm_tu.hip / p5_attn_tu.hip (p5_set_option / environment) ----
extern int g_opt_gemm_v2;
extern int g_opt_gemm_tile;
extern int g_opt_gemm_ring;
extern int g_opt_gemm_xcd_rect;
extern int g_opt_gemm_small_ring;
extern int g_opt_gemm_ring32;
extern int g_opt_gemm_small_ring_tiles;
extern int g_opt_gemm_ring_stages;
extern int g_opt_gemm_ring_wgs;
extern int g_opt_gemm_ksdma;
extern int g_opt_gemm_wide;
extern int g_opt_gemm_wide_min_tiles;
extern int g_opt_gemm_ring_n512;
extern int g_opt_gemm_ring128_min_k;
extern int g_opt_gemm_ring128_min_tiles;
extern int g_opt_g4_nst;
extern int g_opt_g4_wgs;
extern int g_opt_gemm_ws;
extern int g_opt_gemm_ws128;
extern int g_opt_gemm_rect;
extern int g_opt_gemm_ws128_min_k;
extern int g_opt_attn_fwd_wg;
extern int g_opt_attn_fwd_head;
extern int g_opt_attn_bwd_head;
extern int g_opt_attn_keep_bits;
extern int g_opt_attn_op_keep_bits;
extern int g_opt_attn_fused;
extern int g_opt_attn_small;

enum { P5_G4_128x128 = 0, P5_G4_256x128 = 1, P5_G4_128x256 = 2, P5_G5_256x128 = 3, P5_G5_128x128 = 4 };
// ---- launchers ----
int p5l_gemm_bf16(P5GemmArgs g, hipStream_t s);
int p5l_gemm_f32(P5GemmArgs g, hipStream_t s);
bool p5l_gemm_gate_ok(int M, int N, int K, int lda, int ldb);
bool p5l_gemm_ce_ok(int M, int N, int K, int lda, int ldb);
bool p5l_gemm_normbwd_ok(int M, int N, int K, int lda, int ldb);
int launch_gemm4(int cfg, bool ks, P5GemmGroup& grp, hipStream_t s);
int p5l_attn_fwd(int bf16_mode, const P5AttnArgs& a, hipStream_t s);
int p5l_attn_bwd(int bf16_mode, const P5AttnArgs& a, hipStream_t s);
int p5l_attn_bwd_slots(int bf16_mode, int B, int Lq, int Lk);
bool p5l_attn_bwd_dot_ok(int bf16_mode, const P5AttnArgs& a);
template <class T> static inline int launch_gemm(const P5GemmArgs& g, hipStream_t s) { return sizeof(T) == 2 ? p5l_gemm_bf16(g, s) : p5l_gemm_f32(g, s); }
template <class T> static inline int launch_attn_fwd(const P5AttnArgs& a, hipStream_t s) { return p5l_attn_fwd(sizeof(T) == 2, a, s); }
template <class T> static inline int launch_attn_bwd(const P5AttnArgs& a, hipStream_t s) { return p5l_attn_bwd(sizeof(T) == 2, a, s); }
