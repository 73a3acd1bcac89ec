// p5_device.h -- device-side building blocks shared by every kernel file (gfx950 / CDNA4).
//
// Wave = 64 lanes.  All matrix work goes through ONE fragment convention so that the same kernel
// template serves both precisions of the engine:
//   * T = bf16 : v_mfma_f32_16x16x32_bf16   (fast mode; fp32 accumulate)
//   * T = float: v_mfma_f32_16x16x4_f32 x4  (parity mode; exact fp32 FMA chain, MI355X guide section 3)
// A "fragment" is always 16 bytes of reduction-dim-contiguous data per lane; one `mma16` call consumes a
// 64-byte K-chunk (32 bf16 or 16 f32) of a 16-row A tile and a 16-row B^T tile:
//   lane l supplies A[row = l & 15][chunk_k = (l >> 4) * (16 / sizeof(T)) + 0..]  and
//                   B[col = l & 15][same k range],
//   result tile C[row = (l >> 4) * 4 + r][col = l & 15], r = 0..3   (guide section 3, C/D layout).
// For f32 the four MFMAs take element i of the lane's float4, i.e. lane group g contributes
// k = 4g + i to MFMA i -- a k-permutation applied identically to A and B, which leaves the sum unchanged.
#pragma once
#include <stdint.h>

#ifndef P5_EMU
#include <hip/hip_runtime.h>
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct bf16 {
  unsigned short v;
};

// ---- scalar conversions (round-to-nearest-even, written out so host emulation and device agree) ----
__host__ __device__ static __forceinline__ float bf2f(bf16 x) {
  union { unsigned u; float f; } c;
  c.u = ((unsigned)x.v) << 16;
  return c.f;
}
__host__ __device__ static __forceinline__ bf16 f2bf(float f) {
  union { unsigned u; float f; } c;
  c.f = f;
  unsigned u = c.u;
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) {  // NaN stays NaN
    bf16 r; r.v = (unsigned short)((u >> 16) | 0x40); return r;
  }
  u += 0x7FFFu + ((u >> 16) & 1u);
  bf16 r;
  r.v = (unsigned short)(u >> 16);
  return r;
}
// gfx950 converts two fp32 to packed bf16 (round-to-nearest-even, NaN-preserving) in ONE instruction; the integer sequence above
// is ~10 VALU operations per element plus a NaN test that hipcc turns into a branch inside control flow -- every bf16 epilogue
// (GEMM tiles, norms, attention outputs) converts tens of values per lane.  The host emulation keeps the integer sequence; the
// two agree bit for bit on every finite value.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P5_EMU)
__device__ static __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
#define P5_HW_BF16 1
#endif
template <class T> __host__ __device__ static __forceinline__ float to_f(T x);
template <> __host__ __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __host__ __device__ __forceinline__ float to_f<bf16>(bf16 x) { return bf2f(x); }
template <class T> __host__ __device__ static __forceinline__ T from_f(float x);
template <> __host__ __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __host__ __device__ __forceinline__ bf16 from_f<bf16>(float x) {
#ifdef P5_HW_BF16
  bf16 r;
  r.v = (unsigned short)(cvt_pk_bf16(x, x) & 0xFFFFu);
  return r;
#else
  return f2bf(x);
#endif
}

template <class T> struct TT;
template <> struct TT<float> { static constexpr int EPF = 4; static constexpr int KCH = 16; static constexpr int DT = 0; };
template <> struct TT<bf16> { static constexpr int EPF = 8; static constexpr int KCH = 32; static constexpr int DT = 1; };
// EPF = elements per 16-byte fragment, KCH = elements of K consumed per mma16 (64 bytes).

#define P5_NEG_INF (-__builtin_huge_valf())
#define P5_DT_F32 0
#define P5_DT_BF16 1

// ---- launch + block primitives -------------------------------------------------------------------
#ifdef P5_EMU
#define P5_LAUNCH(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define P5_DYN_SMEM(name) char* name = emu::B().dyn_smem
#define P5_LANE() ((int)emu::lane())
#else
// In-run kernel profiler (p5_profile_begin / p5_profile_end, include/p5hip.h): while it is on, every launch of the library is bracketed
// by two HIP events on the launch stream; the report aggregates per (kernel, grid).  bench.py uses it to name the dominant kernel of
// the step and its in-step duration in the very run the driver times (measurement aid; off by default, costs nothing then).
#include <vector>
#include <string>
struct P5Prof {
  struct Rec { const char* name; const char* tag; unsigned gx, gy, gz, bx; int m, n, k; double flops; hipEvent_t a, b; };
  int on = 0;
  double pending_flops = 0.0;       // set by a launcher right before its P5_LAUNCH (algorithmic FLOPs of that launch)
  const char* pending_tag = "";     // ... and what the template parameters of the stringified kernel name stand for
  int pm = 0, pn = 0, pk = 0;       // ... and the problem shape of a single-problem GEMM launch (0 = not given)
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
  }
  void begin(const char* name, dim3 g, dim3 b, hipStream_t s) {
    Rec r{name, pending_tag, g.x, g.y, g.z, b.x, pm, pn, pk, pending_flops, get(), get()};
    pending_flops = 0.0; pending_tag = ""; pm = pn = pk = 0;
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
  }
  void end(hipStream_t s) { (void)hipEventRecord(recs.back().b, s); }
};
// one instance per host thread, shared by all translation units of the library: a launcher notes the FLOPs / shape of the launch it is
// about to make, P5_LAUNCH consumes the note -- both on the calling thread, so two threads driving two engines do not see each other's
inline P5Prof& p5_prof() { static thread_local P5Prof p; return p; }
#define P5_LAUNCH(kern, grid, block, shmem, stream, ...)                                   \
  do {                                                                                     \
    P5Prof& _pf = p5_prof();                                                               \
    if (_pf.on) _pf.begin(#kern, dim3(grid), dim3(block), (stream));                       \
    else { _pf.pending_flops = 0.0; _pf.pending_tag = ""; _pf.pm = _pf.pn = _pf.pk = 0; } \
    hipLaunchKernelGGL(kern, (grid), (block), (shmem), (stream), __VA_ARGS__);             \
    if (_pf.on) _pf.end((stream));                                                         \
  } while (0)
#define P5_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define P5_LANE() ((int)(threadIdx.x & 63))
#endif

// ---- MFMA -------------------------------------------------------------------------------------------
#ifdef P5_EMU
template <class T>
static inline void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
  emu::Wave& w = emu::wave();
  const int l = (int)emu::lane();
  memcpy(w.slot[l], &a, 16);
  memcpy(w.slot[l] + 16, &b, 16);
  emu::wave_barrier();
  const int col = l & 15, rg = l >> 4;
  constexpr int E = TT<T>::EPF;
  for (int r = 0; r < 4; ++r) {
    const int row = rg * 4 + r;
    float s = acc[r];
    for (int g = 0; g < 4; ++g) {
      const T* pa = (const T*)(w.slot[g * 16 + row]);
      const T* pb = (const T*)(w.slot[g * 16 + col] + 16);
      for (int j = 0; j < E; ++j) s += to_f<T>(pa[j]) * to_f<T>(pb[j]);
    }
    acc[r] = s;
  }
  emu::wave_barrier();
}
#else
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <class T> __device__ static __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16<bf16>(f32x4& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
  f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
}
#endif

// ---- LDS transpose read (gfx950 ds_read_b64_tr_b16) ---------------------------------------------------
// Within each 16-lane group the 16 lanes' 8-byte rows form a [4][16] bf16 block (lane i owns row i/4,
// columns (i%4)*4..+3); lane i receives column i, i.e. 4 values that sit in 4 different rows.
#ifdef P5_EMU
static inline u32x2 lds_tr16_b64(const void* p) {
  emu::Wave& w = emu::wave();
  const int l = (int)emu::lane();
  memcpy(w.slot[l], p, 8);
  emu::wave_barrier();
  const int g = l & ~15, i = l & 15;
  unsigned short v[4];
  for (int j = 0; j < 4; ++j) {
    const unsigned short* row = (const unsigned short*)w.slot[g + j * 4 + (i >> 2)];
    v[j] = row[i & 3];
  }
  emu::wave_barrier();
  u32x2 r;
  r[0] = (unsigned)v[0] | ((unsigned)v[1] << 16);
  r[1] = (unsigned)v[2] | ((unsigned)v[3] << 16);
  return r;
}
#else
__device__ static __forceinline__ u32x2 lds_tr16_b64(const void* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(u32x2, r);
}
#endif

// ---- direct global -> LDS copy (gfx950 global_load_lds_dwordx4): lane l's 16 bytes land at lds_wave_base + 16*l.
// The LDS base must be wave-uniform (it travels in M0); completion is tracked by vmcnt, and __syncthreads()
// drains it before the barrier (cdna_hip_programming.md section 5).
#ifdef P5_EMU
static inline void glds16(const void* g, char* lds_wave_base) { memcpy(lds_wave_base + 16 * (int)emu::lane(), g, 16); }
#else
__device__ static __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#endif

// "raw" direct-to-LDS copy: the same instruction issued through inline asm.  hipcc orders every LDS read behind ALL
// direct-to-LDS copies it knows to be in flight (s_waitcnt vmcnt(0) before the read: it cannot prove that the read does
// not alias a copy), which serialises a multi-stage ring.  A raw copy is invisible to that logic; the caller owns the
// ordering: P5_WAIT_VM(n) + P5_BARRIER_LDS() before anybody reads the copied bytes.  (M0 is not otherwise used by the
// kernels that issue raw copies.)
#ifdef P5_EMU
#define glds16_raw glds16
#else
__device__ static __forceinline__ void glds16_raw(const void* g, char* lds_wave_base) {
  const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)lds_wave_base));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory");
}
#endif

// the same copy addressed as SCALAR base + 32-bit per-lane byte offset (the `saddr` form of the instruction): a kernel that issues 16
// copies per K-step keeps ONE offset register per operand instead of sixteen 64-bit pointers
#ifdef P5_EMU
static inline void glds16_raw_s(uint64_t sbase, uint32_t voff, char* lds_wave_base) {
  memcpy(lds_wave_base + 16 * (int)emu::lane(), (const char*)(uintptr_t)sbase + voff, 16);
}
#else
__device__ static __forceinline__ void glds16_raw_s(uint64_t sbase, uint32_t voff, char* lds_wave_base) {    // sbase: wave-uniform
  const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)lds_wave_base));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m) : "memory");
}
#endif

// 16-byte global load the compiler does NOT track (inline asm): its completion is the caller's business -- P5_WAIT_VM(n) with n =
// number of vector-memory operations issued after it, then P5_SCHED_FENCE() before the first use.  This is the only way to
// keep loads in flight across a loop back edge: hipcc's own waitcnt insertion waits for everything older than the current
// block's loads there (cdna_hip_programming.md 5.7 form iii).  The destination must not be touched between issue and wait.
#ifdef P5_EMU
static inline void gload16_raw(u32x4& dst, const void* p) { dst = *(const u32x4*)p; }
#else
__device__ static __forceinline__ void gload16_raw(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
#endif

// compiler scheduling fence: nothing is moved across it (used to keep an end-of-step barrier BELOW the MFMAs it follows
// in program order -- hipcc otherwise hoists "s_waitcnt vmcnt(0); s_barrier" above them and serialises copy and math)
#ifdef P5_EMU
#define P5_SCHED_FENCE() ((void)0)
#else
#define P5_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// lanes of ONE wave exchanging data through LDS: hardware executes a wave's LDS operations in order, the host emulator
// runs lanes one after the other and needs an explicit rendezvous
#ifdef P5_EMU
#define P5_WAVE_SYNC() emu::wave_barrier()
#else
#define P5_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// ---- hand-placed waits / barrier / scheduler hints for software-pipelined main loops -----------------------
// P5_WAIT_VM(n): wait until at most n of this wave's vector-memory ops (incl. direct-to-LDS copies) are outstanding;
// P5_BARRIER_LDS(): workgroup barrier that orders LDS traffic only (no vmcnt(0) like __syncthreads());
// P5_SCHED_GROUP(mask, n): next n instructions of class mask (0x008 MFMA, 0x020 VMEM read, 0x100 DS read) in the schedule.
#ifdef P5_EMU
#define P5_WAIT_VM(n) ((void)0)
#define P5_WAIT_LGKM0() ((void)0)
#define P5_BARRIER_LDS() __syncthreads()
#define P5_SCHED_GROUP(mask, n) ((void)0)
#else
#define P5_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define P5_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P5_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define P5_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

// ---- exponential of the mode: exact expf in the fp32 parity mode, the exp2-based 2-ulp __expf in the bf16 mode (the precise
// one is ~30 instructions; softmax / log-sum-exp kernels call it tens of times per lane) ----
template <class T> __device__ static __forceinline__ float p5_exp(float x);
template <> __device__ __forceinline__ float p5_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float p5_exp<bf16>(float x) { return __expf(x); }

// 2^x on the transcendental unit (v_exp_f32): the long-sequence attention backward folds log2(e) into the terms it adds to the score, so
// a probability costs one fma + one v_exp_f32 instead of add, sub, mul, v_exp_f32
#define P5_LOG2E 1.4426950408889634f
__device__ static __forceinline__ float p5_exp2(float x) {
#ifdef P5_EMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}

// ---- wave reductions (all 64 lanes) -----------------------------------------------------------------
__device__ static __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ static __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
// reduce across the 16 lanes that share (lane >> 4): one accumulator ROW of a 16x16 C tile
__device__ static __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ static __forceinline__ float row16_max(float v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// ---- 16-byte vector load/store helpers --------------------------------------------------------------
__device__ static __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }
__device__ static __forceinline__ void st16(void* p, u32x4 v) { *(u32x4*)p = v; }
__device__ static __forceinline__ u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

template <class T> __device__ static __forceinline__ void unpack16(const u32x4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { union { unsigned u; float f; } c; c.u = v[i]; out[i] = c.f; }
}
template <> __device__ __forceinline__ void unpack16<bf16>(const u32x4& v, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    union { unsigned u; float f; } lo, hi;
    lo.u = v[i] << 16; hi.u = v[i] & 0xFFFF0000u;
    out[2 * i] = lo.f; out[2 * i + 1] = hi.f;
  }
}
// four consecutive T elements (8 bytes of bf16 / 16 bytes of fp32) from fp32, one store
template <class T> __device__ static __forceinline__ void st4(void* p, const float* in);
template <> __device__ __forceinline__ void st4<float>(void* p, const float* in) {
  *(f32x4*)p = (f32x4){in[0], in[1], in[2], in[3]};
}
template <> __device__ __forceinline__ void st4<bf16>(void* p, const float* in) {
  u32x2 v;
#ifdef P5_HW_BF16
  v[0] = cvt_pk_bf16(in[0], in[1]);
  v[1] = cvt_pk_bf16(in[2], in[3]);
#else
  v[0] = (unsigned)f2bf(in[0]).v | ((unsigned)f2bf(in[1]).v << 16);
  v[1] = (unsigned)f2bf(in[2]).v | ((unsigned)f2bf(in[3]).v << 16);
#endif
  *(u32x2*)p = v;
}
template <class T> __device__ static __forceinline__ u32x4 pack16(const float* in);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) { union { unsigned u; float f; } c; c.f = in[i]; v[i] = c.u; }
  return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16>(const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#ifdef P5_HW_BF16
    v[i] = cvt_pk_bf16(in[2 * i], in[2 * i + 1]);
#else
    v[i] = (unsigned)f2bf(in[2 * i]).v | ((unsigned)f2bf(in[2 * i + 1]).v << 16);
#endif
  }
  return v;
}
