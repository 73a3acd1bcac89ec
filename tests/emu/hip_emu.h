// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-OS-thread, fiber-based emulation of the HIP constructs the kernels under
// openp5_amd/csrc use (blocks of threads, __syncthreads, wave64 shuffles, 16x16 MFMA, the LDS
// transpose read, atomics).  It exists because the build container has no GPU: compiling the SAME
// kernel sources against this header lets the not-gpu test-suite exercise every kernel's index
// arithmetic, masking, softmax/backward algebra and the engine's orchestration on the host CPU.
//
// Adversarial modes (environment, read once per process; tests/test_emu_kernels.py::test_kernels_under_adversarial_emulation):
//   P5_EMU_POISON_LDS=1        every __shared__ array holds 0xFF bytes (NaN / -1) at the start of a launch
//   P5_EMU_FIBER_ORDER=reverse the threads of a workgroup are resumed last-to-first between barriers
//   P5_EMU_BLOCK_ORDER=reverse the workgroups of a launch run last-to-first (order of global atomics)
// What the emulation cannot show: timing of asynchronous direct-to-LDS copies (vmcnt bookkeeping), real wave interleavings.
//
// It is NOT a product path: the shipped library (libp5hip.so) is compiled by hipcc for gfx950 only
// and the Python package refuses to run without it.  Nothing under openp5_amd/ loads the emulator;
// only tests/ build and inject it.
//
// Fragment layouts encoded here follow /opt/skills/guides/cdna_hip_programming.md section 3
// (C/D: col = lane & 15, row = (lane >> 4) * 4 + reg; A/B: lane group g = lane >> 4 holds k = g*8..g*8+7
// for 16x16x32 bf16 and k = g for 16x16x4 f32) and section 2 (ds_read_b64_tr_b16).  The GPU tests are
// what validates those assumptions against real hardware.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define P5_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// LDS: function-local statics, all collected in one section so that the emulator can POISON them between workgroups
// (P5_EMU_POISON_LDS=1): on the device a workgroup finds whatever the previous workgroup on that CU left in LDS; here every array
// would otherwise keep its own previous contents (deterministic, plausible values).  0xFF bytes are NaN in bf16 / fp32 and -1 as
// integers, so a kernel that reads LDS it has not written shows up as NaN in its output.
#define __shared__ static __attribute__((section("p5_lds")))
extern "C" char __start_p5_lds[] __attribute__((weak));
extern "C" char __stop_p5_lds[] __attribute__((weak));
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }

namespace emu {

extern "C" void emu_switch(void** from_sp, void* to_sp);

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  unsigned tid = 0;
  bool done = false;
};

struct Wave {
  alignas(16) unsigned char slot[64][64];
  unsigned arrived = 0, gen = 0, live = 64;
};

struct BlockState {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  unsigned nthreads = 0, live = 0, bar_arrived = 0, bar_gen = 0;
  void* sched_sp = nullptr;
  Fiber* cur = nullptr;
  const std::function<void()>* body = nullptr;
  dim3 bdim, gdim, bidx;
  char* dyn_smem = nullptr;
};

inline BlockState& B() {
  static BlockState b;
  return b;
}

inline constexpr size_t kStack = 256 * 1024;

inline void yield() {
  BlockState& b = B();
  emu_switch(&b.cur->sp, b.sched_sp);
}

inline void fiber_main() {
  BlockState& b = B();
  (*b.body)();
  Fiber* f = b.cur;
  f->done = true;
  b.live--;
  Wave& w = b.waves[f->tid / 64];
  w.live--;
  // release barriers that were only waiting for this thread
  if (b.live > 0 && b.bar_arrived == b.live) { b.bar_arrived = 0; b.bar_gen++; }
  if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
  emu_switch(&f->sp, b.sched_sp);
  abort();
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  BlockState& b = B();
  unsigned nt = block.x * block.y * block.z;
  static std::vector<char*> stacks;
  while (stacks.size() < nt) stacks.push_back((char*)aligned_alloc(64, kStack));
  std::vector<char> dyn(shmem + 64);
  // P5_EMU_BLOCK_ORDER=reverse: workgroups run last-to-first.  Any order is a valid execution of kernels whose workgroups only meet in
  // atomics; it changes the order fp32 atomics land in -- the one thing that differs between two runs on the device.
  static const bool reverse = getenv("P5_EMU_BLOCK_ORDER") && !strcmp(getenv("P5_EMU_BLOCK_ORDER"), "reverse");
  const unsigned long long nblk = (unsigned long long)grid.x * grid.y * grid.z;
  // (once per launch, not per workgroup -- the section holds the LDS of every kernel instantiation, ~12 MB: the launch's first
  //  workgroup then runs on poisoned LDS, the later ones on what their predecessor left, like on the device)
  static const bool poison_lds = getenv("P5_EMU_POISON_LDS") && atoi(getenv("P5_EMU_POISON_LDS")) != 0;
  if (poison_lds && __start_p5_lds) memset(__start_p5_lds, 0xFF, (size_t)(__stop_p5_lds - __start_p5_lds));
  for (unsigned long long lin = 0; lin < nblk; ++lin) {
        const unsigned long long id = reverse ? nblk - 1 - lin : lin;
        const unsigned bx = (unsigned)(id % grid.x), by = (unsigned)((id / grid.x) % grid.y), bz = (unsigned)(id / ((unsigned long long)grid.x * grid.y));
        b.fibers.assign(nt, Fiber());
        b.waves.assign((nt + 63) / 64, Wave());
        b.nthreads = b.live = nt;
        b.bar_arrived = 0;
        b.bar_gen = 0;
        b.body = &body;
        b.bdim = block;
        b.gdim = grid;
        b.bidx = dim3(bx, by, bz);
        b.dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
        for (unsigned w = 0; w < b.waves.size(); ++w) {
          unsigned lanes = nt - w * 64;
          b.waves[w].live = lanes > 64 ? 64 : lanes;
        }
        for (unsigned t = 0; t < nt; ++t) {
          Fiber& f = b.fibers[t];
          f.tid = t;
          f.stack = stacks[t];
          uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
          void** sp = (void**)top;
          *(--sp) = nullptr;               // fake return address of fiber_main
          *(--sp) = (void*)&fiber_main;    // `ret` target of the first switch
          for (int r = 0; r < 6; ++r) *(--sp) = nullptr;
          f.sp = sp;
        }
        // P5_EMU_FIBER_ORDER=reverse: the threads of a workgroup are resumed last-to-first.  Between two barriers any order is a valid
        // execution; a write that a LATER thread's read depends on without a barrier in between (a missing __syncthreads, hidden by
        // the first-to-last order) then shows up as a wrong result.
        static const bool frev = getenv("P5_EMU_FIBER_ORDER") && !strcmp(getenv("P5_EMU_FIBER_ORDER"), "reverse");
        unsigned remaining = nt;
        while (remaining) {
          remaining = 0;
          for (unsigned tt = 0; tt < nt; ++tt) {
            const unsigned t = frev ? nt - 1 - tt : tt;
            Fiber& f = b.fibers[t];
            if (f.done) continue;
            b.cur = &f;
            emu_switch(&b.sched_sp, f.sp);
            if (!f.done) remaining++;
          }
        }
      }
}

inline unsigned tid_linear() { return B().cur->tid; }

inline void syncthreads() {
  BlockState& b = B();
  unsigned g = b.bar_gen;
  if (++b.bar_arrived == b.live) {
    b.bar_arrived = 0;
    b.bar_gen++;
    return;
  }
  while (b.bar_gen == g) yield();
}

inline void wave_barrier() {
  BlockState& b = B();
  Wave& w = b.waves[b.cur->tid / 64];
  unsigned g = w.gen;
  if (++w.arrived == w.live) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == g) yield();
}

inline Wave& wave() { return B().waves[B().cur->tid / 64]; }
inline unsigned lane() { return B().cur->tid & 63; }

template <class T>
inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 64, "");
  Wave& w = wave();
  memcpy(w.slot[lane()], &v, sizeof(T));
  wave_barrier();
  T r;
  memcpy(&r, w.slot[src & 63], sizeof(T));
  wave_barrier();
  return r;
}

}  // namespace emu

struct EmuIdx {
  struct X {
    operator unsigned() const {
      auto& b = emu::B();
      return b.cur->tid % b.bdim.x;
    }
  } x;
  struct Y {
    operator unsigned() const {
      auto& b = emu::B();
      return (b.cur->tid / b.bdim.x) % b.bdim.y;
    }
  } y;
  struct Z {
    operator unsigned() const {
      auto& b = emu::B();
      return b.cur->tid / (b.bdim.x * b.bdim.y);
    }
  } z;
};
struct EmuBlockIdx {
  struct X { operator unsigned() const { return emu::B().bidx.x; } } x;
  struct Y { operator unsigned() const { return emu::B().bidx.y; } } y;
  struct Z { operator unsigned() const { return emu::B().bidx.z; } } z;
};
struct EmuBlockDim {
  struct X { operator unsigned() const { return emu::B().bdim.x; } } x;
  struct Y { operator unsigned() const { return emu::B().bdim.y; } } y;
  struct Z { operator unsigned() const { return emu::B().bdim.z; } } z;
};
struct EmuGridDim {
  struct X { operator unsigned() const { return emu::B().gdim.x; } } x;
  struct Y { operator unsigned() const { return emu::B().gdim.y; } } y;
  struct Z { operator unsigned() const { return emu::B().gdim.z; } } z;
};
static EmuIdx threadIdx;
static EmuBlockIdx blockIdx;
static EmuBlockDim blockDim;
static EmuGridDim gridDim;

static inline void __syncthreads() { emu::syncthreads(); }
template <class T> static inline T __shfl_xor(T v, int m) { return emu::shfl_idx(v, (int)emu::lane() ^ m); }
template <class T> static inline T __shfl(T v, int src) { return emu::shfl_idx(v, src); }
// 64-bit lane mask of a predicate (full waves only: the slots of lanes that have exited are not cleared)
static inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::wave();
  w.slot[emu::lane()][0] = pred ? 1 : 0;
  emu::wave_barrier();
  unsigned long long r = 0;
  for (int i = 0; i < 64; ++i) r |= (unsigned long long)(w.slot[i][0] & 1) << i;
  emu::wave_barrier();
  return r;
}
template <class T> static inline T __shfl_down(T v, int d) {
  int s = (int)emu::lane() + d;
  return emu::shfl_idx(v, s > 63 ? (int)emu::lane() : s);
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
#define __expf expf
static inline void __threadfence() {}
#define __logf logf
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
