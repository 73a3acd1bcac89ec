// grid_barrier.hip -- what a device-side grid barrier costs on this part next to a dependent kernel boundary (round 5).
// The decode step is a chain of ~46 dependent launches; the alternative asked for since round 1 is ONE persistent kernel per decoder layer
// whose phases are separated by a grid barrier.  This probe measures both seams with the same tiny phase body:
//   A. N dependent launches of a 256-workgroup kernel replayed from a hipGraph            -> us per kernel boundary
//   B. ONE 256-workgroup launch (one per CU) running N phases separated by a barrier       -> us per barrier, two constructions:
//        flat:  one monotonic counter; lane 0 release-fences, arrives, polls with relaxed agent loads + s_sleep, acquire-fences
//        xcd:   per-XCD counters (workgroup b runs on XCD b % 8), the last arriver of an XCD release-fences and arrives at the top
//               counter, the XCD leader publishes the generation to its XCD; every workgroup acquire-fences
// Phase body: every workgroup writes 1 KiB and reads the 1 KiB its left neighbour wrote in the previous phase (checked: a stale read
// counts as an error), i.e. a real cross-CU dependence.  Every spin is bounded; a timeout is reported, never waited out.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier.hip -o tools/probe/grid_barrier.bin && tools/probe/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static constexpr int WG = 256, NWG = 256, WORDS = 256;      // 1 KiB per workgroup per phase

__device__ static inline void phase_body(unsigned* buf, int phase, int wg, int nwg, unsigned* errors) {
  const unsigned* prev = buf + ((size_t)((phase + 1) & 1) * nwg + (wg + nwg - 1) % nwg) * WORDS;
  unsigned* mine = buf + ((size_t)(phase & 1) * nwg + wg) * WORDS;
  const unsigned got = phase > 0 ? __hip_atomic_load(prev + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (phase > 0 && got != (unsigned)(phase - 1) * 1000u + (unsigned)((wg + nwg - 1) % nwg)) atomicAdd(errors, 1u);
  mine[threadIdx.x] = (unsigned)phase * 1000u + (unsigned)wg;
}
__global__ __launch_bounds__(WG) void phase_kernel(unsigned* buf, int phase, unsigned* errors) { phase_body(buf, phase, blockIdx.x, gridDim.x, errors); }

__device__ static inline bool spin_until(unsigned* p, unsigned target, unsigned* timeouts) {
  for (int it = 0; it < (1 << 20); ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicAdd(timeouts, 1u);
  return false;
}
__global__ __launch_bounds__(WG) void persistent_flat(unsigned* buf, int nphase, unsigned* counter, unsigned* errors, unsigned* timeouts) {
  for (int ph = 0; ph < nphase; ++ph) {
    phase_body(buf, ph, blockIdx.x, gridDim.x, errors);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until(counter, (unsigned)(ph + 1) * gridDim.x, timeouts);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}
// xcd[0..7]: arrivals per XCD; top: arrivals of XCDs; gen[0..7]: generation published per XCD
__global__ __launch_bounds__(WG) void persistent_xcd(unsigned* buf, int nphase, unsigned* xcd, unsigned* top, unsigned* gen, unsigned* errors, unsigned* timeouts) {
  const int x = blockIdx.x & 7, per = gridDim.x >> 3;
  for (int ph = 0; ph < nphase; ++ph) {
    phase_body(buf, ph, blockIdx.x, gridDim.x, errors);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(xcd + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(ph + 1) * per - 1) {            // last arriver of this XCD
        __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(top, (unsigned)(ph + 1) * 8, timeouts);
        __hip_atomic_store(gen + x * 32, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        spin_until(gen + x * 32, (unsigned)(ph + 1), timeouts);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

int main() {
  const int N = 64, REPS = 20;
  unsigned *buf, *ctr;
  CHECK(hipMalloc(&buf, (size_t)2 * NWG * WORDS * 4));
  CHECK(hipMalloc(&ctr, 4096 * 4));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  auto reset = [&]() { CHECK(hipMemsetAsync(ctr, 0, 4096 * 4, s)); CHECK(hipMemsetAsync(buf, 0, (size_t)2 * NWG * WORDS * 4, s)); };
  auto report = [&](const char* what, float ms, int seams) {
    unsigned h[4096];
    CHECK(hipMemcpy(h, ctr, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-34s %8.2f us per run of %d phases  -> %6.2f us per seam   (stale reads %u, spin timeouts %u)\n", what, ms * 1e3 / REPS, N, ms * 1e3 / REPS / seams, h[1024], h[1025]);
  };
  // ---- A. dependent launches from a graph ----
  reset();
  hipGraph_t graph; hipGraphExec_t exec;
  for (int ph = 0; ph < N; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(WG), 0, s, buf, ph, ctr + 1024);      // (loads the code object)
  CHECK(hipStreamSynchronize(s));
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  for (int ph = 0; ph < N; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(WG), 0, s, buf, ph, ctr + 1024);
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  reset();
  for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  reset();
  CHECK(hipEventRecord(a, s));
  for (int r = 0; r < REPS; ++r) CHECK(hipGraphLaunch(exec, s));
  CHECK(hipEventRecord(b, s));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  report("A. 64 dependent launches (graph)", ms, N);
  // ---- B. persistent kernels ----
  for (int variant = 0; variant < 2; ++variant) {
    auto run = [&]() {
      CHECK(hipMemsetAsync(ctr, 0, 1024 * 4, s));      // counters (not the error words)
      if (variant == 0) hipLaunchKernelGGL(persistent_flat, dim3(NWG), dim3(WG), 0, s, buf, N, ctr, ctr + 1024, ctr + 1025);
      else hipLaunchKernelGGL(persistent_xcd, dim3(NWG), dim3(WG), 0, s, buf, N, ctr, ctr + 512, ctr + 600, ctr + 1024, ctr + 1025);
    };
    reset();
    for (int r = 0; r < 3; ++r) run();
    CHECK(hipStreamSynchronize(s));
    reset();
    CHECK(hipEventRecord(a, s));
    for (int r = 0; r < REPS; ++r) run();
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
    report(variant == 0 ? "B1. persistent, flat counter" : "B2. persistent, per-XCD counters", ms, N);
  }
  return 0;
}
