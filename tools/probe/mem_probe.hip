// Memory-system probe (development tool): what does a plain streaming kernel reach on this part, and what do fp32 global atomics cost?
//   read:   16 bytes per lane, UNROLL loads in flight, grid-stride over `bytes`
//   copy:   read + 16-byte store
//   atomic: one fp32 atomicAdd per lane per iteration into `span` consecutive floats (span = n: no address shared inside a pass;
//           small span: the contention of an embedding row that many tokens hit)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool COPY>
__global__ __launch_bounds__(256) void stream_kernel(unsigned* out, const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = i + u * stride < n16 ? src[i + u * stride] : (u32x4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (COPY) { if (i + u * stride < n16) dst[i + u * stride] = v[u]; }
      else acc += v[u][0] ^ v[u][3];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void atomic_kernel(float* __restrict__ dst, size_t n, size_t span, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const size_t i = (t + (size_t)it * gridDim.x * 256) % n;
    atomicAdd(dst + (i % span), 1.0f);
  }
}
extern "C" int mem_stream(unsigned* out, const void* src, void* dst, size_t bytes, int wgs, int unroll, int copy, void* stream) {
  const size_t n16 = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (copy) hipLaunchKernelGGL((stream_kernel<4, true>), dim3(wgs), dim3(256), 0, s, out, (const u32x4*)src, (u32x4*)dst, n16);
  else if (unroll == 8) hipLaunchKernelGGL((stream_kernel<8, false>), dim3(wgs), dim3(256), 0, s, out, (const u32x4*)src, (u32x4*)dst, n16);
  else hipLaunchKernelGGL((stream_kernel<4, false>), dim3(wgs), dim3(256), 0, s, out, (const u32x4*)src, (u32x4*)dst, n16);
  return (int)hipGetLastError();
}
extern "C" int mem_atomic(float* dst, size_t n, size_t span, int wgs, int iters, void* stream) {
  hipLaunchKernelGGL(atomic_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, dst, n, span, iters);
  return (int)hipGetLastError();
}
