// LDS access-pattern probe (development tool, not part of the product library): how many cycles does a CU need per wave
// instruction for the LDS access patterns of p5_attn.h at a given tile row stride?  One workgroup of 1024 threads (16 waves, four
// per SIMD: the LDS pipe is saturated) repeats one pattern `iters` times; the host divides the kernel time by iters.
//   pattern 0: ds_read_b64_tr_b16 pairs as tile_frag_ks (transposed K-strided operand)
//   pattern 1: ds_read_b128 as tile_frag_kc (row = lane & 15, 16-byte slot = lane >> 4)
//   pattern 2: ds_write_b16 as wave_store_16x64 staging (row g*4+r, element dt*16+li)
//   pattern 3: ds_write_b64 as the transposed-score P store (row li, elements t*16+g*4..+3)
//   pattern 4: ds_read_u16 along a diagonal (row qr, element lane-15+qr) as the relative-bias diagonal sums
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void lds_probe_kernel(unsigned* out, int stride, int pattern, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  for (int k = tid; k < 65536 / 4; k += 1024) ((unsigned*)lds)[k] = k * 2654435761u;
  __syncthreads();
  char* base = lds + (wave & 3) * 16 * stride;       // each wave works on its own 16-row slice (as the kernels do) ... within 64 KiB
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (pattern == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char* p = lds + ((g * 8 + (i >> 2)) * stride + (dt * 16 + (i & 3) * 4) * 2 + (it & 1) * 32 * stride) % 60000;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        acc += (unsigned)lo[0] + (unsigned)hi[3];
      }
    } else if (pattern == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const u32x4 v = *(const u32x4*)(base + i * stride + (c & 1) * 64 + g * 16);
        acc += v[0] + v[3];
      }
    } else if (pattern == 2) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) *(unsigned short*)(base + (g * 4 + r) * stride + (dt * 16 + i) * 2) = (unsigned short)(acc + r);
      acc += 1;
    } else if (pattern == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        u32x2 v; v[0] = acc; v[1] = acc + t;
        *(u32x2*)(base + i * stride + (t * 16 + g * 4) * 2) = v;
      }
      acc += 1;
    } else {
#pragma unroll
      for (int qr = 0; qr < 16; ++qr) {
        const int col = lane - 15 + qr;
        acc += *(const unsigned short*)(base + qr * stride + (col < 0 ? 0 : col) * 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc == 0x12345678u) out[tid] = acc;
  if (tid == 0) out[1024] = acc;
}

extern "C" int lds_probe(unsigned* out, int stride, int pattern, int iters, void* stream) {
  hipLaunchKernelGGL(lds_probe_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, stride, pattern, iters);
  return (int)hipGetLastError();
}
