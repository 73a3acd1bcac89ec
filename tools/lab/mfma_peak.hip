// mfma_peak.hip -- what the matrix cores deliver on THIS box, with nothing else in the way: register-resident operands, back-to-back
// MFMAs on independent accumulators.  Answers the round-3 question behind every GEMM roofline statement in DESIGN.md: is the
// 1.35-1.77 PFLOP/s "MFMA-only" ceiling of the ablated GEMM kernels the instruction shape (16x16x32 vs 32x32x16), single-wave
// issue (one MFMA wave per SIMD) or the clock the part sustains under load (zero vs random operands)?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o tools/lab/build/mfma_peak && tools/lab/build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 independent 16x16 accumulators (a 64x64 wave tile: 4 A fragments x 4 B fragments), the shape of the library's inner loops
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(float* out, const bf16x8_t* in, int iters) {
  const int l = threadIdx.x;
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(l * 8 + i) & 4095]; b[i] = in[(l * 8 + 4 + i) & 4095]; }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int it = iters;       // (do-while: a for loop's zero-trip guard makes hipcc rotate the accumulators through copies every iteration)
  do {
#pragma unroll
    for (int i = 0; i < 16; ++i)      // (inline asm, accumulating in place: hipcc's own allocation copies every second accumulator through a[0:3])
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
  } while (--it > 0);
  f32x4 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * blockDim.x + l] = s[0];
}
// 4 independent 32x32 accumulators (the same 64x64 wave tile: 2 x 2 fragments)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float* out, const bf16x8_t* in, int iters) {
  const int l = threadIdx.x;
  bf16x8_t a[2], b[2];
  for (int i = 0; i < 2; ++i) { a[i] = in[(l * 4 + i) & 4095]; b[i] = in[(l * 4 + 2 + i) & 4095]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  int it = iters;
  do {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 1]), "v"(b[i >> 1]));
  } while (--it > 0);
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + l] = s;
}

template <class F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float* out; bf16x8_t* in;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&in, 4096 * 16);
  std::vector<unsigned short> h(4096 * 8);
  const int iters = 4096, cus = 256;
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (auto& v : h) v = data ? (unsigned short)(0x3C00 + (rand() & 0x3FF) + ((rand() & 1) << 15)) : 0;     // +-[0.0078, 0.031) or zeros
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
#define RUN(NAME, KERN, WAVES, WGS_PER_CU, FLOP_PER_WAVE_ITER)                                                              \
    {                                                                                                                        \
      const double ms = time_ms([&] { hipLaunchKernelGGL(KERN<WAVES>, dim3(cus * WGS_PER_CU), dim3(WAVES * 64), 0, 0, out, in, iters); }, 5); \
      const double flop = (double)cus * WGS_PER_CU * WAVES * iters * FLOP_PER_WAVE_ITER;                                     \
      printf("%-34s %s operands: %8.3f ms  %7.1f TFLOP/s  (%.2f cycles/MFMA/SIMD at 2.4 GHz)\n", NAME, data ? "random" : "zero  ", ms,    \
             flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * (FLOP_PER_WAVE_ITER / (FLOP_PER_WAVE_ITER == 16 * 16384.0 ? 16384.0 : 32768.0)) * (WAVES * WGS_PER_CU / 4.0))); \
    }
    RUN("16x16x32, 1 wave/SIMD", k16, 4, 1, 16 * 16384.0);
    RUN("16x16x32, 2 waves/SIMD", k16, 8, 1, 16 * 16384.0);
    RUN("16x16x32, 4 waves/SIMD", k16, 8, 2, 16 * 16384.0);
    RUN("32x32x16, 1 wave/SIMD", k32, 4, 1, 4 * 32768.0);
    RUN("32x32x16, 2 waves/SIMD", k32, 8, 1, 4 * 32768.0);
    RUN("32x32x16, 4 waves/SIMD", k32, 8, 2, 4 * 32768.0);
  }
  return 0;
}
