// Global-load pattern probe (development tool): how long does the operand fetch of the attention forward take, as a function of
// the layout?  grid (2, B*H), 256 threads: every workgroup fetches its 64 Q rows and all 128 K and V rows of its (batch, head),
// 16 bytes per lane, 128-byte rows -- exactly p5_attn_fwd_kernel's address stream at L = 128 -- and reduces them to one word.
//   layout 0: rows of the fused projection output [B*L, 3*H*64] (row stride 3*H*128 bytes; Q | K | V column blocks)   <- today
//   layout 1: head-major [3][B, H, L, 64] (a (batch, head)'s rows are contiguous)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void load_probe_kernel(unsigned* out, const char* buf, int B, int H, int L, int layout) {
  const int tid = threadIdx.x, b = blockIdx.y / H, h = blockIdx.y % H;
  const size_t rs = (size_t)3 * H * 128;                      // row stride of the fused layout
  unsigned acc = 0;
  u32x4 v[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) {                              // 2 pieces of Q (64 rows x 8), 4 of K, 4 of V per thread
    const int which = i < 2 ? 0 : (i < 6 ? 1 : 2);
    const int p = tid + (i < 2 ? i : (i < 6 ? i - 2 : i - 6)) * 256;
    const int row = (which == 0 ? blockIdx.x * 64 : 0) + (p >> 3), pc = p & 7;
    const char* a = layout == 0 ? buf + ((size_t)b * L + row) * rs + (size_t)which * H * 128 + h * 128 + pc * 16
                                : buf + (((size_t)which * B * H + (size_t)b * H + h) * L + row) * 128 + pc * 16;
    v[i] = *(const u32x4*)a;
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) acc += v[i][0] ^ v[i][3];
  if (acc == 0x12345678u) out[blockIdx.y] = acc;
}
extern "C" int load_probe(unsigned* out, const void* buf, int B, int H, int L, int layout, void* stream) {
  hipLaunchKernelGGL(load_probe_kernel, dim3(2, B * H), dim3(256), 0, (hipStream_t)stream, out, (const char*)buf, B, H, L, layout);
  return (int)hipGetLastError();
}
