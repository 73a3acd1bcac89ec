// tr_wait_probe.hip -- does an MFMA behind a PARTIAL `s_waitcnt lgkmcnt(n)` ever read a ds_read_b64_tr_b16 destination before the data has landed?
// (development probe, not part of the product library; round-5 verdict item 8 / profiles/r05_call21_keep_mask_bisect.txt)
//
// The round-5 hardware-only failure: in one variant of the long-sequence attention forward the first P V MFMA of a block multiplied by the
// PREVIOUS contents of its operand registers (ballot words written by VALU just before) instead of the eight transposed LDS reads issued in
// front of it, behind `s_waitcnt lgkmcnt(2)`-style hand-counted waits.  This probe isolates that instruction pattern with physical registers:
//     [SENT]  v_mov sentinel (bf16 +inf pairs) into v[100:115]                      <- "VALU writes of the destination registers just before"
//             8 x ds_read_b64_tr_b16 v[100+2k : 101+2k]
//     [SLOAD] s_load_dwordx4 (SMEM shares lgkmcnt and returns out of order)          <- issued BEFORE the reads (SLOAD=1) or AFTER them (SLOAD=2)
//             s_waitcnt lgkmcnt(6) ; mfma(v[100:103]) ; lgkmcnt(4) ; mfma(v[104:107]) ; lgkmcnt(2) ; mfma(v[108:111]) ; lgkmcnt(0) ; mfma(v[112:115])
// against the same sequence behind one `s_waitcnt lgkmcnt(0)`.  B = bf16 ones, so every accumulator is a row sum of the LDS tile: a stale
// operand shows up as inf / NaN / a different sum.  All four waves of a workgroup run it on their own LDS slice while hammering the LDS pipe,
// `wgs` workgroups, `iters` rounds; out[0] = mismatching (lane, accumulator) pairs, out[1] = rounds run.
//   build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/tr_wait_probe.hip -o tools/probe/libtr_wait_probe.so
//   run  : python tools/probe/tr_wait_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

#define READS(o)                                                                                   \
  "ds_read_b64_tr_b16 v[100:101], v99 offset:" #o "+0\n ds_read_b64_tr_b16 v[102:103], v99 offset:" #o "+512\n"   \
  "ds_read_b64_tr_b16 v[104:105], v99 offset:" #o "+1024\n ds_read_b64_tr_b16 v[106:107], v99 offset:" #o "+1536\n" \
  "ds_read_b64_tr_b16 v[108:109], v99 offset:" #o "+2048\n ds_read_b64_tr_b16 v[110:111], v99 offset:" #o "+2560\n" \
  "ds_read_b64_tr_b16 v[112:113], v99 offset:" #o "+3072\n ds_read_b64_tr_b16 v[114:115], v99 offset:" #o "+3584\n"
#define SENTINEL                                                                                                          \
  "v_mov_b32 v100, v98\n v_mov_b32 v101, v98\n v_mov_b32 v102, v98\n v_mov_b32 v103, v98\n v_mov_b32 v104, v98\n v_mov_b32 v105, v98\n" \
  "v_mov_b32 v106, v98\n v_mov_b32 v107, v98\n v_mov_b32 v108, v98\n v_mov_b32 v109, v98\n v_mov_b32 v110, v98\n v_mov_b32 v111, v98\n" \
  "v_mov_b32 v112, v98\n v_mov_b32 v113, v98\n v_mov_b32 v114, v98\n v_mov_b32 v115, v98\n"
#define ZERO_ACC                                                                                                         \
  "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n" \
  "v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0\n v_mov_b32 v128, 0\n v_mov_b32 v129, 0\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0\n"
#define MFMA(acc, a) "v_mfma_f32_16x16x32_bf16 v[" #acc "], v[" #a "], v[92:95], v[" #acc "]\n"
#define CLOBBERS                                                                                                                        \
  "memory", "v92", "v93", "v94", "v95", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",  \
      "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",  \
      "s40", "s41", "s42", "s43"

// PARTIAL: hand-counted partial waits (1) or one full wait (0); SENT: VALU sentinel writes into the read destinations right before the reads;
// SLOAD: 0 none, 1 an s_load_dwordx4 outstanding from BEFORE the reads, 2 issued AFTER the reads (before the first wait)
template <int PARTIAL, int SENT, int SLOAD>
__device__ __forceinline__ void round_asm(unsigned lds_addr, const void* sptr, float (&r)[16]) {
  asm volatile(
      "v_mov_b32 v99, %16\n v_mov_b32 v98, 0x7f807f80\n v_mov_b32 v92, 0x3f803f80\n v_mov_b32 v93, 0x3f803f80\n v_mov_b32 v94, 0x3f803f80\n v_mov_b32 v95, 0x3f803f80\n" ZERO_ACC
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]),
        "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
      : "v"(lds_addr)
      : CLOBBERS);
  if (SLOAD == 1) asm volatile("s_load_dwordx4 s[40:43], %0, 0x0\n" ::"s"(sptr) : CLOBBERS);
  if (SENT) asm volatile(SENTINEL ::: CLOBBERS);
  asm volatile(READS(0)::: CLOBBERS);
  if (SLOAD == 2) asm volatile("s_load_dwordx4 s[40:43], %0, 0x0\n" ::"s"(sptr) : CLOBBERS);
  if (PARTIAL) {
    // (with an SMEM op in the same counter the hand count is one higher while it is outstanding -- exactly what a hand-counted kernel gets
    //  wrong when the compiler places an s_load next to its reads; SLOAD variants keep the ORIGINAL counts to show that case)
    asm volatile("s_waitcnt lgkmcnt(6)\n" MFMA(116:119, 100:103) "s_waitcnt lgkmcnt(4)\n" MFMA(120:123, 104:107) "s_waitcnt lgkmcnt(2)\n" MFMA(124:127, 108:111)
                 "s_waitcnt lgkmcnt(0)\n" MFMA(128:131, 112:115)::: CLOBBERS);
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)\n" MFMA(116:119, 100:103) MFMA(120:123, 104:107) MFMA(124:127, 108:111) MFMA(128:131, 112:115)::: CLOBBERS);
  }
  asm volatile("s_nop 15\n s_nop 15\n"
               "v_mov_b32 %0, v116\n v_mov_b32 %1, v117\n v_mov_b32 %2, v118\n v_mov_b32 %3, v119\n v_mov_b32 %4, v120\n v_mov_b32 %5, v121\n v_mov_b32 %6, v122\n v_mov_b32 %7, v123\n"
               "v_mov_b32 %8, v124\n v_mov_b32 %9, v125\n v_mov_b32 %10, v126\n v_mov_b32 %11, v127\n v_mov_b32 %12, v128\n v_mov_b32 %13, v129\n v_mov_b32 %14, v130\n v_mov_b32 %15, v131\n"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]),
                 "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
               :
               : CLOBBERS);
}

template <int PARTIAL, int SENT, int SLOAD>
__global__ __launch_bounds__(256) void tr_wait_probe_kernel(unsigned* out, const float* sptr, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4][8 * 256 + 64];      // per wave: 8 x (16 rows x 32 bytes) tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * (8 * 256 + 64); i += 256) (&lds[0][0])[i] = (unsigned short)(0x3c00 + ((i * 37) & 0xff));      // small positive bf16 values
  __syncthreads();
  // lane's 8-byte row piece of a [16 x 16] bf16 block per 16-lane group (the access shape of tile_frag_ks): rows of 32 bytes
  const unsigned addr = (unsigned)(uintptr_t)(&lds[wave][0]) + (unsigned)((lane & 15) * 32 + (lane >> 4) * 8);
  float ref[16], got[16];
  round_asm<0, 0, 0>(addr, sptr, ref);
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    round_asm<PARTIAL, SENT, SLOAD>(addr, sptr, got);
#pragma unroll
    for (int k = 0; k < 16; ++k) bad += (__float_as_uint(got[k]) != __float_as_uint(ref[k])) ? 1u : 0u;
    // keep the LDS pipe and the counters busy between rounds (other waves of the workgroup are inside their rounds meanwhile)
    lds[wave][8 * 256 + (lane & 63)] = (unsigned short)it;
  }
  if (bad) atomicAdd(out, bad);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (unsigned)iters;
}

extern "C" int tr_wait_probe(unsigned* out, const float* sptr, int variant, int wgs, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define CASE(v, P, S, L) case v: hipLaunchKernelGGL((tr_wait_probe_kernel<P, S, L>), dim3(wgs), dim3(256), 0, s, out, sptr, iters); break;
  switch (variant) {
    CASE(0, 0, 0, 0) CASE(1, 1, 0, 0) CASE(2, 1, 1, 0) CASE(3, 1, 1, 1) CASE(4, 1, 1, 2) CASE(5, 1, 0, 1) CASE(6, 1, 0, 2) CASE(7, 0, 1, 1)
    default: return -1;
  }
  return (int)hipGetLastError();
}
