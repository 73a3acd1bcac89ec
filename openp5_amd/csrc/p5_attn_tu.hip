// p5_attn_tu.hip -- translation unit of the attention family (p5_attn.h): kernel selection and launch (declarations in p5_host.h).
#include "p5_host.h"

int g_opt_attn_fwd_wg = getenv("P5_ATTN_FWD_WG") ? atoi(getenv("P5_ATTN_FWD_WG")) : 1;   // whole-(batch, head) attention forward (bf16, L <= 128)
int g_opt_attn_fwd_head = getenv("P5_ATTN_FWD_HEAD") ? atoi(getenv("P5_ATTN_FWD_HEAD")) : 1;   // whole-(batch, head) attention forward with K and V resident in LDS (bf16, 128 < Lk <= 512)
int g_opt_attn_bwd_head = getenv("P5_ATTN_BWD_HEAD") ? atoi(getenv("P5_ATTN_BWD_HEAD")) : 1;   // attention backward with the re-read operands of a (batch, head) resident in LDS (bf16, 128 < L <= 512)
int g_opt_attn_keep_bits = getenv("P5_ATTN_KEEP_BITS") ? atoi(getenv("P5_ATTN_KEEP_BITS")) : 1;   // the long-sequence forward stores its dropout decisions as lane masks for the backward (bf16, L > 128; decided when the workspace is laid out)
int g_opt_attn_op_keep_bits = 0;      // test hook: the standalone attention ops use a keep-mask buffer of their own (p5_lib.hip, op_keep_bits)
int g_opt_attn_small = getenv("P5_ATTN_SMALL") ? atoi(getenv("P5_ATTN_SMALL")) : 1;   // one-launch backward for Lq <= 16
int g_opt_attn_fused = getenv("P5_ATTN_FUSED") ? atoi(getenv("P5_ATTN_FUSED")) : 1;   // fused dQ/dK/dV attention backward (bf16, L <= 128)

template <class T>
static int launch_attn_fwd_impl(const P5AttnArgs& a, hipStream_t s) {
  P5_REQUIRE(a.Lk >= 1 && a.Lk <= 512 && a.Lq >= 1 && a.Lq <= 512, "attention: 1 <= L <= 512");
  P5_PROF_FLOPS(4.0 * a.B * a.H * a.Lq * a.Lk * 64);
  if constexpr (sizeof(T) == 2) {
    // one workgroup per (batch, head): K and V fetched once, every load up front, one barrier (p5_attn.h)
    if (g_opt_attn_fwd_wg && a.Lq <= 128 && a.Lk <= 128) {
      if (a.Lq > 64) P5_LAUNCH((p5_attn_fwd_wg_kernel<T, 8>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else P5_LAUNCH((p5_attn_fwd_wg_kernel<T, 4>), dim3(a.B * a.H), dim3(256), 0, s, a);
      return P5_KCHECK();
    }
    // the same for longer key ranges: K and V of the head resident in LDS (2 x 64 KiB at Lk = 512), the waves loop over the query blocks
    if (g_opt_attn_fwd_head && a.Lk > 128) {
      // (P5AttnArgs::keep_bits is honoured by the head-resident kernels only: the caller passes it only if forward AND backward take them)
      P5_PROF_TAG(a.Lk <= 256 ? "head-resident K/V, 256 keys" : "head-resident K/V, 512 keys");
      const bool bits = a.keep_bits != nullptr && a.drop.state != nullptr && a.drop.thr != 0;
      if (a.Lk <= 256 && bits) P5_LAUNCH((p5_attn_fwd_head_kernel<16, true>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else if (a.Lk <= 256) P5_LAUNCH((p5_attn_fwd_head_kernel<16, false>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else if (bits) P5_LAUNCH((p5_attn_fwd_head_kernel<32, true>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else P5_LAUNCH((p5_attn_fwd_head_kernel<32, false>), dim3(a.B * a.H), dim3(512), 0, s, a);
      return P5_KCHECK();
    }
  }
  dim3 grid((a.Lq + 63) / 64, a.B * a.H), block(256);
  if (a.Lk <= 64) P5_LAUNCH((p5_attn_fwd_kernel<T, 4>), grid, block, 0, s, a);
  else if (a.Lk <= 128) P5_LAUNCH((p5_attn_fwd_kernel<T, 8>), grid, block, 0, s, a);
  else if (a.Lk <= 256) P5_LAUNCH((p5_attn_fwd_kernel<T, 16>), grid, block, 0, s, a);
  else P5_LAUNCH((p5_attn_fwd_kernel<T, 32>), grid, block, 0, s, a);
  return P5_KCHECK();
}
template <class T>
static int launch_attn_bwd_impl(const P5AttnArgs& a, hipStream_t s) {
  P5_REQUIRE(a.Lk >= 1 && a.Lk <= 512 && a.Lq >= 1 && a.Lq <= 512, "attention: 1 <= L <= 512");
  dim3 block(256);
  // short query blocks (the decoder's self- and cross-attention): dQ, dK, dV in one launch, the four waves split the keys (p5_attn.h)
  if (g_opt_attn_small && a.Lq <= 16) {
    P5_LAUNCH((p5_attn_bwd_small_kernel<T>), dim3(1, a.B * a.H), block, 0, s, a);
    return P5_KCHECK();
  }
  if constexpr (sizeof(T) == 2) {
    // one workgroup per (batch, head) that reads Q, K, V, dO once (p5_attn.h)
    if (g_opt_attn_fused && a.Lq <= 128 && a.Lk <= 128 && a.Lq > 16 && a.Lk > 16) {
      P5_REQUIRE(a.dot_out == nullptr || p5l_attn_bwd_dot_ok(1, a), "attention backward: dot_out needs self-attention shapes and 16-byte-aligned gradient rows");
      P5_LAUNCH((p5_attn_bwd_fused_kernel<T>), dim3(a.B * a.H), dim3(512), 0, s, a);
      return P5_KCHECK();
    }
  }
  if constexpr (sizeof(T) == 2) {
    // longer sequences: K / V (dQ pass) and Q / dO (dK, dV pass) of the head resident in LDS, one relative-bias slot per (batch, head)
    if (g_opt_attn_bwd_head && (a.Lq > 128 || a.Lk > 128)) {
      if (a.Lk <= 256) P5_LAUNCH((p5_attn_bwd_dq_head_kernel<16>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else P5_LAUNCH((p5_attn_bwd_dq_head_kernel<32>), dim3(a.B * a.H), dim3(512), 0, s, a);
      P5_TRY(P5_KCHECK());
      if (a.Lq <= 256) P5_LAUNCH((p5_attn_bwd_dkv_head_kernel<16>), dim3(a.B * a.H), dim3(512), 0, s, a);
      else P5_LAUNCH((p5_attn_bwd_dkv_head_kernel<32>), dim3(a.B * a.H), dim3(512), 0, s, a);
      return P5_KCHECK();
    }
  }
  P5_LAUNCH((p5_attn_bwd_dq_kernel<T>), dim3((a.Lq + 63) / 64, a.B * a.H), block, 0, s, a);
  P5_TRY(P5_KCHECK());
  P5_LAUNCH((p5_attn_bwd_dkv_kernel<T>), dim3((a.Lk + 63) / 64, a.B * a.H), block, 0, s, a);
  return P5_KCHECK();
}


// how many slots of the relative-bias partial table one backward launch writes (slot = b * (slots / B) + query block): the whole-head
// kernels have one workgroup per (batch, head), the blocked dQ kernel one per 64 queries -- must mirror launch_attn_bwd_impl's choice
int p5l_attn_bwd_slots(int bf16_mode, int B, int Lq, int Lk) {
  if (g_opt_attn_small && Lq <= 16) return B;
  if (bf16_mode && g_opt_attn_fused && Lq <= 128 && Lk <= 128 && Lq > 16 && Lk > 16) return B;
  if (bf16_mode && g_opt_attn_bwd_head && (Lq > 128 || Lk > 128)) return B;
  return B * ((Lq + 63) / 64);
}
// P5AttnArgs::dot_out (row sums of <d qkv, qkv>) is written by the fused kernel only: self-attention (Lq == Lk), vector stores
bool p5l_attn_bwd_dot_ok(int bf16_mode, const P5AttnArgs& a) {
  return bf16_mode && g_opt_attn_fused && a.Lq == a.Lk && a.Lq <= 128 && a.Lq > 16 && (a.lddq % 8) == 0 && (a.lddk % 8) == 0 && (a.lddv % 8) == 0 &&
         ((uintptr_t)a.dQ % 16) == 0 && ((uintptr_t)a.dK % 16) == 0 && ((uintptr_t)a.dV % 16) == 0;
}
int p5l_attn_fwd(int bf16_mode, const P5AttnArgs& a, hipStream_t s) { return bf16_mode ? launch_attn_fwd_impl<bf16>(a, s) : launch_attn_fwd_impl<float>(a, s); }
int p5l_attn_bwd(int bf16_mode, const P5AttnArgs& a, hipStream_t s) { return bf16_mode ? launch_attn_bwd_impl<bf16>(a, s) : launch_attn_bwd_impl<float>(a, s); }
