// p5_rng.h -- counter-based dropout RNG.  keep(seed, site, idx) is a pure function, so forward and
// backward regenerate the same mask without storing it, and oracle/t5_oracle.py::dropout_keep_mask
// reproduces it bit-for-bit on the CPU for train-mode parity tests.
// The reference draws one Bernoulli mask per dropout site (P5_T5.py:125,180; HF modeling_t5.py:86,140,168,
// 400,431) from torch's Philox stream; bitwise parity with torch's stream is impossible,
// only the distribution (keep prob 1-p, scale 1/(1-p)) is matched.
#pragma once
#include <stdint.h>

#ifndef __host__
#define __host__
#define __device__
#endif

__host__ __device__ static inline uint32_t p5_mix32(uint32_t x) {  // "lowbias32"
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ static inline uint32_t p5_site_key(uint32_t site) { return site * 0x85EBCA6Bu + 0x27D4EB2Fu; }
// keep(seed, site, idx): ONE mixing round per element -- the (seed, site) part is mixed separately and is loop-invariant
// (wave-uniform, hoisted by the compiler), so an element costs an xor and lowbias32's two 32-bit multiplies.  32-bit integer
// multiplies are quarter-rate on CDNA; the first version (idx * odd + seed -> mix -> xor site -> mix: five of them per element)
// made the dropout epilogues of the GEMMs and the attention kernels VALU-bound.
__host__ __device__ static inline bool p5_keep(uint32_t seed, uint32_t site_key, uint32_t idx, uint32_t thr) {
  const uint32_t h = p5_mix32(idx ^ p5_mix32(seed + site_key));
  return (h >> 8) >= thr;
}
__host__ static inline uint32_t p5_drop_thr(float p) { return (uint32_t)(p * 16777216.0f); }

// dropout descriptor passed by value to kernels; seed comes from device memory so a captured hipGraph
// replays with a fresh seed every step (state[0] = base seed, state[1] = step counter).
struct P5Drop {
  const uint32_t* state;  // device pointer or nullptr (=> no dropout)
  uint32_t site_key;
  uint32_t thr;
  float scale;  // 1/(1-p)
};
__device__ static inline uint32_t p5_seed(const P5Drop& d) {
  return d.state ? (d.state[0] + d.state[1] * 0x632BE5ABu) : 0u;
}

// site numbering (shared with oracle.t5_oracle.site_id)
static inline uint32_t p5_site_id(int stack, int layer, int which) { return (uint32_t)((stack * 64 + layer) * 8 + which); }
