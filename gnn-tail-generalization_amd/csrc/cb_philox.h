// Counter-based RNG shared by the dropout kernels: Philox4x32-10 keyed by the 64-bit seed, counter = flat
// element index / 4.  The keep-mask of an element is a pure function of (seed, flat index), so forward and
// backward (and fused epilogues) regenerate it instead of storing it.
#pragma once
#include "cb_common.h"

namespace cb {

#ifndef CB_PHILOX_ROUNDS
#define CB_PHILOX_ROUNDS 10
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  uint32_t c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
  for (int r = 0; r < CB_PHILOX_ROUNDS; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_lo_u32 / v_mul_hi_u32 pair: integer multiplies are
    // quarter rate, and they are what bounds the elementwise kernels that draw several masks per element (the trunk's input stage)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// m[i] = keep(seed, 4*quad + i) ? scale : 0
__device__ __forceinline__ void keep4(uint64_t seed, int64_t quad, uint32_t thresh, float scale, float (&m)[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)quad, (uint32_t)((uint64_t)quad >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = (r[i] >= thresh) ? scale : 0.f;
}

static inline uint32_t dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

}  // namespace cb
