// Counter-based RNG shared by the dropout kernels: Philox4x32-10 keyed by the 64-bit seed, counter = flat
// element index / 4.  The keep-mask of an element is a pure function of (seed, flat index), so forward and
// backward (and fused epilogues) regenerate it instead of storing it.
#pragma once
#include "cb_common.h"

namespace cb {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  uint32_t c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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
