// Shared helpers for libcoldbrew_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/coldbrew_hip.h"

namespace cb {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);
int* device_error_word();            // cb_error.hip: four ints of device-visible host memory (null if the allocation failed)
const char* device_error_text();     // what the word currently says

#define CB_CHECK_ARG(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      cb::set_error(__VA_ARGS__);      \
      return (code);                   \
    }                                  \
  } while (0)

#define CB_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      cb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return CB_E_HIP;                                                                      \
    }                                                                                       \
  } while (0)

#define CB_LAUNCH_CHECK() CB_HIP(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int blocks_for(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// a*x + b*y and x*s + b with ONE fixed rounding sequence wherever they occur (residual mix: cb_axpby_f32 and the fused aggregation
// store; aggregation epilogue: plain and fused store).  Left to the compiler's contraction choice, the fused and the operator-by-
// operator forward differ in the last bit from one build to the next and the paths stop being bit-identical to each other.
__device__ __forceinline__ float mix2(float a, float x, float b, float y) { return __fmaf_rn(a, x, __fmul_rn(b, y)); }
__device__ __forceinline__ float scale_add(float x, float s, float b) { return __fmaf_rn(x, s, b); }

// wave-uniform broadcast helpers (values land in SGPRs)
__device__ __forceinline__ int bcast_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int bcast_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// bf16 storage helpers (round-to-nearest-even, as torch's float -> bfloat16 conversion)
typedef uint16_t bf16_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
  return make_uint2((uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16), (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16));
}

}  // namespace cb
