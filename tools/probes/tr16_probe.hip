// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which lane receives which 16-bit elements).
// LDS holds u16 value = its own element index.  Lane l supplies the address of elements [row(l)][4*(l&3) .. +3] of a
// row-major image with `stride` elements per row, row(l) = (l & 15) >> 2 + 4 * (l >> 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int stride) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, L = l & 15, g = l >> 4;
  const int row = (L >> 2) + 4 * g, col = 4 * (L & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * stride + col));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 64}) {
    k<<<1, 64>>>(d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d(r%d,c%d)", h[l*4+j], h[l*4+j]/stride, h[l*4+j]%stride); printf("\n"); }
  }
  return 0;
}
