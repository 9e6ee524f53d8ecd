// Probe: where does the hardware place the workgroups of a 1-D grid whose footprint allows two workgroups per CU (72 KiB LDS,
// 256 threads, as k_gemm_nn_l3)?  Every workgroup records (XCC id, SE id, CU id, start time); the host prints, for the first
// blocks, which blockIdx values share a CU — the pairs whose phases a start delay would have to separate.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(uint32_t* out, int spin) {
  __shared__ char pad[72 * 1024];
  pad[threadIdx.x] = (char)threadIdx.x;
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (uint32_t)t0;
    out[blockIdx.x * 4 + 3] = (uint32_t)(t0 >> 32) + pad[threadIdx.x & 1] * 0;
  }
}
int main() {
  const int nb = 2048;
  uint32_t* d; hipMalloc(&d, nb * 16);
  k<<<nb, 256>>>(d, 40);
  std::vector<uint32_t> h(nb * 4);
  hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
  std::map<uint32_t, std::vector<int>> by_cu;
  uint64_t tmin = ~0ull;
  for (int b = 0; b < nb; ++b) { uint64_t t = ((uint64_t)h[b*4+3] << 32) | h[b*4+2]; if (t < tmin) tmin = t; }
  for (int b = 0; b < nb; ++b) {
    const uint32_t hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("distinct (xcc, se, sh, cu) = %zu\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 24) break;
    printf("xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf);
    for (int b : kv.second) { uint64_t t = ((uint64_t)h[b*4+3] << 32) | h[b*4+2]; printf(" %d(t=%llu)", b, (unsigned long long)((t - tmin) / 1000)); }
    printf("\n");
  }
  return 0;
}
