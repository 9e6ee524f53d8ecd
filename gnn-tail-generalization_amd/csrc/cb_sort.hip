// Stable LSD radix sort of 64-bit keys, 8 bits per pass (interface + purpose: cb_sort.h).
//
// Per pass three launches, no atomics on global memory, bit-reproducible:
//   k_sort_hist     a block of 4 wavefronts owns 8192 consecutive keys (a wavefront 2048 = 32 rounds of 64 coalesced keys held in
//                   registers); per round the lanes that share a digit find each other with 8 ballots (match), the lowest of them adds the
//                   group's size to the wavefront's private LDS histogram; the block's 256 digit counts go to hist[digit][block]
//   k_sort_scan     one block per digit: exclusive scan of that digit's counts over the blocks (tiles of 1024 with a running carry) + the
//                   digit's total; a last single block turns the 256 totals into digit bases
//   k_sort_scatter  the same blocks re-read their keys, repeat the match per round and write key -> base[digit] + scan[digit][block] +
//                   (earlier wavefronts of the block) + (earlier rounds of the wavefront) + (lower lanes of the match group): stable.
// Bound: HBM — 3 x 8 B per key and pass (two reads, one scattered write in runs of ~32 keys per (block, digit)).
#include "cb_sort.h"

namespace cb {

constexpr int kSortItems = 32;                       // keys per lane
constexpr int kSortWaveKeys = 64 * kSortItems;       // 2048
constexpr int kSortBlockKeys = 4 * kSortWaveKeys;    // 8192

// lanes of the wavefront whose `digit` equals this lane's (inactive lanes: valid = false match nobody)
__device__ __forceinline__ unsigned long long match_digit(unsigned digit, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long v = __ballot(bit);
    m &= bit ? v : ~v;
  }
  return valid ? m : 0ull;
}

__device__ __forceinline__ void load_wave_keys(const uint64_t* __restrict__ in, int64_t n, int64_t wave_base, int lane, uint64_t (&k)[kSortItems],
                                               unsigned long long& vmask_any) {
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = wave_base + (int64_t)r * 64 + lane;
    k[r] = i < n ? in[i] : ~0ull;
  }
  vmask_any = wave_base < n;
}

__global__ void __launch_bounds__(256) k_sort_hist(const uint64_t* __restrict__ in, int64_t n, int shift, uint32_t* __restrict__ hist, int64_t n_blocks) {
  __shared__ uint32_t whist[4][256];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) whist[i][t] = 0;
  __syncthreads();
  const int64_t wave_base = (int64_t)blockIdx.x * kSortBlockKeys + (int64_t)w * kSortWaveKeys;
  if (wave_base < n) {
#pragma unroll 4
    for (int r = 0; r < kSortItems; ++r) {
      const int64_t i = wave_base + (int64_t)r * 64 + lane;
      const bool valid = i < n;
      const unsigned digit = valid ? (unsigned)((in[i] >> shift) & 0xffull) : 0u;
      const unsigned long long m = match_digit(digit, valid);
      if (valid && (m & ((1ull << lane) - 1ull)) == 0ull) whist[w][digit] += (uint32_t)__popcll(m);      // the group's lowest lane; one lane per digit
    }
  }
  __syncthreads();
  hist[(int64_t)t * n_blocks + blockIdx.x] = whist[0][t] + whist[1][t] + whist[2][t] + whist[3][t];
}

// grid = 256 digits; scan[digit][block] = sum of hist[digit][block' < block]; total[digit]
__global__ void __launch_bounds__(1024) k_sort_scan(const uint32_t* __restrict__ hist, int64_t n_blocks, uint32_t* __restrict__ scan,
                                                    uint64_t* __restrict__ total) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint64_t s_carry;
  const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t* h = hist + (int64_t)d * n_blocks;
  uint32_t* o = scan + (int64_t)d * n_blocks;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_blocks; base += 1024) {
    const int64_t i = base + t;
    const uint32_t v = i < n_blocks ? h[i] : 0u;
    uint32_t x = v;      // inclusive scan inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) s_wave[w] = x;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int j = 0; j < w; ++j) wave_off += s_wave[j];
    const uint64_t carry = s_carry;
    // (a digit's exclusive prefix over the blocks stays below n < 2^32: 32 bits; the digit bases are 64 bit)
    if (i < n_blocks) o[i] = (uint32_t)(carry + wave_off + x - v);
    __syncthreads();
    if (t == 1023) s_carry = carry + wave_off + x;
    __syncthreads();
  }
  if (t == 0) total[d] = s_carry;
}

__global__ void __launch_bounds__(256) k_sort_digit_base(const uint64_t* __restrict__ total, uint64_t* __restrict__ base) {
  __shared__ uint64_t s[256];
  const int t = threadIdx.x;
  s[t] = total[t];
  __syncthreads();
  if (t == 0) {
    uint64_t run = 0;
    for (int d = 0; d < 256; ++d) {
      const uint64_t c = s[d];
      s[d] = run;
      run += c;
    }
  }
  __syncthreads();
  base[t] = s[t];
}

__global__ void __launch_bounds__(256) k_sort_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n, int shift,
                                                      const uint32_t* __restrict__ scan, const uint64_t* __restrict__ base, int64_t n_blocks) {
  __shared__ uint32_t whist[4][256];
  __shared__ uint64_t wbase[4][256];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) whist[i][t] = 0;
  __syncthreads();
  const int64_t wave_base = (int64_t)blockIdx.x * kSortBlockKeys + (int64_t)w * kSortWaveKeys;
  uint64_t k[kSortItems];
  unsigned long long any;
  load_wave_keys(in, n, wave_base, lane, k, any);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (any) {
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
      const bool valid = wave_base + (int64_t)r * 64 + lane < n;
      const unsigned digit = valid ? (unsigned)((k[r] >> shift) & 0xffull) : 0u;
      const unsigned long long m = match_digit(digit, valid);
      if (valid && (m & below) == 0ull) whist[w][digit] += (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  {      // first output position of every (wavefront, digit) of this block
    uint64_t run = base[t] + scan[(int64_t)t * n_blocks + blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wbase[i][t] = run;
      run += whist[i][t];
    }
  }
  __syncthreads();
  if (any) {
    uint64_t* mine = wbase[w];      // (only this wavefront touches its row from here on: a running output cursor per digit)
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
      const bool valid = wave_base + (int64_t)r * 64 + lane < n;
      const unsigned digit = valid ? (unsigned)((k[r] >> shift) & 0xffull) : 0u;
      const unsigned long long m = match_digit(digit, valid);
      uint64_t pos = 0;
      if (valid) pos = mine[digit];                                   // every lane of the group reads the cursor ...
      if (valid && (m & below) == 0ull) mine[digit] = pos + (uint64_t)__popcll(m);      // ... before its lowest lane advances it (LDS ops of a wavefront are in order)
      if (valid) out[pos + (uint64_t)__popcll(m & below)] = k[r];
    }
  }
}

static inline int64_t sort_blocks(int64_t n) { return (n + kSortBlockKeys - 1) / kSortBlockKeys; }

size_t sort_u64_temp_bytes(int64_t n) {
  const int64_t nb = sort_blocks(n < 1 ? 1 : n);
  return align_up((size_t)nb * 256 * sizeof(uint32_t), 256) * 2 + 2 * 256 * sizeof(uint64_t) + 256;
}

int sort_u64(void* temp, size_t temp_bytes, uint64_t* keys_in, uint64_t* keys_out, int64_t n, int end_bit, hipStream_t st) {
  CB_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 32) && end_bit > 0 && end_bit <= 64, CB_E_RANGE, "sort_u64: bad size / bit range (n < 2^32: per-digit prefixes are 32 bit)");
  if (n == 0) return CB_OK;
  CB_CHECK_ARG(temp && keys_in && keys_out && temp_bytes >= sort_u64_temp_bytes(n), CB_E_WORKSPACE, "sort_u64: scratch too small");
  const int64_t nb = sort_blocks(n);
  CB_CHECK_ARG(nb < INT32_MAX, CB_E_RANGE, "sort_u64: too many blocks");
  char* w = (char*)temp;
  const size_t hb = align_up((size_t)nb * 256 * sizeof(uint32_t), 256);
  uint32_t* hist = (uint32_t*)w;
  uint32_t* scan = (uint32_t*)(w + hb);
  uint64_t* total = (uint64_t*)(w + 2 * hb);
  uint64_t* base = total + 256;
  uint64_t *src = keys_in, *dst = keys_out;
  const int passes = (end_bit + 7) / 8;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)nb), dim3(256), 0, st, src, n, 8 * p, hist, nb);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(1024), 0, st, hist, nb, scan, total);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_digit_base, dim3(1), dim3(256), 0, st, total, base);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)nb), dim3(256), 0, st, src, dst, n, 8 * p, scan, base, nb);
    CB_LAUNCH_CHECK();
    uint64_t* tmp = src; src = dst; dst = tmp;
  }
  if (src != keys_out) CB_HIP(hipMemcpyAsync(keys_out, src, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));      // even number of passes
  return CB_OK;
}

}  // namespace cb
