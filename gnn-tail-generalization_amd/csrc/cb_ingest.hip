// Device-side graph analysis in front of the path: the head / tail / isolated node split and the
// edge crafting that define the edge_index the TeacherGNN consumes.  The reference does all of it with per-edge Python
// loops over dicts and lists (utils.py:300-334 graph_analyze, :680-729 save_graph_analyze, :731-752 craft_isolation_v2,
// :910-941 get_partial_sorted_idx) — minutes and tens of GB of host objects at 1e8 edges.
//
//   cb_id_count_i64        counts[id] += 1 per edge endpoint           (graph_analyze: out- / in-degree per node)
//   cb_value_hist_i32      hist[v] += 1 per node value                  (all repeated medians of get_partial_sorted_idx come
//                                                                        from ONE degree-value histogram: every subset it
//                                                                        takes the median of is {i : deg_i <= t} or {>= t})
//   cb_select_range_i32    indices i (ascending) with lo <= vals[i] <= hi, + bool mask     (np.where(arr <= med) / (>= med))
//   cb_craft_isolation_i64 edges kept in order unless src != dst and an endpoint is flagged (craft_isolation_v2)
//   cb_symmetrize_i64      union with the transpose, duplicates removed, sorted by (row, col)  (utils.py:667-674 ensure_symmetric;
//                          PyG to_undirected as used by load_ogbn, trainer_node_classification.py:574): both directions packed as
//                          64-bit keys, one radix sort (cb_sort.hip, as the CSR ingest of cb_graph.hip), first-of-run
//                          compaction
//
// All integer work, bit-exact, HBM-bound (one or two passes over the edge list).  Order-preserving compaction in two
// passes: per-block keep counts -> exclusive scan of the block counts (one block) -> scatter at block offset + rank inside
// the block (wave ballots + a 4-entry LDS prefix), so the output order is the input order — what np.where and the
// reference's append loop produce.  Integer atomics only (associative: deterministic results).
#include "cb_sort.h"

#include "cb_common.h"

namespace cb {

constexpr int kCB = 256;      // threads per block
constexpr int kItems = 4;     // consecutive elements per thread
constexpr int kTile = kCB * kItems;

__global__ void __launch_bounds__(kCB) k_id_count(const int64_t* __restrict__ ids, int64_t E, int64_t N, int32_t* __restrict__ counts,
                                                  int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = __builtin_nontemporal_load(ids + i);
    if (v >= 0 && v < N) atomicAdd(counts + v, 1);
    else atomicAdd(bad, 1);
  }
}

__global__ void __launch_bounds__(kCB) k_value_hist(const int32_t* __restrict__ vals, int64_t N, int32_t n_bins, int32_t* __restrict__ hist,
                                                    int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = vals[i];
    if (v >= 0 && v < n_bins) atomicAdd(hist + v, 1);
    else atomicAdd(bad, 1);
  }
}

// ---- order-preserving compaction -------------------------------------------------------------------------------
struct RangePred {      // keep element i iff lo <= vals[i] <= hi
  const int32_t* vals;
  int32_t lo, hi;
  __device__ __forceinline__ bool operator()(int64_t i) const {
    const int32_t v = vals[i];
    return v >= lo && v <= hi;
  }
};

struct CraftPred {      // craft_isolation_v2 keeps an edge unless (ori != dst) and (flag[ori] or flag[dst])
  const int64_t* src;
  const int64_t* dst;
  const uint8_t* flag;
  int64_t n;            // node count: an endpoint outside [0, n) never indexes `flag` (the edge is kept; the caller's degree pass reports it)
  __device__ __forceinline__ bool operator()(int64_t e) const {
    const int64_t a = src[e], b = dst[e];
    const bool fa = a >= 0 && a < n && flag[a], fb = b >= 0 && b < n && flag[b];
    return !(a != b && (fa | fb));
  }
};

template <class P>
__global__ void __launch_bounds__(kCB) k_compact_count(P pred, int64_t n, int32_t* __restrict__ block_counts) {
  __shared__ int s_w[kCB / kWave];
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if (base + k < n && pred(base + k)) ++c;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the block counts in place (64-bit offsets out), total -> *total.  One block.
__global__ void __launch_bounds__(1024) k_scan_counts(const int32_t* __restrict__ counts, int64_t nb, int64_t* __restrict__ offsets,
                                                      int64_t* __restrict__ total) {
  __shared__ long long s_part[1024];
  __shared__ long long s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int64_t start = 0; start < nb; start += blockDim.x) {
    const int64_t i = start + threadIdx.x;
    const long long c = i < nb ? counts[i] : 0;
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {   // Hillis-Steele inclusive scan
      const long long add = (threadIdx.x >= (unsigned)off) ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += add;
      __syncthreads();
    }
    const long long incl = s_part[threadIdx.x], b = s_base;
    if (i < nb) offsets[i] = b + incl - c;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_base = b + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_base;
}

// scatter: element i goes to offsets[block] + (number of kept elements before i inside the block)
template <class P, class W>
__global__ void __launch_bounds__(kCB) k_compact_scatter(P pred, int64_t n, const int64_t* __restrict__ offsets, W write) {
  __shared__ int s_w[kCB / kWave];
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  bool keep[kItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    keep[k] = base + k < n && pred(base + k);
    c += keep[k];
  }
  // exclusive prefix of c over the block's threads (thread order = element order)
  int incl = c;
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane_id() >= off) incl += t;
  }
  if (lane_id() == kWave - 1) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  int wave_base = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_base += s_w[w];
  int64_t pos = offsets[blockIdx.x] + wave_base + incl - c;
#pragma unroll
  for (int k = 0; k < kItems; ++k)
    if (keep[k]) write(pos++, base + k);
}

struct IndexWriter {    // out_idx[pos] = i; mask[i] = 1
  int64_t* out_idx;
  uint8_t* mask;
  __device__ __forceinline__ void operator()(int64_t pos, int64_t i) const {
    if (out_idx) out_idx[pos] = i;
    if (mask) mask[i] = 1;
  }
};

struct EdgeWriter {     // (out_src, out_dst)[pos] = (src, dst)[e]
  const int64_t* src;
  const int64_t* dst;
  int64_t* out_src;
  int64_t* out_dst;
  __device__ __forceinline__ void operator()(int64_t pos, int64_t e) const {
    out_src[pos] = src[e];
    out_dst[pos] = dst[e];
  }
};

__global__ void __launch_bounds__(kCB) k_pack_both(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int64_t N,
                                                   int bits, uint64_t* __restrict__ keys, int32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t a = src[i], b = dst[i];
    if (a < 0 || a >= N || b < 0 || b >= N) {
      atomicAdd(bad, 1);
      a = b = 0;
    }
    keys[i] = ((uint64_t)a << bits) | (uint64_t)b;
    keys[E + i] = ((uint64_t)b << bits) | (uint64_t)a;
  }
}

struct FirstOfRunPred {   // keep sorted key i iff it differs from its predecessor
  const uint64_t* keys;
  __device__ __forceinline__ bool operator()(int64_t i) const { return i == 0 || keys[i] != keys[i - 1]; }
};

struct KeyWriter {        // (out_row, out_col)[pos] = unpack(keys[i])
  const uint64_t* keys;
  int bits;
  int64_t* out_row;
  int64_t* out_col;
  __device__ __forceinline__ void operator()(int64_t pos, int64_t i) const {
    const uint64_t k = keys[i];
    out_row[pos] = (int64_t)(k >> bits);
    out_col[pos] = (int64_t)(k & ((((uint64_t)1) << bits) - 1));
  }
};

static int ingest_key_bits(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) < N) ++b;
  return b;
}

static size_t ingest_sort_temp_bytes(int64_t n, int /*bits*/) { return sort_u64_temp_bytes(n); }

static inline int64_t n_blocks(int64_t n) { return (n + kTile - 1) / kTile; }

}  // namespace cb

using namespace cb;

extern "C" int cb_id_count_i64(const int64_t* ids, int64_t E, int64_t N, int32_t* counts, int32_t* n_bad, void* stream) {
  CB_CHECK_ARG(E >= 0 && N >= 0 && N < INT32_MAX && counts && n_bad && (E == 0 || ids), CB_E_INVALID, "cb_id_count_i64: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CB_HIP(hipMemsetAsync(counts, 0, (size_t)N * sizeof(int32_t), st));
  CB_HIP(hipMemsetAsync(n_bad, 0, sizeof(int32_t), st));
  if (E == 0) return CB_OK;
  int64_t nb = (E + kCB - 1) / kCB;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(k_id_count, dim3((unsigned)nb), dim3(kCB), 0, st, ids, E, N, counts, n_bad);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_value_hist_i32(const int32_t* vals, int64_t N, int32_t n_bins, int32_t* hist, int32_t* n_bad, void* stream) {
  CB_CHECK_ARG(N >= 0 && n_bins > 0 && hist && n_bad && (N == 0 || vals), CB_E_INVALID, "cb_value_hist_i32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CB_HIP(hipMemsetAsync(hist, 0, (size_t)n_bins * sizeof(int32_t), st));
  CB_HIP(hipMemsetAsync(n_bad, 0, sizeof(int32_t), st));
  if (N == 0) return CB_OK;
  int64_t nb = (N + kCB - 1) / kCB;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(k_value_hist, dim3((unsigned)nb), dim3(kCB), 0, st, vals, N, n_bins, hist, n_bad);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_compact_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return align_up((size_t)n_blocks(n) * sizeof(int32_t), 256) + align_up((size_t)n_blocks(n) * sizeof(int64_t), 256);
}

template <class P, class W>
static int run_compact(P pred, W write, int64_t n, int64_t* count, void* ws, size_t ws_bytes, hipStream_t st) {
  const int64_t nb = n_blocks(n);
  int32_t* bc = (int32_t*)ws;
  int64_t* off = (int64_t*)((char*)ws + align_up((size_t)nb * sizeof(int32_t), 256));
  if (n == 0) {
    CB_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), st));
    return CB_OK;
  }
  hipLaunchKernelGGL((k_compact_count<P>), dim3((unsigned)nb), dim3(kCB), 0, st, pred, n, bc);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, (const int32_t*)bc, nb, off, count);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_compact_scatter<P, W>), dim3((unsigned)nb), dim3(kCB), 0, st, pred, n, (const int64_t*)off, write);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_select_range_i32(const int32_t* vals, int64_t N, int32_t lo, int32_t hi, int64_t* out_idx, uint8_t* out_mask,
                                   int64_t* count, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && count && (N == 0 || vals) && (out_idx || out_mask), CB_E_INVALID, "cb_select_range_i32: bad argument");
  CB_CHECK_ARG(ws && ws_bytes >= cb_compact_workspace_bytes(N), CB_E_WORKSPACE, "cb_select_range_i32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (out_mask && N > 0) CB_HIP(hipMemsetAsync(out_mask, 0, (size_t)N, st));
  return run_compact(RangePred{vals, lo, hi}, IndexWriter{out_idx, out_mask}, N, count, ws, ws_bytes, st);
}

extern "C" int cb_craft_isolation_i64(const int64_t* src, const int64_t* dst, int64_t E, const uint8_t* node_flag, int64_t N,
                                      int64_t* out_src, int64_t* out_dst, int64_t* count, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(E >= 0 && N >= 0 && count && (E == 0 || (src && dst && node_flag && out_src && out_dst)), CB_E_INVALID,
               "cb_craft_isolation_i64: bad argument");
  CB_CHECK_ARG(ws && ws_bytes >= cb_compact_workspace_bytes(E), CB_E_WORKSPACE, "cb_craft_isolation_i64: workspace too small");
  return run_compact(CraftPred{src, dst, node_flag, N}, EdgeWriter{src, dst, out_src, out_dst}, E, count, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t cb_symmetrize_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  const int64_t n2 = 2 * (E > 0 ? E : 1);
  const int bits = ingest_key_bits(N < 2 ? 2 : N);
  return 2 * align_up((size_t)n2 * sizeof(uint64_t), 256) + align_up(ingest_sort_temp_bytes(n2, bits), 256) + cb_compact_workspace_bytes(n2) + 256;
}

extern "C" int cb_symmetrize_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t* out_row, int64_t* out_col,
                                 int64_t* count, int32_t* n_bad, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(E >= 0 && N >= 0 && count && n_bad && (E == 0 || (src && dst && out_row && out_col)), CB_E_INVALID,
               "cb_symmetrize_i64: bad argument");
  CB_CHECK_ARG(2 * E < INT32_MAX * (int64_t)2 && N < INT32_MAX, CB_E_RANGE, "cb_symmetrize_i64: size exceeds the index contract");
  CB_CHECK_ARG(ws && ws_bytes >= cb_symmetrize_workspace_bytes(E, N), CB_E_WORKSPACE, "cb_symmetrize_i64: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  CB_HIP(hipMemsetAsync(n_bad, 0, sizeof(int32_t), st));
  if (E == 0) {
    CB_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), st));
    return CB_OK;
  }
  const int64_t n2 = 2 * E;
  const int bits = ingest_key_bits(N < 2 ? 2 : N);
  const size_t kb = align_up((size_t)n2 * sizeof(uint64_t), 256);
  char* w = (char*)ws;
  uint64_t* keys_a = (uint64_t*)w;
  uint64_t* keys_b = (uint64_t*)(w + kb);
  void* temp = w + 2 * kb;
  size_t temp_bytes = align_up(ingest_sort_temp_bytes(n2, bits), 256);
  void* cws = w + 2 * kb + temp_bytes;
  int64_t nb = (E + kCB - 1) / kCB;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(k_pack_both, dim3((unsigned)nb), dim3(kCB), 0, st, src, dst, E, N, bits, keys_a, n_bad);
  CB_LAUNCH_CHECK();
  {
    const int rc = sort_u64(temp, temp_bytes, keys_a, keys_b, n2, 2 * bits, st);
    if (rc != CB_OK) return rc;
  }
  return run_compact(FirstOfRunPred{keys_b}, KeyWriter{keys_b, bits, out_row, out_col}, n2, count, cws, ws_bytes - (2 * kb + temp_bytes), st);
}
