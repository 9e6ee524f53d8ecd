// Graph ingest for the aggregation: edge_index (int64 COO) -> CSR by dst and by src,
// degree norms, and the hub plan.  Replaces the Python-list DGL graph build of
// GNN_model/GCN.py:92-95 and the per-layer degree queries of GCN.py:188,206,243.
//
// Pipeline (all on the caller's stream, no allocation):
//   pack    key[e] = major[e] << bits | minor[e]            (uint64, 2*bits significant)
//   sort    stable LSD radix sort over the 2*bits key bits (cb_sort.hip: 8 bits per pass, ballot-match ranks, no atomics)
//   unpack  col[e] = key[e] & mask
//   rowptr  rowptr[v] = lower_bound(key, v << bits)          (N+1 binary searches)
// so each row lists its neighbours in ascending id; the result depends only on the
// edge multiset (bit-exact contract of SURVEY.md §8 a3).
#include <cstring>

#include "cb_common.h"
#include "cb_sort.h"

namespace cb {

__global__ void k_init_flags(int32_t* flags) {
  if (threadIdx.x == 0) {
    flags[0] = 0;
    flags[1] = 0;
    flags[2] = 1;
    flags[3] = 0;
  }
}

__global__ void k_pack_keys(const int64_t* __restrict__ major, const int64_t* __restrict__ minor, int64_t E, int64_t N,
                            int bits, uint64_t* __restrict__ keys, int32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (i < E) {
    int64_t a = major[i], b = minor[i];
    if (a < 0 || a >= N || b < 0 || b >= N) {
      bad = 1;
      a = 0;
      b = 0;
    }
    keys[i] = ((uint64_t)a << bits) | (uint64_t)b;
  }
  unsigned long long m = __ballot(bad);
  if (m && lane_id() == 0) atomicAdd(&flags[1], (int)__popcll(m));
}

__global__ void k_unpack_cols(const uint64_t* __restrict__ keys, int64_t E, uint64_t mask, int32_t* __restrict__ col) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) col[i] = (int32_t)(keys[i] & mask);
}

template <typename RP>      // int32_t (E < 2^31) or int64_t row pointers
__global__ void k_rowptr_search(const uint64_t* __restrict__ keys, int64_t E, int64_t N, int bits, RP* __restrict__ rowptr) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > N) return;
  uint64_t target = (uint64_t)v << bits;
  int64_t lo = 0, hi = E;  // first index with key >= target
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[v] = (RP)lo;
}

template <typename RP>
__global__ void k_degree_stats(const RP* __restrict__ rowptr, int64_t N, int32_t* flags) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0, zero = 0;
  if (v < N) {
    const int64_t dd = (int64_t)(rowptr[v + 1] - rowptr[v]);
    deg = dd > INT32_MAX ? INT32_MAX : (int)dd;      // (flags[3] saturates: a single row of >= 2^31 edges is refused by the caller)
    zero = deg == 0;
  }
  unsigned long long m = __ballot(zero);
  // wave max of deg
  for (int off = 32; off > 0; off >>= 1) deg = max(deg, __shfl_xor(deg, off));
  if (lane_id() == 0) {
    if (m) atomicAdd(&flags[0], (int)__popcll(m));
    if (deg > 0) atomicMax(&flags[3], deg);
  }
}

template <typename T>
__global__ void k_compare(const T* __restrict__ a, const T* __restrict__ b, int64_t n, int32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int diff = (i < n) ? (a[i] != b[i]) : 0;
  if (__ballot(diff) && lane_id() == 0) atomicAnd(&flags[2], 0);
}

template <typename RP>
__global__ void k_deg_norm(const RP* __restrict__ rowptr, int64_t N, float* __restrict__ norm) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < N) {
    const int64_t deg = (int64_t)(rowptr[v + 1] - rowptr[v]);
    float x = (float)(deg > 1 ? deg : 1);
    norm[v] = 1.0f / sqrtf(x);  // degs.float().clamp(min=1) ** -0.5
  }
}

__global__ void k_hub_count(const int32_t* __restrict__ rowptr, int64_t N, int T, int32_t* counts) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int hub = 0, chunks = 0;
  if (v < N) {
    int deg = rowptr[v + 1] - rowptr[v];
    if (deg > T) {
      hub = 1;
      chunks = (deg + T - 1) / T;
    }
  }
  unsigned long long m = __ballot(hub);
  if (m) {
    for (int off = 32; off > 0; off >>= 1) chunks += __shfl_xor(chunks, off);
    if (lane_id() == 0) {
      atomicAdd(&counts[0], (int)__popcll(m));
      atomicAdd(&counts[1], chunks);
    }
  }
}

// Hub rows in ASCENDING row order, without atomics: per-block counts, a single-block exclusive scan of them, an ordered write.  (Round 6: the order
// of hub_rows decides which rows share a block of k_spmm_hub_finish — and with it the order in which that kernel's column sums are added
// (cb_spmm_csr_store_bwd_mix_f32); a cursor handed out by atomicAdd made that order differ from process to process.)
__global__ void __launch_bounds__(256) k_hub_block_count(const int32_t* __restrict__ rowptr, int64_t N, int T, int32_t* __restrict__ block_count) {
  __shared__ int s_w[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool hub = v < N && rowptr[v + 1] - rowptr[v] > T;
  const unsigned long long m = __ballot(hub);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = (int)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(1024) k_hub_block_scan(int32_t* __restrict__ block_count, int nb) {      // in place: counts -> exclusive prefix
  __shared__ int s_part[1024];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int start = 0; start < nb; start += blockDim.x) {
    const int i = start + threadIdx.x;
    const int c = i < nb ? block_count[i] : 0;
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {
      const int add = (threadIdx.x >= (unsigned)off) ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += add;
      __syncthreads();
    }
    const int incl = s_part[threadIdx.x], base = s_base;
    if (i < nb) block_count[i] = base + incl - c;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_base = base + incl;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_hub_collect(const int32_t* __restrict__ rowptr, int64_t N, int T, int cap, int32_t* __restrict__ hub_rows,
                                                     const int32_t* __restrict__ block_base) {
  __shared__ int s_w[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool hub = v < N && rowptr[v + 1] - rowptr[v] > T;
  const unsigned long long m = __ballot(hub);
  const int lane = lane_id(), w = threadIdx.x >> 6;
  if (lane == 0) s_w[w] = (int)__popcll(m);
  __syncthreads();
  int before = 0;
  for (int j = 0; j < w; ++j) before += s_w[j];
  const int slot = block_base[blockIdx.x] + before + (int)__popcll(m & ((1ull << lane) - 1ull));
  if (hub && slot < cap) hub_rows[slot] = (int32_t)v;
}

// Single block: chunk pointers of the hub rows in the order of hub_rows (ascending row ids, k_hub_collect); results are reduced per hub in chunk order.
__global__ void k_hub_scan(const int32_t* __restrict__ rowptr, int T, int n_hubs, const int32_t* __restrict__ hub_rows,
                           int32_t* __restrict__ hub_chunk_ptr) {
  __shared__ int s_part[1024];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int start = 0; start < n_hubs; start += blockDim.x) {
    int i = start + threadIdx.x;
    int c = 0;
    if (i < n_hubs) {
      int r = hub_rows[i];
      c = (rowptr[r + 1] - rowptr[r] + T - 1) / T;
    }
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {  // Hillis-Steele inclusive scan
      int add = (threadIdx.x >= (unsigned)off) ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += add;
      __syncthreads();
    }
    int incl = s_part[threadIdx.x];
    int base = s_base;
    if (i < n_hubs) hub_chunk_ptr[i] = base + incl - c;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_base = base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) hub_chunk_ptr[n_hubs] = s_base;
}

static int key_bits(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) < N) ++b;
  return b;
}

static size_t sort_temp_bytes(int64_t E, int /*bits*/) { return sort_u64_temp_bytes(E); }

template <typename RP>
static int build_one(const int64_t* major, const int64_t* minor, int64_t E, int64_t N, int bits, RP* rowptr,
                     int32_t* col, int32_t* flags, uint64_t* keys_a, uint64_t* keys_b, void* temp, size_t temp_bytes,
                     hipStream_t st) {
  const int B = 256;
  if (E > 0) {
    hipLaunchKernelGGL(k_pack_keys, dim3(blocks_for(E, B)), dim3(B), 0, st, major, minor, E, N, bits, keys_a, flags);
    CB_LAUNCH_CHECK();
    const int rc = sort_u64(temp, temp_bytes, keys_a, keys_b, E, 2 * bits, st);
    if (rc != CB_OK) return rc;
    uint64_t mask = (((uint64_t)1) << bits) - 1;
    hipLaunchKernelGGL(k_unpack_cols, dim3(blocks_for(E, B)), dim3(B), 0, st, keys_b, E, mask, col);
    CB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_rowptr_search<RP>, dim3(blocks_for(N + 1, B)), dim3(B), 0, st, keys_b, E, N, bits, rowptr);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

}  // namespace cb

using namespace cb;

extern "C" size_t cb_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  int bits = key_bits(N < 2 ? 2 : N);
  size_t keys = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint64_t), 256);
  return 2 * keys + align_up(sort_temp_bytes(E > 0 ? E : 1, bits), 256) + 256;
}

template <typename RP>
static int csr_from_coo(const char* who, const int64_t* src, const int64_t* dst, int64_t E, int64_t N, RP* rowptr, int32_t* col, RP* rowptr_t,
                        int32_t* col_t, int32_t* flags, void* workspace, size_t workspace_bytes, void* stream) {
  CB_CHECK_ARG(E >= 0 && N >= 0, CB_E_INVALID, "%s: negative size (E=%lld, N=%lld)", who, (long long)E, (long long)N);
  CB_CHECK_ARG((sizeof(RP) == 8 ? E < ((int64_t)1 << 32) : E < INT32_MAX) && N < INT32_MAX, CB_E_RANGE,
               "%s: E=%lld / N=%lld exceed the index contract (int32 column ids; %s row pointers)", who, (long long)E, (long long)N,
               sizeof(RP) == 8 ? "int64" : "int32: use cb_csr64_from_coo_i64 for E >= 2^31");
  CB_CHECK_ARG(rowptr && rowptr_t && flags && (E == 0 || (src && dst && col && col_t)), CB_E_INVALID, "%s: null pointer", who);
  CB_CHECK_ARG(workspace && workspace_bytes >= cb_csr_workspace_bytes(E, N), CB_E_WORKSPACE, "%s: workspace %zu < required %zu", who,
               workspace_bytes, cb_csr_workspace_bytes(E, N));
  hipStream_t st = (hipStream_t)stream;
  int bits = key_bits(N < 2 ? 2 : N);
  size_t keys = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint64_t), 256);
  char* w = (char*)workspace;
  uint64_t* keys_a = (uint64_t*)w;
  uint64_t* keys_b = (uint64_t*)(w + keys);
  void* temp = w + 2 * keys;
  size_t temp_bytes = workspace_bytes - 2 * keys;

  hipLaunchKernelGGL(k_init_flags, dim3(1), dim3(64), 0, st, flags);
  CB_LAUNCH_CHECK();
  // by-dst CSR: rows = dst, cols = src (forward aggregation, GCN.py:238)
  int rc = build_one<RP>(dst, src, E, N, bits, rowptr, col, flags, keys_a, keys_b, temp, temp_bytes, st);
  if (rc) return rc;
  // by-src CSR: rows = src, cols = dst (reverse graph, backward of the aggregation)
  rc = build_one<RP>(src, dst, E, N, bits, rowptr_t, col_t, flags, keys_a, keys_b, temp, temp_bytes, st);
  if (rc) return rc;
  const int B = 256;
  if (N > 0) {
    hipLaunchKernelGGL(k_degree_stats<RP>, dim3(blocks_for(N, B)), dim3(B), 0, st, rowptr, N, flags);
    CB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_compare<RP>, dim3(blocks_for(N + 1, B)), dim3(B), 0, st, rowptr, rowptr_t, N + 1, flags);
  CB_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(k_compare<int32_t>, dim3(blocks_for(E, B)), dim3(B), 0, st, col, col_t, E, flags);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

extern "C" int cb_csr_from_coo_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* rowptr,
                                   int32_t* col, int32_t* rowptr_t, int32_t* col_t, int32_t* flags, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return csr_from_coo<int32_t>("cb_csr_from_coo_i64", src, dst, E, N, rowptr, col, rowptr_t, col_t, flags, workspace, workspace_bytes, stream);
}

// The same with int64 row pointers: E >= 2^31 edge_index columns on one device (SURVEY.md 8b "int64 rowptr if E >= 2^31"; the reference's
// graph is int64 throughout, GNN_model/GCN.py:93-94).  Column ids stay int32 (N < 2^31).  The aggregation kernels index edges with 32 bits
// INSIDE a launch: the host cuts the rows into blocks of < 2^31 edges and hands each block over as an ordinary CSR whose row pointers are
// rebased by cb_csr_rebase_i64 (graph.SegmentedCSRGraph) — the form a rank's row block has in the node-sharded path.
extern "C" int cb_csr64_from_coo_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t* rowptr, int32_t* col,
                                     int64_t* rowptr_t, int32_t* col_t, int32_t* flags, void* workspace, size_t workspace_bytes, void* stream) {
  return csr_from_coo<int64_t>("cb_csr64_from_coo_i64", src, dst, E, N, rowptr, col, rowptr_t, col_t, flags, workspace, workspace_bytes, stream);
}

__global__ void k_rebase(const int64_t* __restrict__ rowptr, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(rowptr[i] - rowptr[0]);
}

// out[i] = rowptr[row0 + i] - rowptr[row0] for i <= n_rows: the int32 row pointers of the row block [row0, row0 + n_rows), whose edges
// (fewer than 2^31: checked by the caller, who knows rowptr's values) start at col + rowptr[row0].
extern "C" int cb_csr_rebase_i64(const int64_t* rowptr, int64_t row0, int64_t n_rows, int32_t* out, void* stream) {
  CB_CHECK_ARG(rowptr && out && row0 >= 0 && n_rows >= 0, CB_E_INVALID, "cb_csr_rebase_i64: bad argument");
  hipLaunchKernelGGL(k_rebase, dim3(blocks_for(n_rows + 1, 256)), dim3(256), 0, (hipStream_t)stream, rowptr + row0, n_rows + 1, out);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_deg_norm_i64ptr_f32(const int64_t* rowptr, int64_t N, float* norm, void* stream) {
  CB_CHECK_ARG(N >= 0 && (N == 0 || (rowptr && norm)), CB_E_INVALID, "cb_deg_norm_i64ptr_f32: bad argument");
  if (N == 0) return CB_OK;
  hipLaunchKernelGGL(k_deg_norm<int64_t>, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, N, norm);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_deg_norm_f32(const int32_t* rowptr, int64_t N, float* norm, void* stream) {
  CB_CHECK_ARG(N >= 0 && (N == 0 || (rowptr && norm)), CB_E_INVALID, "cb_deg_norm_f32: bad argument");
  if (N == 0) return CB_OK;
  hipLaunchKernelGGL(k_deg_norm<int32_t>, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, N, norm);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_spmm_hub_count(const int32_t* rowptr, int64_t N, int32_t T, int32_t* counts, void* stream) {
  CB_CHECK_ARG(rowptr && counts && N >= 0 && T > 0, CB_E_INVALID, "cb_spmm_hub_count: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CB_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st));
  if (N > 0) {
    hipLaunchKernelGGL(k_hub_count, dim3(blocks_for(N, 256)), dim3(256), 0, st, rowptr, N, T, counts);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

extern "C" int64_t cb_spmm_hub_fill_scratch_ints(int64_t N) { return N > 0 ? (int64_t)blocks_for(N, 256) : 1; }

extern "C" int cb_spmm_hub_fill(const int32_t* rowptr, int64_t N, int32_t T, int32_t n_hubs, int32_t* hub_rows,
                                int32_t* hub_chunk_ptr, int32_t* cursor, void* stream) {
  CB_CHECK_ARG(rowptr && hub_chunk_ptr && cursor && N >= 0 && T > 0 && n_hubs >= 0 && (n_hubs == 0 || hub_rows), CB_E_INVALID,
               "cb_spmm_hub_fill: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (n_hubs > 0) {      // cursor: scratch of cb_spmm_hub_fill_scratch_ints(N) int32 (per-block hub counts -> their exclusive prefix)
    const int nb = blocks_for(N, 256);
    hipLaunchKernelGGL(k_hub_block_count, dim3(nb), dim3(256), 0, st, rowptr, N, T, cursor);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hub_block_scan, dim3(1), dim3(1024), 0, st, cursor, nb);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hub_collect, dim3(nb), dim3(256), 0, st, rowptr, N, T, n_hubs, hub_rows, (const int32_t*)cursor);
    CB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_hub_scan, dim3(1), dim3(1024), 0, st, rowptr, T, n_hubs, hub_rows, hub_chunk_ptr);
  CB_LAUNCH_CHECK();
  return CB_OK;
}
