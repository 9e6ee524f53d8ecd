// Dense contractions of the TeacherGNN step on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation, 157 TF/s peak = 16x below bf16, so the
// 1e-4 logits parity holds) — replaces th.matmul(feat_src, weight) (GNN_model/GCN.py:225), the
// nn.Linear layers (GCN.py:105,138) and their autograd GEMMs.
//
// Bound: MFMA (2*M*K*N flop per launch) — with M = #nodes, K,N <= a few hundred the A matrix is
// streamed once from HBM and W stays in L2.
//
//   NN : C[M,N] = act( rowscale[m] * (A[M,K] @ B[K,N]) + addend[m,n] + bias[n] )
//   TN : C[K1,K2] = sum_m A[m,K1] * (rowscale[m] * G[m,K2])      (reduction over the long node axis,
//        split over blocks into partial slabs that a second kernel sums in a fixed order)
//
// Tiling: 256 threads = 4 wavefronts arranged WM x WN (2x2 for wide outputs, 4x1 / 1x4 for skinny
// ones such as the 40-class head), each wavefront owns a 64x64 sub-tile = 2x2 MFMA 32x32 tiles
// (64 accumulator registers); block tile (64*WM) x (64*WN), K step 16.  Both operands sit k-major in
// LDS ([k][tile + 4 pad] floats): a fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32), the row-major A of NN is transposed on the way in.  Global loads of
// K-tile t+1 are issued before the MFMAs of tile t and stored to the other LDS buffer after them (one
// barrier per K step); fragment reads of k-step kk+2 are issued before the MFMAs of k-step kk.
// Epilogue (NN): the accumulators are transposed through LDS 32 rows at a time so that every lane
// handles 4 consecutive columns of one row: row scale / addend / bias are applied on float4 values and
// the tile leaves as coalesced 16-byte stores.
#include <stdlib.h>

#include "cb_common.h"
#include "cb_gemm_core.h"
#include "cb_gemm_limb.h"

namespace cb {

// ---- NN ------------------------------------------------------------------------------------
template <int WM, int WN, bool ALIGNED, bool OUT_BF16, int BKT = BK, int WTN = 2>
__global__ void __launch_bounds__(256) k_gemm_nn(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                 void* __restrict__ Cv, int64_t ldc, int64_t M, int N, int K, GemmEpilogue ep,
                                                 int n_row_blocks, int n_col_blocks, int c_vec_ok, int k_chunk) {
  using T = Tile<WM, WN, BKT, WTN>;
  if (k_chunk) {   // split-K launch: block row blockIdx.y contracts k in [y * k_chunk, ...) into its own partial plane [M, ldc]
    const int k0 = blockIdx.y * k_chunk;
    A += k0;
    B += (int64_t)k0 * ldb;
    Cv = reinterpret_cast<float*>(Cv) + (int64_t)blockIdx.y * M * ldc;
    K = min(k_chunk, K - k0);
  }
  constexpr int BM = T::BM, BN = T::BN, LDA = T::LDA, LDB = T::LDB;
  __shared__ __attribute__((aligned(16))) float smem[T::SMEM_FLOATS];
  auto As = [&](int b) { return smem + b * (BKT * LDA); };
  auto Bs = [&](int b) { return smem + 2 * BKT * LDA + b * (BKT * LDB); };
  // XCD-aware tile order: the column blocks of one row block get ids congruent mod 8, i.e. the same
  // XCD / L2 under the observed round-robin dispatch, so the A rows are fetched from HBM once.
  const int per_group = 8 * n_col_blocks;
  const int grp = blockIdx.x / per_group, r = blockIdx.x % per_group;
  const int row_blk = grp * 8 + (r & 7), col_blk = r >> 3;
  if (row_blk >= n_row_blocks) return;
  const int64_t m0 = (int64_t)row_blk * BM;
  const int n0 = col_blk * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / WN, wc = w % WN;

  f32x16 acc[2][WTN];
  zero_acc<WTN>(acc);
  const int nk = (K + BKT - 1) / BKT;
  RowFrag<BM, BKT> fa;
  KFrag<BN, BKT> fb;
  load_rowmajor<ALIGNED, BM, BKT>(fa, A, lda, m0, M, 0, K, t);
  load_kmajor<ALIGNED, BN, BKT>(fb, B, ldb, 0, K, n0, N, t, nullptr);
  store_rowmajor_T<BM, BKT>(fa, As(0), t);
  store_kmajor<BN, BKT>(fb, Bs(0), t);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_rowmajor<ALIGNED, BM, BKT>(fa, A, lda, m0, M, (kt + 1) * BKT, K, t);
      load_kmajor<ALIGNED, BN, BKT>(fb, B, ldb, (int64_t)(kt + 1) * BKT, K, n0, N, t, nullptr);
    }
    mfma_tile_step<LDA, LDB, BKT, WTN>(As(cur), Bs(cur), wr, wc, lane, acc);
    if (kt + 1 < nk) {
      store_rowmajor_T<BM, BKT>(fa, As(cur ^ 1), t);
      store_kmajor<BN, BKT>(fb, Bs(cur ^ 1), t);
    }
    __syncthreads();
  }

  nn_epilogue<WM, WN, WTN, OUT_BF16>(acc, smem, Cv, ldc, m0, n0, M, N, ep, c_vec_ok, t);
}

// ---- TN ------------------------------------------------------------------------------------
// partial[split][K1][K2]: block (tile i, tile j, split s) reduces rows [s*rows_per_split, ...)
template <int WM, int WN, bool ALIGNED>
__global__ void __launch_bounds__(256) k_gemm_tn(const float* __restrict__ A, int64_t lda, const float* __restrict__ G, int64_t ldg,
                                                 const float* __restrict__ rowscale, float* __restrict__ partial, int64_t M, int K1,
                                                 int K2, int64_t rows_per_split, int tiles_j) {
  using T = Tile<WM, WN>;
  constexpr int BM = T::BM, BN = T::BN, LDA = T::LDA, LDB = T::LDB;
  __shared__ __attribute__((aligned(16))) float smem[T::SMEM_FLOATS];
  auto As = [&](int b) { return smem + b * (BK * LDA); };
  auto Bs = [&](int b) { return smem + 2 * BK * LDA + b * (BK * LDB); };
  const int tile = blockIdx.x, split = blockIdx.y;
  const int i0 = (tile / tiles_j) * BM, j0 = (tile % tiles_j) * BN;
  const int64_t r_begin = (int64_t)split * rows_per_split;
  const int64_t r_end = min(M, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / WN, wc = w % WN;

  f32x16 acc[2][2];
  zero_acc<2>(acc);
  const int64_t nk = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;
  KFrag<BM, BK> fa;
  KFrag<BN, BK> fb;
  if (nk > 0) {
    load_kmajor<ALIGNED, BM, BK>(fa, A, lda, r_begin, r_end, i0, K1, t, nullptr);
    load_kmajor<ALIGNED, BN, BK>(fb, G, ldg, r_begin, r_end, j0, K2, t, rowscale);
    store_kmajor<BM, BK>(fa, As(0), t);
    store_kmajor<BN, BK>(fb, Bs(0), t);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) {
      load_kmajor<ALIGNED, BM, BK>(fa, A, lda, r_begin + (kt + 1) * BK, r_end, i0, K1, t, nullptr);
      load_kmajor<ALIGNED, BN, BK>(fb, G, ldg, r_begin + (kt + 1) * BK, r_end, j0, K2, t, rowscale);
    }
    mfma_tile_step<LDA, LDB, BK, 2>(As(cur), Bs(cur), wr, wc, lane, acc);
    if (kt + 1 < nk) {
      store_kmajor<BM, BK>(fa, As(cur ^ 1), t);
      store_kmajor<BN, BK>(fb, Bs(cur ^ 1), t);
    }
    __syncthreads();
  }
  const int l31 = lane & 31, lh = lane >> 5;
  float* P = partial + (int64_t)split * K1 * K2;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = j0 + wc * 64 + tj * 32 + l31;
      if (n >= K2) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = i0 + wr * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (m < K1) P[(int64_t)m * K2 + n] = acc[ti][tj][reg];
      }
    }
}

__global__ void k_sum_partials(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // eight independent running sums (eight loads in flight per lane: with one, 154 slabs of a Pubmed-sized weight gradient took
  // 37 us), combined in a fixed order
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 8 <= nsplit; p += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += partial[(int64_t)(p + j) * n + i];
  for (; p < nsplit; ++p) s[0] += partial[(int64_t)p * n + i];
  out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

static inline int tn_splits(int64_t M, int tiles) {
  // enough blocks for ~4 per CU, each with at least 8 K-steps of work: a K step of a lone block costs ~1.7 us of load latency,
  // so the Cora / Pubmed-sized reductions (M = 2 708 / 19 717 rows) want many short slabs (64 K-steps per slab left them
  // at 3 / 20 blocks on 256 CUs: 90-190 us per weight gradient instead of ~30)
  int64_t want = (256 * 4 + tiles - 1) / tiles;
  int64_t max_by_work = (M + BK * 8 - 1) / (BK * 8);
  int64_t s = want < max_by_work ? want : max_by_work;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

// CB_GEMM_PLAIN_F32=1 keeps every contraction on the fp32-input MFMA (v_mfma_f32_32x32x2_f32) instead of the
// three-limb bf16 path of cb_gemm_limb.hip
static inline bool use_limb3() {
  static const bool on = getenv("CB_GEMM_PLAIN_F32") == nullptr;
  return on;
}

// fp32 outputs too large to stay cached (> 256 MiB: the Infinity Cache) leave with the streaming policy: measured -1.5 % per launch
// on 10M x 256 outputs (7.96 -> 7.84 ms)
static inline int gemm_nt_store(int64_t M, int64_t N) { return M * N * 4 > ((int64_t)256 << 20); }

// tile shape by output width: 2x2 (128x128) by default; for TN 1x4 (64x256) when K1 <= 64, 4x1 (256x64) when K2 <= 64
static inline void tn_tile(int64_t K1, int64_t K2, int& bm, int& bn) {
  if (K1 <= 64 && K2 > 64) { bm = 64; bn = 256; }
  else if (K2 <= 64 && K1 > 64) { bm = 256; bn = 64; }
  else { bm = 128; bn = 128; }
}

// Split-K of the fallback NN kernel: few output tiles and a long contraction (x @ W_0 of a Cora / Citeseer-sized graph:
// 2 708 x 1 433 x 64 = 11 blocks looping 90 K steps, 220 us) — the K range is cut into `splits` chunks of whole K steps, every
// chunk's raw product goes to its own plane of the workspace and k_splitk_finish sums the planes in a fixed order and applies
// the epilogue once.  0 = do not split.
static inline int nn_splitk(int64_t M, int64_t N, int64_t K, int bm, int bn, int* k_chunk) {
  const int64_t blocks = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  if (blocks * 2 > 256 || K < 512) return 0;
  int64_t s = 256 / blocks;
  const int64_t by_work = K / (BK * 8);
  if (s > by_work) s = by_work;
  if (s > 64) s = 64;
  if (s < 2) return 0;
  const int64_t kc = ((K + s - 1) / s + BK - 1) / BK * BK;
  *k_chunk = (int)kc;
  return (int)((K + kc - 1) / kc);
}

__global__ void k_splitk_finish(const float* __restrict__ partial, int nsplit, int64_t M, int N, float* __restrict__ C, int64_t ldc,
                                GemmEpilogue ep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N;
  const int n = (int)(i - m * N);
  float s = 0.f;
#pragma unroll 4
  for (int p = 0; p < nsplit; ++p) s += partial[(int64_t)p * M * N + i];
  if (ep.rowscale) s *= ep.rowscale[m];
  if (ep.addend) s += ep.addend[m * ep.ld_add + n];
  if (ep.bias) s += ep.bias[n];
  if (ep.relu) s = fmaxf(s, 0.f);
  C[m * ldc + n] = s;
}

template <int WM, int WN, bool OUT_BF16 = false, int WTN = 2>
static int launch_nn(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, GemmEpilogue ep, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  using T = Tile<WM, WN, BK, WTN>;
  const int nrb = (int)((M + T::BM - 1) / T::BM), ncb = (int)((N + T::BN - 1) / T::BN);
  const int64_t groups = (nrb + 7) / 8;
  const dim3 grid((unsigned)(groups * 8 * ncb));
  const bool aligned = al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0;
  const int c_vec_ok = ((uintptr_t)C % (OUT_BF16 ? 8 : 16) == 0) && ldc % 4 == 0 && (!ep.addend || (al16(ep.addend) && ep.ld_add % 4 == 0));
  if constexpr (!OUT_BF16) {
    int k_chunk = 0;
    const int splits = nn_splitk(M, N, K, T::BM, T::BN, &k_chunk);
    if (splits && ws && ws_bytes >= (size_t)splits * M * N * sizeof(float)) {
      const GemmEpilogue raw{};
      const dim3 grid2(grid.x, (unsigned)splits);
      const int pv = ((uintptr_t)ws % 16 == 0) && N % 4 == 0;
      if (aligned)
        hipLaunchKernelGGL((k_gemm_nn<WM, WN, true, false, BK, WTN>), grid2, dim3(256), 0, st, A, lda, B, ldb, ws, N, M, (int)N, (int)K, raw, nrb, ncb, pv, k_chunk);
      else
        hipLaunchKernelGGL((k_gemm_nn<WM, WN, false, false, BK, WTN>), grid2, dim3(256), 0, st, A, lda, B, ldb, ws, N, M, (int)N, (int)K, raw, nrb, ncb, pv, k_chunk);
      CB_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, (const float*)ws, splits, M, (int)N, (float*)C, ldc, ep);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
  }
  if (aligned)
    hipLaunchKernelGGL((k_gemm_nn<WM, WN, true, OUT_BF16, BK, WTN>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb, c_vec_ok, 0);
  else
    hipLaunchKernelGGL((k_gemm_nn<WM, WN, false, OUT_BF16, BK, WTN>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb, c_vec_ok, 0);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

template <int WM, int WN>
static int launch_tn(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* C, int64_t M,
                     int64_t K1, int64_t K2, float* ws, hipStream_t st, const DropSpec* gdrop = nullptr, const DropSpec* adrop = nullptr) {
  using T = Tile<WM, WN>;
  const int tiles_i = (int)((K1 + T::BM - 1) / T::BM), tiles_j = (int)((K2 + T::BN - 1) / T::BN);
  const int nsplit = tn_splits(M, tiles_i * tiles_j);
  int64_t rows_per_split = (M + nsplit - 1) / nsplit;
  rows_per_split = (rows_per_split + 31) / 32 * 32;   // whole K steps of either kernel family
  const bool aligned = al16(A) && al16(G) && lda % 4 == 0 && ldg % 4 == 0;
  const dim3 grid((unsigned)(tiles_i * tiles_j), (unsigned)nsplit);
  CB_CHECK_ARG(!(gdrop || adrop) || (use_limb3() && limb3_tn_eligible(A, lda, G, ldg, K1, K2)), CB_E_INVALID,
               "TN contraction with operand dropout: three-limb path only (check cb_gemm_tn_gdrop_supported first)");
  if (use_limb3() && limb3_tn_eligible(A, lda, G, ldg, K1, K2)) {
    const int rc = launch_tn_limb3(A, lda, G, ldg, rowscale, ws, M, K1, K2, T::BM, nsplit, rows_per_split, st, gdrop, adrop);
    if (rc != CB_OK) return rc;
    const int64_t n = K1 * K2;
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, nsplit, n, C);
    CB_LAUNCH_CHECK();
    return CB_OK;
  }
  if (aligned)
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, true>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, ws, M, (int)K1, (int)K2,
                       rows_per_split, tiles_j);
  else
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, false>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, ws, M, (int)K1, (int)K2,
                       rows_per_split, tiles_j);
  CB_LAUNCH_CHECK();
  const int64_t n = K1 * K2;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, nsplit, n, C);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

}  // namespace cb

using namespace cb;

extern "C" size_t cb_gemm_nn_workspace_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return 0;
  return limb3_nn_workspace_bytes(N, K);
}

extern "C" size_t cb_gemm_nn_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int k_chunk = 0;
  const int splits = N <= 64 ? nn_splitk(M, N, K, 256, 64, &k_chunk) : nn_splitk(M, N, K, 128, 128, &k_chunk);
  return (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int cb_gemm_nn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu,
                              void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, CB_E_INVALID, "cb_gemm_nn_f32: negative size");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && (K == 0 || (A && B)) && lda >= K && ldb >= N && ldc >= N && (!addend || ld_add >= N), CB_E_INVALID,
               "cb_gemm_nn_f32: null pointer or leading dimension too small");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, relu, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, gemm_nt_store(M, N)};
  hipStream_t st = (hipStream_t)stream;
  if (use_limb3() && limb3_nn_eligible(A, lda, B, ldb, N, K)) return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, false, st, ws, ws_bytes);
  if (N <= 64) return launch_nn<4, 1>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes);
  return launch_nn<2, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes);
}

// C = act(rowscale * (A @ B) + addend + bias) and C2 = dropout_p(C) from ONE pass: the input Linear + ReLU of the residual
// trunk followed by the dropout in front of the first GCNConv (GNN_model/GCN.py:105-107,110).  C2's keep-mask is the one
// cb_dropout_f32(C, ..., seed, seed_dev, offset = row0 * N) draws.  When the fused epilogue does not apply to the shape
// (three-limb wide tile: N > 128, 16-byte aligned operands, K % 4 == 0, N % 4 == 0) the two kernels run one after the other.
extern "C" int cb_gemm_nn_drop2_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, float* C2, int64_t ldc2,
                                    int64_t M, int64_t N, int64_t K, const float* rowscale, const float* addend, int64_t ld_add,
                                    const float* bias, int relu, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                    void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_gemm_nn_drop2_f32: bad size or p");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_drop2_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && C2 && (K == 0 || (A && B)) && lda >= K && ldb >= N && ldc >= N && ldc2 >= N && (!addend || ld_add >= N), CB_E_INVALID,
               "cb_gemm_nn_drop2_f32: null pointer or leading dimension too small");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, relu, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, gemm_nt_store(M, N)};
  hipStream_t st = (hipStream_t)stream;
  if (use_limb3() && drop_p > 0.f && limb3_nn_dual_eligible(A, lda, B, ldb, C, ldc, C2, ldc2, M, N, K, ep)) {
    ep.out2 = C2; ep.ld_out2 = ldc2; ep.thresh = dropout_threshold(drop_p); ep.keep_scale = 1.f / (1.f - drop_p);
    ep.seed = seed; ep.seed_dev = seed_dev; ep.row0 = row0;
    return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, false, st, ws, ws_bytes);
  }
  int rc = cb_gemm_nn_f32(A, lda, B, ldb, C, ldc, M, N, K, rowscale, addend, ld_add, bias, relu, ws, ws_bytes, stream);
  if (rc != CB_OK) return rc;
  CB_CHECK_ARG(ldc == N && ldc2 == N, CB_E_INVALID, "cb_gemm_nn_drop2_f32: the two-kernel form needs contiguous outputs");
  return cb_dropout_f32(C, C2, M * N, drop_p, seed, seed_dev, row0 * N, stream);
}

// The same with the dropout IN FRONT of the Linear applied to A as it is staged (GCN.py:104: x = F.dropout(x) before layers_MLP[0]):
// C = act((dropout_{a_seed}(A)) @ B + bias), C2 = dropout_{seed}(C).  A's dropped copy is never written (nor kept for the backward:
// cb_gemm_tn_gdrop_f32 regenerates the mask).  Only where the fused form exists — ask cb_gemm_nn_indrop_supported first.
extern "C" int cb_gemm_nn_indrop_supported(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, const float* C2,
                                           int64_t ldc2, int64_t M, int64_t N, int64_t K) {
  GemmEpilogue ep{};
  return use_limb3() && K % 4 == 0 && limb3_nn_dual_eligible(A, lda, B, ldb, C, ldc, C2, ldc2, M, N, K, ep) ? 1 : 0;
}

extern "C" int cb_gemm_nn_indrop_drop2_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, float* C2, int64_t ldc2,
                                           int64_t M, int64_t N, int64_t K, const float* bias, int relu, float a_drop_p, uint64_t a_seed,
                                           float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits,
                                           void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && drop_p > 0.f && drop_p < 1.f && a_drop_p > 0.f && a_drop_p < 1.f && row0 >= 0, CB_E_INVALID,
               "cb_gemm_nn_indrop_drop2_f32: bad size or p");
  CB_CHECK_ARG(!relu_bits || (N == 256 && relu && (uintptr_t)relu_bits % 8 == 0), CB_E_INVALID,
               "cb_gemm_nn_indrop_drop2_f32: mask words of the ReLU exist for N == 256 with relu only");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_indrop_drop2_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && C2 && A && B && lda >= K && ldb >= N && ldc >= N && ldc2 >= N, CB_E_INVALID, "cb_gemm_nn_indrop_drop2_f32: null pointer or bad ld");
  CB_CHECK_ARG(cb_gemm_nn_indrop_supported(A, lda, B, ldb, C, ldc, C2, ldc2, M, N, K), CB_E_INVALID,
               "cb_gemm_nn_indrop_drop2_f32: shape / alignment outside the fused form (cb_gemm_nn_indrop_supported)");
  GemmEpilogue ep{nullptr, nullptr, 0, bias, relu, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, gemm_nt_store(M, N)};
  ep.out2 = C2; ep.ld_out2 = ldc2; ep.thresh = dropout_threshold(drop_p); ep.keep_scale = 1.f / (1.f - drop_p);
  ep.seed = seed; ep.seed_dev = seed_dev; ep.row0 = row0;
  ep.adrop = DropSpec{dropout_threshold(a_drop_p), 1.f / (1.f - a_drop_p), a_seed, seed_dev, row0, K};
  ep.relu_bits_out = (unsigned long long*)relu_bits;
  return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, false, (hipStream_t)stream, nullptr, 0);
}

extern "C" int cb_gemm_nn_bf16out_f32(const float* A, int64_t lda, const float* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M,
                                      int64_t N, int64_t K, const float* rowscale, const float* addend, int64_t ld_add,
                                      const float* bias, int relu, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, CB_E_INVALID, "cb_gemm_nn_bf16out_f32: negative size");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_bf16out_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && (K == 0 || (A && B)) && lda >= K && ldb >= N && ldc >= N && (!addend || ld_add >= N), CB_E_INVALID,
               "cb_gemm_nn_bf16out_f32: null pointer or leading dimension too small");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, relu, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, gemm_nt_store(M, N)};
  hipStream_t st = (hipStream_t)stream;
  if (use_limb3() && limb3_nn_eligible(A, lda, B, ldb, N, K)) return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, true, st, ws, ws_bytes);
  if (N <= 64) return launch_nn<4, 1, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
  return launch_nn<2, 2, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
}

extern "C" size_t cb_gemm_tn_workspace_bytes(int64_t M, int64_t K1, int64_t K2) {
  if (M <= 0 || K1 <= 0 || K2 <= 0) return 0;
  int bm, bn;
  tn_tile(K1, K2, bm, bn);
  const int tiles = (int)(((K1 + bm - 1) / bm) * ((K2 + bn - 1) / bn));
  return (size_t)tn_splits(M, tiles) * (size_t)K1 * (size_t)K2 * sizeof(float);
}

// The trunk's store on a SUBSET of the node rows as the epilogue of the dense transform in front of it (rows-only forward, trunk.py _layer_on_rows):
//   act = relu(rowscale * (A @ B) + addend + bias)  (-> out_act if given);   C = dropout(c_act * act + c_mix * mix_src[mix_index[m] | row_index[m]])
// A, C, out_act, rowscale, addend: compact [M, .] over the rows row_index[0 .. M) of the node rows; relu_bits ([all node rows][4]) and the dropout
// mask are taken at the node row.  N == 256 on the three-limb wide tile only: ask cb_gemm_nn_store_rows_supported first (else cb_gemm_nn_f32 followed
// by cb_trunk_store_rows_f32, whose results this reproduces bit for bit).
extern "C" int cb_gemm_nn_store_rows_supported(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, int64_t M, int64_t N,
                                               int64_t K) {
  return use_limb3() && N == 256 && M > 0 && limb3_nn_eligible(A, lda, B, ldb, N, K) && al16(C) && ldc % 4 == 0 ? 1 : 0;
}

extern "C" int cb_gemm_nn_store_rows_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                         const float* rowscale, const float* addend, int64_t ld_add, const float* bias, const int64_t* row_index,
                                         const float* mix_src, int64_t ld_mix, const int64_t* mix_index, float c_act, float c_mix, float drop_p,
                                         uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int bits_relu_only, float* out_act,
                                         int64_t ld_act, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M >= 0 && N == 256 && K > 0 && drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_gemm_nn_store_rows_f32: bad size (N must be 256) or p");
  CB_CHECK_ARG(K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_store_rows_f32: size out of range");
  if (M == 0) return CB_OK;
  CB_CHECK_ARG(A && B && C && row_index && lda >= K && ldb >= N && ldc >= N && (!addend || ld_add >= N), CB_E_INVALID,
               "cb_gemm_nn_store_rows_f32: null pointer or leading dimension too small");
  CB_CHECK_ARG(cb_gemm_nn_store_rows_supported(A, lda, B, ldb, C, ldc, M, N, K) && (!addend || (al16(addend) && ld_add % 4 == 0)) &&
                   (!mix_src || (al16(mix_src) && ld_mix % 4 == 0 && ld_mix >= N)) && (!out_act || (al16(out_act) && ld_act % 4 == 0 && ld_act >= N)) &&
                   (!relu_bits || (uintptr_t)relu_bits % 8 == 0),
               CB_E_INVALID, "cb_gemm_nn_store_rows_f32: shape / alignment outside the fused form (cb_gemm_nn_store_rows_supported)");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, 1, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, 0};
  ep.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  ep.keep_scale = 1.f / (1.f - drop_p);
  ep.seed = seed; ep.seed_dev = seed_dev; ep.row0 = row0;
  ep.relu_bits_out = (unsigned long long*)relu_bits;
  ep.row_ids = row_index; ep.mix_src = mix_src; ep.ld_mix = ld_mix; ep.mix_index = mix_index; ep.c_act = c_act; ep.c_mix = c_mix;
  ep.bits_relu_only = bits_relu_only; ep.out_act = out_act; ep.ld_act = ld_act;
  return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, false, (hipStream_t)stream, ws, ws_bytes);
}

extern "C" int cb_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* C, int64_t M,
                              int64_t K1, int64_t K2, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M >= 0 && K1 >= 0 && K2 >= 0, CB_E_INVALID, "cb_gemm_tn_f32: negative size");
  CB_CHECK_ARG(K1 < (1 << 20) && K2 < (1 << 20), CB_E_RANGE, "cb_gemm_tn_f32: size out of range");
  if (K1 == 0 || K2 == 0) return CB_OK;
  CB_CHECK_ARG(C && (M == 0 || (A && G)) && lda >= K1 && ldg >= K2, CB_E_INVALID, "cb_gemm_tn_f32: null pointer or bad ld");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    CB_HIP(hipMemsetAsync(C, 0, (size_t)K1 * K2 * sizeof(float), st));
    return CB_OK;
  }
  CB_CHECK_ARG(ws && ws_bytes >= cb_gemm_tn_workspace_bytes(M, K1, K2), CB_E_WORKSPACE, "cb_gemm_tn_f32: workspace too small");
  int bm, bn;
  tn_tile(K1, K2, bm, bn);
  if (bm == 64) return launch_tn<1, 4>(A, lda, G, ldg, rowscale, C, M, K1, K2, (float*)ws, st);
  if (bm == 256) return launch_tn<4, 1>(A, lda, G, ldg, rowscale, C, M, K1, K2, (float*)ws, st);
  return launch_tn<2, 2>(A, lda, G, ldg, rowscale, C, M, K1, K2, (float*)ws, st);
}

// C = A^T @ dropout_{g_seed}(G): the weight gradient of the input Linear from the UNdropped features (mask regenerated while G is
// staged; the forward's cb_gemm_nn_indrop_drop2_f32 drew the same one).  Three-limb path only: cb_gemm_tn_gdrop_supported.
extern "C" int cb_gemm_tn_gdrop_supported(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2) {
  return use_limb3() && K2 % 4 == 0 && limb3_tn_eligible(A, lda, G, ldg, K1, K2) ? 1 : 0;
}

extern "C" int cb_gemm_tn_gdrop_f32(const float* A, int64_t lda, const float* G, int64_t ldg, float* C, int64_t M, int64_t K1, int64_t K2,
                                    float g_drop_p, uint64_t g_seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes,
                                    void* stream) {
  CB_CHECK_ARG(M >= 0 && K1 >= 0 && K2 >= 0 && g_drop_p > 0.f && g_drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_gemm_tn_gdrop_f32: bad size or p");
  CB_CHECK_ARG(K1 < (1 << 20) && K2 < (1 << 20), CB_E_RANGE, "cb_gemm_tn_gdrop_f32: size out of range");
  if (K1 == 0 || K2 == 0) return CB_OK;
  CB_CHECK_ARG(C && (M == 0 || (A && G)) && lda >= K1 && ldg >= K2, CB_E_INVALID, "cb_gemm_tn_gdrop_f32: null pointer or bad ld");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    CB_HIP(hipMemsetAsync(C, 0, (size_t)K1 * K2 * sizeof(float), st));
    return CB_OK;
  }
  CB_CHECK_ARG(cb_gemm_tn_gdrop_supported(A, lda, G, ldg, K1, K2), CB_E_INVALID, "cb_gemm_tn_gdrop_f32: operands outside the three-limb path");
  CB_CHECK_ARG(ws && ws_bytes >= cb_gemm_tn_workspace_bytes(M, K1, K2), CB_E_WORKSPACE, "cb_gemm_tn_gdrop_f32: workspace too small");
  const DropSpec gd{dropout_threshold(g_drop_p), 1.f / (1.f - g_drop_p), g_seed, seed_dev, row0, K2};
  int bm, bn;
  tn_tile(K1, K2, bm, bn);
  if (bm == 64) return launch_tn<1, 4>(A, lda, G, ldg, nullptr, C, M, K1, K2, (float*)ws, st, &gd);
  if (bm == 256) return launch_tn<4, 1>(A, lda, G, ldg, nullptr, C, M, K1, K2, (float*)ws, st, &gd);
  return launch_tn<2, 2>(A, lda, G, ldg, nullptr, C, M, K1, K2, (float*)ws, st, &gd);
}

// The input stage of the fused trunk's backward INSIDE the input Linear's weight gradient (autograd of GCN.py:104-110 and of the mixes res_tricks.py:23):
//   gy = (X0 > 0) * ( dropout_bwd_{g_seed}(g) + mfold )          [M, 256], never written
//   C  = gy^T @ dropout_{x_seed}(X)                               [256, K2]   (= layers_MLP[0].weight.grad)
//   colsum = column sums of gy                                    [256]       (= layers_MLP[0].bias.grad)
// g = dL/d dropout(X0) (the dX of the first GCNConv), mfold = the folded mix gradients (cb_spmm_csr_store_bwd_mix_f32), x0_bits = mask words of (X0 > 0)
// ([M][4] words), X = the UNdropped input features.  Replaces cb_trunk_input_bwd_multi_f32 + cb_gemm_tn_gdrop_f32: gy's 4 * 256 * M bytes are neither
// written nor re-read.  One 256 x 128 tile of eight wavefronts per row slab: 64 < K2 <= 128, K2 % 4 == 0, both dropouts with p > 0, M large enough for
// >= 256 slabs — cb_gemm_tn_instage_supported first.
static inline int instage_splits(int64_t M) { return tn_splits(M, 2); }

extern "C" int cb_gemm_tn_instage_supported(const float* g, const float* mfold, const float* X, int64_t ldx, int64_t M, int64_t K2) {
  return use_limb3() && K2 > 64 && K2 <= 128 && K2 % 4 == 0 && ldx % 4 == 0 && ldx >= K2 && ldx < (1 << 22) && al16(g) && al16(mfold) && al16(X) && M > 0 &&
                 instage_splits(M) >= 256
             ? 1
             : 0;
}

extern "C" size_t cb_gemm_tn_instage_workspace_bytes(int64_t M, int64_t K2) {
  if (M <= 0 || K2 <= 0) return 0;
  return (size_t)instage_splits(M) * (size_t)256 * (size_t)(K2 + 1) * sizeof(float);
}

extern "C" int cb_gemm_tn_instage_f32(const float* g, const float* mfold, const uint64_t* x0_bits, const float* X, int64_t ldx, float* C, float* colsum, int64_t M,
                                      int64_t K2, float g_drop_p, uint64_t g_seed, float x_drop_p, uint64_t x_seed, const uint64_t* seed_dev, int64_t row0, void* ws,
                                      size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(M > 0 && K2 > 0 && g_drop_p > 0.f && g_drop_p < 1.f && x_drop_p > 0.f && x_drop_p < 1.f && row0 >= 0, CB_E_INVALID,
               "cb_gemm_tn_instage_f32: bad size or p (both dropouts must be active)");
  CB_CHECK_ARG(g && mfold && x0_bits && X && C && colsum && (uintptr_t)x0_bits % 8 == 0, CB_E_INVALID, "cb_gemm_tn_instage_f32: null or misaligned pointer");
  CB_CHECK_ARG(cb_gemm_tn_instage_supported(g, mfold, X, ldx, M, K2), CB_E_INVALID, "cb_gemm_tn_instage_f32: shape / alignment outside the fused form (cb_gemm_tn_instage_supported)");
  CB_CHECK_ARG(ws && ws_bytes >= cb_gemm_tn_instage_workspace_bytes(M, K2), CB_E_WORKSPACE, "cb_gemm_tn_instage_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = instage_splits(M);
  int64_t rows_per_split = (M + nsplit - 1) / nsplit;
  rows_per_split = (rows_per_split + 31) / 32 * 32;
  const DropSpec xd{dropout_threshold(x_drop_p), 1.f / (1.f - x_drop_p), x_seed, seed_dev, row0, K2};
  const DropSpec gd{dropout_threshold(g_drop_p), 1.f / (1.f - g_drop_p), g_seed, seed_dev, row0, 256};
  float* partial = (float*)ws;
  float* cs_partial = partial + (size_t)nsplit * 256 * K2;
  const int rc = launch_tn_instage(g, mfold, x0_bits, X, ldx, partial, cs_partial, M, K2, nsplit, rows_per_split, st, xd, gd);
  if (rc != CB_OK) return rc;
  const int64_t n = 256 * K2;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)partial, nsplit, n, C);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, (const float*)cs_partial, nsplit, (int64_t)256, colsum);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

// C = act(rowscale * (dropout_{a_seed}(A) @ B) + addend + bias) with the dropout applied to A while it is staged (no dropped copy of A is
// written, kept or re-read) — the input Linear of the residual trunk (GCN.py:104-107: A = x) and the first GCNConv's transform
// (GCN.py:110 then :213,225,230-235: A = X0, the dropout in front of layer 0).  relu_bits (may be NULL; N == 256 and relu): [M][4] mask words
// of (C > 0).  Bit-identical to cb_dropout_f32 followed by cb_gemm_nn_f32.  Only where the fused form exists: cb_gemm_nn_indrop_supported
// (pass C for C2).
extern "C" int cb_gemm_nn_indrop_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                     const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu, float a_drop_p,
                                     uint64_t a_seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && a_drop_p > 0.f && a_drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_gemm_nn_indrop_f32: bad size or p");
  CB_CHECK_ARG(!relu_bits || (N == 256 && relu && (uintptr_t)relu_bits % 8 == 0), CB_E_INVALID,
               "cb_gemm_nn_indrop_f32: mask words of the ReLU exist for N == 256 with relu only");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + 63) / 64 < (1 << 24), CB_E_RANGE, "cb_gemm_nn_indrop_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && A && B && lda >= K && ldb >= N && ldc >= N && (!addend || ld_add >= N), CB_E_INVALID, "cb_gemm_nn_indrop_f32: null pointer or bad ld");
  CB_CHECK_ARG(cb_gemm_nn_indrop_supported(A, lda, B, ldb, C, ldc, C, ldc, M, N, K) && (!addend || (al16(addend) && ld_add % 4 == 0)), CB_E_INVALID,
               "cb_gemm_nn_indrop_f32: shape / alignment outside the fused form (cb_gemm_nn_indrop_supported)");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, relu, nullptr, 0, 0u, 1.f, 0ull, nullptr, 0, gemm_nt_store(M, N)};
  ep.adrop = DropSpec{dropout_threshold(a_drop_p), 1.f / (1.f - a_drop_p), a_seed, seed_dev, row0, K};
  ep.relu_bits_out = (unsigned long long*)relu_bits;
  return launch_nn_limb3(A, lda, B, ldb, C, ldc, M, N, K, ep, false, (hipStream_t)stream, nullptr, 0);
}

// C = dropout_{a_seed}(A)^T @ (rowscale * G): the weight gradient of the first GCNConv (autograd of GCN.py:225 behind the dropout of :110)
// from the UNdropped X0 — the mask cb_gemm_nn_indrop_f32 drew is regenerated while A is staged.  cb_gemm_tn_adrop_supported first.
extern "C" int cb_gemm_tn_adrop_supported(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2) {
  int bm, bn;
  tn_tile(K1, K2, bm, bn);
  return use_limb3() && K1 % 4 == 0 && bm == 128 && limb3_tn_eligible(A, lda, G, ldg, K1, K2) ? 1 : 0;
}

extern "C" int cb_gemm_tn_adrop_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* C, int64_t M, int64_t K1,
                                    int64_t K2, float a_drop_p, uint64_t a_seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes,
                                    void* stream) {
  CB_CHECK_ARG(M >= 0 && K1 >= 0 && K2 >= 0 && a_drop_p > 0.f && a_drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_gemm_tn_adrop_f32: bad size or p");
  CB_CHECK_ARG(K1 < (1 << 20) && K2 < (1 << 20), CB_E_RANGE, "cb_gemm_tn_adrop_f32: size out of range");
  if (K1 == 0 || K2 == 0) return CB_OK;
  CB_CHECK_ARG(C && (M == 0 || (A && G)) && lda >= K1 && ldg >= K2, CB_E_INVALID, "cb_gemm_tn_adrop_f32: null pointer or bad ld");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    CB_HIP(hipMemsetAsync(C, 0, (size_t)K1 * K2 * sizeof(float), st));
    return CB_OK;
  }
  CB_CHECK_ARG(cb_gemm_tn_adrop_supported(A, lda, G, ldg, K1, K2), CB_E_INVALID, "cb_gemm_tn_adrop_f32: operands outside the three-limb 128-row tile path");
  CB_CHECK_ARG(ws && ws_bytes >= cb_gemm_tn_workspace_bytes(M, K1, K2), CB_E_WORKSPACE, "cb_gemm_tn_adrop_f32: workspace too small");
  const DropSpec ad{dropout_threshold(a_drop_p), 1.f / (1.f - a_drop_p), a_seed, seed_dev, row0, K1};
  return launch_tn<2, 2>(A, lda, G, ldg, rowscale, C, M, K1, K2, (float*)ws, st, nullptr, &ad);
}
