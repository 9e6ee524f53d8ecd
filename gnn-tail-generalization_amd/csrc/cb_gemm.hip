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
// Tiling: 256 threads = 4 wavefronts (2x2), block tile 128x128, K step 16, each wavefront owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).  Both operands sit k-major in LDS
// ([k][128 + 4 pad] floats): a fragment read is 32 consecutive floats per half-wave (conflict-free
// ds_read_b32), A is transposed on the way in (global row-major [m][k] -> LDS [k][m]).  Global loads
// of tile t+1 are issued before the MFMAs of tile t and stored to the other LDS buffer after them:
// one barrier per K step.
#include "cb_common.h"

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;  // LDT: padded LDS row (floats)

struct GemmEpilogue {
  const float* rowscale;  // [M] or null
  const float* addend;    // [M, ld_add] or null
  int64_t ld_add;
  const float* bias;      // [N] or null
  int relu;
};

__device__ __forceinline__ void mfma_tile_step(const float* __restrict__ As, const float* __restrict__ Bs, int wr, int wc,
                                               int lane, f32x16 (&acc)[2][2]) {
  const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    const float* ar = As + (kk + kh) * LDT + wr * 64 + l31;
    const float* br = Bs + (kk + kh) * LDT + wc * 64 + l31;
    const float a0 = ar[0], a1 = ar[32];
    const float b0 = br[0], b1 = br[32];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// ---- operand staging -----------------------------------------------------------------------
// "row" operand: global [rows of the tile dimension][k], 16 k per tile row -> needs the transpose into [k][m]
struct RowFrag {
  float v[2][4];
};
// thread t: m = t/4 + 64*j (j = 0,1), k quad = t%4
template <bool ALIGNED>
__device__ __forceinline__ void load_rowmajor(RowFrag& f, const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M, int k0,
                                              int K, int t) {
  const int kq = (t & 3) * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t m = m0 + (t >> 2) + 64 * j;
    const int k = k0 + kq;
    if (ALIGNED && m < M && k + 4 <= K) {
      const float4 x = *reinterpret_cast<const float4*>(A + m * lda + k);
      f.v[j][0] = x.x; f.v[j][1] = x.y; f.v[j][2] = x.z; f.v[j][3] = x.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) f.v[j][i] = (m < M && k + i < K) ? A[m * lda + k + i] : 0.f;
    }
  }
}
__device__ __forceinline__ void store_rowmajor_T(const RowFrag& f, float* __restrict__ S, int t) {
  const int kq = (t & 3) * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = (t >> 2) + 64 * j;
#pragma unroll
    for (int i = 0; i < 4; ++i) S[(kq + i) * LDT + m] = f.v[j][i];
  }
}

// "k-major" operand: global [k][n] with n contiguous (B of NN; both operands of TN): straight copy
struct KFrag {
  float4 v[2];
};
// thread t: k = t/32 + 8*j, n quad = t%32
template <bool ALIGNED>
__device__ __forceinline__ void load_kmajor(KFrag& f, const float* __restrict__ B, int64_t ldb, int64_t k0, int64_t Kdim, int n0,
                                            int N, int t, const float* __restrict__ kscale) {
  const int n = n0 + (t & 31) * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t k = k0 + (t >> 5) + 8 * j;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < Kdim) {
      if (ALIGNED && n + 4 <= N) {
        x = *reinterpret_cast<const float4*>(B + k * ldb + n);
      } else {
        if (n + 0 < N) x.x = B[k * ldb + n + 0];
        if (n + 1 < N) x.y = B[k * ldb + n + 1];
        if (n + 2 < N) x.z = B[k * ldb + n + 2];
        if (n + 3 < N) x.w = B[k * ldb + n + 3];
      }
      if (kscale) {
        const float s = kscale[k];
        x.x *= s; x.y *= s; x.z *= s; x.w *= s;
      }
    }
    f.v[j] = x;
  }
}
__device__ __forceinline__ void store_kmajor(const KFrag& f, float* __restrict__ S, int t) {
#pragma unroll
  for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(S + ((t >> 5) + 8 * j) * LDT + (t & 31) * 4) = f.v[j];
}

// ---- NN ------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_gemm_nn(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                 float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, GemmEpilogue ep,
                                                 int n_row_blocks, int n_col_blocks) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LDT];
  auto As = [&](int b) { return smem + b * (BK * LDT); };
  auto Bs = [&](int b) { return smem + (2 + b) * (BK * LDT); };
  // XCD-aware tile order: the column blocks of one row block get ids congruent mod 8, i.e. the same
  // XCD / L2 under the observed round-robin dispatch, so the A rows are fetched from HBM once.
  const int per_group = 8 * n_col_blocks;
  const int grp = blockIdx.x / per_group, r = blockIdx.x % per_group;
  const int row_blk = grp * 8 + (r & 7), col_blk = r >> 3;
  if (row_blk >= n_row_blocks) return;
  const int64_t m0 = (int64_t)row_blk * BM;
  const int n0 = col_blk * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (K + BK - 1) / BK;
  RowFrag fa;
  KFrag fb;
  load_rowmajor<ALIGNED>(fa, A, lda, m0, M, 0, K, t);
  load_kmajor<ALIGNED>(fb, B, ldb, 0, K, n0, N, t, nullptr);
  store_rowmajor_T(fa, As(0), t);
  store_kmajor(fb, Bs(0), t);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_rowmajor<ALIGNED>(fa, A, lda, m0, M, (kt + 1) * BK, K, t);
      load_kmajor<ALIGNED>(fb, B, ldb, (int64_t)(kt + 1) * BK, K, n0, N, t, nullptr);
    }
    mfma_tile_step(As(cur), Bs(cur), wr, wc, lane, acc);
    if (kt + 1 < nk) {
      store_rowmajor_T(fa, As(cur ^ 1), t);
      store_kmajor(fb, Bs(cur ^ 1), t);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = n0 + wc * 64 + tj * 32 + l31;
      if (n >= N) continue;
      const float bv = ep.bias ? ep.bias[n] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int64_t m = m0 + wr * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[ti][tj][reg];
        if (ep.rowscale) v *= ep.rowscale[m];
        if (ep.addend) v += ep.addend[m * ep.ld_add + n];
        v += bv;
        if (ep.relu) v = fmaxf(v, 0.f);
        C[m * ldc + n] = v;
      }
    }
  }
}

// ---- TN ------------------------------------------------------------------------------------
// partial[split][K1][K2]: block (tile i, tile j, split s) reduces rows [s*rows_per_split, ...)
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_gemm_tn(const float* __restrict__ A, int64_t lda, const float* __restrict__ G, int64_t ldg,
                                                 const float* __restrict__ rowscale, float* __restrict__ partial, int64_t M, int K1,
                                                 int K2, int64_t rows_per_split, int tiles_j) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LDT];
  auto As = [&](int b) { return smem + b * (BK * LDT); };
  auto Bs = [&](int b) { return smem + (2 + b) * (BK * LDT); };
  const int tile = blockIdx.x, split = blockIdx.y;
  const int i0 = (tile / tiles_j) * BM, j0 = (tile % tiles_j) * BN;
  const int64_t r_begin = (int64_t)split * rows_per_split;
  const int64_t r_end = min(M, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int64_t nk = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;
  KFrag fa, fb;
  if (nk > 0) {
    load_kmajor<ALIGNED>(fa, A, lda, r_begin, r_end, i0, K1, t, nullptr);
    load_kmajor<ALIGNED>(fb, G, ldg, r_begin, r_end, j0, K2, t, rowscale);
    store_kmajor(fa, As(0), t);
    store_kmajor(fb, Bs(0), t);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) {
      load_kmajor<ALIGNED>(fa, A, lda, r_begin + (kt + 1) * BK, r_end, i0, K1, t, nullptr);
      load_kmajor<ALIGNED>(fb, G, ldg, r_begin + (kt + 1) * BK, r_end, j0, K2, t, rowscale);
    }
    mfma_tile_step(As(cur), Bs(cur), wr, wc, lane, acc);
    if (kt + 1 < nk) {
      store_kmajor(fa, As(cur ^ 1), t);
      store_kmajor(fb, Bs(cur ^ 1), t);
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
  float s = 0.f;
  for (int p = 0; p < nsplit; ++p) s += partial[(int64_t)p * n + i];
  out[i] = s;
}

static inline int tn_splits(int64_t M, int tiles) {
  // enough blocks for ~4 per CU, each with at least 64 K-steps of work
  int64_t want = (256 * 4 + tiles - 1) / tiles;
  int64_t max_by_work = (M + BK * 64 - 1) / (BK * 64);
  int64_t s = want < max_by_work ? want : max_by_work;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace cb

using namespace cb;

extern "C" int cb_gemm_nn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu,
                              void* stream) {
  CB_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, CB_E_INVALID, "cb_gemm_nn_f32: negative size");
  CB_CHECK_ARG(N < (1 << 24) && K < (1 << 24) && (M + BM - 1) / BM < (1 << 24), CB_E_RANGE, "cb_gemm_nn_f32: size out of range");
  if (M == 0 || N == 0) return CB_OK;
  CB_CHECK_ARG(C && (K == 0 || (A && B)) && lda >= K && ldb >= N && ldc >= N && (!addend || ld_add >= N), CB_E_INVALID,
               "cb_gemm_nn_f32: null pointer or leading dimension too small");
  GemmEpilogue ep{rowscale, addend, ld_add, bias, relu};
  const int nrb = (int)((M + BM - 1) / BM), ncb = (int)((N + BN - 1) / BN);
  const int64_t groups = (nrb + 7) / 8;
  const dim3 grid((unsigned)(groups * 8 * ncb));
  const bool aligned = al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0;
  hipStream_t st = (hipStream_t)stream;
  if (aligned)
    hipLaunchKernelGGL((k_gemm_nn<true>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb);
  else
    hipLaunchKernelGGL((k_gemm_nn<false>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_gemm_tn_workspace_bytes(int64_t M, int64_t K1, int64_t K2) {
  if (M <= 0 || K1 <= 0 || K2 <= 0) return 0;
  const int tiles = (int)(((K1 + BM - 1) / BM) * ((K2 + BN - 1) / BN));
  return (size_t)tn_splits(M, tiles) * (size_t)K1 * (size_t)K2 * sizeof(float);
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
  const int tiles_i = (int)((K1 + BM - 1) / BM), tiles_j = (int)((K2 + BN - 1) / BN);
  const int nsplit = tn_splits(M, tiles_i * tiles_j);
  int64_t rows_per_split = (M + nsplit - 1) / nsplit;
  rows_per_split = (rows_per_split + BK - 1) / BK * BK;
  const bool aligned = al16(A) && al16(G) && lda % 4 == 0 && ldg % 4 == 0;
  const dim3 grid((unsigned)(tiles_i * tiles_j), (unsigned)nsplit);
  if (aligned)
    hipLaunchKernelGGL((k_gemm_tn<true>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, (float*)ws, M, (int)K1, (int)K2,
                       rows_per_split, tiles_j);
  else
    hipLaunchKernelGGL((k_gemm_tn<false>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, (float*)ws, M, (int)K1, (int)K2,
                       rows_per_split, tiles_j);
  CB_LAUNCH_CHECK();
  const int64_t n = K1 * K2;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, nsplit, n, C);
  CB_LAUNCH_CHECK();
  return CB_OK;
}
