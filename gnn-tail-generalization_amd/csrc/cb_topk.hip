// Teacher -> student hand-off: the "virtual neighbour" lookup of Cold Brew's SEMLP,
// MLP_model/__init__.py:143-156 `SEMLP.replacement` — a per-node Python loop of [1,N] matmuls + argsort in the reference:
//     s_j   = <q_i, T_j>  for all N teacher embeddings        (:150)
//     sel   = the K largest s_j                                (:151-152, K = --SEMLP_topK_2_replace)
//     out_i = sum_k softmax(s_sel)_k * T_sel_k                 (:153-154)
// Here: one MFMA GEMM sweep that never materialises the B x N score matrix.  Each block owns 128 queries and a slab of
// teacher rows; per 128x128 score tile the epilogue folds the tile into a running top-K per query held in LDS; a second
// kernel merges the slabs' lists, applies the softmax and combines the K selected teacher rows.  The scores come from the
// three-limb bf16-MFMA core of cb_gemm_limb.hip (fp32 operands split exactly, error at the level of an fp32 GEMM; both
// operands are k-contiguous "row" operands) when the operands allow float4 access, else from the fp32-input MFMA core of
// cb_gemm.hip (operands transposed into LDS).
// Bound: MFMA (2*B*N*D flop, x6 bf16 passes on the limb path).  Ordering of ties: larger score first, then larger index (what an ascending stable argsort
// followed by [-K:] selects).
#include <stdlib.h>

#include "cb_common.h"
#include "cb_gemm_core.h"
#include "cb_limb_core.h"

namespace cb {

constexpr int KMAX = 8;

struct Cand {
  float v;
  int i;
};

__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i > j); }

// insert (v,i) into the descending list best[0..K)
template <int K_>
__device__ __forceinline__ void push(Cand (&best)[K_], int K, float v, int i) {
  if (!better(v, i, best[K - 1].v, best[K - 1].i)) return;
  int p = K - 1;
#pragma unroll
  for (int s = K_ - 1; s > 0; --s) {
    if (s < K && s <= p && better(v, i, best[s - 1].v, best[s - 1].i)) {
      best[s] = best[s - 1];
      p = s - 1;
    }
  }
  best[p].v = v;
  best[p].i = i;
}

// fold one 128x128 score tile (accumulators of the four wavefronts) into the running top-K lists, 32 query rows per pass
__device__ __forceinline__ void fold_tile(f32x16 (&acc)[2][2], float* __restrict__ Cs, Cand (&s_cand)[32][8][KMAX],
                                          Cand (&s_run)[128][KMAX], int n0, int N, int K, int t) {
  constexpr int LDB = 128 + 4;
  const int lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int pass = 0; pass < 4; ++pass) {
    const int wr_sel = pass >> 1, ti = pass & 1;
    if (wr == wr_sel) {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Cs[((reg & 3) + 8 * (reg >> 2) + 4 * lh) * LDB + wc * 64 + tj * 32 + l31] = acc[ti][tj][reg];
    }
    __syncthreads();
    {
      const int row = t >> 3, seg = t & 7;
      Cand best[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { best[k].v = -INFINITY; best[k].i = -1; }
      for (int c = 0; c < 16; ++c) {
        const int col = seg * 16 + c, n = n0 + col;
        if (n < N) push<KMAX>(best, K, Cs[row * LDB + col], n);
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) s_cand[row][seg][k] = best[k];
    }
    __syncthreads();
    if ((t & 7) == 0) {
      const int row = t >> 3, grow = wr_sel * 64 + ti * 32 + row;
      Cand best[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) best[k] = s_run[grow][k];
      for (int seg = 0; seg < 8; ++seg)
        for (int k = 0; k < K; ++k) {
          const Cand c = s_cand[row][seg][k];
          if (c.i >= 0) push<KMAX>(best, K, c.v, c.i);
        }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) s_run[grow][k] = best[k];
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_topk_scores(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ T, int64_t ldt,
                                                     int64_t B, int N, int D, int K, int tiles_per_split, int n_col_tiles,
                                                     Cand* __restrict__ partial, int aligned) {
  using TL = Tile<2, 2, BK, 2>;
  constexpr int BM = TL::BM, BN = TL::BN, LDA = TL::LDA, LDB = TL::LDB;
  __shared__ __attribute__((aligned(16))) float smem[TL::SMEM_FLOATS];
  __shared__ Cand s_cand[32][8][KMAX];
  __shared__ Cand s_run[BM][KMAX];
  auto As = [&](int b) { return smem + b * (BK * LDA); };
  auto Bs = [&](int b) { return smem + 2 * BK * LDA + b * (BK * LDB); };
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int split = blockIdx.y;
  for (int i = t; i < BM * KMAX; i += 256) {
    s_run[i / KMAX][i % KMAX].v = -INFINITY;
    s_run[i / KMAX][i % KMAX].i = -1;
  }
  __syncthreads();
  const int ct_begin = split * tiles_per_split, ct_end = min(n_col_tiles, ct_begin + tiles_per_split);
  const int nk = (D + BK - 1) / BK;
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    const int n0 = ct * BN;
    f32x16 acc[2][2];
    zero_acc<2>(acc);
    RowFrag<BM, BK> fa;
    RowFrag<BN, BK> fb;
    if (aligned) {
      load_rowmajor<true, BM, BK>(fa, Q, ldq, m0, B, 0, D, t);
      load_rowmajor<true, BN, BK>(fb, T, ldt, n0, N, 0, D, t);
    } else {
      load_rowmajor<false, BM, BK>(fa, Q, ldq, m0, B, 0, D, t);
      load_rowmajor<false, BN, BK>(fb, T, ldt, n0, N, 0, D, t);
    }
    store_rowmajor_T<BM, BK>(fa, As(0), t);
    store_rowmajor_T<BN, BK>(fb, Bs(0), t);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
        if (aligned) {
          load_rowmajor<true, BM, BK>(fa, Q, ldq, m0, B, (kt + 1) * BK, D, t);
          load_rowmajor<true, BN, BK>(fb, T, ldt, n0, N, (kt + 1) * BK, D, t);
        } else {
          load_rowmajor<false, BM, BK>(fa, Q, ldq, m0, B, (kt + 1) * BK, D, t);
          load_rowmajor<false, BN, BK>(fb, T, ldt, n0, N, (kt + 1) * BK, D, t);
        }
      }
      mfma_tile_step<LDA, LDB, BK, 2>(As(cur), Bs(cur), wr, wc, lane, acc);
      if (kt + 1 < nk) {
        store_rowmajor_T<BM, BK>(fa, As(cur ^ 1), t);
        store_rowmajor_T<BN, BK>(fb, Bs(cur ^ 1), t);
      }
      __syncthreads();
    }
    fold_tile(acc, smem, s_cand, s_run, n0, N, K, t);
  }
  for (int i = t; i < BM * K; i += 256) {
    const int row = i / K, k = i % K;
    const int64_t m = m0 + row;
    if (m < B) partial[((int64_t)split * B + m) * K + k] = s_run[row][k];
  }
}

// Same sweep on the three-limb core: Q rows and T rows are both k-contiguous "row" operands.
__global__ void __launch_bounds__(256, 2) k_topk_scores_l3(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ T,
                                                           int64_t ldt, int64_t B, int N, int D, int K, int tiles_per_split,
                                                           int n_col_tiles, Cand* __restrict__ partial) {
  using OA = RowOperand<128>;
  using OB = RowOperand<128>;
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * (OA::BYTES + OB::BYTES)];
  static_assert(32 * (BN + 4) * 4 <= 2 * (OA::BYTES + OB::BYTES), "fold staging must fit");
  __shared__ Cand s_cand[32][8][KMAX];
  __shared__ Cand s_run[BM][KMAX];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int split = blockIdx.y;
  for (int i = t; i < BM * KMAX; i += 256) {
    s_run[i / KMAX][i % KMAX].v = -INFINITY;
    s_run[i / KMAX][i % KMAX].i = -1;
  }
  __syncthreads();
  OA oa;
  oa.init(ldq, B - m0, t);
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  const uint32_t baddr[2] = {OB::frag_addr(wc * 64, lane), OB::frag_addr(wc * 64 + 32, lane)};
  const int ct_begin = split * tiles_per_split, ct_end = min(n_col_tiles, ct_begin + tiles_per_split);
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    const int n0 = ct * BN;
    OB ob;
    ob.init(ldt, N - n0, t);
    f32x16 acc[2][2];
    zero_acc_n<2>(acc);
    limb_k_loop<2, 1, OA, OB>(oa, ob, smem, Q + m0 * ldq, KS, ldq, T + (int64_t)n0 * ldt, KS, ldt, nullptr, D, aaddr, baddr, acc, t);
    fold_tile(acc, reinterpret_cast<float*>(smem), s_cand, s_run, n0, N, K, t);
  }
  for (int i = t; i < BM * K; i += 256) {
    const int row = i / K, k = i % K;
    const int64_t m = m0 + row;
    if (m < B) partial[((int64_t)split * B + m) * K + k] = s_run[row][k];
  }
}

// One wavefront per query: merge the slabs' lists, softmax over the K winners (ascending order, as argsort()[-K:] yields
// them), combine the selected teacher rows.
__global__ void __launch_bounds__(256) k_topk_finish(const Cand* __restrict__ partial, int n_splits, int64_t B, int K,
                                                     const float* __restrict__ T, int64_t ldt, int D, float* __restrict__ out,
                                                     int* __restrict__ out_idx, float* __restrict__ out_w) {
  const int lane = threadIdx.x & 63;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= B) return;
  const int total = n_splits * K;
  Cand win[KMAX];
  float lv = -INFINITY;   // the last winner: candidates strictly "worse" than it remain eligible
  int li = INT32_MAX;
  for (int r = 0; r < K; ++r) {
    Cand mine{-INFINITY, -1};
    for (int c = lane; c < total; c += 64) {
      const Cand x = partial[((int64_t)(c / K) * B + q) * K + (c % K)];
      if (x.i < 0) continue;
      const bool eligible = (r == 0) || better(lv, li, x.v, x.i);
      if (eligible && better(x.v, x.i, mine.v, mine.i)) mine = x;
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(mine.v, off);
      const int oi = __shfl_xor(mine.i, off);
      if (better(ov, oi, mine.v, mine.i)) { mine.v = ov; mine.i = oi; }
    }
    win[r] = mine;
    lv = mine.v;
    li = mine.i;
  }
  // win[] is descending; the reference's order is ascending (sortidx[-K:])
  const float mx = win[0].v;
  float den = 0.f, wgt[KMAX];
  for (int r = K - 1; r >= 0; --r) { wgt[r] = (win[r].i >= 0) ? expf(win[r].v - mx) : 0.f; den += wgt[r]; }
  for (int r = 0; r < K; ++r) wgt[r] = den > 0.f ? wgt[r] / den : 0.f;
  for (int c = lane; c < D; c += 64) {
    float s = 0.f;
    for (int r = K - 1; r >= 0; --r)
      if (win[r].i >= 0) s += wgt[r] * T[(int64_t)win[r].i * ldt + c];
    out[q * D + c] = s;
  }
  if (lane == 0) {
    for (int r = 0; r < K; ++r) {   // ascending order, like `select`
      if (out_idx) out_idx[q * K + r] = win[K - 1 - r].i;
      if (out_w) out_w[q * K + r] = wgt[K - 1 - r];
    }
  }
}

static void topk_geometry(int64_t B, int64_t N, int& n_row_blocks, int& n_col_tiles, int& n_splits, int& tiles_per_split) {
  n_row_blocks = (int)((B + 127) / 128);
  n_col_tiles = (int)((N + 127) / 128);
  int want = (1024 + n_row_blocks - 1) / n_row_blocks;   // ~4 blocks per CU
  if (want < 1) want = 1;
  if (want > n_col_tiles) want = n_col_tiles;
  if (want > 256) want = 256;
  tiles_per_split = (n_col_tiles + want - 1) / want;
  n_splits = (n_col_tiles + tiles_per_split - 1) / tiles_per_split;
}

}  // namespace cb

using namespace cb;

extern "C" size_t cb_topk_replace_workspace_bytes(int64_t B, int64_t N, int64_t K) {
  if (B <= 0 || N <= 0 || K <= 0) return 0;
  int rb, ctl, ns, tps;
  topk_geometry(B, N, rb, ctl, ns, tps);
  return (size_t)ns * (size_t)B * (size_t)K * sizeof(Cand);
}

extern "C" int cb_topk_replace_f32(const float* q, int64_t ldq, const float* t, int64_t ldt, int64_t B, int64_t N, int64_t D, int32_t K,
                                   float* out, int32_t* out_idx, float* out_w, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(B >= 0 && N > 0 && D > 0 && K >= 1 && K <= KMAX && K <= N, CB_E_INVALID,
               "cb_topk_replace_f32: bad size (1 <= K <= min(%d, N) required)", KMAX);
  CB_CHECK_ARG(N < INT32_MAX && D < (1 << 24), CB_E_RANGE, "cb_topk_replace_f32: size out of range");
  if (B == 0) return CB_OK;
  CB_CHECK_ARG(q && t && out && ldq >= D && ldt >= D, CB_E_INVALID, "cb_topk_replace_f32: null pointer or bad ld");
  CB_CHECK_ARG(ws && ws_bytes >= cb_topk_replace_workspace_bytes(B, N, K), CB_E_WORKSPACE, "cb_topk_replace_f32: workspace too small");
  int rb, ctl, ns, tps;
  topk_geometry(B, N, rb, ctl, ns, tps);
  hipStream_t st = (hipStream_t)stream;
  const int aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)t % 16 == 0) && ldq % 4 == 0 && ldt % 4 == 0;
  static const bool plain = getenv("CB_GEMM_PLAIN_F32") != nullptr;
  if (aligned && !plain && D % 4 == 0 && ldq < (1 << 22) && ldt < (1 << 22))
    hipLaunchKernelGGL(k_topk_scores_l3, dim3((unsigned)rb, (unsigned)ns), dim3(256), 0, st, q, ldq, t, ldt, B, (int)N, (int)D, (int)K,
                       tps, ctl, (Cand*)ws);
  else
    hipLaunchKernelGGL(k_topk_scores, dim3((unsigned)rb, (unsigned)ns), dim3(256), 0, st, q, ldq, t, ldt, B, (int)N, (int)D, (int)K,
                       tps, ctl, (Cand*)ws, aligned);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_topk_finish, dim3((unsigned)((B * 64 + 255) / 256)), dim3(256), 0, st, (const Cand*)ws, ns, B, (int)K, t, ldt,
                     (int)D, out, out_idx, out_w);
  CB_LAUNCH_CHECK();
  return CB_OK;
}
