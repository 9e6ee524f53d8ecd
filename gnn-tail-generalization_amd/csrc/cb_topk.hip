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

// K-th best of a list without indexing it at run time
template <int K_>
__device__ __forceinline__ float best_kth_v(const Cand (&best)[K_], int K) {
  float r = best[0].v;
#pragma unroll
  for (int s = 1; s < K_; ++s) r = (s == K - 1) ? best[s].v : r;
  return r;
}
template <int K_>
__device__ __forceinline__ int best_kth_i(const Cand (&best)[K_], int K) {
  int r = best[0].i;
#pragma unroll
  for (int s = 1; s < K_; ++s) r = (s == K - 1) ? best[s].i : r;
  return r;
}

// insert (v,i) into the descending list best[0..K): a chain of compare-and-swaps with compile-time indices (the element carried
// along is always the smaller one, so the list stays sorted and the K-th falls off the end).  Register-only: an insertion point
// computed at run time indexes the array dynamically, which puts `best` into scratch memory — every fold then waits on scratch
// round trips (measured: 464 -> 331 ms at arxiv scale only by filtering, the rest of the gap was this).
template <int K_>
__device__ __forceinline__ void push(Cand (&best)[K_], int K, float v, int i) {
  if (!better(v, i, best_kth_v(best, K), best_kth_i(best, K))) return;
#pragma unroll
  for (int s = 0; s < K_; ++s) {
    if (s < K) {
      const bool b = better(v, i, best[s].v, best[s].i);
      const float tv = best[s].v;
      const int ti = best[s].i;
      best[s].v = b ? v : tv;
      best[s].i = b ? i : ti;
      v = b ? tv : v;
      i = b ? ti : i;
    }
  }
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

// The fold above costs as much as the K loop of a 128 x 128 x 768 tile (measured: 464 ms with it, 240 ms without, arxiv scale):
// it stages and scans all 16 384 scores of every tile although, once a query has seen n teacher rows, a new score enters its top K
// with probability K / n.  Filtered fold: every lane compares its 64 accumulator values with the K-th best value of their rows
// (s_thr) where they sit — no staging — and the few that pass are appended to a small LDS queue; one thread per query row then
// merges its queue entries.  The candidate set is exactly "scores not worse than the current K-th best", insertion uses the same
// strict total order (value, then index), so the lists equal the unfiltered fold's whatever order the queue was filled in.  A
// tile with more than QCAP candidates (the first tiles of a block) takes the unfiltered fold.
constexpr int QCAP = 1024;
struct QCand {
  float v;
  int i, row;
};

struct FoldState {
  Cand (*cand)[8][KMAX];     // [32][8][KMAX] scratch of the unfiltered fold
  Cand (*run)[KMAX];         // [128][KMAX] running lists
  float* thr;                // [128] value of the K-th best of each row (-inf while the list is short)
  QCand* q;                  // [QCAP]
  int* qn;
};

__device__ __forceinline__ void fold_filtered(f32x16 (&acc)[2][2], float* __restrict__ Cs, const FoldState& fs, int n0, int N, int K, int t) {
  const int lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // (entry: a block barrier separates this call from the last writes to thr / qn == 0)
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int rbase = wr * 64 + ti * 32 + 8 * q4 + 4 * lh;
      const float4 th = *reinterpret_cast<const float4*>(fs.thr + rbase);
      const float thv[4] = {th.x, th.y, th.z, th.w};
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const float v = acc[ti][tj][4 * q4 + r4];
          const int n = n0 + wc * 64 + tj * 32 + l31;
          if (v >= thv[r4] && n < N) {
            const int slot = atomicAdd(fs.qn, 1);
            if (slot < QCAP) fs.q[slot] = QCand{v, n, rbase + r4};
          }
        }
    }
  __syncthreads();
  const int qn = *fs.qn;
  if (qn > QCAP) {                      // block-uniform: too many candidates for the queue -> the unfiltered fold of the whole tile
    fold_tile(acc, Cs, *reinterpret_cast<Cand (*)[32][8][KMAX]>(fs.cand), *reinterpret_cast<Cand (*)[128][KMAX]>(fs.run), n0, N, K, t);
  } else if (qn > 0 && t < 128) {
    Cand best[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) best[k] = fs.run[t][k];
    bool touched = false;
    for (int c = 0; c < qn; ++c) {
      const QCand x = fs.q[c];
      if (x.row == t) { push<KMAX>(best, K, x.v, x.i); touched = true; }
    }
    if (touched) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) fs.run[t][k] = best[k];
    }
  }
  if (qn > 0) {
    __syncthreads();
    if (t < 128) fs.thr[t] = fs.run[t][K - 1].v;
    if (t == 0) *fs.qn = 0;
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
  __shared__ __attribute__((aligned(16))) float s_thr[BM];
  __shared__ int s_qn;
  static_assert(sizeof(QCand) * QCAP <= sizeof(s_cand), "the candidate queue lives in the unfiltered fold's scratch (never used together)");
  const FoldState fs{s_cand, s_run, s_thr, reinterpret_cast<QCand*>(&s_cand[0][0][0]), &s_qn};
  auto As = [&](int b) { return smem + b * (BK * LDA); };
  auto Bs = [&](int b) { return smem + 2 * BK * LDA + b * (BK * LDB); };
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int split = blockIdx.y;
  for (int i = t; i < BM * KMAX; i += 256) {
    s_run[i / KMAX][i % KMAX].v = -INFINITY;
    s_run[i / KMAX][i % KMAX].i = -1;
  }
  if (t < BM) s_thr[t] = -INFINITY;
  if (t == 0) s_qn = 0;
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
    fold_filtered(acc, smem, fs, n0, N, K, t);
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
  __shared__ __attribute__((aligned(16))) float s_thr[BM];
  __shared__ int s_qn;
  static_assert(sizeof(QCand) * QCAP <= sizeof(s_cand), "the candidate queue lives in the unfiltered fold's scratch (never used together)");
  const FoldState fs{s_cand, s_run, s_thr, reinterpret_cast<QCand*>(&s_cand[0][0][0]), &s_qn};
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int split = blockIdx.y;
  for (int i = t; i < BM * KMAX; i += 256) {
    s_run[i / KMAX][i % KMAX].v = -INFINITY;
    s_run[i / KMAX][i % KMAX].i = -1;
  }
  if (t < BM) s_thr[t] = -INFINITY;
  if (t == 0) s_qn = 0;
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
    if (K > 0) fold_filtered(acc, reinterpret_cast<float*>(smem), fs, n0, N, K, t);
    else if (acc[0][0][0] == 12345.678f) s_run[0][0].v = 1.f;      // (K = -1: measurement launch without the fold, see cb_topk_replace_f32)
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
