// fp32 contractions on the bf16 matrix cores by exact operand decomposition ("three-limb" GEMM).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA (2.5 PF/s vs 157 TF/s dense).  An fp32
// value has a 24-bit significand and a bf16 keeps 8 significant bits with the full fp32 exponent range, so taking three
// times the round-to-nearest bf16 of what is left gives   a = a_hi + a_mid + a_lo   EXACTLY (every residual subtraction is
// exact; the last residual fits 8 bits), with |a_mid| <= 2^-8 |a| and |a_lo| <= 2^-17 |a|.  Then
//     a*b = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) + [mid*lo + lo*mid + lo*lo]
// and the bracket is below 2^-24 |a*b| — half an ulp of the fp32 product, without bias (tests/test_host_logic.py restates
// the split in numpy: worst case 2^-24.3, mean 2^-28 of |a*b|).
// The kernels here issue the six leading limb products as bf16 MFMAs that accumulate in fp32 (each bf16 x bf16 product is
// exact in the fp32 accumulator datapath), i.e. 6/16 of the fp32-MFMA time.  Measured against fp64 the result is at least as
// accurate as an fp32 GEMM (this repository's fp32-MFMA kernel and hipBLASLt) on every shape and value distribution tried
// (tools/gemm_accuracy.py, tests/test_gpu_kernels.py::test_gemm_three_limb_*).
// Same contractions, epilogue and tile order as cb_gemm.hip (th.matmul(feat_src, weight) GNN_model/GCN.py:225, the
// nn.Linear layers GCN.py:105,138 and their autograd GEMMs); cb_gemm.hip dispatches here when the shape qualifies.
//
// Bound: at K = N = 256 and M = 10^7 the six-pass MFMA time (~3.6 ms at 2.1 GHz) is next to the HBM time of streaming
// A in and C out once (20.5 GB, ~3.2 ms), so the kernel sits on the ridge; the fp32-MFMA kernel is 10 ms on the same shape.
//
// Tiling: 256 threads = 4 wavefronts WM x WN, wave tile 64 x 64 = 2x2 MFMA tiles, K step 16 (= one MFMA k-step, 24 MFMAs
// per wavefront between barriers).  Every operand is fetched with coalesced float4 loads in its natural row-major
// orientation, split into limbs in registers and written as three bf16 planes into LDS, again in its natural orientation:
//   * "row" operand (A of NN, k contiguous in memory): plane [tile row][16 k], 32 B per row; a fragment (8 consecutive k
//     of one row) is one ds_read_b128; the two 16-byte slots of a row are swapped when (row >> 3) & 1;
//   * "col" operand (B of NN, both operands of TN: k runs down the rows): plane [16 k][tile cols]; a fragment (8
//     consecutive k of one column) is gathered by two ds_read_b64_tr_b16 — the gfx950 LDS transpose read hands lane L of
//     each 16-lane group column L of a 4 x 16 block whose rows the group's lanes address — so no transposition ever
//     happens in registers or in the global access pattern; the 64-byte (32-column) chunks of a row are XOR-swizzled
//     with k & 3 so that the four rows of one transpose read sit on disjoint banks.
// Software pipeline inside every wavefront (two LDS stages of 24 KB for 128 x 128 tiles, three blocks per CU, one barrier
// per K step): while the MFMAs of K step k read stage k & 1, the same instruction stream splits the registers that hold
// step k+1 and writes them to the other stage, and re-fills those registers with the global loads of step k+2 — one
// slice of that work after every group of four MFMAs.  The matrix core, the vector ALU (the split), the LDS and the
// memory pipeline are therefore all busy at any instant instead of taking turns (thread blocks that share a CU fall
// into lock step, so phases that use only one of these units are not hidden by the neighbours).
// Addressing inside the K loop: a wave-uniform base pointer (scalar registers) per K step plus lane offsets that never change.
#include <utility>

#include "cb_common.h"
#include "cb_gemm_core.h"
#include "cb_gemm_limb.h"
#include "cb_limb_core.h"

namespace cb {

// wave tile 64 x (32 * WTN); block tile (64 * WM) x (32 * WTN * WN)
template <int WM, int WN, int WTN = 2, int PD = 1>
struct LTile {
  static constexpr int BM = 64 * WM, BN = 32 * WTN * WN;
  static constexpr int MINW = (WM == 2 && WTN == 2 && PD < 3 ? 3 : 2);   // wavefronts per SIMD the registers must allow
  static constexpr int NTHR = 64 * WM * WN;                                // four wavefronts per block; eight for the 256 x 256 TN tile (round 5)
  static_assert(WM * WN == 4 || WM * WN == 8, "four or eight wavefronts per block");
};

// ---- NN ------------------------------------------------------------------------------------
// ADROP: F.dropout of the A operand while it is staged (ep.adrop) — the dropout of the input features in front of the input Linear
template <int WM, int WN, bool OUT_BF16, int WTN = 2, int PD = 1, int EPI = 0, bool ADROP = false>
__global__ void __launch_bounds__(256, (LTile<WM, WN, WTN, PD>::MINW)) k_gemm_nn_l3(const float* __restrict__ A, int64_t lda,
                                                                                 const float* __restrict__ B, int64_t ldb,
                                                                                 void* __restrict__ Cv, int64_t ldc, int64_t M, int N,
                                                                                 int K, GemmEpilogue ep, int n_row_blocks,
                                                                                 int n_col_blocks, int c_vec_ok) {
  using T = LTile<WM, WN, WTN>;
  using OA = RowOperand<T::BM, ADROP>;
  using OB = ColOperand<T::BN, false>;
  constexpr int BM = T::BM, BN = T::BN;
  constexpr int SMEM = 2 * (OA::BYTES + OB::BYTES);
  static_assert(32 * (BN + 4) * 4 <= SMEM, "epilogue staging must fit");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int per_group = 8 * n_col_blocks;   // XCD-aware order, see k_gemm_nn
  const int grp = blockIdx.x / per_group, r = blockIdx.x % per_group;
  const int row_blk = grp * 8 + (r & 7), col_blk = r >> 3;
  if (row_blk >= n_row_blocks) return;
  const int64_t m0 = (int64_t)row_blk * BM;
  const int n0 = col_blk * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / WN, wc = w % WN;

  f32x16 acc[2][WTN];
  zero_acc_n<WTN>(acc);
  OA oa;
  OB ob;
  oa.init(lda, M - m0, t);
  if constexpr (ADROP) oa.set_drop(ep.adrop, m0, K, t);
  ob.init(ldb, N - n0, t);
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  uint32_t baddr[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) baddr[j] = OB::frag_addr(wc * (32 * WTN) + 32 * j, lane);
  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + m0 * lda, KS, lda, B + n0, (int64_t)KS * ldb, ldb, nullptr, K, aaddr, baddr, acc, t);
  nn_epilogue<WM, WN, WTN, OUT_BF16, EPI>(acc, reinterpret_cast<float*>(smem), Cv, ldc, m0, n0, M, N, ep, c_vec_ok, t);
}

// ---- TN ------------------------------------------------------------------------------------
// 1-D grid, XCD-aware: the tiles of one row split get block ids congruent mod 8 (same XCD / L2), so each operand
// panel is fetched from HBM once although tiles_i (tiles_j) tiles consume it.
// GDROP: F.dropout of the G operand while it is staged (gd): the weight gradient of the input Linear reads the undropped features
// ADROP: the same for the A operand (ad): the weight gradient of the first GCNConv reads X0 and regenerates the mask of F.dropout(X0)
template <int WM, int WN, bool SCALED, int WTN = 2, int PD = 1, bool GDROP = false, bool ADROP = false>
__global__ void __launch_bounds__((LTile<WM, WN, WTN, PD>::NTHR), (LTile<WM, WN, WTN, PD>::MINW)) k_gemm_tn_l3(const float* __restrict__ A, int64_t lda,
                                                                                 const float* __restrict__ G, int64_t ldg,
                                                                                 const float* __restrict__ rowscale,
                                                                                 float* __restrict__ partial, int64_t M, int K1, int K2,
                                                                                 int64_t rows_per_split, int tiles_j, int n_tiles,
                                                                                 int nsplit, DropSpec gd) {
  using T = LTile<WM, WN, WTN>;
  using OA = ColOperand<T::BM, false, ADROP, T::NTHR>;
  using OB = ColOperand<T::BN, SCALED, GDROP, T::NTHR>;
  constexpr int BM = T::BM, BN = T::BN;
  __shared__ __attribute__((aligned(16))) char smem[2 * (OA::BYTES + OB::BYTES)];
  const int b = blockIdx.x;
  const int tile = (b >> 3) % n_tiles, split = (b & 7) + 8 * (b / (8 * n_tiles));
  if (split >= nsplit) return;
  const int i0 = (tile / tiles_j) * BM, j0 = (tile % tiles_j) * BN;
  const int64_t r_begin = (int64_t)split * rows_per_split;
  const int64_t r_end = min(M, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / WN, wc = w % WN;

  f32x16 acc[2][WTN];
  zero_acc_n<WTN>(acc);
  OA oa;
  OB ob;
  oa.init(lda, K1 - i0, t);
  ob.init(ldg, K2 - j0, t);
  if constexpr (GDROP) ob.set_drop(gd, r_begin, r_end - r_begin, j0, t);
  if constexpr (ADROP) oa.set_drop(gd, r_begin, r_end - r_begin, i0, t);      // (one DropSpec: the two operand dropouts never occur together)
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  uint32_t baddr[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) baddr[j] = OB::frag_addr(wc * (32 * WTN) + 32 * j, lane);
  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + r_begin * lda + i0, (int64_t)KS * lda, lda, G + r_begin * ldg + j0, (int64_t)KS * ldg, ldg,
                           SCALED ? rowscale + r_begin : nullptr, r_end - r_begin, aaddr, baddr, acc, t);
  const int l31 = lane & 31, lh = lane >> 5;
  float* P = partial + (int64_t)split * K1 * K2;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < WTN; ++tj) {
      const int n = j0 + wc * (32 * WTN) + tj * 32 + l31;
      if (n >= K2) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = i0 + wr * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (m < K1) P[(int64_t)m * K2 + n] = acc[ti][tj][reg];
      }
    }
}

// ---- TN with the trunk's input stage as its A operand (cb_gemm_tn_instage_f32) ----------------------------------------------------------------
// dW_in^T = gy^T @ dropout(x) with gy = (X0 > 0) * (dropout_bwd(g) + mfold) COMPUTED while it is staged (ColOperandInStage) — the [M, 256] matrix gy is
// never written nor re-read; cs_partial[split][256] receives each slab's column sums of gy (the input Linear's bias gradient).  One 256 x 128 tile of
// eight wavefronts per slab (the round-5 wide tile: each element staged once).
__global__ void __launch_bounds__(512, 2) k_gemm_tn_instage(const float* __restrict__ g, const float* __restrict__ mfold,
                                                            const unsigned long long* __restrict__ x0_bits, const float* __restrict__ X, int64_t ldx,
                                                            float* __restrict__ partial, float* __restrict__ cs_partial, int64_t M, int K2,
                                                            int64_t rows_per_split, int nsplit, DropSpec xd, DropSpec gd) {
  using T = LTile<4, 2, 2>;
  using OA = ColOperandInStage<T::NTHR>;
  using OB = ColOperand<T::BN, false, true, T::NTHR>;
  constexpr int K1 = 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * (OA::BYTES + OB::BYTES)];
  const int split = blockIdx.x;
  if (split >= nsplit) return;
  const int64_t r_begin = (int64_t)split * rows_per_split;
  const int64_t r_end = min(M, r_begin + rows_per_split);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / 2, wc = w % 2;

  f32x16 acc[2][2];
  zero_acc_n<2>(acc);
  OA oa;
  OB ob;
  oa.init(K1, K1, t);
  ob.init(ldx, K2, t);
  oa.set_stage(g, mfold, x0_bits + r_begin * 4, gd, r_begin, r_end - r_begin, t);
  ob.set_drop(xd, r_begin, r_end - r_begin, 0, t);
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  uint32_t baddr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) baddr[j] = OB::frag_addr(wc * 64 + 32 * j, lane);
  limb_k_loop<2, 1, OA, OB>(oa, ob, smem, g + r_begin * K1, (int64_t)KS * K1, K1, X + r_begin * ldx, (int64_t)KS * ldx, ldx, nullptr, r_end - r_begin, aaddr, baddr,
                            acc, t);
  const int l31 = lane & 31, lh = lane >> 5;
  float* P = partial + (int64_t)split * K1 * K2;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = wc * 64 + tj * 32 + l31;
      if (n >= K2) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = wr * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        P[(int64_t)m * K2 + n] = acc[ti][tj][reg];
      }
    }
  // the slab's column sums: wavefront w staged the k rows w, w + 8 of every K step; fixed order (the K loop ended with a block barrier: smem is free)
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i) red[w * 256 + lane * 4 + i] = oa.cs[i];
  __syncthreads();
  if (t < 256) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j * 256 + t];
    cs_partial[(int64_t)split * 256 + t] = s;
  }
}

int launch_tn_instage(const float* g, const float* mfold, const uint64_t* x0_bits, const float* X, int64_t ldx, float* partial, float* cs_partial, int64_t M, int64_t K2,
                      int nsplit, int64_t rows_per_split, hipStream_t st, const DropSpec& xd, const DropSpec& gd) {
  hipLaunchKernelGGL(k_gemm_tn_instage, dim3((unsigned)nsplit), dim3(512), 0, st, g, mfold, (const unsigned long long*)x0_bits, X, ldx, partial, cs_partial, M, (int)K2,
                     rows_per_split, nsplit, xd, gd);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

// Register prefetch depth 1 everywhere (2 / 3 measured no gain: profiles/r01_*); the weight operand split once per launch and staged by
// LDS-DMA measured neutral (8.23 vs 8.08 ms on 10M x 256 x 256, profiles/r02_gemm_presplit.md: the K loop is bound by the six MFMA passes
// at the power-limited clock, not by instruction issue) and is not kept.
template <int WM, int WN, bool OUT_BF16, int WTN = 2>
static int launch_nn_l3_t(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                          int64_t K, GemmEpilogue ep, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  using T = LTile<WM, WN, WTN>;
  const int nrb = (int)((M + T::BM - 1) / T::BM), ncb = (int)((N + T::BN - 1) / T::BN);
  const int64_t groups = (nrb + 7) / 8;
  const int c_vec_ok = ((uintptr_t)C % (OUT_BF16 ? 8 : 16) == 0) && ldc % 4 == 0 && (!ep.addend || (al16(ep.addend) && ep.ld_add % 4 == 0));
  const dim3 grid((unsigned)(groups * 8 * ncb));
  if constexpr (!OUT_BF16 && WM == 2 && WTN == 4) {
    if (ep.out2) {    // dual-output epilogue (dropped copy): the wide-tile fp32 kernel only, see limb3_nn_dual_eligible
      if (ep.adrop.thresh)      // + dropout of the A operand in its staging (input Linear of the residual trunk)
        hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, false, WTN, 1, 1, true>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep,
                           nrb, ncb, c_vec_ok);
      else
        hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, false, WTN, 1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb,
                           ncb, c_vec_ok);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
    if (ep.row_ids) {   // the trunk's store on a subset of the node rows in the epilogue (cb_gemm_nn_store_rows_f32)
      hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, false, WTN, 1, 2>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb, c_vec_ok);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
    if (ep.adrop.thresh) {      // single output, dropout of the A operand in its staging (no dropped copy of A anywhere)
      hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, false, WTN, 1, 0, true>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb,
                         ncb, c_vec_ok);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
  }
  CB_CHECK_ARG(!ep.adrop.thresh, CB_E_INVALID, "NN contraction with operand dropout: wide fp32 tile only (check cb_gemm_nn_indrop_supported first)");
  hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, OUT_BF16, WTN, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, (int)N, (int)K, ep, nrb, ncb,
                     c_vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

// float4 access to both operands, lane offsets in 32 bits
bool limb3_nn_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t N, int64_t K) {
  return al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 && N % 4 == 0 && K > 0 && lda < (1 << 22) && ldb < (1 << 22);
}

// the dual-output epilogue exists for the wide (128 x 256) fp32 tile with vector stores on both outputs
bool limb3_nn_dual_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, const float* C2, int64_t ldc2,
                            int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep) {
  // below one wide tile per CU the 128 x 128 tile + a separate elementwise pass is faster (see launch_nn_limb3)
  return N > 128 && ((M + 127) / 128) * ((N + 255) / 256) >= 256 && limb3_nn_eligible(A, lda, B, ldb, N, K) && al16(C) && al16(C2) && ldc % 4 == 0 && ldc2 % 4 == 0 &&
         (!ep.addend || (al16(ep.addend) && ep.ld_add % 4 == 0));
}

size_t limb3_nn_workspace_bytes(int64_t, int64_t) { return 0; }      // (the three-limb NN kernels need no workspace)

int launch_nn_limb3(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    const GemmEpilogue& ep, bool out_bf16, hipStream_t st, void* ws, size_t ws_bytes) {
  if (N <= 64) {
    return out_bf16 ? launch_nn_l3_t<4, 1, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st)
                    : launch_nn_l3_t<4, 1, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
  }
  // 128 x 256 block tile (wave tile 64 x 128) where it fills the chip, else 128 x 128:
  // fewer than one wide tile per CU (a Pubmed-sized M = 19 717: 155 tiles): the 128 x 128 tile doubles the blocks in flight (-6 % on
  // the S-pubmed step); the dual-output epilogues exist for the wide tile only
  const bool fills = ((M + 127) / 128) * ((N + 255) / 256) >= 256 || ep.out2 || ep.adrop.thresh || ep.row_ids;
  if (N > 128 && fills)
    return out_bf16 ? launch_nn_l3_t<2, 2, true, 4>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes)
                    : launch_nn_l3_t<2, 2, false, 4>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes);
  return out_bf16 ? launch_nn_l3_t<2, 2, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes)
                  : launch_nn_l3_t<2, 2, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, st, ws, ws_bytes);
}

template <int WM, int WN, int WTN = 2>
static void launch_tn_l3_t(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* partial, int64_t M,
                           int64_t K1, int64_t K2, int nsplit, int64_t rows_per_split, hipStream_t st, const DropSpec& gd, const DropSpec& ad) {
  using T = LTile<WM, WN, WTN>;
  const int ti = (int)((K1 + T::BM - 1) / T::BM), tj = (int)((K2 + T::BN - 1) / T::BN);
  const dim3 grid((unsigned)(((nsplit + 7) / 8) * 8 * ti * tj));
#define CB_TN_LAUNCH(SC_, GD_, AD_)                                                                                                  \
  hipLaunchKernelGGL((k_gemm_tn_l3<WM, WN, SC_, WTN, 1, GD_, AD_>), grid, dim3(T::NTHR), 0, st, A, lda, G, ldg, rowscale, partial, M, (int)K1, \
                     (int)K2, rows_per_split, tj, ti * tj, nsplit, gd.thresh ? gd : ad)
  if (gd.thresh) CB_TN_LAUNCH(false, true, false);      // (launch_tn_limb3 admits it without a row scale only)
  else if (ad.thresh) {
    if constexpr (WM == 2 && WN == 2 && WTN == 2) { if (rowscale) CB_TN_LAUNCH(true, false, true); else CB_TN_LAUNCH(false, false, true); }
  } else if (rowscale) CB_TN_LAUNCH(true, false, false);
  else CB_TN_LAUNCH(false, false, false);
#undef CB_TN_LAUNCH
}

bool limb3_tn_eligible(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2) {
  return al16(A) && al16(G) && lda % 4 == 0 && ldg % 4 == 0 && K1 % 4 == 0 && K2 % 4 == 0 && lda < (1 << 22) && ldg < (1 << 22);
}

int launch_tn_limb3(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* partial, int64_t M,
                    int64_t K1, int64_t K2, int bm, int nsplit, int64_t rows_per_split, hipStream_t st, const DropSpec* gdrop, const DropSpec* adrop) {
  const DropSpec gd = gdrop ? *gdrop : DropSpec{};
  const DropSpec ad = adrop ? *adrop : DropSpec{};
  CB_CHECK_ARG(!gd.thresh || (!rowscale && gd.width % 4 == 0 && !ad.thresh), CB_E_INVALID,
               "TN contraction: dropout of the G operand needs width %% 4 == 0, no row scale and no dropout of A");
  CB_CHECK_ARG(!ad.thresh || (ad.width % 4 == 0 && bm == 128), CB_E_INVALID, "TN contraction: dropout of the A operand needs width %% 4 == 0 and the 128-row tile");
  if (bm == 64) launch_tn_l3_t<1, 4>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
  else if (bm == 256) launch_tn_l3_t<4, 1>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
  else {
    // Round 5: the whole 256 x 256 weight gradient of a hidden layer as ONE tile of eight wavefronts (wave tile 64 x 128 as before): each operand
    // element is staged — split into limbs — once, where the 128 x 256 tile staged the gradient operand twice: 2.1 instead of 3.1 staging VALU per
    // MFMA on a kernel whose SIMDs are issue-bound (PMC: matrix pipe 62 % + VALU 39 % of the cycles, profiles/r05_gemm_tn_pmc_raw.txt).
    // 7.29 -> 6.55 ms at M = 10^7, bit-identical (same K order per output element).  CB_GEMM_TN_WIDE=0 keeps the 128 x 256 tile.
    static const bool wide = !(getenv("CB_GEMM_TN_WIDE") && getenv("CB_GEMM_TN_WIDE")[0] == '0');
    if (wide && !ad.thresh && !gd.thresh && K1 == 256 && K2 == 256 && nsplit >= 256) {      // (fewer slabs than CUs — a Pubmed-sized M —: the 128 x 256 tile has twice the blocks)
      launch_tn_l3_t<4, 2, 4>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
    // the same for the input Linear's weight gradient (256 x F, F <= 128, the dropout mask of the features regenerated while they are staged): one
    // 256 x 128 tile of eight wavefronts draws every mask quad once instead of once per 128-row tile
    if (wide && gd.thresh && !ad.thresh && !rowscale && K1 == 256 && K2 <= 128 && K2 > 64 && nsplit >= 256) {
      launch_tn_l3_t<4, 2, 2>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
    // (operand dropout: the 128 x 128 tile — the wide one is at the register cap and the mask's Philox rounds would spill)
    if (!ad.thresh && K2 > 128 && ((K1 + 127) / 128) * ((K2 + 255) / 256) * nsplit >= 256) launch_tn_l3_t<2, 2, 4>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
    else launch_tn_l3_t<2, 2>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st, gd, ad);
  }
  CB_LAUNCH_CHECK();
  return CB_OK;
}

}  // namespace cb
