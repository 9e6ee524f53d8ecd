// Three-limb bf16-MFMA implementation of the fp32 contractions (cb_gemm_limb.hip); cb_gemm.hip dispatches to it.
#pragma once
#include "cb_gemm_core.h"

namespace cb {

// operands 16-byte aligned with leading dimensions, K and N multiples of 4 (float4 access); any M
bool limb3_nn_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t N, int64_t K);
size_t limb3_nn_workspace_bytes(int64_t N, int64_t K);
int launch_nn_limb3(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    const GemmEpilogue& ep, bool out_bf16, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
bool limb3_nn_dual_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, const float* C2, int64_t ldc2,
                            int64_t M, int64_t N, int64_t K, const GemmEpilogue& ep);
bool limb3_tn_eligible(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2);
// partial slabs [nsplit][K1][K2] exactly as k_gemm_tn writes them; bm = tile rows chosen by tn_tile()
int launch_tn_limb3(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* partial, int64_t M,
                    int64_t K1, int64_t K2, int bm, int nsplit, int64_t rows_per_split, hipStream_t st, const DropSpec* gdrop = nullptr,
                    const DropSpec* adrop = nullptr);

// the input Linear's weight gradient with the trunk's input stage computed in its A operand's staging (k_gemm_tn_instage): partial [nsplit][256][K2],
// cs_partial [nsplit][256]
int launch_tn_instage(const float* g, const float* mfold, const uint64_t* x0_bits, const float* X, int64_t ldx, float* partial, float* cs_partial, int64_t M, int64_t K2,
                      int nsplit, int64_t rows_per_split, hipStream_t st, const DropSpec& xd, const DropSpec& gd);

}  // namespace cb
