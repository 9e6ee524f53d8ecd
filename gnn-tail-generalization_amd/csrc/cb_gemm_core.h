// Shared fp32-MFMA GEMM building blocks (tile geometry, LDS operand staging, the MFMA inner step) used by
// cb_gemm.hip (NN / TN contractions) and cb_topk.hip (scores + running top-K).  See cb_gemm.hip for the design notes.
#pragma once
#include "cb_common.h"
#include "cb_philox.h"

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;   // K step (32 measured slower: 100 vs 115 TF/s, profiles/r01_*)

// F.dropout applied to an OPERAND on its way into LDS (value * keep(seed, flat index) / (1 - p), the mask cb_dropout_f32 draws): the
// dropped copy of the matrix is never written or read (the input features' dropout of the residual trunk, GCN.py:104).
struct DropSpec {
  uint32_t thresh;           // 0 = off
  float scale;               // 1 / (1 - p)
  uint64_t seed;
  const uint64_t* seed_dev;  // hipGraph mode: per-step part in device memory (added to `seed`), or null
  int64_t row0;              // global index of the matrix's row 0 (node-sharded runs draw the unsharded mask)
  int64_t width;             // row length of the dropped matrix in elements (multiple of 4)
};

struct GemmEpilogue {
  const float* rowscale;  // [M] or null
  const float* addend;    // [M, ld_add] or null
  int64_t ld_add;
  const float* bias;      // [N] or null
  int relu;
  // EPI == 1 kernels only: second output C2 = dropout(C) (F.dropout of the value just stored, GCN.py:110 after :105-107); the
  // keep-mask is the one cb_dropout_f32 draws for (seed, flat index (row0 + m) * N + n) — N % 4 == 0
  float* out2;
  int64_t ld_out2;
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
  const uint64_t* seed_dev;
  int64_t row0;
  int nt_store;                     // fp32 C leaves with the streaming (nt) policy
  DropSpec adrop;                   // ADROP kernels only: dropout of the A operand (thresh = 0: off)
  unsigned long long* relu_bits_out;   // 256-column tiles only: [M][4] mask words of (C > 0) after the ReLU (word q, bit L <-> column 4 L + q:
                                    // the layout of the aggregation's fused store) — the trunk's input stage reads them instead of C itself
  // EPI == 2 kernels only (N == 256): the trunk's fused store on the rows of a SUBSET of the node rows (cb_trunk_store_rows_f32's pass in this
  // epilogue): act = relu(rowscale * acc + addend + bias) -> out_act (optional); C = dropout(c_act * act + c_mix * mix_src[mix_index[m] | row_ids[m]]);
  // relu_bits_out is indexed by the NODE row row_ids[m]; the dropout mask (thresh, keep_scale, seed, seed_dev, row0 above) is drawn there too
  const int64_t* row_ids;
  const float* mix_src;
  int64_t ld_mix;
  const int64_t* mix_index;
  float c_act, c_mix;
  int bits_relu_only;
  float* out_act;
  int64_t ld_act;
};

template <int WM, int WN, int BKT = BK, int WTN = 2>
struct Tile {
  static constexpr int BM = 64 * WM, BN = 32 * WTN * WN, LDA = BM + 4, LDB = BN + 4;   // wave tile = 64 x (32*WTN)
  static constexpr int SMEM_FLOATS = 2 * BKT * (LDA + LDB);
  static_assert(WM * WN == 4, "four wavefronts per block");
  static_assert(32 * LDB <= SMEM_FLOATS, "epilogue staging (32 rows) must fit the operand buffers");
};

template <int LDA, int LDB, int BKT, int WTN>
__device__ __forceinline__ void mfma_tile_step(const float* __restrict__ As, const float* __restrict__ Bs, int wr, int wc,
                                               int lane, f32x16 (&acc)[2][WTN]) {
  const int l31 = lane & 31, kh = lane >> 5;
  const float* ar = As + kh * LDA + wr * 64 + l31;
  const float* br = Bs + kh * LDB + wc * (32 * WTN) + l31;
  float a[2], b[WTN];
  a[0] = ar[0]; a[1] = ar[32];
#pragma unroll
  for (int j = 0; j < WTN; ++j) b[j] = br[32 * j];
#pragma unroll
  for (int kk = 0; kk < BKT; kk += 2) {
    float na[2] = {0.f, 0.f}, nb[WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) nb[j] = 0.f;
    if (kk + 2 < BKT) {  // next k-step's fragments are in flight while this k-step's MFMAs issue
      na[0] = ar[(kk + 2) * LDA];
      na[1] = ar[(kk + 2) * LDA + 32];
#pragma unroll
      for (int j = 0; j < WTN; ++j) nb[j] = br[(kk + 2) * LDB + 32 * j];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    a[0] = na[0]; a[1] = na[1];
#pragma unroll
    for (int j = 0; j < WTN; ++j) b[j] = nb[j];
  }
}

// ---- operand staging -----------------------------------------------------------------------
// "row-major" operand (A of NN): global [tile rows][k], 16 k per K step -> transposed into LDS [k][row].
// thread t: k quad = t % 4, rows (t / 4) + 64 * j
template <int BMT, int BKT>
struct RowFrag {
  float v[BMT * BKT / 1024][4];
};
template <bool ALIGNED, int BMT, int BKT>
__device__ __forceinline__ void load_rowmajor(RowFrag<BMT, BKT>& f, const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M,
                                              int k0, int K, int t) {
  constexpr int TPR = BKT / 4, RPP = 256 / TPR;   // threads per row (one float4 each), rows per pass
  const int k = k0 + (t % TPR) * 4;
#pragma unroll
  for (int j = 0; j < BMT / RPP; ++j) {
    const int64_t m = m0 + t / TPR + RPP * j;
    if (ALIGNED && m < M && k + 4 <= K) {
      const float4 x = *reinterpret_cast<const float4*>(A + m * lda + k);
      f.v[j][0] = x.x; f.v[j][1] = x.y; f.v[j][2] = x.z; f.v[j][3] = x.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) f.v[j][i] = (m < M && k + i < K) ? A[m * lda + k + i] : 0.f;
    }
  }
}
template <int BMT, int BKT>
__device__ __forceinline__ void store_rowmajor_T(const RowFrag<BMT, BKT>& f, float* __restrict__ S, int t) {
  constexpr int TPR = BKT / 4, RPP = 256 / TPR;
  const int kq = (t % TPR) * 4;
#pragma unroll
  for (int j = 0; j < BMT / RPP; ++j) {
    const int m = t / TPR + RPP * j;
#pragma unroll
    for (int i = 0; i < 4; ++i) S[(kq + i) * (BMT + 4) + m] = f.v[j][i];
  }
}

// "k-major" operand: global [k][n] with n contiguous (B of NN; both operands of TN): straight copy.
// thread t: n quad = t % (BNT/4), k = t / (BNT/4) + (1024/BNT) * j
template <int BNT, int BKT>
struct KFrag {
  float4 v[BNT * BKT / 1024];
};
template <bool ALIGNED, int BNT, int BKT>
__device__ __forceinline__ void load_kmajor(KFrag<BNT, BKT>& f, const float* __restrict__ B, int64_t ldb, int64_t k0, int64_t Kdim,
                                            int n0, int N, int t, const float* __restrict__ kscale) {
  constexpr int TPR = BNT / 4, RPP = 256 / TPR;   // threads per row, rows per pass
  const int n = n0 + (t % TPR) * 4;
#pragma unroll
  for (int j = 0; j < BKT / RPP; ++j) {
    const int64_t k = k0 + t / TPR + RPP * j;
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
template <int BNT, int BKT>
__device__ __forceinline__ void store_kmajor(const KFrag<BNT, BKT>& f, float* __restrict__ S, int t) {
  constexpr int TPR = BNT / 4, RPP = 256 / TPR;
#pragma unroll
  for (int j = 0; j < BKT / RPP; ++j)
    *reinterpret_cast<float4*>(S + (t / TPR + RPP * j) * (BNT + 4) + (t % TPR) * 4) = f.v[j];
}

template <int WTN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][WTN]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// ---- NN epilogue through LDS: 32 rows x BN per pass -------------------------------------------
// C/D layout of the 32x32 MFMA (any input dtype): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// The accumulators are transposed through LDS so that every lane handles 4 consecutive columns of one row: row scale /
// addend / bias / relu are applied on float4 values and the tile leaves as coalesced 16-byte stores (8-byte for bf16 output).
// `Cs` must hold 32 x (BN + 4) floats and must no longer be read as operand storage by any wavefront of the block.
__device__ __forceinline__ void store4(float* __restrict__ p, float a, float b, float c, float d, int nt) {
  if (nt) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t v4 = {a, b, c, d};
    __builtin_nontemporal_store(v4, reinterpret_cast<f4_t*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
}

// EPI: 0 = plain; 1 = second output out2 = dropout(C) (+ the mask words of C > 0); 2 = the trunk's store on a subset of the node rows (see GemmEpilogue)
template <int WM, int WN, int WTN, bool OUT_BF16, int EPI = 0>
__device__ __forceinline__ void nn_epilogue(f32x16 (&acc)[2][WTN], float* __restrict__ Cs, void* __restrict__ Cv, int64_t ldc,
                                            int64_t m0, int n0, int64_t M, int N, const GemmEpilogue& ep, int c_vec_ok, int t) {
  constexpr int BN = 32 * WTN * WN, LDB = BN + 4;
  constexpr int G = 1;     // wave rows staged per pass (all WM at once measured neutral and pushed the dual-output epilogue into scratch)
  float* C = (float*)Cv;
  const int lane = t & 63, w = t >> 6, wr = w / WN, wc = w % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int TPR = BN / 4;             // threads per staged row
  constexpr int NV = 32 * TPR / 256;      // float4 per thread per pass
  // a thread owns the same 4 columns in every pass whenever 256 % TPR == 0 (all tiles here): its bias values are loaded once
  constexpr bool FIXED_COLS = 256 % TPR == 0;
  float bfix[4] = {0.f, 0.f, 0.f, 0.f};
  if (FIXED_COLS && ep.bias) {
    const int n = n0 + (t % TPR) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (n + q < N) bfix[q] = ep.bias[n + q];
  }
#pragma unroll
  for (int pass = 0; pass < 2 * (WM / G); ++pass) {
    const int grp = pass >> 1, ti = pass & 1;
    if (wr / G == grp) {
#pragma unroll
      for (int tj = 0; tj < WTN; ++tj)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Cs[((wr % G) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * LDB + wc * (32 * WTN) + tj * 32 + l31] = acc[ti][tj][reg];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G * NV; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / TPR, c4 = (idx % TPR) * 4;
      const int64_t m = m0 + (grp * G + row / 32) * 64 + ti * 32 + (row % 32);
      const int n = n0 + c4;
      if (m < M && n < N) {
        const float4 v = *reinterpret_cast<const float4*>(Cs + row * LDB + c4);
        float o[4] = {v.x, v.y, v.z, v.w};
        const float rs = ep.rowscale ? ep.rowscale[m] : 1.f;
        const bool full4 = n + 4 <= N;
        float ad[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (ep.addend) {
          const float* ap = ep.addend + m * ep.ld_add + n;
          if (full4 && c_vec_ok) {
            const float4 a4 = *reinterpret_cast<const float4*>(ap);
            ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n + q < N) ad[q] = ap[q];
          }
        }
        if (FIXED_COLS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = bfix[q];
        } else if (ep.bias) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (n + q < N) bv[q] = ep.bias[n + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = o[q] * rs + ad[q] + bv[q];
          if (ep.relu) o[q] = fmaxf(o[q], 0.f);
        }
        if constexpr (OUT_BF16) {   // Z stored as bf16 for the aggregation (build extension, fp32 accumulate downstream)
          bf16_t* cp = (bf16_t*)Cv + m * ldc + n;
          if (full4 && c_vec_ok) {
            *reinterpret_cast<uint2*>(cp) = pack4_bf16(o[0], o[1], o[2], o[3]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n + q < N) cp[q] = f32_to_bf16(o[q]);
          }
        } else if constexpr (EPI == 2) {   // launch contract: N == 256 (TPR == 64: a wavefront holds one whole row), vector stores, ep.relu set
          const int64_t gm = ep.row_ids[m];
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (ep.thresh) keep4(ep.seed_dev ? ep.seed + *ep.seed_dev : ep.seed, ((ep.row0 + gm) * N + n) >> 2, ep.thresh, ep.keep_scale, mk);
          if (ep.relu_bits_out) {
            unsigned long long mine = 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned long long wq = __ballot(o[q] > 0.f && (ep.bits_relu_only || mk[q] != 0.f));
              if ((t & 63) == q) mine = wq;
            }
            if ((t & 63) < 4) ep.relu_bits_out[gm * 4 + (t & 63)] = mine;
          }
          if (ep.out_act) store4(ep.out_act + m * ep.ld_act + n, o[0], o[1], o[2], o[3], 0);
          float x[4] = {o[0], o[1], o[2], o[3]};
          if (ep.mix_src) {
            const int64_t mr = ep.mix_index ? ep.mix_index[m] : gm;
            const float4 qv = *reinterpret_cast<const float4*>(ep.mix_src + mr * ep.ld_mix + n);
            x[0] = mix2(ep.c_act, o[0], ep.c_mix, qv.x); x[1] = mix2(ep.c_act, o[1], ep.c_mix, qv.y);
            x[2] = mix2(ep.c_act, o[2], ep.c_mix, qv.z); x[3] = mix2(ep.c_act, o[3], ep.c_mix, qv.w);
          }
          if (ep.thresh) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] *= mk[q];
          }
          store4(C + m * ldc + n, x[0], x[1], x[2], x[3], 0);
        } else {
          float* cp = C + m * ldc + n;
          if (full4 && c_vec_ok) {
            store4(cp, o[0], o[1], o[2], o[3], ep.nt_store);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n + q < N) cp[q] = o[q];
          }
          if constexpr (TPR == 64) {   // a wavefront holds one whole 256-column row: four ballots are its mask words
            if (ep.relu_bits_out) {
              unsigned long long mine = 0ull;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const unsigned long long wq = __ballot(o[q] > 0.f);
                if ((t & 63) == q) mine = wq;
              }
              if ((t & 63) < 4) ep.relu_bits_out[m * 4 + (t & 63)] = mine;
            }
          }
          if constexpr (EPI == 1) {   // launch contract: N % 4 == 0, 16-byte aligned out2 rows
            float mk[4];
            keep4(ep.seed_dev ? ep.seed + *ep.seed_dev : ep.seed, ((ep.row0 + m) * N + n) >> 2, ep.thresh, ep.keep_scale, mk);
            store4(ep.out2 + m * ep.ld_out2 + n, o[0] * mk[0], o[1] * mk[1], o[2] * mk[2], o[3] * mk[3], ep.nt_store);
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace cb
