// fp32 contractions on the bf16 matrix cores by exact operand decomposition ("three-limb" GEMM).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA (2.5 PF/s vs 157 TF/s dense).  An fp32
// value has a 24-bit significand = three 8-bit fields, and a bf16 holds exactly one such field with the full fp32 exponent
// range, so   a = a_hi + a_mid + a_lo   with three bf16 limbs is an EXACT identity (truncating split, every residual
// subtraction exact).  Then
//     a*b = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) + [mid*lo + lo*mid + lo*lo]
// and the bracket is below 2^-23 |a*b| — the rounding an fp32 multiply commits anyway.  The kernels here issue the six
// leading limb products as bf16 MFMAs that accumulate in fp32 (each bf16 x bf16 product is exact in the fp32 accumulator
// datapath), i.e. 6/16 of the fp32-MFMA time for results that differ from an fp32 GEMM only by the order of the fp32
// additions and the dropped 2^-23 terms (tests compare both kernels against an fp64 reference).
// Same contractions, epilogue and tile order as cb_gemm.hip (th.matmul(feat_src, weight) GNN_model/GCN.py:225, the
// nn.Linear layers GCN.py:105,138 and their autograd GEMMs); cb_gemm.hip dispatches here when the shape qualifies.
//
// Bound: at K = N = 256 and M = 10^7 the six-pass MFMA time (~3.6 ms at 2.1 GHz) is next to the HBM time of streaming
// A in and C out once (20.5 GB, ~3.2 ms), so the kernel sits on the ridge; the fp32-MFMA kernel is 10 ms on the same shape.
//
// Tiling: 256 threads = 4 wavefronts WM x WN, wave tile 64 x 64 = 2x2 MFMA tiles, K step 32 (= two MFMA k-steps, 48 MFMAs
// per wavefront between barriers).  Every operand is fetched with coalesced float4 loads in its natural row-major
// orientation, split into limbs in registers and written as three bf16 planes into LDS, again in its natural orientation:
//   * "row" operand (A of NN, k contiguous in memory): plane [tile row][32 k], 64 B per row; a fragment (8 consecutive k
//     of one row) is one ds_read_b128; the four 16-byte slots of a row are XOR-swizzled with (row >> 2) & 3;
//   * "col" operand (B of NN, both operands of TN: k runs down the rows): plane [32 k][tile cols]; a fragment (8
//     consecutive k of one column) is gathered by two ds_read_b64_tr_b16 — the gfx950 LDS transpose read hands lane L of
//     each 16-lane group column L of a 4 x 16 block whose rows the group's lanes address — so no transposition ever
//     happens in registers or in the global access pattern; the 64-byte (32-column) chunks of a row are XOR-swizzled
//     with k & 3 so that the four rows of one transpose read sit on disjoint banks.
// One LDS stage (48 KB for 128 x 128 tiles, three blocks per CU); the next K step's global loads are in flight in registers
// during the MFMAs.  Addressing inside the K loop: a wave-uniform base pointer (scalar registers) per K step plus lane
// offsets that never change.
#include <stdlib.h>

#include <utility>

#include "cb_common.h"
#include "cb_gemm_core.h"
#include "cb_gemm_limb.h"

namespace cb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// a == hi + mid + lo exactly; each limb is an fp32 bit pattern whose low 16 bits are zero (= a bf16 in the high half)
__device__ __forceinline__ void split3(float a, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));   // <= 8 significant bits left: the pack below keeps all of them
}
// two limbs (high halves of x0, x1) -> one dword {bf16(x0), bf16(x1)}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&pl)[3]) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3(v[i], h[i], m[i], l[i]);
  pl[0] = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
  pl[1] = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
  pl[2] = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int KS = 32;   // K step per barrier pair

// ---- "row" operand: global [rows][k] (k contiguous) -> LDS planes [R rows][32 k] -------------------
// Loads are split into NV pieces (one float4 per lane each) so that the K loop can issue them between MFMA groups; a piece
// only moves raw data into registers, masking of out-of-range elements happens when the registers are staged.
template <int R>
struct RowOperand {
  static constexpr int PLANE = R * 64, BYTES = 3 * PLANE, NV = R / 32;   // float4 per thread per K step
  uint32_t voff;      // element offset of this lane's first float4 from (tile row 0, k0): row (t/8), k quad (t%8)
  uint32_t woff;      // LDS byte offset of its 8-byte store in plane 0 (swizzled), j-th store: + j * 32 * 64
  uint32_t rmask;     // bit j: tile row t/8 + 32 j lies inside the matrix
  __device__ __forceinline__ void init(int64_t ld, int64_t rows_left, int t) {
    const int row = t >> 3, kq = t & 7;
    voff = (uint32_t)(row * ld + kq * 4);
    woff = row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + ((kq & 1) << 3);
    rmask = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) rmask |= (row + 32 * j < rows_left) ? (1u << j) : 0u;
  }
  // base = &A[tile row 0][k0] (uniform); k_left = K - k0 (FULL => k_left >= 32); K % 4 == 0
  template <bool FULL, int J>
  __device__ __forceinline__ void piece(float4 (&f)[NV], const float* __restrict__ base, int64_t ld, int64_t k_left, const float*,
                                        int t) const {
    const bool in = ((rmask >> J) & 1) && (FULL || (t & 7) * 4 < k_left);
    f[J] = *reinterpret_cast<const float4*>((base + (int64_t)(32 * J) * ld) + (in ? voff : 0u));
  }
  __device__ __forceinline__ void stage(const float4 (&f)[NV], char* __restrict__ S, int64_t k_left, int t) const {
    const bool kin = (t & 7) * 4 < k_left;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool live = ((rmask >> j) & 1) && kin;
      const float v[4] = {live ? f[j].x : 0.f, live ? f[j].y : 0.f, live ? f[j].z : 0.f, live ? f[j].w : 0.f};
      uint2 pl[3];
      split4(v, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + woff + j * (32 * 64)) = pl[p];
    }
  }
  // fragment address (plane 0, k-substep 0) of tile rows r0 + (lane & 31); substep 1 = address ^ 32
  static __device__ __forceinline__ uint32_t frag_addr(int r0, int lane) {
    const int row = r0 + (lane & 31);
    return row * 64 + (((lane >> 5) ^ ((row >> 2) & 3)) << 4);
  }
  static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, uint32_t addr, int plane, int s) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(S + plane * PLANE + (addr ^ (s << 5))));
  }
};

// ---- "col" operand: global [k][cols] (cols contiguous) -> LDS planes [32 k][C cols] ------------------
template <int C, bool SCALED>
struct ColOperand {
  static constexpr int ROWB = C * 2, PLANE = 32 * ROWB, BYTES = 3 * PLANE;
  static constexpr int TPR = C / 4, KPP = 256 / TPR, NV = 32 / KPP;   // threads per k row, k rows per pass, float4 per thread
  static constexpr int NC = C / 32;                                    // 64-byte chunks per row
  static_assert(NC == 2 || NC == 4 || NC == 8, "tile widths 64 / 128 / 256");
  static __device__ __forceinline__ int swz(int k) { return NC == 2 ? ((k >> 1) & 1) : (k & 3); }
  uint32_t voff;      // element offset of this lane's first float4 from (step row 0, tile col 0): k = t / TPR, quad t % TPR
  uint32_t woff;      // LDS byte offset of its 8-byte store in plane 0; j-th store: + j * KPP * ROWB  (KPP % 4 == 0)
  bool cok;           // its 4 columns lie inside the matrix (N % 4 == 0)
  float sc[SCALED ? NV : 1];   // raw per-row scales of the fetched step
  __device__ __forceinline__ void init(int64_t ld, int cols_left, int t) {
    const int k = t / TPR, nq = t % TPR;
    cok = nq * 4 < cols_left;
    voff = (uint32_t)(k * ld + (cok ? nq * 4 : 0));
    woff = k * ROWB + ((((nq >> 3) ^ swz(k)) & (NC - 1)) << 6) + ((nq & 7) << 3);
  }
  // base = &B[step row 0][tile col 0] (uniform); k_left = operand rows from there (FULL => >= 32); kscale = their scales
  template <bool FULL, int J>
  __device__ __forceinline__ void piece(float4 (&f)[NV], const float* __restrict__ base, int64_t ld, int64_t k_left,
                                        const float* __restrict__ kscale, int t) {
    const int k = t / TPR;
    const bool kin = FULL || k + KPP * J < k_left;
    f[J] = *reinterpret_cast<const float4*>((base + (int64_t)(KPP * J) * ld) + (kin ? voff : 0u));
    if constexpr (SCALED) sc[J] = (kscale + KPP * J)[kin ? k : 0];
  }
  __device__ __forceinline__ void stage(const float4 (&f)[NV], char* __restrict__ S, int64_t k_left, int t) const {
    const int k = t / TPR;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool live = cok && k + KPP * j < k_left;
      const float m = SCALED ? sc[j] : 1.f;
      const float v[4] = {live ? f[j].x * m : 0.f, live ? f[j].y * m : 0.f, live ? f[j].z * m : 0.f, live ? f[j].w * m : 0.f};
      uint2 pl[3];
      split4(v, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + woff + j * (KPP * ROWB)) = pl[p];
    }
  }
  // address (plane 0, k-substep 0, first transpose read) of the fragment of tile columns c0 .. c0+31 (c0 % 32 == 0):
  // lane L of a 16-lane group addresses row (L >> 2), column quad (L & 3) of the group's 4 x 16 block
  static __device__ __forceinline__ uint32_t frag_addr(int c0, int lane) {
    const int L = lane & 15, k = 8 * (lane >> 5) + (L >> 2);
    return k * ROWB + ((((c0 >> 5) ^ swz(k)) & (NC - 1)) << 6) + (((lane >> 4) & 1) << 5) + ((L & 3) << 3);
  }
  // k rows 16 s + 8 (lane >> 5) + {0..3} and {4..7}
  static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, uint32_t addr, int plane, int s) {
    const char* q = S + plane * PLANE + addr + s * (16 * ROWB);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * ROWB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
};

// The next K step's global loads, cut into NV_A + NV_B pieces that limb_tile_step issues between its MFMA groups: a
// wavefront that issued them in one burst would sit in the memory pipeline's queue instead of feeding the matrix core.
template <class OPA, class OPB>
struct Prefetch {
  OPA& oa;
  OPB& ob;
  float4 (&fa)[OPA::NV];
  float4 (&fb)[OPB::NV];
  const float* abase;
  const float* bbase;
  const float* bscale;
  int64_t lda, ldb, k_left;   // k_left <= 0: nothing to fetch
  int t;
  static constexpr int P = OPA::NV + OPB::NV;
  template <int PIECE>
  __device__ __forceinline__ void one() const {
    if (k_left <= 0) return;
    if constexpr (PIECE < OPA::NV) {
      if (k_left >= KS) oa.template piece<true, PIECE>(fa, abase, lda, k_left, nullptr, t);
      else oa.template piece<false, PIECE>(fa, abase, lda, k_left, nullptr, t);
    } else {
      constexpr int J = PIECE - OPA::NV;
      if (k_left >= KS) ob.template piece<true, J>(fb, bbase, ldb, k_left, bscale, t);
      else ob.template piece<false, J>(fb, bbase, ldb, k_left, bscale, t);
    }
  }
  template <int G, int NG, int... Is>
  __device__ __forceinline__ void group_impl(std::integer_sequence<int, Is...>) const {
    ((Is * NG / P == G ? one<Is>() : void()), ...);
  }
  // pieces scheduled with MFMA group G of NG
  template <int G, int NG>
  __device__ __forceinline__ void group() const { group_impl<G, NG>(std::make_integer_sequence<int, P>{}); }
  __device__ __forceinline__ void all() const { group<0, 1>(); }
};

// ---- one K step (32 = two MFMA k-steps) of a 64x64 wave tile -------------------------------------
// OPA / OPB: RowOperand or ColOperand; aaddr[i] / baddr[j]: fragment addresses of the wave's two 32-row / 32-column blocks.
// Limb products in increasing magnitude; planes are read just before their first use so that at most eight fragments
// are live; consecutive MFMAs go to different accumulators; one slice of the next K step's global loads per MFMA group.
template <int WTN, class OPA, class OPB, class PF>
__device__ __forceinline__ void limb_tile_step(const char* __restrict__ As, const char* __restrict__ Bs, const uint32_t (&aaddr)[2],
                                               const uint32_t (&baddr)[WTN], f32x16 (&acc)[2][WTN], const PF& pf) {
  constexpr int NH = WTN / 2, NG = 12 * NH;   // column halves of the wave tile; MFMA groups (of four) per K step
#define CB_MFMA4(A_, B_, H_, G_)                                                                             \
  pf.template group<G_, NG>();                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
      acc[i][2 * H_ + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][2 * H_ + j], 0, 0, 0);
#define CB_HALF(S_, H_)                                                                                                      \
  {                                                                                                                          \
    bf16x8 b_hi[2], b_mid[2], b_lo[2];                                                                                       \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_lo[i] = OPA::frag(As, aaddr[i], 2, S_); }                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_hi[j] = OPB::frag(Bs, baddr[2 * H_ + j], 0, S_);                         \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_hi[i] = OPA::frag(As, aaddr[i], 0, S_); }                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_lo[j] = OPB::frag(Bs, baddr[2 * H_ + j], 2, S_);                         \
    CB_MFMA4(a_lo, b_hi, H_, (S_ * NH + H_) * 6 + 0)                                                                             \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_mid[i] = OPA::frag(As, aaddr[i], 1, S_); }                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_mid[j] = OPB::frag(Bs, baddr[2 * H_ + j], 1, S_);                        \
    CB_MFMA4(a_hi, b_lo, H_, (S_ * NH + H_) * 6 + 1)                                                                             \
    CB_MFMA4(a_mid, b_mid, H_, (S_ * NH + H_) * 6 + 2)                                                                           \
    CB_MFMA4(a_mid, b_hi, H_, (S_ * NH + H_) * 6 + 3)                                                                            \
    CB_MFMA4(a_hi, b_mid, H_, (S_ * NH + H_) * 6 + 4)                                                                            \
    CB_MFMA4(a_hi, b_hi, H_, (S_ * NH + H_) * 6 + 5)                                                                             \
  }
#define CB_STEP(S_)                         \
  {                                         \
    bf16x8 a_hi[2], a_mid[2], a_lo[2];      \
    CB_HALF(S_, 0)                          \
    if constexpr (NH > 1) CB_HALF(S_, 1)    \
  }
  CB_STEP(0)
  CB_STEP(1)
#undef CB_STEP
#undef CB_HALF
#undef CB_MFMA4
}

template <int WTN>
__device__ __forceinline__ void zero_acc_n(f32x16 (&acc)[2][WTN]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// wave tile 64 x (32 * WTN); block tile (64 * WM) x (32 * WTN * WN)
template <int WM, int WN, int WTN = 2>
struct LTile {
  static constexpr int BM = 64 * WM, BN = 32 * WTN * WN;
  static constexpr int MINW = (WM == 2 && WTN == 2 ? 3 : 2);   // blocks per CU that the LDS stage allows
  static_assert(WM * WN == 4, "four wavefronts per block");
};

// ---- NN ------------------------------------------------------------------------------------
template <int WM, int WN, bool OUT_BF16, int WTN = 2>
__global__ void __launch_bounds__(256, (LTile<WM, WN, WTN>::MINW)) k_gemm_nn_l3(const float* __restrict__ A, int64_t lda,
                                                                            const float* __restrict__ B, int64_t ldb,
                                                                            void* __restrict__ Cv, int64_t ldc, int64_t M, int N, int K,
                                                                            GemmEpilogue ep, int n_row_blocks, int n_col_blocks,
                                                                            int c_vec_ok) {
  using T = LTile<WM, WN, WTN>;
  using OA = RowOperand<T::BM>;
  using OB = ColOperand<T::BN, false>;
  constexpr int BM = T::BM, BN = T::BN;
  constexpr int SMEM = OA::BYTES + OB::BYTES;
  static_assert(32 * (BN + 4) * 4 <= SMEM, "epilogue staging must fit");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  char* As = smem;
  char* Bs = smem + OA::BYTES;
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
  ob.init(ldb, N - n0, t);
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  uint32_t baddr[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) baddr[j] = OB::frag_addr(wc * (32 * WTN) + 32 * j, lane);
  const float* a_tile = A + m0 * lda;
  const float* b_tile = B + n0;
  float4 fa[OA::NV], fb[OB::NV];
  const int nk = (K + KS - 1) / KS;
  Prefetch<OA, OB>{oa, ob, fa, fb, a_tile, b_tile, nullptr, lda, ldb, (int64_t)K, t}.all();
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * KS;
    oa.stage(fa, As, K - k0, t);
    ob.stage(fb, Bs, K - k0, t);
    __syncthreads();
    // next K step's operands travel while this one's MFMAs run
    const Prefetch<OA, OB> pf{oa, ob, fa, fb, a_tile + (k0 + KS), b_tile + (int64_t)(k0 + KS) * ldb, nullptr, lda, ldb,
                              (int64_t)K - (k0 + KS), t};
    limb_tile_step<WTN, OA, OB>(As, Bs, aaddr, baddr, acc, pf);
    __syncthreads();
  }
  nn_epilogue<WM, WN, WTN, OUT_BF16>(acc, reinterpret_cast<float*>(smem), Cv, ldc, m0, n0, M, N, ep, c_vec_ok, t);
}

// ---- TN ------------------------------------------------------------------------------------
// 1-D grid, XCD-aware: the tiles of one row split get block ids congruent mod 8 (same XCD / L2), so each operand
// panel is fetched from HBM once although tiles_i (tiles_j) tiles consume it.
template <int WM, int WN, bool SCALED, int WTN = 2>
__global__ void __launch_bounds__(256, (LTile<WM, WN, WTN>::MINW)) k_gemm_tn_l3(const float* __restrict__ A, int64_t lda,
                                                                            const float* __restrict__ G, int64_t ldg,
                                                                            const float* __restrict__ rowscale,
                                                                            float* __restrict__ partial, int64_t M, int K1, int K2,
                                                                            int64_t rows_per_split, int tiles_j, int n_tiles,
                                                                            int nsplit) {
  using T = LTile<WM, WN, WTN>;
  using OA = ColOperand<T::BM, false>;
  using OB = ColOperand<T::BN, SCALED>;
  constexpr int BM = T::BM, BN = T::BN;
  __shared__ __attribute__((aligned(16))) char smem[OA::BYTES + OB::BYTES];
  char* As = smem;
  char* Bs = smem + OA::BYTES;
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
  const uint32_t aaddr[2] = {OA::frag_addr(wr * 64, lane), OA::frag_addr(wr * 64 + 32, lane)};
  uint32_t baddr[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) baddr[j] = OB::frag_addr(wc * (32 * WTN) + 32 * j, lane);
  float4 fa[OA::NV], fb[OB::NV];
  const int64_t nk = r_end > r_begin ? (r_end - r_begin + KS - 1) / KS : 0;
  const float* a_col = A + i0;
  const float* g_col = G + j0;
  Prefetch<OA, OB>{oa, ob, fa, fb, a_col + r_begin * lda, g_col + r_begin * ldg, SCALED ? rowscale + r_begin : nullptr, lda, ldg,
                   r_end - r_begin, t}.all();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int64_t row = r_begin + kt * KS;
    oa.stage(fa, As, r_end - row, t);
    ob.stage(fb, Bs, r_end - row, t);
    __syncthreads();
    const int64_t nrow = row + KS;
    const Prefetch<OA, OB> pf{oa, ob, fa, fb, a_col + nrow * lda, g_col + nrow * ldg, SCALED ? rowscale + nrow : nullptr, lda, ldg,
                              r_end - nrow, t};
    limb_tile_step<WTN, OA, OB>(As, Bs, aaddr, baddr, acc, pf);
    __syncthreads();
  }
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

static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

template <int WM, int WN, bool OUT_BF16, int WTN = 2>
static int launch_nn_l3_t(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                          int64_t K, GemmEpilogue ep, hipStream_t st) {
  using T = LTile<WM, WN, WTN>;
  const int nrb = (int)((M + T::BM - 1) / T::BM), ncb = (int)((N + T::BN - 1) / T::BN);
  const int64_t groups = (nrb + 7) / 8;
  const int c_vec_ok = ((uintptr_t)C % (OUT_BF16 ? 8 : 16) == 0) && ldc % 4 == 0 && (!ep.addend || (al16(ep.addend) && ep.ld_add % 4 == 0));
  hipLaunchKernelGGL((k_gemm_nn_l3<WM, WN, OUT_BF16, WTN>), dim3((unsigned)(groups * 8 * ncb)), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M,
                     (int)N, (int)K, ep, nrb, ncb, c_vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

// float4 access to both operands, lane offsets in 32 bits
bool limb3_nn_eligible(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t N, int64_t K) {
  return al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 && N % 4 == 0 && K > 0 && lda < (1 << 22) && ldb < (1 << 22);
}

int launch_nn_limb3(const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    const GemmEpilogue& ep, bool out_bf16, hipStream_t st) {
  if (N <= 64) {
    return out_bf16 ? launch_nn_l3_t<4, 1, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st)
                    : launch_nn_l3_t<4, 1, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
  }
  static const int wide = getenv("CB_LIMB_WIDE") ? atoi(getenv("CB_LIMB_WIDE")) : 0;   // measurement hook: 128 x 256 block tile
  if (wide && N > 128)
    return out_bf16 ? launch_nn_l3_t<2, 2, true, 4>(A, lda, B, ldb, C, ldc, M, N, K, ep, st)
                    : launch_nn_l3_t<2, 2, false, 4>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
  return out_bf16 ? launch_nn_l3_t<2, 2, true>(A, lda, B, ldb, C, ldc, M, N, K, ep, st)
                  : launch_nn_l3_t<2, 2, false>(A, lda, B, ldb, C, ldc, M, N, K, ep, st);
}

template <int WM, int WN, int WTN = 2>
static void launch_tn_l3_t(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* partial, int64_t M,
                           int64_t K1, int64_t K2, int nsplit, int64_t rows_per_split, hipStream_t st) {
  using T = LTile<WM, WN, WTN>;
  const int ti = (int)((K1 + T::BM - 1) / T::BM), tj = (int)((K2 + T::BN - 1) / T::BN);
  const dim3 grid((unsigned)(((nsplit + 7) / 8) * 8 * ti * tj));
  if (rowscale)
    hipLaunchKernelGGL((k_gemm_tn_l3<WM, WN, true, WTN>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, partial, M, (int)K1, (int)K2,
                       rows_per_split, tj, ti * tj, nsplit);
  else
    hipLaunchKernelGGL((k_gemm_tn_l3<WM, WN, false, WTN>), grid, dim3(256), 0, st, A, lda, G, ldg, rowscale, partial, M, (int)K1, (int)K2,
                       rows_per_split, tj, ti * tj, nsplit);
}

bool limb3_tn_eligible(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2) {
  return al16(A) && al16(G) && lda % 4 == 0 && ldg % 4 == 0 && K1 % 4 == 0 && K2 % 4 == 0 && lda < (1 << 22) && ldg < (1 << 22);
}

int launch_tn_limb3(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* partial, int64_t M,
                    int64_t K1, int64_t K2, int bm, int nsplit, int64_t rows_per_split, hipStream_t st) {
  if (bm == 64) launch_tn_l3_t<1, 4>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st);
  else if (bm == 256) launch_tn_l3_t<4, 1>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st);
  else {
    static const int wide = getenv("CB_LIMB_WIDE") ? atoi(getenv("CB_LIMB_WIDE")) : 0;
    if (wide && K2 > 128) launch_tn_l3_t<2, 2, 4>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st);
    else launch_tn_l3_t<2, 2>(A, lda, G, ldg, rowscale, partial, M, K1, K2, nsplit, rows_per_split, st);
  }
  CB_LAUNCH_CHECK();
  return CB_OK;
}

}  // namespace cb
