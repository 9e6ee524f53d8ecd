// "An fp32 tile in LDS times a 256-column weight on the matrix cores": the dense part shared by the aggregation + GEMM kernels
// (cb_agg_gemm.hip: the tile is the block's 64 aggregated rows) and the forward front (cb_front.hip: the tile is 64 rows of x, then of X0).
// Arithmetic = cb_gemm_limb.hip's, product by product (three exact bf16 limbs per fp32 operand, the six leading limb products per K step
// in limb_tile_step's order, fp32 MFMA accumulators), so results are bit-identical to cb_gemm_nn_f32 on the same operands.
//   A operand: the LDS tile; a fragment (8 consecutive k of one row) = two ds_read_b128, split into limbs in registers (tile rows of
//              TLD floats, TLD * 4 = 16 mod 256 bytes: the 16 lanes of a b128 group hit 16 distinct 16-byte bank columns);
//   B operand: the weight, split ONCE per launch by k_weight_image into MFMA fragment order (K x 256 x 6 bytes, L2 resident): a fragment
//              is one coalesced global_load_dwordx4 per limb, no LDS, no conversion in the K loop.
#pragma once
#include "cb_common.h"
#include "cb_limb_core.h"

namespace cb {

constexpr int kTM = 64;      // rows per tile
constexpr int kTLD = 260;    // floats per LDS tile row (256-wide tiles)
constexpr int kKD = 256;     // width of the aggregated rows = K of the dense part of cb_agg_gemm.hip
constexpr int kND = 256;     // output width of the dense part
constexpr int kNT = kND / 32, kNS = kKD / 16;

// image[((s * kNT + j) * 3 + p) * 64 + lane] = limb p of B[16 s + 8 (lane >> 5) + e][32 j + (lane & 31)], e = 0..7 (B[k][n] = W[k * sk + n * sn]),
// s < n_steps (K = 16 n_steps rows of B)
__global__ void __launch_bounds__(256) k_weight_image(const float* __restrict__ W, int64_t sk, int64_t sn, uint4* __restrict__ image, int n_steps);

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

// 16-byte store of a value that is written once and read by a later kernel: streaming (nt) policy
__device__ __forceinline__ void store_stream4(float* __restrict__ p, const float (&v)[4]) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const f4_t q = {v[0], v[1], v[2], v[3]};
  __builtin_nontemporal_store(q, reinterpret_cast<f4_t*>(p));
}

// acc[i][j] = rows 32 i .. 32 i + 31 of the tile  x  columns 64 w + 32 j .. + 31 of B, over NSTEPS K steps of 16.
// PF = K steps of B fragments in flight per wavefront (ring of PF + 1 register buffers of 24 registers each).  PF = 1 (K loop unrolled by
// two) where the wavefront's B latency is covered by others (cb_agg_gemm.hip: the gathering wavefronts bound the kernel); PF = 2 where the multiplying wavefronts ARE the kernel (cb_front.hip): an L2 hit comes back after ~1 us, a K step's 24 MFMAs
// take 0.4 us.
template <int NSTEPS, int TLD, int PF = 1>
__device__ __forceinline__ void tile_times_image(const float* __restrict__ tile, const uint4* __restrict__ image, int w, int lane, f32x16 (&acc)[2][2]) {
  static_assert(PF >= 1 && PF <= 3 && PF < NSTEPS, "prefetch distance");
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* a_row[2] = {tile + l31 * TLD + 8 * lh, tile + (32 + l31) * TLD + 8 * lh};
  // B fragments: a RUNNING pointer, advanced every K step (fixed per-step addresses would all be loop invariants of the persistent
  // tile loop: the compiler hoists them — 96 address pairs — and spills)
  const uint4* bp = image + ((int64_t)(2 * w) * 3) * 64 + lane;
  uint4 bq[PF + 1][2][3];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[d][j][p] = bp[j * 192 + p * 64];
    bp += kNT * 192;
  }
  auto step = [&](int s, int cur, int nx) {
    if (s + PF < NSTEPS) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[nx][j][p] = bp[j * 192 + p * 64];
    }
    bp += kNT * 192;
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // one 32-row block of A at a time: its limbs live only across its twelve MFMAs
      const float4 x0 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s);
      const float4 x1 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s + 4);
      uint32_t hh[4], mm[4], ll[4];
      split3x2(x0.x, x0.y, hh[0], mm[0], ll[0]);
      split3x2(x0.z, x0.w, hh[1], mm[1], ll[1]);
      split3x2(x1.x, x1.y, hh[2], mm[2], ll[2]);
      split3x2(x1.z, x1.w, hh[3], mm[3], ll[3]);
      const bf16x8 a_hi = as_bf16x8(make_uint4(hh[0], hh[1], hh[2], hh[3]));
      const bf16x8 a_mid = as_bf16x8(make_uint4(mm[0], mm[1], mm[2], mm[3]));
      const bf16x8 a_lo = as_bf16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
      // limb products in increasing magnitude, the order of limb_tile_step (cb_limb_core.h)
#define CB_TG_MFMA2(A_, P_) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, as_bf16x8(bq[cur][j][P_]), acc[i][j], 0, 0, 0);
      CB_TG_MFMA2(a_lo, 0)
      CB_TG_MFMA2(a_hi, 2)
      CB_TG_MFMA2(a_mid, 1)
      CB_TG_MFMA2(a_mid, 0)
      CB_TG_MFMA2(a_hi, 1)
      CB_TG_MFMA2(a_hi, 0)
#undef CB_TG_MFMA2
    }
  };
  // rolled loop, unrolled by the ring size (ring indices are compile-time constants); a FULL unroll lets the compiler hoist every B load to the
  // top and spill (724 bytes of scratch per lane)
  constexpr int R = PF + 1, MAIN = NSTEPS / R * R;
#pragma unroll 1
  for (int s0 = 0; s0 < MAIN; s0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(s0 + r, r, (r + PF) % R);
  }
#pragma unroll
  for (int r = 0; r < NSTEPS - MAIN; ++r) step(MAIN + r, r, (r + PF) % R);      // (MAIN % R == 0: step s sits in ring slot s % R)
}

// The same product with the A limbs split ONCE per element instead of once per multiplying wavefront (round 6): before the MFMAs of K step s the
// block's 256 threads split the tile's 64 x 16 slab of step s + 1 cooperatively (one float4 each: row t >> 2, k quad t & 3) into three bf16 planes in
// LDS (RowOperand<64>'s layout: 2 KB per plane, two stages = 12 KB), and every wavefront reads its A fragments as bf16 (three ds_read_b128 per
// 32-row block) — the K loop's limb-split VALU work drops from 4 x (the whole slab per wavefront) to 1 x, for one block barrier per K step.  Same limb
// values (split4), same limb products in the same order: bit-identical to tile_times_image.  All four wavefronts of the block must call it together.
// planes: 2 * RowOperand<kTM>::BYTES bytes of LDS.
template <int NSTEPS, int TLD, int PF = 1>
__device__ __forceinline__ void tile_times_image_coop(const float* __restrict__ tile, char* __restrict__ planes, const uint4* __restrict__ image, int w, int lane,
                                                      int t, f32x16 (&acc)[2][2]) {
  using OA = RowOperand<kTM>;
  static_assert(OA::NV == 1 && PF >= 1 && PF <= 3 && PF < NSTEPS, "64-row tiles, one float4 of the slab per thread");
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int row = t >> 2, kq = t & 3;
  const float* a_src = tile + row * TLD + 4 * kq;                                  // this thread's float4 of slab 0
  const uint32_t woff = row * 32 + (((kq >> 1) ^ ((row >> 3) & 1)) << 4) + ((kq & 1) << 3);      // RowOperand<64>::init's store offset
  const uint32_t aaddr[2] = {OA::frag_addr(0, lane), OA::frag_addr(32, lane)};
  auto stage = [&](int s, char* dst) {
    const float4 x = *reinterpret_cast<const float4*>(a_src + 16 * s);
    const float v[4] = {x.x, x.y, x.z, x.w};
    uint2 pl[3];
    split4(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * OA::PLANE + woff) = pl[p];
  };
  const uint4* bp = image + ((int64_t)(2 * w) * 3) * 64 + lane;
  uint4 bq[PF + 1][2][3];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[d][j][p] = bp[j * 192 + p * 64];
    bp += kNT * 192;
  }
  stage(0, planes);
  __syncthreads();
  auto step = [&](int s, int cur, int nx) {
    if (s + PF < NSTEPS) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[nx][j][p] = bp[j * 192 + p * 64];
    }
    bp += kNT * 192;
    const char* S = planes + (s & 1) * OA::BYTES;
    if (s + 1 < NSTEPS) stage(s + 1, planes + ((s + 1) & 1) * OA::BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16x8 a_lo = OA::frag(S, aaddr[i], 2), a_hi = OA::frag(S, aaddr[i], 0), a_mid = OA::frag(S, aaddr[i], 1);
#define CB_TG_MFMA2(A_, P_) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, as_bf16x8(bq[cur][j][P_]), acc[i][j], 0, 0, 0);
      CB_TG_MFMA2(a_lo, 0)
      CB_TG_MFMA2(a_hi, 2)
      CB_TG_MFMA2(a_mid, 1)
      CB_TG_MFMA2(a_mid, 0)
      CB_TG_MFMA2(a_hi, 1)
      CB_TG_MFMA2(a_hi, 0)
#undef CB_TG_MFMA2
    }
    __syncthreads();      // the slab of step s + 1 is staged; every wavefront has read the planes of step s
  };
  constexpr int R = PF + 1, MAIN = NSTEPS / R * R;
#pragma unroll 1
  for (int s0 = 0; s0 < MAIN; s0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(s0 + r, r, (r + PF) % R);
  }
#pragma unroll
  for (int r = 0; r < NSTEPS - MAIN; ++r) step(MAIN + r, r, (r + PF) % R);
}

// ---- narrow tail (round 5: the output Linear 256 -> C <= 64 as the tail of the LAST layer's aggregation, GCN.py:133-138) ----------------
// image layout as above with kNTn = 2 column blocks (columns >= C are zero): image[((s * kNTn + j) * 3 + p) * 64 + lane]
constexpr int kNTn = 2;
__global__ void __launch_bounds__(256) k_weight_image_narrow(const float* __restrict__ W, int64_t sk, int64_t sn, uint4* __restrict__ image, int n_steps,
                                                             int n_cols);

// acc = rows 32 i .. 32 i + 31 of the tile  x  columns 32 j .. 32 j + 31 of the narrow image, over NSTEPS K steps of 16: the four multiplying
// wavefronts of a block take the four (i, j) blocks of a 64 x 64 output.  Same limb products in the same order as tile_times_image.
template <int NSTEPS, int TLD>
__device__ __forceinline__ void tile_times_image_block(const float* __restrict__ tile, const uint4* __restrict__ image, int i, int j, int lane, f32x16& acc) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* a_row = tile + (32 * i + l31) * TLD + 8 * lh;
  const uint4* bp = image + ((int64_t)j * 3) * 64 + lane;
  uint4 bq[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) bq[0][p] = bp[p * 64];
  bp += kNTn * 192;
  auto step = [&](int s, int cur, int nx) {
    if (s + 1 < NSTEPS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[nx][p] = bp[p * 64];
    }
    bp += kNTn * 192;
    const float4 x0 = *reinterpret_cast<const float4*>(a_row + 16 * s);
    const float4 x1 = *reinterpret_cast<const float4*>(a_row + 16 * s + 4);
    uint32_t hh[4], mm[4], ll[4];
    split3x2(x0.x, x0.y, hh[0], mm[0], ll[0]);
    split3x2(x0.z, x0.w, hh[1], mm[1], ll[1]);
    split3x2(x1.x, x1.y, hh[2], mm[2], ll[2]);
    split3x2(x1.z, x1.w, hh[3], mm[3], ll[3]);
    const bf16x8 a_hi = as_bf16x8(make_uint4(hh[0], hh[1], hh[2], hh[3]));
    const bf16x8 a_mid = as_bf16x8(make_uint4(mm[0], mm[1], mm[2], mm[3]));
    const bf16x8 a_lo = as_bf16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, as_bf16x8(bq[cur][0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, as_bf16x8(bq[cur][2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, as_bf16x8(bq[cur][1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, as_bf16x8(bq[cur][0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, as_bf16x8(bq[cur][1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, as_bf16x8(bq[cur][0]), acc, 0, 0, 0);
  };
  static_assert(NSTEPS % 2 == 0, "K loop unrolled by two");
#pragma unroll 1
  for (int s0 = 0; s0 < NSTEPS; s0 += 2) {
    step(s0, 0, 1);
    step(s0 + 1, 1, 0);
  }
}

constexpr int kCLD = 68;     // floats per row of a wavefront's private C strip (8 rows x 64 columns)

// Epilogue of a wavefront's 64 x 64 block of accumulators through a WAVE-PRIVATE staging strip: 8 rows x 64 columns per pass, transposed so
// that a lane holds a float4 of one row; fn(m, n, v) receives tile row m (0 .. 63), column n (64 w + 4 (idx & 15)) and the four values.
template <class F>
__device__ __forceinline__ void acc_rows_through_strip(const f32x16 (&acc)[2][2], float* __restrict__ cs, int w, int lane, F&& fn) {
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int j = 0; j < 2; ++j) cs[(r4 + 4 * lh) * kCLD + 32 * j + l31] = acc[i][j][4 * q + r4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int idx = lane + 64 * half, row = idx >> 4, c4 = (idx & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(cs + row * kCLD + c4);
        fn(32 * i + 8 * q + row, 64 * w + c4, v);
      }
    }
}

}  // namespace cb
