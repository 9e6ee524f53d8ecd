// Aggregation + the NEXT dense transform in one kernel ("the matrix cores run under the gathers").
//
// Replaces, per layer of the residual trunk, the pair
//     forward :  X_{l+1} = store( A^T Z_l )                     GNN_model/GCN.py:238-253,127-133   (cb_spmm_csr_fused_f32)
//                Z_{l+1} = a . (X_{l+1} W_{l+1}) + E_{l+1}      GNN_model/GCN.py:213,225,230-235    (cb_gemm_nn_f32)
//     backward:  dZ_l    = A (b . dY'_l)                        autograd of :238                    (cb_spmm_csr_f32, reverse CSR)
//                dX_l    = a . (dZ_l W_l^T)                     autograd of :213,225                (cb_gemm_nn_f32)
// by ONE launch each: a block of four wavefronts aggregates 64 consecutive rows exactly as k_spmm_rows does (same edge-stream walk,
// same stores: X_{l+1} / dZ_l still go to memory, the weight-gradient GEMM needs them), keeps the finished rows in LDS as fp32
// (64 x 256, 65 KB) and multiplies the tile by the 256 x 256 weight before anything else is read.  What disappears: the second
// kernel's 10 GB read of the matrix just written, and the matrix cores' time as a term of its own — the aggregation is bound by
// HBM gathers at ~1.1 kW with the matrix cores idle, the three-limb GEMM by the power-limited clock (profiles/r02_power_probe.txt);
// two blocks share a CU, so one block's MFMA phase runs while the other one gathers (the aggregation does not need more
// residency than that: profiles/r03_fused_agg_gemm.md, occupancy sweep).
//
// Arithmetic of the dense part = cb_gemm_limb.hip's, product by product: fp32 operands as three exact bf16 limbs, the six leading
// limb products per K step in the same order into fp32 MFMA accumulators, `rowscale * acc + addend` on the way out — results are
// bit-identical to cb_gemm_nn_f32 on the same inputs (tests/test_gpu_agg_gemm.py).
//   A operand: the LDS tile; a fragment (8 consecutive k of one row) = two ds_read_b128, split into limbs in registers
//              (1040-byte tile rows: the 16 lanes of a b128 group hit 16 distinct 16-byte bank columns);
//   B operand: the weight, split ONCE per launch by k_agg_gemm_image into MFMA fragment order (384 KB, L2 resident): a fragment is
//              one coalesced global_load_dwordx4 per limb, no LDS, no conversion in the K loop;
//   C: accumulators -> the same LDS tile -> row-major float4 -> epilogue -> 1 KiB streaming row stores (the store pattern of
//      the aggregation itself).
// Hub rows (more edges than the hub threshold) are reduced by the hub kernels, which run BEFORE this kernel here; their finished
// rows are read back from memory into the tile.
#include <stdlib.h>
#include <string.h>

#include "cb_common.h"
#include "cb_limb_core.h"
#include "cb_spmm_core.h"

namespace cb {

constexpr int kTM = 64;      // rows per block (4 wavefronts x 16 rows: the row block of k_spmm_rows)
constexpr int kTLD = 260;    // floats per LDS tile row
constexpr int kKD = 256;     // width of the aggregated rows = K of the dense part
constexpr int kND = 256;     // output width of the dense part
constexpr int kNT = kND / 32, kNS = kKD / 16;

struct GemmTail {
  const uint4* image;      // weight limbs in fragment order (k_agg_gemm_image)
  const float* rowscale;   // [rows] or null
  const float* addend;     // [rows, ld_add] or null
  int64_t ld_add;
  float* out;              // [rows, ld_out]
  int64_t ld_out;
  // TB kernels only (trunk backward): the value just computed is dL/dx of the stage above layer l-1; the backward of that layer's fused
  // store — what cb_trunk_layer_bwd_f32 does in a pass of its own — leaves the same epilogue:
  //   out2 = c_act * keep(seed, m, n) * g * relu_bit_{l-1}(m, n) * rowscale2[m];   colsum partial[block][n] += (the same without rowscale2)
  const unsigned long long* bits;   // [rows][4] mask words of the forward store of layer l-1 (d = 256: one tile)
  float c_act;
  uint32_t thresh;
  float keep_scale;
  uint64_t seed;
  const uint64_t* seed_dev;
  int64_t row0;
  const float* rowscale2;
  float* out2;
  int64_t ld_out2;
  float* colsum_partial;   // [gridDim.x][256] or null
  int out_masked;          // TB: `out` itself leaves as keep(seed, m, n) * g (dropout backward applied: the form the trunk's input stage consumes)
  int dbg;                 // measurement hook CB_AGG_GEMM_DBG (bit 0: B fragments loaded once, bit 1: A fragments split once, bit 2: no K loop)
};

// image[((s * kNT + j) * 3 + p) * 64 + lane] = limb p of B[16 s + 8 (lane >> 5) + e][32 j + (lane & 31)], e = 0..7 (B[k][n] = W[k * sk + n * sn])
__global__ void __launch_bounds__(256) k_agg_gemm_image(const float* __restrict__ W, int64_t sk, int64_t sn, uint4* __restrict__ image) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = idx & 63, sj = idx >> 6, j = sj % kNT, s = sj / kNT;
  if (s >= kNS) return;
  const int k0 = 16 * s + 8 * (lane >> 5), n = 32 * j + (lane & 31);
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    split3x2(W[(int64_t)(k0 + 2 * e) * sk + (int64_t)n * sn], W[(int64_t)(k0 + 2 * e + 1) * sk + (int64_t)n * sn], h[e], m[e], l[e]);
  uint4* o = image + ((int64_t)(s * kNT + j) * 3) * 64 + lane;
  o[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o[64] = make_uint4(m[0], m[1], m[2], m[3]);
  o[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

template <bool FUSED, int GP>
__global__ void __launch_bounds__(256, 2) k_agg_gemm(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ h,
                                                     int64_t ld_h, float* __restrict__ out, int64_t ld_out, int n_rows, Epilogue ep, int hub_T,
                                                     FusedEpi fe, GemmTail gt) {
  __shared__ __attribute__((aligned(16))) float tile[kTM * kTLD];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * kTM + w * 16;
  const int nr = max(0, min(16, n_rows - r0));
  const int c0 = lane * 4;
  float* tile_lane = tile + (w * 16) * kTLD + c0;

  // ---- phase 1: the aggregation of k_spmm_rows<4, 16, 8, ...>; every finished row also lands in the tile ----------------------
  unsigned long long hubmask = 0ull;
  if (nr > 0) {
    int my_ptr = __builtin_nontemporal_load(rowptr + r0 + min(lane, nr));
    float my_scale = 1.f;
    if (ep.row_scale && lane < nr) my_scale = __builtin_nontemporal_load(ep.row_scale + r0 + lane);
    const int nxt = __shfl_down(my_ptr, 1);
    hubmask = __ballot(lane < nr && (nxt - my_ptr) > hub_T);
    float bvec[4] = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bvec[i] = ep.bias[c0 + i];
    }
    const float* h_lane = h + c0;
    float* out_lane = out + c0;
    if (hubmask == 0) {
      stream_rows<4, 8, true, FUSED, false, float, GP, false, kTLD>(0, nr, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, true, ep.relu,
                                                                    bvec, fe, c0, nullptr, 0, ep, tile_lane);
    } else {
      int r = 0;
      while (r < nr) {
        const unsigned long long m = hubmask >> r;
        const int nh = m ? r + (__ffsll((long long)m) - 1) : nr;
        if (nh > r)
          stream_rows<4, 8, true, FUSED, false, float, GP, false, kTLD>(r, nh, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, true,
                                                                        ep.relu, bvec, fe, c0, nullptr, 0, ep, tile_lane);
        r = nh + 1;
      }
    }
  }
  // rows the stream did not produce: hub rows (finished by the hub kernels, which ran before this launch) and rows past the end
  if (hubmask != 0ull || nr < 16) {
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < nr) {
        if (!((hubmask >> i) & 1ull)) continue;
        const float* src = FUSED ? fe.out_next + (int64_t)(r0 + i) * fe.ld_next + c0 : out + (int64_t)(r0 + i) * ld_out + c0;
        v = *reinterpret_cast<const float4*>(src);
      }
      *reinterpret_cast<float4*>(tile_lane + i * kTLD) = v;
    }
  }
  __syncthreads();

  // ---- phase 2: tile (64 x 256) @ W (256 x 256): wavefront w owns output columns [64 w, 64 w + 64) ---------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* a_row[2] = {tile + l31 * kTLD + 8 * lh, tile + (32 + l31) * kTLD + 8 * lh};
  const uint4* bimg = gt.image + ((int64_t)(2 * w) * 3) * 64 + lane;       // + s * (kNT * 192) + j * 192 + p * 64
  uint4 bq[2][2][3];                                                       // [buffer][column tile][limb]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) bq[0][j][p] = bimg[j * 192 + p * 64];
  const bool dbg_b = gt.dbg & 1, dbg_a = gt.dbg & 2;
  bf16x8 a_hi[2], a_mid[2], a_lo[2];
#pragma unroll
  for (int s = 0; s < kNS; ++s) {
    if (gt.dbg & 4) break;
    const int cur = dbg_b ? 0 : (s & 1), nx = cur ^ 1;
    if (s + 1 < kNS && !dbg_b) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[nx][j][p] = bimg[(s + 1) * (kNT * 192) + j * 192 + p * 64];
    }
    if (!dbg_a || s == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 x0 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s);
      const float4 x1 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s + 4);
      uint32_t hh[4], mm[4], ll[4];
      split3x2(x0.x, x0.y, hh[0], mm[0], ll[0]);
      split3x2(x0.z, x0.w, hh[1], mm[1], ll[1]);
      split3x2(x1.x, x1.y, hh[2], mm[2], ll[2]);
      split3x2(x1.z, x1.w, hh[3], mm[3], ll[3]);
      a_hi[i] = as_bf16x8(make_uint4(hh[0], hh[1], hh[2], hh[3]));
      a_mid[i] = as_bf16x8(make_uint4(mm[0], mm[1], mm[2], mm[3]));
      a_lo[i] = as_bf16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    }
    }
    // limb products in increasing magnitude, the order of limb_tile_step (cb_limb_core.h)
#define CB_AG_MFMA4(A_, P_)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], as_bf16x8(bq[cur][j][P_]), acc[i][j], 0, 0, 0);
    CB_AG_MFMA4(a_lo, 0)
    CB_AG_MFMA4(a_hi, 2)
    CB_AG_MFMA4(a_mid, 1)
    CB_AG_MFMA4(a_mid, 0)
    CB_AG_MFMA4(a_hi, 1)
    CB_AG_MFMA4(a_hi, 0)
#undef CB_AG_MFMA4
  }
  __syncthreads();      // every wavefront has read its last A fragment: the tile becomes the C staging buffer
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        tile[(32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * kTLD + 64 * w + 32 * j + l31] = acc[i][j][reg];
  __syncthreads();
  const float zero_bias = ep.bias ? 0.f : 0.f;   // (keeps the epilogue expression of cb_gemm_core.h's nn_epilogue: o * rs + addend + bias)
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    if (i >= nr) break;
    const int64_t m = r0 + i;
    const float4 v = *reinterpret_cast<const float4*>(tile_lane + i * kTLD);
    float o[4] = {v.x, v.y, v.z, v.w};
    const float rs = gt.rowscale ? gt.rowscale[m] : 1.f;
    float ad[4] = {0.f, 0.f, 0.f, 0.f};
    if (gt.addend) {
      const float4 a4 = *reinterpret_cast<const float4*>(gt.addend + m * gt.ld_add + c0);
      ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = o[q] * rs + ad[q] + zero_bias;
    store_stream<4>(gt.out + m * gt.ld_out + c0, o);
  }
}

// Hand-over of the LDS tile buffers WITHOUT a block barrier (ASYNC form of k_agg_gemm2): two counters per buffer in LDS.
//   ready[b]: +1 by every gathering wavefront that has written its rows of the tile in buffer b   (tile complete at NG * (use + 1))
//   freed[b]: +1 by every multiplying wavefront that has read the tile in buffer b for the last time (buffer reusable at 4 * use)
// A wavefront's LDS instructions are executed in order, so a counter increment issued after the tile writes (or reads) is seen after them;
// the asm statements only keep the COMPILER from moving LDS accesses across the hand-over.  No wait on outstanding global loads / stores
// (a block barrier drains them): a gathering wavefront that has finished its rows moves on to the next tile while its row stores are still
// in flight and while the other wavefronts finish theirs, so the eight gathering wavefronts of a CU drift apart and cover each other's
// start-of-tile latencies (rowptr -> column ids -> first gathers), which a barrier lines up.  Spins are bounded: a lost hand-over produces
// wrong numbers that the parity tests catch, never a hung GPU.
__device__ __forceinline__ void ag_wait(int* flag, int target) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 26)) break;      // ~7 s of s_sleep: only a lost hand-over gets here
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ag_signal(int* flag, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- version 2: one persistent 8-wavefront block per CU, wavefront-specialised ------------------------------------------------
// Two co-resident blocks that each alternate gather / MFMA phases fall into lock step (both gather, then both multiply: measured,
// profiles/r03_fused_agg_gemm.md), so nothing overlaps.  Here the roles are fixed instead: wavefronts 0-3 only gather (tile t+1
// into one LDS buffer), wavefronts 4-7 only multiply (tile t from the other buffer) and store; ONE block barrier per tile hands the
// buffers over.  The 64 rows of a tile are cut among the four gathering wavefronts at equal EDGE counts (row boundaries from
// the tile's rowptr values, three ballots), so that they reach the barrier together — a fixed 16 rows each leaves the block
// waiting for the wavefront with the heaviest rows.
template <int U, bool FUSED, int GP, int NG>
__device__ __forceinline__ void ag2_gather_tile(int t, float* __restrict__ tile, int w, int lane, const int* __restrict__ rowptr,
                                                const int* __restrict__ col, const float* __restrict__ h, int64_t ld_h, float* __restrict__ out,
                                                int64_t ld_out, int n_rows, const Epilogue& ep, int hub_T, const FusedEpi& fe) {
  const int r0 = t * kTM;
  const int nrt = min(kTM, n_rows - r0);
  const int c0 = lane * 4;
  const int my_ptr = __builtin_nontemporal_load(rowptr + r0 + min(lane, nrt));
  const int ptr_hi = rowptr[r0 + nrt];                      // (uniform address: scalar load)
  float my_scale = 1.f;
  if (ep.row_scale && lane < nrt) my_scale = __builtin_nontemporal_load(ep.row_scale + r0 + lane);
  int nxt = __shfl_down(my_ptr, 1);
  if (lane == kWave - 1) nxt = ptr_hi;
  const unsigned long long hubmask = __ballot(lane < nrt && (nxt - my_ptr) > hub_T);
  // this wavefront's rows [ra, rb): the rows whose first edge falls into its quarter of the tile's edge range
  const int e0 = bcast_lane(my_ptr, 0), e1 = ptr_hi;
  const int64_t span = (int64_t)e1 - e0;
  const int ta = e0 + (int)(span * w / NG), tb = e0 + (int)(span * (w + 1) / NG);
  const int ra = w == 0 ? 0 : (int)__popcll(__ballot(lane < nrt && my_ptr < ta));
  const int rb = w == NG - 1 ? nrt : (int)__popcll(__ballot(lane < nrt && my_ptr < tb));
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (ep.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bvec[i] = ep.bias[c0 + i];
  }
  const float* h_lane = h + c0;
  float* out_lane = out + c0;
  float* tile_lane = tile + c0;
  int r = ra;
  while (r < rb) {      // maximal hub-free runs of [ra, rb)
    const unsigned long long m = (hubmask >> r) & (rb - r >= 64 ? ~0ull : ((1ull << (rb - r)) - 1ull));
    const int nh = m ? r + (__ffsll((long long)m) - 1) : rb;
    if (nh > r)
      stream_rows<4, U, true, FUSED, false, float, GP, false, kTLD, true>(r, nh, nrt, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, true,
                                                                          ep.relu, bvec, fe, c0, nullptr, 0, ep, tile_lane, ptr_hi);
    if (nh < rb) {      // hub row nh: finished by the hub kernels, which ran before this launch
      const float* src = FUSED ? fe.out_next + (int64_t)(r0 + nh) * fe.ld_next + c0 : out + (int64_t)(r0 + nh) * ld_out + c0;
      *reinterpret_cast<float4*>(tile_lane + nh * kTLD) = *reinterpret_cast<const float4*>(src);
    }
    r = nh + 1;
  }
  if (w == NG - 1) {
    for (int i = nrt; i < kTM; ++i) *reinterpret_cast<float4*>(tile_lane + i * kTLD) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

constexpr int kCLD = 68;     // floats per row of a multiplying wavefront's private C strip (8 rows x 64 columns)

// PF = K steps of B fragments in flight per multiplying wavefront (ring of PF + 1 register buffers, K loop unrolled by PF + 1): next
// to wavefronts that keep dozens of gathers outstanding, a load of this CU — L2 hit or not — comes back after microseconds.
template <int PF, bool TB>
__device__ __forceinline__ void ag2_mfma_tile(int t, const float* __restrict__ tile, float* __restrict__ cs, int w, int lane, int n_rows,
                                              const GemmTail& gt, float (&colsum)[4], uint64_t seed_eff, int* freed = nullptr) {
  static_assert(PF == 1 || PF == 3, "ring of 2 or 4 fragment buffers (16 K steps)");
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* a_row[2] = {tile + l31 * kTLD + 8 * lh, tile + (32 + l31) * kTLD + 8 * lh};
  // B fragments: a RUNNING pointer, advanced every K step (fixed per-step addresses would all be loop invariants of the persistent
  // tile loop: the compiler hoists them — 96 address pairs — and spills)
  const uint4* bp = gt.image + ((int64_t)(2 * w) * 3) * 64 + lane;
  uint4 bq[PF + 1][2][3];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[d][j][p] = bp[d * (kNT * 192) + j * 192 + p * 64];
  }
  bp += PF * (kNT * 192);
#pragma unroll(PF + 1)
  for (int s = 0; s < kNS; ++s) {
    const int cur = s % (PF + 1), nx = (s + PF) % (PF + 1);
    if (s + PF < kNS && !(gt.dbg & 1)) {      // (dbg bit 0, measurement: the fragments of the first steps are reused)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[nx][j][p] = bp[j * 192 + p * 64];
    }
    bp += kNT * 192;
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // one 32-row block of A at a time: its limbs live only across its twelve MFMAs
      float4 x0, x1;
      if (!(gt.dbg & 64)) {
        x0 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s);
        x1 = *reinterpret_cast<const float4*>(a_row[i] + 16 * s + 4);
      } else {      // (dbg bit 6, measurement: no LDS reads of the tile, no limb split in the loop)
        x0 = make_float4(1.f + lane, 2.f, 3.f, 4.f);
        x1 = x0;
      }
      uint32_t hh[4], mm[4], ll[4];
      split3x2(x0.x, x0.y, hh[0], mm[0], ll[0]);
      split3x2(x0.z, x0.w, hh[1], mm[1], ll[1]);
      split3x2(x1.x, x1.y, hh[2], mm[2], ll[2]);
      split3x2(x1.z, x1.w, hh[3], mm[3], ll[3]);
      const bf16x8 a_hi = as_bf16x8(make_uint4(hh[0], hh[1], hh[2], hh[3]));
      const bf16x8 a_mid = as_bf16x8(make_uint4(mm[0], mm[1], mm[2], mm[3]));
      const bf16x8 a_lo = as_bf16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
      // limb products in increasing magnitude, the order of limb_tile_step (cb_limb_core.h)
#define CB_AG_MFMA2(A_, P_) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, as_bf16x8(bq[cur][j][P_]), acc[i][j], 0, 0, 0);
      CB_AG_MFMA2(a_lo, 0)
      CB_AG_MFMA2(a_hi, 2)
      CB_AG_MFMA2(a_mid, 1)
      CB_AG_MFMA2(a_mid, 0)
      CB_AG_MFMA2(a_hi, 1)
      CB_AG_MFMA2(a_hi, 0)
#undef CB_AG_MFMA2
    }
  }
  if (freed) ag_signal(freed, lane);      // ASYNC form: the tile has been read for the last time
  // epilogue through a WAVE-PRIVATE staging strip (the tile itself is still being read by the other three multiplying wavefronts
  // and there is no barrier among four of eight wavefronts): 8 rows x 64 columns per pass, transposed so that a lane applies
  // `rowscale * acc + addend` on a float4 and the strip leaves as 256-byte row segments
  const int r0 = t * kTM;
  const float zero_bias = 0.f;
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
        const int64_t m = r0 + 32 * i + 8 * q + row;
        const float4 v = *reinterpret_cast<const float4*>(cs + row * kCLD + c4);
        if (m < n_rows) {
          const int n = 64 * w + c4;
          const float rs = gt.rowscale ? gt.rowscale[m] : 1.f;
          float ad[4] = {0.f, 0.f, 0.f, 0.f};
          if (gt.addend) {
            const float4 a4 = *reinterpret_cast<const float4*>(gt.addend + m * gt.ld_add + n);
            ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
          }
          float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] * rs + ad[e] + zero_bias;
          float gm[4] = {o[0], o[1], o[2], o[3]};
          if constexpr (TB) {      // the arithmetic of k_trunk_bwd<0> (cb_elementwise.hip), element for element
            if (gt.thresh) {
              float mk[4];
              keep4(seed_eff, ((gt.row0 + m) * kND + n) >> 2, gt.thresh, gt.keep_scale, mk);
#pragma unroll
              for (int e = 0; e < 4; ++e) gm[e] *= mk[e];
            }
            if (gt.out_masked) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = gm[e];
            }
          }
          if (!(gt.dbg & 2)) store_stream<4>(gt.out + m * gt.ld_out + n, o);      // (dbg bit 1, measurement: no output store)
          if constexpr (TB) {
            const unsigned long long* bw = gt.bits + m * 4;      // word e, bit L <-> column 4 L + e
            const int L = n >> 2;
            const float sc2 = gt.rowscale2 ? gt.rowscale2[m] : 1.f;
            float gy[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              gy[e] = ((bw[e] >> L) & 1ull) ? gt.c_act * gm[e] : 0.f;
              colsum[e] += gy[e];
              gy[e] *= sc2;
            }
            store_stream<4>(gt.out2 + m * gt.ld_out2 + n, gy);
          }
        }
      }
    }
}

template <int U, bool FUSED, int GP, int NG, bool TB = false, bool ASYNC = false>
__global__ void __launch_bounds__(64 * (NG + 4), (NG + 4) / 4) k_agg_gemm2(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                          const float* __restrict__ h, int64_t ld_h, float* __restrict__ out,
                                                                          int64_t ld_out, int n_rows, Epilogue ep, int hub_T, FusedEpi fe,
                                                                          GemmTail gt, int n_tiles) {
  // NG gathering wavefronts (ids 0 .. NG-1) + 4 multiplying ones: NG = 4 -> 2 wavefronts per SIMD and up to 256 registers each
  // (deep B look-ahead); NG = 8 -> 3 per SIMD and 168 registers (one K step of look-ahead), but twice the gathers in flight
  __shared__ __attribute__((aligned(16))) float tiles[2][kTM * kTLD];
  __shared__ __attribute__((aligned(16))) float cstrip[4][8 * kCLD];
  __shared__ int ready[2], freed[2];
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform by construction: tell the compiler, so that what derives from it stays in SGPRs)
  const bool gathers = wv < NG;
  const int w = gathers ? wv : wv - NG;
  const int n_it = ((int)blockIdx.x < n_tiles) ? (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};      // TB: sums of this lane's 4 output columns (64 w + 4 (lane & 15) ..) over the rows it stored
  const uint64_t seed_eff = TB ? (gt.seed_dev ? gt.seed + *gt.seed_dev : gt.seed) : 0ull;
  if constexpr (ASYNC) {
    if (threadIdx.x < 2) ready[threadIdx.x] = freed[threadIdx.x] = 0;
    __syncthreads();
    if (gathers) {
      for (int it = 0; it < n_it; ++it) {
        const int b = it & 1, use = it >> 1;
        if (use > 0) ag_wait(&freed[b], 4 * use);
        if (!(gt.dbg & 16))
          ag2_gather_tile<U, FUSED, GP, NG>(blockIdx.x + it * gridDim.x, tiles[b], w, lane, rowptr, col, h, ld_h, out, ld_out, n_rows, ep, hub_T, fe);
        ag_signal(&ready[b], lane);
      }
    } else {
      for (int it = 0; it < n_it; ++it) {
        const int b = it & 1, use = it >> 1;
        ag_wait(&ready[b], NG * (use + 1));
        if (!(gt.dbg & 8))
          ag2_mfma_tile<(NG == 4 ? 3 : 1), TB>(blockIdx.x + it * gridDim.x, tiles[b], cstrip[w], w, lane, n_rows, gt, colsum, seed_eff, &freed[b]);
        else
          ag_signal(&freed[b], lane);
      }
    }
  } else {
    for (int it = 0; it <= n_it; ++it) {
      if (gathers) {
        if (it < n_it && !(gt.dbg & 16))
          ag2_gather_tile<U, FUSED, GP, NG>(blockIdx.x + it * gridDim.x, tiles[it & 1], w, lane, rowptr, col, h, ld_h, out, ld_out, n_rows, ep, hub_T, fe);
      } else if (it >= 1 && !(gt.dbg & 8)) {
        ag2_mfma_tile<(NG == 4 ? 3 : 1), TB>(blockIdx.x + (it - 1) * gridDim.x, tiles[(it - 1) & 1], cstrip[w], w, lane, n_rows, gt, colsum, seed_eff);
      }
      __syncthreads();
    }
  }
  if constexpr (TB) {
    // column sums of the block's rows: the four lanes that own the same column quad are added in a fixed order, then one partial row per
    // block (summed over the blocks by k_agg_colsum_finish in block order: no float atomics, bit-reproducible)
    if (!gathers && gt.colsum_partial) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = colsum[e];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        colsum[e] = v;
      }
      if (lane < 16) {
        float* pp = gt.colsum_partial + (int64_t)blockIdx.x * kND + 64 * w + 4 * lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] = colsum[e];
      }
    }
  }
}

// out[c] = sum over blocks of partial[block][c], blocks in ascending order (one thread per column)
__global__ void __launch_bounds__(256) k_agg_colsum_finish(const float* __restrict__ partial, int n_blocks, float* __restrict__ out) {
  const int c = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < n_blocks; ++b) s += partial[(int64_t)b * kND + c];
  out[c] = s;
}


static inline int64_t ag_partial_ld(int64_t d) { return (d + 3) / 4 * 4; }

// blocks of the persistent kernel: one per CU
static int g_cu_limit = 0;      // cb_agg_gemm_set_cu_limit: CUs the launching stream may use (0 = all)

static int ag_n_blocks(int n_tiles) {
  if (g_cu_limit > 0) return n_tiles < g_cu_limit ? n_tiles : g_cu_limit;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
    if (n_cu > 1024) n_cu = 1024;      // (cb_spmm_gemm_trunkbwd_workspace_bytes: one partial row per block)
  }
  return n_tiles < n_cu ? n_tiles : n_cu;
}

template <bool FUSED>
static int launch_agg_gemm(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, Epilogue ep, float* out,
                           int64_t ld_out, int hub_T, int n_hubs, int n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                           float* partial, hipStream_t st, FusedEpi fe, GemmTail gt) {
  const int d = kKD;
  const dim3 blk(256);
  if (n_hubs > 0) {      // hub rows first: the main kernel reads their finished rows back
    const int64_t ld_p = ag_partial_ld(d);
    const dim3 gridc((unsigned)((n_chunks + 3) / 4), 1);
    if (ep.col_flags)
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, 8, float, 2>), gridc, blk, 0, st, rowptr, col, h, ld_h, d, hub_T, n_hubs, n_chunks, hub_rows,
                         hub_chunk_ptr, partial, ld_p, ep);
    else
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, 8, float, 0>), gridc, blk, 0, st, rowptr, col, h, ld_h, d, hub_T, n_hubs, n_chunks, hub_rows,
                         hub_chunk_ptr, partial, ld_p, ep);
    CB_LAUNCH_CHECK();
    const dim3 grid2((unsigned)((n_hubs + 3) / 4), 1);
    hipLaunchKernelGGL((k_spmm_hub_finish<4, FUSED>), grid2, blk, 0, st, d, n_hubs, hub_rows, hub_chunk_ptr, partial, ld_p, out, ld_out, ep, fe);
    CB_LAUNCH_CHECK();
  }
  const int n_tiles = (int)((N + kTM - 1) / kTM);
  static const int version = getenv("CB_AGG_GEMM_V") ? atoi(getenv("CB_AGG_GEMM_V")) : 2;      // measurement hook: 1 = the two-phase block form
  if (version == 1) {
    const dim3 grid((unsigned)n_tiles);
    if (ep.col_flags)
      hipLaunchKernelGGL((k_agg_gemm<FUSED, 2>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt);
    else
      hipLaunchKernelGGL((k_agg_gemm<FUSED, 0>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt);
    CB_LAUNCH_CHECK();
    return CB_OK;
  }
  const dim3 grid2((unsigned)ag_n_blocks(n_tiles));      // one persistent block per CU (139 KB of LDS each)
  // measured on S-pl10M (profiles/r03_fused_agg_gemm.md): 8 gathering wavefronts; with the flag hand-over 12 gathers in flight per wavefront
  // (160 registers, no scratch) = 16 (168 + 20-92 B of scratch) > 8; with the block barrier 8 (16 spills there)
  static const bool async = !(getenv("CB_AGG_GEMM_SYNC") && !strcmp(getenv("CB_AGG_GEMM_SYNC"), "barrier"));      // measurement hook: the block-barrier hand-over
  static const int u16 = getenv("CB_AGG_GEMM_U") ? atoi(getenv("CB_AGG_GEMM_U")) : (async ? 12 : 8);
  static const int ng = getenv("CB_AGG_GEMM_NG") ? atoi(getenv("CB_AGG_GEMM_NG")) : 8;
#define CB_AG2(U_, GP_, NG_) hipLaunchKernelGGL((k_agg_gemm2<U_, FUSED, GP_, NG_>), grid2, dim3(64 * (NG_ + 4)), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles)
#define CB_AG2A(U_, GP_) hipLaunchKernelGGL((k_agg_gemm2<U_, FUSED, GP_, 8, false, true>), grid2, dim3(64 * 12), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles)
  if constexpr (!FUSED) {
    if (gt.out2) {      // + the trunk backward of the layer below in the dense tail's epilogue
      static const bool tb_async = !(getenv("CB_AGG_GEMM_SYNC") && !strcmp(getenv("CB_AGG_GEMM_SYNC"), "barrier"));
      if (tb_async) {
        if (ep.col_flags)
          hipLaunchKernelGGL((k_agg_gemm2<12, false, 2, 8, true, true>), grid2, dim3(64 * 12), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
        else
          hipLaunchKernelGGL((k_agg_gemm2<12, false, 0, 8, true, true>), grid2, dim3(64 * 12), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
      } else if (ep.col_flags)
        hipLaunchKernelGGL((k_agg_gemm2<8, false, 2, 8, true>), grid2, dim3(64 * 12), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
      else
        hipLaunchKernelGGL((k_agg_gemm2<8, false, 0, 8, true>), grid2, dim3(64 * 12), 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
      CB_LAUNCH_CHECK();
      return CB_OK;
    }
  }
  if (async && ng == 8) {
    if (u16 == 16) { if (ep.col_flags) CB_AG2A(16, 2); else CB_AG2A(16, 0); }
    else if (u16 == 12) { if (ep.col_flags) CB_AG2A(12, 2); else CB_AG2A(12, 0); }
    else { if (ep.col_flags) CB_AG2A(8, 2); else CB_AG2A(8, 0); }
  } else if (ng == 8) {
    if (u16 == 16) { if (ep.col_flags) CB_AG2(16, 2, 8); else CB_AG2(16, 0, 8); }
    else { if (ep.col_flags) CB_AG2(8, 2, 8); else CB_AG2(8, 0, 8); }
  } else {
    if (u16 == 8) { if (ep.col_flags) CB_AG2(8, 2, 4); else CB_AG2(8, 0, 4); }
    else { if (ep.col_flags) CB_AG2(16, 2, 4); else CB_AG2(16, 0, 4); }
  }
#undef CB_AG2
#undef CB_AG2A
  CB_LAUNCH_CHECK();
  return CB_OK;
}

static inline bool ag_al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
static int ag_dbg() {
  static const int v = getenv("CB_AGG_GEMM_DBG") ? atoi(getenv("CB_AGG_GEMM_DBG")) : 0;
  return v;
}

}  // namespace cb

using namespace cb;

// The persistent kernels launch one block per CU.  On a stream confined to a CU subset (hipExtStreamCreateWithCUMask: the backward runs its
// weight-gradient GEMMs beside the aggregation chain on disjoint CU sets) more blocks than CUs would queue behind whole blocks: the host tells
// how many CUs the stream has.  0 restores "all CUs of the device".  Not thread-safe (one training loop per process, as in the reference).
extern "C" int cb_agg_gemm_set_cu_limit(int32_t n_cus) {
  CB_CHECK_ARG(n_cus >= 0 && n_cus <= 1024, CB_E_INVALID, "cb_agg_gemm_set_cu_limit: 0 .. 1024");
  g_cu_limit = n_cus;
  return CB_OK;
}

// A stream whose kernels run on the CUs whose bit is set in mask[0 .. words) only (bit i of word j = CU 32 j + i).
extern "C" int cb_stream_create_cu_mask(const uint32_t* mask, int32_t words, void** stream) {
  CB_CHECK_ARG(mask && stream && words > 0 && words <= 32, CB_E_INVALID, "cb_stream_create_cu_mask: bad argument");
  hipStream_t st = nullptr;
  CB_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  *stream = (void*)st;
  return CB_OK;
}

extern "C" size_t cb_agg_gemm_image_bytes(int64_t K, int64_t N) {
  if (K != kKD || N != kND) return 0;
  return (size_t)kNS * kNT * 3 * 64 * sizeof(uint4);
}

extern "C" int cb_agg_gemm_image_f32(const float* W, int64_t ld, int64_t K, int64_t N, int transpose, void* image, size_t image_bytes,
                                     void* stream) {
  CB_CHECK_ARG(K == kKD && N == kND, CB_E_INVALID, "cb_agg_gemm_image_f32: the fused dense part is built for 256 x 256 weights (got %lld x %lld)",
               (long long)K, (long long)N);
  CB_CHECK_ARG(W && image && ld >= (transpose ? K : N), CB_E_INVALID, "cb_agg_gemm_image_f32: null pointer / bad leading dimension");
  CB_CHECK_ARG(image_bytes >= cb_agg_gemm_image_bytes(K, N) && ag_al16(image), CB_E_WORKSPACE, "cb_agg_gemm_image_f32: image buffer too small or misaligned");
  // B[k][n] = W[k][n] (transpose = 0) or W[n][k] (transpose = 1: the dX contraction multiplies by W^T)
  const int64_t sk = transpose ? 1 : ld, sn = transpose ? ld : 1;
  hipLaunchKernelGGL(k_agg_gemm_image, dim3(kNS * kNT * 64 / 256), dim3(256), 0, (hipStream_t)stream, W, sk, sn, (uint4*)image);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

static int agg_gemm_common_checks(const char* who, int64_t N, int64_t E, int64_t d, const void* rowptr, const void* col, const void* h, int64_t ld_h,
                                  int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const void* hub_rows, const void* hub_chunk_ptr, const void* ws,
                                  size_t ws_bytes, const void* image, const float* g_addend, int64_t ld_add, const float* g_out, int64_t ld_gout) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d == kKD, CB_E_INVALID, "%s: the fused dense part needs d == 256", who);
  CB_CHECK_ARG(N < INT32_MAX - kTM && E < INT32_MAX, CB_E_RANGE, "%s: size exceeds the int32 contract", who);
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && image && g_out && (E == 0 || col), CB_E_INVALID, "%s: null pointer", who);
  CB_CHECK_ARG(ag_al16(h) && ld_h % 4 == 0 && ld_h >= d && ag_al16(image) && ag_al16(g_out) && ld_gout % 4 == 0 && ld_gout >= kND &&
                   (!g_addend || (ag_al16(g_addend) && ld_add % 4 == 0 && ld_add >= kND)),
               CB_E_INVALID, "%s: 16-byte aligned rows of at least 256 floats required", who);
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "%s: bad hub plan", who);
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= (size_t)n_chunks * ag_partial_ld(d) * sizeof(float)), CB_E_WORKSPACE,
               "%s: hub plan given but workspace missing/too small", who);
  return CB_OK;
}

// Plain aggregation (out = act(row_scale * sum + bias), as cb_spmm_csr_f32) + g_out = g_rowscale * (out @ B) + g_addend, B = the
// 256 x 256 matrix whose fragment image cb_agg_gemm_image_f32 wrote.
extern "C" int cb_spmm_gemm_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                int64_t d, const float* row_scale, const float* bias, int relu, float* out, int64_t ld_out, int32_t hub_T,
                                int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes,
                                const void* image, const float* g_rowscale, const float* g_addend, int64_t ld_add, float* g_out, int64_t ld_gout,
                                void* stream) {
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws,
                                        ws_bytes, image, g_addend, ld_add, g_out, ld_gout);
  if (rc != CB_OK || N == 0) return rc;
  CB_CHECK_ARG(out && ag_al16(out) && ld_out % 4 == 0 && ld_out >= d, CB_E_INVALID, "cb_spmm_gemm_f32: 16-byte aligned output rows required");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, bias, relu, nullptr, 0, col_flags};
  GemmTail gt{(const uint4*)image, g_rowscale, g_addend, ld_add, g_out, ld_gout};
  gt.dbg = ag_dbg();
  return launch_agg_gemm<false>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                (hipStream_t)stream, FusedEpi{}, gt);
}

// cb_spmm_gemm_f32 + the trunk backward of the layer below from the dense tail's epilogue (TB): g_out = g_rowscale * (out @ B) is dL/dx of
// the stage above layer l-1, and gr_out = c_act * dropout_bwd_{seed}(g_out) * relu_bits * rowscale2 (the input of the next reverse
// aggregation) with colsum = column sums of the same without rowscale2 (that layer's bias gradient) — cb_trunk_layer_bwd_f32 without
// its 10 GB read of g_out.  ws2: cb_spmm_gemm_trunkbwd_workspace_bytes() for the per-block partial column sums.
extern "C" size_t cb_spmm_gemm_trunkbwd_workspace_bytes(void) { return (size_t)1024 * kND * sizeof(float); }

extern "C" int cb_spmm_gemm_trunkbwd_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                         int64_t ld_h, int64_t d, float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                         const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* image,
                                         const float* g_rowscale, float* g_out, int64_t ld_gout, const uint64_t* relu_bits, float c_act,
                                         float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, const float* rowscale2,
                                         float* gr_out, int64_t ld_gr, float* colsum, void* ws2, size_t ws2_bytes, int32_t g_masked, void* stream) {
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_trunkbwd_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                        ws, ws_bytes, image, nullptr, 0, g_out, ld_gout);
  if (rc != CB_OK) return rc;
  if (N == 0) {
    if (colsum) CB_HIP(hipMemsetAsync(colsum, 0, kND * sizeof(float), (hipStream_t)stream));
    return CB_OK;
  }
  CB_CHECK_ARG(out && ag_al16(out) && ld_out % 4 == 0 && ld_out >= d && relu_bits && gr_out && ag_al16(gr_out) && ld_gr % 4 == 0 && ld_gr >= kND &&
                   drop_p >= 0.f && drop_p < 1.f && row0 >= 0,
               CB_E_INVALID, "cb_spmm_gemm_trunkbwd_f32: null pointer, misaligned rows or bad p");
  CB_CHECK_ARG(!colsum || (ws2 && ws2_bytes >= cb_spmm_gemm_trunkbwd_workspace_bytes()), CB_E_WORKSPACE, "cb_spmm_gemm_trunkbwd_f32: workspace too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{nullptr, nullptr, 0, nullptr, 0, col_flags};
  GemmTail gt{(const uint4*)image, g_rowscale, nullptr, 0, g_out, ld_gout};
  gt.bits = (const unsigned long long*)relu_bits; gt.c_act = c_act; gt.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  gt.keep_scale = 1.f / (1.f - drop_p); gt.seed = seed; gt.seed_dev = seed_dev; gt.row0 = row0; gt.rowscale2 = rowscale2;
  gt.out2 = gr_out; gt.ld_out2 = ld_gr; gt.colsum_partial = colsum ? (float*)ws2 : nullptr; gt.out_masked = g_masked; gt.dbg = ag_dbg();
  hipStream_t st = (hipStream_t)stream;
  const int rc2 = launch_agg_gemm<false>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws, st,
                                         FusedEpi{}, gt);
  if (rc2 != CB_OK) return rc2;
  if (colsum) {
    const int n_tiles = (int)((N + kTM - 1) / kTM);
    hipLaunchKernelGGL(k_agg_colsum_finish, dim3(1), dim3(256), 0, st, (const float*)ws2, ag_n_blocks(n_tiles), colsum);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// Fused trunk store (cb_spmm_csr_fused_f32: ReLU / mix / dropout, mask words, out_next) + g_out = g_rowscale * (out_next @ B) + g_addend.
extern "C" int cb_spmm_gemm_fused_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                      int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                                      float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                      uint64_t* relu_bits, float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                      const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* image,
                                      const float* g_rowscale, const float* g_addend, int64_t ld_add, float* g_out, int64_t ld_gout, void* stream) {
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_fused_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                        ws, ws_bytes, image, g_addend, ld_add, g_out, ld_gout);
  if (rc != CB_OK || N == 0) return rc;
  CB_CHECK_ARG(out_next && ag_al16(out_next) && ld_next % 4 == 0 && ld_next >= d && (!mix_src || (ag_al16(mix_src) && ld_mix % 4 == 0)), CB_E_INVALID,
               "cb_spmm_gemm_fused_f32: 16-byte aligned rows required");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_spmm_gemm_fused_f32: dropout p out of range");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, bias, 1, nullptr, 0, col_flags};
  FusedEpi fe{};
  fe.mix_src = mix_src; fe.ld_mix = ld_mix; fe.c_act = c_act; fe.c_mix = c_mix;
  fe.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fe.keep_scale = 1.f / (1.f - drop_p);
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits;
  fe.out_act = nullptr; fe.ld_act = 0; fe.out_next = out_next; fe.ld_next = ld_next; fe.d = (int)d;
  GemmTail gt{(const uint4*)image, g_rowscale, g_addend, ld_add, g_out, ld_gout};
  gt.dbg = ag_dbg();
  return launch_agg_gemm<true>(rowptr, col, N, h, ld_h, ep, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                               (hipStream_t)stream, fe, gt);
}
