// Aggregation + the NEXT dense transform in one kernel ("the matrix cores run under the gathers").
//
// Replaces, per layer of the residual trunk, the pair
//     forward :  X_{l+1} = store( A^T Z_l )                     GNN_model/GCN.py:238-253,127-133   (cb_spmm_csr_fused_f32)
//                Z_{l+1} = a . (X_{l+1} W_{l+1}) + E_{l+1}      GNN_model/GCN.py:213,225,230-235    (cb_gemm_nn_f32)
//     backward:  dZ_l    = A (b . dY'_l)                        autograd of :238                    (cb_spmm_csr_f32, reverse CSR)
//                dX_l    = a . (dZ_l W_l^T)                     autograd of :213,225                (cb_gemm_nn_f32)
// by ONE launch each: one persistent block of 12 wavefronts per CU.  Wavefronts 0-7 aggregate 64-row tiles exactly as k_spmm_rows does
// (same edge-stream walk, same stores: X_{l+1} / dZ_l still go to memory, the weight-gradient GEMM needs them) and leave every finished row
// in one of two fp32 LDS tiles (64 x 256, 65 KB each); wavefronts 8-11 multiply the other tile by the 256 x 256 weight and store.  The tile
// buffers change hands through two LDS counters per buffer (no block barrier).  What disappears: the second kernel's 10 GB read of the
// matrix just written, and the matrix cores' time as a term of its own (profiles/r03_fused_agg_gemm.md).
//
// Arithmetic of the dense part = cb_gemm_limb.hip's, product by product: fp32 operands as three exact bf16 limbs, the six leading
// limb products per K step in the same order into fp32 MFMA accumulators, `rowscale * acc + addend` on the way out — results are
// bit-identical to cb_gemm_nn_f32 on the same inputs (tests/test_gpu_agg_gemm.py, tests/test_gpu_fullsize.py at 10^7 rows).
//   A operand: the LDS tile; a fragment (8 consecutive k of one row) = two ds_read_b128, split into limbs in registers
//              (1040-byte tile rows: the 16 lanes of a b128 group hit 16 distinct 16-byte bank columns);
//   B operand: the weight, split ONCE per launch by k_weight_image into MFMA fragment order (384 KB, L2 resident): a fragment is
//              one coalesced global_load_dwordx4 per limb, no LDS, no conversion in the K loop;
//   C: accumulators -> wave-private LDS strips -> row-major float4 -> epilogue -> 256-byte streaming row segments.
// Hub rows (more edges than the hub threshold) are reduced by the hub kernels, which run BEFORE this kernel; their finished rows are
// read back from memory into the tile.
// ACC forms (node-sharded path, dist.py): the reduction of a row starts from the partial sums of the earlier passes (interior columns,
// earlier halo slices) — the LAST halo pass of a rank's aggregation then also produces the next layer's Z / this layer's dX.
#include <stdlib.h>
#include <string.h>

#include "cb_common.h"
#include "cb_limb_core.h"
#include "cb_spmm_core.h"
#include "cb_tile_gemm.h"

namespace cb {

struct GemmTail {
  const uint4* image;      // weight limbs in fragment order (k_weight_image)
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
  int* err;                // device-visible error word (cb_error.hip): a tile hand-over that timed out is recorded here, never silent
  // NARROW kernels only (the output Linear as the tail of the last layer's aggregation, GCN.py:133-138): out[m][n] = acc + bias[n], n < n_out <= 64
  const float* bias;       // [n_out] or null
  int n_out;
};

// (declared in cb_tile_gemm.h)
__global__ void __launch_bounds__(256) k_weight_image(const float* __restrict__ W, int64_t sk, int64_t sn, uint4* __restrict__ image, int n_steps) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = idx & 63, sj = idx >> 6, j = sj % kNT, s = sj / kNT;
  if (s >= n_steps) return;
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

// B[k][n] = W[k * sk + n * sn] for n < n_cols, zero beyond: the narrow image (kNTn = 2 column blocks) of a 256 x C weight, C <= 64
__global__ void __launch_bounds__(256) k_weight_image_narrow(const float* __restrict__ W, int64_t sk, int64_t sn, uint4* __restrict__ image, int n_steps,
                                                             int n_cols) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = idx & 63, sj = idx >> 6, j = sj % kNTn, s = sj / kNTn;
  if (s >= n_steps) return;
  const int k0 = 16 * s + 8 * (lane >> 5), n = 32 * j + (lane & 31);
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float w0 = n < n_cols ? W[(int64_t)(k0 + 2 * e) * sk + (int64_t)n * sn] : 0.f;
    const float w1 = n < n_cols ? W[(int64_t)(k0 + 2 * e + 1) * sk + (int64_t)n * sn] : 0.f;
    split3x2(w0, w1, h[e], m[e], l[e]);
  }
  uint4* o = image + ((int64_t)(s * kNTn + j) * 3) * 64 + lane;
  o[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o[64] = make_uint4(m[0], m[1], m[2], m[3]);
  o[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

// Hand-over of the LDS tile buffers WITHOUT a block barrier: two counters per buffer in LDS.
//   ready[b]: +1 by every gathering wavefront that has written its rows of the tile in buffer b   (tile complete at kNG * (use + 1))
//   freed[b]: +1 by every multiplying wavefront that has read the tile in buffer b for the last time (buffer reusable at 4 * use)
// A wavefront's LDS instructions are executed in order, so a counter increment issued after the tile writes (or reads) is seen after them;
// the asm statements only keep the COMPILER from moving LDS accesses across the hand-over.  No wait on outstanding global loads / stores
// (a block barrier drains them): a gathering wavefront that has finished its rows moves on to the next tile while its row stores are still
// in flight and while the other wavefronts finish theirs, so the eight gathering wavefronts of a CU drift apart and cover each other's
// start-of-tile latencies (rowptr -> column ids -> first gathers), which a barrier lines up.
// Spins are bounded so that a lost hand-over can never hang the GPU — and it is never silent: the wavefront that gives up records
// CB_DEVERR_HANDOVER (+ block and counter target) in the device error word; every later cb_spmm_gemm_* call and cb_device_status()
// then return CB_E_DEVICE (the results of the launch that timed out are invalid).
constexpr int kSpinLimit = 1 << 26;      // ~7 s of s_sleep
__device__ __forceinline__ void ag_wait(int* flag, int target, int* err, int spin_limit = kSpinLimit) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > spin_limit) {
      if (err && lane_id() == 0) {
        __hip_atomic_store(err + 1, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err + 2, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err, CB_DEVERR_HANDOVER, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      break;
    }
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ag_signal(int* flag, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- one persistent 12-wavefront block per CU, wavefront-specialised ------------------------------------------------------------
// Two co-resident blocks that each alternate gather / MFMA phases fall into lock step (both gather, then both multiply: measured,
// profiles/r03_fused_agg_gemm.md), so nothing overlaps.  Here the roles are fixed instead: wavefronts 0-7 only gather (tile t+1
// into one LDS buffer), wavefronts 8-11 only multiply (tile t from the other buffer) and store.  The 64 rows of a tile are cut among the
// gathering wavefronts at equal EDGE counts (row boundaries from the tile's rowptr values, ballots), so that they finish together —
// a fixed 8 rows each leaves the tile waiting for the wavefront with the heaviest rows.
constexpr int kNG = 8;       // gathering wavefronts (3 wavefronts per SIMD at <= 168 registers; 12 gathers in flight each)
constexpr int kU = 12;       // gathers in flight per gathering wavefront (16: spills; 8: slower — profiles/r03_fused_agg_gemm.md)

template <bool FUSED, int GP, bool ACC>
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
  // this wavefront's rows [ra, rb): the rows whose first edge falls into its share of the tile's edge range
  const int e0 = bcast_lane(my_ptr, 0), e1 = ptr_hi;
  const int64_t span = (int64_t)e1 - e0;
  const int ta = e0 + (int)(span * w / kNG), tb = e0 + (int)(span * (w + 1) / kNG);
  const int ra = w == 0 ? 0 : (int)__popcll(__ballot(lane < nrt && my_ptr < ta));
  const int rb = w == kNG - 1 ? nrt : (int)__popcll(__ballot(lane < nrt && my_ptr < tb));
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (ep.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bvec[i] = ep.bias[c0 + i];
  }
  const float* h_lane = h + c0;
  float* out_lane = out + c0;
  float* tile_lane = tile + c0;
  const float* init_lane = ACC ? ep.acc_init + c0 : nullptr;
  int r = ra;
  while (r < rb) {      // maximal hub-free runs of [ra, rb)
    const unsigned long long m = (hubmask >> r) & (rb - r >= 64 ? ~0ull : ((1ull << (rb - r)) - 1ull));
    const int nh = m ? r + (__ffsll((long long)m) - 1) : rb;
    if (nh > r)
      stream_rows<4, kU, true, FUSED, ACC, float, GP, kTLD, true>(r, nh, nrt, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, true,
                                                                         ep.relu, bvec, fe, c0, init_lane, ep.ld_init, ep, tile_lane, ptr_hi);
    if (nh < rb) {      // hub row nh: finished by the hub kernels, which ran before this launch
      const float* src = FUSED ? fe.out_next + (int64_t)(r0 + nh) * fe.ld_next + c0 : out + (int64_t)(r0 + nh) * ld_out + c0;
      *reinterpret_cast<float4*>(tile_lane + nh * kTLD) = *reinterpret_cast<const float4*>(src);
    }
    r = nh + 1;
  }
  if (w == kNG - 1) {
    for (int i = nrt; i < kTM; ++i) *reinterpret_cast<float4*>(tile_lane + i * kTLD) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// One K step of B fragments in flight per multiplying wavefront (two register buffers, K loop unrolled by two): next to wavefronts that
// keep dozens of gathers outstanding, a load of this CU — L2 hit or not — comes back after microseconds.
template <bool TB>
__device__ __forceinline__ void ag2_mfma_tile(int t, const float* __restrict__ tile, float* __restrict__ cs, int w, int lane, int n_rows,
                                              const GemmTail& gt, float (&colsum)[4], uint64_t seed_eff, int* freed) {
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  tile_times_image<kNS, kTLD>(tile, gt.image, w, lane, acc);
  ag_signal(freed, lane);      // the tile has been read for the last time
  // epilogue through a WAVE-PRIVATE staging strip (the tile itself is still being read by the other three multiplying wavefronts
  // and there is no barrier among four of twelve wavefronts): 8 rows x 64 columns per pass, transposed so that a lane applies
  // `rowscale * acc + addend` on a float4 and the strip leaves as 256-byte row segments
  const int r0 = t * kTM;
  const float zero_bias = 0.f;      // (keeps the epilogue expression of cb_gemm_core.h's nn_epilogue: o * rs + addend + bias)
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
          }
          store_stream<4>(gt.out + m * gt.ld_out + n, o);
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

// NARROW tail: multiplying wavefront w takes the 32 x 32 block (rows 32 (w & 1), columns 32 (w >> 1)) of the tile's 64 x 64 output = the
// logits of 64 rows (C <= 64 classes): out = acc + bias, the epilogue expression of cb_gemm_nn_f32 with a bias and neither row scale nor addend.
__device__ __forceinline__ void ag2_mfma_tile_head(int t, const float* __restrict__ tile, float* __restrict__ cs, int w, int lane, int n_rows,
                                                   const GemmTail& gt, int* freed) {
  const int l31 = lane & 31, lh = lane >> 5, bi = w & 1, bj = w >> 1;
  f32x16 acc;
  tile_times_image_block<kNS, kTLD>(tile, gt.image, bi, bj, lane, acc);
  ag_signal(freed, lane);      // the tile has been read for the last time
  const int r0 = t * kTM + 32 * bi;
#pragma unroll
  for (int q = 0; q < 4; ++q) {      // 8 rows x 32 columns per pass through the wave-private strip
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) cs[(r4 + 4 * lh) * kCLD + l31] = acc[4 * q + r4];
    const int row = lane >> 3, c4 = (lane & 7) * 4;
    const float4 v = *reinterpret_cast<const float4*>(cs + row * kCLD + c4);
    const int64_t m = r0 + 8 * q + row;
    const int n = 32 * bj + c4;
    if (m < n_rows && n < gt.n_out) {
      const float zero_add = 0.f;
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = o[e] * 1.f + zero_add + ((gt.bias && n + e < gt.n_out) ? gt.bias[n + e] : 0.f);
      float* dst = gt.out + m * gt.ld_out + n;
      if (n + 4 <= gt.n_out && (gt.ld_out & 3) == 0) {
        store_stream<4>(dst, o);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < gt.n_out) dst[e] = o[e];
      }
    }
  }
}

template <bool FUSED, int GP, bool ACC, bool TB, bool NARROW = false>
__global__ void __launch_bounds__(64 * (kNG + 4), (kNG + 4) / 4) k_agg_gemm2(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                            const float* __restrict__ h, int64_t ld_h, float* __restrict__ out,
                                                                            int64_t ld_out, int n_rows, Epilogue ep, int hub_T, FusedEpi fe,
                                                                            GemmTail gt, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float tiles[2][kTM * kTLD];
  __shared__ __attribute__((aligned(16))) float cstrip[4][8 * kCLD];
  __shared__ int ready[2], freed[2];
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform by construction: tell the compiler, so that what derives from it stays in SGPRs)
  const bool gathers = wv < kNG;
  const int w = gathers ? wv : wv - kNG;
  const int n_it = ((int)blockIdx.x < n_tiles) ? (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};      // TB: sums of this lane's 4 output columns (64 w + 4 (lane & 15) ..) over the rows it stored
  const uint64_t seed_eff = TB ? (gt.seed_dev ? gt.seed + *gt.seed_dev : gt.seed) : 0ull;
  if (threadIdx.x < 2) ready[threadIdx.x] = freed[threadIdx.x] = 0;
  __syncthreads();
  if (gathers) {
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1, use = it >> 1;
      if (use > 0) ag_wait(&freed[b], 4 * use, gt.err);
      ag2_gather_tile<FUSED, GP, ACC>(blockIdx.x + it * gridDim.x, tiles[b], w, lane, rowptr, col, h, ld_h, out, ld_out, n_rows, ep, hub_T, fe);
      ag_signal(&ready[b], lane);
    }
  } else {
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1, use = it >> 1;
      ag_wait(&ready[b], kNG * (use + 1), gt.err);
      if constexpr (NARROW) ag2_mfma_tile_head(blockIdx.x + it * gridDim.x, tiles[b], cstrip[w], w, lane, n_rows, gt, &freed[b]);
      else ag2_mfma_tile<TB>(blockIdx.x + it * gridDim.x, tiles[b], cstrip[w], w, lane, n_rows, gt, colsum, seed_eff, &freed[b]);
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

// Fault injection for the hand-over's failure path (tests only reach it through cb_agg_gemm_handover_selftest): one wavefront waits, with a
// short spin bound, for a counter nobody increments.
__global__ void __launch_bounds__(64) k_agg_handover_selftest(int* err) {
  __shared__ int never;
  if (threadIdx.x == 0) never = 0;
  __syncthreads();
  ag_wait(&never, 1, err, 1 << 10);
}

static inline int64_t ag_partial_ld(int64_t d) { return (d + 3) / 4 * 4; }
constexpr int kMaxBlocks = 1024;      // (cb_spmm_gemm_trunkbwd_workspace_bytes: one partial row per block)

// blocks of the persistent kernel: one per CU of the current device (139 KB of LDS each)
static int ag_n_blocks(int n_tiles) {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  if (n_cu > kMaxBlocks) n_cu = kMaxBlocks;
  return n_tiles < n_cu ? n_tiles : n_cu;
}

template <bool FUSED, bool ACC>
static int launch_agg_gemm(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, Epilogue ep, float* out,
                           int64_t ld_out, int hub_T, int n_hubs, int n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                           float* partial, hipStream_t st, FusedEpi fe, GemmTail gt) {
  const int d = kKD;
  const dim3 blk(256);
  int* err = device_error_word();
  CB_CHECK_ARG(err != nullptr, CB_E_HIP, "cb_spmm_gemm: the device error word could not be allocated (%s)", cb_last_error());
  CB_CHECK_ARG(*(volatile int*)err == 0, CB_E_DEVICE, "cb_spmm_gemm: an earlier launch recorded a device-side error (%s); results since then are invalid",
               device_error_text());
  gt.err = err;
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
    FusedEpi fe_hub = fe;
    fe_hub.skip_next = 0;      // hub rows always go through memory: the persistent kernel reads them back into its tile
    hipLaunchKernelGGL((k_spmm_hub_finish<4, FUSED>), grid2, blk, 0, st, d, n_hubs, hub_rows, hub_chunk_ptr, partial, ld_p, out, ld_out, ep, fe_hub);
    CB_LAUNCH_CHECK();
  }
  const int n_tiles = (int)((N + kTM - 1) / kTM);
  const dim3 grid((unsigned)ag_n_blocks(n_tiles)), block(64 * (kNG + 4));
#define CB_AG2(GP_, TB_) \
  hipLaunchKernelGGL((k_agg_gemm2<FUSED, GP_, ACC, TB_>), grid, block, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles)
  bool tb = false, narrow = false;
  if constexpr (!FUSED) tb = gt.out2 != nullptr;      // + the trunk backward of the layer below in the dense tail's epilogue
  if constexpr (!FUSED) {
    if (tb) { if (ep.col_flags) CB_AG2(2, true); else CB_AG2(0, true); }
  }
  if constexpr (FUSED) {      // the output Linear (<= 64 classes) as the tail: gt.n_out > 0
    narrow = gt.n_out > 0;
    if (narrow) {
      if (ep.col_flags) hipLaunchKernelGGL((k_agg_gemm2<true, 2, ACC, false, true>), grid, block, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
      else hipLaunchKernelGGL((k_agg_gemm2<true, 0, ACC, false, true>), grid, block, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, ep, hub_T, fe, gt, n_tiles);
    }
  }
  if (!tb && !narrow) { if (ep.col_flags) CB_AG2(2, false); else CB_AG2(0, false); }
#undef CB_AG2
  CB_LAUNCH_CHECK();
  return CB_OK;
}

static inline bool ag_al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace cb

using namespace cb;

extern "C" int cb_agg_gemm_handover_selftest(void* stream) {
  int* err = device_error_word();
  CB_CHECK_ARG(err != nullptr, CB_E_HIP, "cb_agg_gemm_handover_selftest: the device error word could not be allocated");
  hipLaunchKernelGGL(k_agg_handover_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, err);
  CB_LAUNCH_CHECK();
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
  hipLaunchKernelGGL(k_weight_image, dim3(kNS * kNT * 64 / 256), dim3(256), 0, (hipStream_t)stream, W, sk, sn, (uint4*)image, kNS);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

static int agg_gemm_common_checks(const char* who, int64_t N, int64_t E, int64_t d, const void* rowptr, const void* col, const void* h, int64_t ld_h,
                                  int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const void* hub_rows, const void* hub_chunk_ptr, const void* ws,
                                  size_t ws_bytes, const void* image, const float* g_addend, int64_t ld_add, const float* g_out, int64_t ld_gout,
                                  const float* acc_init, int64_t ld_init) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d == kKD, CB_E_INVALID, "%s: the fused dense part needs d == 256", who);
  CB_CHECK_ARG(N < INT32_MAX - kTM && E < INT32_MAX, CB_E_RANGE, "%s: size exceeds the int32 contract", who);
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && image && g_out && (E == 0 || col), CB_E_INVALID, "%s: null pointer", who);
  CB_CHECK_ARG(ag_al16(h) && ld_h % 4 == 0 && ld_h >= d && ag_al16(image) && ag_al16(g_out) && ld_gout % 4 == 0 && ld_gout >= kND &&
                   (!g_addend || (ag_al16(g_addend) && ld_add % 4 == 0 && ld_add >= kND)) &&
                   (!acc_init || (ag_al16(acc_init) && ld_init % 4 == 0 && ld_init >= d)),
               CB_E_INVALID, "%s: 16-byte aligned rows of at least 256 floats required", who);
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "%s: bad hub plan", who);
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= (size_t)n_chunks * ag_partial_ld(d) * sizeof(float)), CB_E_WORKSPACE,
               "%s: hub plan given but workspace missing/too small", who);
  return CB_OK;
}

// Plain aggregation (out = act(row_scale * (acc_init + sum) + bias), as cb_spmm_csr_f32 / cb_spmm_csr_acc_f32) + g_out = g_rowscale *
// (out @ B) + g_addend, B = the 256 x 256 matrix whose fragment image cb_agg_gemm_image_f32 wrote.  acc_init (may be NULL): the partial
// sums of the earlier passes of a node-sharded aggregation ([N, ld_init]); `out` may alias it (a row's partial sums are read before the
// row is stored, by the same wavefront).
extern "C" int cb_spmm_gemm_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init, float* out,
                                int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* image, const float* g_rowscale,
                                const float* g_addend, int64_t ld_add, float* g_out, int64_t ld_gout, void* stream) {
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws,
                                        ws_bytes, image, g_addend, ld_add, g_out, ld_gout, acc_init, ld_init);
  if (rc != CB_OK || N == 0) return rc;
  CB_CHECK_ARG(out && ag_al16(out) && ld_out % 4 == 0 && ld_out >= d, CB_E_INVALID, "cb_spmm_gemm_f32: 16-byte aligned output rows required");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, bias, relu, acc_init, ld_init, col_flags};
  GemmTail gt{(const uint4*)image, g_rowscale, g_addend, ld_add, g_out, ld_gout};
  if (acc_init)
    return launch_agg_gemm<false, true>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                        (hipStream_t)stream, FusedEpi{}, gt);
  return launch_agg_gemm<false, false>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                       (hipStream_t)stream, FusedEpi{}, gt);
}

// cb_spmm_gemm_f32 + the trunk backward of the layer below from the dense tail's epilogue (TB): g_out = g_rowscale * (out @ B) is dL/dx of
// the stage above layer l-1, and gr_out = c_act * dropout_bwd_{seed}(g_out) * relu_bits * rowscale2 (the input of the next reverse
// aggregation) with colsum = column sums of the same without rowscale2 (that layer's bias gradient) — cb_trunk_layer_bwd_f32 without
// its 10 GB read of g_out.  ws2: cb_spmm_gemm_trunkbwd_workspace_bytes() for the per-block partial column sums.
extern "C" size_t cb_spmm_gemm_trunkbwd_workspace_bytes(void) { return (size_t)kMaxBlocks * kND * sizeof(float); }

extern "C" int cb_spmm_gemm_trunkbwd_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                         int64_t ld_h, int64_t d, const float* acc_init, int64_t ld_init, float* out, int64_t ld_out, int32_t hub_T,
                                         int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                         size_t ws_bytes, const void* image, const float* g_rowscale, float* g_out, int64_t ld_gout,
                                         const uint64_t* relu_bits, float c_act, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                         const float* rowscale2, float* gr_out, int64_t ld_gr, float* colsum, void* ws2, size_t ws2_bytes,
                                         void* stream) {
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_trunkbwd_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                        ws, ws_bytes, image, nullptr, 0, g_out, ld_gout, acc_init, ld_init);
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
  Epilogue ep{nullptr, nullptr, 0, acc_init, ld_init, col_flags};
  GemmTail gt{(const uint4*)image, g_rowscale, nullptr, 0, g_out, ld_gout};
  gt.bits = (const unsigned long long*)relu_bits; gt.c_act = c_act; gt.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  gt.keep_scale = 1.f / (1.f - drop_p); gt.seed = seed; gt.seed_dev = seed_dev; gt.row0 = row0; gt.rowscale2 = rowscale2;
  gt.out2 = gr_out; gt.ld_out2 = ld_gr; gt.colsum_partial = colsum ? (float*)ws2 : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc2 = acc_init ? launch_agg_gemm<false, true>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                                           (float*)ws, st, FusedEpi{}, gt)
                           : launch_agg_gemm<false, false>(rowptr, col, N, h, ld_h, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                                            (float*)ws, st, FusedEpi{}, gt);
  if (rc2 != CB_OK) return rc2;
  if (colsum) {
    const int n_tiles = (int)((N + kTM - 1) / kTM);
    hipLaunchKernelGGL(k_agg_colsum_finish, dim3(1), dim3(256), 0, st, (const float*)ws2, ag_n_blocks(n_tiles), colsum);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// Fused trunk store (cb_spmm_csr_fused_f32 / cb_spmm_csr_fused_acc_f32: ReLU / mix / dropout, mask words, out_next) + g_out = g_rowscale *
// (out_next @ B) + g_addend.
static int spmm_gemm_fused_impl(int skip_next, const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                                int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias,
                                const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act,
                                float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                size_t ws_bytes, const void* image, const float* g_rowscale, const float* g_addend, int64_t ld_add,
                                float* g_out, int64_t ld_gout, void* stream, const float* head_bias = nullptr, int n_out = 0) {
  // (n_out > 0: the narrow tail — g_out = logits [N, ld_gout >= n_out]; the common checks see a stand-in leading dimension)
  const int rc = agg_gemm_common_checks("cb_spmm_gemm_fused_f32", N, E, d, rowptr, col, h, ld_h, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                                        ws, ws_bytes, image, g_addend, ld_add, g_out, n_out > 0 ? kND : ld_gout, acc_init, ld_init);
  if (rc != CB_OK || N == 0) return rc;
  // (the evaluation form writes out_next for hub rows only: without a hub plan it may be NULL)
  CB_CHECK_ARG((out_next || (skip_next && n_hubs == 0)) && ag_al16(out_next) && ld_next % 4 == 0 && ld_next >= d &&
                   (!mix_src || (ag_al16(mix_src) && ld_mix % 4 == 0)) && (!out_act || (ag_al16(out_act) && ld_act % 4 == 0 && ld_act >= d)),
               CB_E_INVALID, "cb_spmm_gemm_fused_f32: 16-byte aligned rows required");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_spmm_gemm_fused_f32: dropout p out of range");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, bias, 1, acc_init, ld_init, col_flags};
  FusedEpi fe{};
  fe.mix_src = mix_src; fe.ld_mix = ld_mix; fe.c_act = c_act; fe.c_mix = c_mix;
  fe.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fe.keep_scale = 1.f / (1.f - drop_p);
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits; fe.bits_relu_only = bits_relu_only;
  fe.out_act = out_act; fe.ld_act = ld_act; fe.out_next = out_next; fe.ld_next = ld_next; fe.d = (int)d;
  fe.skip_next = skip_next;
  GemmTail gt{(const uint4*)image, g_rowscale, g_addend, ld_add, g_out, ld_gout};
  gt.bias = head_bias; gt.n_out = n_out;
  if (acc_init)
    return launch_agg_gemm<true, true>(rowptr, col, N, h, ld_h, ep, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                       (hipStream_t)stream, fe, gt);
  return launch_agg_gemm<true, false>(rowptr, col, N, h, ld_h, ep, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                      (hipStream_t)stream, fe, gt);
}

#define CB_SGF_PARAMS                                                                                                                              \
  const float *acc_init, int64_t ld_init, const int32_t *rowptr, const int32_t *col, int32_t col_flags, int64_t N, int64_t E, const float *h,      \
      int64_t ld_h, int64_t d, const float *row_scale, const float *bias, const float *mix_src, int64_t ld_mix, float c_act, float c_mix,          \
      float drop_p, uint64_t seed, const uint64_t *seed_dev, int64_t row0, uint64_t *relu_bits, int32_t bits_relu_only, float *out_act,          \
      int64_t ld_act, float *out_next, int64_t ld_next, int32_t hub_T,                                                                            \
      int32_t n_hubs, int32_t n_chunks, const int32_t *hub_rows, const int32_t *hub_chunk_ptr, void *ws, size_t ws_bytes, const void *image,       \
      const float *g_rowscale, const float *g_addend, int64_t ld_add, float *g_out, int64_t ld_gout, void *stream
#define CB_SGF_ARGS                                                                                                                                \
  acc_init, ld_init, rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed, seed_dev, row0,      \
      relu_bits, bits_relu_only, out_act, ld_act, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, image, g_rowscale, g_addend, ld_add, g_out,    \
      ld_gout, stream
extern "C" int cb_spmm_gemm_fused_f32(CB_SGF_PARAMS) { return spmm_gemm_fused_impl(0, CB_SGF_ARGS); }
// The same for a forward that no backward follows (evaluation / metrics passes): the stored activations X_{l+1} have no reader — the next
// layer's Z leaves this kernel —, so the rows the persistent kernel finishes stay on chip (out_next: still the hub rows' way into the
// tile, contents otherwise undefined; may be NULL when the plan has no hub rows).  10 GB less written per launch at the headline size.
extern "C" int cb_spmm_gemm_fused_eval_f32(CB_SGF_PARAMS) { return spmm_gemm_fused_impl(1, CB_SGF_ARGS); }
#undef CB_SGF_PARAMS
#undef CB_SGF_ARGS

// ---- the output Linear as the tail of the LAST layer's aggregation (round 5; GCN.py:133-138: Linear(dropout(X_L)) on the rows the store just made) ----
extern "C" size_t cb_agg_gemm_head_image_bytes(int64_t K, int64_t C) {
  if (K != kKD || C < 1 || C > 32 * kNTn) return 0;
  return (size_t)kNS * kNTn * 3 * 64 * sizeof(uint4);
}

// image of B = W^T for an nn.Linear weight W [C, 256] (transpose = 1) or of B = W [256, C] (transpose = 0); columns C .. 63 are zero
extern "C" int cb_agg_gemm_head_image_f32(const float* W, int64_t ld, int64_t K, int64_t C, int transpose, void* image, size_t image_bytes, void* stream) {
  CB_CHECK_ARG(cb_agg_gemm_head_image_bytes(K, C) > 0, CB_E_INVALID, "cb_agg_gemm_head_image_f32: K must be 256 and 1 <= C <= 64 (got %lld x %lld)", (long long)K,
               (long long)C);
  CB_CHECK_ARG(W && image && ld >= (transpose ? K : C), CB_E_INVALID, "cb_agg_gemm_head_image_f32: null pointer / bad leading dimension");
  CB_CHECK_ARG(image_bytes >= cb_agg_gemm_head_image_bytes(K, C) && ag_al16(image), CB_E_WORKSPACE, "cb_agg_gemm_head_image_f32: image buffer too small or misaligned");
  const int64_t sk = transpose ? 1 : ld, sn = transpose ? ld : 1;
  hipLaunchKernelGGL(k_weight_image_narrow, dim3(kNS * kNTn * 64 / 256), dim3(256), 0, (hipStream_t)stream, W, sk, sn, (uint4*)image, kNS, (int)C);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

#define CB_SGH_PARAMS                                                                                                                              \
  const float *acc_init, int64_t ld_init, const int32_t *rowptr, const int32_t *col, int32_t col_flags, int64_t N, int64_t E, const float *h,      \
      int64_t ld_h, int64_t d, const float *row_scale, const float *bias, const float *mix_src, int64_t ld_mix, float c_act, float c_mix,          \
      float drop_p, uint64_t seed, const uint64_t *seed_dev, int64_t row0, uint64_t *relu_bits, int32_t bits_relu_only, float *out_act,          \
      int64_t ld_act, float *out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t *hub_rows,                \
      const int32_t *hub_chunk_ptr, void *ws, size_t ws_bytes, const void *head_image, const float *head_bias, int64_t C, float *logits,          \
      int64_t ld_logits, void *stream
static int spmm_gemm_head_impl(int skip_next, CB_SGH_PARAMS) {
  CB_CHECK_ARG(C >= 1 && C <= 32 * kNTn && (N == 0 || (logits && ld_logits >= C)), CB_E_INVALID, "cb_spmm_gemm_fused_head_f32: 1 <= C <= 64 logits per row expected");
  // (the common checks want a 256-wide 16-byte aligned tail output: the narrow tail has its own rule — any ld >= C, float4 stores where ld % 4 == 0)
  CB_CHECK_ARG(N == 0 || ((uintptr_t)logits % 16) == 0, CB_E_INVALID, "cb_spmm_gemm_fused_head_f32: logits must be 16-byte aligned");
  return spmm_gemm_fused_impl(skip_next, acc_init, ld_init, rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed,
                              seed_dev, row0, relu_bits, bits_relu_only, out_act, ld_act, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws,
                              ws_bytes, head_image, nullptr, nullptr, 0, logits, ld_logits, stream, head_bias, (int)C);
}
// Fused trunk store of the LAST layer (as cb_spmm_gemm_fused_f32) + logits = out_next @ B + head_bias, B = the 256 x C matrix behind head_image.
extern "C" int cb_spmm_gemm_fused_head_f32(CB_SGH_PARAMS) {
  return spmm_gemm_head_impl(0, acc_init, ld_init, rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed, seed_dev, row0,
                             relu_bits, bits_relu_only, out_act, ld_act, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, head_image,
                             head_bias, C, logits, ld_logits, stream);
}
// The same for a forward that no backward follows: the last layer's activations are not written at all (hub rows excepted), only the logits leave.
extern "C" int cb_spmm_gemm_fused_head_eval_f32(CB_SGH_PARAMS) {
  return spmm_gemm_head_impl(1, acc_init, ld_init, rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed, seed_dev, row0,
                             relu_bits, bits_relu_only, out_act, ld_act, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, head_image,
                             head_bias, C, logits, ld_logits, stream);
}
#undef CB_SGH_PARAMS
