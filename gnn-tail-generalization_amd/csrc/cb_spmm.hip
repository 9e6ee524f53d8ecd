// CSR sum-aggregation for gfx950 — replaces DGL's gspmm behind
// `graph.update_all(fn.copy_src('h','m'), fn.sum('m','h'))` (GNN_model/GCN.py:198,238)
// with the `* norm` and `+ bias` of GCN.py:242-253 and the following ReLU (GCN.py:127-128)
// fused into the store.
//
// Bound: HBM.  Algorithmic bytes per launch = E*(d*4 + 4) + N*(d*4 + 4) [+ 4N row scale].
//
// Mapping (d = 256 fp32 is the tuned case):
//   * one 64-lane wavefront owns RPW consecutive destination rows; lane l owns columns
//     [VEC*l, VEC*l + VEC) of the 64*VEC-wide column tile, so one neighbour row is ONE
//     fully coalesced 1 KiB global_load_dwordx4 per wavefront (d = 256, VEC = 4);
//   * the rows' edges are contiguous in CSR, so the wavefront walks them as ONE edge
//     stream: 64 column ids per coalesced index load, wave-uniform broadcast
//     (v_readlane -> SGPR base address), U independent gathers in flight before the first
//     add, row boundaries handled while consuming (wave-uniform scalar compares).  Short
//     rows therefore do not serialise on the rowptr -> col -> gather latency chain;
//   * rows longer than the hub threshold (power-law hubs) are skipped here and reduced by
//     k_spmm_hub_chunks (one wavefront per chunk of T edges -> partial row in the
//     workspace) + k_spmm_hub_finish (sums a hub's partials in chunk order + epilogue).
//   * no atomics anywhere: every output element is produced by one lane in a fixed order,
//     so results are bit-reproducible run to run.
//   * streaming data (indices, output rows) uses non-temporal accesses so the 4 MiB L2s and
//     the 256 MiB Infinity Cache keep the re-used neighbour rows (hub sources).
#include <stdlib.h>
#include <string.h>

#include "cb_common.h"
#include "cb_philox.h"
#include "cb_spmm_small.h"

namespace cb {

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<2> {
  using T = float2;
};
template <>
struct Vec<4> {
  using T = float4;
};

template <int VEC>
__device__ __forceinline__ void zero(float (&a)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) a[i] = 0.f;
}

template <int VEC>
__device__ __forceinline__ void gather(float (&v)[VEC], const float* __restrict__ p) {
  using T = typename Vec<VEC>::T;
  T t = *reinterpret_cast<const T*>(p);
  if constexpr (VEC == 1) {
    v[0] = t;
  } else if constexpr (VEC == 2) {
    v[0] = t.x;
    v[1] = t.y;
  } else {
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
}

// Row that is read exactly once (the mixed-in X0 row of the fused store): keep it out of L2 / Infinity Cache, which hold the
// re-used hub source rows
template <int VEC>
__device__ __forceinline__ void gather_stream(float (&v)[VEC], const float* __restrict__ p) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = __builtin_nontemporal_load(p + i);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// fp32 x4 source row chunk with the streaming (nt) cache policy: one global_load_dwordx4 ... nt
__device__ __forceinline__ void gather_nt4(float (&v)[4], const float* __restrict__ p) {
  const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}

// bf16 x4 source row chunk (8 bytes), streaming policy
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gather_nt4_bf16(float (&v)[4], const void* __restrict__ p) {
  const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
  v[0] = __uint_as_float(t[0] << 16);
  v[1] = __uint_as_float(t[0] & 0xFFFF0000u);
  v[2] = __uint_as_float(t[1] << 16);
  v[3] = __uint_as_float(t[1] & 0xFFFF0000u);
}

// Gather policy of the source-row loads (GP): 0 = default cache policy for every row; 1 = every row streaming (nt);
// 2 = the CSR's column ids carry a "hot source" flag in bit 31 (set at graph build for the most-referenced source rows):
// hot rows default policy, all others streaming, so that the rows that ARE re-used keep L2 / Infinity Cache to themselves.
constexpr int kColMask = 0x7fffffff;

// Source-row load of the aggregation: fp32 rows, or bf16-stored rows widened to fp32 (accumulation stays fp32)
template <int VEC, typename HT>
__device__ __forceinline__ void gather_in(float (&v)[VEC], const HT* __restrict__ p) {
  if constexpr (sizeof(HT) == 4) {
    gather<VEC>(v, reinterpret_cast<const float*>(p));
  } else if constexpr (VEC == 4) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16);
    v[1] = __uint_as_float(t.x & 0xFFFF0000u);
    v[2] = __uint_as_float(t.y << 16);
    v[3] = __uint_as_float(t.y & 0xFFFF0000u);
  } else if constexpr (VEC == 2) {
    const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    v[0] = __uint_as_float(t << 16);
    v[1] = __uint_as_float(t & 0xFFFF0000u);
  } else {
    v[0] = bf16_to_f32(*reinterpret_cast<const bf16_t*>(p));
  }
}

template <int VEC>
__device__ __forceinline__ void store_stream(float* __restrict__ p, const float (&v)[VEC]) {
  // written once, read by a later kernel: keep it out of the caches
#pragma unroll
  for (int i = 0; i < VEC; ++i) __builtin_nontemporal_store(v[i], p + i);
}

struct Epilogue {
  const float* row_scale;  // [N] or null
  const float* bias;       // [d] or null
  int relu;
  // ACC kernels only: partial sums [N, ld_init] the reduction starts from (the interior-column pass of the node-sharded
  // path, dist.py: out = act(row_scale * (acc_init + sum over THIS CSR's columns) + bias))
  const float* acc_init;
  int64_t ld_init;
  int col_flags;           // 1: bit 31 of every column id marks a hot source row (gather policy 2, see gather_pol)
  // MASK kernels only (cb_spmm_csr_masked_f32): every gathered SOURCE row u enters the sum as src_scale[u] * (src_bits[u] ? h[u] : 0)
  // and the finished row is multiplied by out_coef — the backward of the fused trunk store (dropout keep & ReLU mask, c_act / (1-p),
  // the source row's degree norm) applied while the reverse aggregation gathers dL/dx_l, instead of in a pass of its own
  const unsigned long long* src_bits;   // [n_cols][d / 256][4] mask words of the forward store (word k, bit L <-> column 256 t + 4 L + k)
  const float* src_scale;               // [n_cols]
  float out_coef;
};

// Uniform (scalar-cache) reads of the per-source-row mask words and scale: the row id is wave-uniform, the arrays are read-only
// for the whole launch, so they are addressed through the constant address space (s_load)
typedef const __attribute__((address_space(4))) unsigned long long* ConstU64Ptr;
typedef const __attribute__((address_space(4))) float* ConstF32Ptr;

// lane l keeps v when bit l of the wave-uniform 64-bit word is set: the word IS the lane mask of one v_cndmask
__device__ __forceinline__ float keep_if_bit(float v, unsigned long long w) {
  float r;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(w));
  return r;
}

struct SrcRowMask {     // what the MASK kernels fetch per gathered source row (all wave-uniform: SGPRs)
  unsigned long long w[4];
  float s;
};
__device__ __forceinline__ SrcRowMask load_src_mask(const Epilogue& ep, int col_id, int tiles, int tile) {
  SrcRowMask m;
  ConstU64Ptr bw = (ConstU64Ptr)(ep.src_bits + ((int64_t)col_id * tiles + tile) * 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) m.w[k] = bw[k];
  m.s = ((ConstF32Ptr)ep.src_scale)[col_id];
  return m;
}

// Extended epilogue of the fused residual trunk (GCN.py:127-133 folded into the aggregation's store):
//   act    = relu(row_scale * acc + bias)                      -> ReLU mask bits and/or the activation itself
//   x_next = dropout_{seed}((1-alpha) * act + alpha * mix_src[row])   -> the next stage's input
// Only for VEC = 4 full tiles (d % 256 == 0): lane l owns columns 4l..4l+3 of its 256-wide tile, the mask
// word k of a (row, tile) holds the ballot of component k over the 64 lanes.
struct FusedEpi {
  const float* mix_src;   // [N, ld_mix] or null (no mix)
  int64_t ld_mix;
  float c_act, c_mix;     // (1 - alpha), alpha
  uint32_t thresh;        // dropout threshold (0 = keep everything)
  float keep_scale;       // 1 / (1 - p)
  uint64_t seed;
  const uint64_t* seed_dev;  // hipGraph mode: per-step seed part in device memory (added to `seed`), or null
  int64_t row0;           // global index of local row 0 (node-sharded runs draw the unsharded mask)
  unsigned long long* bits;  // [N][d/256][4] or null
  float* out_act;         // [N, ld_act] or null
  int64_t ld_act;
  float* out_next;        // [N, ld_next]
  int64_t ld_next;
  int d;
};

__device__ __forceinline__ void fused_store(const FusedEpi& fe, int64_t row, int c0, const float (&acc)[4], float scale,
                                            const float (&b)[4], const float (&rmix)[4]) {
  float a[4], x[4], m[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = fmaxf(scale_add(acc[i], scale, b[i]), 0.f);
  if (fe.thresh) keep4(fe.seed_dev ? fe.seed + *fe.seed_dev : fe.seed, ((fe.row0 + row) * fe.d + c0) >> 2, fe.thresh, fe.keep_scale, m);
  if (fe.bits) {
    // mask word k of (row, tile), bit l: the element (column 4 l + k) passes gradient to the pre-activation — ReLU positive AND kept
    // by the dropout.  The backward kernels that also regenerate the keep-mask are unaffected (masking twice is masking once);
    // cb_spmm_csr_masked_f32 needs nothing but these words.
    const int lane = lane_id();
    unsigned long long mine = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long w = __ballot(a[k] > 0.f && m[k] != 0.f);
      if (lane == k) mine = w;
    }
    if (lane < 4) fe.bits[(row * (fe.d >> 8) + (c0 >> 8)) * 4 + lane] = mine;
  }
  if (fe.out_act) store_stream<4>(fe.out_act + row * fe.ld_act + c0, a);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = fe.mix_src ? mix2(fe.c_act, a[i], fe.c_mix, rmix[i]) : a[i];
  if (fe.thresh) {      // kept as a statement of its own: the same rounding sequence as cb_axpby_f32 followed by cb_dropout_f32
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] *= m[i];
  }
  store_stream<4>(fe.out_next + row * fe.ld_next + c0, x);
}

template <int VEC>
__device__ __forceinline__ void write_row(float* __restrict__ out_row, const float (&acc)[VEC], float scale,
                                          const float (&b)[VEC], int relu) {
  float r[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float t = scale_add(acc[i], scale, b[i]);   // rst * norm + bias   (GCN.py:250,253)
    r[i] = relu ? fmaxf(t, 0.f) : t;
  }
  store_stream<VEC>(out_row, r);
}

// Walks the contiguous edge range of local rows [rlo, rhi) of this wavefront's row block.
// my_ptr: lane i holds rowptr[r0 + i] (i <= nr).  All control flow is wave-uniform.
// wave-uniform `craw` = column id as stored (GP == 2: bit 31 = hot flag)
template <int VEC, typename HT, int GP>
__device__ __forceinline__ void gather_pol(float (&v)[VEC], const HT* __restrict__ h_lane, int64_t ld_h, int craw) {
  if constexpr (GP == 0 || VEC != 4) {
    gather_in<VEC, HT>(v, h_lane + (int64_t)craw * ld_h);
  } else if constexpr (GP == 1) {
    if constexpr (sizeof(HT) == 4) gather_nt4(v, reinterpret_cast<const float*>(h_lane) + (int64_t)craw * ld_h);
    else gather_nt4_bf16(v, h_lane + (int64_t)craw * ld_h);
  } else {
    const HT* p = h_lane + (int64_t)(craw & kColMask) * ld_h;
    if (craw < 0) gather_in<VEC, HT>(v, p);
    else if constexpr (sizeof(HT) == 4) gather_nt4(v, reinterpret_cast<const float*>(p));
    else gather_nt4_bf16(v, p);
  }
}

template <int VEC, int U, bool FULL, bool FUSED, bool ACC, typename HT, int GP = 0, bool MASK = false>
__device__ __forceinline__ void stream_rows(int rlo, int rhi, int nr, int my_ptr, float my_scale, int r0, const int* __restrict__ col,
                                            const HT* __restrict__ h_lane, int64_t ld_h, float* __restrict__ out_lane,
                                            int64_t ld_out, bool active_in, int relu, const float (&bvec)[VEC], const FusedEpi& fe,
                                            int c0, const float* __restrict__ init_lane, int64_t ld_init, const Epilogue& ep) {
  static_assert(!MASK || (VEC == 4 && FULL && !FUSED && !ACC), "masked gather: d % 256 == 0, plain store");
  const int m_tiles = MASK ? (int)gridDim.y : 1, m_tile = MASK ? (int)blockIdx.y : 0;
  const bool active = FULL ? true : active_in;
  float ainit[VEC];                          // ACC: partial sums of local row `cur`, fetched one row ahead (read once: streaming)
  zero<VEC>(ainit);
  if constexpr (ACC) {
    if (active) gather_stream<VEC>(ainit, init_lane + (int64_t)(r0 + rlo) * ld_init);
  }
  float rmix[4] = {0.f, 0.f, 0.f, 0.f};   // FUSED: mix_src row of local row `cur`, fetched one row ahead
  if constexpr (FUSED) {
    if (fe.mix_src) {
      float t[VEC];
      gather_stream<VEC>(t, fe.mix_src + (int64_t)(r0 + rlo) * fe.ld_mix + c0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) rmix[i] = t[i];
    }
  }
  const int lane = lane_id();
  const int e_begin = bcast_lane(my_ptr, rlo);
  const int e_end = bcast_lane(my_ptr, rhi);
  int cur = rlo;
  int cur_end = bcast_lane(my_ptr, rlo + 1);
  float acc[VEC];
  zero<VEC>(acc);

  auto flush = [&]() {
    const float s = __int_as_float(bcast_lane(__float_as_int(my_scale), cur));  // row scale of local row `cur`
    if constexpr (ACC) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += ainit[i];
      if (active && cur + 1 < rhi) gather_stream<VEC>(ainit, init_lane + (int64_t)(r0 + cur + 1) * ld_init);
    }
    if constexpr (FUSED) {
      float a4[4], b4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a4[i] = acc[i % VEC]; b4[i] = bvec[i % VEC]; }
      fused_store(fe, (int64_t)(r0 + cur), c0, a4, s, b4, rmix);
      if (fe.mix_src && cur + 1 < nr) {
        float t[VEC];
        gather_stream<VEC>(t, fe.mix_src + (int64_t)(r0 + cur + 1) * fe.ld_mix + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) rmix[i] = t[i];
      }
    } else if (active) {
      write_row<VEC>(out_lane + (int64_t)(r0 + cur) * ld_out, acc, s, bvec, relu);
    }
    zero<VEC>(acc);
    ++cur;
    cur_end = bcast_lane(my_ptr, cur + 1);
  };

  for (int base = e_begin; base < e_end; base += kWave) {
    const int cnt = min(kWave, e_end - base);
    int my_col = 0;
    if (lane < cnt) my_col = __builtin_nontemporal_load(col + base + lane);
    int k = 0;
    for (; k + U <= cnt; k += U) {
      float v[U][VEC];
      SrcRowMask sm[MASK ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = bcast_lane(my_col, k + u);
        if (active) gather_pol<VEC, HT, GP>(v[u], h_lane, ld_h, c);
        else zero<VEC>(v[u]);
        if constexpr (MASK) sm[u] = load_src_mask(ep, GP == 2 ? (c & kColMask) : c, m_tiles, m_tile);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + k + u;
        while (e == cur_end) flush();
        if constexpr (MASK) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += sm[u].s * keep_if_bit(v[u][i], sm[u].w[i]);
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
        }
      }
    }
    for (; k < cnt; ++k) {
      const int c = bcast_lane(my_col, k);
      float v[VEC];
      if (active) gather_pol<VEC, HT, GP>(v, h_lane, ld_h, c);
      else zero<VEC>(v);
      const int e = base + k;
      while (e == cur_end) flush();
      if constexpr (MASK) {
        const SrcRowMask m1 = load_src_mask(ep, GP == 2 ? (c & kColMask) : c, m_tiles, m_tile);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += m1.s * keep_if_bit(v[i], m1.w[i]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += v[i];
      }
    }
  }
  while (cur < rhi) flush();  // last row + trailing empty rows
}

template <int VEC, int RPW, int U, bool FULL, bool FUSED, typename HT, bool ACC = false, int GP = 0, bool MASK = false>
__global__ void __launch_bounds__(256) k_spmm_rows(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const HT* __restrict__ h, int64_t ld_h, float* __restrict__ out,
                                                   int64_t ld_out, int n_rows, int d, Epilogue ep, int hub_T, FusedEpi fe) {
  static_assert(!FUSED || (VEC == 4 && FULL), "fused epilogue: d % 256 == 0, float4 lanes");
  static_assert(RPW < kWave, "row block must fit the lanes of one wavefront (+1 end pointer)");
  const int lane = lane_id();
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int r0 = wave * RPW;
  if (r0 >= n_rows) return;
  const int nr = min(RPW, n_rows - r0);
  const int c0 = (blockIdx.y * kWave + lane) * VEC;  // this lane's first column
  const bool active = c0 < d;

  int my_ptr = __builtin_nontemporal_load(rowptr + r0 + min(lane, nr));
  float my_scale = 1.f;  // lane i: row_scale[r0 + i], broadcast at flush time (no load on the flush path)
  if (ep.row_scale && lane < nr) my_scale = __builtin_nontemporal_load(ep.row_scale + r0 + lane);
  if constexpr (MASK) my_scale *= ep.out_coef;
  const int nxt = __shfl_down(my_ptr, 1);
  const unsigned long long hubmask = __ballot(lane < nr && (nxt - my_ptr) > hub_T);

  float bvec[VEC];
  zero<VEC>(bvec);
  if (ep.bias && active) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) bvec[i] = ep.bias[c0 + i];
  }
  const HT* h_lane = h + c0;
  float* out_lane = out + c0;
  const float* init_lane = ACC ? ep.acc_init + c0 : nullptr;

  if (hubmask == 0) {
    stream_rows<VEC, U, FULL, FUSED, ACC, HT, GP, MASK>(0, nr, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, active, ep.relu, bvec, fe, c0,
                                                    init_lane, ep.ld_init, ep);
  } else {
    int r = 0;
    while (r < nr) {  // maximal hub-free runs; hub rows are written by the hub kernels
      unsigned long long m = hubmask >> r;
      int nh = m ? r + (__ffsll((long long)m) - 1) : nr;
      if (nh > r)
        stream_rows<VEC, U, FULL, FUSED, ACC, HT, GP, MASK>(r, nh, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, active, ep.relu, bvec, fe,
                                                        c0, init_lane, ep.ld_init, ep);
      r = nh + 1;
    }
  }
}

// One wavefront per chunk of T edges of a hub row -> one partial row in `partial`.
template <int VEC, int U, typename HT, int GP = 0, bool MASK = false>
__global__ void __launch_bounds__(256) k_spmm_hub_chunks(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const HT* __restrict__ h, int64_t ld_h, int d, int hub_T,
                                                         int n_hubs, int n_chunks, const int* __restrict__ hub_rows,
                                                         const int* __restrict__ hub_chunk_ptr, float* __restrict__ partial,
                                                         int64_t ld_p, Epilogue ep) {
  const int m_tiles = MASK ? (int)gridDim.y : 1, m_tile = MASK ? (int)blockIdx.y : 0;
  const int lane = lane_id();
  const int chunk = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (chunk >= n_chunks) return;
  const int c0 = (blockIdx.y * kWave + lane) * VEC;
  const bool active = c0 < d;
  // hub index: last i with hub_chunk_ptr[i] <= chunk (wave-uniform binary search)
  int lo = 0, hi = n_hubs;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (hub_chunk_ptr[mid] <= chunk) lo = mid; else hi = mid;
  }
  const int row = hub_rows[lo];
  const int j = chunk - hub_chunk_ptr[lo];
  const int e_begin = rowptr[row] + j * hub_T;
  const int e_end = min(e_begin + hub_T, rowptr[row + 1]);
  const HT* h_lane = h + c0;
  float acc[VEC];
  zero<VEC>(acc);
  for (int base = e_begin; base < e_end; base += kWave) {
    const int cnt = min(kWave, e_end - base);
    int my_col = 0;
    if (lane < cnt) my_col = __builtin_nontemporal_load(col + base + lane);
    int k = 0;
    for (; k + U <= cnt; k += U) {
      float v[U][VEC];
      SrcRowMask sm[MASK ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = bcast_lane(my_col, k + u);
        if (active) gather_pol<VEC, HT, GP>(v[u], h_lane, ld_h, c);
        else zero<VEC>(v[u]);
        if constexpr (MASK) sm[u] = load_src_mask(ep, GP == 2 ? (c & kColMask) : c, m_tiles, m_tile);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (MASK) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += sm[u].s * keep_if_bit(v[u][i], sm[u].w[i]);
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
        }
      }
    }
    for (; k < cnt; ++k) {
      const int c = bcast_lane(my_col, k);
      float v[VEC];
      if (active) gather_pol<VEC, HT, GP>(v, h_lane, ld_h, c);
      else zero<VEC>(v);
      if constexpr (MASK) {
        const SrcRowMask m1 = load_src_mask(ep, GP == 2 ? (c & kColMask) : c, m_tiles, m_tile);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += m1.s * keep_if_bit(v[i], m1.w[i]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += v[i];
      }
    }
  }
  if (active) {
    float* p = partial + (int64_t)chunk * ld_p + c0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = acc[i];
  }
}

// One wavefront per hub row: partials summed in chunk order, then the epilogue.
template <int VEC, bool FUSED>
__global__ void __launch_bounds__(256) k_spmm_hub_finish(int d, int n_hubs, const int* __restrict__ hub_rows,
                                                         const int* __restrict__ hub_chunk_ptr,
                                                         const float* __restrict__ partial, int64_t ld_p,
                                                         float* __restrict__ out, int64_t ld_out, Epilogue ep, FusedEpi fe) {
  const int lane = lane_id();
  const int i = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= n_hubs) return;
  const int c0 = (blockIdx.y * kWave + lane) * VEC;
  if (c0 >= d) return;
  const int row = hub_rows[i];
  float acc[VEC];
  zero<VEC>(acc);
  for (int c = hub_chunk_ptr[i]; c < hub_chunk_ptr[i + 1]; ++c) {
    float v[VEC];
    gather<VEC>(v, partial + (int64_t)c * ld_p + c0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
  }
  if (ep.acc_init) {   // node-sharded path: partial sums of the interior-column pass
    float v[VEC];
    gather<VEC>(v, ep.acc_init + (int64_t)row * ep.ld_init + c0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
  }
  float bvec[VEC];
  zero<VEC>(bvec);
  if (ep.bias) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) bvec[k] = ep.bias[c0 + k];
  }
  float s = ep.row_scale ? ep.row_scale[row] : 1.f;
  if (ep.src_bits) s *= ep.out_coef;
  if constexpr (FUSED) {
    float a4[4], b4[4], rmix[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { a4[k] = acc[k % VEC]; b4[k] = bvec[k % VEC]; }
    if (fe.mix_src) {
      float t[VEC];
      gather_stream<VEC>(t, fe.mix_src + (int64_t)row * fe.ld_mix + c0);
#pragma unroll
      for (int k = 0; k < VEC; ++k) rmix[k] = t[k];
    }
    fused_store(fe, (int64_t)row, c0, a4, s, b4, rmix);
  } else {
    write_row<VEC>(out + (int64_t)row * ld_out + c0, acc, s, bvec, ep.relu);
  }
}

static inline int64_t partial_ld(int64_t d) { return (d + 3) / 4 * 4; }

// Gather policy of a launch: 2 when the caller hands over flagged column ids (col_flags), else 0; the measurement hook
// CB_SPMM_GATHER=1 turns plain-id launches into all-streaming ones (measured slower: profiles/r02_spmm_gather_policy.md).
static int gather_policy(int col_flags) {
  static const int env = getenv("CB_SPMM_GATHER") ? atoi(getenv("CB_SPMM_GATHER")) : 0;
  return col_flags ? 2 : (env == 1 ? 1 : 0);
}

template <int VEC, bool FUSED, int RPW, int U, typename HT>
static int launch_spmm_cfg(const int32_t* rowptr, const int32_t* col, int64_t N, const HT* h, int64_t ld_h, int64_t d,
                           Epilogue ep, float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks,
                           const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial, hipStream_t st, FusedEpi fe) {
  const int tile = kWave * VEC;
  const int ny = (int)((d + tile - 1) / tile);
  const int waves_per_block = 4;
  {
    int64_t n_waves = (N + RPW - 1) / RPW;
    dim3 grid((unsigned)((n_waves + waves_per_block - 1) / waves_per_block), ny);
#define CB_ROWS_LAUNCH_GP(FULL_, FUSED_, ACC_, GP_)                                                                                 \
  hipLaunchKernelGGL((k_spmm_rows<VEC, RPW, U, FULL_, FUSED_, HT, ACC_, GP_>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h, \
                     ld_h, out, ld_out, (int)N, (int)d, ep, hub_T, fe)
#define CB_ROWS_LAUNCH(FULL_, FUSED_, ACC_) CB_ROWS_LAUNCH_GP(FULL_, FUSED_, ACC_, 0)
    const bool acc = ep.acc_init != nullptr;
    constexpr bool kGP = VEC == 4 && RPW == 16 && U == 8;   // gather-policy variants: the d % 256 == 0 kernels (fp32 and bf16-stored rows)
    const int gp = kGP && d % tile == 0 ? (acc ? (ep.col_flags ? 2 : 0) : gather_policy(ep.col_flags)) : 0;
    CB_CHECK_ARG(!ep.col_flags || gp == 2, CB_E_INVALID, "flagged column ids are only understood by the d %% 256 == 0 kernels");
    if constexpr (FUSED) {
      if (acc) { if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, true, true, 2); else CB_ROWS_LAUNCH(true, true, true); } else CB_ROWS_LAUNCH(true, true, true); }
      else if constexpr (kGP) { if (gp == 1) CB_ROWS_LAUNCH_GP(true, true, false, 1); else if (gp == 2) CB_ROWS_LAUNCH_GP(true, true, false, 2); else CB_ROWS_LAUNCH(true, true, false); }
      else CB_ROWS_LAUNCH(true, true, false);
    } else if (d % tile == 0) {
      if (acc) { if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, false, true, 2); else CB_ROWS_LAUNCH(true, false, true); } else CB_ROWS_LAUNCH(true, false, true); }
      else if constexpr (kGP) { if (gp == 1) CB_ROWS_LAUNCH_GP(true, false, false, 1); else if (gp == 2) CB_ROWS_LAUNCH_GP(true, false, false, 2); else CB_ROWS_LAUNCH(true, false, false); }
      else CB_ROWS_LAUNCH(true, false, false);
    } else {
      if (acc) CB_ROWS_LAUNCH(false, false, true); else CB_ROWS_LAUNCH(false, false, false);
    }
#undef CB_ROWS_LAUNCH
#undef CB_ROWS_LAUNCH_GP
    CB_LAUNCH_CHECK();
  }
  if (n_hubs > 0) {
    const int64_t ld_p = partial_ld(d);
    dim3 grid((unsigned)((n_chunks + waves_per_block - 1) / waves_per_block), ny);
    constexpr bool kGPh = VEC == 4 && RPW == 16 && U == 8;
    const int gph = kGPh && d % tile == 0 ? gather_policy(ep.col_flags) : 0;
#define CB_HUB_LAUNCH(GP_)                                                                                                       \
  hipLaunchKernelGGL((k_spmm_hub_chunks<VEC, 8, HT, GP_>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h, ld_h, (int)d, \
                     hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, ld_p, ep)
    if constexpr (kGPh) { if (gph == 1) CB_HUB_LAUNCH(1); else if (gph == 2) CB_HUB_LAUNCH(2); else CB_HUB_LAUNCH(0); }
    else CB_HUB_LAUNCH(0);
#undef CB_HUB_LAUNCH
    CB_LAUNCH_CHECK();
    dim3 grid2((unsigned)((n_hubs + waves_per_block - 1) / waves_per_block), ny);
    hipLaunchKernelGGL((k_spmm_hub_finish<VEC, FUSED>), grid2, dim3(kWave * waves_per_block), 0, st, (int)d, n_hubs, hub_rows,
                       hub_chunk_ptr, partial, ld_p, out, ld_out, ep, fe);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// Tuning hook (measurement only): CB_SPMM_VARIANT = "<RPW>x<U>" selects another row-block / unroll shape of the
// d = 256 kernel; unset = the tuned default.
static int spmm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CB_SPMM_VARIANT");
    v = 0;
    if (e) {
      if (!strcmp(e, "8x8")) v = 1;
      else if (!strcmp(e, "32x8")) v = 2;
      else if (!strcmp(e, "16x4")) v = 3;
      else if (!strcmp(e, "16x16")) v = 4;
      else if (!strcmp(e, "8x4")) v = 5;
    }
  }
  return v;
}

template <int VEC, bool FUSED = false, typename HT = float>
static int launch_spmm(const int32_t* rowptr, const int32_t* col, int64_t N, const HT* h, int64_t ld_h, int64_t d,
                       Epilogue ep, float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks,
                       const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial, hipStream_t st,
                       FusedEpi fe = FusedEpi{}) {
#define CB_SPMM_ARGS rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st, fe
  if constexpr (VEC == 4 && !FUSED && sizeof(HT) == 4) {
    switch (spmm_variant()) {
      case 1: return launch_spmm_cfg<VEC, FUSED, 8, 8, HT>(CB_SPMM_ARGS);
      case 2: return launch_spmm_cfg<VEC, FUSED, 32, 8, HT>(CB_SPMM_ARGS);
      case 3: return launch_spmm_cfg<VEC, FUSED, 16, 4, HT>(CB_SPMM_ARGS);
      case 4: return launch_spmm_cfg<VEC, FUSED, 16, 16, HT>(CB_SPMM_ARGS);
      case 5: return launch_spmm_cfg<VEC, FUSED, 8, 4, HT>(CB_SPMM_ARGS);
      default: break;
    }
  }
  return launch_spmm_cfg<VEC, FUSED, 16, 8, HT>(CB_SPMM_ARGS);
#undef CB_SPMM_ARGS
}

// Masked-source launch (cb_spmm_csr_masked_f32): fp32 rows, d % 256 == 0, plain store.  MU = gathers in flight per wavefront
// (each carries 9 SGPRs of mask words + scale next to its 4 VGPRs).
template <int MU>
static int launch_spmm_masked(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, int64_t d, Epilogue ep,
                              float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks, const int32_t* hub_rows,
                              const int32_t* hub_chunk_ptr, float* partial, hipStream_t st) {
  constexpr int RPW = 16, waves_per_block = 4;
  const int ny = (int)(d / 256);
  const FusedEpi fe{};
  const int64_t n_waves = (N + RPW - 1) / RPW;
  const dim3 grid((unsigned)((n_waves + waves_per_block - 1) / waves_per_block), ny), blk(kWave * waves_per_block);
  if (ep.col_flags)
    hipLaunchKernelGGL((k_spmm_rows<4, RPW, MU, true, false, float, false, 2, true>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N,
                       (int)d, ep, hub_T, fe);
  else
    hipLaunchKernelGGL((k_spmm_rows<4, RPW, MU, true, false, float, false, 0, true>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N,
                       (int)d, ep, hub_T, fe);
  CB_LAUNCH_CHECK();
  if (n_hubs > 0) {
    const int64_t ld_p = partial_ld(d);
    const dim3 gridc((unsigned)((n_chunks + waves_per_block - 1) / waves_per_block), ny);
    if (ep.col_flags)
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, MU, float, 2, true>), gridc, blk, 0, st, rowptr, col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks,
                         hub_rows, hub_chunk_ptr, partial, ld_p, ep);
    else
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, MU, float, 0, true>), gridc, blk, 0, st, rowptr, col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks,
                         hub_rows, hub_chunk_ptr, partial, ld_p, ep);
    CB_LAUNCH_CHECK();
    const dim3 grid2((unsigned)((n_hubs + waves_per_block - 1) / waves_per_block), ny);
    hipLaunchKernelGGL((k_spmm_hub_finish<4, false>), grid2, blk, 0, st, (int)d, n_hubs, hub_rows, hub_chunk_ptr, partial, ld_p, out, ld_out,
                       ep, fe);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

}  // namespace cb

using namespace cb;

extern "C" size_t cb_spmm_workspace_bytes(int64_t n_chunks, int64_t d) {
  if (n_chunks <= 0 || d <= 0) return 0;
  return (size_t)n_chunks * (size_t)partial_ld(d) * sizeof(float);
}

static int spmm_plain_impl(const char* who, const int32_t* rowptr, const int32_t* col, int col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                           int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init,
                           float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                           const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d >= 0, CB_E_INVALID, "%s: negative size", who);
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "%s: size exceeds the int32 contract", who);
  if (N == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out && (E == 0 || col), CB_E_INVALID, "%s: null pointer", who);
  CB_CHECK_ARG(ld_h >= d && ld_out >= d && (!acc_init || ld_init >= d), CB_E_INVALID, "%s: leading dimension smaller than d", who);
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "%s: bad hub plan", who);
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)),
               CB_E_WORKSPACE, "%s: hub plan given but workspace missing/too small (%zu < %zu)", who, ws_bytes,
               cb_spmm_workspace_bytes(n_chunks, d));
  Epilogue ep{row_scale, bias, relu, acc_init, ld_init, col_flags};
  CB_CHECK_ARG(!col_flags || d % 256 == 0, CB_E_INVALID, "%s: flagged column ids need d %% 256 == 0", who);
  hipStream_t st = (hipStream_t)stream;
  if (n_hubs == 0) hub_T = INT32_MAX;  // no plan given (or no hub rows): every row is reduced whole by one wavefront
  const bool ini16 = !acc_init || (((uintptr_t)acc_init % 16 == 0) && ld_init % 4 == 0);
  const bool ini8 = !acc_init || (((uintptr_t)acc_init % 8 == 0) && ld_init % 2 == 0);
  const bool al16 = ((uintptr_t)h % 16 == 0) && ((uintptr_t)out % 16 == 0) && (ld_h % 4 == 0) && (ld_out % 4 == 0) && (d % 4 == 0) && ini16;
  const bool al8 = ((uintptr_t)h % 8 == 0) && ((uintptr_t)out % 8 == 0) && (ld_h % 2 == 0) && (ld_out % 2 == 0) && (d % 2 == 0) && ini8;
  float* partial = (float*)ws;
  static const bool small_off = getenv("CB_SPMM_NO_SMALL") != nullptr;   // measurement hook: one wavefront per gathered row at every width
  CB_CHECK_ARG(!col_flags || al16, CB_E_INVALID, "%s: flagged column ids need 16-byte aligned rows", who);
  if (!small_off && !acc_init && spmm_small_eligible(d, al16))
    return launch_spmm_small(rowptr, col, N, h, ld_h, d, row_scale, bias, relu, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows,
                             hub_chunk_ptr, partial, partial_ld(d), al16, st);
  if (al16 && d >= 256)
    return launch_spmm<4>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
  if (al8 && d >= 128)
    return launch_spmm<2>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
  return launch_spmm<1>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
}

extern "C" int cb_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                               int64_t ld_h, int64_t d, const float* row_scale, const float* bias, int relu, float* out, int64_t ld_out,
                               int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                               const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  return spmm_plain_impl("cb_spmm_csr_f32", rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, relu, nullptr, 0, out, ld_out, hub_T, n_hubs,
                         n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream);
}

// out = act(row_scale * (acc_init + sum over this CSR's columns) + bias): the second (halo-column) pass of the node-sharded
// aggregation; acc_init holds the raw sums of the interior-column pass (dist.py).  acc_init may alias out.
extern "C" int cb_spmm_csr_acc_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                   int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init,
                                   int64_t ld_init, float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                   const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0 || d == 0, CB_E_INVALID, "cb_spmm_csr_acc_f32: acc_init is null");
  return spmm_plain_impl("cb_spmm_csr_acc_f32", rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, relu, acc_init, ld_init, out, ld_out, hub_T,
                         n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream);
}

// out[v] = out_coef * sum_{u in row v} src_scale[u] * (src_bits[u] ? h[u] : 0): the reverse aggregation of the fused trunk's
// backward with the layer-below's store backward (dropout keep & ReLU mask bits of the forward store, c_act / (1 - p), degree norm
// of the source row) applied to every gathered row — what cb_trunk_layer_bwd_f32 followed by cb_spmm_csr_f32 computes, without
// the [N, d] intermediate (8 bytes / element less traffic per layer; + 36 bytes per edge of scalar-cache reads).
extern "C" int cb_spmm_csr_masked_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                      int64_t ld_h, int64_t d, const uint64_t* src_bits, const float* src_scale, float out_coef,
                                      float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                      const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d > 0 && d % 256 == 0, CB_E_INVALID, "cb_spmm_csr_masked_f32: d must be a positive multiple of 256");
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "cb_spmm_csr_masked_f32: size exceeds the int32 contract");
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out && src_bits && src_scale && (E == 0 || col), CB_E_INVALID, "cb_spmm_csr_masked_f32: null pointer");
  CB_CHECK_ARG(((uintptr_t)h % 16 == 0) && ((uintptr_t)out % 16 == 0) && ld_h % 4 == 0 && ld_out % 4 == 0 && ld_h >= d && ld_out >= d &&
                   ((uintptr_t)src_bits % 8 == 0) && ((uintptr_t)src_scale % 4 == 0),
               CB_E_INVALID, "cb_spmm_csr_masked_f32: 16-byte aligned rows of at least d floats required");
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "cb_spmm_csr_masked_f32: bad hub plan");
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)), CB_E_WORKSPACE,
               "cb_spmm_csr_masked_f32: hub plan given but workspace missing/too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{nullptr, nullptr, 0, nullptr, 0, col_flags, (const unsigned long long*)src_bits, src_scale, out_coef};
  static const int mu = getenv("CB_SPMM_MASK_U") ? atoi(getenv("CB_SPMM_MASK_U")) : 4;     // measurement hook
  if (mu == 4)
    return launch_spmm_masked<4>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                 (hipStream_t)stream);
  return launch_spmm_masked<8>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                               (hipStream_t)stream);
}

static int spmm_fused_impl(int h_bf16, const float* acc_init, int64_t ld_init, int col_flags, const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, const void* h, int64_t ld_h,
                                     int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                                     float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits,
                                     float* out_act, int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_T,
                                     int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                                     void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d > 0 && d % 256 == 0, CB_E_INVALID, "cb_spmm_csr_fused_f32: d must be a positive multiple of 256");
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "cb_spmm_csr_fused_f32: size exceeds the int32 contract");
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out_next && (E == 0 || col), CB_E_INVALID, "cb_spmm_csr_fused_f32: null pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_spmm_csr_fused_f32: dropout p out of range");
  const bool al = ((uintptr_t)h % (h_bf16 ? 8 : 16) == 0) && ((uintptr_t)out_next % 16 == 0) && ld_h % 4 == 0 && ld_next % 4 == 0 &&
                  (!mix_src || ((uintptr_t)mix_src % 16 == 0 && ld_mix % 4 == 0)) &&
                  (!out_act || ((uintptr_t)out_act % 16 == 0 && ld_act % 4 == 0));
  CB_CHECK_ARG(al && ld_h >= d && ld_next >= d, CB_E_INVALID, "cb_spmm_csr_fused_f32: 16-byte aligned rows required");
  CB_CHECK_ARG(!acc_init || ((uintptr_t)acc_init % 16 == 0 && ld_init % 4 == 0 && ld_init >= d), CB_E_INVALID,
               "cb_spmm_csr_fused_acc_f32: acc_init must be 16-byte aligned rows of at least d floats");
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "cb_spmm_csr_fused_f32: bad hub plan");
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)),
               CB_E_WORKSPACE, "cb_spmm_csr_fused_f32: hub plan given but workspace missing/too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, bias, 1, acc_init, ld_init, col_flags};
  FusedEpi fe{};
  fe.mix_src = mix_src; fe.ld_mix = ld_mix; fe.c_act = c_act; fe.c_mix = c_mix;
  fe.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fe.keep_scale = 1.f / (1.f - drop_p);
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits;
  fe.out_act = out_act; fe.ld_act = ld_act; fe.out_next = out_next; fe.ld_next = ld_next; fe.d = (int)d;
  if (h_bf16)
    return launch_spmm<4, true, bf16_t>(rowptr, col, N, (const bf16_t*)h, ld_h, d, ep, out_next, ld_next, hub_T, n_hubs, n_chunks,
                                        hub_rows, hub_chunk_ptr, (float*)ws, (hipStream_t)stream, fe);
  return launch_spmm<4, true, float>(rowptr, col, N, (const float*)h, ld_h, d, ep, out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows,
                                     hub_chunk_ptr, (float*)ws, (hipStream_t)stream, fe);
}

#define CB_FUSED_PARAMS                                                                                                          \
  const int32_t *rowptr, const int32_t *col, int64_t N, int64_t E, const void *h, int64_t ld_h, int64_t d, const float *row_scale, \
      const float *bias, const float *mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,               \
      const uint64_t *seed_dev, int64_t row0,                                                                                      \
      uint64_t *relu_bits, float *out_act, int64_t ld_act, float *out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs,        \
      int32_t n_chunks, const int32_t *hub_rows, const int32_t *hub_chunk_ptr, void *ws, size_t ws_bytes, void *stream
#define CB_FUSED_ARGS                                                                                                            \
  rowptr, col, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed, seed_dev, row0, relu_bits, out_act, ld_act,  \
      out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream

extern "C" int cb_spmm_csr_fused_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                     int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                                     float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                     uint64_t* relu_bits, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                     int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                     const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  return spmm_fused_impl(0, nullptr, 0, col_flags, CB_FUSED_ARGS);
}

// Fused store of the residual trunk on top of the interior-column partial sums (second pass of the node-sharded aggregation)
extern "C" int cb_spmm_csr_fused_acc_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags,
                                         int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias,
                                         const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                         const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, float* out_act, int64_t ld_act,
                                         float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                         const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0, CB_E_INVALID, "cb_spmm_csr_fused_acc_f32: acc_init is null");
  return spmm_fused_impl(0, acc_init, ld_init, col_flags, CB_FUSED_ARGS);
}

extern "C" int cb_spmm_csr_fused_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h,
                                          int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                                          int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                          const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                          int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                          const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  return spmm_fused_impl(1, nullptr, 0, col_flags, CB_FUSED_ARGS);
}

// bf16-stored source rows, fp32 accumulation and output (build extension: BASELINE config 2; also the halo pass of the node-sharded
// aggregation when the halo rows crossed the links as bf16: acc_init = the interior-column sums, dist.py)
static int spmm_bf16_impl(const char* who, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h,
                          int64_t ld_h, int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init,
                          float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                          const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d >= 0, CB_E_INVALID, "%s: negative size", who);
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "%s: size exceeds the int32 contract", who);
  if (N == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out && (E == 0 || col), CB_E_INVALID, "%s: null pointer", who);
  CB_CHECK_ARG(ld_h >= d && ld_out >= d && (!acc_init || ld_init >= d), CB_E_INVALID, "%s: leading dimension smaller than d", who);
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "%s: bad hub plan", who);
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)),
               CB_E_WORKSPACE, "%s: hub plan given but workspace missing/too small", who);
  const bool ini16 = !acc_init || (((uintptr_t)acc_init % 16 == 0) && ld_init % 4 == 0);
  const bool ini8 = !acc_init || (((uintptr_t)acc_init % 8 == 0) && ld_init % 2 == 0);
  const bool al8 = ((uintptr_t)h % 8 == 0) && ((uintptr_t)out % 16 == 0) && (ld_h % 4 == 0) && (ld_out % 4 == 0) && (d % 4 == 0) && ini16;
  CB_CHECK_ARG(!col_flags || (al8 && d % 256 == 0), CB_E_INVALID, "%s: flagged column ids need d %% 256 == 0 and 8-byte aligned rows", who);
  Epilogue ep{row_scale, bias, relu, acc_init, ld_init, col_flags};
  hipStream_t st = (hipStream_t)stream;
  if (n_hubs == 0) hub_T = INT32_MAX;
  const bf16_t* hb = (const bf16_t*)h;
  const bool al4 = ((uintptr_t)h % 4 == 0) && ((uintptr_t)out % 8 == 0) && (ld_h % 2 == 0) && (ld_out % 2 == 0) && (d % 2 == 0) && ini8;
  float* partial = (float*)ws;
  if (al8 && d >= 256)
    return launch_spmm<4, false, bf16_t>(rowptr, col, N, hb, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
  if (al4 && d >= 128)
    return launch_spmm<2, false, bf16_t>(rowptr, col, N, hb, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
  return launch_spmm<1, false, bf16_t>(rowptr, col, N, hb, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
}

extern "C" int cb_spmm_csr_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h, int64_t ld_h,
                                    int64_t d, const float* row_scale, const float* bias, int relu, float* out, int64_t ld_out,
                                    int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                    const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  return spmm_bf16_impl("cb_spmm_csr_bf16_f32", rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, relu, nullptr, 0, out, ld_out, hub_T,
                        n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream);
}

// cb_spmm_csr_acc_f32 over bf16-stored source rows: the halo-column pass when the halo rows travelled as bf16 (the wire buffer
// is read as it arrived, no widening pass).  acc_init may alias out.
extern "C" int cb_spmm_csr_acc_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h,
                                        int64_t ld_h, int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init,
                                        int64_t ld_init, float* out, int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                        const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0 || d == 0, CB_E_INVALID, "cb_spmm_csr_acc_bf16_f32: acc_init is null");
  return spmm_bf16_impl("cb_spmm_csr_acc_bf16_f32", rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, bias, relu, acc_init, ld_init, out, ld_out,
                        hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream);
}

// cb_spmm_csr_fused_acc_f32 over bf16-stored source rows (fused trunk store on top of the interior sums, bf16 wire buffer)
extern "C" int cb_spmm_csr_fused_acc_bf16_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags,
                                              int64_t N, int64_t E, const uint16_t* h, int64_t ld_h, int64_t d, const float* row_scale,
                                              const float* bias, const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p,
                                              uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, float* out_act,
                                              int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                              const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0, CB_E_INVALID, "cb_spmm_csr_fused_acc_bf16_f32: acc_init is null");
  return spmm_fused_impl(1, acc_init, ld_init, col_flags, CB_FUSED_ARGS);
}
