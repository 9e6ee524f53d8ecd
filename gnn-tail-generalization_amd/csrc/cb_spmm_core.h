// Device side of the CSR sum-aggregation (design notes: cb_spmm.hip): vector helpers, the epilogues, the edge-stream walk of a
// wavefront's row block (stream_rows), the row / hub kernels.  Shared by cb_spmm.hip (the aggregation entry points) and
// cb_agg_gemm.hip (aggregation + next dense transform in one kernel).
#pragma once
#include "cb_common.h"
#include "cb_philox.h"

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
  // label-propagation step (outcome_correlation.py:137-143 with alpha_term, post_step = clamp(0, 1)), narrow rows (VEC < 4 kernels) only:
  //   out[v] = lp_post[v] * clamp(row_scale[v] * acc + lp_c_mix * lp_mix[v], 0, 1)        (lp_post = null: 1)
  // i.e. alpha * D^-1/2 A (.) + (1 - alpha) * y0, clamped, and already scaled by D^-1/2 for the next step's gather
  const float* lp_mix;      // [N, ld_lp] or null (null: plain epilogue)
  int64_t ld_lp;
  float lp_c_mix;
  const float* lp_post;     // [N] or null
  // ACC kernels, raw in-place accumulation only (out == acc_init, no scale / bias / ReLU: the intermediate halo passes of the node-sharded
  // aggregation): a row without edges in THIS CSR keeps its running sums as they are — neither read nor written (a halo slice touches only a
  // fraction of the rows; the passes are otherwise bound by reading and re-writing all of them)
  int acc_skip_empty;
  // CS kernels (d % 256 == 0, fp32 rows, plain store): out[v] = row_scale[v] * sum_u col_scale[u] * h[u] — the factor of a SOURCE row applied
  // as the row is gathered (the row-sparse backward's A (a * X) on the loss rows: no scaled copy of X, trunk.py)
  const float* col_scale;   // [n_cols] or null
};

template <int VEC>
__device__ __forceinline__ void write_row_lp(float* __restrict__ out_row, const float (&acc)[VEC], float scale, const Epilogue& ep, int64_t row,
                                             int c0, bool (&on)[VEC]) {
  const float post = ep.lp_post ? ep.lp_post[row] : 1.f;
  float r[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (!on[i]) continue;
    float t = acc[i] * scale;
    t = t + ep.lp_c_mix * ep.lp_mix[row * ep.ld_lp + c0 + i];
    t = fminf(fmaxf(t, 0.f), 1.f);
    r[i] = t * post;
    __builtin_nontemporal_store(r[i], out_row + i);
  }
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
  int bits_relu_only;     // mask words hold (act > 0) alone, not (act > 0 AND kept by this store's dropout): the 'Residual' trunk, whose backward
                          // also sends the NEXT layer's mix gradient through this ReLU (under another dropout mask)
  float* out_act;         // [N, ld_act] or null
  int64_t ld_act;
  float* out_next;        // [N, ld_next]
  int64_t ld_next;
  int d;
  int skip_next;          // cb_agg_gemm.hip, forwards without a backward: the finished row only goes to the on-chip tile, not to out_next
  int bwd;                // BACKWARD of a store applied to the row this (reverse) aggregation just summed (cb_spmm_csr_store_bwd_f32): g = scale * acc goes to
                          // out_act (the raw gradient w.r.t. the stored activation: the input stage's mix operand), and
                          // out_next = bwd_rowscale[row] * c_act * keep(seed, row) * g where the mask word `bits` (READ here) has the element's bit set, else 0 —
                          // cb_trunk_layer_bwd_f32's pass without its read of g
  const float* bwd_rowscale;
  const int32_t* row_ids; // or null.  The CSR's rows are a SUBSET of the node rows (row r = node row_ids[r]; rows-only forward, trunk.py): mix_src, the mask
                          // words and the dropout mask are taken at the node row, out_act / out_next (and row_scale, rowptr) at the compact row
  // MIXB kernels (cb_spmm_csr_store_bwd_mix_f32: bwd, all node rows): the gradients that reach X0 through the residual mixes are FOLDED into one stream —
  //   out_act = mx_c * ( keep(seed, row) * g  +  sum_q keep(mx_seed[q], row) * mx_g[q][mx_pos[q][row]] )        (a row with mx_pos < 0 is absent: zero)
  // instead of the raw g: what the input stage adds to the gradient of dropout(X0) (the compact operands mx_g are the mix gradients of the layers above,
  // each under its own store's dropout mask), so that the input stage reads ONE matrix instead of n + 1 — and cs_partial ([blocks][d]) receives the
  // column sums of out_next / bwd_rowscale per block: the bias gradient of the store whose backward this epilogue applies
  int mx_n;
  const float* mx_g[2];
  const int* mx_pos[2];
  uint64_t mx_seed[2];
  float mx_c;
  float* cs_partial;
  int cs_block0;          // (hub-finish launch: its blocks' partial rows follow the row kernel's)
};

// The MIXB form of fused_store's bwd branch (see FusedEpi).  xm[q] / xp[q]: operand q's row of this node row and its position (< 0: absent), fetched one
// row ahead by the caller; cs: the wavefront's running column sums of c_act * keep * g under the mask word.
__device__ __forceinline__ void fused_store_bwd_mix(const FusedEpi& fe, int64_t row, int c0, const float (&acc)[4], float scale, float (&x)[4],
                                                    const float (&xm)[2][4], const int (&xp)[2], float (&cs)[4]) {
  const uint64_t sd = fe.seed_dev ? *fe.seed_dev : 0ull;
  const int64_t quad = ((fe.row0 + row) * fe.d + c0) >> 2;
  float m[4] = {1.f, 1.f, 1.f, 1.f}, mm[4] = {0.f, 0.f, 0.f, 0.f};
  if (fe.thresh) keep4(fe.seed + sd, quad, fe.thresh, fe.keep_scale, m);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q < fe.mx_n && xp[q] >= 0) {      // (wave-uniform)
      float mq[4] = {1.f, 1.f, 1.f, 1.f};
      if (fe.thresh) keep4(fe.mx_seed[q] + sd, quad, fe.thresh, fe.keep_scale, mq);
#pragma unroll
      for (int i = 0; i < 4; ++i) mm[i] += fe.mx_c * (xm[q][i] * mq[i]);
    }
  }
  const unsigned long long* bw = fe.bits + (row * (fe.d >> 8) + (c0 >> 8)) * 4;
  const float rs = fe.bwd_rowscale ? fe.bwd_rowscale[row] : 1.f;
  const int lane = lane_id();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float gm = scale_add(acc[i], scale, 0.f) * m[i];
    mm[i] += fe.mx_c * gm;
    const float t = ((bw[i] >> lane) & 1ull) ? fe.c_act * gm : 0.f;
    cs[i] += t;
    x[i] = t * rs;                       // (cb_trunk_layer_bwd_f32's expressions, in its order)
  }
  store_stream<4>(fe.out_act + row * fe.ld_act + c0, mm);
  store_stream<4>(fe.out_next + row * fe.ld_next + c0, x);
}

// x: the values stored to out_next (also handed to the caller: cb_agg_gemm.hip keeps the finished row on chip)
// grow: the node row of `row` (== row unless fe.row_ids)
__device__ __forceinline__ void fused_store(const FusedEpi& fe, int64_t row, int c0, const float (&acc)[4], float scale,
                                            const float (&b)[4], const float (&rmix)[4], float (&x)[4], int64_t grow) {
  float a[4], m[4] = {1.f, 1.f, 1.f, 1.f};
  if (fe.bwd) {
    float g4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g4[i] = scale_add(acc[i], scale, b[i]);      // (b = 0: what the plain store writes)
    if (fe.out_act) store_stream<4>(fe.out_act + row * fe.ld_act + c0, g4);
    if (fe.thresh) keep4(fe.seed_dev ? fe.seed + *fe.seed_dev : fe.seed, ((fe.row0 + grow) * fe.d + c0) >> 2, fe.thresh, fe.keep_scale, m);
    const unsigned long long* bw = fe.bits + (grow * (fe.d >> 8) + (c0 >> 8)) * 4;
    const float rs = fe.bwd_rowscale ? fe.bwd_rowscale[grow] : 1.f;
    const int lane = lane_id();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gm = g4[i] * m[i];
      x[i] = ((bw[i] >> lane) & 1ull) ? (fe.c_act * gm) * rs : 0.f;      // (cb_trunk_layer_bwd_f32's expressions, in its order)
    }
    store_stream<4>(fe.out_next + row * fe.ld_next + c0, x);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = fmaxf(scale_add(acc[i], scale, b[i]), 0.f);
  if (fe.thresh) keep4(fe.seed_dev ? fe.seed + *fe.seed_dev : fe.seed, ((fe.row0 + grow) * fe.d + c0) >> 2, fe.thresh, fe.keep_scale, m);
  if (fe.bits) {
    // mask word k of (row, tile), bit l: the element (column 4 l + k) passes gradient to the pre-activation — ReLU positive AND kept
    // by the dropout.  The backward kernels that also regenerate the keep-mask are unaffected (masking twice is masking once).
    const int lane = lane_id();
    unsigned long long mine = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long w = __ballot(a[k] > 0.f && (fe.bits_relu_only || m[k] != 0.f));
      if (lane == k) mine = w;
    }
    if (lane < 4) fe.bits[(grow * (fe.d >> 8) + (c0 >> 8)) * 4 + lane] = mine;
  }
  if (fe.out_act) store_stream<4>(fe.out_act + row * fe.ld_act + c0, a);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = fe.mix_src ? mix2(fe.c_act, a[i], fe.c_mix, rmix[i]) : a[i];
  if (fe.thresh) {      // kept as a statement of its own: the same rounding sequence as cb_axpby_f32 followed by cb_dropout_f32
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] *= m[i];
  }
  if (!fe.skip_next) store_stream<4>(fe.out_next + row * fe.ld_next + c0, x);
}

template <int VEC>
__device__ __forceinline__ void write_row(float* __restrict__ out_row, const float (&acc)[VEC], float scale,
                                          const float (&b)[VEC], int relu, float (&r)[VEC]) {
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

// TLD > 0 (cb_agg_gemm.hip): every finished row is also written to an LDS tile — tile_lane = this lane's 4 columns of the
// wavefront's local row 0, TLD floats per tile row.
// P65 (64-row blocks, cb_agg_gemm.hip): lane i holds rowptr[r0 + i] for i < 64 and ptr_hi = rowptr[r0 + 64].
template <int VEC, int U, bool FULL, bool FUSED, bool ACC, typename HT, int GP = 0, int TLD = 0, bool P65 = false, bool CS = false, bool MIXB = false>
__device__ __forceinline__ void stream_rows(int rlo, int rhi, int nr, int my_ptr_v, float my_scale, int r0, const int* __restrict__ col,
                                            const HT* __restrict__ h_lane, int64_t ld_h, float* __restrict__ out_lane,
                                            int64_t ld_out, bool active_in, int relu, const float (&bvec)[VEC], const FusedEpi& fe,
                                            int c0, const float* __restrict__ init_lane, int64_t ld_init, const Epilogue& ep,
                                            float* tile_lane = nullptr, int ptr_hi = 0, int my_gid_v = 0, int my_xp0 = 0, int my_xp1 = 0,
                                            float* cs_acc = nullptr) {
  // my_gid_v (FUSED with fe.row_ids): lane i holds the node row of local row i
  // MIXB (FusedEpi): my_xp0 / my_xp1: lane i holds the position of local row i in the compact operands; cs_acc: the wavefront's 4 running column sums
  static_assert(!MIXB || (FUSED && VEC == 4 && FULL && !ACC && TLD == 0 && !P65), "folded mix gradients: the fused d % 256 == 0 kernel on all node rows");
  static_assert(TLD == 0 || (VEC == 4 && FULL), "on-chip row tile: d == 256, float4 lanes");
  static_assert(!CS || (!P65 && !FUSED && !ACC && sizeof(HT) == 4), "source-row factor: plain fp32 aggregation only");
  struct PtrAt {      // rowptr of local row i (wave-uniform i)
    int v, hi;
    __device__ __forceinline__ int operator()(int i) const {
      if constexpr (P65) return i >= kWave ? hi : bcast_lane(v, i);
      else return bcast_lane(v, i);
    }
  };
  const PtrAt my_ptr_at{my_ptr_v, ptr_hi};
  const bool active = FULL ? true : active_in;
  float ainit[VEC];                          // ACC: partial sums of local row `cur`, fetched one row ahead (read once: streaming)
  zero<VEC>(ainit);
  const bool skip_empty = ACC && !FUSED && ep.acc_skip_empty;
  if constexpr (ACC) {
    if (active && !(skip_empty && my_ptr_at(rlo) == my_ptr_at(rlo + 1))) gather_stream<VEC>(ainit, init_lane + (int64_t)(r0 + rlo) * ld_init);
  }
  float rmix[4] = {0.f, 0.f, 0.f, 0.f};   // FUSED: mix_src row of local row `cur`, fetched one row ahead
  auto gid_of = [&](int i) -> int64_t { return fe.row_ids ? (int64_t)bcast_lane(my_gid_v, i) : (int64_t)(r0 + i); };
  if constexpr (FUSED) {
    if (fe.mix_src) {
      float t[VEC];
      gather_stream<VEC>(t, fe.mix_src + gid_of(rlo) * fe.ld_mix + c0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) rmix[i] = t[i];
    }
  }
  float xm[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // MIXB: the compact operands' rows of local row `cur`, fetched one row ahead
  int xp[2] = {-1, -1};
  auto fetch_mix = [&](int i) {
    if constexpr (MIXB) {
      xp[0] = fe.mx_n > 0 ? bcast_lane(my_xp0, i) : -1;
      xp[1] = fe.mx_n > 1 ? bcast_lane(my_xp1, i) : -1;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (xp[q] >= 0) gather_stream<4>(xm[q], fe.mx_g[q] + (int64_t)xp[q] * fe.d + c0);
    }
  };
  if constexpr (MIXB) fetch_mix(rlo);
  const int lane = lane_id();
  const int e_begin = my_ptr_at(rlo);
  const int e_end = my_ptr_at(rhi);
  int cur = rlo;
  int cur_end = my_ptr_at(rlo + 1);
  float acc[VEC];
  zero<VEC>(acc);

  int cur_begin = e_begin;                   // first edge of local row `cur`
  auto flush = [&]() {
    const float s = __int_as_float(bcast_lane(__float_as_int(my_scale), cur));  // row scale of local row `cur`
    if constexpr (ACC) {
      const int nxt_end = my_ptr_at(cur + 2);      // (clamped beyond the block's last row: lanes past nr hold the block's end pointer)
      const bool fetch_next = active && cur + 1 < rhi && !(skip_empty && cur_end == nxt_end);
      if (skip_empty && cur_begin == cur_end) {      // raw in-place pass, row without edges here: its running sums stay untouched
        if (fetch_next) gather_stream<VEC>(ainit, init_lane + (int64_t)(r0 + cur + 1) * ld_init);
        ++cur;
        cur_begin = cur_end;
        cur_end = nxt_end;
        return;
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += ainit[i];
      if (fetch_next) gather_stream<VEC>(ainit, init_lane + (int64_t)(r0 + cur + 1) * ld_init);
    }
    if constexpr (FUSED) {
      float a4[4], b4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a4[i] = acc[i % VEC]; b4[i] = bvec[i % VEC]; }
      float x4[4];
      if constexpr (MIXB) {
        float (&csr)[4] = *reinterpret_cast<float (*)[4]>(cs_acc);
        fused_store_bwd_mix(fe, (int64_t)(r0 + cur), c0, a4, s, x4, xm, xp, csr);
        if (cur + 1 < rhi) fetch_mix(cur + 1);
      } else {
        fused_store(fe, (int64_t)(r0 + cur), c0, a4, s, b4, rmix, x4, gid_of(cur));
      }
      if constexpr (TLD > 0) *reinterpret_cast<float4*>(tile_lane + cur * TLD) = make_float4(x4[0], x4[1], x4[2], x4[3]);
      if (fe.mix_src && cur + 1 < nr) {
        float t[VEC];
        gather_stream<VEC>(t, fe.mix_src + gid_of(cur + 1) * fe.ld_mix + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) rmix[i] = t[i];
      }
    } else if (active) {
      bool lp_done = false;
      if constexpr (VEC < 4 && !ACC && TLD == 0) {
        if (ep.lp_mix) {
          bool on[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) on[i] = true;
          write_row_lp<VEC>(out_lane + (int64_t)(r0 + cur) * ld_out, acc, s, ep, (int64_t)(r0 + cur), c0, on);
          lp_done = true;
        }
      }
      if (!lp_done) {
        float rv[VEC];
        write_row<VEC>(out_lane + (int64_t)(r0 + cur) * ld_out, acc, s, bvec, relu, rv);
        if constexpr (TLD > 0) *reinterpret_cast<float4*>(tile_lane + cur * TLD) = make_float4(rv[0], rv[1], rv[2], rv[3]);
      }
    }
    zero<VEC>(acc);
    ++cur;
    cur_begin = cur_end;
    cur_end = my_ptr_at(cur + 1);
  };

  // P65 kernels run ONE wavefront per SIMD (cb_agg_gemm.hip): no neighbour hides a dependent load, so the next window's column ids are
  // requested before this window's gathers and the remainder of a window goes out as one predicated batch instead of one load at a time
  int col_next = 0;
  if constexpr (P65) {
    if (e_begin + lane < e_end) col_next = __builtin_nontemporal_load(col + e_begin + lane);
  }
  for (int base = e_begin; base < e_end; base += kWave) {
    const int cnt = min(kWave, e_end - base);
    int my_col = 0;
    if constexpr (P65) {
      my_col = col_next;
      if (base + kWave + lane < e_end) col_next = __builtin_nontemporal_load(col + base + kWave + lane);
    } else {
      if (lane < cnt) my_col = __builtin_nontemporal_load(col + base + lane);
    }
    float my_cs = 0.f;                         // CS: lane i holds the factor of the window's i-th source row
    if constexpr (CS) {
      if (lane < cnt) my_cs = ep.col_scale[my_col & kColMask];
    }
    int k = 0;
    for (; k + U <= cnt; k += U) {
      float v[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = bcast_lane(my_col, k + u);
        if (active) gather_pol<VEC, HT, GP>(v[u], h_lane, ld_h, c);
        else zero<VEC>(v[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + k + u;
        while (e == cur_end) flush();
        if constexpr (CS) {
          const float cs = __int_as_float(bcast_lane(__float_as_int(my_cs), k + u));
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += cs * v[u][i];
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
        }
      }
    }
    if constexpr (P65) {
      if (k < cnt) {
        const int nb = cnt - k;      // 1 .. U-1 edges left in this window
        float v[U][VEC];
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
          if (u < nb) gather_pol<VEC, HT, GP>(v[u], h_lane, ld_h, bcast_lane(my_col, k + u));
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
          if (u < nb) {
            const int e = base + k + u;
            while (e == cur_end) flush();
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
          }
        }
        k = cnt;
      }
    }
    for (; k < cnt; ++k) {
      const int c = bcast_lane(my_col, k);
      float v[VEC];
      if (active) gather_pol<VEC, HT, GP>(v, h_lane, ld_h, c);
      else zero<VEC>(v);
      const int e = base + k;
      while (e == cur_end) flush();
      if constexpr (CS) {
        const float cs = __int_as_float(bcast_lane(__float_as_int(my_cs), k));
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += cs * v[i];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += v[i];
      }
    }
  }
  while (cur < rhi) flush();  // last row + trailing empty rows
}

// The wavefronts' running column sums (MIXB: FusedEpi::cs_partial) -> one partial row per block: fixed order, no atomics.  All threads of the block call it.
__device__ __forceinline__ void block_colsum_store(const float (&cs)[4], float* __restrict__ partial_row, int c0, bool live) {
  __shared__ float s_cs[4][256 + 4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) s_cs[w][lane * 4 + i] = live ? cs[i] : 0.f;
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      partial_row[c0 + i] = ((s_cs[0][lane * 4 + i] + s_cs[1][lane * 4 + i]) + s_cs[2][lane * 4 + i]) + s_cs[3][lane * 4 + i];
  }
}

template <int VEC, int RPW, int U, bool FULL, bool FUSED, typename HT, bool ACC = false, int GP = 0, bool CS = false, bool MIXB = false>
__global__ void __launch_bounds__(256) k_spmm_rows(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const HT* __restrict__ h, int64_t ld_h, float* __restrict__ out,
                                                   int64_t ld_out, int n_rows, int d, Epilogue ep, int hub_T, FusedEpi fe) {
  static_assert(!FUSED || (VEC == 4 && FULL), "fused epilogue: d % 256 == 0, float4 lanes");
  static_assert(RPW < kWave, "row block must fit the lanes of one wavefront (+1 end pointer)");
  const int lane = lane_id();
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int r0 = wave * RPW;
  float cs_acc[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MIXB) {      // (no early exit: every wavefront of the block takes part in the column sums' hand-over)
    if (r0 >= n_rows) {
      if (fe.cs_partial) block_colsum_store(cs_acc, fe.cs_partial + (int64_t)(fe.cs_block0 + blockIdx.x) * d + blockIdx.y * 256, lane * 4, false);
      return;
    }
  } else {
    if (r0 >= n_rows) return;
  }
  const int nr = min(RPW, n_rows - r0);
  const int c0 = (blockIdx.y * kWave + lane) * VEC;  // this lane's first column
  const bool active = c0 < d;

  int my_ptr = __builtin_nontemporal_load(rowptr + r0 + min(lane, nr));
  float my_scale = 1.f;  // lane i: row_scale[r0 + i], broadcast at flush time (no load on the flush path)
  if (ep.row_scale && lane < nr) my_scale = __builtin_nontemporal_load(ep.row_scale + r0 + lane);
  const int nxt = __shfl_down(my_ptr, 1);
  const unsigned long long hubmask = __ballot(lane < nr && (nxt - my_ptr) > hub_T);
  int my_gid = 0;
  if constexpr (FUSED) {
    if (fe.row_ids && lane < nr) my_gid = fe.row_ids[r0 + lane];
  }
  int my_xp0 = -1, my_xp1 = -1;      // MIXB: lane i holds the position of local row i in the compact operands
  if constexpr (MIXB) {
    if (fe.mx_n > 0 && lane < nr) my_xp0 = __builtin_nontemporal_load(fe.mx_pos[0] + r0 + lane);
    if (fe.mx_n > 1 && lane < nr) my_xp1 = __builtin_nontemporal_load(fe.mx_pos[1] + r0 + lane);
  }

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
    stream_rows<VEC, U, FULL, FUSED, ACC, HT, GP, 0, false, CS, MIXB>(0, nr, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, active, ep.relu, bvec, fe, c0,
                                                    init_lane, ep.ld_init, ep, nullptr, 0, my_gid, my_xp0, my_xp1, cs_acc);
  } else {
    int r = 0;
    while (r < nr) {  // maximal hub-free runs; hub rows are written by the hub kernels
      unsigned long long m = hubmask >> r;
      int nh = m ? r + (__ffsll((long long)m) - 1) : nr;
      if (nh > r)
        stream_rows<VEC, U, FULL, FUSED, ACC, HT, GP, 0, false, CS, MIXB>(r, nh, nr, my_ptr, my_scale, r0, col, h_lane, ld_h, out_lane, ld_out, active, ep.relu, bvec, fe,
                                                        c0, init_lane, ep.ld_init, ep, nullptr, 0, my_gid, my_xp0, my_xp1, cs_acc);
      r = nh + 1;
    }
  }
  if constexpr (MIXB) {
    if (fe.cs_partial) block_colsum_store(cs_acc, fe.cs_partial + (int64_t)(fe.cs_block0 + blockIdx.x) * d + blockIdx.y * 256, lane * 4, true);
  }
}

// One wavefront per chunk of T edges of a hub row -> one partial row in `partial`.
template <int VEC, int U, typename HT, int GP = 0, bool CS = false>
__global__ void __launch_bounds__(256) k_spmm_hub_chunks(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const HT* __restrict__ h, int64_t ld_h, int d, int hub_T,
                                                         int n_hubs, int n_chunks, const int* __restrict__ hub_rows,
                                                         const int* __restrict__ hub_chunk_ptr, float* __restrict__ partial,
                                                         int64_t ld_p, Epilogue ep) {
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
    float my_cs = 0.f;
    if constexpr (CS) {
      if (lane < cnt) my_cs = ep.col_scale[my_col & kColMask];
    }
    int k = 0;
    for (; k + U <= cnt; k += U) {
      float v[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = bcast_lane(my_col, k + u);
        if (active) gather_pol<VEC, HT, GP>(v[u], h_lane, ld_h, c);
        else zero<VEC>(v[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (CS) {
          const float cs = __int_as_float(bcast_lane(__float_as_int(my_cs), k + u));
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += cs * v[u][i];
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
      if constexpr (CS) {
        const float cs = __int_as_float(bcast_lane(__float_as_int(my_cs), k));
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += cs * v[i];
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
template <int VEC, bool FUSED, bool MIXB = false>
__global__ void __launch_bounds__(256) k_spmm_hub_finish(int d, int n_hubs, const int* __restrict__ hub_rows,
                                                         const int* __restrict__ hub_chunk_ptr,
                                                         const float* __restrict__ partial, int64_t ld_p,
                                                         float* __restrict__ out, int64_t ld_out, Epilogue ep, FusedEpi fe) {
  const int lane = lane_id();
  const int i = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  float cs_acc[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MIXB) {      // (d % 256 == 0: every lane has columns; every wavefront of the block hands its column sums over)
    if (i >= n_hubs) {
      if (fe.cs_partial) block_colsum_store(cs_acc, fe.cs_partial + (int64_t)(fe.cs_block0 + blockIdx.x) * d + blockIdx.y * 256, lane * 4, false);
      return;
    }
  } else {
    if (i >= n_hubs) return;
  }
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
  const float s = ep.row_scale ? ep.row_scale[row] : 1.f;
  if constexpr (FUSED) {
    float a4[4], b4[4], rmix[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { a4[k] = acc[k % VEC]; b4[k] = bvec[k % VEC]; }
    const int64_t grow = fe.row_ids ? (int64_t)fe.row_ids[row] : (int64_t)row;
    if (fe.mix_src) {
      float t[VEC];
      gather_stream<VEC>(t, fe.mix_src + grow * fe.ld_mix + c0);
#pragma unroll
      for (int k = 0; k < VEC; ++k) rmix[k] = t[k];
    }
    float x4[4];
    if constexpr (MIXB) {
      float xm[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      int xp[2] = {-1, -1};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < fe.mx_n) xp[q] = fe.mx_pos[q][row];
        if (xp[q] >= 0) gather_stream<4>(xm[q], fe.mx_g[q] + (int64_t)xp[q] * fe.d + c0);
      }
      fused_store_bwd_mix(fe, (int64_t)row, c0, a4, s, x4, xm, xp, cs_acc);
      if (fe.cs_partial) block_colsum_store(cs_acc, fe.cs_partial + (int64_t)(fe.cs_block0 + blockIdx.x) * d + blockIdx.y * 256, lane * 4, true);
    } else {
      fused_store(fe, (int64_t)row, c0, a4, s, b4, rmix, x4, grow);
    }
  } else {
    {
      if (ep.lp_mix) {      // label-propagation store (narrow rows only ever set it)
        bool on[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) on[k] = true;
        write_row_lp<VEC>(out + (int64_t)row * ld_out + c0, acc, s, ep, (int64_t)row, c0, on);
        return;
      }
    }
    float rv[VEC];
    write_row<VEC>(out + (int64_t)row * ld_out + c0, acc, s, bvec, ep.relu, rv);
  }
}

}  // namespace cb
