// Narrow-feature CSR sum-aggregation (d <= 16) for gfx950: the width of the last GCNConv(H, C) in the
// non-residual mode (GNN_model/GCN.py:70-71, C = 7 on Cora) and of every label-propagation step
// (Label_propagation_model/outcome_correlation.py:128-145, d = C = 40 / 47).  Same contract and epilogue as
// k_spmm_rows in cb_spmm.hip (out[v] = act(row_scale[v] * sum_{u in row v} h[u] + bias), GCN.py:238-253).
//
// Why a second kernel: with one 64-lane wavefront per gathered row, a 7-float row keeps 7 lanes busy and moves 28 bytes
// per load instruction.  Here a wavefront is cut into W = 64 / G groups of G lanes; one group gathers one edge's row
// (G x VEC floats), so ONE load instruction brings W neighbour rows.  Bound: HBM (random d*4-byte reads: whole 64-byte
// sectors move, so the useful fraction is d*4 / sectors touched); algorithmic bytes E*(4d + 4) + N*(4d + 4) [+4N].
//
// Mapping: a wavefront owns RPW consecutive destination rows and walks their contiguous CSR edges as one stream,
// W edges per window (group g takes edge base + g), U windows of loads in flight.  Row boundaries inside a window
// are resolved with a segmented inclusive scan across the groups (log2 W shuffle steps; rows are ascending along the
// stream, so "partner has my row" is the segment test); the tail group of every finished row applies the epilogue and
// stores; the unfinished last segment is carried into the next window.  Work per wavefront is therefore the edge
// count, not the longest row, and there are no atomics (bit-reproducible).  Rows above the hub threshold are cut out
// of the stream exactly as in cb_spmm.hip and reduced by k_small_hub_chunks (one wavefront per chunk of T edges, groups
// take edges round-robin, one cross-group reduction at the end) + k_small_hub_finish.
#include "cb_common.h"
#include "cb_spmm_small.h"

namespace cb {

namespace {

template <int VEC>
__device__ __forceinline__ void ldrow(float (&v)[VEC], const float* __restrict__ p) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void strow(float* __restrict__ p, const float (&v)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) __builtin_nontemporal_store(v[i], p + i);
}

struct SmallEpi {
  const float* row_scale;
  const float* bias;
  int relu;
};

// epilogue + store of one finished row by the lanes of one group (`on` = this lane's group owns the row and its columns exist)
template <int VEC>
__device__ __forceinline__ void finish_row(bool on, float* __restrict__ out, int64_t ld_out, int64_t row, int c0, const float (&acc)[VEC],
                                           float scale, const float (&b)[VEC], int relu) {
  if (!on) return;
  float r[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float t = scale_add(acc[i], scale, b[i]);   // rst * norm + bias   (GCN.py:250,253)
    r[i] = relu ? fmaxf(t, 0.f) : t;
  }
  strow<VEC>(out + row * ld_out + c0, r);
}

// Streams the edges of local rows [rlo, rhi) (no hub row inside).  my_ptr: lane i holds rowptr[r0 + i], i <= nr.
template <int G, int VEC, int U>
__device__ __forceinline__ void stream_small(int rlo, int rhi, int nr, int my_ptr, float my_scale, int r0, const int* __restrict__ col,
                                             const float* __restrict__ h, int64_t ld_h, float* __restrict__ out, int64_t ld_out,
                                             int d, const SmallEpi& ep, const float (&bvec)[VEC]) {
  constexpr int W = kWave / G;
  const int lane = lane_id(), g = lane / G, l = lane % G, c0 = l * VEC;
  const bool colok = c0 < d;
  const int e_begin = bcast_lane(my_ptr, rlo), e_end = bcast_lane(my_ptr, rhi);
  // empty rows are never met by the stream: write act(bias) for them now (lanes of group 0)
  {
    const int nxt = __shfl_down(my_ptr, 1);
    unsigned long long em = __ballot(lane >= rlo && lane < rhi && nxt == my_ptr);
    while (em) {
      const int r = __ffsll((long long)em) - 1;
      em &= em - 1;
      float z[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) z[i] = 0.f;
      finish_row<VEC>(g == 0 && colok, out, ld_out, (int64_t)r0 + r, c0, z, 1.f, bvec, ep.relu);
    }
  }
  if (e_begin >= e_end) return;
  int carry_row = -1;            // wave-uniform: local row whose partial sum is carried (replicated in every group), -1 = none
  float carry[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) carry[i] = 0.f;
  const unsigned long long ptr_lanes = (nr >= 63) ? ~0ull : ((1ull << (nr + 1)) - 1ull);

  for (int base = e_begin; base < e_end; base += U * W) {
    // ---- issue the loads of U windows ----
    float v[U][VEC];
    int ecol[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * W + g;
      ecol[u] = (e < e_end) ? __builtin_nontemporal_load(col + e) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ecol[u] >= 0 && colok) ldrow<VEC>(v[u], h + (int64_t)ecol[u] * ld_h + c0);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
      }
    }
    // ---- consume them window by window ----
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int wbase = base + u * W;
      if (wbase >= e_end) break;                       // wave-uniform
      const int nvalid = min(W, e_end - wbase);        // groups 0 .. nvalid-1 hold an edge
      // local row of every group's edge: rid_k = #{rows i : rowptr[r0+i] <= e_k} - 1 (one ballot per group, wave-uniform)
      int rid = 0x3fffffff - g;                        // invalid groups: distinct, never equal to a real row
      int first_rid = 0, last_rid = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (k < nvalid) {
          const unsigned long long m = __ballot(my_ptr <= wbase + k) & ptr_lanes;
          const int rk = __popcll(m) - 1;
          if (g == k) rid = rk;
          if (k == 0) first_rid = rk;
          last_rid = rk;
        }
      }
      // segmented inclusive scan across the groups
      float s[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] = v[u][i];
#pragma unroll
      for (int st = 1; st < W; st <<= 1) {
        const int pr = __shfl_up(rid, st * G);
        float pv[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) pv[i] = __shfl_up(s[i], st * G);
        if (g >= st && pr == rid) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) s[i] += pv[i];
        }
      }
      // the carried row: continues in this window (first segment) or is finished
      if (carry_row >= 0) {
        if (carry_row == first_rid) {
          if (rid == first_rid) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] += carry[i];
          }
        } else {
          const float sc = __int_as_float(bcast_lane(__float_as_int(my_scale), carry_row));
          finish_row<VEC>(g == 0 && colok, out, ld_out, (int64_t)r0 + carry_row, c0, carry, sc, bvec, ep.relu);
        }
      }
      // tails of finished rows store; the last segment becomes the carry
      const int nrid = __shfl_down(rid, G);
      const bool tail = (g < nvalid) && (g == nvalid - 1 || nrid != rid);
      const bool done = tail && rid != last_rid;
      const float sc = __shfl(my_scale, done ? rid : 0);
      finish_row<VEC>(done && colok, out, ld_out, (int64_t)r0 + (done ? rid : 0), c0, s, sc, bvec, ep.relu);
#pragma unroll
      for (int i = 0; i < VEC; ++i) carry[i] = __shfl(s[i], (nvalid - 1) * G + l);
      carry_row = last_rid;
    }
  }
  if (carry_row >= 0) {
    const float sc = __int_as_float(bcast_lane(__float_as_int(my_scale), carry_row));
    finish_row<VEC>(g == 0 && colok, out, ld_out, (int64_t)r0 + carry_row, c0, carry, sc, bvec, ep.relu);
  }
}

template <int G, int VEC, int RPW, int U>
__global__ void __launch_bounds__(256) k_spmm_small(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ h,
                                                    int64_t ld_h, float* __restrict__ out, int64_t ld_out, int n_rows, int d, SmallEpi ep,
                                                    int hub_T) {
  static_assert(RPW < kWave, "row block + end pointer must fit the lanes");
  const int lane = lane_id();
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int r0 = wave * RPW;
  if (r0 >= n_rows) return;
  const int nr = min(RPW, n_rows - r0);
  const int c0 = (lane % G) * VEC;
  const int my_ptr = __builtin_nontemporal_load(rowptr + r0 + min(lane, nr));
  float my_scale = 1.f;
  if (ep.row_scale && lane < nr) my_scale = __builtin_nontemporal_load(ep.row_scale + r0 + lane);
  const int nxt = __shfl_down(my_ptr, 1);
  const unsigned long long hubmask = __ballot(lane < nr && (nxt - my_ptr) > hub_T);
  float bvec[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) bvec[i] = (ep.bias && c0 + i < d) ? ep.bias[c0 + i] : 0.f;
  if (hubmask == 0) {
    stream_small<G, VEC, U>(0, nr, nr, my_ptr, my_scale, r0, col, h, ld_h, out, ld_out, d, ep, bvec);
  } else {
    int r = 0;
    while (r < nr) {   // maximal hub-free runs; hub rows are written by the hub kernels
      const unsigned long long m = hubmask >> r;
      const int nh = m ? r + (__ffsll((long long)m) - 1) : nr;
      if (nh > r) stream_small<G, VEC, U>(r, nh, nr, my_ptr, my_scale, r0, col, h, ld_h, out, ld_out, d, ep, bvec);
      r = nh + 1;
    }
  }
}

// One wavefront per chunk of T edges of a hub row -> one raw partial row in `partial`.
template <int G, int VEC, int U>
__global__ void __launch_bounds__(256) k_small_hub_chunks(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const float* __restrict__ h, int64_t ld_h, int d, int hub_T, int n_hubs,
                                                          int n_chunks, const int* __restrict__ hub_rows,
                                                          const int* __restrict__ hub_chunk_ptr, float* __restrict__ partial, int64_t ld_p) {
  constexpr int W = kWave / G;
  const int lane = lane_id(), g = lane / G, l = lane % G, c0 = l * VEC;
  const int chunk = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (chunk >= n_chunks) return;
  int lo = 0, hi = n_hubs;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (hub_chunk_ptr[mid] <= chunk) lo = mid; else hi = mid;
  }
  const int row = hub_rows[lo];
  const int e_begin = rowptr[row] + (chunk - hub_chunk_ptr[lo]) * hub_T;
  const int e_end = min(e_begin + hub_T, rowptr[row + 1]);
  const bool colok = c0 < d;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int base = e_begin; base < e_end; base += U * W) {
    float v[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * W + g;
      const int c = (e < e_end) ? __builtin_nontemporal_load(col + e) : -1;
      if (c >= 0 && colok) ldrow<VEC>(v[u], h + (int64_t)c * ld_h + c0);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
  }
#pragma unroll
  for (int st = G; st < kWave; st <<= 1)      // fixed-order tree across the groups
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], st);
  if (g == 0 && colok) {
    float* p = partial + (int64_t)chunk * ld_p + c0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = acc[i];
  }
}

// One lane per (hub row, column): partials summed in chunk order, then the epilogue.
__global__ void __launch_bounds__(256) k_small_hub_finish(int d, int n_hubs, const int* __restrict__ hub_rows,
                                                          const int* __restrict__ hub_chunk_ptr, const float* __restrict__ partial,
                                                          int64_t ld_p, float* __restrict__ out, int64_t ld_out, SmallEpi ep) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(t / d), c = (int)(t % d);
  if (i >= n_hubs) return;
  const int row = hub_rows[i];
  float acc = 0.f;
  for (int k = hub_chunk_ptr[i]; k < hub_chunk_ptr[i + 1]; ++k) acc += partial[(int64_t)k * ld_p + c];
  const float v = scale_add(acc, ep.row_scale ? ep.row_scale[row] : 1.f, ep.bias ? ep.bias[c] : 0.f);
  out[(int64_t)row * ld_out + c] = ep.relu ? fmaxf(v, 0.f) : v;
}

template <int G, int VEC>
int launch_small_t(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, int64_t d, SmallEpi ep, float* out,
                   int64_t ld_out, int hub_T, int n_hubs, int n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                   float* partial, int64_t ld_p, hipStream_t st) {
  constexpr int RPW = 32, U = 4, WPB = 4;
  const int64_t n_waves = (N + RPW - 1) / RPW;
  hipLaunchKernelGGL((k_spmm_small<G, VEC, RPW, U>), dim3((unsigned)((n_waves + WPB - 1) / WPB)), dim3(kWave * WPB), 0, st, rowptr, col, h,
                     ld_h, out, ld_out, (int)N, (int)d, ep, hub_T);
  CB_LAUNCH_CHECK();
  if (n_hubs > 0) {
    hipLaunchKernelGGL((k_small_hub_chunks<G, VEC, U>), dim3((unsigned)((n_chunks + WPB - 1) / WPB)), dim3(kWave * WPB), 0, st, rowptr,
                       col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, ld_p);
    CB_LAUNCH_CHECK();
    const int64_t work = (int64_t)n_hubs * d;
    hipLaunchKernelGGL(k_small_hub_finish, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, (int)d, n_hubs, hub_rows, hub_chunk_ptr,
                       partial, ld_p, out, ld_out, ep);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

}  // namespace

// Measured on the ogbn-products-shaped graph (2.45 M rows, 1.24e8 edges; profiles/r02_spmm_narrow_widths.md): the grouped stream
// wins for d <= 16 (d = 7: 2.60 vs 2.90 ms, d = 16: 2.40 vs 2.91 ms); from d = 32 up the per-window bookkeeping (one ballot
// per group, log2 W shuffle steps per component) costs more than the idle lanes of one-wavefront-per-row (d = 40: 5.37 vs
// 4.32 ms = 0.60 of the HBM roofline on the S 8(d) bytes; d = 64: 5.67 vs 4.32 ms), so wider rows stay on k_spmm_rows.
bool spmm_small_eligible(int64_t d, bool /*al16*/) { return d >= 1 && d <= 16; }

int launch_spmm_small(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, int64_t d, const float* row_scale,
                      const float* bias, int relu, float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks,
                      const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial, int64_t ld_p, bool al16, hipStream_t st) {
  SmallEpi ep{row_scale, bias, relu};
#define CB_SMALL_ARGS rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, ld_p, st
  if (al16 && d > 4) return launch_small_t<4, 4>(CB_SMALL_ARGS);   // float4 lanes: 16-byte aligned rows, d in {8, 12, 16}
  if (d <= 4) return launch_small_t<4, 1>(CB_SMALL_ARGS);
  if (d <= 8) return launch_small_t<8, 1>(CB_SMALL_ARGS);
  return launch_small_t<16, 1>(CB_SMALL_ARGS);
#undef CB_SMALL_ARGS
}

}  // namespace cb
