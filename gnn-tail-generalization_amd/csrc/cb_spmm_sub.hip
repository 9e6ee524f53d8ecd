// Sum-aggregation for rows of 17 .. 64 floats (d % 4 == 0, 16-byte aligned): the width of a label-propagation step (d = number of classes,
// Label_propagation_model/outcome_correlation.py:128-156: 40 on ogbn-arxiv, 47 -> 48 padded on ogbn-products) and of the last GCNConv(H, C)
// of the non-residual mode (GNN_model/GCN.py:70-71).  Same contract and epilogues as k_spmm_rows (plain: act(row_scale * sum + bias); the
// label-propagation store of cb_spmm_csr_lp_f32).
//
// An EXPERIMENT kept opt-in (CB_SPMM_SUB=1, see spmm_sub_eligible below for the numbers): with one wavefront per gathered row
// (k_spmm_rows<1>) a launch at d = 40, 48 and 64 takes the SAME 4.3 – 4.4 ms on the ogbn-products shape, and the question was whether that is a
// limit of one gathered edge per wave-wide load instruction.  It is not: this kernel gathers 4 – 8 edges per instruction and is no faster,
// because those widths are bound by the 128-byte lines they touch (two per edge).  The grouped-stream kernel (cb_spmm_small.hip) brings several
// edges per instruction too but pays a segmented scan across the groups per window, which loses from d = 32 up.  Here the wavefront is cut into
// S = 64 / L independent STREAMS of L lanes (L x float4 >= d): every stream walks its own run of consecutive destination rows, so ONE load
// instruction gathers S neighbour rows and nothing is ever reduced across lanes.  The price is per-lane control: column ids are loaded per
// lane (the L lanes of a stream read the same address), row boundaries are per-lane predicates resolved in a wave-wide "while any lane must
// flush" loop, rowptr / row scales are looked up with ds_bpermute from the wave's 64 preloaded values.
// The wavefront owns 64 consecutive rows; inside every hub-free run of them the rows are cut among the streams at equal EDGE counts (ballots
// over the preloaded rowptr values).  Hub rows go to the hub kernels of cb_spmm.hip as before.  No atomics: bit-reproducible.
#include "cb_common.h"
#include "cb_spmm_core.h"
#include "cb_spmm_small.h"

namespace cb {

namespace {

constexpr int kSubRows = 64;     // destination rows per wavefront

template <int L, int U>
__global__ void __launch_bounds__(256) k_spmm_sub(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ h,
                                                  int64_t ld_h, float* __restrict__ out, int64_t ld_out, int n_rows, int d, Epilogue ep, int hub_T) {
  constexpr int S = kWave / L;
  const int lane = lane_id();
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int r0 = wave * kSubRows;
  if (r0 >= n_rows) return;
  const int nr = min(kSubRows, n_rows - r0);
  const int g = lane / L, l = lane % L, c0 = 4 * l;
  const bool colok = c0 < d;
  // lane i: rowptr[r0 + i] (i < 64) — looked up per lane with ds_bpermute; ptr_hi = rowptr[r0 + nr]
  const int my_ptr = rowptr[r0 + min(lane, nr)];
  const int ptr_hi = rowptr[r0 + nr];
  float my_scale = 1.f;
  if (ep.row_scale && lane < nr) my_scale = ep.row_scale[r0 + lane];
  int nxt = __shfl_down(my_ptr, 1);
  if (lane == kWave - 1) nxt = ptr_hi;
  const unsigned long long hubmask = __ballot(lane < nr && (nxt - my_ptr) > hub_T);
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (ep.bias && colok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bvec[i] = ep.bias[c0 + i];
  }
  // per-lane i.  The shuffle is issued by EVERY lane before the select: inside the ternary the lanes with i == 64 would sit it out, and a
  // ds_bpermute that reads a lane outside EXEC returns 0
  auto ptr_at = [&](int i) {
    const int v = __shfl(my_ptr, i & (kWave - 1));
    return i >= kWave ? ptr_hi : v;
  };

  auto store_row = [&](bool on, int row_local, const float (&acc)[4]) {      // epilogue + store by the lanes of one stream
    const float s = __shfl(my_scale, row_local);      // (every lane takes part in the shuffle)
    if (!(on && colok)) return;
    const int64_t row = r0 + row_local;
    float r[4];
    if (ep.lp_mix) {
      const float post = ep.lp_post ? ep.lp_post[row] : 1.f;
      const float4 m = *reinterpret_cast<const float4*>(ep.lp_mix + row * ep.ld_lp + c0);
      const float mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = acc[i] * s;
        t = t + ep.lp_c_mix * mv[i];
        r[i] = fminf(fmaxf(t, 0.f), 1.f) * post;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = scale_add(acc[i], s, bvec[i]);      // rst * norm + bias (GCN.py:250,253)
        r[i] = ep.relu ? fmaxf(t, 0.f) : t;
      }
    }
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t v4 = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(v4, reinterpret_cast<f4_t*>(out + row * ld_out + c0));
  };

  int run = 0;
  while (run < nr) {      // maximal hub-free runs [run, nh) of the wavefront's rows (wave-uniform)
    const unsigned long long hm = hubmask >> run;
    const int nh = hm ? run + (__ffsll((long long)hm) - 1) : nr;
    if (nh > run) {
      // cut rows [run, nh) among the S streams at equal edge counts: stream g takes the rows whose first edge lies in its share
      const int e_lo = bcast_lane(my_ptr, run), e_hi = nh >= kWave ? ptr_hi : bcast_lane(my_ptr, nh);
      const int64_t span = (int64_t)e_hi - e_lo;
      int ra = run, rb = nh;
#pragma unroll
      for (int q = 1; q < S; ++q) {
        const int t = e_lo + (int)(span * q / S);
        const int cut = run + (int)__popcll(__ballot(lane >= run && lane < nh && my_ptr < t));      // first row of stream q (uniform)
        if (g >= q) ra = cut;
        if (g < q && cut < rb) rb = cut;
      }
      // this lane's stream: rows [ra, rb), edges [e, e_end)
      int cur = ra;
      int e = ptr_at(ra);
      const int e_end = ptr_at(rb);
      int cur_end = ptr_at(min(ra + 1, kWave));
      bool live = ra < rb;                      // rows left to finish
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      while (__any(live && e < e_end)) {
        int cidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cidx[u] = (live && e + u < e_end) ? col[e + u] : -1;
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cidx[u] >= 0 && colok) v[u] = *reinterpret_cast<const float4*>(h + (int64_t)cidx[u] * ld_h + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // rows of this stream that end in front of edge e + u (also empty ones) are finished first
          bool need = live && (e + u) == cur_end && (e + u) < e_end;
          while (__any(need)) {
            store_row(need, cur, acc);
            if (need) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[i] = 0.f;
              ++cur;
            }
            const int ne = ptr_at(min(cur + 1, kWave));
            if (need) cur_end = ne;
            need = need && (e + u) == cur_end && cur < rb;
          }
          acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w;
        }
        e += U;
      }
      // the stream's last row and its trailing empty rows
      bool need = live && cur < rb;
      while (__any(need)) {
        store_row(need, cur, acc);
        if (need) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = 0.f;
          ++cur;
        }
        need = need && cur < rb;
      }
    }
    run = nh + 1;
  }
}

}  // namespace

bool spmm_sub_eligible(int64_t d, bool al16) {
  // Opt-in (CB_SPMM_SUB=1).  Measured on the ogbn-products shape (profiles/r03_spmm_narrow_widths.md): 2.88 vs 3.00 ms at d = 32, but 4.62 vs
  // 4.34 ms at d = 40, 4.66 vs 4.38 at d = 48, 4.72 vs 4.35 at d = 64 and 3.55 vs 3.11 at d = 20 — from d = 40 a gathered row spans two
  // 128-byte lines and k_spmm_rows already moves ~57 G lines/s (7.3 TB/s of lines), so more edges per load instruction buy nothing and the
  // per-lane control costs; no width the reference runs (C = 3 .. 47) lies in the 24 < d <= 32 window where it wins.
  static const bool on = getenv("CB_SPMM_SUB") != nullptr;
  return on && al16 && d > 16 && d <= 64 && d % 4 == 0;
}

int launch_spmm_sub(const int32_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t ld_h, int64_t d, const Epilogue& ep, float* out,
                    int64_t ld_out, int hub_T, int n_hubs, int n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial,
                    int64_t ld_p, hipStream_t st) {
  const int64_t n_waves = (N + kSubRows - 1) / kSubRows;
  const dim3 grid((unsigned)((n_waves + 3) / 4)), blk(256);
  static const int u = getenv("CB_SPMM_SUB_U") ? atoi(getenv("CB_SPMM_SUB_U")) : 8;
  if (d <= 32) {
    if (u == 4) hipLaunchKernelGGL((k_spmm_sub<8, 4>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, (int)d, ep, hub_T);
    else hipLaunchKernelGGL((k_spmm_sub<8, 8>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, (int)d, ep, hub_T);
  } else {
    if (u == 4) hipLaunchKernelGGL((k_spmm_sub<16, 4>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, (int)d, ep, hub_T);
    else hipLaunchKernelGGL((k_spmm_sub<16, 8>), grid, blk, 0, st, rowptr, col, h, ld_h, out, ld_out, (int)N, (int)d, ep, hub_T);
  }
  CB_LAUNCH_CHECK();
  if (n_hubs > 0) {      // hub rows: the chunk / finish kernels of the wide path (float4 lanes; the finish kernel knows the lp store)
    const dim3 gridc((unsigned)((n_chunks + 3) / 4), 1);
    hipLaunchKernelGGL((k_spmm_hub_chunks<4, 8, float, 0>), gridc, blk, 0, st, rowptr, col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks, hub_rows,
                       hub_chunk_ptr, partial, ld_p, ep);
    CB_LAUNCH_CHECK();
    const dim3 grid2((unsigned)((n_hubs + 3) / 4), 1);
    hipLaunchKernelGGL((k_spmm_hub_finish<4, false>), grid2, blk, 0, st, (int)d, n_hubs, hub_rows, hub_chunk_ptr, partial, ld_p, out, ld_out, ep,
                       FusedEpi{});
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

}  // namespace cb
