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

#include "cb_spmm_core.h"

namespace cb {

static inline int64_t partial_ld(int64_t d) { return (d + 3) / 4 * 4; }

// Gather policy of a launch: 2 when the caller hands over flagged column ids (col_flags: hot source rows keep the default cache policy,
// every other gather streams), else 0 (all-streaming measured slower: profiles/r02_spmm_gather_policy.md).
static inline int gather_policy(int col_flags) { return col_flags ? 2 : 0; }

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
    bool cs_done = false;      // source-row factor (ep.col_scale): the d % 256 == 0 fp32 kernels with the plain store
    if constexpr (kGP && !FUSED && sizeof(HT) == 4) {
      if (ep.col_scale) {
        CB_CHECK_ARG(!acc && d % tile == 0, CB_E_INVALID, "a source-row factor needs d %% 256 == 0 and no running sums");
        if (gp == 2)
          hipLaunchKernelGGL((k_spmm_rows<VEC, RPW, U, true, false, HT, false, 2, true>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h,
                             ld_h, out, ld_out, (int)N, (int)d, ep, hub_T, fe);
        else
          hipLaunchKernelGGL((k_spmm_rows<VEC, RPW, U, true, false, HT, false, 0, true>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h,
                             ld_h, out, ld_out, (int)N, (int)d, ep, hub_T, fe);
        cs_done = true;
      }
    }
    if (cs_done) {
    } else if constexpr (FUSED) {
      if (acc) { if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, true, true, 2); else CB_ROWS_LAUNCH(true, true, true); } else CB_ROWS_LAUNCH(true, true, true); }
      else if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, true, false, 2); else CB_ROWS_LAUNCH(true, true, false); }
      else CB_ROWS_LAUNCH(true, true, false);
    } else if (d % tile == 0) {
      if (acc) { if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, false, true, 2); else CB_ROWS_LAUNCH(true, false, true); } else CB_ROWS_LAUNCH(true, false, true); }
      else if constexpr (kGP) { if (gp == 2) CB_ROWS_LAUNCH_GP(true, false, false, 2); else CB_ROWS_LAUNCH(true, false, false); }
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
    bool cs_hub = false;
    if constexpr (kGPh && !FUSED && sizeof(HT) == 4) {
      if (ep.col_scale) {
        if (gph == 2)
          hipLaunchKernelGGL((k_spmm_hub_chunks<VEC, 8, HT, 2, true>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h, ld_h, (int)d, hub_T,
                             n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, ld_p, ep);
        else
          hipLaunchKernelGGL((k_spmm_hub_chunks<VEC, 8, HT, 0, true>), grid, dim3(kWave * waves_per_block), 0, st, rowptr, col, h, ld_h, (int)d, hub_T,
                             n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, ld_p, ep);
        cs_hub = true;
      }
    }
    if (cs_hub) {
    } else if constexpr (kGPh) { if (gph == 2) CB_HUB_LAUNCH(2); else CB_HUB_LAUNCH(0); }
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

template <int VEC, bool FUSED = false, typename HT = float>
static int launch_spmm(const int32_t* rowptr, const int32_t* col, int64_t N, const HT* h, int64_t ld_h, int64_t d,
                       Epilogue ep, float* out, int64_t ld_out, int hub_T, int n_hubs, int n_chunks,
                       const int32_t* hub_rows, const int32_t* hub_chunk_ptr, float* partial, hipStream_t st,
                       FusedEpi fe = FusedEpi{}) {
  // 16 rows per wavefront, 8 gathers in flight: the sweep of row-block / unroll shapes is in profiles/r01_* (all within +-1 %)
  return launch_spmm_cfg<VEC, FUSED, 16, 8, HT>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st, fe);
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
                           const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream, const float* col_scale = nullptr) {
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
  ep.col_scale = col_scale;
  ep.acc_skip_empty = acc_init && acc_init == out && ld_init == ld_out && !row_scale && !bias && !relu;      // raw in-place pass: rows without edges stay untouched
  CB_CHECK_ARG(!col_flags || d % 256 == 0, CB_E_INVALID, "%s: flagged column ids need d %% 256 == 0", who);
  hipStream_t st = (hipStream_t)stream;
  if (n_hubs == 0) hub_T = INT32_MAX;  // no plan given (or no hub rows): every row is reduced whole by one wavefront
  const bool ini16 = !acc_init || (((uintptr_t)acc_init % 16 == 0) && ld_init % 4 == 0);
  const bool ini8 = !acc_init || (((uintptr_t)acc_init % 8 == 0) && ld_init % 2 == 0);
  const bool al16 = ((uintptr_t)h % 16 == 0) && ((uintptr_t)out % 16 == 0) && (ld_h % 4 == 0) && (ld_out % 4 == 0) && (d % 4 == 0) && ini16;
  const bool al8 = ((uintptr_t)h % 8 == 0) && ((uintptr_t)out % 8 == 0) && (ld_h % 2 == 0) && (ld_out % 2 == 0) && (d % 2 == 0) && ini8;
  float* partial = (float*)ws;
  CB_CHECK_ARG(!col_flags || al16, CB_E_INVALID, "%s: flagged column ids need 16-byte aligned rows", who);
  if (col_scale) {
    CB_CHECK_ARG(al16 && d % 256 == 0 && !acc_init && !bias && !relu, CB_E_INVALID,
                 "%s: a source-row factor needs 16-byte aligned rows with d %% 256 == 0, no bias / ReLU / running sums", who);
    return launch_spmm<4>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, partial, st);
  }
  if (!acc_init && spmm_small_eligible(d, al16))
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

// out[v, :] = row_scale[v] * sum_{u in row v} col_scale[u] * h[u, :]: the aggregation with a factor per SOURCE row applied as the row is
// gathered (d % 256 == 0).  The row-sparse backward takes A (a * X_l) on the loss rows with it — the weight gradient of the level
// contracted over the loss rows (trunk.py; autograd of GCN.py:213,238 re-associated) — without a scaled copy of X_l.
extern "C" int cb_spmm_csr_colscale_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                        int64_t ld_h, int64_t d, const float* col_scale, const float* row_scale, float* out, int64_t ld_out,
                                        int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                        const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(col_scale != nullptr || N == 0 || d == 0, CB_E_INVALID, "cb_spmm_csr_colscale_f32: col_scale is null");
  return spmm_plain_impl("cb_spmm_csr_colscale_f32", rowptr, col, col_flags, N, E, h, ld_h, d, row_scale, nullptr, 0, nullptr, 0, out, ld_out, hub_T,
                         n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream, col_scale);
}

// One label-propagation step with its two elementwise passes folded into the store (Label_propagation_model/outcome_correlation.py:137-143,
// alpha_term = True, post_step = clamp(0, 1); trainer :33-63):
//     out[v, :] = post_scale[v] * clamp(row_scale[v] * sum_{u in row v} h[u, :] + c_mix * mix[v, :], 0, 1)
// With h = D^-1/2 result_t, row_scale = alpha D^-1/2, mix = y0, c_mix = 1 - alpha and post_scale = D^-1/2 the output IS the next step's
// gather operand D^-1/2 result_{t+1}; post_scale = NULL on the last step returns result itself.  Narrow rows (d = number of classes):
// one lane per column (the VEC = 1 instantiation of k_spmm_rows + hub kernels).
extern "C" int cb_spmm_csr_lp_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d,
                                  const float* row_scale, const float* mix, int64_t ld_mix, float c_mix, const float* post_scale, float* out,
                                  int64_t ld_out, int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                  const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d >= 0, CB_E_INVALID, "cb_spmm_csr_lp_f32: negative size");
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "cb_spmm_csr_lp_f32: size exceeds the int32 contract");
  if (N == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out && mix && (E == 0 || col), CB_E_INVALID, "cb_spmm_csr_lp_f32: null pointer");
  CB_CHECK_ARG(ld_h >= d && ld_out >= d && ld_mix >= d, CB_E_INVALID, "cb_spmm_csr_lp_f32: leading dimension smaller than d");
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "cb_spmm_csr_lp_f32: bad hub plan");
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)), CB_E_WORKSPACE,
               "cb_spmm_csr_lp_f32: hub plan given but workspace missing/too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, nullptr, 0, nullptr, 0, 0};
  ep.lp_mix = mix; ep.ld_lp = ld_mix; ep.lp_c_mix = c_mix; ep.lp_post = post_scale;
  return launch_spmm<1>(rowptr, col, N, h, ld_h, d, ep, out, ld_out, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws, (hipStream_t)stream);
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

static int spmm_fused_impl(int h_bf16, const float* acc_init, int64_t ld_init, int col_flags, const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, const void* h, int64_t ld_h,
                                     int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                                     float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only,
                                     float* out_act, int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_T,
                                     int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                                     void* ws, size_t ws_bytes, void* stream, const int32_t* row_ids = nullptr) {
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
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits; fe.bits_relu_only = bits_relu_only;
  fe.out_act = out_act; fe.ld_act = ld_act; fe.out_next = out_next; fe.ld_next = ld_next; fe.d = (int)d; fe.row_ids = row_ids;
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
      uint64_t *relu_bits, int32_t bits_relu_only, float *out_act, int64_t ld_act, float *out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs,        \
      int32_t n_chunks, const int32_t *hub_rows, const int32_t *hub_chunk_ptr, void *ws, size_t ws_bytes, void *stream
#define CB_FUSED_ARGS                                                                                                            \
  rowptr, col, N, E, h, ld_h, d, row_scale, bias, mix_src, ld_mix, c_act, c_mix, drop_p, seed, seed_dev, row0, relu_bits, bits_relu_only, out_act, ld_act,  \
      out_next, ld_next, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, ws, ws_bytes, stream

extern "C" int cb_spmm_csr_fused_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h,
                                     int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                                     float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                     uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                     int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                                     const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  return spmm_fused_impl(0, nullptr, 0, col_flags, CB_FUSED_ARGS);
}

// The same over a CSR whose rows are a SUBSET of the node rows (rows-only forward, trunk.py): row r of the CSR / of row_scale / of out_act / out_next is
// node row row_ids[r] (ascending); mix_src, relu_bits and the dropout mask are taken at the node row.
extern "C" int cb_spmm_csr_fused_rows_f32(const int32_t* row_ids, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E,
                                          const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                                          int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                          uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                          int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                          size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(row_ids != nullptr || N == 0, CB_E_INVALID, "cb_spmm_csr_fused_rows_f32: row_ids is null");
  return spmm_fused_impl(0, nullptr, 0, col_flags, CB_FUSED_ARGS, row_ids);
}

// A (reverse) aggregation whose epilogue is the BACKWARD of the trunk's store of the rows it writes (the row-sparse backward's dense level, trunk.py):
//   g = row_scale * sum  -> out_g (the gradient w.r.t. the stored, dropped activation: kept for the input stage's mix gather)
//   out_gr = bwd_rowscale * c_act * dropout_bwd_seed(g) where relu_bits (READ: written by the forward's store) has the element's bit, else 0
// = cb_spmm_csr_f32 followed by cb_trunk_layer_bwd_f32 without the pass's read of g (bit-identical values); the bias gradient of that pass (column sums of
// the masked gradient) is taken by cb_trunk_input_bwd_multi_cs_f32, which reads g anyway.  d % 256 == 0, fp32 rows.  row_ids (may be null): the CSR's rows are a
// subset of the node rows (a compact level: row r = node row_ids[r]) — mask words, dropout mask and bwd_rowscale at the node row, row_scale / out_g / out_gr compact.
extern "C" int cb_spmm_csr_store_bwd_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                         int64_t d, const float* row_scale, const uint64_t* relu_bits, const float* bwd_rowscale, float c_act, float drop_p,
                                         uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* out_g, int64_t ld_g, float* out_gr, int64_t ld_gr,
                                         int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                         size_t ws_bytes, const int32_t* row_ids, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d > 0 && d % 256 == 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_f32: d must be a positive multiple of 256");
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "cb_spmm_csr_store_bwd_f32: size exceeds the int32 contract");
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out_gr && relu_bits && (E == 0 || col), CB_E_INVALID, "cb_spmm_csr_store_bwd_f32: null pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_f32: dropout p / row offset out of range");
  CB_CHECK_ARG(((uintptr_t)h % 16 == 0) && ((uintptr_t)out_gr % 16 == 0) && ld_h % 4 == 0 && ld_gr % 4 == 0 && ld_h >= d && ld_gr >= d &&
                   (!out_g || ((uintptr_t)out_g % 16 == 0 && ld_g % 4 == 0 && ld_g >= d)) && ((uintptr_t)relu_bits % 8 == 0),
               CB_E_INVALID, "cb_spmm_csr_store_bwd_f32: 16-byte aligned rows required");
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_f32: bad hub plan");
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)), CB_E_WORKSPACE,
               "cb_spmm_csr_store_bwd_f32: hub plan given but workspace missing/too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, nullptr, 0, nullptr, 0, col_flags};
  FusedEpi fe{};
  fe.bwd = 1; fe.bwd_rowscale = bwd_rowscale; fe.c_act = c_act;
  fe.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fe.keep_scale = 1.f / (1.f - drop_p);
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits;
  fe.out_act = out_g; fe.ld_act = ld_g; fe.out_next = out_gr; fe.ld_next = ld_gr; fe.d = (int)d; fe.row_ids = row_ids;
  return launch_spmm<4, true, float>(rowptr, col, N, h, ld_h, d, ep, out_gr, ld_gr, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr, (float*)ws,
                                     (hipStream_t)stream, fe);
}

namespace cb {
// partial[p][c], p < nparts, summed in groups of `per` consecutive rows: thread = column (coalesced), fixed order -> folded[g][c]
__global__ void __launch_bounds__(256) k_colsum_fold(const float* __restrict__ partial, int nparts, int d, int per, float* __restrict__ folded) {
  const int g = blockIdx.x;
  const int p0 = g * per, p1 = min(nparts, p0 + per);
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = p0;
    for (; p + 4 <= p1; p += 4) {
      s0 += partial[(int64_t)p * d + c]; s1 += partial[(int64_t)(p + 1) * d + c];
      s2 += partial[(int64_t)(p + 2) * d + c]; s3 += partial[(int64_t)(p + 3) * d + c];
    }
    for (; p < p1; ++p) s0 += partial[(int64_t)p * d + c];
    folded[(int64_t)g * d + c] = (s0 + s1) + (s2 + s3);
  }
}
// out[c] = sum_g folded[g][c]: 32 columns x 8 strided group lanes per block, fixed order
__global__ void __launch_bounds__(256) k_colsum_last(const float* __restrict__ folded, int ngroups, int d, float* __restrict__ out) {
  __shared__ float s_t[8][32];
  const int cl = threadIdx.x & 31, gl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < d) {
#pragma unroll 4
    for (int g = gl; g < ngroups; g += 8) s += folded[(int64_t)g * d + c];
  }
  s_t[gl][cl] = s;
  __syncthreads();
  if (gl == 0 && c < d)
    out[c] = ((s_t[0][cl] + s_t[1][cl]) + (s_t[2][cl] + s_t[3][cl])) + ((s_t[4][cl] + s_t[5][cl]) + (s_t[6][cl] + s_t[7][cl]));
}
constexpr int kFoldGroups = 1024;
static inline int64_t mix_row_blocks(int64_t N) { return ((N + 15) / 16 + 3) / 4; }
static inline int64_t mix_hub_blocks(int64_t n_hubs) { return (n_hubs + 3) / 4; }
}  // namespace cb

extern "C" size_t cb_spmm_store_bwd_mix_workspace_bytes(int64_t N, int64_t n_hubs, int64_t d) {
  if (N <= 0 || d <= 0) return 0;
  return (size_t)(mix_row_blocks(N) + mix_hub_blocks(n_hubs > 0 ? n_hubs : 0) + kFoldGroups) * (size_t)d * sizeof(float);
}

// cb_spmm_csr_store_bwd_f32 on ALL node rows whose first output is not the raw gradient g but the FOLDED mix gradient
//   out_m = c_mix * ( dropout_bwd_seed(g) + sum_q dropout_bwd_{mix_seeds[q]}(mix_g[q][mix_pos[q][row]]) )        (n_mix <= 2 compact operands; pos < 0: absent)
// — everything the layers above and this store send to X0 through their residual mixes (res_tricks.py:23, under each store's own dropout GCN.py:110,133),
// so that the input stage (cb_gemm_tn_instage_f32) reads ONE [N, d] matrix beside dL/d dropout(X0) — and colsum (may be null) = the column sums of
// out_gr / bwd_rowscale: the bias gradient of the store whose backward this is (autograd of GCN.py:253).  ws2: cb_spmm_store_bwd_mix_workspace_bytes.
extern "C" int cb_spmm_csr_store_bwd_mix_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                             int64_t d, const float* row_scale, const uint64_t* relu_bits, const float* bwd_rowscale, float c_act, float drop_p,
                                             uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* out_m, int64_t ld_m, float* out_gr, int64_t ld_gr,
                                             int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                             size_t ws_bytes, int32_t n_mix, const float* const* mix_g, const int32_t* const* mix_pos, const uint64_t* mix_seeds,
                                             float c_mix, float* colsum, void* ws2, size_t ws2_bytes, void* stream) {
  CB_CHECK_ARG(N >= 0 && E >= 0 && d > 0 && d % 256 == 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: d must be a positive multiple of 256");
  CB_CHECK_ARG(N < INT32_MAX && E < INT32_MAX && d < (1 << 20), CB_E_RANGE, "cb_spmm_csr_store_bwd_mix_f32: size exceeds the int32 contract");
  if (N == 0) return CB_OK;
  CB_CHECK_ARG(rowptr && h && out_gr && out_m && relu_bits && (E == 0 || col), CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: null pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: dropout p / row offset out of range");
  CB_CHECK_ARG(((uintptr_t)h % 16 == 0) && ((uintptr_t)out_gr % 16 == 0) && ((uintptr_t)out_m % 16 == 0) && ld_h % 4 == 0 && ld_gr % 4 == 0 && ld_m % 4 == 0 &&
                   ld_h >= d && ld_gr >= d && ld_m >= d && ((uintptr_t)relu_bits % 8 == 0),
               CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: 16-byte aligned rows required");
  CB_CHECK_ARG(n_mix >= 0 && n_mix <= 2 && (n_mix == 0 || (mix_g && mix_pos && mix_seeds)), CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: 0..2 compact mix operands");
  CB_CHECK_ARG(hub_T > 0 && n_hubs >= 0 && n_chunks >= 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: bad hub plan");
  CB_CHECK_ARG(n_hubs == 0 || (hub_rows && hub_chunk_ptr && ws && ws_bytes >= cb_spmm_workspace_bytes(n_chunks, d)), CB_E_WORKSPACE,
               "cb_spmm_csr_store_bwd_mix_f32: hub plan given but workspace missing/too small");
  CB_CHECK_ARG(!colsum || (ws2 && ws2_bytes >= cb_spmm_store_bwd_mix_workspace_bytes(N, n_hubs, d)), CB_E_WORKSPACE,
               "cb_spmm_csr_store_bwd_mix_f32: column-sum workspace missing/too small");
  if (n_hubs == 0) hub_T = INT32_MAX;
  Epilogue ep{row_scale, nullptr, 0, nullptr, 0, col_flags};
  FusedEpi fe{};
  fe.bwd = 1; fe.bwd_rowscale = bwd_rowscale; fe.c_act = c_act;
  fe.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fe.keep_scale = 1.f / (1.f - drop_p);
  fe.seed = seed; fe.seed_dev = seed_dev; fe.row0 = row0; fe.bits = (unsigned long long*)relu_bits;
  fe.out_act = out_m; fe.ld_act = ld_m; fe.out_next = out_gr; fe.ld_next = ld_gr; fe.d = (int)d;
  fe.mx_n = n_mix; fe.mx_c = c_mix;
  for (int q = 0; q < n_mix; ++q) {
    CB_CHECK_ARG(mix_g[q] && mix_pos[q] && (uintptr_t)mix_g[q] % 16 == 0, CB_E_INVALID, "cb_spmm_csr_store_bwd_mix_f32: null or misaligned mix operand %d", q);
    fe.mx_g[q] = mix_g[q]; fe.mx_pos[q] = mix_pos[q]; fe.mx_seed[q] = mix_seeds[q];
  }
  fe.cs_partial = colsum ? (float*)ws2 : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int ny = (int)(d / 256);
  const int64_t nb_rows = mix_row_blocks(N), nb_hub = n_hubs > 0 ? mix_hub_blocks(n_hubs) : 0;
  {
    dim3 grid((unsigned)nb_rows, ny);
    fe.cs_block0 = 0;
    if (col_flags)
      hipLaunchKernelGGL((k_spmm_rows<4, 16, 8, true, true, float, false, 2, false, true>), grid, dim3(256), 0, st, rowptr, col, h, ld_h, out_gr, ld_gr, (int)N, (int)d, ep,
                         hub_T, fe);
    else
      hipLaunchKernelGGL((k_spmm_rows<4, 16, 8, true, true, float, false, 0, false, true>), grid, dim3(256), 0, st, rowptr, col, h, ld_h, out_gr, ld_gr, (int)N, (int)d, ep,
                         hub_T, fe);
    CB_LAUNCH_CHECK();
  }
  if (n_hubs > 0) {
    const int64_t ld_p = partial_ld(d);
    dim3 grid((unsigned)((n_chunks + 3) / 4), ny);
    if (col_flags)
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, 8, float, 2>), grid, dim3(256), 0, st, rowptr, col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                         (float*)ws, ld_p, ep);
    else
      hipLaunchKernelGGL((k_spmm_hub_chunks<4, 8, float, 0>), grid, dim3(256), 0, st, rowptr, col, h, ld_h, (int)d, hub_T, n_hubs, n_chunks, hub_rows, hub_chunk_ptr,
                         (float*)ws, ld_p, ep);
    CB_LAUNCH_CHECK();
    fe.cs_block0 = (int)nb_rows;
    hipLaunchKernelGGL((k_spmm_hub_finish<4, true, true>), dim3((unsigned)nb_hub, ny), dim3(256), 0, st, (int)d, n_hubs, hub_rows, hub_chunk_ptr, (const float*)ws, ld_p,
                       out_gr, ld_gr, ep, fe);
    CB_LAUNCH_CHECK();
  }
  if (colsum) {
    const int nparts = (int)(nb_rows + nb_hub);
    const int per = (nparts + kFoldGroups - 1) / kFoldGroups;
    const int ngroups = (nparts + per - 1) / per;
    float* folded = (float*)ws2 + (size_t)nparts * d;
    hipLaunchKernelGGL(k_colsum_fold, dim3((unsigned)ngroups), dim3(256), 0, st, (const float*)ws2, nparts, (int)d, per, folded);
    CB_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_last, dim3((unsigned)((d + 31) / 32)), dim3(256), 0, st, (const float*)folded, ngroups, (int)d, colsum);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// Fused store of the residual trunk on top of the interior-column partial sums (second pass of the node-sharded aggregation)
extern "C" int cb_spmm_csr_fused_acc_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags,
                                         int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias,
                                         const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                         const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act,
                                         float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                         const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0, CB_E_INVALID, "cb_spmm_csr_fused_acc_f32: acc_init is null");
  return spmm_fused_impl(0, acc_init, ld_init, col_flags, CB_FUSED_ARGS);
}

extern "C" int cb_spmm_csr_fused_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h,
                                          int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                                          int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                          const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
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
  ep.acc_skip_empty = acc_init && acc_init == out && ld_init == ld_out && !row_scale && !bias && !relu;
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
                                              uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only,
                                              float* out_act, int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_T, int32_t n_hubs, int32_t n_chunks,
                                              const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(acc_init != nullptr || N == 0, CB_E_INVALID, "cb_spmm_csr_fused_acc_bf16_f32: acc_init is null");
  return spmm_fused_impl(1, acc_init, ld_init, col_flags, CB_FUSED_ARGS);
}
