/*
 * coldbrew_hip.h — C ABI of libcoldbrew_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the Cold Brew TeacherGNN hot path.  The reference has no FFI
 * of its own: its hot path bottoms out in third-party Python packages (dgl 0.7.0's
 * gspmm behind `graph.update_all`, torch's matmul / elementwise / Adam).  Each entry
 * point below replaces one of those call sites; the citation names the reference line
 * (paths relative to the reference root) whose work it does.
 *
 * Conventions (all functions):
 *   - plain pointers are DEVICE pointers unless the name says `host`; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is
 *     asynchronous on that stream and re-entrant across streams;
 *   - nothing is allocated or freed: scratch memory comes from the caller
 *     (cb_*_workspace_bytes) and must stay alive until the stream has passed the call;
 *   - row-major matrices, leading dimension `ld` in elements; fp32 rows must be 4-byte
 *     aligned (16-byte alignment + ld % 4 == 0 enables the vector paths);
 *   - indices are int32 (E < 2^31, N < 2^31); edge_index arrives as int64 (PyG contract);
 *   - return value 0 = ok, negative = error (CB_E_*); cb_last_error() gives the
 *     thread-local message of the last failing call.
 */
#ifndef COLDBREW_HIP_H
#define COLDBREW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CB_OK 0
#define CB_E_INVALID (-1)   /* bad argument (null pointer, negative size, misaligned)  */
#define CB_E_RANGE (-2)     /* size does not fit the int32 index contract              */
#define CB_E_WORKSPACE (-3) /* workspace too small                                     */
#define CB_E_HIP (-4)       /* a HIP runtime call / kernel launch failed               */

int cb_version(void);
const char* cb_last_error(void);

/* ------------------------------------------------------------------------------------
 * Graph ingest — replaces `dgl.graph((src_list, dst_list))` built through Python lists
 * (GNN_model/GCN.py:92-95) and the degree queries (GCN.py:188,206,243).
 *
 * src = edge_index[0], dst = edge_index[1], E columns, multigraph (duplicates kept).
 * Produces the by-dst CSR (rowptr/col: row v lists the sources u of edges u->v — the
 * forward aggregation) and the by-src CSR (rowptr_t/col_t: the reverse graph used by the
 * backward).  In-row order is ascending column id, so the arrays are a deterministic
 * function of the edge multiset.
 * flags[0] = #nodes with in-degree 0   (GCN.py:187-197 raises if non-zero)
 * flags[1] = #edges with an endpoint outside [0, N)  (arrays are invalid if non-zero)
 * flags[2] = 1 if the edge multiset is symmetric (by-src CSR == by-dst CSR), else 0
 * flags[3] = max in-degree
 * ---------------------------------------------------------------------------------- */
size_t cb_csr_workspace_bytes(int64_t E, int64_t N);
int cb_csr_from_coo_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                        int32_t* rowptr, int32_t* col, int32_t* rowptr_t, int32_t* col_t,
                        int32_t* flags /*[4]*/, void* workspace, size_t workspace_bytes, void* stream);

/* norm[v] = max(rowptr[v+1]-rowptr[v], 1)^-1/2 as fp32 — `degs.float().clamp(min=1)` then
 * `th.pow(degs, -0.5)` (GCN.py:206-208 with the by-src rowptr, GCN.py:243-245 with the by-dst rowptr). */
int cb_deg_norm_f32(const int32_t* rowptr, int64_t N, float* norm, void* stream);

/* ------------------------------------------------------------------------------------
 * Hub plan for the aggregation: rows longer than `hub_threshold` edges are split into
 * chunks of `hub_threshold` edges that separate wavefronts reduce (power-law graphs).
 * counts[0] = #hub rows, counts[1] = #chunks.  Call cb_spmm_hub_count, read counts on the
 * host, allocate hub_rows[counts[0]] and hub_chunk_ptr[counts[0]+1], call cb_spmm_hub_fill.
 * ---------------------------------------------------------------------------------- */
int cb_spmm_hub_count(const int32_t* rowptr, int64_t N, int32_t hub_threshold, int32_t* counts /*[2]*/, void* stream);
int cb_spmm_hub_fill(const int32_t* rowptr, int64_t N, int32_t hub_threshold, int32_t n_hubs,
                     int32_t* hub_rows, int32_t* hub_chunk_ptr, int32_t* cursor /*[1] scratch*/, void* stream);

/* ------------------------------------------------------------------------------------
 * Sum aggregation — replaces `graph.update_all(fn.copy_src('h','m'), fn.sum('m','h'))`
 * (GCN.py:198,238: DGL gspmm copy_lhs/sum) with the post-scale, bias (GCN.py:242-253) and
 * the ReLU that follows it in TricksComb.forward (GCN.py:127-128) fused as an epilogue:
 *
 *     out[v, :] = act( row_scale[v] * sum_{j in [rowptr[v], rowptr[v+1])} h[col[j], :] + bias[:] )
 *
 * row_scale / bias may be NULL (factor 1 / no bias); relu = 0/1.  Called with the by-dst
 * CSR for the forward and with the by-src CSR (no epilogue) for the backward
 * (autograd of gspmm = SpMM on the reverse graph).
 * `ws` holds the hub partial sums: cb_spmm_workspace_bytes(n_chunks, d).
 * Deterministic: every row is reduced in CSR order by one wavefront, hub rows in chunk order.
 * ---------------------------------------------------------------------------------- */
size_t cb_spmm_workspace_bytes(int64_t n_chunks, int64_t d);
int cb_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E,
                    const float* h, int64_t ld_h, int64_t d,
                    const float* row_scale, const float* bias, int relu,
                    float* out, int64_t ld_out,
                    int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                    const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                    void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COLDBREW_HIP_H */
