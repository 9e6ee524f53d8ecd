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
#define CB_E_DEVICE (-5)    /* a kernel recorded a device-side error (cb_device_status) */

/* device-side error codes (first int of the device error word) */
#define CB_DEVERR_HANDOVER 1 /* the LDS tile hand-over of the aggregation + GEMM kernel gave up its bounded wait */
#define CB_DEVERR_GRADROWS 2 /* cb_rows_zero_outside_mask_f32 found a non-zero gradient row outside the loss rows */

int cb_version(void);
const char* cb_last_error(void);
/* Device-side errors are never silent: a kernel that gives up a bounded wait (the only such wait is the LDS tile hand-over of the
 * cb_spmm_gemm_* kernels: a lost hand-over would otherwise mean wrong numbers in a training run that has no parity test beside it)
 * records the reason in a word of device-visible host memory.  Every later cb_spmm_gemm_* call returns CB_E_DEVICE until
 * cb_device_status() has been called; cb_device_status() returns CB_OK, or CB_E_DEVICE with the reason in cb_last_error(), and clears
 * the word.  It does not synchronise — call it after the synchronisation that ends a step.  New relative to the reference (torch raises
 * on device faults by itself; GNN_model/GCN.py:238 + :225 are the work of those kernels). */
int cb_device_status(void);

/* Row-sparse backward (new; autograd of trainer_node_classification.py:390-391 `nll_loss(log_softmax(out[train_mask]))`): the gradient that
 * enters the model is zero in every row outside the loss rows, and so is everything the head and the last trunk store make of it — the
 * first reverse aggregation of the backward (autograd of GCN.py:238) then only has to gather the loss rows (graph.CSRGraph.filtered_t).
 * This call verifies the claim on the device: any non-zero element of g [rows, d] in a row with mask[row] == 0 records
 * CB_DEVERR_GRADROWS in the device error word (cb_device_status).  guard (may be NULL): an int32 in DEVICE memory the caller owns; a
 * violation also sets it to 1.  Handed to the optimiser launch that follows on the same stream (cb_adam_multi_norm_f32), it keeps the
 * truncated gradients of such a step away from the parameters and the moments without a host synchronisation in between; the caller
 * clears it when it handles the error. */
int cb_rows_zero_outside_mask_f32(const float* g, int64_t ld, int64_t rows, int64_t d, const uint8_t* mask, int32_t* guard, void* stream);

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

/* E >= 2^31 on one device ("indices int32 (int64 rowptr if E >= 2^31)"; the reference's graph is int64 throughout,
 * GCN.py:93-94): the same ingest with int64 row pointers (column ids stay int32: N < 2^31; E < 2^32), the degree norms from them, and
 * cb_csr_rebase_i64: out[i] = rowptr[row0 + i] - rowptr[row0], i <= n_rows — the int32 row pointers of a row block with fewer than 2^31
 * edges, whose column ids start at col + rowptr[row0].  The aggregation kernels index edges with 32 bits inside a launch; the host cuts the
 * rows into such blocks (graph.SegmentedCSRGraph) and every entry point above runs per block, as it does on a rank's row block of the
 * node-sharded path. */
int cb_csr64_from_coo_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t* rowptr, int32_t* col, int64_t* rowptr_t,
                          int32_t* col_t, int32_t* flags /*[4]*/, void* workspace, size_t workspace_bytes, void* stream);
int cb_csr_rebase_i64(const int64_t* rowptr, int64_t row0, int64_t n_rows, int32_t* out, void* stream);
int cb_deg_norm_i64ptr_f32(const int64_t* rowptr, int64_t N, float* norm, void* stream);

/* ------------------------------------------------------------------------------------
 * Hub plan for the aggregation: rows longer than `hub_threshold` edges are split into
 * chunks of `hub_threshold` edges that separate wavefronts reduce (power-law graphs).
 * counts[0] = #hub rows, counts[1] = #chunks.  Call cb_spmm_hub_count, read counts on the
 * host, allocate hub_rows[counts[0]] and hub_chunk_ptr[counts[0]+1], call cb_spmm_hub_fill.
 * hub_rows comes back in ASCENDING row order (no atomics: the plan — and with it the order of every
 * sum a kernel takes over the hub rows — is the same in every process); scratch: int32
 * [cb_spmm_hub_fill_scratch_ints(N)].
 * ---------------------------------------------------------------------------------- */
int cb_spmm_hub_count(const int32_t* rowptr, int64_t N, int32_t hub_threshold, int32_t* counts /*[2]*/, void* stream);
int64_t cb_spmm_hub_fill_scratch_ints(int64_t N);
int cb_spmm_hub_fill(const int32_t* rowptr, int64_t N, int32_t hub_threshold, int32_t n_hubs,
                     int32_t* hub_rows, int32_t* hub_chunk_ptr, int32_t* scratch, void* stream);

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
 * `ws` holds the hub partial sums: cb_spmm_workspace_bytes(n_chunks, d).  With n_hubs == 0 (no plan) every
 * row is reduced whole by one wavefront — correct for any graph, slow for power-law hubs.
 * Deterministic: every row is reduced in CSR order by one wavefront, hub rows in chunk order.
 * col_flags = 0: `col` holds plain column ids.  col_flags = 1 (fp32 rows, d % 256 == 0, 16-byte aligned only): bit 31 of every
 * id marks a HOT source row (one of the most-referenced rows, chosen at graph build so that together they fit the 256 MiB
 * Infinity Cache); hot rows are gathered with the default cache policy, all others with the streaming (nt) policy, which
 * keeps the re-used rows resident instead of letting 1e8 single-use 1 KiB rows evict them (-11 % per launch on the
 * 10M-node power-law graph, profiles/r02_spmm_gather_policy.md).  Results do not depend on the flags.
 * ---------------------------------------------------------------------------------- */
size_t cb_spmm_workspace_bytes(int64_t n_chunks, int64_t d);
int cb_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E,
                    const float* h, int64_t ld_h, int64_t d,
                    const float* row_scale, const float* bias, int relu,
                    float* out, int64_t ld_out,
                    int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                    const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                    void* ws, size_t ws_bytes, void* stream);
/* out[v, :] = row_scale[v] * sum_{u in row v} col_scale[u] * h[u, :] — a factor per SOURCE row, applied as the row is gathered
 * (fp32 rows, d % 256 == 0, 16-byte aligned; col_scale: [number of columns]).  New: the row-sparse backward takes A (a * X_l) on the
 * loss rows with it, i.e. the weight gradient X_l^T (a * A^T dY) of GCN.py:213,238 contracted over the loss rows as
 * ((A (a * X_l))[S_0])^T dY[S_0] — without a scaled copy of X_l (trunk.py). */
int cb_spmm_csr_colscale_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E,
                             const float* h, int64_t ld_h, int64_t d, const float* col_scale, const float* row_scale,
                             float* out, int64_t ld_out, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                             const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);


/* ------------------------------------------------------------------------------------
 * Elementwise / reduction stages around the GEMM + aggregation pairs (HBM-bound).
 * ---------------------------------------------------------------------------------- */

/* out[i] = x[i] * keep(seed, offset + i) / (1 - p) — `F.dropout(x, p, training=True)` of GCN.py:104,110,133.
 * keep() is a counter-based Philox4x32-10 draw of (seed, flat index): the backward calls the same
 * function on the gradient with the same seed instead of storing a mask; `offset` = flat index of x[0]
 * in the unsharded tensor (0 on one GPU), so row shards draw the mask of the full tensor.
 * In-place (out == x) is allowed. */
int cb_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, int64_t offset,
                   void* stream);
/* seed_dev (all dropout-bearing entry points; may be NULL): a device word added to `seed` at kernel time, so a step
 * captured in a hipGraph draws fresh masks on every replay (the host advances *seed_dev inside the graph). */

/* out = a*x + b*y — ResidualConnection / InitialConnection (res_tricks.py:14,23) and gradient sums. */
int cb_axpby_f32(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream);

/* Backward of the aggregation epilogue  Y = act(row_scale * R + bias)  (autograd of GCN.py:250,253 and
 * F.relu GCN.py:128) in one pass over [rows, d] contiguous matrices:
 *     gm = g * (act > 0)  (act == NULL: gm = g);  colsum[c] = sum_r gm[r,c]  (= dbias; NULL to skip);
 *     out[r,c] = gm[r,c] * row_scale[r]  (row_scale NULL: factor 1; out NULL: column sums only).
 * Column sums are reduced in a fixed order (ws: cb_colsum_workspace_bytes(rows, d)). */
size_t cb_colsum_workspace_bytes(int64_t rows, int64_t d);
int cb_act_bwd_f32(const float* g, const float* act, const float* row_scale, float* out, int64_t rows, int64_t d,
                   float* colsum, void* ws, size_t ws_bytes, void* stream);

/* out2[0] = ||x||_F (`th.norm(self.le)`, GCN.py:232), out2[1] = sum x^2; ws: cb_reduce_workspace_bytes(). */
size_t cb_reduce_workspace_bytes(void);
int cb_frobenius_norm_f32(const float* x, int64_t n, float* out2, void* ws, size_t ws_bytes, void* stream);

/* Fused `F.nll_loss(F.log_softmax(logits[mask], 1), y[mask])` (trainer_node_classification.py:390-391)
 * and its gradient: loss[0] = mean over the `count` rows with mask != 0 (mask NULL: all rows);
 * grad [rows, C] contiguous (NULL to skip) = (softmax - onehot) / count on masked rows, 0 elsewhere.
 * mask is one byte per row (torch.bool). */
int cb_nll_logsoftmax_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask, int64_t rows, int64_t C,
                          int64_t count, float* loss, float* grad, void* ws, size_t ws_bytes, void* stream);

/* One fused Adam update of a parameter tensor — `torch.optim.Adam(params, lr, weight_decay)` of
 * trainer_node_classification.py:310,430 (L2-style weight decay added to the gradient, bias-corrected
 * moments, step >= 1). */
int cb_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int64_t step, const int64_t* step_dev, void* stream);
/* The same update for n_tensors parameter tensors in ONE launch (host arrays of device pointers; the table travels in the
 * kernel arguments, 24 tensors per launch).  Identical arithmetic to cb_adam_step_f32.
 * extra_decay (may be NULL, entries may be NULL): per tensor, a device float added to weight_decay for that tensor.  With
 * extra_decay[i] = se_reg / ||le_i||_F the update of a structural-embedding table includes the gradient of the regulariser
 * `se_reg * ||le||_F` (GCN.py:232,236; trainer_node_classification.py:393) without materialising `le / ||le||` and without the
 * autograd accumulation into le.grad: 50 bytes / element less HBM traffic per table per step. */
int cb_adam_multi_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                      const float* const* extra_decay, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      const int64_t* step_dev, const int32_t* guard, void* stream);
/* step_dev (may be NULL): the step count read from device memory instead of `step` (hipGraph replay).
 * guard (may be NULL): int32 in device memory; while it is non-zero the launch writes nothing (see cb_rows_zero_outside_mask_f32). */
/* The same, and for every tensor with norm_out[i] != NULL also norm_out[i][0] = ||p_i||_F, [1] = ||p_i||_F^2 of the UPDATED tensor
 * (cb_frobenius_norm_f32's pair, same thread-to-element map and summation order when the tensor is the largest of its launch): the
 * next forward's `th.norm(self.le)` (GCN.py:232) costs no pass of its own over the table.  ws: cb_adam_norm_workspace_bytes(number of
 * non-NULL entries). */
size_t cb_adam_norm_workspace_bytes(int32_t n_norms);
int cb_adam_multi_norm_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                           const float* const* extra_decay, float* const* norm_out, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, const int64_t* step_dev, const int32_t* guard, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------
 * Dense fp32 contractions on the matrix cores.  Default: every fp32 operand is decomposed exactly into three
 * bf16 limbs and the six leading limb products run as v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error
 * at the level of an fp32 GEMM, csrc/cb_gemm_limb.hip).  Operands that are not 16-byte aligned, or whose leading
 * dimensions / K / N are not multiples of 4, and every call when CB_GEMM_PLAIN_F32=1 is set in the environment,
 * use the fp32-input MFMA (v_mfma_f32_32x32x2_f32) instead.
 * Differences from an IEEE fp32 GEMM on the limb path: an infinite operand, or one above the largest bf16
 * (|x| > 3.39e38), yields NaN (inf - inf in the split) where fp32 would yield +-inf / a finite value; limbs that
 * fall below the smallest normal fp32 (|x| < ~2^-110) are flushed (absolute error < 2^-126).
 * ---------------------------------------------------------------------------------- */

/* C[M,N] = act( rowscale[m] * (A[M,K] @ B[K,N]) + addend[m,n] + bias[n] ) — `feat_src = feat * norm`,
 * `th.matmul(feat_src, weight)`, `+ self.le` (GCN.py:213,225,231) in one kernel (the row scale commutes
 * with the product); with bias/relu it is also nn.Linear + F.relu (GCN.py:105-106,138) and, fed with
 * gradients, the dX GEMMs of their backward.  rowscale/addend/bias may be NULL.
 * ws (optional, cb_gemm_nn_workspace_bytes(N, K) = 6 * K * N bytes + padding): room for B split ONCE per launch into its three
 * bf16 planes, pre-arranged as the kernel's LDS image — B is the small weight matrix, so this replaces the per-block,
 * per-K-step split of the same values by a straight 16-byte copy.  ws == NULL: every block splits its own copy. */
size_t cb_gemm_nn_workspace_bytes(int64_t N, int64_t K);
/* Few output tiles, long contraction (x @ W_0 of a Cora-sized graph, 2 708 x 1 433 x 64: GCN.py:105) on the fp32-input
 * fallback kernel: with ws of cb_gemm_nn_splitk_workspace_bytes(M, N, K) bytes (0 = the shape is not split) the K range is
 * cut into chunks contracted by different blocks and summed in a fixed order before the epilogue.  Without ws: one block
 * per output tile walks the whole K range (same result up to the summation order). */
size_t cb_gemm_nn_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int cb_gemm_nn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu,
                   void* ws, size_t ws_bytes, void* stream);
/* The same contraction with a second output C2 = dropout_p(C) written by the same epilogue (the value is in registers
 * anyway): `X = relu(Linear_0(x)); X_dropped = F.dropout(X)` of the residual trunk (GCN.py:105-107,110) without re-reading
 * X.  C2's keep-mask equals the one cb_dropout_f32 draws for (seed, seed_dev, offset = row0 * N).  Falls back to
 * cb_gemm_nn_f32 + cb_dropout_f32 (contiguous outputs) when the fused epilogue does not cover the shape. */
int cb_gemm_nn_drop2_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, float* C2, int64_t ldc2,
                         int64_t M, int64_t N, int64_t K, const float* rowscale, const float* addend, int64_t ld_add, const float* bias,
                         int relu, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes, void* stream);
/* C[K1,K2] = sum_m A[m,K1] * rowscale[m] * G[m,K2] — the weight gradients (autograd of GCN.py:225 and
 * of nn.Linear): a reduction over the node axis, split into row slabs whose partial products are summed
 * in a fixed order (ws: cb_gemm_tn_workspace_bytes).  rowscale may be NULL. */
size_t cb_gemm_tn_workspace_bytes(int64_t M, int64_t K1, int64_t K2);
int cb_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* C, int64_t M,
                   int64_t K1, int64_t K2, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused residual trunk (type_trick with 'Initial', no bare norm: the default configs of
 * Pubmed / ogbn-arxiv / the synthetic power-law benchmark).  Folds GCN.py:127-133 of layer l
 * and GCN.py:110 of layer l+1 into the aggregation's store:
 *     act      = relu(row_scale[v] * sum_j h[col[j]] + bias)                       (GCN.py:238-253,128)
 *     out_next = dropout_{seed,p}( c_act * act + c_mix * mix_src[v] )              (res_tricks.py:23, GCN.py:110/133)
 * relu_bits [N][d/256][4] uint64 receives the backward mask of the store (word k, bit l = column 256*tile + 4*l + k): set where
 * the element passes gradient to the pre-activation, i.e. act > 0 AND the dropout keeps it (p = 0: the ReLU mask); bits_relu_only != 0: act > 0
 * alone (the 'Residual' connection, res_tricks.py:7-14: layer l+1's mix sends a second gradient through this ReLU under ANOTHER dropout mask —
 * the backward kernels regenerate the keep masks anyway); out_act (nullable) the activation itself (the mix source of the next 'Residual' layer).  d must be a multiple of 256; rows 16-byte aligned.  row0 = global index
 * of local row 0 (dropout mask of the unsharded tensor).  mix_src NULL: no mix; drop_p 0: no dropout.
 * ---------------------------------------------------------------------------------- */
int cb_spmm_csr_fused_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                          int64_t d,
                          const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix, float c_act,
                          float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits,
                          int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_threshold, int32_t n_hubs,
                          int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes,
                          void* stream);

/* Backward of that epilogue in one pass over contiguous [rows, d]:
 *     gm = dropout_bwd(g);  gx0 = (accumulate ? gx0 : 0) + c_mix * gm  (gx0 NULL: skipped);
 *     gy = c_act * gm * relu_bit;  colsum = sum_rows gy (dbias; NULL to skip);  out = gy * row_scale[r]. */
int cb_trunk_layer_bwd_f32(const float* g, const uint64_t* relu_bits, const float* row_scale, void* out, int out_bf16, float* gx0,
                           int accumulate, int64_t rows, int64_t d, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                           int64_t row0, float c_act, float c_mix, const float* g2, uint64_t seed2, float c2, const int32_t* g2_pos, float* colsum,
                           void* ws, size_t ws_bytes, void* stream);
/* g2 (may be NULL): 'Residual' connection (res_tricks.py:7-14) — the layer's ReLU output also is the mix source of the NEXT layer, so
 *     gy = (c_act * dropout_bwd_seed(g) + c2 * dropout_bwd_seed2(g2)) * relu_bit     (g2 = gradient w.r.t. the next layer's stored output;
 * relu_bits written with bits_relu_only).  g2_pos (may be NULL; int32 [rows of the full matrix]): g2 is a COMPACT matrix of a row-sparse backward —
 * full row r sits at g2_pos[r], absent (= zero) where that is negative. */
/* (out_bf16 != 0: `out` is a bf16 [rows, d] matrix — gradient rows stored in bf16 for the bf16 aggregation variant.) */
/* cb_trunk_layer_bwd_f32 for layer 0 of the 'Initial' trunk (all rows, fp32 out, no in-place accumulator) that also FOLDS the gradients the residual mixes send
 * to X0 (InitialConnection, res_tricks.py:19-23, each under its own store's dropout GCN.py:110,133) — round 6, the elementwise form of
 * cb_spmm_csr_store_bwd_mix_f32's epilogue for the levels whose reverse aggregation does not carry the store backward:
 *   out_m = c_mix * ( dropout_bwd_seed(g) + sum_q dropout_bwd_{mix_seeds[q]}(mix_g[q][mix_pos[q][r] | r]) )      n_mix <= 2 operands (host arrays); mix_pos[q] NULL: a
 * dense [rows, d] operand, else int32 [rows] positions in a compact one (< 0: absent).  out / colsum: exactly cb_trunk_layer_bwd_f32's (bit-identical).  The
 * input stage then reads out_m beside dL/d dropout(X0): cb_gemm_tn_instage_f32.  colsum2 (may be NULL): the column sums of cs_c * dropout_bwd(mix_g[cs_src]) through
 * the mask words cs_bits (indexed by the node row) — the bias gradient (GCN.py:253) of a store whose backward cb_spmm_csr_store_bwd_f32 applied, as
 * cb_trunk_input_bwd_multi_cs_f32 returns it; ws2: cb_colsum_workspace_bytes(rows, d). */
int cb_trunk_layer_bwd_fold_f32(const float* g, const uint64_t* relu_bits, const float* row_scale, float* out, int64_t rows, int64_t d, float drop_p,
                                uint64_t seed, const uint64_t* seed_dev, int64_t row0, float c_act, float c_mix, int32_t n_mix, const float* const* mix_g,
                                const int32_t* const* mix_pos, const uint64_t* mix_seeds, float* out_m, float* colsum, void* ws, size_t ws_bytes, int32_t cs_src,
                                const uint64_t* cs_bits, float cs_c, float* colsum2, void* ws2, size_t ws2_bytes, void* stream);
/* The same over a SUBSET of the rows (row-sparse backward: the loss rows): g / out are compact [n_rows, d] matrices holding rows
 * row_index[0 .. n_rows) (ascending) of the full ones; relu_bits / row_scale are the full arrays; the dropout mask is the global row's. */
int cb_trunk_layer_bwd_rows_f32(const float* g, const int64_t* row_index, int64_t n_rows, const uint64_t* relu_bits, const float* row_scale, float* out,
                                int64_t d, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, float c_act, const float* g2, uint64_t seed2,
                                float c2, const int32_t* g2_pos, float* colsum, void* ws, size_t ws_bytes, void* stream);
/* (g2 / seed2 / c2 / g2_pos as in cb_trunk_layer_bwd_f32; g2_pos is required with g2: the two compact matrices live on different supports.) */

/* Backward into the trunk's input stage X0 = relu(Linear(dropout(x))) (GCN.py:104-107,110):
 *     out = (add + dropout_bwd(g)) * (act > 0);  colsum = sum_rows out  (bias gradient of the input Linear). */
int cb_trunk_input_bwd_f32(const float* g, const float* add, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                           uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                           void* stream);

/* The same input stage with the X0 gradient gathered in one pass (no [rows, d] accumulator is read-modify-written per layer):
 *     out = ( dropout_bwd_seed(g) + c_mix * sum_{l < n_mix} dropout_bwd_seeds_mix[l](g_mix[l]) ) * (act > 0)
 * g_mix[l] = gradient w.r.t. the output of layer l's fused store (host array of n_mix <= 7 device pointers); used with
 * cb_trunk_layer_bwd_f32(gx0 = NULL).  Autograd of GCN.py:104-110 + res_tricks.py:23 for every layer at once.
 * act_bits (may be NULL): [rows][d / 256][4] mask words of (act > 0) used instead of act (act may then be NULL).
 * g_mix_pos (may be NULL): host array of n_mix device pointers (entries may be NULL); where g_mix_pos[l] is set (int32 [rows]), g_mix[l] is a
 * COMPACT matrix that holds only some rows (the support rows of a row-sparse backward) — row r at position g_mix_pos[l][r], absent (= zero)
 * where that is negative. */
int cb_trunk_input_bwd_multi_f32(const float* g, uint64_t seed, int32_t n_mix, const float* const* g_mix, const uint64_t* seeds_mix,
                                 float c_mix, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                                 const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                                 const uint64_t* act_bits, const int32_t* const* g_mix_pos, void* stream);
/* The same, which also returns n_cs (<= 2) extra column sums (host arrays of n_cs entries): colsum2[q] = the column sums of cs_c[q] *
 * dropout_bwd_{seeds_mix[cs_src[q]]}(g_mix[cs_src[q]]) where the mask words cs_bits[q] ([rows][d / 256][4], indexed by the node row also for a compact operand)
 * have the element's bit: the bias gradients (GCN.py:253) of the layers whose store backward was applied by the reverse aggregation's own epilogue
 * (cb_spmm_csr_store_bwd_f32) — this pass reads those gradients anyway.  ws2: n_cs planes of cb_colsum_workspace_bytes(rows, d).  A dense operand's sum
 * has the partial-sum order of cb_trunk_layer_bwd_f32's (bit-identical). */
int cb_trunk_input_bwd_multi_cs_f32(const float* g, uint64_t seed, int32_t n_mix, const float* const* g_mix, const uint64_t* seeds_mix,
                                    float c_mix, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                                    const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                                    const uint64_t* act_bits, const int32_t* const* g_mix_pos, int32_t n_cs, const int32_t* cs_src,
                                    const uint64_t* const* cs_bits, const float* cs_c, float* const* colsum2, void* ws2, size_t ws2_bytes, void* stream);
/* A (reverse) aggregation whose epilogue is the BACKWARD of the trunk's store of the rows it writes — autograd of GCN.py:127-133 applied to the output of
 * the autograd of GCN.py:238 in one kernel:
 *   g = row_scale * sum_{u in row v} h[u]              -> out_g (may be NULL): dL/d(stored, dropped activation), the input stage's mix operand
 *   out_gr = bwd_rowscale[v] * c_act * dropout_bwd_seed(g) where relu_bits (READ: the forward store's mask words) has the element's bit, else 0
 * = cb_spmm_csr_f32 followed by cb_trunk_layer_bwd_f32 (gx0 = NULL, no second gradient) without that pass's read of g; values bit-identical.  The
 * pass's column sums (the bias gradient) come from cb_trunk_input_bwd_multi_cs_f32.  d % 256 == 0, fp32 rows, 16-byte aligned.  row_ids (may be NULL): the
 * CSR's rows are a subset of the node rows (row r = node row_ids[r]): relu_bits, the dropout mask and bwd_rowscale are taken at the node row. */
int cb_spmm_csr_store_bwd_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                              int64_t d, const float* row_scale, const uint64_t* relu_bits, const float* bwd_rowscale, float c_act, float drop_p,
                              uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* out_g, int64_t ld_g, float* out_gr, int64_t ld_gr,
                              int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                              size_t ws_bytes, const int32_t* row_ids, void* stream);
/* The same on ALL node rows, with the mix gradients FOLDED (round 6): the first output is not the raw g but everything this store and the layers above
 * send to X0 through their residual mixes (InitialConnection, res_tricks.py:19-23: X = (1 - alpha) X_l + alpha X_0, each under its own store's dropout
 * GCN.py:110,133):
 *   out_m = c_mix * ( dropout_bwd_seed(g) + sum_q dropout_bwd_{mix_seeds[q]}(mix_g[q][mix_pos[q][v]]) )          n_mix <= 2 compact operands (host arrays);
 * mix_pos[q] (int32 [N]): position of node row v in operand q, < 0 where it holds no such row.  out_gr as above (bit-identical).  colsum (may be NULL; [d]) =
 * column sums of out_gr / bwd_rowscale: the bias gradient (GCN.py:253) of the store whose backward this is; ws2 = cb_spmm_store_bwd_mix_workspace_bytes.
 * The input stage then reads ONE [N, d] matrix beside dL/d dropout(X0): cb_gemm_tn_instage_f32. */
size_t cb_spmm_store_bwd_mix_workspace_bytes(int64_t N, int64_t n_hubs, int64_t d);
int cb_spmm_csr_store_bwd_mix_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                                  int64_t d, const float* row_scale, const uint64_t* relu_bits, const float* bwd_rowscale, float c_act, float drop_p,
                                  uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* out_m, int64_t ld_m, float* out_gr, int64_t ld_gr,
                                  int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                                  size_t ws_bytes, int32_t n_mix, const float* const* mix_g, const int32_t* const* mix_pos, const uint64_t* mix_seeds,
                                  float c_mix, float* colsum, void* ws2, size_t ws2_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * bf16-storage variant of the aggregation (build extension = BASELINE config 2; the reference is fp32-only):
 * the rows that are gathered (Z in the forward, b*dY' in the backward) are stored as bf16 (round-to-nearest-
 * even), accumulation, outputs and everything else stay fp32.  Halves the dominant gather traffic.
 * Same semantics and arguments as cb_gemm_nn_f32 / cb_spmm_csr_f32 / cb_spmm_csr_fused_f32 otherwise
 * (ld in elements of the respective type; bf16 rows must be 8-byte aligned for the vector paths).
 * ---------------------------------------------------------------------------------- */
int cb_gemm_nn_bf16out_f32(const float* A, int64_t lda, const float* B, int64_t ldb, uint16_t* C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu,
                           void* ws, size_t ws_bytes, void* stream);
int cb_spmm_csr_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h, int64_t ld_h, int64_t d,
                         const float* row_scale, const float* bias, int relu, float* out, int64_t ld_out, int32_t hub_threshold,
                         int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                         size_t ws_bytes, void* stream);
int cb_spmm_csr_fused_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h, int64_t ld_h,
                               int64_t d, const float* row_scale, const float* bias, const float* mix_src, int64_t ld_mix,
                               float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                               uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                               int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                               const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Normalisation tricks (GNN_model/norm_tricks.py) as fused reductions; all matrices contiguous [rows, d].
 * ---------------------------------------------------------------------------------- */
/* node_norm (norm_tricks.py:53-84): y = (x - c*mu_r) * std_r^-q, mu/std over the features of row r
 * (std = sqrt(biased var + eps)); 'n': c=1,q=1; 'v': c=0,q=1; 'm': c=1,q=0; 'srv'/'pr': c=0,q=0.5.
 * stats2 [rows][2] = (mu, std) is written by the forward (may be NULL) and read by the backward. */
int cb_node_norm_fwd_f32(const float* x, float* y, float* stats2, int64_t rows, int64_t d, float c, float q, float eps, void* stream);
int cb_node_norm_bwd_f32(const float* x, const float* g, const float* stats2, float* dx, int64_t rows, int64_t d, float c, float q,
                         void* stream);
/* Column statistics in one pass: colsum[c] = sum_r x[r,c]; colsum2[c] = sum_r x[r,c]^2 (w == NULL) or
 * sum_r x[r,c]*w[r,c] — the reductions behind mean_norm / pair_norm / BatchNorm1d (norm_tricks.py:25-41,106,132)
 * and their backward; fixed-order two-stage reduce (ws: cb_colstats_workspace_bytes). */
size_t cb_colstats_workspace_bytes(int64_t rows, int64_t d);
int cb_colstats_f32(const float* x, const float* w, int64_t rows, int64_t d, float* colsum, float* colsum2, void* ws, size_t ws_bytes,
                    void* stream);
/* y[r,c] = ((x[r,c] - shift[c]) * scale[c]) * gscale + bias[c]  (vectors may be NULL). */
int cb_col_affine_f32(const float* x, const float* shift, const float* scale, const float* bias, float gscale, float* y, int64_t rows,
                      int64_t d, void* stream);
/* dx[r,c] = a[c]*ga*g[r,c] + b[c]*gb*xh[r,c] + e[c]  (xh, a, b, e may be NULL: a,b default 1) — the backward of the column norms. */
int cb_col_bwd_combine_f32(const float* g, const float* xh, const float* a, const float* b, const float* e, float ga, float gb,
                           float* dx, int64_t rows, int64_t d, void* stream);

/* ------------------------------------------------------------------------------------
 * Teacher -> student hand-off: `SEMLP.replacement` (MLP_model/__init__.py:143-156), a per-node
 * Python loop of [1,N] matmuls + argsort in the reference.  For every query row q_i [D]:
 *     s_j = <q_i, t_j> over the N teacher rows;  sel = the K largest (ties: larger index);
 *     out_i = sum_k softmax(s_sel)_k * t_sel_k          (contiguous out [B, D])
 * One fp32-MFMA sweep with a running top-K in LDS (the B x N score matrix is never written) + a merge kernel.
 * out_idx [B,K] / out_w [B,K] (nullable) receive the selection and the softmax weights in ascending score order
 * (the order of `sortidx[-K:]`).  1 <= K <= 8.  ws: cb_topk_replace_workspace_bytes(B, N, K).
 * ---------------------------------------------------------------------------------- */
size_t cb_topk_replace_workspace_bytes(int64_t B, int64_t N, int64_t K);
int cb_topk_replace_f32(const float* q, int64_t ldq, const float* t, int64_t ldt, int64_t B, int64_t N, int64_t D, int32_t K,
                        float* out, int32_t* out_idx, float* out_w, void* ws, size_t ws_bytes, void* stream);

/* cb_spmm_csr_fused_f32 over a CSR whose rows are a SUBSET of the node rows: row r of rowptr / row_scale / out_act / out_next is node row row_ids[r]
 * (ascending ids); mix_src, relu_bits ([all node rows][d/256][4]) and the dropout mask are taken at the node row.  The rows-only training forward
 * (trunk.py): a GCNConv + store (GCN.py:205-256,127-133) evaluated only on the rows the layers above read. */
int cb_spmm_csr_fused_rows_f32(const int32_t* row_ids, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E,
                               const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                               int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                               uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                               int32_t hub_T, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                               size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Node-sharded aggregation in two passes (new; the reference is single-device).  A rank's row block of
 * the CSR is split by column owner: the interior-column pass (plain cb_spmm_csr_f32, no epilogue) runs while the halo
 * rows travel over xGMI; the halo-column pass starts from those raw sums and applies the epilogue once:
 *     out[v, :] = act( row_scale[v] * (acc_init[v, :] + sum_{j in row v of THIS csr} h[col[j], :]) + bias[:] )
 * Same arguments as cb_spmm_csr_f32 / cb_spmm_csr_fused_f32 plus acc_init [N, ld_init] (fp32, read once; the plain
 * variant allows acc_init == out; a raw in-place pass — acc_init == out, no row scale / bias / ReLU: the intermediate halo slices —
 * neither reads nor writes rows that have no edge in this CSR).  Replaces the same reference lines as those two
 * (GCN.py:198,238-253,127-133).
 * ---------------------------------------------------------------------------------- */
int cb_spmm_csr_acc_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d,
                        const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init, float* out,
                        int64_t ld_out, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                        const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);
int cb_spmm_csr_fused_acc_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                              int64_t E,
                              const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias,
                              const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                              const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act,
                              float* out_next, int64_t ld_next, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                              const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);

/* The halo-column pass over bf16-stored halo rows: with the bf16 wire (COLDBREW_HALO_WIRE=bf16, opt-in, outside the 1e-4
 * parity) the rows leave cb_gather_rows_bf16_f32 narrowed (round-to-nearest-even), cross the links at 2 bytes per element and
 * are read by these passes exactly as they arrived (widened in registers; accumulation and outputs fp32) — no conversion pass
 * on either side.  Arguments as the fp32 forms, h = uint16_t rows (8-byte aligned for the vector paths). */
int cb_spmm_csr_acc_bf16_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const uint16_t* h, int64_t ld_h,
                             int64_t d, const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init,
                             float* out, int64_t ld_out, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                             const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);
int cb_spmm_csr_fused_acc_bf16_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags,
                                   int64_t N, int64_t E, const uint16_t* h, int64_t ld_h, int64_t d, const float* row_scale,
                                   const float* bias, const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p,
                                   uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act,
                                   int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_threshold, int32_t n_hubs,
                                   int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes,
                                   void* stream);

/* ------------------------------------------------------------------------------------
 * Device-side graph analysis in front of the path (per-edge Python dict / list loops in the reference).
 * Integer work, bit-exact, order-preserving (outputs list elements in input order, as np.where and the reference's append
 * loops do).  ws for the compactions: cb_compact_workspace_bytes(n elements scanned).
 * ---------------------------------------------------------------------------------- */
/* counts[v] = #{e : ids[e] == v}, v in [0, N)  — graph_analyze (utils.py:300-334): out-degree from edge_index[0], in-degree
 * from edge_index[1].  *n_bad = ids outside [0, N) (not counted). */
int cb_id_count_i64(const int64_t* ids, int64_t E, int64_t N, int32_t* counts, int32_t* n_bad, void* stream);
/* hist[b] = #{i : vals[i] == b}, b in [0, n_bins)  — the degree-value histogram from which every repeated median of
 * get_partial_sorted_idx (utils.py:910-941) is read. */
int cb_value_hist_i32(const int32_t* vals, int64_t N, int32_t n_bins, int32_t* hist, int32_t* n_bad, void* stream);
size_t cb_compact_workspace_bytes(int64_t n);
/* out_idx[0 .. *count) = ascending indices i with lo <= vals[i] <= hi; out_mask[i] = 1 for them, 0 otherwise (either output
 * may be NULL)  — np.where(arr <= median) / (arr >= median) of get_partial_sorted_idx and the *_deg_mask vectors of
 * save_graph_analyze (utils.py:694-717). */
int cb_select_range_i32(const int32_t* vals, int64_t N, int32_t lo, int32_t hi, int64_t* out_idx, uint8_t* out_mask, int64_t* count,
                        void* ws, size_t ws_bytes, void* stream);
/* craft_isolation_v2 (utils.py:731-752): keeps edge e, in order, unless src[e] != dst[e] and (node_flag[src[e]] or
 * node_flag[dst[e]]).  out_src / out_dst need room for E entries; *count = edges kept. */
int cb_craft_isolation_i64(const int64_t* src, const int64_t* dst, int64_t E, const uint8_t* node_flag, int64_t N, int64_t* out_src,
                           int64_t* out_dst, int64_t* count, void* ws, size_t ws_bytes, void* stream);
/* ensure_symmetric (utils.py:667-674) / to_undirected as load_ogbn uses it (trainer_node_classification.py:574): the edge set
 * united with its transpose, duplicates removed, sorted by (row, col).  out_row / out_col need room for 2E entries. */
size_t cb_symmetrize_workspace_bytes(int64_t E, int64_t N);
int cb_symmetrize_i64(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t* out_row, int64_t* out_col, int64_t* count,
                      int32_t* n_bad, void* ws, size_t ws_bytes, void* stream);

/* out[i, :] = src[idx[i], :] (contiguous out [n_idx, d]) — packs the rows a peer asked for before the
 * all-to-all of the node-sharded halo exchange (new; the reference is single-device). */
int cb_gather_rows_f32(const float* src, int64_t ld, const int64_t* idx, int64_t n_idx, int64_t d, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Aggregation + the NEXT dense transform in one kernel (csrc/cb_agg_gemm.hip): a block aggregates 64 rows exactly as
 * cb_spmm_csr_f32 / cb_spmm_csr_fused_f32 do (same stores: the aggregated matrix still goes to memory), keeps them in LDS and
 * multiplies the tile by a 256 x 256 matrix B on the matrix cores before anything else is read:
 *     g_out[v, :] = g_rowscale[v] * (out[v, :] @ B) + g_addend[v, :]           (bit-identical to cb_gemm_nn_f32 on `out`)
 * Replaces, per layer of the residual trunk: forward  GCN.py:238-253,127-133 followed by :213,225,230-235 of the next layer
 * (cb_spmm_csr_fused_f32 + cb_gemm_nn_f32); backward autograd of :238 followed by autograd of :213,225
 * (cb_spmm_csr_f32 on the reverse CSR + cb_gemm_nn_f32 with W^T).  d must be 256; fp32 rows, 16-byte aligned.
 * image: B split once into bf16 limbs in MFMA fragment order by cb_agg_gemm_image_f32 (transpose = 1: B = W^T);
 * cb_agg_gemm_image_bytes(256, 256) bytes, 16-byte aligned, L2 resident (384 KB).
 * acc_init (may be NULL; [N, ld_init] fp32, may alias the aggregated output): the reduction of a row starts from these partial sums — the
 * LAST pass of a node-sharded aggregation (cb_spmm_csr_acc_f32 / cb_spmm_csr_fused_acc_f32 + the GEMM), so that a rank's last halo pass
 * also produces the next layer's Z / this layer's dX.
 * A tile hand-over that times out is recorded in the device error word (cb_device_status); these calls return CB_E_DEVICE while it is set.
 * ---------------------------------------------------------------------------------- */
size_t cb_agg_gemm_image_bytes(int64_t K, int64_t N);
int cb_agg_gemm_image_f32(const float* W, int64_t ld, int64_t K, int64_t N, int transpose, void* image, size_t image_bytes, void* stream);
int cb_spmm_gemm_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d,
                     const float* row_scale, const float* bias, int relu, const float* acc_init, int64_t ld_init, float* out, int64_t ld_out,
                     int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws,
                     size_t ws_bytes, const void* image, const float* g_rowscale, const float* g_addend, int64_t ld_add, float* g_out,
                     int64_t ld_gout, void* stream);
int cb_spmm_gemm_fused_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                           int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                           int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                           uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                           int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                           const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* image,
                           const float* g_rowscale, const float* g_addend, int64_t ld_add, float* g_out, int64_t ld_gout, void* stream);
/* cb_spmm_gemm_fused_f32 for a forward that no backward follows (evaluation / metrics passes, GCN.py:100-140 under no_grad): the stored
 * activations have no reader — the next layer's Z leaves this kernel — so the rows the persistent kernel finishes are NOT written to
 * out_next (the hub rows pass through it, its other contents are undefined afterwards; NULL is accepted when n_hubs == 0).  g_out as above. */
int cb_spmm_gemm_fused_eval_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                                int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                                int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                                const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* image,
                                const float* g_rowscale, const float* g_addend, int64_t ld_add, float* g_out, int64_t ld_gout, void* stream);
/* The output Linear as the tail of the LAST layer's aggregation (round 5): GCN.py:133-138 `layers_MLP[-1](F.dropout(x))` reads exactly the rows
 * the last trunk store has just made — logits = out_next @ W_out^T + b_out leave the aggregation kernel (C <= 64 classes; the four multiplying
 * wavefronts of a block take the four 32 x 32 blocks of a tile's 64 x 64 output), the separate head GEMM and its re-read of the [N, 256]
 * activations disappear.  Same limb products in the same order as cb_gemm_nn_f32.  head_image: cb_agg_gemm_head_image_f32 of the nn.Linear
 * weight [C, 256] (transpose = 1) — 256 x 64 with zero columns beyond C; logits [N, ld_logits >= C], 16-byte aligned.  Other arguments as
 * cb_spmm_gemm_fused_f32.  _eval: a forward that no backward follows — the last layer's activations are not written at all. */
size_t cb_agg_gemm_head_image_bytes(int64_t K, int64_t C);
int cb_agg_gemm_head_image_f32(const float* W, int64_t ld, int64_t K, int64_t C, int transpose, void* image, size_t image_bytes, void* stream);
int cb_spmm_gemm_fused_head_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                                int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias, const float* mix_src,
                                int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0,
                                uint64_t* relu_bits, int32_t bits_relu_only, float* out_act, int64_t ld_act, float* out_next, int64_t ld_next,
                                int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr,
                                void* ws, size_t ws_bytes, const void* head_image, const float* head_bias, int64_t C, float* logits,
                                int64_t ld_logits, void* stream);
int cb_spmm_gemm_fused_head_eval_f32(const float* acc_init, int64_t ld_init, const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N,
                                     int64_t E, const float* h, int64_t ld_h, int64_t d, const float* row_scale, const float* bias,
                                     const float* mix_src, int64_t ld_mix, float c_act, float c_mix, float drop_p, uint64_t seed,
                                     const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int32_t bits_relu_only, float* out_act,
                                     int64_t ld_act, float* out_next, int64_t ld_next, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks,
                                     const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, const void* head_image,
                                     const float* head_bias, int64_t C, float* logits, int64_t ld_logits, void* stream);
/* Fault injection for the failure path above (tests): one wavefront waits with a short spin bound for a hand-over that never comes;
 * cb_device_status() must then report CB_E_DEVICE. */
int cb_agg_gemm_handover_selftest(void* stream);

/* ------------------------------------------------------------------------------------
 * The forward front of the residual trunk in one kernel (csrc/cb_front.hip): GNN_model/GCN.py:104-107 (dropout of x, input Linear, ReLU),
 * :110 (dropout in front of layer 0) and :213,225,230-235 (first GCNConv's transform):
 *     X0 = relu(dropout_{seed_x}(x) @ W_in^T + bias_in)                  -> x0 [M, 256] (+ relu_bits [M][4]: mask words of X0 > 0, may be NULL)
 *     Z0 = rowscale * (dropout_{seed_x0}(X0) @ W_0) + addend             -> z0 [M, 256]
 * dropout(X0) never leaves the chip unless x0_drop is given (then it is also stored: the backward's weight gradient can read it instead of
 * regenerating the mask, cb_gemm_tn_adrop_f32).  drop_p = 0: no dropout.  Bit-identical to cb_gemm_nn_indrop_drop2_f32 + cb_gemm_nn_f32.
 * image_in / image_0: cb_front_image_f32 of W_in (nn.Linear layout [256, K], transpose = 1) and of W_0 ([256, 256], transpose = 0);
 * K (input features) in {64, 128} — cb_front_image_bytes(K) == 0 says "no forward-front kernel for this K" — and hidden width 256.
 * ---------------------------------------------------------------------------------- */
size_t cb_front_image_bytes(int64_t K);
int cb_front_image_f32(const float* W, int64_t ld, int64_t K, int transpose, void* image, size_t image_bytes, void* stream);
int cb_trunk_front_f32(const float* x, int64_t ld_x, int64_t M, int64_t K, const void* image_in, const float* bias_in, const void* image_0,
                       const float* rowscale, const float* addend, int64_t ld_add, float* x0, int64_t ld_x0, uint64_t* relu_bits, float* x0_drop,
                       int64_t ld_drop, float* z0, int64_t ld_z, float drop_p, uint64_t seed_x, uint64_t seed_x0, const uint64_t* seed_dev,
                       int64_t row0, void* stream);

/* Edge-weighted aggregation — the `edge_weight` argument of GCNConv.forward (GNN_model/GCN.py:199-202: fn.u_mul_e + fn.sum):
 *     out[v, :] = act( row_scale[v] * sum_{j in row v} w[j] * h[col[j], :] + bias[:] ),   w in CSR order ([E], fp32)
 * and the gradient of the weights, dw[j] = <h[col[j], :], g[row of j, :]>.  The reference asserts len(edge_weight) == E (:200);
 * TricksComb never passes one, so these are plain one-wavefront-per-row kernels (exact, deterministic), not the tuned stream. */
int cb_spmm_csr_weighted_f32(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int64_t E, const float* h, int64_t ld_h,
                             int64_t d, const float* row_scale, const float* bias, int relu, float* out, int64_t ld_out, void* stream);
int cb_spmm_edge_dot_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, const float* h, int64_t ld_h, const float* g,
                         int64_t ld_g, int64_t d, float* dw, void* stream);

/* Dropout of an OPERAND while it is staged (GCN.py:104: `x = F.dropout(x)` in front of layers_MLP[0] of the residual trunk): the input
 * Linear reads the undropped features and applies the keep-mask cb_dropout_f32(x, ..., a_seed, seed_dev, offset = row0 * K) draws as it
 * splits them into limbs — no dropped copy is written, kept or re-read; the weight gradient regenerates the same mask:
 *     C = act(dropout(A) @ B + bias), C2 = dropout_{seed}(C)            cb_gemm_nn_indrop_drop2_f32   (else: cb_dropout_f32 + cb_gemm_nn_drop2_f32)
 *     C = A^T @ dropout(G)                                              cb_gemm_tn_gdrop_f32          (else: cb_dropout_f32 + cb_gemm_tn_f32)
 * Results are bit-identical to the two-kernel forms.  The *_supported queries (1 / 0) say whether the fused form exists for a shape.  * relu_bits (may be NULL; N == 256 and relu only): [M][4] mask words of (C > 0), in the layout of the aggregation's fused store — the input stage
 * of the trunk backward (cb_trunk_input_bwd_multi_f32, act_bits) then reads 32 bytes per row instead of C's 1 KiB. */
int cb_gemm_nn_indrop_supported(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, const float* C2, int64_t ldc2,
                                int64_t M, int64_t N, int64_t K);
int cb_gemm_nn_indrop_drop2_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, float* C2, int64_t ldc2,
                                int64_t M, int64_t N, int64_t K, const float* bias, int relu, float a_drop_p, uint64_t a_seed, float drop_p,
                                uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, void* stream);
/* Single-output forms of the same (no dropped copy of the OUTPUT either): the input Linear writes X0 (+ its mask words) only, and the first
 * GCNConv's transform applies the dropout in front of layer 0 (GCN.py:110) to X0 while IT stages it; its weight gradient regenerates that mask:
 *     C = act(rowscale * (dropout(A) @ B) + addend + bias)             cb_gemm_nn_indrop_f32   (cb_gemm_nn_indrop_supported with C2 = C)
 *     C = dropout(A)^T @ (rowscale * G)                                cb_gemm_tn_adrop_f32    (cb_gemm_tn_adrop_supported)
 * X0's dropped copy (10 GB at the headline size) is then never written, kept or read.  Bit-identical to the forms with the copy. */
int cb_gemm_nn_indrop_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const float* rowscale, const float* addend, int64_t ld_add, const float* bias, int relu, float a_drop_p, uint64_t a_seed,
                          const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, void* stream);
int cb_gemm_tn_adrop_supported(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2);
int cb_gemm_tn_adrop_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* rowscale, float* C, int64_t M, int64_t K1, int64_t K2,
                         float a_drop_p, uint64_t a_seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes, void* stream);
int cb_gemm_tn_gdrop_supported(const float* A, int64_t lda, const float* G, int64_t ldg, int64_t K1, int64_t K2);
int cb_gemm_tn_gdrop_f32(const float* A, int64_t lda, const float* G, int64_t ldg, float* C, int64_t M, int64_t K1, int64_t K2, float g_drop_p,
                         uint64_t g_seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes, void* stream);
/* The input stage of the fused trunk's backward INSIDE the input Linear's weight gradient (round 6; autograd of GCN.py:104-110 — F.dropout(x),
 * layers_MLP[0], F.relu, F.dropout — and of the mixes res_tricks.py:23):
 *     gy = (X0 > 0) * ( dropout_bwd_{g_seed}(g) + mfold )          [M, 256], computed while it is staged as the A operand: never written, never re-read
 *     C  = gy^T @ dropout_{x_seed}(X)                               [256, K2] = layers_MLP[0].weight.grad
 *     colsum = column sums of gy                                    [256]     = layers_MLP[0].bias.grad
 * g = dL/d dropout(X0) (the dX of the first GCNConv), mfold = the folded mix gradients (cb_spmm_csr_store_bwd_mix_f32), x0_bits = [M][4] mask words of
 * (X0 > 0) (cb_gemm_nn_indrop_*'s relu_bits), X = the undropped features.  Replaces cb_trunk_input_bwd_multi_f32 + cb_gemm_tn_gdrop_f32 (2 * 4 * 256 * M bytes
 * less traffic).  64 < K2 <= 128, K2 % 4 == 0, both dropouts active, >= 256 row slabs: cb_gemm_tn_instage_supported first. */
int cb_gemm_tn_instage_supported(const float* g, const float* mfold, const float* X, int64_t ldx, int64_t M, int64_t K2);
size_t cb_gemm_tn_instage_workspace_bytes(int64_t M, int64_t K2);
int cb_gemm_tn_instage_f32(const float* g, const float* mfold, const uint64_t* x0_bits, const float* X, int64_t ldx, float* C, float* colsum, int64_t M, int64_t K2,
                           float g_drop_p, uint64_t g_seed, float x_drop_p, uint64_t x_seed, const uint64_t* seed_dev, int64_t row0, void* ws, size_t ws_bytes,
                           void* stream);

/* cb_spmm_gemm_f32 (reverse aggregation + dX contraction) + the trunk backward of the layer below from the same epilogue: g_out is
 * dL/dx of the stage above layer l-1; gr_out = c_act * dropout_bwd_{seed}(g_out) * relu_bits * rowscale2 (input of the next reverse
 * aggregation) and colsum = the column sums of the same without rowscale2 (bias gradient of layer l-1) — what cb_trunk_layer_bwd_f32
 * computes in a pass of its own (autograd of GCN.py:127-133,250-253), without its 10 GB read of g_out.  relu_bits: the mask words
 * cb_spmm_csr_fused_f32 wrote for layer l-1 ([N][4], d = 256).  ws2: cb_spmm_gemm_trunkbwd_workspace_bytes() (partial column sums).
 * acc_init: as for cb_spmm_gemm_f32 (node-sharded: the last halo pass of the reverse aggregation). */
size_t cb_spmm_gemm_trunkbwd_workspace_bytes(void);
int cb_spmm_gemm_trunkbwd_f32(const int32_t* rowptr, const int32_t* col, int32_t col_flags, int64_t N, int64_t E, const float* h, int64_t ld_h,
                              int64_t d, const float* acc_init, int64_t ld_init, float* out, int64_t ld_out, int32_t hub_threshold,
                              int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows, const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes,
                              const void* image, const float* g_rowscale, float* g_out, int64_t ld_gout, const uint64_t* relu_bits, float c_act,
                              float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, const float* rowscale2, float* gr_out,
                              int64_t ld_gr, float* colsum, void* ws2, size_t ws2_bytes, void* stream);

/* One label-propagation step, elementwise passes folded into the aggregation's store (Label_propagation_model/outcome_correlation.py:137-143
 * with alpha_term and post_step = clamp(0, 1), as trainer_node_classification.py:33-63 drives it):
 *     out[v, :] = post_scale[v] * clamp(row_scale[v] * sum_{u in row v} h[u, :] + c_mix * mix[v, :], 0, 1)      (post_scale NULL: 1) */
int cb_spmm_csr_lp_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, const float* h, int64_t ld_h, int64_t d,
                       const float* row_scale, const float* mix, int64_t ld_mix, float c_mix, const float* post_scale, float* out,
                       int64_t ld_out, int32_t hub_threshold, int32_t n_hubs, int32_t n_chunks, const int32_t* hub_rows,
                       const int32_t* hub_chunk_ptr, void* ws, size_t ws_bytes, void* stream);

/* The same pack with the rows narrowed to bf16 (round-to-nearest-even) as they are written: the send buffer of the bf16 halo wire. */
int cb_gather_rows_bf16_f32(const float* src, int64_t ld, const int64_t* idx, int64_t n_idx, int64_t d, uint16_t* out, void* stream);
/* out[r, :] = pos[r] >= 0 ? src[pos[r], :] : fill for r < n_rows (contiguous [n, d] src and [n_rows, d] out, d % 4 == 0): a matrix over a
 * row subset written back to all rows in one pass — fill = 0: the gradient of a structural-embedding table (dL/dZ_l, GCN.py:230-232) when the level
 * of the row-sparse backward that produces it is compact; fill = NaN: the logits (GCN.py:138) of a rows-only training forward, evaluated on the loss
 * rows of trainer_node_classification.py:390-391 only — a reader of any other row gets NaN, not a plausible number (trunk.py). */
int cb_expand_rows_f32(const float* src, const int32_t* pos, int64_t n_rows, int64_t d, float fill, float* out, void* stream);
/* The trunk's fused store — ReLU, mask words, residual mix, dropout (GCN.py:127-133, res_tricks.py:7-23) — on a SUBSET of the rows, applied to the
 * output of a dense transform instead of inside an aggregation: y / out are compact [n_rows, d] matrices of the rows row_index[0 .. n_rows) (ascending
 * global ids), relu_bits ([N][d/256][4], may be NULL) is the full array, the dropout mask is drawn at the global row; mix_src (may be NULL) is read at row
 * mix_index[r] (mix_index NULL: at row_index[r], i.e. mix_src is a full array too).
 *   act = relu(y[r]) (-> out_act[r] if given: the next 'Residual' layer's mix source);  out[r] = dropout((c_act * act + c_mix * mix_src[mix_index[r]]));
 *   bits as cb_spmm_csr_fused_f32 writes them.
 * The rows-only forward of the training step (trunk.py): the last GCNConv (GCN.py:205-256) evaluated on the loss rows of trainer…:390-391. */
int cb_trunk_store_rows_f32(const float* y, const int64_t* row_index, int64_t n_rows, int64_t d, const float* mix_src, int64_t ld_mix,
                            const int64_t* mix_index, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int bits_relu_only,
                            float* out, float* out_act, void* stream);
/* The same store as the EPILOGUE of the dense transform in front of it:
 *   act = relu(rowscale * (A @ B) + addend + bias) (-> out_act if given);  C = dropout(c_act * act + c_mix * mix_src[mix_index[m] | row_index[m]])
 * with A, C, out_act, rowscale, addend compact over the rows row_index[0 .. M) — a GCNConv evaluated sum-first on a subset of the rows (GCN.py:213-256
 * with the aggregation taken before the transform) and the trunk's ReLU / mix / dropout (:127-133) in one kernel.  N == 256; results equal
 * cb_gemm_nn_f32 followed by cb_trunk_store_rows_f32 bit for bit.  cb_gemm_nn_store_rows_supported tells whether the fused form exists for a shape. */
int cb_gemm_nn_store_rows_supported(const float* A, int64_t lda, const float* B, int64_t ldb, const float* C, int64_t ldc, int64_t M, int64_t N, int64_t K);
int cb_gemm_nn_store_rows_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                              const float* rowscale, const float* addend, int64_t ld_add, const float* bias, const int64_t* row_index,
                              const float* mix_src, int64_t ld_mix, const int64_t* mix_index, float c_act, float c_mix, float drop_p, uint64_t seed,
                              const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits, int bits_relu_only, float* out_act, int64_t ld_act, void* ws,
                              size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COLDBREW_HIP_H */
