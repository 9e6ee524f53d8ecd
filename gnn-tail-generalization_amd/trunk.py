"""Fused residual trunk of TricksComb.forward for the 'Initial' and 'Residual' connections without a bare norm — the
configurations the reference's best-config table selects for Pubmed / ogbn-arxiv / chameleon / squirrel ('Initial…') and WISCONSIN /
CORNELL / TEXAS ('Residual…', base_options.py:416-421) and the benchmark graphs.  One autograd node for

    X0   = relu(Linear_0(dropout(x)))                                   GCN.py:103-107
    for l: X  = dropout(X);  Y = GCNConv_l(X);  A_l = relu(Y);  X = (1-a) A_l + a M_l      GCN.py:109-131
           M_l = X0 ('Initial', res_tricks.py:16-23)  |  M_0 = X0, M_l = A_{l-1} ('Residual', res_tricks.py:7-14: x_list holds the ReLU outputs)
    out  = Linear_1(dropout(X))                                          GCN.py:133-138

with the hand-written backward.  Forward: ONE kernel for the front (dropout(x), input Linear, ReLU, dropout(X0) and layer 0's
transform: cb_trunk_front_f32; hidden 256, 64 / 128 input features), then one kernel per layer — the aggregation with its
ReLU-mask / mix / dropout store AND the next layer's transform (cb_spmm_gemm_fused_f32; the last layer: cb_spmm_csr_fused_f32) —
and the output Linear (in forwards without a backward: the narrow tail of the last aggregation, cb_spmm_gemm_fused_head_f32).  Backward per layer: reverse aggregation + dX contraction in one kernel (cb_spmm_gemm_f32), the weight-gradient
GEMM, one fused elementwise pass (cb_trunk_layer_bwd_f32); ReLU masks are kept as bits, dropout masks are regenerated, the gradient
w.r.t. X0 is gathered in one pass by the input stage.  Other widths, bf16-stored rows and node-sharded pull plans take the same node with
one kernel per stage.  Same arithmetic and the same sequence of dropout seeds as the modular path (ops.py), which stays the general
fallback; every fused form is bit-identical to the kernels it replaces (tests/test_gpu_agg_gemm.py, test_gpu_kernels.py, test_gpu_fullsize.py).
The backward (_Backward) runs on the rows that carry gradient when the caller promised `loss_rows` ("Row-sparse backward"); with the
second promise, `rows_only`, a training forward evaluates its last layers on the rows the loss reads (_rows_only_decision; DESIGN.md section 1,
"Rows-only training forward").
"""
import ctypes
import os

import torch

from . import _lib, gemm, ops
from .tuning import T


def connection(tc):
    """'residual' | 'initial' | None: which mix TricksComb builds for this type_trick (GCN.py:57-60: 'Residual' is tested first)."""
    t = tc.type_trick
    return 'residual' if 'Residual' in t else 'initial' if 'Initial' in t else None


def eligible(tc, x, want_les):
    t = tc.type_trick
    return (tc.has_residual_MLP and connection(tc) is not None and 'Jumping' not in t and not want_les
            and tc.args.type_trick not in ('BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm')
            and tc.dim_hidden % 256 == 0 and x.is_cuda and x.dtype == torch.float32 and len(tc.layers_GCN) == tc.num_layers
            and len(tc.layers_MLP) == 2
            # one dropout rate everywhere (else the modular path, which takes the rates one by one)
            and tc.embedding_dropout == tc.dropout and tc.args.dropout == tc.dropout)


def _exchanged(graph, n_rows, d=256):
    """An [n_rows, d] fp32 matrix the next aggregation of `graph` will exchange: node-sharded, with room behind it for the first halo slice
    (dist.alloc_exchanged: the interior pass and the first halo pass then are one); otherwise a plain matrix."""
    if hasattr(graph, 'part'):
        from .dist import alloc_exchanged
        return alloc_exchanged(graph, n_rows, d)
    return torch.empty((n_rows, d), dtype=torch.float32, device=graph.norm_in.device)


def _fused_spmm(graph, z, bias, x0, c_act, c_mix, p, seed, want_act=False, produce=None, want_bits=True, relu_only=False):
    """(bits, out_next[, act]) of cb_spmm_csr_fused_f32 on the (possibly node-sharded) graph.  Node-sharded + overlapped:
    the exchange runs as the sliced pipeline of dist.ShardedGraph (produce(k, r0, r1), if given, fills rows [r0, r1) of z — the
    row-chunked layer GEMM — right before slice k is packed and sent); the interior-column pass (plain kernel, raw sums) and
    the halo passes of the earlier slices run while the later slices travel, the fused store is the last slice's pass
    (cb_spmm_csr_fused_acc_f32 / _bf16_f32 when the halo rows crossed the links as bf16)."""
    lib = _lib.load()
    sh = graph if hasattr(graph, 'part') else None
    if sh is not None and sh.overlap and z.dtype == torch.float32:
        flights = sh.start_halo(z, False, produce)
        return sh.finish_halo(flights, sh.f, None, lambda g, recv, a_: _fused_launch(lib, graph, g, recv, a_, bias, x0, c_act, c_mix, p, seed, want_act, want_bits, relu_only),
                              x_local=z)
    if produce is not None:
        produce(0, 0, z.shape[0])
    if sh is None:
        return _fused_launch(lib, graph, graph, z, None, bias, x0, c_act, c_mix, p, seed, want_act, want_bits, relu_only)
    return _fused_launch(lib, graph, sh.f.whole, sh.exchange(z, False), None, bias, x0, c_act, c_mix, p, seed, want_act, want_bits, relu_only)


def _fused_gemm_launch(graph, z, bias, x0, c_act, c_mix, p, seed, image, g_rowscale, g_addend, want_bits=True, g=None, acc=None, want_act=False,
                       relu_only=False, head=None):
    """(bits, out_next, z_next[, act]) of cb_spmm_gemm_fused_f32: the fused trunk store of layer l and Z_{l+1} = g_rowscale * (out_next @ W_{l+1})
    + g_addend from one kernel (d = 256, fp32 rows).  g: the CSR to run on (default: the graph itself; node-sharded: the last halo slice,
    z = its receive buffer) with acc = the running sums of the earlier passes.  want_bits=False (a forward that no backward follows):
    cb_spmm_gemm_fused_eval_f32 — no mask words, and out_next is not written either (it has no reader: returned as None).  want_act: a
    fourth result, the ReLU output A_l itself (the next 'Residual' layer's mix source); relu_only: mask words of A_l > 0 alone.
    head = (b_out, C) (the LAST layer; image = graph.head_image(w_out)): the tail is the output Linear — the third result is the logits [N, C]
    (cb_spmm_gemm_fused_head_f32, GCN.py:133-138), g_rowscale / g_addend are ignored."""
    lib = _lib.load()
    g = graph if g is None else g
    if head is not None:
        fn = lib.cb_spmm_gemm_fused_head_f32 if want_bits else lib.cb_spmm_gemm_fused_head_eval_f32
    else:
        fn = lib.cb_spmm_gemm_fused_f32 if want_bits else lib.cb_spmm_gemm_fused_eval_f32
    n, d = g.N, z.shape[1]
    dev = z.device
    bits = torch.empty((n, d // 256, 4), dtype=torch.int64, device=dev) if want_bits else None
    plan = g._plan
    # (the evaluation form keeps the finished rows on chip: X_{l+1} goes to memory only as the hub rows' way into the tile)
    out_next = torch.empty((n, d), dtype=torch.float32, device=dev) if (want_bits or plan.n_hubs > 0) else None
    z_next = _exchanged(graph, n) if head is None else torch.empty((n, int(head[1])), dtype=torch.float32, device=dev)     # (Z_{l+1}: the next aggregation exchanges it | the logits)
    act = torch.empty((n, d), dtype=torch.float32, device=dev) if want_act else None
    wsb = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
    ws = g._workspace(wsb)
    prof = getattr(graph, 'profile', None)
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    col_k = g.flagged_cols(False, d * 4)
    if g_addend is not None and g_addend.stride(1) != 1:
        g_addend = g_addend.contiguous()
    with torch.cuda.device(dev):
        _lib.check(fn(_lib.ptr(acc), acc.stride(0) if acc is not None else 0, _lib.ptr(g.rowptr),
                      _lib.ptr(col_k if col_k is not None else g.col), int(col_k is not None), n, g.E,
                      _lib.ptr(z), z.stride(0), d, _lib.ptr(graph.norm_in), _lib.ptr(bias), _lib.ptr(x0),
                      x0.stride(0) if x0 is not None else 0, float(c_act), float(c_mix), float(p), ctypes.c_uint64(seed),
                      ops.seed_dev_ptr(), int(getattr(graph, 'row_offset', 0)), _lib.ptr(bits), int(bool(relu_only)), _lib.ptr(act), d,
                      _lib.ptr(out_next), d, g.hub_threshold, plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr),
                      _lib.ptr(ws), wsb, _lib.ptr(image),
                      *((_lib.ptr(head[0]), int(head[1])) if head is not None
                        else (_lib.ptr(g_rowscale), _lib.ptr(g_addend), g_addend.stride(0) if g_addend is not None else 0)),
                      _lib.ptr(z_next), z_next.stride(0), _lib.stream_ptr()),
                   'cb_spmm_gemm_fused_f32')
    if prof is not None:
        ev1.record()
        from .graph import prof_rec
        # (the evaluation form writes neither the mask words nor X_{l+1}: 8(d)'s output stream is not there, its rows stay on chip)
        prof.append(prof_rec(ev0, ev1, g, ('agg_gemm_head' if head is not None else 'agg_gemm_fused') + ('' if want_bits else '_eval'),
                             g.algorithmic_bytes(d) - (0 if want_bits else n * d * 4), n * d * 4 + (n * d // 8 if want_bits else 0),
                             (n * int(head[1]) * 4) if head is not None
                             else n * 256 * 4 * (2 if g_addend is not None else 1) + (4 * n if g_rowscale is not None else 0)))
    res = (bits, out_next if want_bits else None, z_next)
    return res + (act,) if want_act else res


def _fused_gemm(graph, z, bias, x0, c_act, c_mix, p, seed, image, g_rowscale, g_addend, want_bits=True, want_act=False, relu_only=False, head=None):
    """_fused_gemm_launch on the (possibly node-sharded) graph: sharded, the exchange of z runs as the sliced pipeline of dist.ShardedGraph
    (pack / push-sum, all-to-all, interior pass, halo passes of the earlier slices) and the LAST halo pass is the fused kernel on top of the
    running sums — the rank's last pass over its rows also yields the next layer's Z, so no GEMM stands between this aggregation and the next
    layer's first send."""
    if not hasattr(graph, 'part'):
        return _fused_gemm_launch(graph, z, bias, x0, c_act, c_mix, p, seed, image, g_rowscale, g_addend, want_bits, want_act=want_act, relu_only=relu_only,
                                  head=head)
    sh = graph
    flights = sh.start_halo(z, False)
    return sh.finish_halo(flights, sh.f, None, lambda g, recv, a_: _fused_gemm_launch(graph, recv, bias, x0, c_act, c_mix, p, seed, image, g_rowscale,
                                                                                       g_addend, want_bits, g=g, acc=a_, want_act=want_act, relu_only=relu_only, head=head),
                          x_local=z)


def head_tail_enabled(bwd):
    """The output Linear as the tail of the last layer's aggregation (cb_spmm_gemm_fused_head_f32).  Default: in forwards that no backward follows
    (the metrics and evaluation forwards: two of the three forwards of the reference's epoch) — there the last layer's activations are not
    written at all and the head's 10 GB re-read disappears (reference epoch 318.2 / 315.9 -> 313.3 / 314.5 ms, A/B on one box).  In the training
    forward the activations must be stored anyway and the persistent kernel moves its bytes slower than the plain aggregation kernel it would
    replace (5.9 against 7.5 TB/s): 160.6 / 160.5 against 159.7 / 160.1 ms per step, so it stays two kernels there.  CB_AGG_GEMM_HEAD=0: never;
    =2: also in the training forward."""
    mode = os.environ.get('CB_AGG_GEMM_HEAD', '1')
    return mode == '2' or (mode == '1' and not bwd)


def agg_gemm_eligible(graph, hidden, agg_bf16):
    """The aggregation + next-dense-transform kernels (cb_agg_gemm.hip): hidden = 256, fp32 rows; one GPU, or node-sharded with the
    overlapped halo exchange on the fp32 wire under a push / pull cover plan or an unsliced pull plan (the last halo pass is then the
    fused kernel on top of the running sums).  CB_AGG_GEMM=0
    keeps the two-kernel form (aggregation, then GEMM)."""
    if os.environ.get('CB_AGG_GEMM', '1') == '0' or hidden != 256 or agg_bf16:
        return False
    if hasattr(graph, 'part'):
        # with the pull plan cut by owner row chunks the row-chunked producers win instead (the GEMM chunks run under the link time of a
        # link-bound exchange: 12.8 vs 11.4 predicted steps/s on the ogbn-products shape at P = 2, profiles/r04_shard_probe_S-products.txt)
        plan = graph.f.plan
        return bool(graph.overlap) and graph.wire == 'f32' and plan is not None and (plan.cover or plan.n_slices == 1)
    return hasattr(graph, 'spmm_gemm')


def _fused_launch(lib, graph, g, z, acc, bias, x0, c_act, c_mix, p, seed, want_act, want_bits=True, relu_only=False, row_ids=None, row_scale=None):
    """One fused-store launch over CSR g (the whole graph, a rank's single-pass block, or the last halo slice on top of acc).
    want_bits=False (forward without a backward: eval / metrics forwards): the backward's mask words are not written.
    row_ids (int32 [g.N], with row_scale = norm_in on those rows): g's rows are a subset of the node rows (cb_spmm_csr_fused_rows_f32) — out_next is
    compact, x0 / the mask words / the dropout mask are taken at the node row (the mask words of the other rows are not written)."""
    n, d = g.N, z.shape[1]
    dev = z.device
    bits = torch.empty((n if row_ids is None else graph.N, d // 256, 4), dtype=torch.int64, device=dev) if want_bits else None
    out_next = torch.empty((n, d), dtype=torch.float32, device=dev)
    act = torch.empty((n, d), dtype=torch.float32, device=dev) if want_act else None
    plan = g._plan
    wsb = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
    ws = g._workspace(wsb)
    prof = getattr(graph, 'profile', None)
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    bf16 = z.dtype == torch.bfloat16
    col_k = g.flagged_cols(False, d * z.element_size()) if hasattr(g, 'flagged_cols') else None
    head = (_lib.ptr(g.rowptr), _lib.ptr(col_k if col_k is not None else g.col), int(col_k is not None))
    args = head + (n, g.E, _lib.ptr(z), z.stride(0), d, _lib.ptr(graph.norm_in if row_scale is None else row_scale), _lib.ptr(bias),
            _lib.ptr(x0), x0.stride(0) if x0 is not None else 0, float(c_act), float(c_mix), float(p), ctypes.c_uint64(seed),
            ops.seed_dev_ptr(), int(getattr(graph, 'row_offset', 0)), _lib.ptr(bits), int(bool(relu_only)), _lib.ptr(act), d, _lib.ptr(out_next), d,
            g.hub_threshold,
            plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), wsb, _lib.stream_ptr())
    with torch.cuda.device(dev):
        if row_ids is not None:
            _lib.check(lib.cb_spmm_csr_fused_rows_f32(_lib.ptr(row_ids), *args), 'cb_spmm_csr_fused_rows_f32')
        elif acc is not None:
            fn = lib.cb_spmm_csr_fused_acc_bf16_f32 if bf16 else lib.cb_spmm_csr_fused_acc_f32
            _lib.check(fn(_lib.ptr(acc), d, *args), 'cb_spmm_csr_fused_acc_f32')
        else:
            fn = lib.cb_spmm_csr_fused_bf16_f32 if bf16 else lib.cb_spmm_csr_fused_f32
            _lib.check(fn(*args), 'cb_spmm_csr_fused_f32')
    if prof is not None:
        ev1.record()
        # SURVEY §8(d) bytes of the aggregation; the fused store's own streams (mixed-in row read + mask bits) are kept apart
        from .graph import prof_rec
        prof.append(prof_rec(ev0, ev1, g, 'fused_store', g.algorithmic_bytes(d, src_elem=2 if bf16 else 4), n * d * 4 + (n * d // 8 if want_bits else 0)))
    return bits, out_next, act


def _spmm_t(graph, gr):
    if hasattr(graph, 'part'):
        return graph.aggregate(gr, True)
    return graph.spmm(gr, transpose=True)


def _layer_bwd_fold(g, bits, row_scale, p, seed, row0, c_act, c_mix, want_colsum, mix_g, mix_pos, mix_seeds, out=None, cs=None):
    """cb_trunk_layer_bwd_fold_f32: layer 0's store backward on all rows that also folds the mix gradients — (b * dY', dbias, m) with
    m = c_mix * (dropout_bwd(g) + sum_q dropout_bwd_q(mix_g[q])); mix_pos[q]: None for a dense operand, else its int32 position map.
    cs = (index into mix_g, mask words, factor): a fourth result — the column sums of factor * dropout_bwd(mix_g[index]) through those mask words (the bias
    gradient of a store whose backward left a reverse aggregation's epilogue)."""
    lib = _lib.load()
    rows, d = g.shape
    if out is None:
        out = torch.empty_like(g)
    m = torch.empty_like(g)
    colsum = torch.empty(d, dtype=torch.float32, device=g.device) if want_colsum else None
    wsb = lib.cb_colsum_workspace_bytes(rows, d) if want_colsum else 0
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    k = len(mix_g)
    colsum2 = torch.empty(d, dtype=torch.float32, device=g.device) if cs is not None else None
    ws2b = lib.cb_colsum_workspace_bytes(rows, d) if cs is not None else 0
    ws2 = torch.empty(max(ws2b, 16), dtype=torch.uint8, device=g.device) if cs is not None else None
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_trunk_layer_bwd_fold_f32(_lib.ptr(g), _lib.ptr(bits), _lib.ptr(row_scale), _lib.ptr(out), rows, d, float(p), ctypes.c_uint64(seed),
                                                   ops.seed_dev_ptr(), int(row0), float(c_act), float(c_mix), k,
                                                   (ctypes.c_void_p * max(k, 1))(*[t.data_ptr() for t in mix_g]),
                                                   (ctypes.c_void_p * max(k, 1))(*[(q.data_ptr() if q is not None else None) for q in mix_pos]),
                                                   (ctypes.c_uint64 * max(k, 1))(*[int(s_) for s_ in mix_seeds]), _lib.ptr(m), _lib.ptr(colsum), _lib.ptr(ws), wsb,
                                                   int(cs[0]) if cs is not None else -1, _lib.ptr(cs[1]) if cs is not None else None,
                                                   float(cs[2]) if cs is not None else 0.0, _lib.ptr(colsum2), _lib.ptr(ws2), ws2b,
                                                   _lib.stream_ptr()), 'cb_trunk_layer_bwd_fold_f32')
    if cs is not None:
        return out, colsum, m, colsum2
    return out, colsum, m


def _layer_bwd(g, bits, row_scale, gx0, accumulate, p, seed, row0, c_act, c_mix, want_colsum, out_bf16=False, out=None, g2=None, seed2=0, c2=0.0, g2_pos=None):
    """cb_trunk_layer_bwd_f32: (b * dY' of the layer's store, dbias).  g2 ('Residual'): the gradient w.r.t. the NEXT layer's stored output, which
    reaches this layer's ReLU output through that layer's mix (c2 = alpha) under that layer's dropout mask (seed2); g2_pos (int32 [rows]): g2 is
    a compact matrix of a row-sparse backward, row r at g2_pos[r] (absent where negative)."""
    lib = _lib.load()
    rows, d = g.shape
    if out is None:
        out = torch.empty(g.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=g.device)
    colsum = torch.empty(d, dtype=torch.float32, device=g.device) if want_colsum else None
    wsb = lib.cb_colsum_workspace_bytes(rows, d) if want_colsum else 0
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_trunk_layer_bwd_f32(_lib.ptr(g), _lib.ptr(bits), _lib.ptr(row_scale), _lib.ptr(out), int(out_bf16),
                                              _lib.ptr(gx0), int(accumulate), rows, d, float(p), ctypes.c_uint64(seed), ops.seed_dev_ptr(),
                                              int(row0), float(c_act), float(c_mix), _lib.ptr(g2), ctypes.c_uint64(seed2), float(c2), _lib.ptr(g2_pos),
                                              _lib.ptr(colsum), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                   'cb_trunk_layer_bwd_f32')
    return out, colsum


def _layer_bwd_rows(g_c, rows_idx, bits, row_scale, p, seed, row0, c_act, want_colsum, out=None, g2=None, seed2=0, c2=0.0, g2_pos=None):
    """_layer_bwd over the compact rows rows_idx of a row-sparse backward (cb_trunk_layer_bwd_rows_f32): (b * dY' of those rows, dbias)."""
    lib = _lib.load()
    n_c, d = g_c.shape
    out = torch.empty_like(g_c) if out is None else out
    colsum = torch.empty(d, dtype=torch.float32, device=g_c.device) if want_colsum else None
    wsb = lib.cb_colsum_workspace_bytes(n_c, d) if want_colsum else 0
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g_c.device)
    with torch.cuda.device(g_c.device):
        _lib.check(lib.cb_trunk_layer_bwd_rows_f32(_lib.ptr(g_c), _lib.ptr(rows_idx), n_c, _lib.ptr(bits), _lib.ptr(row_scale), _lib.ptr(out), d, float(p),
                                                   ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), float(c_act), _lib.ptr(g2), ctypes.c_uint64(seed2),
                                                   float(c2), _lib.ptr(g2_pos), _lib.ptr(colsum), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                   'cb_trunk_layer_bwd_rows_f32')
    return out, colsum


def _input_bwd(g, add, act, p, seed, row0):
    lib = _lib.load()
    rows, d = g.shape
    out = torch.empty_like(g)
    colsum = torch.empty(d, dtype=torch.float32, device=g.device)
    wsb = lib.cb_colsum_workspace_bytes(rows, d)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_trunk_input_bwd_f32(_lib.ptr(g), _lib.ptr(add), _lib.ptr(act), _lib.ptr(out), rows, d, float(p),
                                              ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(colsum), _lib.ptr(ws), wsb,
                                              _lib.stream_ptr()), 'cb_trunk_input_bwd_f32')
    return out, colsum


def _input_bwd_multi(g, seed, g_mix, seeds_mix, c_mix, act, p, row0, act_bits=None, mix_pos=None, cs=None):
    """cb_trunk_input_bwd_multi_f32: (dropout_bwd(g) + c_mix * sum_l dropout_bwd_l(g_mix[l])) * (act > 0) and its column sums.
    act_bits: int64 [rows, d/256, 4] mask words of (act > 0), read instead of act.  mix_pos (list, entries None or int32 [rows]): where set,
    g_mix[l] is a compact matrix of the rows with mix_pos[l] >= 0 (support rows of a row-sparse backward); its other rows are zero.
    cs = list of up to two (index into g_mix, mask words, factor): a third result — per entry the column sums of factor * dropout_bwd(g_mix[index]) through
    those mask words, the bias gradient of a store whose backward left the reverse aggregation's epilogue (cb_trunk_input_bwd_multi_cs_f32)."""
    lib = _lib.load()
    rows, d = g.shape
    out = torch.empty_like(g)
    colsum = torch.empty(d, dtype=torch.float32, device=g.device)
    wsb = lib.cb_colsum_workspace_bytes(rows, d)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    n = len(g_mix)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in g_mix])
    seeds = (ctypes.c_uint64 * max(n, 1))(*[int(s) for s in seeds_mix])
    pos = None
    if mix_pos is not None and any(q is not None for q in mix_pos):
        pos = (ctypes.c_void_p * max(n, 1))(*[(q.data_ptr() if q is not None else None) for q in mix_pos])
    with torch.cuda.device(g.device):
        if cs:
            k = len(cs)
            colsum2 = [torch.empty(d, dtype=torch.float32, device=g.device) for _ in range(k)]
            ws2 = torch.empty(max(k * wsb, 16), dtype=torch.uint8, device=g.device)
            _lib.check(lib.cb_trunk_input_bwd_multi_cs_f32(_lib.ptr(g), ctypes.c_uint64(seed), n, ptrs, seeds, float(c_mix), _lib.ptr(None if act_bits is not None else act),
                                                           _lib.ptr(out), rows, d, float(p), ops.seed_dev_ptr(), int(row0), _lib.ptr(colsum), _lib.ptr(ws), wsb,
                                                           _lib.ptr(act_bits), pos, k, (ctypes.c_int32 * k)(*[int(c[0]) for c in cs]),
                                                           (ctypes.c_void_p * k)(*[c[1].data_ptr() for c in cs]), (ctypes.c_float * k)(*[float(c[2]) for c in cs]),
                                                           (ctypes.c_void_p * k)(*[t.data_ptr() for t in colsum2]), _lib.ptr(ws2), k * wsb, _lib.stream_ptr()),
                       'cb_trunk_input_bwd_multi_cs_f32')
            return out, colsum, colsum2
        _lib.check(lib.cb_trunk_input_bwd_multi_f32(_lib.ptr(g), ctypes.c_uint64(seed), n, ptrs, seeds, float(c_mix), _lib.ptr(None if act_bits is not None else act), _lib.ptr(out),
                                                    rows, d, float(p), ops.seed_dev_ptr(), int(row0), _lib.ptr(colsum), _lib.ptr(ws), wsb,
                                                    _lib.ptr(act_bits), pos, _lib.stream_ptr()), 'cb_trunk_input_bwd_multi_f32')
    return out, colsum


# The trunk backward of the layer below (dropout / mix / ReLU backward, row scale, bias column sums: cb_trunk_layer_bwd_f32's pass over dL/dx)
# in the epilogue of the reverse aggregation + dX kernel (cb_spmm_gemm_trunkbwd_f32): opt-in, CB_AGG_GEMM_TRUNKBWD=1.  Measured neutral on one
# GPU (202.0-203.2 vs 201.8-202.2 ms per step on S-pl10M, profiles/r03_fused_agg_gemm.md) and slower as the last halo pass of a sharded
# aggregation (2.40 vs 1.62 + 0.48 ms at P = 8, profiles/r04_shard_probe_S-pl10M.txt): whatever leaves through the multiplying wavefronts'
# epilogue is paid at their store rate.  Bias gradients are summed in another order than by the pass: equal to rounding, not bit for bit.
def tail_trunk_bwd(graph):
    return os.environ.get('CB_AGG_GEMM_TRUNKBWD', '0') == '1'


# Thresholds of the row-sparse backward and the gather mode: tuning.T (rowsparse_*, fwd0_*, mix_max, gather_mem_frac), each documented there with
# the graph it was tuned on.


_GATHER_OK = {}


def _gather_fits(L, x0, graph=None):
    """Gather mode keeps every layer's [N, d] gradient alive until the input stage: (L - 1) * N * d * 4 bytes more than accumulating
    layer by layer.  Allowed while that stays below a quarter of the device memory that is free at the first backward of
    this shape (decided once per shape: no driver query per step); otherwise the in-place accumulate path.  Node-sharded: ONE decision
    for the group (all-reduce MIN), so that every rank runs the same backward.  CB_TRUNK_GATHER=0/1 forces it."""
    key = (L, tuple(x0.shape), x0.device.index)
    ok = _GATHER_OK.get(key)
    if ok is None:
        env = os.environ.get('CB_TRUNK_GATHER')
        if env in ('0', '1'):
            ok = env == '1'
        else:
            free, _total = torch.cuda.mem_get_info(x0.device)
            ok = (L - 1) * x0.numel() * 4 <= T.gather_mem_frac * free
        if graph is not None and hasattr(graph, 'part') and graph.part.world > 1:
            from .dist import _all_reduce
            import torch.distributed as dist
            flag = torch.tensor([int(ok)], dtype=torch.int64, device=x0.device)
            _all_reduce(flag, op=dist.ReduceOp.MIN, group=graph.group)
            ok = bool(int(flag.item()))
        _GATHER_OK[key] = ok
    return ok


def _chunked(graph, agg_bf16):
    """Row-chunk the producers of the exchanged matrices (layer GEMM forward; dX GEMM + trunk layer backward) so that chunk k
    ships while chunk k+1 is computed: node-sharded overlapped graphs whose plans are cut by owner row chunk into more than one slice
    (the pull-only plan, COLDBREW_HALO_COVER=0, without the aggregation + GEMM kernels)."""
    return (hasattr(graph, 'part') and graph.overlap and not agg_bf16 and graph.f.plan is not None and graph.f.plan.n_slices > 1
            and not graph.f.plan.cover and os.environ.get('COLDBREW_CHUNKED_PRODUCERS', '1') != '0')


def _gather_and_tail(graph, L, residual, h, x0):
    """(gather, tail_tb) of a backward: gather mode of the gradients that reach X0 through the mixes; the trunk backward in the dX kernel's epilogue."""
    gather = residual or (L <= T.mix_max and _gather_fits(L, x0, graph))
    return gather, agg_gemm_eligible(graph, h, False) and gather and not residual and tail_trunk_bwd(graph)


def _support_plan(graph, loss_rows, n_rows, L, residual, h, x0, committed=False):
    """(plan, gather, tail_tb) of a backward — and of a rows-only forward — under the caller's loss_rows promise: ONE decision for both, so that a
    forward that evaluated its last layer on the loss rows finds the same plan in its backward.  plan: CSRGraph.grad_support_plan or None (dense).
    committed (the backward of a rows-only forward): the plan's build / hit bookkeeping (support_plan_pays) is not asked again — the forward's own
    build may just have tipped it."""
    gather, tail_tb = _gather_and_tail(graph, L, residual, h, x0)
    hint = loss_rows if (not hasattr(graph, 'part') and hasattr(graph, 'grad_support_plan') and graph.rowptr_t is not None) else None
    if hint is not None and (not ops.loss_rows_enabled() or hint[0].shape[0] != n_rows):
        hint = None
    if not (hint is not None and 1 <= hint[1] <= T.rowsparse_s0_limit * n_rows
            and (n_rows >= T.rowsparse_min_nodes or getattr(graph, 'rowsparse_small_ok', False)) and gather
            and agg_gemm_eligible(graph, h, False) and not tail_tb and (committed or graph.support_plan_pays())):
        return None, gather, tail_tb
    return hint, gather, tail_tb


def rows_only_enabled():
    """CB_ROWS_ONLY_FWD=0: the training forward evaluates every row of every layer even when the caller reads the loss rows only."""
    return os.environ.get('CB_ROWS_ONLY_FWD', '1') != '0'


def _rows_only_decision(graph, cfg, x, x0, h, ag, bwd, layer_params):
    """What a training forward under both promises of the caller (gradient AND reads in the loss rows only) evaluates on fewer rows:
    (plan, below, sharded, sum_first_below), each None / False where not taken.
      plan    one GPU: the backward's row-support plan (the SAME decision, _support_plan) — the last layer, its store and the output Linear run on the
              loss rows (_last_layer_on_loss_rows), where the backward runs its level 0 through the source rows' side on the saved aggregate (a
              structural-embedding table on that layer: its rows are summed too, the level stays on the compact form);
      below   ... and the layer below it on the rows the last layer reads (CSRGraph.rows_only_fwd: S_1, while the plan keeps that support compact;
              'Residual': its ReLU output — the last layer's mix source — lives on those rows too).  CB_ROWS_ONLY_BELOW=0: that layer on all rows;
      sum_first_below   ... with its sum taken FIRST as well (L >= 3, no table on it, enough edges: tuning.T.sum_first_below_min_edges): its weight
              gradient then contracts the saved aggregate over |S_1| rows (level 1 through the source rows' side), its dX is a GEMM on |S_1| rows
              in front of the plain reverse aggregation, the layer under it loses its dense tail.  CB_ROWS_ONLY_BELOW=1: Z-first on S_1;
      sharded (space, orientation): row shards — the last layer on the rank's loss rows, where its backward runs compact levels; the exchange
              ships only the in-neighbours of those rows (dist.ShardedGraph.loss_rows_forward).  The layers below keep all local rows."""
    L, _alpha, _p, _seeds, agg_bf16, _track, loss_rows, residual, rows_only = cfg
    none = (None, None, None, False)
    le_last, sharded = layer_params[3 * (L - 1) + 2], hasattr(graph, 'part')
    # (bf16-stored rows, one GPU: the last layer alone — its sum is taken over the fp32 activations, the layers below keep their bf16-stored Z)
    bf16_last_only = bool(agg_bf16) and not sharded and agg_gemm_eligible(graph, h, False)
    if not (rows_only and bwd and loss_rows is not None and L >= 2 and (ag or bf16_last_only) and rows_only_enabled()
            and (sharded or x.shape[0] >= T.rows_only_min_nodes)):
        return none
    if sharded:
        if not (hasattr(graph, 'loss_rows_forward') and ops.loss_rows_enabled() and loss_rows[0].shape[0] == x.shape[0]):
            return none
        gather, tail_tb = _gather_and_tail(graph, L, residual, h, x0)
        levels = graph.support_levels(loss_rows[0], L, compact=ag and gather and not tail_tb, cumulative=residual)
        if not (levels and levels[0].src is not None):
            return none
        return None, None, (levels[0].src, graph.loss_rows_forward(levels)), False
    if not T.rowsparse_loss_side:
        return none
    hint = _support_plan(graph, loss_rows, x.shape[0], L, residual, h, x0)[0]
    if hint is None:
        return none
    plan = graph.grad_support_plan(hint[0], L, max_frac=T.rowsparse_max_frac, cumulative=residual)
    mode = '0' if bf16_last_only else os.environ.get('CB_ROWS_ONLY_BELOW', '2')
    below = graph.rows_only_fwd(plan) if mode != '0' else None
    sum_first = (below is not None and mode == '2' and L >= 3 and layer_params[3 * (L - 2) + 2] is None and len(plan.levels) >= 2
                 and below[0].E >= T.sum_first_below_min_edges)
    return plan, below, None, sum_first


def _store_rows(y, idx, mix, c_act, c_mix, p, seed, row0, bits, relu_only, mix_index=None, want_act=False):
    """cb_trunk_store_rows_f32: the trunk's fused store (ReLU, mask words, mix, dropout) on the compact rows idx of a dense transform's output.
    mix_index: the rows of `mix` to read when that is a compact matrix itself (default: the node rows idx).  Returns (stored rows, ReLU output | None)."""
    lib = _lib.load()
    out = torch.empty_like(y)
    act = torch.empty_like(y) if want_act else None
    with torch.cuda.device(y.device):
        _lib.check(lib.cb_trunk_store_rows_f32(_lib.ptr(y), _lib.ptr(idx), y.shape[0], y.shape[1], _lib.ptr(mix), mix.stride(0) if mix is not None else 0,
                                               _lib.ptr(mix_index), float(c_act), float(c_mix), float(p), ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(bits),
                                               int(bool(relu_only)), _lib.ptr(out), _lib.ptr(act), _lib.stream_ptr()), 'cb_trunk_store_rows_f32')
    return out, act


def _layer_on_rows(graph, space, fwd, col_scale, cur, w, b, mix, mix_index, alpha, p, seed, row0, residual, want_act=False, le=None, fwd_le=None):
    """One GCNConv + store on the rows of `space` with the sum taken FIRST: H = (A (a * X))[space] over the forward orientation `fwd` restricted to
    those rows (col_scale = a on fwd's source space), Y = b * (H W) + bias on |space| rows, then the store on those rows.  le (a structural-embedding
    table on all node rows, GCN.py:230-232: Z = a * (X W) + le): its rows are summed the same way, Y = b * (H W + (A le)[space]) + bias, over fwd_le —
    the same orientation with node-row sources.  Returns (mask words [N, d/256, 4] with the rows of `space` written, stored rows, ReLU output | None, H)."""
    fwd.profile = getattr(graph, 'profile', None)
    h_agg = fwd.spmm(cur, col_scale=col_scale)
    b_rows = getattr(space, '_norm_in', None)
    if b_rows is None:
        b_rows = space._norm_in = graph.norm_in[space.idx].contiguous()
    le_sum = None
    if le is not None:
        fwd_le.profile = fwd.profile
        le_sum = fwd_le.spmm(le, row_scale=b_rows)
    bits = torch.empty((graph.N, w.shape[1] // 256, 4), dtype=torch.int64, device=cur.device)
    # the store as the epilogue of the transform where that form exists (hidden 256), else the transform and then the elementwise pass: same values
    fused = gemm.mm_nn_store_rows(h_agg, w, b_rows, le_sum, b, space.idx, mix, mix_index, 1 - alpha, alpha, p, seed, row0, bits, residual, want_act)
    if fused is not None:
        return bits, fused[0], fused[1], h_agg
    y = gemm.mm_nn(h_agg, w, rowscale=b_rows, addend=le_sum, bias=b)
    del le_sum
    x_next, act = _store_rows(y, space.idx, mix, 1 - alpha, alpha, p, seed, row0, bits, residual, mix_index, want_act)
    return bits, x_next, act, h_agg


def _last_layer_on_loss_rows(graph, plan, cur, w, b, mix, alpha, p, seed, row0, residual, w_out, b_out, below=None, le=None):
    """The LAST GCNConv, its store and the output Linear on the loss rows S_0 only (rows-only forward): aggregation and transform commute,
        Y[S_0] = b * ((A (a * X))[S_0] W) + bias        (GCN.py:213-256 with the sum taken first)
    so the layer is one aggregation over the edges that ENTER the loss rows (10 % of the edges under a 10 % mask) into a compact [|S_0|, H] matrix, a
    GEMM on |S_0| rows, the store on those rows and the head on those rows.  Z_{L-1} — the previous layer's dense tail — is never formed.
    Returns (mask words [N, H/256, 4] (rows of S_0 written), dropped X_L on S_0, logits [N, C] with ops.unread_rows_fill() (NaN) outside S_0, H = (A (a * X))[S_0]:
    the operand of the level's weight gradient in the backward's source-side form).  below (CSRGraph.rows_only_fwd): `cur` holds the rows of S_1 only."""
    sp = plan.space0
    fwd0 = graph.loss_rows_fwd(plan) if below is None else below[1]
    mix_index = None
    if below is not None and residual:      # the mix source (the layer below's ReLU output) lives on S_1 too: the loss rows' positions in it
        mix_index = getattr(plan, '_pos0_in_1', None)
        if mix_index is None:
            mix_index = plan._pos0_in_1 = below[4].pos[sp.idx].long().contiguous()
    bits, x_l, _act, h_agg = _layer_on_rows(graph, sp, fwd0, graph.norm_out if below is None else below[4].a, cur, w, b, mix, mix_index, alpha, p, seed, row0,
                                           residual, le=le, fwd_le=graph.loss_rows_fwd(plan) if le is not None else None)
    n = graph.N
    logits_c = gemm.mm_nn(x_l, w_out.t().contiguous(), bias=b_out)
    out = ops.expand_unread(logits_c, sp, n)      # the rows nobody may read: NaN (ops.unread_rows_fill)
    graph.rows_only_forwards = getattr(graph, 'rows_only_forwards', 0) + 1
    return bits, x_l, (out, logits_c), h_agg


def _last_layer_on_loss_rows_sharded(graph, s0, orient, cur, w, b, mix, alpha, p, seed, row0, residual, w_out, b_out, le=None):
    """_last_layer_on_loss_rows on a rank's row block: the aggregation of a * X over the edges that enter the rank's loss rows is a level orientation
    of the forward exchange (dist.ShardedGraph.loss_rows_forward: the peers ship only the in-neighbours of those rows).  Returns (mask words, dropped
    X_L on the rank's loss rows, logits [n_local, C] with ops.unread_rows_fill() (NaN) elsewhere)."""
    lib = _lib.load()
    xs = _exchanged(graph, cur.shape[0], cur.shape[1])
    with torch.cuda.device(cur.device):      # xs = a * X (the source rows' factor, applied before the rows travel)
        _lib.check(lib.cb_act_bwd_f32(_lib.ptr(cur), None, _lib.ptr(graph.norm_out), _lib.ptr(xs), cur.shape[0], cur.shape[1], None, None, 0, _lib.stream_ptr()),
                   'cb_act_bwd_f32')
    h_agg = graph.aggregate_finish(graph.aggregate_start(xs, False, orient=orient), False)
    del xs
    b0 = getattr(s0, '_norm_in', None)
    if b0 is None:
        b0 = s0._norm_in = graph.norm_in[s0.idx].contiguous()
    # (a structural-embedding table on the layer: its rows are summed over the same level orientation — a second, equally small exchange)
    le_sum = graph.aggregate_finish(graph.aggregate_start(le, False, orient=orient), False, row_scale=b0) if le is not None else None
    y = gemm.mm_nn(h_agg, w, rowscale=b0, addend=le_sum, bias=b)
    del h_agg, le_sum
    n, d = graph.N, w.shape[1]
    bits = torch.empty((n, d // 256, 4), dtype=torch.int64, device=cur.device)
    x_l, _act = _store_rows(y, s0.idx, mix, 1 - alpha, alpha, p, seed, row0, bits, residual)
    del y
    logits_c = gemm.mm_nn(x_l, w_out.t().contiguous(), bias=b_out)
    out = ops.expand_unread(logits_c, s0, n)
    graph.rows_only_forwards += 1
    return bits, x_l, out


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, cfg, x, w_in, b_in, w_out, b_out, *layer_params):
        """layer_params = (W_0, bias_0, le_0 | None, W_1, ...).  cfg = (L, alpha, p, seeds, agg_bf16, track, loss_rows, residual): track = autograd
        was recording when the trunk was called (inside a Function's forward it never is, and needs_input_grad does not know about no_grad);
        loss_rows = None or (bool mask [N], count): the caller's promise that the output receives gradient in those rows only (ops.py);
        residual: the 'Residual' connection (mix source of layer l > 0 = the previous layer's ReLU output) instead of 'Initial' (X0);
        rows_only: the caller's second promise — it READS the output in the rows of loss_rows only (the other rows are returned as NaN: ops.unread_rows_fill)."""
        L, alpha, p, seeds, agg_bf16, track, _loss_rows, residual, _rows_only = cfg
        row0 = int(getattr(graph, 'row_offset', 0))
        a = graph.norm_out
        x = x.contiguous()
        bwd = bool(track) and any(ctx.needs_input_grad)          # eval / metrics forwards (no_grad): no mask words, nothing kept
        # the dropout of the input features (GCN.py:104) is applied by the input Linear's GEMM while it stages x (no dropped copy of x
        # is written, kept or re-read: the weight gradient regenerates the mask) where that form exists; CB_TRUNK_INDROP=0 keeps the pass.
        # Round 4: the dropout in front of layer 0 (GCN.py:110) is applied to X0 the same way by layer 0's GEMM, so X0's dropped copy
        # (10 GB at the headline size: one more output stream of the input Linear, one more tensor kept for the backward) does not exist
        # either; `cur` stays None until a path that has no such form asks for the copy (dropped_x0()).
        fused_in = x0_bits = cur = z_front = out_rows = None
        indrop = p > 0 and os.environ.get('CB_TRUNK_INDROP', '1') != '0'
        # Round 4: the whole forward front — dropout(x), input Linear, ReLU, dropout(X0), layer 0's transform — in ONE kernel
        # (cb_trunk_front_f32): a block keeps its 64 rows of dropout(X0) in LDS and multiplies them by W_0 at once.  The dropped copy is
        # also stored when a backward follows (the layer-0 weight gradient reads it: 14.8 + 6.3 ms against 14.4 + 8.7 ms with the mask
        # regenerated while X0 is staged, profiles/r04_front_kernel.md); CB_TRUNK_X0_COPY=0 never materialises it (-10 GB of peak memory).
        if (indrop or p == 0) and not agg_bf16 and L >= 1 and w_in.shape[0] == 256 and os.environ.get('CB_TRUNK_FRONT', '1') != '0':
            w0_, _b0, le0_ = layer_params[0:3]
            want_copy = bwd and p > 0 and os.environ.get('CB_TRUNK_X0_COPY', '1') == '1'
            fr = gemm.trunk_front(x, w_in, b_in, w0_, a, le0_, p, seeds[0], seeds[1], row0, want_bits=bwd, want_drop=want_copy,
                                  z_out=_exchanged(graph, x.shape[0]) if hasattr(graph, 'part') else None)
            if fr is not None:
                x0, x0_bits, cur, z_front = fr
                fused_in = True
                if p == 0:
                    cur = x0
        if fused_in is None and indrop:
            wb = bwd and w_in.shape[0] == 256
            r = None if os.environ.get('CB_TRUNK_X0_COPY', '1') == '1' else gemm.mm_nn_indrop(x, w_in.t().contiguous(), p, seeds[0], row0, bias=b_in,
                                                                                              relu=True, want_bits=wb)
            if r is not None:
                fused_in = True
                x0, x0_bits = r if wb else (r, None)
            else:
                fused_in = gemm.mm_nn_indrop_drop2(x, w_in.t().contiguous(), p, seeds[0], seeds[1], row0, bias=b_in, relu=True,
                                                   want_bits=bwd and w_in.shape[0] == 256)
                if fused_in is not None:
                    x0, cur = fused_in[0], fused_in[1]
                    x0_bits = fused_in[2] if len(fused_in) > 2 else None      # mask words of (X0 > 0): what the input stage of the backward reads instead of X0
        if fused_in is not None:
            xd = x                    # saved for the backward: the UNdropped features
        else:
            xd = ops._dropout_raw(x, p, seeds[0], row0 * x.shape[1]) if p > 0 else x
            if p > 0:    # X0 and its dropped copy leave the same GEMM epilogue (X0 is not re-read by a dropout pass)
                x0, cur = gemm.mm_nn_drop2(xd, w_in.t().contiguous(), p, seeds[1], row0, bias=b_in, relu=True)
            else:
                x0 = cur = gemm.mm_nn(xd, w_in.t().contiguous(), bias=b_in, relu=True)

        def dropped_x0():
            return ops._dropout_raw(x0, p, seeds[1], row0 * x0.shape[1])
        ctx.indrop = fused_in is not None
        h = x0.shape[1]
        saved_in, saved_bits = [cur], []
        ag = agg_gemm_eligible(graph, h, agg_bf16)
        ro_plan, ro_below, ro_sh, agg_first_below = _rows_only_decision(graph, cfg, x, x0, h, ag, bwd, layer_params)
        ro_any = ro_plan is not None or ro_sh is not None
        le_last = layer_params[3 * (L - 1) + 2]
        h_last = None
        h_below = None
        z_ready = None                           # Z_l already produced by layer l-1's aggregation kernel (cb_spmm_gemm_fused_f32)
        out_head = None                          # the logits, when the output Linear left the last layer's aggregation kernel
        mix = x0                                 # mix source of the layer: X0 ('Initial', and layer 0 of 'Residual'), else the previous ReLU output
        for l in range(L):
            w, b, le = layer_params[3 * l: 3 * l + 3]
            sd_l = seeds[l + 2] if p > 0 else 0
            # 'Residual': this layer's ReLU output is the next layer's mix source (stored by the same kernel); its mask words hold the ReLU
            # alone, because the next layer's mix sends a second gradient through it under another dropout mask (cb_trunk_layer_bwd_f32, g2)
            keep_act = residual and l + 1 < L
            z0 = act = None
            if l == 0 and z_front is not None:      # left the forward-front kernel
                z0 = z_front
            elif l == 0 and cur is None:      # layer 0 with no dropped copy of X0: its GEMM draws the mask while it stages X0
                # (l > 0 with cur None: a forward that no backward follows — the activations stayed on chip and Z_l is in z_ready)
                z0 = None if (agg_bf16 or (not ag and _chunked(graph, agg_bf16))) else gemm.mm_nn_indrop(x0, w, p, seeds[1], row0, rowscale=a, addend=le,
                                                                                                         out=_exchanged(graph, x0.shape[0], w.shape[1]))
                if z0 is None:
                    cur = saved_in[0] = dropped_x0()
            if ag and agg_first_below and l == L - 2:
                bits, cur, act, h_below = _layer_on_rows(graph, ro_below[4], ro_below[0], a, cur, w, b, mix, None, alpha, p, sd_l, row0, residual, want_act=keep_act)
                saved_in[L - 2] = None        # X_{L-2}: read by the aggregation above only
                z = None
            elif ag and ro_sh is not None and l == L - 1:
                bits, cur, out_head = _last_layer_on_loss_rows_sharded(graph, ro_sh[0], ro_sh[1], cur, w, b, mix, alpha, p, sd_l, row0, residual, w_out, b_out, le=le)
                z = None
            elif ro_plan is not None and l == L - 1:
                bits, cur, (out_head, out_rows), h_last = _last_layer_on_loss_rows(graph, ro_plan, cur, w, b, mix, alpha, p, sd_l, row0, residual, w_out, b_out,
                                                                                   below=ro_below, le=le_last)
                if le_last is None:
                    saved_in[L - 1] = None    # X_{L-1}: read by the aggregation above only (the level's weight gradient contracts h_last)
                else:
                    h_last = None             # the table's gradient is dL/dZ itself: level 0 stays on the compact form, which reads X_{L-1} on S_1
                z = None
            elif ag:
                from .graph import weight_image
                z = (z_ready if z_ready is not None else z0 if z0 is not None
                     else gemm.mm_nn(cur, w, rowscale=a, addend=le, out=_exchanged(graph, cur.shape[0], w.shape[1])))
                z_ready = None
                if agg_first_below and l + 3 == L:      # the layer under the aggregate-first one: plain fused store, no dense tail
                    bits, cur, act = _fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, want_act=keep_act, want_bits=bwd, relu_only=residual)
                elif ro_plan is not None and l + 2 == L and ro_below is not None:
                    # ... and this layer's outputs are read on S_1 only (the in-neighbours of the loss rows): its aggregation + store on those rows, compact
                    ro_below[0].profile = getattr(graph, 'profile', None)
                    bits, cur, act = _fused_launch(_lib.load(), graph, ro_below[0], z, None, b, mix, 1 - alpha, alpha, p, sd_l, keep_act, want_bits=bwd,
                                                   relu_only=residual, row_ids=ro_below[2], row_scale=ro_below[3])
                elif ro_any and l + 2 == L:     # rows-only: the last layer aggregates X_{L-1} itself — no dense tail under this store
                    bits, cur, act = _fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, want_act=keep_act, want_bits=bwd, relu_only=residual)
                elif l + 1 < L:     # this layer's store + the next layer's transform in one kernel
                    w1, _, le1 = layer_params[3 * (l + 1): 3 * (l + 1) + 3]
                    res = _fused_gemm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, weight_image(w1), a, le1, want_bits=bwd, want_act=keep_act,
                                      relu_only=residual)
                    bits, cur, z_ready = res[:3]
                    act = res[3] if keep_act else None
                else:
                    # the last layer: the output Linear (GCN.py:133-138) is the tail of its aggregation where that form exists (<= 64 classes)
                    from .graph import head_image
                    himg = head_image(w_out) if head_tail_enabled(bwd) else None
                    if himg is not None:
                        bits, cur, out_head = _fused_gemm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, himg, None, None, want_bits=bwd, relu_only=residual,
                                                          head=(b_out, w_out.shape[0]))[:3]
                    else:
                        bits, cur, act = _fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, want_bits=bwd, relu_only=residual)
            elif z0 is None and _chunked(graph, agg_bf16):
                # node-sharded pipeline: row chunk k of Z leaves the GEMM, is packed and put on the links while chunk k+1 multiplies
                z = _exchanged(graph, cur.shape[0], w.shape[1])

                def produce(k, r0, r1, cur=cur, w=w, le=le, z=z):
                    if r1 > r0:
                        gemm.mm_nn(cur[r0:r1], w, rowscale=a[r0:r1], addend=le[r0:r1] if le is not None else None, out=z[r0:r1])
                bits, cur, act = _fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, want_act=keep_act, produce=produce, want_bits=bwd,
                                             relu_only=residual)
            else:
                z = z0 if z0 is not None else gemm.mm_nn(cur, w, rowscale=a, addend=le, out_bf16=agg_bf16,
                                                         out=None if agg_bf16 else _exchanged(graph, cur.shape[0], w.shape[1]))
                bits, cur, act = _fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, sd_l, want_act=keep_act, want_bits=bwd, relu_only=residual)
            del z, z0
            z_front = None
            if residual:
                mix = act
            if bwd:
                saved_bits.append(bits)
                saved_in.append(cur)
        del mix
        out = out_head if out_head is not None else gemm.mm_nn(cur, w_out.t().contiguous(), bias=b_out)
        ctx.graph, ctx.cfg, ctx.row0 = graph, cfg, row0
        ctx.n_layer_params = len(layer_params)
        if bwd:
            ctx.save_for_backward(xd, x0, w_in, w_out, *saved_in, *saved_bits, *[t for t in layer_params if t is not None],
                                  *([h_last] if h_last is not None else []), *([h_below] if h_below is not None else []),
                                  *([x0_bits] if x0_bits is not None else []))
        ctx.has_x0_bits = bwd and x0_bits is not None
        ctx.has_h_below = bwd and h_below is not None
        ctx.rows_only = bwd and h_last is not None       # one GPU: saved_in[L] compact, saved_in[L - 1] absent, h_last saved
        ctx.rows_only_sharded = bwd and (ro_sh is not None or (ro_plan is not None and h_last is None))      # row shards / a table on the last layer: saved_in[L] compact only
        ctx.in_last_compact = bwd and ro_plan is not None and h_last is None and ro_below is not None       # ... and saved_in[L - 1] holds the rows of S_1 (= level 0's destination)
        ctx.le_present = [layer_params[3 * l + 2] is not None for l in range(L)]
        # Second output (rows-only forward on one GPU, else None): the logits of the loss rows as the compact [|S_0|, C] matrix they were computed as — what
        # the reference calls res.emb4classi = emb4classi_full[mask] (GNN_normalizations.py:45-47).  A loss built on it hands its gradient back compact:
        # no [N, C] loss pass, no zero-row check, no row gather in the backward's head.
        ctx.set_materialize_grads(False)
        return out, out_rows

    @staticmethod
    def backward(ctx, gout, gout_rows=None):
        if gout is None and gout_rows is None:
            return (None,) * (7 + ctx.n_layer_params)
        return _Backward(ctx, gout, gout_rows).run()


class _Backward:
    """The hand-written backward of the trunk, one object per call.  run() = head stage, then the layers from the last to the first, then the
    input stage.  What a layer does depends on where its operands live:
      * _layer_source_side / _layer_compact: a level of the row-sparse PLAN (one GPU): compact [|S_j|, .] matrices, the level's own orientation;
      * _layer_fused: the reverse aggregation + dX contraction in one kernel (cb_spmm_gemm_f32) on all rows — one GPU, or as the last halo pass
        of a node-sharded exchange (whose level orientations restrict the exchange to the support's rows: dist.ShardedGraph.support_orients);
      * _layer_plain: aggregation, then the dX GEMM (+ row-chunked producers of the node-sharded pull pipeline; bf16-stored rows).
    All share the same stage helpers (_store_bwd*, _dw, the bookkeeping of the gradients that reach X0)."""

    def __init__(self, ctx, gout, gout_rows=None):
        self.ctx = ctx
        self.gout_rows = gemm._rowmajor(gout_rows) if gout_rows is not None else None      # gradient of the compact loss-row logits (forward's second output)
        graph, (L, alpha, p, seeds, agg_bf16, _track, loss_rows, residual, _rows_only), row0 = ctx.graph, ctx.cfg, ctx.row0
        self.graph, self.L, self.alpha, self.p, self.seeds, self.agg_bf16, self.row0, self.residual = graph, L, alpha, p, seeds, agg_bf16, row0, residual
        sv = list(ctx.saved_tensors)
        self.xd, self.x0, self.w_in, self.w_out = sv[:4]
        self.saved_in = sv[4: 4 + L + 1]
        self.saved_bits = sv[4 + L + 1: 4 + 2 * L + 1]
        rest = sv[4 + 2 * L + 1:]
        self.x0_bits = rest.pop() if ctx.has_x0_bits else None
        self.rows_only = ctx.rows_only                      # the forward ran its last layer on the loss rows: saved_in[L] is compact, saved_in[L - 1] absent
        self.h_below = rest.pop() if ctx.has_h_below else None  # (A (a * X_{L-2}))[S_1]: the layer below the last one took its sum first too
        self.h_last = rest.pop() if ctx.rows_only else None  # (A (a * X_{L-1}))[S_0]
        self.xl_compact = ctx.rows_only or ctx.rows_only_sharded
        self.in_last_compact = ctx.in_last_compact
        self.lp, k = [], 0
        for l in range(L):
            w, b = rest[k], rest[k + 1]
            k += 2
            le = None
            if ctx.le_present[l]:
                le = rest[k]
                k += 1
            self.lp.append((w, b, le))
        self.a, self.bnorm = graph.norm_out, graph.norm_in
        self.need = ctx.needs_input_grad       # (graph, cfg, x, w_in, b_in, w_out, b_out, *layer_params)
        self.gout = gemm._rowmajor(gout) if gout is not None else None
        self.n_rows = self.x0.shape[0]
        self.h = self.x0.shape[1]
        self.sharded = hasattr(graph, 'part')
        if loss_rows is not None and not self.xl_compact and (not ops.loss_rows_enabled() or loss_rows[0].shape[0] != self.x0.shape[0]):
            loss_rows = None
        self.loss_rows = loss_rows
        # the gradient reaching X0 through the mixes: 'Initial' — every layer's, gathered in one pass by the input stage (the per-layer gradients
        # stay alive until then) when L <= mix_max and they fit, else accumulated in place layer by layer; 'Residual' — layer 0's alone
        self.gather = residual or (L <= T.mix_max and _gather_fits(L, self.x0, graph))
        self.gx0 = None if self.gather else torch.empty_like(self.x0)
        self.g_mix, self.seeds_mix, self.mix_pos = [], [], []
        self.grads_layers = [None] * (3 * L)
        self.ag = agg_gemm_eligible(graph, self.h, agg_bf16)
        self.chunked = (self.gather and _chunked(graph, agg_bf16) and graph.b.plan.n_slices > 1 and not self.ag)
        self.tail_tb = self.ag and self.gather and not residual and tail_trunk_bwd(graph)      # (the accumulate-in-place form needs the pass: it also adds into gx0)
        # Round 6 — the input stage without a pass of its own: the reverse aggregation that writes all rows and applies layer 0's store backward in its epilogue
        # (_layer_source_side) FOLDS every mix gradient known by then into one matrix (cb_spmm_csr_store_bwd_mix_f32: it holds layer 0's in registers, the
        # compact ones of the layers above are gathered per row) and takes layer 0's bias gradient; the input Linear's weight gradient then computes
        # gy = (X0 > 0) * (dropout_bwd(dL/d dropout(X0)) + fold) while it stages it (cb_gemm_tn_instage_f32).  gy's [N, 256] write + read and the n + 1
        # operand reads of cb_trunk_input_bwd_multi_f32 disappear (S-pl10M: 20 GB per step).  'Initial', gathered mix gradients, dropout active,
        # features staged undropped, no gradient w.r.t. the features; where no reverse aggregation carries layer 0's store backward (dense levels, mid-size graphs,
        # row shards) that store backward's own pass folds (cb_trunk_layer_bwd_fold_f32, _dx_and_store_bwd).  CB_INSTAGE_FOLD=0: the separate pass.
        self.fold_ok = (os.environ.get('CB_INSTAGE_FOLD', '1') != '0' and self.gather and not residual and not agg_bf16 and p > 0
                        and ctx.indrop and self.need[3] and not self.need[2] and self.x0_bits is not None and self.h == 256
                        and gemm.mm_tn_instage_supported(self.x0, self.xd, self.x0.shape[0]))
        self.mfold = None      # the folded mix gradients, once a level produced them
        self.plan_taken = False      # (the fold belongs to the row-sparse plan: the dense backward keeps the passes whose sums are the bit-for-bit witness of rounds 2 - 5)

    # -- small helpers -------------------------------------------------------------------------------------------------------------------
    def seed(self, i):
        return self.seeds[i] if self.p > 0 else 0

    def need_w(self, l):
        return self.need[7 + 3 * l]

    def need_b(self, l):
        return self.need[7 + 3 * l + 1]

    def need_le(self, l):
        return self.lp[l][2] is not None and self.need[7 + 3 * l + 2]

    def _dw(self, l, x_in, gz):
        """X_l^T (a * dZ_l).  Layer 0 without a dropped copy of X0 (x_in is None): the mask of the dropout in front of layer 0 is
        regenerated from X0 while the GEMM stages it (cb_gemm_tn_adrop_f32)."""
        if x_in is not None:
            return gemm.mm_tn(x_in, gz, rowscale=self.a)
        sd0 = self.seed(1)
        dw = gemm.mm_tn_adrop(self.x0, gz, self.p, sd0, self.row0, rowscale=self.a)
        return dw if dw is not None else gemm.mm_tn(ops._dropout_raw(self.x0, self.p, sd0, self.row0 * self.x0.shape[1]), gz, rowscale=self.a)

    def _second(self, below, g_above, pos_above=None):
        """'Residual': keyword arguments of the second gradient that reaches layer `below`'s ReLU output — through the mix of layer below + 1,
        under that layer's dropout mask.  pos_above: the position map of g_above's row space when it is a compact matrix (row-sparse plan)."""
        if not self.residual or g_above is None:
            return {}
        return dict(g2=g_above, seed2=self.seed(below + 3), c2=self.alpha, **({'g2_pos': pos_above} if pos_above is not None else {}))

    def _dx_and_store_bwd(self, src, wt, rowscale, below, g_ready=None, g_above=None, orient=None, pos_above=None):
        """dL/dx of the stage above layer `below` and that layer's store backward: (g, gr, dbias, handle); handle = the already started
        exchange of gr (row-chunked producers of the node-sharded pull pipeline), else None.  g_ready: dL/dx left the reverse aggregation's kernel."""
        p, alpha, row0, bnorm = self.p, self.alpha, self.row0, self.bnorm
        sd = self.seed(below + 2)
        want_b = self.need_b(below)
        bits = self.saved_bits[below]
        if self.chunked:
            g_ = torch.empty((src.shape[0], wt.shape[1]), dtype=torch.float32, device=src.device)
            gr_ = _exchanged(self.graph, g_.shape[0], g_.shape[1])
            colsums = []
            sec = self._second(below, g_above, pos_above)

            def produce(k, r0, r1):
                if r1 <= r0:
                    return
                gemm.mm_nn(src[r0:r1], wt, rowscale=rowscale[r0:r1] if rowscale is not None else None, out=g_[r0:r1])
                sec_k = dict(sec, g2=sec['g2'][r0:r1]) if sec else {}
                _, cs = _layer_bwd(g_[r0:r1], bits[r0:r1], bnorm[r0:r1], None, False, p, sd, row0 + r0, 1 - alpha, alpha, want_b, out=gr_[r0:r1], **sec_k)
                if want_b:
                    colsums.append(cs)
            h_ = self.graph.aggregate_start(gr_, True, produce=produce, orient=orient)
            db_ = None
            if want_b:      # (a rank that owns no rows produces no chunk: its share of the bias gradient is zero)
                db_ = (torch.zeros(wt.shape[1], dtype=torch.float32, device=src.device) if not colsums
                       else colsums[0] if len(colsums) == 1 else torch.stack(colsums).sum(0))
            return g_, gr_, db_, h_
        g_ = g_ready if g_ready is not None else gemm.mm_nn(src, wt, rowscale=rowscale)
        cs_ = getattr(self, '_cs', [])
        if (below == 0 and self.fold_ok and self.mfold is None and self.plan_taken and len(self.g_mix) <= 2 and len(cs_) <= 1
                and all(t.is_contiguous() for t in self.g_mix) and g_.is_contiguous()):
            # layer 0's store backward as a pass of its own (a level whose reverse aggregation does not carry it): the pass reads layer 0's mix gradient
            # anyway and folds the ones of the layers above into one matrix (cb_trunk_layer_bwd_fold_f32); the input stage is then computed inside the
            # input Linear's weight gradient (see __init__)
            res = _layer_bwd_fold(g_, bits, bnorm, p, sd, row0, 1 - alpha, alpha, want_b, self.g_mix, self.mix_pos, self.seeds_mix,
                                  out=_exchanged(self.graph, g_.shape[0], g_.shape[1]) if self.sharded else None, cs=cs_[0][:3] if cs_ else None)
            gr_, db_, self.mfold = res[:3]
            if cs_:      # (the bias gradient the input-stage pass would have taken for a store whose backward left a reverse aggregation's epilogue)
                self.grads_layers[3 * cs_[0][3] + 1] = res[3]
                self._cs = []
            self.g_mix, self.mix_pos, self.seeds_mix = [], [], []
            return None, gr_, db_, None
        gr_, db_ = _layer_bwd(g_, bits, bnorm, self.gx0, below != self.L - 1, p, sd, row0, 1 - alpha, alpha, want_b, out_bf16=self.agg_bf16,
                              out=_exchanged(self.graph, g_.shape[0], g_.shape[1]) if (self.sharded and not self.agg_bf16) else None,
                              **self._second(below, g_above, pos_above))
        return g_, gr_, db_, None

    # -- head ----------------------------------------------------------------------------------------------------------------------------
    def _head(self, space):
        """Output Linear (GCN.py:133-138) and the store backward of the last layer: sets d_w_out, d_b_out and returns (g, gr, dbias, handle, space).
        space: the compact row space of the loss rows (a plan's space0 / a rank's share of it), or None = all rows."""
        L, need, gout, xl, w_out = self.L, self.need, self.gout, self.saved_in[self.L], self.w_out
        if space is not None:      # loss rows only
            gout_c = ops.gather_rows_by_index(gout, space.idx) if gout is not None else None
            if self.gout_rows is not None:      # (the compact output's gradient arrives on those rows already)
                gout_c = self.gout_rows if gout_c is None else gout_c + self.gout_rows
            xl_c = xl if self.xl_compact else ops.gather_rows_by_index(xl, space.idx)
            self.d_w_out = gemm.mm_tn(gout_c, xl_c) if need[5] else None
            self.d_b_out = ops.act_bwd(gout_c, None, None, want_out=False, want_colsum=True)[1] if need[6] else None
            g = gemm.mm_nn(gout_c, w_out)                                        # dL/d(dropped X_L), loss rows only
            gr, dbias = _layer_bwd_rows(g, space.idx, self.saved_bits[L - 1], self.bnorm, self.p, self.seed(L + 1), self.row0, 1 - self.alpha, self.need_b(L - 1),
                                        out=_exchanged(self.graph, g.shape[0], g.shape[1]) if self.sharded else None)
            return g, gr, dbias, None, space
        self.d_w_out = gemm.mm_tn(gout, xl) if need[5] else None
        self.d_b_out = ops.act_bwd(gout, None, None, want_out=False, want_colsum=True)[1] if need[6] else None
        g, gr, dbias, handle = self._dx_and_store_bwd(gout, w_out, None, L - 1, orient=self._orient_of(L - 1))
        return g, gr, dbias, handle, None

    def _sh_level(self, layer):
        """Node-sharded: the dist.SupportLevel of the reverse aggregation of `layer` (orientation + row spaces), or None."""
        j = self.L - 1 - layer
        return self.sh_levels[j] if 0 <= j < len(self.sh_levels) else None

    def _orient_of(self, layer):
        lv = self._sh_level(layer)
        return lv.orient if lv is not None else None

    def _dw_rows(self, l, x_in, gz, dst):
        """The layer's weight gradient with dL/dZ_l on the rows of `dst` (None: all rows — _dw)."""
        if dst is None:
            return self._dw(l, x_in, gz)
        return gemm.mm_tn(ops.gather_rows_by_index(x_in, dst.idx), gz, rowscale=dst.a)

    # -- the layer forms -----------------------------------------------------------------------------------------------------------------
    def _layer_source_side(self, l, gr, level, fwd_j):
        """A plan level through its SOURCE rows' side (CSRGraph._support_fwd; level 0: the loss rows): a * (A^T dY) W^T = a * A^T (dY W^T) and
        X^T (a * A^T dY) = ((A (a * X))[S_j])^T dY[S_j] — the GEMM and the weight gradient contract over |S_j| rows instead of |S_{j+1}|,
        dL/dZ_l itself is never formed (so not with a table gradient, which IS dL/dZ_l).  Same sums, associated differently."""
        w = self.lp[l][0]
        dst = level[1]
        level[0].profile = getattr(self.graph, 'profile', None)
        if fwd_j is not None:
            fwd_j.profile = level[0].profile
        t = gemm.mm_nn(gr, w.t().contiguous())
        self._fused_store_bwd = None
        # 0: never; 1 (default): levels that write all rows; 2: compact levels too (measured equal on S-pl10M: 119.2 - 119.4 ms per step either way)
        mode = os.environ.get('CB_SPMM_STORE_BWD', '1')
        if (l > 0 and not self.residual and self.gather and len(getattr(self, '_cs', [])) < 2 and (mode == '2' or (mode == '1' and dst is None))
                and hasattr(level[0], 'spmm_store_bwd') and t.shape[1] % 256 == 0):
            # the store backward of the layer below (its dropout / mix / ReLU backward and row factor) leaves this reverse aggregation's own epilogue —
            # the separate pass's read of g disappears; its bias gradient is taken by the input stage, which reads g anyway.  A compact destination:
            # mask words, row factor and dropout mask at the node rows dst.idx
            ids = None
            if dst is not None:
                ids = getattr(dst, '_ids32', None)
                if ids is None:
                    ids = dst._ids32 = dst.idx.to(torch.int32).contiguous()
            fold = (self.fold_ok and l == 1 and dst is None and len(self.g_mix) <= 2 and all(q is not None for q in self.mix_pos)
                    and not getattr(self, '_cs', []))
            if fold:      # (g_mix holds the mix gradients of the layers above, compact on their supports; layer 0's own is this aggregation's sum)
                self.mfold, gr_below, db0 = level[0].spmm_store_bwd(t, self.a, self.saved_bits[0], self.bnorm, 1 - self.alpha, self.p, self.seed(2), self.row0,
                                                                    mix=(self.g_mix, self.mix_pos, self.seeds_mix, self.alpha, self.need_b(0)))
                self.grads_layers[1] = db0
                self.g_mix, self.mix_pos, self.seeds_mix = [], [], []      # folded: the input stage reads self.mfold instead
                g_new = None
            else:
                g_new, gr_below = level[0].spmm_store_bwd(t, dst.a if dst is not None else self.a, self.saved_bits[l - 1], self.bnorm, 1 - self.alpha, self.p,
                                                          self.seed(l + 1), self.row0, row_ids=ids)
            self._fused_store_bwd = (gr_below, l - 1)
        else:
            g_new = level[0].spmm(t, row_scale=dst.a if dst is not None else self.a)
        del t
        if self.need_w(l):
            # (the aggregate the rows-only forward saved, else taken now)
            x_agg = (self.h_last if (self.rows_only and l == self.L - 1) else self.h_below if (self.h_below is not None and l == self.L - 2)
                     else fwd_j.spmm(self.saved_in[l], col_scale=self.a))
            self.grads_layers[3 * l] = gemm.mm_tn(x_agg, gr)
        return None, g_new

    def _layer_compact(self, l, gr, level):
        """A plan level on compact matrices: dL/dZ_l = A (b * dY') over the level's own orientation (rows / columns renumbered to positions in
        S_{j+1} / S_j) and a * (dL/dZ_l @ W_l^T) from the same kernel; the weight gradient over the rows of S_{j+1}."""
        from .graph import weight_image
        w = self.lp[l][0]
        dst = level[1]
        level[0].profile = getattr(self.graph, 'profile', None)
        gz, g_new = level[0].spmm_gemm(gr, weight_image(w, transpose=True), transpose=False, g_rowscale=dst.a if dst is not None else self.a)
        if self.need_w(l):
            if dst is not None:      # dL/dZ_l lives on S_{j+1}: X_l^T (a * dZ_l) over those rows (all others contribute zeros)
                x_rows = self.saved_in[l] if (self.in_last_compact and l == self.L - 1) else ops.gather_rows_by_index(self.saved_in[l], dst.idx)
                self.grads_layers[3 * l] = gemm.mm_tn(x_rows, gz, rowscale=dst.a)
            else:
                self.grads_layers[3 * l] = self._dw(l, self.saved_in[l], gz)
        return gz, g_new

    def _layer_fused(self, l, gr, handle, dst=None):
        """dL/dZ_l = A (b * dY') and a * (dL/dZ_l @ W_l^T) from one kernel (cb_spmm_gemm_f32) on all rows; for l > 0 and tail_tb the trunk backward
        of layer l-1's store leaves the same epilogue (cb_spmm_gemm_trunkbwd_f32).  Node-sharded: the kernel is the LAST halo pass of the reverse
        aggregation, on top of the running sums of the earlier passes; dst (a compact level of the rank, dist.SupportLevel): its rows are the
        positions of the level's destination support.  Returns (gz, g_new, (gr_next, dbias_next) | None)."""
        from .graph import weight_image
        graph = self.graph
        a = dst.a if dst is not None else self.a
        w = self.lp[l][0]
        img = weight_image(w, transpose=True)
        use_tb = l > 0 and self.tail_tb
        sd_b = self.seed(l + 1)

        def tail(csr, src, acc, tr):
            csr.profile = getattr(graph, 'profile', None)
            if use_tb:
                return csr.spmm_gemm_trunkbwd(src, img, a, self.saved_bits[l - 1], 1 - self.alpha, self.p, sd_b, self.row0, self.bnorm, self.need_b(l - 1),
                                              transpose=tr, acc_init=acc)
            return csr.spmm_gemm(src, img, transpose=tr, g_rowscale=a, acc_init=acc)
        res = (graph.aggregate_finish(handle, True, last_pass=lambda csr, recv, acc: tail(csr, recv, acc, False)) if self.sharded
               else tail(graph, gr, None, True))
        if use_tb:
            gz, g_new, gr_n, db_n = res
            return gz, g_new, (gr_n, db_n)
        gz, g_new = res
        return gz, g_new, None

    def _layer_plain(self, gr, handle):
        """dL/dZ_l = A (b * dY') by the plain aggregation (the dX GEMM is a kernel of its own)."""
        return self.graph.aggregate_finish(handle, True) if self.sharded else _spmm_t(self.graph, gr)

    # -- the backward --------------------------------------------------------------------------------------------------------------------
    def run(self):
        ctx, graph, L, alpha, p, need = self.ctx, self.graph, self.L, self.alpha, self.p, self.need
        a, sharded, gather = self.a, self.sharded, self.gather
        gout = self.gout
        # Node-sharded: the row-sparse backward as LEVEL ORIENTATIONS of the reverse exchange (dist.ShardedGraph.support_orients) — level j
        # ships and gathers only the rows of the support S_j; matrices keep all local rows, every other branch is unchanged.
        # Round 5: while a support is a small share of the nodes the level is also COMPACT in the rank's rows (SupportLevel.src / .dst), so the
        # rank's head, store backward, weight gradients and GEMM tails run on the support's rows as on one GPU.
        self.sh_levels = []
        if self.gout_rows is not None and not self.rows_only and not ctx.rows_only_sharded:
            raise RuntimeError('gradient for the compact loss-row logits, but the forward did not run rows-only')
        if sharded and hasattr(graph, 'support_levels') and self.loss_rows is not None:
            ops.check_rows_zero(gout, self.loss_rows[0])
            # ('Residual': CUMULATIVE supports, as on one GPU below)
            self.sh_levels = graph.support_levels(self.loss_rows[0], L, compact=self.ag and gather and not self.tail_tb, cumulative=self.residual)
        # Row-sparse backward (one GPU): when the caller promised that only the loss rows of gout carry gradient, what the backward makes of it
        # stays zero outside the rows those can reach: after the j-th reverse aggregation only the rows with a neighbour in the previous support
        # carry gradient (CSRGraph.grad_support_plan: S_0 = loss rows, S_1, ... — 10 % / 45 % / 94 % of the rows on the bench's graph with its
        # 10 % train mask).  The head, the store backward, the aggregation + dX kernels and the weight gradients of those levels run on compact
        # [|S_j|, .] matrices, and the input stage takes the per-layer gradients as compact operands.  The promise is checked on the device
        # (ops.check_rows_zero: a violation ends in the device error word and stops the optimiser launch, never in silent wrong gradients).
        # Hidden 256, gathered per-layer gradients, loss rows <= rowsparse_s0_limit of the nodes.
        plan = None
        hint = None if sharded else _support_plan(graph, self.loss_rows, self.n_rows, L, self.residual, self.h, self.x0, committed=self.xl_compact)[0]
        # (with bf16-stored rows the compact levels still run on fp32 matrices through the aggregation + GEMM kernel; the dense levels below
        # them go on as the bf16 path does)
        if hint is not None:
            if gout is not None:      # (no gradient for the [N, C] output at all: the loss was built on the compact output — nothing to check)
                ops.check_rows_zero(gout, hint[0])
            # ('Residual': a layer's store backward also takes the gradient of the layer above, so the supports are CUMULATIVE — W_{j+1} = N(W_j) ∪ W_j,
            # a superset of both; every matrix of level j lives on W_j)
            # (count: one use per step — the forward of a rows-only step has looked the plan up already)
            plan = graph.grad_support_plan(hint[0], L, max_frac=T.rowsparse_max_frac, cumulative=self.residual, count=not self.xl_compact)
        self.plan_taken = plan is not None or bool(self.sh_levels)      # (row shards: the level orientations of the reverse exchange)
        if self.rows_only and plan is None:
            raise RuntimeError('the forward evaluated its last layer on the loss rows (rows_only), but its backward finds no row-support plan: '
                               'CB_LOSS_ROWS / tuning.T / the mask changed between the forward and the backward')

        space0 = plan.space0 if plan is not None else (self.sh_levels[0].src if self.sh_levels else None)
        if self.xl_compact and space0 is None:
            raise RuntimeError('the forward evaluated its last layer on the loss rows (rows_only), but its backward has no compact level 0')
        g, gr, dbias, handle, space = self._head(space0)
        g_above = None         # 'Residual': dL/d(stored output) of the layer above the one whose store backward comes next
        deferred = None        # (layer, X_l, dZ_l): weight gradient of the layer above, computed under this layer's halo exchange
        for l in range(L - 1, -1, -1):
            w, b, le = self.lp[l]
            if gather and (not self.residual or l == 0) and self.mfold is None:      # this layer's mix reads X0: its gradient is gathered by the input stage
                self.g_mix.append(g)
                self.mix_pos.append(space.pos if space is not None else None)
                self.seeds_mix.append(self.seed(l + 2))
            j = L - 1 - l
            level = plan.levels[j] if (plan is not None and j < len(plan.levels)) else None
            lv = self._sh_level(l)
            dst = level[1] if level is not None else lv.dst if lv is not None else None
            if sharded and handle is None:
                handle = graph.aggregate_start(gr, True, orient=self._orient_of(l))   # node-sharded: the exchange is in flight from here
            if deferred is not None:
                self.grads_layers[3 * deferred[0]] = self._dw_rows(*deferred)
                deferred = None
            tb_next = None
            fwd_j = plan.fwd[j] if (level is not None and j < len(plan.fwd)) else None
            source_side = (self.rows_only and l == L - 1) or (self.h_below is not None and l == L - 2) or (T.rowsparse_loss_side and fwd_j is not None and not self.need_le(l)
                                                              and not (self.need_w(l) and self.saved_in[l] is None)      # (layer 0 without a stored dropped copy of X0)
                                                              and not (self.in_last_compact and l == L - 1))
            if source_side:
                gz, g_new = self._layer_source_side(l, gr, level, fwd_j)
            elif level is not None:
                gz, g_new = self._layer_compact(l, gr, level)
            elif self.ag:
                gz, g_new, tb_next = self._layer_fused(l, gr, handle, dst)
                if self.need_w(l):
                    if sharded:
                        deferred = (l, self.saved_in[l], gz, dst)
                    else:
                        self.grads_layers[3 * l] = self._dw(l, self.saved_in[l], gz)
            else:
                gz, g_new = self._layer_plain(gr, handle), None
                if self.need_w(l):
                    if sharded:
                        deferred = (l, self.saved_in[l], gz, None)
                    else:
                        self.grads_layers[3 * l] = self._dw(l, self.saved_in[l], gz)
            g_above = g if self.residual else None
            pos_above = space.pos if (self.residual and space is not None) else None      # (g lives on the space of this level's source rows)
            del g, gr
            handle = None
            self.grads_layers[3 * l + 1] = dbias
            space = dst
            if l > 0 and source_side and getattr(self, '_fused_store_bwd', None) is not None:
                g, (gr, _below), dbias = g_new, self._fused_store_bwd, None      # (dbias of layer l - 1: an extra column sum of the input stage)
                if self.mfold is not None:      # (folded: the epilogue took that bias gradient itself)
                    dbias = self.grads_layers[3 * (l - 1) + 1]
                elif self.need_b(l - 1):
                    self._cs = getattr(self, '_cs', []) + [(len(self.g_mix), self.saved_bits[l - 1], 1 - alpha, l - 1)]
                self._fused_store_bwd = None
            elif l > 0 and dst is not None:      # the store backward of layer l-1 on the rows of S_{j+1}
                g = g_new
                gr, dbias = _layer_bwd_rows(g, dst.idx, self.saved_bits[l - 1], self.bnorm, p, self.seed(l + 1), self.row0, 1 - alpha, self.need_b(l - 1),
                                            out=_exchanged(graph, g.shape[0], g.shape[1]) if sharded else None,
                                            **self._second(l - 1, g_above, pos_above))
            elif l > 0 and tb_next is not None:
                g, (gr, dbias) = g_new, tb_next
            elif l > 0:      # dL/d(dropped X_l) and the backward of layer l-1's store
                g, gr, dbias, handle = self._dx_and_store_bwd(gz, w.t().contiguous(), a, l - 1, g_new, g_above, orient=self._orient_of(l - 1),
                                                              pos_above=pos_above)
            else:            # dL/d(dropped X_0): consumed by the input stage
                g = g_new if g_new is not None else gemm.mm_nn(gz, w.t().contiguous(), rowscale=a)
            g_above = None
            if self.need_le(l):
                if dst is not None:      # the table's gradient is dL/dZ_l on ALL rows: the support's rows, zeros elsewhere
                    gz = ops.expand_rows(gz, dst.pos)
                self.grads_layers[3 * l + 2] = gz
            del gz
        if deferred is not None:
            self.grads_layers[3 * deferred[0]] = self._dw_rows(*deferred)
        # input stage: X0 feeds layer 0 (through its dropout) and the mixes
        if self.mfold is not None:      # folded mix gradients (see __init__): the stage is computed inside the input Linear's weight gradient — no pass, no gpre
            d_w_in, d_b_in = gemm.mm_tn_instage(g, self.mfold, self.x0_bits, self.xd, p, self.seed(1), p, self.seeds[0], self.row0)
            del g
            self.gx0 = self.g_mix = self.mfold = None
            graph.instage_folds = getattr(graph, 'instage_folds', 0) + 1      # (tests, bench)
            return (None, None, None, d_w_in, d_b_in if need[4] else None, self.d_w_out, self.d_b_out, *self.grads_layers)
        if gather:
            cs = getattr(self, '_cs', [])
            res = _input_bwd_multi(g, self.seed(1), self.g_mix, self.seeds_mix, alpha, self.x0, p, self.row0, act_bits=self.x0_bits,
                                   mix_pos=self.mix_pos, cs=[c[:3] for c in cs])
            gpre, d_b_in = res[0], res[1]
            for c, db in zip(cs, res[2] if cs else []):
                self.grads_layers[3 * c[3] + 1] = db
        else:
            gpre, d_b_in = _input_bwd(g, self.gx0, self.x0, p, self.seed(1), self.row0)
        del g
        self.gx0 = self.g_mix = None
        d_w_in = None
        xd, row0 = self.xd, self.row0
        if need[3]:
            if ctx.indrop and p > 0:      # xd holds the undropped features: the mask is regenerated while the GEMM stages them
                d_w_in = gemm.mm_tn_gdrop(gpre, xd, p, self.seeds[0], row0)
                if d_w_in is None:
                    d_w_in = gemm.mm_tn(gpre, ops._dropout_raw(xd, p, self.seeds[0], row0 * xd.shape[1]))
            else:
                d_w_in = gemm.mm_tn(gpre, xd)
        d_x = None
        if need[2]:
            d_x = gemm.mm_nn(gpre, self.w_in)
            if p > 0:
                d_x = ops._dropout_raw(d_x, p, self.seeds[0], row0 * d_x.shape[1])
        return (None, None, d_x, d_w_in, d_b_in if need[4] else None, self.d_w_out, self.d_b_out, *self.grads_layers)


def forward(tc, x, graph, loss_rows=None, rows_only=False):
    """TricksComb.forward on the fused trunk; returns (logits, se_reg_all).  loss_rows: None, (bool mask [N], count) or the mask alone — the caller's promise
    that the logits receive gradient in the rows of the mask only (the masked loss, trainer_node_classification.py:390-391).  rows_only (with loss_rows):
    the caller also READS the logits in those rows only — a training forward may then evaluate its last layer on them; the other rows come back as NaN (ops.unread_rows_fill)."""
    L = tc.num_layers
    p = float(tc.dropout) if tc.training else 0.0
    seeds = tuple(ops.next_seed() for _ in range(L + 2)) if p > 0 else (0,) * (L + 2)
    params, se_reg_all = [], None
    for conv in tc.layers_GCN:
        le = conv.le if conv.whetherHasSE else None
        params += [conv.weight, conv.bias, le]
        if le is not None:
            reg = ops.frobenius_norm(le)
            if hasattr(graph, 'part'):
                from .dist import allreduce_sum
                reg = allreduce_sum(reg * reg, graph.group).sqrt()
            conv.se_norm = reg.detach()
            se_reg_all = reg if se_reg_all is None else se_reg_all + reg
    if not all(c._allow_zero_in_degree for c in tc.layers_GCN):      # GCN.py:187-197; set_allow_zero_in_degree(True) lifts it
        graph.check_zero_in_degree()
    agg_bf16 = getattr(tc.args, 'agg_dtype', 'f32') == 'bf16'
    if loss_rows is not None:
        mask, count = loss_rows if isinstance(loss_rows, (tuple, list)) else (loss_rows, None)
        if count is None:      # (a host sync: the trainers hand the count, computed once per run)
            count = int(mask.sum().item())
        if mask.dtype != torch.bool or mask.dim() != 1 or mask.shape[0] != x.shape[0]:
            raise ValueError(f'loss_rows: a bool mask over the {x.shape[0]} rows expected, got {tuple(mask.shape)} {mask.dtype}')
        loss_rows = (mask, int(count))
    out, out_rows = _TrunkFn.apply(graph, (L, float(tc.alpha), p, seeds, agg_bf16, torch.is_grad_enabled(), loss_rows, connection(tc) == 'residual',
                                           bool(rows_only) and loss_rows is not None), x, tc.layers_MLP[0].weight, tc.layers_MLP[0].bias,
                                   tc.layers_MLP[1].weight, tc.layers_MLP[1].bias, *params)
    if out_rows is not None:      # (rows-only forward: the logits of the loss rows as they were computed, == out[mask]; TeacherGNN.get_3_embs hands them on)
        out._cb_rows = (out_rows, loss_rows[0])
    return out, se_reg_all
