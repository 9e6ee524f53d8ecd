"""Fused residual trunk of TricksComb.forward for the 'Initial' connection without a bare norm — the
configuration the reference's best-config table selects for Pubmed / ogbn-arxiv ('InitialBatchNorm',
base_options.py:416) and the benchmark graphs.  One autograd node for

    X0   = relu(Linear_0(dropout(x)))                                   GCN.py:103-107
    for l: X  = dropout(X);  Y = GCNConv_l(X);  A = relu(Y);  X = (1-a) A + a X0      GCN.py:109-131
    out  = Linear_1(dropout(X))                                          GCN.py:133-138

with the hand-written backward.  Per layer the forward is exactly two kernels (MFMA GEMM with
row-scale/+E epilogue, aggregation with ReLU-mask/mix/dropout epilogue) and the backward four
(fused elementwise, reverse aggregation, two GEMMs); ReLU masks are kept as bits, dropout masks are
regenerated, the gradient w.r.t. X0 is accumulated in place.  Same arithmetic and the same sequence
of dropout seeds as the modular path (ops.py), which stays the general fallback.
"""
import ctypes

import torch

from . import _lib, gemm, ops, streams


def eligible(tc, x, want_les):
    t = tc.type_trick
    return (tc.has_residual_MLP and 'Initial' in t and 'Residual' not in t and 'Jumping' not in t and not want_les
            and tc.args.type_trick not in ('BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm')
            and tc.dim_hidden % 256 == 0 and x.is_cuda and x.dtype == torch.float32 and len(tc.layers_GCN) == tc.num_layers
            and len(tc.layers_MLP) == 2
            # one dropout rate everywhere (else the modular path, which takes the rates one by one)
            and tc.embedding_dropout == tc.dropout and tc.args.dropout == tc.dropout)


def _fused_spmm(graph, z, bias, x0, c_act, c_mix, p, seed, want_act=False, produce=None, want_bits=True):
    """(bits, out_next[, act]) of cb_spmm_csr_fused_f32 on the (possibly node-sharded) graph.  Node-sharded + overlapped:
    the exchange runs as the sliced pipeline of dist.ShardedGraph (produce(k, r0, r1), if given, fills rows [r0, r1) of z — the
    row-chunked layer GEMM — right before slice k is packed and sent); the interior-column pass (plain kernel, raw sums) and
    the halo passes of the earlier slices run while the later slices travel, the fused store is the last slice's pass
    (cb_spmm_csr_fused_acc_f32 / _bf16_f32 when the halo rows crossed the links as bf16)."""
    lib = _lib.load()
    sh = graph if hasattr(graph, 'part') else None
    if sh is not None and sh.overlap and z.dtype == torch.float32:
        flights = sh.start_halo(z, False, produce)
        sh.f.interior.profile = getattr(graph, 'profile', None)
        acc = sh.f.interior.spmm(z)
        return sh.finish_halo(flights, sh.f, acc, lambda g, recv, a_: _fused_launch(lib, graph, g, recv, a_, bias, x0, c_act, c_mix, p, seed, want_act, want_bits))
    if produce is not None:
        produce(0, 0, z.shape[0])
    if sh is None:
        return _fused_launch(lib, graph, graph, z, None, bias, x0, c_act, c_mix, p, seed, want_act, want_bits)
    return _fused_launch(lib, graph, sh.f.whole, sh.exchange(z, False), None, bias, x0, c_act, c_mix, p, seed, want_act, want_bits)


def _fused_gemm_launch(graph, z, bias, x0, c_act, c_mix, p, seed, image, g_rowscale, g_addend, want_bits=True):
    """(bits, out_next, z_next) of cb_spmm_gemm_fused_f32: the fused trunk store of layer l and Z_{l+1} = g_rowscale * (out_next @ W_{l+1})
    + g_addend from one kernel (single GPU, d = 256, fp32 rows)."""
    lib = _lib.load()
    g = graph
    n, d = g.N, z.shape[1]
    dev = z.device
    bits = torch.empty((n, d // 256, 4), dtype=torch.int64, device=dev) if want_bits else None
    out_next = torch.empty((n, d), dtype=torch.float32, device=dev)
    z_next = torch.empty((n, 256), dtype=torch.float32, device=dev)
    plan = g._plan
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
        _lib.check(lib.cb_spmm_gemm_fused_f32(_lib.ptr(g.rowptr), _lib.ptr(col_k if col_k is not None else g.col), int(col_k is not None), n, g.E,
                                              _lib.ptr(z), z.stride(0), d, _lib.ptr(graph.norm_in), _lib.ptr(bias), _lib.ptr(x0),
                                              x0.stride(0) if x0 is not None else 0, float(c_act), float(c_mix), float(p), ctypes.c_uint64(seed),
                                              ops.seed_dev_ptr(), 0, _lib.ptr(bits), _lib.ptr(out_next), d, g.hub_threshold, plan.n_hubs,
                                              plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), wsb,
                                              _lib.ptr(image), _lib.ptr(g_rowscale), _lib.ptr(g_addend),
                                              g_addend.stride(0) if g_addend is not None else 0, _lib.ptr(z_next), 256, _lib.stream_ptr()),
                   'cb_spmm_gemm_fused_f32')
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, g.algorithmic_bytes(d), n * d * 4 + n * d // 8,
                     n * 256 * 4 * (2 if g_addend is not None else 1) + (4 * n if g_rowscale is not None else 0)))
    return bits, out_next, z_next


def agg_gemm_eligible(graph, hidden, agg_bf16):
    """The aggregation + next-dense-transform kernels (cb_agg_gemm.hip): one GPU, hidden = 256, fp32 rows.  CB_AGG_GEMM=0 keeps the
    two-kernel form (aggregation, then GEMM)."""
    return (os.environ.get('CB_AGG_GEMM', '1') != '0' and not hasattr(graph, 'part') and hidden == 256 and not agg_bf16
            and hasattr(graph, 'spmm_gemm'))


def _fused_launch(lib, graph, g, z, acc, bias, x0, c_act, c_mix, p, seed, want_act, want_bits=True):
    """One fused-store launch over CSR g (the whole graph, a rank's single-pass block, or the last halo slice on top of acc).
    want_bits=False (forward without a backward: eval / metrics forwards): the backward's mask words are not written."""
    n, d = g.N, z.shape[1]
    dev = z.device
    bits = torch.empty((n, d // 256, 4), dtype=torch.int64, device=dev) if want_bits else None
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
    args = head + (n, g.E, _lib.ptr(z), z.stride(0), d, _lib.ptr(graph.norm_in), _lib.ptr(bias),
            _lib.ptr(x0), x0.stride(0) if x0 is not None else 0, float(c_act), float(c_mix), float(p), ctypes.c_uint64(seed),
            ops.seed_dev_ptr(), int(getattr(graph, 'row_offset', 0)), _lib.ptr(bits), _lib.ptr(act), d, _lib.ptr(out_next), d, g.hub_threshold,
            plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), wsb, _lib.stream_ptr())
    with torch.cuda.device(dev):
        if acc is not None:
            fn = lib.cb_spmm_csr_fused_acc_bf16_f32 if bf16 else lib.cb_spmm_csr_fused_acc_f32
            _lib.check(fn(_lib.ptr(acc), d, *args), 'cb_spmm_csr_fused_acc_f32')
        else:
            fn = lib.cb_spmm_csr_fused_bf16_f32 if bf16 else lib.cb_spmm_csr_fused_f32
            _lib.check(fn(*args), 'cb_spmm_csr_fused_f32')
    if prof is not None:
        ev1.record()
        # SURVEY §8(d) bytes of the aggregation; the fused store's own streams (mixed-in row read + mask bits) are kept apart
        prof.append((ev0, ev1, g.algorithmic_bytes(d, src_elem=2 if bf16 else 4), n * d * 4 + n * d // 8))
    return bits, out_next, act


def _spmm_t(graph, gr):
    if hasattr(graph, 'part'):
        return graph.aggregate(gr, True)
    return graph.spmm(gr, transpose=True)


def _layer_bwd(g, bits, row_scale, gx0, accumulate, p, seed, row0, c_act, c_mix, want_colsum, out_bf16=False, out=None):
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
                                              int(row0), float(c_act), float(c_mix), _lib.ptr(colsum), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                   'cb_trunk_layer_bwd_f32')
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


def _input_bwd_multi(g, seed, g_mix, seeds_mix, c_mix, act, p, row0, premasked=0, act_bits=None):
    """cb_trunk_input_bwd_multi_f32: (dropout_bwd(g) + c_mix * sum_l dropout_bwd_l(g_mix[l])) * (act > 0) and its column sums.
    premasked: bit l set = g_mix[l] already is dropout_bwd_l(.) (stored so by the kernel that produced it).  act_bits: int64 [rows, d/256, 4] mask
    words of (act > 0), read instead of act."""
    lib = _lib.load()
    rows, d = g.shape
    out = torch.empty_like(g)
    colsum = torch.empty(d, dtype=torch.float32, device=g.device)
    wsb = lib.cb_colsum_workspace_bytes(rows, d)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    n = len(g_mix)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in g_mix])
    seeds = (ctypes.c_uint64 * max(n, 1))(*[int(s) for s in seeds_mix])
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_trunk_input_bwd_multi_f32(_lib.ptr(g), ctypes.c_uint64(seed), n, ptrs, seeds, float(c_mix), _lib.ptr(None if act_bits is not None else act), _lib.ptr(out),
                                                    rows, d, float(p), ops.seed_dev_ptr(), int(row0), _lib.ptr(colsum), _lib.ptr(ws), wsb,
                                                    int(premasked), _lib.ptr(act_bits), _lib.stream_ptr()), 'cb_trunk_input_bwd_multi_f32')
    return out, colsum


import os
# cb_gemm_nn_trunkbwd_f32 (the layer-below's trunk backward in the dX GEMM's epilogue) measured 2-3 ms per step SLOWER than the
# GEMM + cb_trunk_layer_bwd_f32 as a pass of its own (215.1-216.3 vs 212.4-214.1 ms on S-pl10M): the extra 10 GB leave through the
# GEMM's store phase, its least efficient part.  Off unless CB_TRUNK_FUSE_BWD=1; both forms are tested.
FUSE_BWD_EPILOGUE = os.environ.get('CB_TRUNK_FUSE_BWD', '0') == '1'
# Store backward of the layer below applied by the reverse aggregation to the rows it gathers (cb_spmm_csr_masked_f32) + its bias
# column sums from the dX GEMM's epilogue (one GPU, fp32 rows).  Measured on S-pl10M: 210.8-213.0 vs 214.7 ms per step (the 3 x 4.0 ms
# passes go, the masked launches cost +1.7 ms each for their 36 bytes of mask words / scale per edge, the epilogue ~1 ms) — but the
# reverse launches then move SURVEY 8(d)'s bytes at 0.78 instead of 0.90 of the roofline, so it stays opt-in: CB_TRUNK_MASKED_GATHER=1.
MASKED_GATHER = os.environ.get('CB_TRUNK_MASKED_GATHER', '0') == '1'
# Three ways of sparing the trunk backward its own [N, d] passes.  All are built, tested bit for bit / to summation order
# (tests/test_gpu_agg_gemm.py, tests/test_gpu_kernels.py) and OFF by default: alternating runs on one box (tools/probes/ab.sh, 3 rounds of 6 steps,
# S-pl10M; box-to-box spread of the step time is ~1.5 %) give
#     all off                                    202.2 / 201.8 / 202.1 ms      (= CB_AGG_GEMM_TRUNKBWD=0 CB_TRUNK_FUSE_OUT_BWD=0: 198.3 / 201.7 / 202.0)
#     CB_AGG_GEMM_TRUNKBWD=1                     = the first line above (it was on in those runs)
#     + CB_TRUNK_FUSE_OUT_BWD=1                  202.0 / 202.7 / 202.4
#     + CB_TRUNK_PREMASKED=1                     203.1 / 203.2 / 202.3
# i.e. nothing outside the noise: the passes that go (3 x 3.9 ms of k_trunk_bwd, 10 GB re-reads) come back as longer epilogues of kernels whose
# multiplying wavefronts are not hidden (profiles/r03_fused_agg_gemm.md), and the input stage is bound by its six 10 GB streams (5.3 TB/s), not by
# the Philox rounds CB_TRUNK_PREMASKED removes (11.66 -> 11.41 ms).
#  * CB_AGG_GEMM_TRUNKBWD=1: trunk backward of the layer below in the epilogue of the reverse aggregation + dX kernel (cb_spmm_gemm_trunkbwd_f32).
#    With the block-barrier form of that kernel it measured slower (207.8 vs 203.9 ms); with the flag hand-over (registers allocated per role) it
#    is neutral.  Bias gradients are summed in another order than by the pass (block partials): equal to rounding, not bit for bit.
#  * CB_TRUNK_FUSE_OUT_BWD=1: the same epilogue on the output Linear's dX GEMM (K = num_classes: store-bound) — cb_gemm_nn_trunkbwd_f32.
#  * CB_TRUNK_PREMASKED=1: those kernels store dL/d(dropped X_{l+1}) as its dropout backward keep * g / (1 - p), the only form it is consumed
#    in, so the input stage draws one mask per row quad instead of one per layer.  Same products in the same order: bit-identical.
TAIL_TRUNK_BWD = os.environ.get('CB_AGG_GEMM_TRUNKBWD', '0') == '1'
FUSE_OUT_BWD = os.environ.get('CB_TRUNK_FUSE_OUT_BWD', '0') == '1'
PREMASKED = os.environ.get('CB_TRUNK_PREMASKED', '0') == '1'
# The input Linear's GEMM writes the mask words of (X0 > 0) from its epilogue (a wavefront holds a whole 256-column row there: four ballots),
# and the input stage of the backward reads those 32 bytes per row instead of X0's 1 KiB (one of its six 10 GB streams).  CB_TRUNK_X0_BITS=0: off.
X0_BITS = os.environ.get('CB_TRUNK_X0_BITS', '1') == '1'
MIX_MAX = 7      # mixed-in gradients one cb_trunk_input_bwd_multi_f32 launch gathers (deeper trunks accumulate layer by layer)


_GATHER_OK = {}


def _gather_fits(L, x0):
    """Gather mode keeps every layer's [N, d] gradient alive until the input stage: (L - 1) * N * d * 4 bytes more than accumulating
    layer by layer.  Allowed while that stays below a quarter of the device memory that is free at the first backward of
    this shape (decided once per shape: no driver query per step); otherwise the in-place accumulate path."""
    key = (L, tuple(x0.shape), x0.device.index)
    ok = _GATHER_OK.get(key)
    if ok is None:
        free, _total = torch.cuda.mem_get_info(x0.device)
        ok = _GATHER_OK[key] = (L - 1) * x0.numel() * 4 <= 0.25 * free or os.environ.get('CB_TRUNK_GATHER') == '1'
        if os.environ.get('CB_TRUNK_GATHER') == '0':
            ok = _GATHER_OK[key] = False
    return ok


def _chunked(graph, agg_bf16):
    """Row-chunk the producers of the exchanged matrices (layer GEMM forward; dX GEMM + trunk layer backward) so that chunk k
    ships while chunk k+1 is computed: node-sharded overlapped graphs whose plans have more than one slice."""
    return (hasattr(graph, 'part') and graph.overlap and not agg_bf16 and graph.f.plan is not None and graph.f.plan.n_slices > 1
            and os.environ.get('COLDBREW_CHUNKED_PRODUCERS', '1') != '0')


OVERLAP_MIN_ROWS = int(os.environ.get('CB_BWD_OVERLAP_MIN_ROWS', 1 << 20))      # smaller graphs: a handful of events costs what the overlap buys


class _Overlap:
    """The backward on two CU-partitioned streams (streams.py).  Everything the chain enqueues goes to `main` (the caller's stream waits
    for both at the end); side(fn, ...) enqueues a weight-gradient GEMM on `sidestream` after what the chain has produced so far."""

    def __init__(self, device):
        self.main, self.sidestream, self.main_cus = streams.partition(device)
        self.device = device
        self.caller = torch.cuda.current_stream(device)
        self._ctx = None

    def __enter__(self):
        ev = torch.cuda.Event()
        ev.record(self.caller)
        self.main.wait_event(ev)
        self.sidestream.wait_event(ev)
        _lib.check(_lib.load().cb_agg_gemm_set_cu_limit(self.main_cus), 'cb_agg_gemm_set_cu_limit')      # persistent kernels: one block per CU of `main`
        self._ctx = torch.cuda.stream(self.main)
        self._ctx.__enter__()
        return self

    def side(self, fn, *produced):
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.sidestream.wait_event(ev)
        for t in produced:
            t.record_stream(self.sidestream)      # (the caching allocator must not hand the block to the chain while the GEMM reads it)
        with torch.cuda.stream(self.sidestream):
            return fn()

    def finish(self, grads):
        """Called inside the `with`: the caller's stream continues after both streams; the gradients are used there."""
        for ev_stream in (self.main, self.sidestream):
            ev = torch.cuda.Event()
            ev.record(ev_stream)
            self.caller.wait_event(ev)
        for t in grads:
            if isinstance(t, torch.Tensor):
                t.record_stream(self.caller)
        return grads

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        _lib.check(_lib.load().cb_agg_gemm_set_cu_limit(0), 'cb_agg_gemm_set_cu_limit')
        return False


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, cfg, x, w_in, b_in, w_out, b_out, *layer_params):
        """layer_params = (W_0, bias_0, le_0 | None, W_1, ...).  cfg = (L, alpha, p, seeds, agg_bf16)."""
        L, alpha, p, seeds, agg_bf16 = cfg
        row0 = int(getattr(graph, 'row_offset', 0))
        a = graph.norm_out
        x = x.contiguous()
        # the dropout of the input features (GCN.py:104) is applied by the input Linear's GEMM while it stages x (no dropped copy of x
        # is written, kept or re-read: the weight gradient regenerates the mask) where that form exists; CB_TRUNK_INDROP=0 keeps the pass
        fused_in = None
        if p > 0 and os.environ.get('CB_TRUNK_INDROP', '1') != '0':
            fused_in = gemm.mm_nn_indrop_drop2(x, w_in.t().contiguous(), p, seeds[0], seeds[1], row0, bias=b_in, relu=True,
                                               want_bits=w_in.shape[0] == 256 and X0_BITS)
        x0_bits = None
        if fused_in is not None:
            x0, cur = fused_in[0], fused_in[1]
            x0_bits = fused_in[2] if len(fused_in) > 2 else None      # mask words of (X0 > 0): what the input stage of the backward reads instead of X0
            xd = x                    # saved for the backward: the UNdropped features
        else:
            xd = ops._dropout_raw(x, p, seeds[0], row0 * x.shape[1]) if p > 0 else x
            if p > 0:    # X0 and its dropped copy leave the same GEMM epilogue (X0 is not re-read by a dropout pass)
                x0, cur = gemm.mm_nn_drop2(xd, w_in.t().contiguous(), p, seeds[1], row0, bias=b_in, relu=True)
            else:
                x0 = cur = gemm.mm_nn(xd, w_in.t().contiguous(), bias=b_in, relu=True)
        ctx.indrop = fused_in is not None
        h = x0.shape[1]
        saved_in, saved_bits = [cur], []
        bwd = any(ctx.needs_input_grad)          # eval / metrics forwards (no_grad): no mask words, nothing kept
        ag = agg_gemm_eligible(graph, h, agg_bf16)
        z_ready = None                           # Z_l already produced by layer l-1's aggregation kernel (cb_spmm_gemm_fused_f32)
        for l in range(L):
            w, b, le = layer_params[3 * l: 3 * l + 3]
            if ag:
                from .graph import weight_image
                z = z_ready if z_ready is not None else gemm.mm_nn(cur, w, rowscale=a, addend=le)
                z_ready = None
                sd_l = seeds[l + 2] if p > 0 else 0
                if l + 1 < L:     # this layer's store + the next layer's transform in one kernel
                    w1, _, le1 = layer_params[3 * (l + 1): 3 * (l + 1) + 3]
                    bits, cur, z_ready = _fused_gemm_launch(graph, z, b, x0, 1 - alpha, alpha, p, sd_l, weight_image(w1), a, le1, want_bits=bwd)
                else:
                    bits, cur, _ = _fused_spmm(graph, z, b, x0, 1 - alpha, alpha, p, sd_l, want_bits=bwd)
            elif _chunked(graph, agg_bf16):
                # node-sharded pipeline: row chunk k of Z leaves the GEMM, is packed and put on the links while chunk k+1 multiplies
                z = torch.empty((cur.shape[0], w.shape[1]), dtype=torch.float32, device=cur.device)

                def produce(k, r0, r1, cur=cur, w=w, le=le, z=z):
                    if r1 > r0:
                        gemm.mm_nn(cur[r0:r1], w, rowscale=a[r0:r1], addend=le[r0:r1] if le is not None else None, out=z[r0:r1])
                bits, cur, _ = _fused_spmm(graph, z, b, x0, 1 - alpha, alpha, p, seeds[l + 2] if p > 0 else 0, produce=produce, want_bits=bwd)
            else:
                z = gemm.mm_nn(cur, w, rowscale=a, addend=le, out_bf16=agg_bf16)
                bits, cur, _ = _fused_spmm(graph, z, b, x0, 1 - alpha, alpha, p, seeds[l + 2] if p > 0 else 0, want_bits=bwd)
            del z
            if bwd:
                saved_bits.append(bits)
                saved_in.append(cur)
        out = gemm.mm_nn(cur, w_out.t().contiguous(), bias=b_out)
        ctx.graph, ctx.cfg, ctx.row0 = graph, cfg, row0
        ctx.n_layer_params = len(layer_params)
        if bwd:
            ctx.save_for_backward(xd, x0, w_in, w_out, *saved_in, *saved_bits, *[t for t in layer_params if t is not None],
                                  *([x0_bits] if x0_bits is not None else []))
        ctx.has_x0_bits = bwd and x0_bits is not None
        ctx.le_present = [layer_params[3 * l + 2] is not None for l in range(L)]
        return out

    @staticmethod
    def backward(ctx, gout):
        """Runs the backward either on the caller's stream, or — one GPU, aggregation + GEMM kernels on, a graph large enough for a few
        events not to matter, no hipGraph capture — with the chain on a stream confined to three quarters of the CUs and the weight-gradient
        GEMMs beside it on the rest (streams.py)."""
        graph, (L, alpha, p, seeds, agg_bf16) = ctx.graph, ctx.cfg
        x0 = ctx.saved_tensors[1]
        ov = None
        if (streams.enabled() and not hasattr(graph, 'part') and ops._graph_seed is None and not torch.cuda.is_current_stream_capturing()
                and agg_gemm_eligible(graph, x0.shape[1], agg_bf16) and x0.shape[0] >= OVERLAP_MIN_ROWS):
            ov = _Overlap(x0.device)
        if ov is None:
            return _TrunkFn._backward_impl(ctx, gout, None)
        with ov:
            return ov.finish(_TrunkFn._backward_impl(ctx, gout, ov))

    @staticmethod
    def _backward_impl(ctx, gout, ov):
        graph, (L, alpha, p, seeds, agg_bf16), row0 = ctx.graph, ctx.cfg, ctx.row0
        sv = list(ctx.saved_tensors)
        xd, x0, w_in, w_out = sv[:4]
        saved_in = sv[4: 4 + L + 1]
        saved_bits = sv[4 + L + 1: 4 + 2 * L + 1]
        rest = sv[4 + 2 * L + 1:]
        x0_bits = rest.pop() if ctx.has_x0_bits else None
        lp, k = [], 0
        for l in range(L):
            w, b = rest[k], rest[k + 1]
            k += 2
            le = None
            if ctx.le_present[l]:
                le = rest[k]
                k += 1
            lp.append((w, b, le))
        a, bnorm = graph.norm_out, graph.norm_in
        need = ctx.needs_input_grad       # (graph, cfg, x, w_in, b_in, w_out, b_out, *layer_params)
        gout = gemm._rowmajor(gout)
        h = x0.shape[1]
        # output Linear (GCN.py:138)
        xl = saved_in[L]
        def side(fn, *produced):
            """A weight-gradient GEMM: beside the chain when the backward runs on CU-partitioned streams (`produced`: operands written by
            the chain so far)."""
            return ov.side(fn, *produced) if ov is not None else fn()

        d_w_out = side(lambda: gemm.mm_tn(gout, xl)) if need[5] else None
        d_b_out = ops.act_bwd(gout, None, None, want_out=False, want_colsum=True)[1] if need[6] else None
        # the gradient reaching X0 through the L mixes: gathered in one pass by the input stage (the per-layer gradients stay
        # alive until then) when L <= MIX_MAX, else accumulated in place layer by layer
        gather = L <= MIX_MAX and _gather_fits(L, x0)
        # opt-in (CB_TRUNK_FUSE_BWD=1): the layer-below's trunk backward leaves the epilogue of the GEMM that produces dL/dx
        fuse = gather and not agg_bf16 and FUSE_BWD_EPILOGUE
        gx0 = None if gather else torch.empty_like(x0)
        g_mix, seeds_mix = [], []
        grads_layers = [None] * (3 * L)
        sharded = hasattr(graph, 'part')

        # opt-in (CB_TRUNK_MASKED_GATHER=1; one GPU, fp32 rows): the store backward of the layer below is applied by the reverse aggregation
        # to the rows it gathers (cb_spmm_csr_masked_f32) and its bias column sums leave the dX GEMM's epilogue — no [N, d] pass of its own
        masked = gather and not agg_bf16 and not sharded and not fuse and MASKED_GATHER and hasattr(graph, 'spmm_masked')
        coef = (1 - alpha) / (1 - p)

        chunked = gather and not fuse and _chunked(graph, agg_bf16) and graph.b.plan.n_slices > 1

        ag_bwd = agg_gemm_eligible(graph, h, agg_bf16) and not masked and not fuse

        def dx_gemm(src, wt, rowscale, below, g_ready=None, fuse=fuse):
            """dL/dx of the stage above layer `below` (+ that layer's trunk backward when fused): (g, gr, dbias, handle); handle =
            the already started exchange of gr (row-chunked producers of the node-sharded pipeline), else None."""
            sd = seeds[below + 2] if p > 0 else 0
            if chunked:
                g_ = torch.empty((src.shape[0], wt.shape[1]), dtype=torch.float32, device=src.device)
                gr_ = torch.empty_like(g_)
                want_b = need[7 + 3 * below + 1]
                colsums = []

                def produce(k, r0, r1):
                    if r1 <= r0:
                        return
                    gemm.mm_nn(src[r0:r1], wt, rowscale=rowscale[r0:r1] if rowscale is not None else None, out=g_[r0:r1])
                    _, cs = _layer_bwd(g_[r0:r1], saved_bits[below][r0:r1], bnorm[r0:r1], None, False, p, sd, row0 + r0, 1 - alpha, alpha,
                                       want_b, out=gr_[r0:r1])
                    if want_b:
                        colsums.append(cs)
                h_ = graph.aggregate_start(gr_, True, produce=produce)
                db_ = None
                if want_b:
                    db_ = colsums[0] if len(colsums) == 1 else torch.stack(colsums).sum(0)
                return g_, gr_, db_, h_
            if masked:
                if need[7 + 3 * below + 1]:
                    g_, _, db_ = gemm.mm_nn_trunkbwd(src, wt, rowscale, saved_bits[below], coef, 0.0, 0, row0, None, True, want_gr=False)
                else:
                    g_, db_ = gemm.mm_nn(src, wt, rowscale=rowscale), None
                return g_, None, db_, None
            if fuse:      # (gather mode: g is consumed by the input stage only, as its dropout backward — stored in that form, PREMASKED)
                return gemm.mm_nn_trunkbwd(src, wt, rowscale, saved_bits[below], 1 - alpha, p, sd, row0, bnorm, need[7 + 3 * below + 1],
                                           g_masked=PREMASKED) + (None,)
            g_ = g_ready if g_ready is not None else gemm.mm_nn(src, wt, rowscale=rowscale)     # g_ready: left the reverse aggregation's kernel
            gr_, db_ = _layer_bwd(g_, saved_bits[below], bnorm, gx0, below != L - 1, p, sd, row0, 1 - alpha, alpha,
                                  need[7 + 3 * below + 1], out_bf16=agg_bf16)
            return g_, gr_, db_, None

        # dL/d(dropped X_L) and the backward of layer L-1's store.  The output Linear's dX GEMM has K = C (40): it is bound by its 10 GB store, so
        # the trunk backward leaves its epilogue (no re-read of the matrix just written) where the later, MFMA-bound dX GEMMs keep the pass
        fuse_out = FUSE_OUT_BWD and gather and not agg_bf16 and not sharded and not masked and not chunked and h % 256 == 0
        g, gr, dbias, handle = dx_gemm(gout, w_out, None, L - 1, fuse=fuse or fuse_out)
        g_pm = PREMASKED and (fuse or fuse_out)      # g already is dropout_bwd(g): no mask drawn for it by the input stage
        premasked = 0
        deferred = None        # (layer, X_l, dZ_l): weight gradient of the layer above, computed under this layer's halo exchange
        for l in range(L - 1, -1, -1):
            w, b, le = lp[l]
            if gather:
                premasked |= int(bool(g_pm)) << len(g_mix)
                g_mix.append(g)
                seeds_mix.append(seeds[l + 2] if p > 0 else 0)
            if sharded and handle is None:
                handle = graph.aggregate_start(gr, True)                        # node-sharded: the exchange is in flight from here
            if deferred is not None:
                grads_layers[3 * deferred[0]] = gemm.mm_tn(deferred[1], deferred[2], rowscale=a)
                deferred = None
            g_fused = None
            if masked:
                gz = graph.spmm_masked(g, saved_bits[l], bnorm, coef)                         # dL/dZ_l = A (b * dY'), dY' formed on the fly
            elif ag_bwd:
                # dL/dZ_l = A (b * dY') and a * (dL/dZ_l @ W_l^T) from one kernel (cb_spmm_gemm_f32); for l > 0 the trunk backward of layer
                # l-1's store leaves the same epilogue (cb_spmm_gemm_trunkbwd_f32: no pass of its own over dL/dx_l)
                from .graph import weight_image
                tb_fused = None
                if l > 0 and TAIL_TRUNK_BWD and gather:      # (the accumulate-in-place form needs the pass: it also adds into gx0)
                    gz, g_fused, gr_n, db_n = graph.spmm_gemm_trunkbwd(gr, weight_image(w, transpose=True), a, saved_bits[l - 1], 1 - alpha, p,
                                                                         seeds[l + 1] if p > 0 else 0, row0, bnorm, need[7 + 3 * (l - 1) + 1],
                                                                         g_masked=PREMASKED and gather)
                    tb_fused = (gr_n, db_n)
                else:
                    gz, g_fused = graph.spmm_gemm(gr, weight_image(w, transpose=True), transpose=True, g_rowscale=a)
            else:
                gz = graph.aggregate_finish(handle, True) if sharded else _spmm_t(graph, gr)  # dL/dZ_l = A (b * dY')
            del g, gr
            handle = None
            if need[7 + 3 * l]:
                if sharded:
                    deferred = (l, saved_in[l], gz)
                else:
                    grads_layers[3 * l] = side(lambda gz=gz, l=l: gemm.mm_tn(saved_in[l], gz, rowscale=a), gz)
            grads_layers[3 * l + 1] = dbias
            if l > 0 and ag_bwd and tb_fused is not None:
                g, (gr, dbias), handle = g_fused, tb_fused, None
                g_pm = PREMASKED and gather
            elif l > 0:
                g, gr, dbias, handle = dx_gemm(gz, w.t().contiguous(), a, l - 1, g_fused)   # dL/d(dropped X_l) and the backward of layer l-1's store
                g_pm = PREMASKED and fuse
            else:
                g = g_fused if g_fused is not None else gemm.mm_nn(gz, w.t().contiguous(), rowscale=a)   # dL/d(dropped X_0): consumed by the input stage
            if le is not None and need[7 + 3 * l + 2]:
                grads_layers[3 * l + 2] = gz
            else:
                del gz
        if deferred is not None:
            grads_layers[3 * deferred[0]] = gemm.mm_tn(deferred[1], deferred[2], rowscale=a)
            deferred = None
        # input stage: X0 feeds layer 0 (through its dropout) and every mix
        if gather:
            gpre, d_b_in = _input_bwd_multi(g, seeds[1] if p > 0 else 0, g_mix, seeds_mix, alpha, x0, p, row0, premasked, act_bits=x0_bits)
        else:
            gpre, d_b_in = _input_bwd(g, gx0, x0, p, seeds[1] if p > 0 else 0, row0)
        del g, gx0, g_mix
        d_w_in = None
        if need[3]:
            if ctx.indrop:      # xd holds the undropped features: the mask is regenerated while the GEMM stages them
                d_w_in = gemm.mm_tn_gdrop(gpre, xd, p, seeds[0], row0)
                if d_w_in is None:
                    d_w_in = gemm.mm_tn(gpre, ops._dropout_raw(xd, p, seeds[0], row0 * xd.shape[1]))
            else:
                d_w_in = gemm.mm_tn(gpre, xd)
        d_x = None
        if need[2]:
            d_x = gemm.mm_nn(gpre, w_in)
            if p > 0:
                d_x = ops._dropout_raw(d_x, p, seeds[0], row0 * d_x.shape[1])
        return (None, None, d_x, d_w_in, d_b_in if need[4] else None, d_w_out, d_b_out, *grads_layers)


def forward(tc, x, graph):
    """TricksComb.forward on the fused trunk; returns (logits, se_reg_all)."""
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
    out = _TrunkFn.apply(graph, (L, float(tc.alpha), p, seeds, agg_bf16), x, tc.layers_MLP[0].weight, tc.layers_MLP[0].bias,
                         tc.layers_MLP[1].weight, tc.layers_MLP[1].bias, *params)
    return out, se_reg_all
