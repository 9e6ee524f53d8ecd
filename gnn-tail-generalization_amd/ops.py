"""Autograd-visible operators of the TeacherGNN hot path, each backed by the C ABI
(include/coldbrew_hip.h).  One torch.autograd.Function per fused stage; tensors must live
on the MI355X (no CPU fallback — _lib.require_device raises otherwise).

Stage map (reference lines are GNN_model/GCN.py unless noted):
  dropout     Xd = X * keep / (1-p)                  :104,:110,:133  (counter-based RNG, mask never stored)
  transform   Z  = (Xd * a) @ W + E ; ||E||_F        :213,:225,:230-232  (dense, MFMA-bound)
  aggregate   Y  = act(b * (A^T Z) + bias)           :238,:250,:253(+:128)  (sparse, HBM-bound)
  res_mix     X' = (1-alpha) * Y + alpha * R         res_tricks.py:14,23
  nll_logsoftmax  mean nll(log_softmax(out[mask]))   trainer_node_classification.py:390-391
"""
import ctypes
import os

import torch

from . import _lib


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# dropout
# ---------------------------------------------------------------------------------------------
_seed_override = []
_graph_seed = None      # hipGraph mode: int64 device tensor [1]; kernels add its value to the per-site host seed


_graph_site = 0         # hipGraph mode: running number of the dropout site whose host seed is being fixed


def next_seed():
    """63-bit seed drawn from torch's CPU generator: follows torch.manual_seed, costs no device sync.  In hipGraph mode the per-site
    host seeds are baked into the captured kernels and all variation comes from the device seed word, so they are taken from a fixed
    sequence instead: a resumed run then captures the very same graphs as the run that wrote the checkpoint."""
    global _graph_site
    if _seed_override:
        return _seed_override.pop(0)
    if _graph_seed is not None:
        _graph_site += 1
        return (0x9E3779B97F4A7C15 * _graph_site) % (2 ** 62)
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def set_graph_seed(t):
    """Install (or clear with None) the device-resident per-step seed word used while a training step is
    captured / replayed as a hipGraph: the captured kernels keep their per-site host seeds and add *t."""
    global _graph_seed, _graph_site
    _graph_seed = t
    _graph_site = 0


def seed_dev_ptr():
    return _lib.ptr(_graph_seed) if _graph_seed is not None else None


def _dropout_raw(x, p, seed, offset=0):
    lib = _lib.load()
    x = _c(x)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.cb_dropout_f32(_lib.ptr(x), _lib.ptr(out), x.numel(), float(p), ctypes.c_uint64(seed), seed_dev_ptr(),
                                      int(offset), _lib.stream_ptr()), 'cb_dropout_f32')
    return out


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        return _dropout_raw(x, p, seed, offset)

    @staticmethod
    def backward(ctx, g):
        return _dropout_raw(g, ctx.p, ctx.seed, ctx.offset), None, None, None


def dropout(x, p, training=True, seed=None, offset=0):
    """F.dropout semantics; the keep-mask is a pure function of (seed, flat element index)."""
    if not training or p == 0.0:
        return x
    _lib.require_device(x)
    if x.dtype != torch.float32:
        raise TypeError('dropout expects float32')
    if p >= 1.0:
        return x * 0
    if seed is None:
        seed = next_seed()
    return _DropoutFn.apply(x, float(p), int(seed), int(offset))


def dropout_keep_mask(shape, p, seed, device, offset=0):
    """The boolean keep-mask `dropout` uses for (seed, shape) — for parity tests that inject the same
    mask into the oracle."""
    ones = torch.ones(shape, dtype=torch.float32, device=device)
    return _dropout_raw(ones, p, seed, offset) > 0


# ---------------------------------------------------------------------------------------------
# aggregation
# ---------------------------------------------------------------------------------------------
def act_bwd(g, act=None, row_scale=None, want_out=True, want_colsum=False):
    """gm = g * (act > 0); returns (gm * row_scale[:,None] or None, colsum(gm) or None) in one pass."""
    lib = _lib.load()
    g = _c(g)
    rows, d = g.shape
    if act is not None:
        act = _c(act)
    out = torch.empty_like(g) if want_out else None
    colsum = torch.empty(d, dtype=torch.float32, device=g.device) if want_colsum else None
    wsb = lib.cb_colsum_workspace_bytes(rows, d) if want_colsum else 0
    ws = _ws(wsb, g.device) if want_colsum else None
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_act_bwd_f32(_lib.ptr(g), _lib.ptr(act), _lib.ptr(row_scale), _lib.ptr(out), rows, d,
                                      _lib.ptr(colsum), _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_act_bwd_f32')
    return out, colsum


class _AggregateFn(torch.autograd.Function):
    """Forward: by-dst CSR SpMM with fused `* norm_in`, `+ bias`, optional ReLU.
    Backward (autograd of DGL's gspmm = SpMM on the reverse graph):
        dY' = dY * (Y > 0) if relu;  dbias = colsum(dY');  dZ = A (b * dY').
    grad_rows (bool mask over the rows, or None): the loss_rows promise below for the gradient this node receives.  The backward then works on
    the rows of the mask only: it checks the claim on the device (check_rows_zero), takes dY' and dbias on the compact rows and gathers over
    the edges that leave them (CSRGraph.grad_support_plan with one level: rows S_0 -> all rows) — level 0 of the fused trunk's row-sparse
    backward for the one-operator-per-stage path."""

    @staticmethod
    def forward(ctx, graph, h, row_scale, bias, relu, grad_rows=None):
        out = graph.spmm(h, transpose=False, row_scale=row_scale, bias=bias, relu=relu)
        ctx.graph, ctx.relu, ctx.grad_rows = graph, relu, grad_rows
        ctx.has_bias = bias is not None
        ctx.save_for_backward(out if relu else None, row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        out, row_scale = ctx.saved_tensors
        need_h, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.grad_rows is not None and need_h:
            check_rows_zero(g, ctx.grad_rows)
            plan = ctx.graph.grad_support_plan(ctx.grad_rows, 1, max_frac=0.0)
            idx = plan.space0.idx
            gc = gather_rows_by_index(g, idx)
            if ctx.relu or row_scale is not None or need_b:
                rs = None
                if row_scale is not None:
                    rs = getattr(plan, '_row_scale0', None)
                    if rs is None or rs[0] is not row_scale:
                        rs = plan._row_scale0 = (row_scale, row_scale[idx].contiguous())
                    rs = rs[1]
                gc, dbias = act_bwd(gc, gather_rows_by_index(out, idx) if ctx.relu else None, rs, want_out=True, want_colsum=need_b)
            else:
                dbias = None
            return None, plan.levels[0][0].spmm(gc, transpose=False), None, dbias, None, None
        if not (ctx.relu or row_scale is not None or need_b):
            gs, dbias = _c(g), None
        else:
            gs, dbias = act_bwd(g, out if ctx.relu else None, row_scale, want_out=need_h, want_colsum=need_b)
        dh = ctx.graph.spmm(gs, transpose=True) if need_h else None
        return None, dh, None, dbias, None, None


class _WeightedAggregateFn(torch.autograd.Function):
    """The `edge_weight` form of the aggregation (GCN.py:199-202, fn.u_mul_e + fn.sum): Y = act(b * (sum_e w_e h[src_e]) + bias).
    edge_weight is indexed like the columns of edge_index (as graph.edata in DGL).  Backward: dh = sum over the reverse CSR with the
    same weights; dw_e = <h[src_e], b[dst_e] * dY'[dst_e]>."""

    @staticmethod
    def forward(ctx, graph, h, edge_weight, row_scale, bias, relu):
        w = edge_weight.detach().to(torch.float32)
        out = graph.spmm_weighted(h, w[graph.edge_perm(False)], False, row_scale=row_scale, bias=bias, relu=relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, relu, bias is not None
        ctx.save_for_backward(h, w, out if relu else None, row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        h, w, out, row_scale = ctx.saved_tensors
        graph = ctx.graph
        need_h, need_w, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[4]
        if ctx.relu or row_scale is not None or need_b:
            gs, dbias = act_bwd(g, out if ctx.relu else None, row_scale, want_out=True, want_colsum=need_b)
        else:
            gs, dbias = _c(g), None
        dh = graph.spmm_weighted(gs, w[graph.edge_perm(True)], True) if need_h else None
        dw = None
        if need_w:
            dw = torch.empty_like(w)
            dw[graph.edge_perm(False)] = graph.edge_dot(h, gs, False)
        return None, dh, dw, None, dbias, None


def aggregate(graph, h, row_scale=None, bias=None, relu=False, edge_weight=None, grad_rows=None):
    """grad_rows: see _AggregateFn (ignored by the weighted and the node-sharded forms: their backward is dense)."""
    _lib.require_device(h)
    if edge_weight is not None:
        if hasattr(graph, 'part'):
            raise NotImplementedError('edge_weight on a node-sharded graph (never passed by TricksComb, GCN.py:115)')
        return _WeightedAggregateFn.apply(graph, h, edge_weight, row_scale, bias, bool(relu))
    if hasattr(graph, 'part'):      # node-sharded graph: all-gather exchange + local rows (dist.py)
        from .dist import sharded_aggregate
        return sharded_aggregate(graph, h, row_scale, bias, relu)
    return _AggregateFn.apply(graph, h, row_scale, bias, bool(relu), grad_rows)


# ---------------------------------------------------------------------------------------------
# structural-embedding regulariser
# ---------------------------------------------------------------------------------------------
class _FrobeniusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, known):
        if known is not None:      # left by the fused Adam when it wrote x (optim.Adam.step): no pass over the table
            norm = known[0].clone()
        else:
            lib = _lib.load()
            xc = _c(x)
            out2 = torch.empty(2, dtype=torch.float32, device=x.device)
            wsb = lib.cb_reduce_workspace_bytes()
            ws = _ws(wsb, x.device)
            with torch.cuda.device(x.device):
                _lib.check(lib.cb_frobenius_norm_f32(_lib.ptr(xc), xc.numel(), _lib.ptr(out2), _lib.ptr(ws), wsb,
                                                     _lib.stream_ptr()), 'cb_frobenius_norm_f32')
            norm = out2[0]
        ctx.save_for_backward(x, norm)
        return norm.clone()

    @staticmethod
    def backward(ctx, g):
        x, norm = ctx.saved_tensors
        # d||x||/dx = x / ||x||; 0 at x = 0 (the subgradient torch.norm's backward picks), not NaN
        return x * torch.where(norm > 0, g / norm, torch.zeros_like(norm)), None


# Norms the fused Adam left behind: parameter._cb_norm = (parameter._version at that time, [2] device tensor {||p||_F, ||p||_F^2}).  The Adam kernel
# computes the sum of squares of the values it writes (cb_adam_multi_norm_f32) for every parameter whose norm a forward has asked for
# (`_cb_want_norm`), so the next forward's th.norm(le) (GCN.py:232) reads two floats instead of the table.  An entry is trusted only while
# the parameter's version counter is the one recorded — every torch in-place write (load_state_dict, copy_, another optimiser) bumps it —,
# never during a hipGraph capture, and never for a parameter a captured graph updates behind Python's back (`_cb_norm_off`).
# (A write through `p.data` has a version counter of its own and is not seen: code that edits a table that way calls forget_norms.)
ADAM_NORMS = True


def forget_norms(module_or_params):
    """Drops the norms the fused Adam left for these parameters (after writing them through a path that does not bump their version)."""
    params = module_or_params.parameters() if hasattr(module_or_params, 'parameters') else module_or_params
    for p in params:
        p._cb_norm = None


def known_norm(x):
    if not ADAM_NORMS or getattr(x, '_cb_norm_off', False) or torch.cuda.is_current_stream_capturing():
        return None
    hit = getattr(x, '_cb_norm', None)
    return hit[1] if hit is not None and hit[0] == x._version else None


def frobenius_norm(x):
    _lib.require_device(x)
    known = None
    if isinstance(x, torch.nn.Parameter) and x.is_leaf:
        x._cb_want_norm = True
        known = known_norm(x)
    return _FrobeniusFn.apply(x, known)


def fold_se_reg(model, optimizer, coef, se_reg_all):
    """The regulariser term `coef * sum_l ||le_l||_F` of the training loss (trainer_node_classification.py:393) WITHOUT its autograd
    path: returns the detached value to add to the loss and hands `coef / ||le_l||` to the fused Adam (optim.Adam.extra_decay_buffer),
    whose kernel then adds `coef * le / ||le||` — the term's gradient — to the data gradient it reads anyway (a non-finite
    coefficient, i.e. an all-zero table, counts as 0: the subgradient torch.norm's backward picks).  Mathematically the update of
    `loss.backward()` through `th.norm` (GCN.py:232); what is saved per table and step is the `le / ||le||` tensor (8 B/element)
    and autograd's accumulation of it into le.grad (12 B/element).  One extra launch per table.  None if the model has no SE layer
    or the optimiser is not the fused Adam (the caller then keeps the autograd path).  CB_SE_REG_FOLD=0 switches it off."""
    import os
    from . import optim
    if not isinstance(optimizer, optim.Adam):
        return None
    convs = [m for m in model.modules() if getattr(m, 'whetherHasSE', False) and getattr(m, 'se_norm', None) is not None]
    if (os.environ.get('CB_SE_REG_FOLD', '1') == '0' or not convs
            or any(c.le.grad_fn is not None or not c.le.requires_grad for c in convs)):
        # the caller keeps the autograd path for this step: coefficients left over from an earlier folded step must not act again
        for buf in optimizer._extra_decay.values():
            buf.zero_()
        return None
    for conv in convs:
        buf = optimizer.extra_decay_buffer(conv.le)
        c0 = getattr(conv, '_se_coef', None)
        if c0 is None or float(conv._se_coef_val) != float(coef) or c0.device != buf.device:
            c0 = conv._se_coef = torch.tensor(float(coef), dtype=torch.float32, device=buf.device)
            conv._se_coef_val = float(coef)
        torch.div(c0, conv.se_norm, out=buf.view(()))
    return coef * se_reg_all.detach()


def transform(feat, norm_out, weight, le=None, graph=None):
    """Z = (feat * a[:,None]) @ W (+ le) and se_reg = ||le||_F (GCN.py:213,225,230-236).  With a
    node-sharded graph `le` holds the local rows and the norm is taken over all ranks."""
    _lib.require_device(feat, weight)
    from . import gemm
    z = gemm.linear_rowscale(feat, weight, norm_out, le)
    if le is None:
        return z, None
    reg = frobenius_norm(le)
    if graph is not None and hasattr(graph, 'part'):
        from .dist import allreduce_sum
        reg = allreduce_sum(reg * reg, graph.group).sqrt()
    return z, reg


# ---------------------------------------------------------------------------------------------
# residual mixes
# ---------------------------------------------------------------------------------------------
def _axpby_raw(a, x, b, y):
    lib = _lib.load()
    x, y = _c(x), _c(y)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.cb_axpby_f32(float(a), _lib.ptr(x), float(b), _lib.ptr(y), _lib.ptr(out), x.numel(),
                                    _lib.stream_ptr()), 'cb_axpby_f32')
    return out


class _AxpbyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, x, b, y):
        ctx.a, ctx.b = a, b
        return _axpby_raw(a, x, b, y)

    @staticmethod
    def backward(ctx, g):
        gx = _axpby_raw(ctx.a, g, 0.0, g) if ctx.needs_input_grad[1] else None
        gy = _axpby_raw(ctx.b, g, 0.0, g) if ctx.needs_input_grad[3] else None
        return None, gx, None, gy


def axpby(a, x, b, y):
    """a*x + b*y for same-shape float32 device tensors."""
    _lib.require_device(x, y)
    if x.shape != y.shape or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise ValueError('axpby expects same-shape float32 tensors')
    return _AxpbyFn.apply(float(a), x, float(b), y)


# ---------------------------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------------------------
# Row-sparse backward: the masked loss's gradient w.r.t. the logits is zero outside the loss rows, and so is everything row-wise stages
# make of it.  The caller that builds such an objective SAYS so: `loss_rows=(mask, count)` handed to the forward (TeacherGNN.get_3_embs ->
# TricksComb.forward -> trunk) is the promise that the logits of that forward receive gradient in the rows of `mask` only; the trunk's
# backward then may skip the other rows — after check_rows_zero() has put the claim under the device's own eyes: a violation ends in the
# device error word AND in the guard word that keeps the optimiser launch of that step from writing (_lib.grad_guard), never in silent wrong
# gradients or updated weights.  CB_LOSS_ROWS=0 makes every backward dense.
def loss_rows_enabled():
    return os.environ.get('CB_LOSS_ROWS', '1') != '0'


def check_rows_zero(g, mask):
    """Device-side check that every row of g outside `mask` is exactly zero (cb_rows_zero_outside_mask_f32).  A violation is recorded in
    the device error word (the trainers raise where they read the loss) and in the device's guard word: the fused Adam launch that follows
    on the stream then leaves parameters and moments untouched (_lib.grad_guard)."""
    lib = _lib.load()
    _lib.require_device(g, mask)
    if g.stride(1) != 1:
        g = g.contiguous()
    m8 = mask.view(torch.uint8)
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_rows_zero_outside_mask_f32(_lib.ptr(g), g.stride(0) if g.shape[0] > 1 else g.shape[1], g.shape[0], g.shape[1], _lib.ptr(m8),
                                                     _lib.ptr(_lib.grad_guard(g.device)), _lib.stream_ptr()), 'cb_rows_zero_outside_mask_f32')


class _NllFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, mask, count, unit_grad=False):
        ctx.unit_grad = bool(unit_grad)
        lib = _lib.load()
        z = logits if logits.stride(1) == 1 else logits.contiguous()
        rows, C = z.shape
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        grad = torch.empty((rows, C), dtype=torch.float32, device=z.device) if ctx.needs_input_grad[0] else None
        wsb = lib.cb_reduce_workspace_bytes()
        ws = _ws(wsb, z.device)
        m8 = mask.view(torch.uint8) if mask is not None else None
        with torch.cuda.device(z.device):
            _lib.check(lib.cb_nll_logsoftmax_f32(_lib.ptr(z), z.stride(0), _lib.ptr(y), _lib.ptr(m8), rows, C, int(count),
                                                 _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                       'cb_nll_logsoftmax_f32')
        ctx.save_for_backward(grad)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        # unit_grad: the caller promised that this loss enters its objective with coefficient 1 and that backward() is seeded
        # with 1 (the trainers' own step): the [N, C] pass that would multiply by that 1 is skipped
        # (rows outside the mask are exactly zero: what a forward called with loss_rows=(mask, count) relies on)
        return (grad if ctx.unit_grad else grad * g), None, None, None, None


def nll_logsoftmax(logits, y, mask=None, count=None, unit_grad=False):
    """mean_{r: mask[r]} nll(log_softmax(logits[r]), y[r]) without materialising logits[mask].
    `count` = number of masked rows (precomputed once per dataset to avoid a per-step host sync).
    unit_grad=True: promise that the upstream gradient of this loss is exactly 1 (loss used with coefficient 1, backward() called on
    the objective without a seed) — the backward then hands the stored gradient on without a pass over [N, C]."""
    _lib.require_device(logits, y, mask)
    if logits.dtype != torch.float32 or y.dtype != torch.int64:
        raise TypeError('nll_logsoftmax expects float32 logits and int64 labels')
    if mask is not None and mask.dtype != torch.bool:
        raise TypeError('mask must be a bool tensor')
    if count is None:
        count = int(mask.sum().item()) if mask is not None else logits.shape[0]
    return _NllFn.apply(logits, _c(y), _c(mask) if mask is not None else None, count, bool(unit_grad))


def gather_rows_by_index(x, idx, out_bf16=False):
    """x[idx] for a float32 device matrix and an int64 index vector (row pack of the halo exchange).  out_bf16: the rows are
    narrowed to bfloat16 (round-to-nearest-even) by the same kernel — the send buffer of the bf16 halo wire."""
    lib = _lib.load()
    _lib.require_device(x, idx)
    if x.stride(1) != 1:
        x = x.contiguous()
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), x.shape[1]), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    fn = lib.cb_gather_rows_bf16_f32 if out_bf16 else lib.cb_gather_rows_f32
    with torch.cuda.device(x.device):
        _lib.check(fn(_lib.ptr(x), x.stride(0) if x.shape[0] > 1 else x.shape[1], _lib.ptr(idx), idx.numel(),
                      x.shape[1], _lib.ptr(out), _lib.stream_ptr()), 'cb_gather_rows_f32')
    return out


def expand_rows(src, pos, fill=0.0):
    """[len(pos), d] matrix whose row r is src[pos[r]] where pos[r] >= 0 and `fill` elsewhere (cb_expand_rows_f32): a compact matrix over a
    row subset (graph.RowSpace: pos int32 [N]) written back to all rows."""
    lib = _lib.load()
    _lib.require_device(src, pos)
    if src.dtype != torch.float32 or src.dim() != 2 or src.shape[1] % 4 or pos.dtype != torch.int32:
        raise ValueError('expand_rows: float32 [n, d] matrix with d % 4 == 0 and an int32 position vector expected')
    src = src.contiguous()
    out = torch.empty((pos.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(lib.cb_expand_rows_f32(_lib.ptr(src), _lib.ptr(pos.contiguous()), pos.numel(), src.shape[1], float(fill), _lib.ptr(out), _lib.stream_ptr()),
                   'cb_expand_rows_f32')
    return out


def unread_rows_fill():
    """What a rows-only training forward writes into the rows of its output that the caller promised not to read (trunk.py "Rows-only forward"):
    NaN (tuning.T.rows_only_poison, the default) — a consumer that reads them after all (TeacherGNN.out, res.commonEmb, an edge-wise loss) gets NaN
    instead of a plausible zero / bias-only row; 0.0 otherwise."""
    from .tuning import T
    return float('nan') if T.rows_only_poison else 0.0


def expand_unread(rows, space, n):
    """[n, C] output of a rows-only forward: `rows` on the loss rows (graph.RowSpace `space`), unread_rows_fill() elsewhere."""
    if rows.shape[1] % 4 == 0:
        return expand_rows(rows, space.pos, unread_rows_fill())
    return torch.full((n, rows.shape[1]), unread_rows_fill(), dtype=torch.float32, device=rows.device).index_copy_(0, space.idx, rows)


def label_propagation(graph, y0, deg_inv_sqrt, alpha, num_propagations):
    """result <- clamp(alpha * D^-1/2 A D^-1/2 result + (1 - alpha) * y0, 0, 1), `num_propagations` times, starting from
    y0 (Label_propagation_model/outcome_correlation.py:128-156 with post_step = clamp(0,1), alpha_term=True).
    Every product with the adjacency is the aggregation kernel on the cached CSR (symmetric graph)."""
    _lib.require_device(y0, deg_inv_sqrt)
    y0 = _c(y0.float())
    dis = _c(deg_inv_sqrt.float())
    a_dis = _c(dis * float(alpha))
    T = int(num_propagations)
    if T <= 0:
        return y0.clone()
    if not hasattr(graph, 'spmm_lp'):            # (a graph object without the fused step: three passes per step)
        result = y0.clone()
        for _ in range(T):
            h, _ = act_bwd(result, None, dis, want_out=True, want_colsum=False)          # D^-1/2 result
            prop = graph.spmm(h, row_scale=a_dis)                                        # alpha * D^-1/2 A (.)
            result = _axpby_raw(1.0, prop, 1.0 - float(alpha), y0).clamp_(0, 1)
        return result
    # rows padded to a multiple of 16 floats (64-byte rows: a 47-class row of 188 bytes straddles sectors — 5.16 vs 4.26 ms per
    # aggregation on the ogbn-products shape, profiles/r03_spmm_narrow_widths.md); the padding columns stay exactly zero
    c = y0.shape[1]
    cp = (c + 15) // 16 * 16 if c > 16 else c
    if cp != c:
        y0 = torch.nn.functional.pad(y0, (0, cp - c))
    # the state carried from step to step is h_t = D^-1/2 result_t: the aggregation's store forms clamp(alpha D^-1/2 (A h_t) + (1 - alpha) y0,
    # 0, 1) and scales it by D^-1/2 for the next gather (cb_spmm_csr_lp_f32); the last step leaves the scale off and returns result_T
    h, _ = act_bwd(y0, None, dis, want_out=True, want_colsum=False)                      # h_0 = D^-1/2 y0
    buf = torch.empty_like(h)
    for t in range(T):
        last = t == T - 1
        graph.spmm_lp(h, a_dis, y0, 1.0 - float(alpha), None if last else dis, out=buf)
        h, buf = buf, h
    return h[:, :c].contiguous() if cp != c else h


def se_topk_replace(le_guess, teacher_se, k, return_selection=False):
    """`SEMLP.replacement` (MLP_model/__init__.py:143-156) for all rows of `le_guess` at once: softmax-weighted mix of the K
    teacher structural embeddings with the largest inner product.  No gradient (the reference detaches both sides)."""
    lib = _lib.load()
    _lib.require_device(le_guess, teacher_se)
    q = le_guess.detach().float()
    t = teacher_se.detach().float()
    if q.stride(1) != 1:
        q = q.contiguous()
    if t.stride(1) != 1:
        t = t.contiguous()
    B, D = q.shape
    N = t.shape[0]
    if t.shape[1] != D:
        raise ValueError(f'embedding widths differ: {tuple(q.shape)} vs {tuple(t.shape)}')
    out = torch.empty((B, D), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=q.device) if return_selection else None
    wgt = torch.empty((B, k), dtype=torch.float32, device=q.device) if return_selection else None
    wsb = lib.cb_topk_replace_workspace_bytes(B, N, k)
    ws = _ws(wsb, q.device)
    with torch.cuda.device(q.device):
        _lib.check(lib.cb_topk_replace_f32(_lib.ptr(q), q.stride(0) if B > 1 else D, _lib.ptr(t), t.stride(0) if N > 1 else D, B, N, D,
                                           int(k), _lib.ptr(out), _lib.ptr(idx), _lib.ptr(wgt), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                   'cb_topk_replace_f32')
    return (out, idx, wgt) if return_selection else out
