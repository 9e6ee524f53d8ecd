"""Fused non-residual stack of TricksComb.forward — the configuration the reference's best-config table selects for Cora / Citeseer / ACTOR
('NoRes…', base_options.py:416-421; no input / output Linear, no mix, GCN.py:44-50,70-71):

    for l < L:  X = dropout(X);  Y = GCNConv_l(X)  (widths F -> H -> ... -> H -> C);  X = relu(Y) for l < L - 1      GCN.py:109-131
    out = dropout(X)                                                                    (on the logits)             GCN.py:133

One autograd node with the fused trunk's kernels (trunk.py): the dropout in front of layer 0 is applied by its GEMM while it stages x
(cb_gemm_nn_indrop_f32), every hidden layer's aggregation stores ReLU mask words and the DROPPED activation in one pass (cb_spmm_csr_fused_f32
without a mix source) and — between two hidden layers — also yields the next layer's transform (cb_spmm_gemm_fused_f32); the backward runs the
reverse aggregation + dX contraction as one kernel (cb_spmm_gemm_f32), keeps ReLU masks as bits and regenerates dropout masks.  Hidden width
256 (the fused store's row layout), one GPU or row shards (dist.ShardedGraph: the exchanges run inside the same helpers as the trunk's, the
reverse aggregation + dX kernel as the last halo pass); every other shape takes the operator path (GCN.py _forward_modular), whose arithmetic and sequence
of dropout seeds this node reproduces (tests/test_gpu_model.py::test_fused_stack_equals_modular_path; goldens case_nr_h256_*)."""
import torch

from . import _lib, gemm, ops
from .trunk import _exchanged, _fused_gemm, _fused_launch, _fused_spmm, _layer_bwd, _layer_bwd_rows, _spmm_t, agg_gemm_eligible, rows_only_enabled
from .tuning import T


def _plan_hint(graph, loss_rows, n_rows, ag, committed=False):
    """loss_rows if the backward (and a rows-only forward) may run on the row-support plan, else None — one decision for both.  committed (the backward of
    a rows-only forward): the plan's build / hit bookkeeping is not asked again."""
    if (loss_rows is not None and ops.loss_rows_enabled() and not hasattr(graph, 'part') and loss_rows[0].shape[0] == n_rows
            and getattr(graph, 'rowptr_t', None) is not None and 1 <= loss_rows[1] <= T.rowsparse_s0_limit * n_rows
            and (n_rows >= T.rowsparse_min_nodes or getattr(graph, 'rowsparse_small_ok', False)) and ag and (committed or graph.support_plan_pays())):
        return loss_rows
    return None


def eligible(tc, x, graph, want_les):
    return (not tc.has_residual_MLP and not want_les and tc.dim_hidden == 256 and tc.num_layers >= 2 and len(tc.layers_GCN) == tc.num_layers
            and tc.args.type_trick not in ('BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm')
            and x.is_cuda and x.dtype == torch.float32 and getattr(tc.args, 'agg_dtype', 'f32') == 'f32'
            and tc.args.dropout == tc.dropout and (hasattr(graph, 'spmm_gemm') or hasattr(graph, 'part')))


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, cfg, x, *layer_params):
        """layer_params = (W_0, bias_0, le_0 | None, W_1, ...).  cfg = (L, p, seeds[L + 1], track, loss_rows, rows_only): loss_rows = None or (bool mask [N],
        count), the caller's promise that the output receives gradient in those rows only (ops.py "Row-sparse backward"); rows_only: ... and that it
        READS the output in those rows only (trunk.py "Rows-only forward")."""
        L, p, seeds, track, loss_rows_, rows_only = cfg
        a, b = graph.norm_out, graph.norm_in
        row0 = int(getattr(graph, 'row_offset', 0))       # first global row of this rank's block (dropout masks are those of the unsharded tensor)
        x = x.contiguous()
        bwd = bool(track) and any(ctx.needs_input_grad)
        w0, _b0, le0 = layer_params[0:3]
        # layer 0: Z_0 = a * (dropout(x) W_0) + E_0, the dropout applied while the GEMM stages x where that form exists
        z = gemm.mm_nn_indrop(x, w0, p, seeds[0], row0, rowscale=a, addend=le0, out=_exchanged(graph, x.shape[0], w0.shape[1])) if p > 0 else None
        ctx.indrop = z is not None
        if z is None:
            xd0 = ops._dropout_raw(x, p, seeds[0], row0 * x.shape[1]) if p > 0 else x
            z = gemm.mm_nn(xd0, w0, rowscale=a, addend=le0, out=_exchanged(graph, x.shape[0], w0.shape[1]))
        else:
            xd0 = x                       # kept for the backward: the UNdropped features (the weight gradient regenerates the mask)
        ag = agg_gemm_eligible(graph, 256, False)
        saved_in, saved_bits = [xd0], []
        z_ready = None
        # Rows-only forward (trunk.py; one GPU, under the plan decision of the backward): the logits are read on S_0 only, so the last aggregation (class
        # width) runs on those rows, the last layer's transform on the rows it gathers — S_1 —, and the last HIDDEN layer's aggregation + store on S_1 too
        # (cb_spmm_csr_fused_rows_f32), wherever the plan keeps S_1 compact.
        ro = None
        if rows_only and bwd and L >= 2 and rows_only_enabled() and x.shape[0] >= T.rows_only_min_nodes:
            hint = _plan_hint(graph, loss_rows_, x.shape[0], ag)
            if hint is not None:
                plan_ = graph.grad_support_plan(hint[0], L, max_frac=T.rowsparse_max_frac)
                ro = graph.rows_only_fwd(plan_)      # (fwd1, fwd0c, ids1, b1, s1) or None
                if ro is not None:
                    ro = ro + (plan_,)
        for l in range(L - 1):            # hidden layers: aggregation with the ReLU / dropout store (+ the next hidden layer's transform)
            _w, bias, _le = layer_params[3 * l: 3 * l + 3]
            w1, _b1, le1 = layer_params[3 * (l + 1): 3 * (l + 1) + 3]
            sd = seeds[l + 1] if p > 0 else 0
            z = z_ready if z_ready is not None else z
            z_ready = None
            if ro is not None and l == L - 2:      # the last hidden layer: read by the last aggregation only, on S_1
                ro[0].profile = getattr(graph, 'profile', None)
                bits, cur, _ = _fused_launch(_lib.load(), graph, ro[0], z, None, bias, None, 1.0, 0.0, p, sd, False, want_bits=bwd, row_ids=ro[2], row_scale=ro[3])
                s1 = ro[4]
                le_c = ops.gather_rows_by_index(le1, s1.idx) if le1 is not None else None
                z = gemm.mm_nn(cur, w1, rowscale=s1.a, addend=le_c)          # Z_{L-1} on the rows of S_1
            elif ag and l + 1 < L - 1:      # the next layer is H -> H: its transform leaves this layer's aggregation kernel
                from .graph import weight_image
                # (a forward that no backward follows leaves cur = None: the activations stayed on chip)
                bits, cur, z_ready = _fused_gemm(graph, z, bias, None, 1.0, 0.0, p, sd, weight_image(w1), a, le1, want_bits=bwd)[:3]
            else:
                bits, cur, _ = _fused_spmm(graph, z, bias, None, 1.0, 0.0, p, sd, want_bits=bwd)
                z = gemm.mm_nn(cur, w1, rowscale=a, addend=le1, out=_exchanged(graph, cur.shape[0], w1.shape[1]) if w1.shape[1] % 4 == 0 else None)
            if bwd:
                saved_bits.append(bits)
                saved_in.append(cur)
        z = z_ready if z_ready is not None else z
        bias_last = layer_params[3 * (L - 1) + 1]
        # the last layer: no ReLU (GCN.py:127); then the dropout on the logits (GCN.py:133)
        if ro is not None:      # the logits on the loss rows (gathering the compact Z over the orientation renumbered to S_1)
            plan_ = ro[5]
            sp = plan_.space0
            ro[1].profile = getattr(graph, 'profile', None)
            b0 = getattr(sp, '_norm_in', None)
            if b0 is None:
                b0 = sp._norm_in = b[sp.idx].contiguous()
            y_c = ro[1].spmm(z, row_scale=b0, bias=bias_last)
            y = ops.expand_unread(y_c, sp, x.shape[0])      # the rows nobody may read: NaN (ops.unread_rows_fill)
            graph.rows_only_forwards = getattr(graph, 'rows_only_forwards', 0) + 1
        else:
            y = graph.aggregate(z, False, b, bias_last, False) if hasattr(graph, 'part') else graph.spmm(z, row_scale=b, bias=bias_last)
        out = ops._dropout_raw(y, p, seeds[L], row0 * y.shape[1]) if p > 0 else y
        ctx.in_last_compact = bwd and ro is not None      # saved_in[L - 1] holds the rows of S_1 (= level 0's destination)
        ctx.graph, ctx.cfg = graph, cfg
        if bwd:
            ctx.save_for_backward(*saved_in, *saved_bits, *[t for t in layer_params if t is not None])
        ctx.le_present = [layer_params[3 * l + 2] is not None for l in range(L)]
        return out

    @staticmethod
    def backward(ctx, gout):
        graph, (L, p, seeds, _track, loss_rows, _rows_only) = ctx.graph, ctx.cfg
        sv = list(ctx.saved_tensors)
        saved_in, saved_bits, rest = sv[:L], sv[L: 2 * L - 1], sv[2 * L - 1:]
        lp, k = [], 0
        for l in range(L):
            w, bias = rest[k], rest[k + 1]
            k += 2
            le = None
            if ctx.le_present[l]:
                le = rest[k]
                k += 1
            lp.append((w, bias, le))
        a, b = graph.norm_out, graph.norm_in
        row0 = int(getattr(graph, 'row_offset', 0))
        sharded = hasattr(graph, 'part')
        need = ctx.needs_input_grad          # (graph, cfg, x, *layer_params)
        nw = lambda l: need[3 + 3 * l]       # noqa: E731
        nb = lambda l: need[3 + 3 * l + 1]   # noqa: E731
        nle = lambda l: lp[l][2] is not None and need[3 + 3 * l + 2]      # noqa: E731
        grads = [None] * (3 * L)
        ag = agg_gemm_eligible(graph, 256, False)
        gout = gemm._rowmajor(gout)
        # Row-sparse backward (one GPU; trunk.py / DESIGN.md section 1): under the caller's loss_rows promise the levels of the backward whose support
        # is small run on compact [|S_j|, .] matrices through the plan's renumbered orientations; the promise is checked on the device.
        plan = None
        hint = _plan_hint(graph, loss_rows, gout.shape[0], ag, committed=ctx.in_last_compact)
        if hint is not None:
            ops.check_rows_zero(gout, hint[0])
            plan = graph.grad_support_plan(hint[0], L, max_frac=T.rowsparse_max_frac, count=not ctx.in_last_compact)      # (one use per step)
        if ctx.in_last_compact and (plan is None or plan.levels[0][1] is None):
            raise RuntimeError('the forward ran its last layers on the loss rows\' supports (rows_only), but its backward finds no such plan: '
                               'CB_LOSS_ROWS / tuning.T / the mask changed between the forward and the backward')
        space = plan.space0 if plan is not None else None                        # row space of g / gr (None: all rows)

        def level_of(j):
            return plan.levels[j] if (plan is not None and space is not None and j < len(plan.levels)) else None

        def rows_of(x_full, sp):
            return ops.gather_rows_by_index(x_full, sp.idx) if sp is not None else x_full

        def all_rows(gz_c, sp):      # a table's gradient is dL/dZ_l on ALL rows: the support's rows, zeros elsewhere
            if sp is None:
                return gz_c
            if gz_c.shape[1] % 4 == 0:
                return ops.expand_rows(gz_c, sp.pos)
            out_ = torch.zeros((sp.pos.numel(), gz_c.shape[1]), dtype=gz_c.dtype, device=gz_c.device)
            out_[sp.idx] = gz_c
            return out_
        # last layer: dropout on the logits, bias, degree norm, reverse aggregation at the class width
        gd = ops._dropout_raw(gout, p, seeds[L], row0 * gout.shape[1]) if p > 0 else gout
        gr, grads[3 * (L - 1) + 1] = ops.act_bwd(gd, None, b, want_out=True, want_colsum=nb(L - 1))
        level = level_of(0)
        if level is not None:            # gathers the loss rows only; writes S_1 (or all rows)
            csr, dst = level
            csr.profile = getattr(graph, 'profile', None)
            gz = csr.spmm(rows_of(gr, space))
            space = dst
        else:
            gz = _spmm_t(graph, gr)
        a_sp = space.a if space is not None else a
        w_last = lp[L - 1][0]
        if nw(L - 1):
            x_last = saved_in[L - 1] if ctx.in_last_compact else rows_of(saved_in[L - 1], space)      # (rows-only forward: already the rows of S_1)
            grads[3 * (L - 1)] = gemm.mm_tn(x_last, gz, rowscale=a_sp)
        if nle(L - 1):
            grads[3 * (L - 1) + 2] = all_rows(gz, space)
        g = gemm.mm_nn(gz, w_last.t().contiguous(), rowscale=a_sp)               # dL/d(dropped X_{L-1}), on the rows of `space`
        del gd, gr, gz
        for l in range(L - 2, -1, -1):
            w = lp[l][0]
            sd_l = seeds[l + 1] if p > 0 else 0
            if space is not None:
                gr, grads[3 * l + 1] = _layer_bwd_rows(g, space.idx, saved_bits[l], b, p, sd_l, row0, 1.0, nb(l))
            else:
                gr, grads[3 * l + 1] = _layer_bwd(g, saved_bits[l], b, None, False, p, sd_l, row0, 1.0, 0.0, nb(l),
                                                  out=_exchanged(graph, g.shape[0], g.shape[1]) if sharded else None)
            del g
            g = None
            level = level_of(L - 1 - l)
            if level is not None:        # a compact level of the plan: the level's own orientation, rows of S_{j+1} (or all rows) out
                from .graph import weight_image
                csr, dst = level
                csr.profile = getattr(graph, 'profile', None)
                a_dst = dst.a if dst is not None else a
                if l > 0:
                    gz, g = csr.spmm_gemm(gr, weight_image(w, transpose=True), transpose=False, g_rowscale=a_dst)
                else:
                    gz = csr.spmm(gr)
                    if need[2]:
                        g = gemm.mm_nn(gz, w.t().contiguous(), rowscale=a_dst)
                del gr
                if nw(l):
                    if l == 0 and ctx.indrop and dst is None:
                        dw = gemm.mm_tn_adrop(saved_in[0], gz, p, seeds[0], row0, rowscale=a)
                        if dw is None:
                            dw = gemm.mm_tn(ops._dropout_raw(saved_in[0], p, seeds[0], row0 * saved_in[0].shape[1]), gz, rowscale=a)
                        grads[0] = dw
                    else:
                        x_in = saved_in[l] if not (l == 0 and ctx.indrop) else ops._dropout_raw(saved_in[0], p, seeds[0], row0 * saved_in[0].shape[1])
                        grads[3 * l] = gemm.mm_tn(rows_of(x_in, dst), gz, rowscale=a_dst)
                if nle(l):
                    grads[3 * l + 2] = all_rows(gz, dst)
                del gz
                space = dst
                continue
            if ag and l > 0:             # dL/dZ_l = A (b * dY') and a * (dL/dZ_l W_l^T) from one kernel (sharded: as the last halo pass)
                from .graph import weight_image
                img = weight_image(w, transpose=True)
                if sharded:
                    gz, g = graph.aggregate_finish(graph.aggregate_start(gr, True), True,
                                                   last_pass=lambda csr, recv, acc: csr.spmm_gemm(recv, img, transpose=False, g_rowscale=a, acc_init=acc))
                else:
                    gz, g = graph.spmm_gemm(gr, img, transpose=True, g_rowscale=a)
            else:
                gz = _spmm_t(graph, gr)
                if l > 0 or need[2]:
                    g = gemm.mm_nn(gz, w.t().contiguous(), rowscale=a)
            del gr
            if nw(l):
                if l == 0 and ctx.indrop:      # the mask of the dropout in front of layer 0 is regenerated while the GEMM stages x
                    dw = gemm.mm_tn_adrop(saved_in[0], gz, p, seeds[0], row0, rowscale=a)
                    if dw is None:
                        dw = gemm.mm_tn(ops._dropout_raw(saved_in[0], p, seeds[0], row0 * saved_in[0].shape[1]), gz, rowscale=a)
                    grads[0] = dw
                else:
                    grads[3 * l] = gemm.mm_tn(saved_in[l], gz, rowscale=a)
            if nle(l):
                grads[3 * l + 2] = gz
            del gz
        d_x = None
        if need[2]:
            d_x = ops._dropout_raw(g, p, seeds[0], row0 * g.shape[1]) if p > 0 else g
        return (None, None, d_x, *grads)


def forward(tc, x, graph, loss_rows=None, rows_only=False):
    """TricksComb.forward on the fused non-residual stack; returns (logits, se_reg_all).  loss_rows, rows_only: as trunk.forward."""
    L = tc.num_layers
    p = float(tc.dropout) if tc.training else 0.0
    seeds = tuple(ops.next_seed() for _ in range(L + 1)) if p > 0 else (0,) * (L + 1)
    params, se_reg_all = [], None
    for conv in tc.layers_GCN:
        le = conv.le if conv.whetherHasSE else None
        params += [conv.weight, conv.bias, le]
        if le is not None:
            reg = ops.frobenius_norm(le)
            if hasattr(graph, 'part'):      # row shards: the norm is over all ranks' rows
                from .dist import allreduce_sum
                reg = allreduce_sum(reg * reg, graph.group).sqrt()
            conv.se_norm = reg.detach()
            se_reg_all = reg if se_reg_all is None else se_reg_all + reg
    if not all(c._allow_zero_in_degree for c in tc.layers_GCN):      # GCN.py:187-197
        graph.check_zero_in_degree()
    if loss_rows is not None:
        mask, count = loss_rows if isinstance(loss_rows, (tuple, list)) else (loss_rows, None)
        if mask.dtype != torch.bool or mask.dim() != 1 or mask.shape[0] != x.shape[0]:
            raise ValueError(f'loss_rows: a bool mask over the {x.shape[0]} rows expected, got {tuple(mask.shape)} {mask.dtype}')
        loss_rows = (mask, int(count) if count is not None else int(mask.sum().item()))
    out = _StackFn.apply(graph, (L, p, seeds, torch.is_grad_enabled(), loss_rows, bool(rows_only) and loss_rows is not None), x, *params)
    return out, se_reg_all
