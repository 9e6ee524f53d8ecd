"""GPU: the row-sparse backward (trunk.py, CSRGraph.grad_support_plan, the loss_rows promise of the forward / ops.check_rows_zero).
The masked loss of trainer_node_classification.py:390-391 has a gradient that is zero in every row outside the train rows, and the
backward keeps it zero outside the rows those can reach; the levels of the backward whose support is small run on compact matrices.
Gradients equal the dense backward's up to the order in which sums are associated, the claim is verified on the device, and a violation
is reported instead of training on."""
import contextlib
import io
import os

import pytest
import torch

from gnn_tail_generalization_amd import tuning

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step_grads(flag, dataset='S-pl1M', se='000', layers=3, extra=(), n_loss_rows=None, rows_only=False):
    """One training step's (loss, gradients, plan used).  flag: CB_LOSS_ROWS.  rows_only=False pins the FORWARD to all rows (CB_ROWS_ONLY_FWD=0), so
    that the tests of the backward compare like with like (the same loss bit for bit); the rows-only forward has its own tests below."""
    import bench
    from gnn_tail_generalization_amd import _lib, ops
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    old, old_ro = os.environ.get('CB_LOSS_ROWS'), os.environ.get('CB_ROWS_ONLY_FWD')
    os.environ['CB_LOSS_ROWS'] = flag
    os.environ['CB_ROWS_ONLY_FWD'] = '1' if rows_only else '0'
    try:
        args = bench.make_args(dataset, ['--manual_assign_GPU=0'] + list(extra), se=se, layers=layers)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            t = tnc.trainer(args, 0)
            if n_loss_rows is not None:      # sparse labels: every 7919-th row
                m = torch.zeros_like(t.data.train_mask)
                m[(torch.arange(n_loss_rows, device=m.device) * 7919) % m.numel()] = True
                t.data.train_mask, t.data.test_mask = m, ~m
            t.setup_teacherGNN()
        t.teacherGNN.train()
        ops._seed_override[:] = [11, 12, 13, 14, 15]
        loss = t.training_loss()
        loss.backward()
        ops._seed_override[:] = []
        torch.cuda.synchronize()
        assert _lib.load().cb_device_status() == 0
        used = getattr(t.graph(), '_support_plan', None) is not None
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in t.teacherGNN.named_parameters() if p.grad is not None}, used
    finally:
        ops._seed_override[:] = []
        for k, v in (('CB_LOSS_ROWS', old), ('CB_ROWS_ONLY_FWD', old_ro)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('conn', ['Initial', 'Residual'])
@pytest.mark.parametrize('max_frac,se,loss_side', [(0.6, '000', True), (0.6, '000', False), (0.0, '000', True), (0.6, '111', True)])
def test_row_sparse_backward_equals_the_dense_backward(max_frac, se, loss_side, conn, monkeypatch):
    """max_frac 0.6: the supports S_0 (10 % of the rows) and S_1 (45 %) compact, from S_2 (94 %) on dense; 0: only the gathered side of the
    first aggregation compact.  se 111: structural-embedding tables, whose gradient dL/dZ_l is scattered from the compact level to all rows.
    loss_side: level 0's GEMM and weight gradient contracted over the loss rows (plan.fwd0), whether its destination is compact or dense —
    not with a table gradient."""
    from gnn_tail_generalization_amd import trunk
    monkeypatch.setattr(tuning.T, 'rowsparse_max_frac', max_frac)
    monkeypatch.setattr(tuning.T, 'rowsparse_loss_side', loss_side)
    spmm_rows = []
    from gnn_tail_generalization_amd.graph import CSRGraph
    real = CSRGraph.spmm
    monkeypatch.setattr(CSRGraph, 'spmm', lambda self, h, *a, **k: (spmm_rows.append((self.N, self.n_cols)), real(self, h, *a, **k))[1])
    extra = () if conn == 'Initial' else ('--force_set_to_best_config=0', '--type_trick=Residual')      # (Residual: cumulative supports, a second compact gradient)
    loss_s, g_s, used_s = _step_grads('1', se=se, extra=extra)
    took_loss_side = any(n < c for n, c in spmm_rows)  # fwd0: one row per loss row, all columns
    assert took_loss_side == (loss_side and se == '000')
    loss_d, g_d, used_d = _step_grads('0', se=se, extra=extra)
    assert used_s and not used_d                       # the 10 % train mask of the stand-in: the plan was built and used
    assert loss_s == loss_d
    assert set(g_s) == set(g_d)
    for k in g_d:
        scale = float(g_d[k].abs().max())
        # same addends; the association of sums differs (hub chunks, the slabs of the weight-gradient reductions)
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * scale, k


@pytest.mark.parametrize('se,layers', [('000', 3), ('111', 3), ('000', 2), ('100', 4)])
def test_row_sparse_backward_of_the_non_residual_stack_equals_its_dense_backward(se, layers):
    """stack.py (NoRes: F -> H -> ... -> C, dropout on the logits): level 0 of the plan runs at the class width on the loss rows, the hidden levels
    on compact matrices through the aggregation + dX kernel; gradients equal the dense backward's up to the association of sums."""
    extra = ('--force_set_to_best_config=0', '--type_trick=NoResNodeNorm')
    loss_s, g_s, used_s = _step_grads('1', se=se, layers=layers, extra=extra)
    loss_d, g_d, used_d = _step_grads('0', se=se, layers=layers, extra=extra)
    assert used_s and not used_d and loss_s == loss_d and set(g_s) == set(g_d)
    for k in g_d:
        scale = float(g_d[k].abs().max())
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * scale, k


@pytest.mark.parametrize('extra,se', [(('--type_trick=Initial',), '000'), (('--type_trick=Residual',), '111'), (('--type_trick=NodeNorm',), '000'),
                                      (('--type_trick=DenseNodeNorm',), '000'), (('--type_trick=NoRes',), '001'), (('--type_trick=JumpingBatchNorm',), '000')])
def test_row_sparse_last_aggregation_of_the_operator_path_equals_the_dense_backward(extra, se, monkeypatch):
    """configurations outside the fused nodes (other hidden widths — the dataset presets fix them, base_options.py:192-224 —, norms, 'Dense'
    / 'Jumping' connections) run one operator per stage; the backward of their LAST aggregation (ops._AggregateFn with grad_rows) works on the loss rows: dY'
    and the bias gradient on the compact rows, the gather over the edges that leave them.  Everything below is dense.  Forced here for the fusable
    shapes too (use_fused_trunk = False).  Same gradients as CB_LOSS_ROWS=0 up to the association of the hub rows' sums."""
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.GNN_model.GCN import TricksComb
    monkeypatch.setattr(TricksComb, 'use_fused_trunk', False)
    hinted = []
    real = ops._AggregateFn.apply
    monkeypatch.setattr(ops._AggregateFn, 'apply', lambda *a: (hinted.append(a[5] is not None), real(*a))[1])
    extra = ('--force_set_to_best_config=0',) + extra
    loss_s, g_s, used_s = _step_grads('1', se=se, extra=extra)
    assert hinted == [False, False, True]              # three layers: the last aggregation alone carries the promise
    loss_d, g_d, used_d = _step_grads('0', se=se, extra=extra)
    assert used_s and not used_d and loss_s == loss_d and set(g_s) == set(g_d)
    for k in g_d:
        scale = float(g_d[k].abs().max())
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * scale, k


def test_column_statistic_norms_keep_the_dense_backward():
    """BatchNorm / PairNorm / ... take column statistics after the last aggregation: their backward reaches every row, the promise about the logits says
    nothing about the aggregation's gradient, and the operator path does not pass it on."""
    extra = ('--force_set_to_best_config=0', '--type_trick=BatchNorm')
    _, _, used = _step_grads('1', extra=extra)
    assert not used


def _close_up_to_relu_flips(got, ref, name):
    """(sum a X) W and sum a (X W) differ in the last bits, and a pre-activation that is zero to rounding may land on either side of the ReLU: one such
    element among the 2.6 * 10^7 of S-pl1M's loss rows moves every gradient by ~ 1 / sqrt(2.6 * 10^7) = 2 * 10^-4 of its norm (measured:
    tools/probes/rows_only_dbg.py — no flipped mask word: 1e-7; two: 1.8e-4).  A wrong term would show at 1e-2 and more."""
    assert float((got - ref).norm()) <= 1e-3 * float(ref.norm()), name
    if not name.endswith('.le'):      # (a table's gradient is per node row, not a sum over rows: a flipped element IS an entry of it)
        assert float((got - ref).abs().max()) <= 5e-3 * float(ref.abs().max()), name


@pytest.mark.parametrize('conn,se,layers,n_loss_rows,below', [('Initial', '000', 3, None, '2'), ('Residual', '000', 3, None, '2'), ('Initial', '100', 3, None, '2'),
                                                              ('Initial', '000', 2, None, '2'), ('Residual', '100', 4, None, '2'), ('Initial', '000', 3, 200, '2'),
                                                              ('Initial', '000', 3, None, '1'), ('Residual', '000', 4, None, '1'),
                                                              ('Initial', '000', 3, None, '0'), ('Residual', '000', 2, None, '0')])
def test_rows_only_forward_equals_the_dense_step(conn, se, layers, n_loss_rows, below, monkeypatch):
    """Rows-only forward (trunk._last_layer_on_loss_rows): the trainer promises that it reads — not only differentiates — the logits in the train rows
    only, and the training forward evaluates its LAST layer (aggregation of the inputs over the edges that enter the loss rows, transform, store,
    output Linear) on those rows; the backward's level 0 contracts the saved aggregate.  Against the all-rows step with the dense backward: the
    same loss and gradients up to the association of sums — (sum a X) W against sum a (X W) — and the ReLUs that association flips at zero.
    below = CB_ROWS_ONLY_BELOW, what the layer UNDER the last one does: '2' (default) — its sum first too, on the rows the last layer reads (S_1: aggregate,
    GEMM on |S_1| rows, store; the backward's level 1 contracts the saved aggregate; needs a layer under it and no table on it, else as '1');
    '1' — Z-first on S_1 (cb_spmm_csr_fused_rows_f32: the fused store over a subset of the rows; 'Residual': its ReLU output, the last layer's mix
    source, lives there too); '0' — on all rows."""
    from gnn_tail_generalization_amd import trunk
    calls, subset_launches, on_rows = [], [], []
    real = trunk._last_layer_on_loss_rows
    monkeypatch.setattr(trunk, '_last_layer_on_loss_rows', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    real_launch = trunk._fused_launch
    monkeypatch.setattr(trunk, '_fused_launch', lambda *a, **k: (subset_launches.append(k.get('row_ids') is not None), real_launch(*a, **k))[1])
    real_rows = trunk._layer_on_rows
    monkeypatch.setattr(trunk, '_layer_on_rows', lambda *a, **k: (on_rows.append(1), real_rows(*a, **k))[1])
    monkeypatch.setenv('CB_ROWS_ONLY_BELOW', below)
    monkeypatch.setenv('CB_SPMM_STORE_BWD', '2' if n_loss_rows is None else '1')      # (2: the store backward also in the epilogue of compact levels)
    monkeypatch.setattr(tuning.T, 'sum_first_below_min_edges', 0)      # (S-pl1M sits below the break-even of the form)
    sum_first_below = below == '2' and layers >= 3 and se[1] == '0'      # (the residual trunks' GCNConvs all take the middle flag of whetherHasSE, GCN.py:58-60)
    extra = () if conn == 'Initial' else ('--force_set_to_best_config=0', '--type_trick=Residual')
    loss_s, g_s, used_s = _step_grads('1', se=se, layers=layers, extra=extra, n_loss_rows=n_loss_rows, rows_only=True)
    assert calls == [1] and used_s
    assert len(on_rows) == (2 if sum_first_below else 1) and any(subset_launches) == (below != '0' and not sum_first_below)
    loss_d, g_d, used_d = _step_grads('0', se=se, layers=layers, extra=extra, n_loss_rows=n_loss_rows)
    assert calls == [1] and not used_d
    assert abs(loss_s - loss_d) <= 2e-6 * abs(loss_d) and set(g_s) == set(g_d)
    for k in g_d:
        _close_up_to_relu_flips(g_s[k], g_d[k], k)


def test_rows_only_forward_returns_the_loss_rows_and_poison(monkeypatch):
    """What the caller gets under both promises: the logits of the all-rows forward (same dropout masks: they are drawn at the global row) in the rows
    of the mask, NaN in every other row (a reader that breaks the promise — TeacherGNN.out, res.commonEmb, an edge-wise loss — must not get
    plausible numbers; tuning.T.rows_only_poison = False: zeros).  Without rows_only — the gradient promise alone — every row is evaluated."""
    import bench
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    args = bench.make_args('S-pl1M', ['--manual_assign_GPU=0'])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    t.teacherGNN.train()
    mask, n = t.data.train_mask, int(t.data.train_mask.sum())
    outs = {}
    for ro in (True, False):
        ops._seed_override[:] = [21, 22, 23, 24, 25]
        outs[ro] = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, loss_rows=(mask, n), rows_only=ro).emb4classi_full
        ops._seed_override[:] = []
    assert bool(torch.isnan(outs[True].detach()[~mask]).all()) and bool(torch.isfinite(outs[False].detach()).all())
    assert bool(torch.isfinite(t.teacherGNN.out.detach()).all())      # (.out is the LAST forward's: the all-rows one)
    torch.testing.assert_close(outs[True].detach()[mask], outs[False].detach()[mask], atol=2e-5, rtol=2e-5)
    monkeypatch.setattr(tuning.T, 'rows_only_poison', False)
    ops._seed_override[:] = [21, 22, 23, 24, 25]
    res = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, loss_rows=(mask, n), rows_only=True)
    ops._seed_override[:] = []
    assert float(res.emb4classi_full.detach()[~mask].abs().max()) == 0.0 and t.teacherGNN.out is res.commonEmb
    monkeypatch.setattr(tuning.T, 'rows_only_poison', True)
    # the trainer's own step makes the promise (and a loss that stays finite shows that nothing of it reads the poisoned rows) ...
    assert bool(torch.isfinite(t.training_loss().detach()))
    assert bool(torch.isnan(t.teacherGNN.out.detach()[~mask]).all())
    # ... withdraws it on request ...
    t.rows_only_forward = False
    assert bool(torch.isfinite(t.training_loss().detach())) and bool(torch.isfinite(t.teacherGNN.out.detach()).all())
    del t.rows_only_forward
    # ... and refuses the objective that would read every row of the same forward (trainer…:417-418), inside training_loss itself
    t.args.has_loss_component_edgewise = True
    with pytest.raises(NotImplementedError, match='only the train rows'):
        t.training_loss()
    t.args.has_loss_component_edgewise = False
    with torch.no_grad():      # a forward that no backward follows keeps every row, whatever was promised
        ops._seed_override[:] = [21, 22, 23, 24, 25]
        out_ng = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, loss_rows=(mask, n), rows_only=True).emb4classi_full
        ops._seed_override[:] = []
    torch.testing.assert_close(out_ng, outs[False].detach(), atol=0, rtol=0)


@pytest.mark.parametrize('se,layers', [('000', 3), ('111', 3), ('000', 2), ('100', 4)])
def test_rows_only_forward_of_the_non_residual_stack(se, layers, monkeypatch):
    """stack.py (NoRes: F -> H -> ... -> C): the last aggregation (class width) on the loss rows, the last transform on the rows it gathers (S_1), the last
    hidden layer's aggregation + store on S_1 (cb_spmm_csr_fused_rows_f32).  Same sums in the same order as the all-rows forward: the loss is the same to
    the last bit, the gradients equal the dense backward's as the row-sparse backward's do."""
    from gnn_tail_generalization_amd import stack
    subset_launches = []
    real_launch = stack._fused_launch
    monkeypatch.setattr(stack, '_fused_launch', lambda *a, **k: (subset_launches.append(k.get('row_ids') is not None), real_launch(*a, **k))[1])
    extra = ('--force_set_to_best_config=0', '--type_trick=NoResNodeNorm')
    loss_s, g_s, used_s = _step_grads('1', se=se, layers=layers, extra=extra, rows_only=True)
    assert subset_launches == [True] and used_s
    loss_d, g_d, used_d = _step_grads('0', se=se, layers=layers, extra=extra)
    assert not used_d and loss_s == loss_d and set(g_s) == set(g_d)
    for k in g_d:
        scale = float(g_d[k].abs().max())
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * scale, k


@pytest.mark.parametrize('conn,below', [('Initial', '2'), ('Residual', '2'), ('Initial', '0')])
def test_rows_only_forward_with_structural_embedding_tables(conn, below, monkeypatch):
    """whetherHasSE=111 (a table on every GCNConv, GCN.py:230-232): the last layer's table rows are summed over the same edges as its inputs,
    Y[S_0] = b * (H W + (A le)[S_0]) + bias; the table's gradient is dL/dZ on all of S_1, so the backward's level 0 stays on the compact form (aggregation +
    dX kernel over the level's own orientation) and reads the layer's input on S_1 — which the layer below, Z-first on S_1, left compact (below '0': on
    all rows)."""
    from gnn_tail_generalization_amd import trunk
    calls = []
    real = trunk._last_layer_on_loss_rows
    monkeypatch.setattr(trunk, '_last_layer_on_loss_rows', lambda *a, **k: (calls.append(k.get('le') is not None), real(*a, **k))[1])
    monkeypatch.setenv('CB_ROWS_ONLY_BELOW', below)
    extra = () if conn == 'Initial' else ('--force_set_to_best_config=0', '--type_trick=Residual')
    loss_s, g_s, used_s = _step_grads('1', se='111', extra=extra, rows_only=True)
    assert calls == [True] and used_s
    loss_d, g_d, _ = _step_grads('0', se='111', extra=extra)
    assert abs(loss_s - loss_d) <= 2e-6 * abs(loss_d) and set(g_s) == set(g_d)
    for k in g_d:
        _close_up_to_relu_flips(g_s[k], g_d[k], k)


@pytest.mark.parametrize('n_loss_rows', [3, 200, 20000])
def test_row_sparse_backward_with_sparse_labels(n_loss_rows, monkeypatch):
    """Few loss rows (the public Planetoid splits label 0.3 - 5 % of the nodes): supports of 3 / 200 / 20 000 rows that grow by orders of
    magnitude per level — deeper levels than the first take the source-side form too (plan.fwd[j]) — against the dense backward."""
    from gnn_tail_generalization_amd import graph
    monkeypatch.setattr(tuning.T, 'fwd0_min_edges', 0)
    loss_s, g_s, used_s = _step_grads('1', n_loss_rows=n_loss_rows)
    loss_d, g_d, used_d = _step_grads('0', n_loss_rows=n_loss_rows)
    assert used_s and not used_d and loss_s == loss_d
    for k in g_d:
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * float(g_d[k].abs().max()), k


def test_row_sparse_backward_with_bf16_stored_rows():
    """--agg_dtype bf16 (BASELINE config 2's storage of the gathered rows): the plan's compact levels run on fp32 matrices, the dense levels
    below them as the bf16 path does — the gradients differ from the all-bf16 dense backward by what bf16 rows cost there, not more."""
    loss_s, g_s, used_s = _step_grads('1', extra=['--agg_dtype=bf16'])
    loss_d, g_d, used_d = _step_grads('0', extra=['--agg_dtype=bf16'])
    loss_f, g_f, _ = _step_grads('0')                                      # fp32 rows, dense backward: the yardstick
    assert used_s and not used_d and loss_s == loss_d
    for k in g_d:
        scale = float(g_f[k].abs().max())
        err_dense_bf16 = float((g_d[k] - g_f[k]).abs().max())
        assert float((g_s[k] - g_f[k]).abs().max()) <= 1.5 * err_dense_bf16 + 1e-5 * scale, k


def test_rows_only_forward_with_bf16_stored_rows(monkeypatch):
    """--agg_dtype bf16: the last layer alone runs on the loss rows — its sum over the fp32 activations (closer to the fp32 result than the bf16-stored Z it
    replaces), the layers below as the bf16 path does.  Loss and gradients stay within what bf16 rows cost the all-rows step."""
    from gnn_tail_generalization_amd import trunk
    calls = []
    real = trunk._last_layer_on_loss_rows
    monkeypatch.setattr(trunk, '_last_layer_on_loss_rows', lambda *a, **k: (calls.append(k.get('below') is None), real(*a, **k))[1])
    loss_s, g_s, used_s = _step_grads('1', extra=['--agg_dtype=bf16'], rows_only=True)
    assert calls == [True] and used_s
    loss_d, g_d, _ = _step_grads('0', extra=['--agg_dtype=bf16'])
    loss_f, g_f, _ = _step_grads('0')                                      # fp32 rows, dense backward: the yardstick
    assert abs(loss_s - loss_f) <= 1.5 * abs(loss_d - loss_f) + 1e-5 * abs(loss_f)
    for k in g_d:
        err_dense_bf16 = float((g_d[k] - g_f[k]).norm())
        assert float((g_s[k] - g_f[k]).norm()) <= 1.5 * err_dense_bf16 + 1e-3 * float(g_f[k].norm()), k


def test_row_sparse_backward_two_layers_small_graph(monkeypatch):
    """BASELINE config 2's shape (S-pubmed: 19 717 nodes, 2 layers, structural embeddings): the plan has two levels, the second one dense by
    construction (the stage below the first layer needs all rows); taken at this size only under hipGraph replay, forced here."""
    from gnn_tail_generalization_amd import trunk
    monkeypatch.setattr(tuning.T, 'rowsparse_min_nodes', 0)
    loss_s, g_s, used_s = _step_grads('1', dataset='S-pubmed', se='111', layers=2)
    loss_d, g_d, used_d = _step_grads('0', dataset='S-pubmed', se='111', layers=2)
    assert used_s and not used_d and loss_s == loss_d
    for k in g_d:
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * float(g_d[k].abs().max()), k


@pytest.mark.parametrize('case', ['case_r_initialbn_h256_L3_train10', 'case_r_initialbn_h256_L3_train10_se111',
                                  'case_r_residual_h256_L3_train10', 'case_r_residual_h256_L3_train10_se111',      # (Residual, round 5: cumulative supports)
                                  'case_nr_h256_L3_train10', 'case_nr_h256_L3_train10_se111'])                    # (the non-residual stack, stack.py)
@pytest.mark.parametrize('rows_only', [False, True])
def test_row_sparse_backward_matches_the_unmodified_reference(case, rows_only, monkeypatch):
    """The reference's own gradients (goldens case_r_initialbn_h256_L3_train10[_se111]: hidden 256, 3 layers, 8 – 10 % train rows, without
    and with structural-embedding tables on every layer, generated from the unmodified reference by tests/golden/make_golden.py) against
    the product's fused trunk with the row-sparse backward switched on at this small size: supports S_0 and S_1 compact, then dense; without
    tables level 0 runs through the loss rows' side, with tables its table gradient is scattered from the compact level."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from conftest import load_golden
    from helpers import product_model
    from gnn_tail_generalization_amd import _lib, graph, ops, trunk
    monkeypatch.setattr(tuning.T, 'rowsparse_min_nodes', 0)
    monkeypatch.setattr(tuning.T, 'rows_only_min_nodes', 0)
    monkeypatch.setattr(tuning.T, 'fwd0_min_edges', 0)                  # (the source-side form at this small size too)
    monkeypatch.setenv('CB_LOSS_ROWS', '1')
    g = load_golden(case)
    args, model = product_model(g['cfg'], g['sd'], DEV)
    x, ei, y, mask = g['x'].to(DEV), g['edge_index'].to(DEV), g['y'].to(DEV), g['train_mask'].to(DEV)
    model.train()
    # rows_only: the last layer(s) on the loss rows' supports
    out = model.get_3_embs(x, ei, mask, loss_rows=mask, rows_only=rows_only).emb4classi_full
    loss = ops.nll_logsoftmax(out, y, mask)
    if model.se_reg_all is not None:                                  # trainer_node_classification.py:393-394
        loss = loss + args.se_reg * model.se_reg_all
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.load().cb_device_status() == 0
    plan = getattr(model.model.model._graph(ei), '_support_plan', None)
    assert plan is not None and plan.levels[0][1] is not None       # the backward ran on the plan, S_1 compact
    took_rows_only = rows_only      # (the residual trunks, with and without tables, and the non-residual stack)
    if took_rows_only:
        assert bool(torch.isnan(out.detach()[~mask]).any(dim=1).all())      # the rows nobody may read are poisoned (ops.unread_rows_fill)
        torch.testing.assert_close(out.detach()[mask].cpu(), g['train_out'][mask.cpu()], atol=1e-4, rtol=1e-4)
    else:
        torch.testing.assert_close(out.detach().cpu(), g['train_out'], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), g['train_loss'], atol=1e-4, rtol=1e-5)
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(g['grads'])
    for k, ref in g['grads'].items():
        torch.testing.assert_close(got[k].cpu(), ref, atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('loss_side', [True, False])
def test_row_sparse_backward_on_a_directed_multigraph(loss_side, monkeypatch):
    """A DIRECTED graph with repeated edges (the reverse orientation is a CSR of its own, plan.fwd0 is cut from the forward one): the
    row-sparse backward — both forms of level 0 — against the dense backward of the same model, dropout on."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from conftest import load_golden
    from helpers import product_model
    from gnn_tail_generalization_amd import _lib, graph, ops, trunk
    monkeypatch.setattr(tuning.T, 'rowsparse_min_nodes', 0)
    monkeypatch.setattr(tuning.T, 'fwd0_min_edges', 0)
    monkeypatch.setattr(tuning.T, 'rowsparse_loss_side', loss_side)
    g = load_golden('case_r_initialbn_h256_L3_train10')
    cfg = dict(g['cfg'], dropout=0.3)
    n = cfg['N_nodes']
    gen = torch.Generator().manual_seed(5)
    src = torch.randint(0, n, (6 * n,), generator=gen)
    dst = (torch.rand(6 * n, generator=gen) ** 3 * n).long()                       # skewed in-degrees
    ei = torch.cat([torch.stack([src, dst]), torch.stack([src[:200], dst[:200]]),    # + repeated edges
                    torch.arange(n).repeat(2, 1)], 1).to(DEV)                        # + self-loops: no zero in-degree
    x, y = g['x'].to(DEV), g['y'].to(DEV)
    mask = torch.zeros(n, dtype=torch.bool, device=DEV)
    perm = torch.randperm(n, generator=gen)
    mask[perm[perm >= 40][:24]] = True                                              # (no top hub among the loss rows: supports 24 / 94 / 256 of 400 rows)
    grads, losses, used = {}, {}, {}
    for flag in ('1', '0'):
        monkeypatch.setenv('CB_LOSS_ROWS', flag)
        args, model = product_model(cfg, g['sd'], DEV)
        model.train()
        ops._seed_override[:] = [21, 22, 23, 24, 25]
        try:
            out = model.get_3_embs(x, ei, mask, loss_rows=mask).emb4classi_full
            loss = ops.nll_logsoftmax(out, y, mask)
            loss.backward()
        finally:
            ops._seed_override[:] = []
        torch.cuda.synchronize()
        assert _lib.load().cb_device_status() == 0
        graph = model.model.model._graph(ei)
        assert not graph.symmetric
        plan = getattr(graph, '_support_plan', None)
        used[flag] = plan is not None
        if flag == '1':
            assert plan is not None and plan.levels[0][1] is not None and plan.fwd0 is not None
        losses[flag] = float(loss.detach())
        grads[flag] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert used == {'1': True, '0': False} and losses['1'] == losses['0'] and set(grads['1']) == set(grads['0'])
    for k, ref in grads['0'].items():
        assert float((grads['1'][k] - ref).abs().max()) <= 5e-6 * float(ref.abs().max()), k


def test_row_sparse_backward_at_the_headline_size():
    """S-pl10M (10^7 nodes, 10^8 edges): one training step's gradients, row-sparse (supports 10 % / 45 % compact, then dense) against dense."""
    import gc
    loss_s, g_s, used_s = _step_grads('1', dataset='S-pl10M')
    gc.collect()
    torch.cuda.empty_cache()
    loss_d, g_d, used_d = _step_grads('0', dataset='S-pl10M')
    gc.collect()
    torch.cuda.empty_cache()
    assert used_s and not used_d and loss_s == loss_d
    for k in g_d:
        assert float((g_s[k] - g_d[k]).abs().max()) <= 5e-6 * float(g_d[k].abs().max()), k
    del g_s
    loss_r, g_r, used_r = _step_grads('1', dataset='S-pl10M', rows_only=True)      # + the rows-only forward (the default of the trainer's step)
    gc.collect()
    torch.cuda.empty_cache()
    assert used_r and abs(loss_r - loss_d) <= 2e-6 * abs(loss_d)
    for k in g_d:
        _close_up_to_relu_flips(g_r[k], g_d[k], k)


def test_support_plan_levels_are_the_reverse_graph_restricted_and_renumbered(monkeypatch):
    from gnn_tail_generalization_amd import graph
    monkeypatch.setattr(tuning.T, 'fwd0_min_edges', 0)
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=70000)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    keep = data.train_mask
    plan = G.grad_support_plan(keep, 3, max_frac=0.6)
    assert plan is G.grad_support_plan(keep, 3, max_frac=0.6)                    # cached per mask
    rp, col = G.rowptr_t.long(), G.col_t[:G.E].long()
    rows = torch.repeat_interleave(torch.arange(G.N, device=DEV), rp[1:] - rp[:-1])
    assert torch.equal(plan.space0.idx, keep.nonzero().flatten())
    assert plan.levels[-1][1] is None and len(plan.levels) <= 3                   # the plan ends with a dense destination
    # the forward orientation on the loss rows: (A (a * X))[S_0] with the factor of a source row applied as it is gathered
    assert plan.fwd0 is not None and plan.fwd0.N == plan.space0.n and plan.fwd0.n_cols == G.N
    xf = torch.randn(G.N, 256, device=DEV)
    want = G.spmm(xf * G.norm_out.unsqueeze(1))[plan.space0.idx]
    got = plan.fwd0.spmm(xf, col_scale=G.norm_out)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    got_rs = plan.fwd0.spmm(xf, col_scale=G.norm_out, row_scale=plan.space0.a)
    assert float((got_rs - want * plan.space0.a.unsqueeze(1)).abs().max()) <= 2e-6 * float(want.abs().max())
    with pytest.raises(ValueError):
        plan.fwd0.spmm(xf[:, :64].contiguous(), col_scale=G.norm_out)
    src_mask, src, h_full = keep, plan.space0, None
    h_full = torch.randn(G.N, 256, device=DEV) * keep.float().unsqueeze(1)       # a matrix supported on S_0
    h_c = h_full[src.idx].contiguous()
    for csr, dst in plan.levels:
        m = src_mask[col]
        want_mask = torch.zeros(G.N, dtype=torch.bool, device=DEV)
        want_mask[rows[m]] = True
        full = G.spmm(h_full, transpose=True)                                     # the dense orientation on the zero-padded matrix
        part = csr.spmm(h_c)
        if dst is None:
            assert csr.N == G.N
            got = part
        else:
            assert torch.equal(dst.idx, want_mask.nonzero().flatten()) and csr.N == dst.n
            assert bool((dst.pos[dst.idx] == torch.arange(dst.n, device=DEV, dtype=torch.int32)).all()) and int((dst.pos < 0).sum()) == G.N - dst.n
            assert bool((full[~want_mask] == 0).all())                            # nothing outside the support
            got = torch.zeros_like(full)
            got[dst.idx] = part
        assert float((got - full).abs().max()) <= 2e-6 * float(full.abs().max())
        if dst is None:
            break
        src_mask, src = want_mask, dst
        h_full = full
        h_c = part


def test_expand_rows_is_the_inverse_of_the_row_pack():
    from gnn_tail_generalization_amd import ops
    n = 10007
    mask = torch.rand(n, device=DEV) < 0.3
    pos = torch.where(mask, torch.cumsum(mask, 0, dtype=torch.int32) - 1, torch.full((n,), -1, dtype=torch.int32, device=DEV))
    src = torch.randn(int(mask.sum()), 256, device=DEV)
    out = ops.expand_rows(src, pos)
    want = torch.zeros(n, 256, device=DEV)
    want[mask] = src
    assert torch.equal(out, want)
    assert torch.equal(ops.expand_rows(src[:0], torch.full((5,), -1, dtype=torch.int32, device=DEV)), torch.zeros(5, 256, device=DEV))
    poisoned = ops.expand_rows(src, pos, fill=float('nan'))      # the logits of a rows-only forward: NaN where nobody may read
    assert torch.equal(poisoned[mask], src) and bool(torch.isnan(poisoned[~mask]).all())


def test_a_new_mask_every_step_falls_back_to_the_dense_backward():
    """The supports are built once per (graph, mask): a caller that changes the loss rows every step would pay the build every step, so
    after four builds without re-use the graph reports that planning does not pay and the trunk takes the dense backward."""
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=70000)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    same = data.train_mask
    for _ in range(20):                                    # one mask, re-used: pays
        G.grad_support_plan(same, 3)
    assert G.support_plan_pays() and G._support_builds == 1
    G2 = CSRGraph(data.edge_index, data.x.shape[0])
    for i in range(4):                                     # a new mask each time
        assert G2.support_plan_pays()
        G2.grad_support_plan(torch.rand(G2.N, device=DEV) < 0.1, 3)
    assert not G2.support_plan_pays()


def test_a_trainer_whose_mask_changes_every_step_keeps_stepping(monkeypatch):
    """The rows-only forward commits its backward to the plan it was evaluated on: the build / hit bookkeeping (support_plan_pays) is asked ONCE per step,
    in the forward — the build it triggers may be the one that tips the balance.  Six steps with a fresh train mask each: the first four take the plan
    (forward and backward alike), the rest run dense; no step fails, every loss is finite."""
    from gnn_tail_generalization_amd import trunk
    monkeypatch.setattr(tuning.T, 'rowsparse_min_nodes', 0)
    taken = []
    real = trunk._last_layer_on_loss_rows
    monkeypatch.setattr(trunk, '_last_layer_on_loss_rows', lambda *a, **k: (taken.append(1), real(*a, **k))[1])
    import bench
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    args = bench.make_args('S-arxiv', ['--manual_assign_GPU=0'])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    g = torch.Generator(device='cpu').manual_seed(1)
    losses = []
    for _ in range(6):
        m = (torch.rand(t.data.train_mask.shape[0], generator=g) < 0.1).to(t.data.train_mask.device)
        t.data.train_mask, t.data.test_mask, t._n_train = m, ~m, None
        losses.append(float(t.train_step()))
    assert all(l == l and abs(l) < 1e3 for l in losses) and 1 <= len(taken) <= 4, (losses, taken)


def test_violated_claim_is_reported_not_silent():
    from gnn_tail_generalization_amd import _lib, ops
    lib = _lib.load()
    g = torch.randn(5000, 40, device=DEV)
    mask = torch.rand(5000, device=DEV) < 0.2
    ok = g * mask.float().unsqueeze(1)
    ops.check_rows_zero(ok, mask)
    torch.cuda.synchronize()
    assert lib.cb_device_status() == 0
    bad = ok.clone()
    row = int((~mask).nonzero()[7])
    bad[row, 3] = 1e-30
    ops.check_rows_zero(bad, mask)
    torch.cuda.synchronize()
    assert int(_lib.grad_guard(DEV).item()) == 1        # ... and the word the optimiser launch of such a step looks at is set
    with pytest.raises(_lib.HipExtensionError, match=f'row {row}'):
        _lib.device_status()
    assert lib.cb_device_status() == 0                  # reported once, then cleared — the guard with it
    assert int(_lib.grad_guard(DEV).item()) == 0


def _small_trainer(dataset='S-pl1M'):
    import bench
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    args = bench.make_args(dataset, ['--manual_assign_GPU=0'])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    t.teacherGNN.train()
    if getattr(t, '_n_train', None) is None:
        t._n_train = int(t.data.train_mask.sum().item())
    return t


def test_without_the_promise_the_backward_is_dense(monkeypatch):
    """The row hint is an ARGUMENT of the forward (loss_rows=...), not something the backward finds out: a caller that builds another
    objective on the logits — here the masked loss plus a dense term — simply does not pass it and gets the dense backward, whose gradients
    equal those of the same objective with CB_LOSS_ROWS=0 bit for bit (it IS the same code path); no plan is built, nothing is checked."""
    from gnn_tail_generalization_amd import _lib, ops
    monkeypatch.setenv('CB_LOSS_ROWS', '1')
    grads = []
    for flag in ('1', '0'):
        monkeypatch.setenv('CB_LOSS_ROWS', flag)
        t = _small_trainer()
        ops._seed_override[:] = [11, 12, 13, 14, 15]
        out = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index).emb4classi_full            # the reference's call: no loss_rows
        loss = ops.nll_logsoftmax(out, t.data.y, t.data.train_mask, t._n_train) + 1e-3 * out.square().mean()
        loss.backward()
        ops._seed_override[:] = []
        torch.cuda.synchronize()
        _lib.device_status()
        assert getattr(t.graph(), '_support_plan', None) is None
        grads.append({k: p.grad.detach().clone() for k, p in t.teacherGNN.named_parameters() if p.grad is not None})
        del t
    assert grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_a_broken_promise_raises_and_never_reaches_the_weights(monkeypatch):
    """loss_rows handed to the forward, then a DENSE term added to the objective: the device-side check of the backward fails.  The error
    word reports it (the trainers raise where they read the loss) and the guard word keeps the fused Adam launch that follows on the
    stream from writing — parameters and moments are bit for bit what they were (the truncated gradients must never be applied)."""
    from gnn_tail_generalization_amd import _lib, ops
    monkeypatch.setenv('CB_LOSS_ROWS', '1')
    t = _small_trainer()
    t.train_step()                                               # one good step: moments exist
    torch.cuda.synchronize()
    _lib.device_status()
    before = {k: v.detach().clone() for k, v in t.teacherGNN.state_dict().items()}
    mom = [{k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for st in t.optimizer.state.values()]
    out = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, loss_rows=(t.data.train_mask, t._n_train)).emb4classi_full
    loss = ops.nll_logsoftmax(out, t.data.y, t.data.train_mask, t._n_train) + 1e-3 * out.square().mean()      # ... and breaks the promise
    steps_before = {id(p): st['step'] for p, st in t.optimizer.state.items()}
    t.optimizer.zero_grad()
    loss.backward()
    t.optimizer.step()
    torch.cuda.synchronize()
    with pytest.raises(_lib.HipExtensionError, match='outside the loss rows'):
        _lib.device_status()
    # ... and the skipped launch is not a step: the counts behind the bias corrections are what they were
    assert {id(p): st['step'] for p, st in t.optimizer.state.items()} == steps_before
    for k, v in t.teacherGNN.state_dict().items():
        assert torch.equal(v, before[k]), k
    for st, old in zip(t.optimizer.state.values(), mom):
        for k, v in old.items():
            assert torch.equal(st[k], v), k
    # the error is handled: the guard is clear again and the next (honest) step updates
    assert int(_lib.grad_guard(DEV).item()) == 0
    t.train_step()
    torch.cuda.synchronize()
    _lib.device_status()
    assert any(not torch.equal(v, before[k]) for k, v in t.teacherGNN.state_dict().items())

