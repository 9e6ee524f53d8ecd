"""GPU: the trunk's input stage without a pass of its own (round 6).  The reverse aggregation that applies layer 0's store backward folds every mix
gradient into one matrix (cb_spmm_csr_store_bwd_mix_f32) and the input Linear's weight gradient computes (X0 > 0) * (dropout_bwd(g) + fold) while it
stages it (cb_gemm_tn_instage_f32) — autograd of GCN.py:104-110 and res_tricks.py:19-23.  Both kernels against plain torch formulas built from the
product's own keep-masks (cb_dropout_f32 of ones), and the fused step against the step with the separate pass (CB_INSTAGE_FOLD=0)."""
import os

import pytest
import torch

from gnn_tail_generalization_amd import tuning

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _mask_words(active):
    """[rows, 256] bool -> int64 [rows, 1, 4]: word k, bit l <-> column 4 l + k (the layout of the fused stores' mask words)."""
    rows = active.shape[0]
    a = active.view(rows, 64, 4).permute(0, 2, 1).to(torch.int64)                    # [rows, k, l]
    w = (a << torch.arange(64, device=active.device, dtype=torch.int64)).sum(-1)     # bit 63 wraps into the sign: the same 64 bits
    return w.view(rows, 1, 4).contiguous()


def _keep(shape, p, seed, offset=0):
    from gnn_tail_generalization_amd import ops
    return ops._dropout_raw(torch.ones(shape, device=DEV), p, seed, offset)         # keep ? 1 / (1 - p) : 0


@pytest.mark.parametrize('rows,feats', [(300000, 128), (262144 + 77, 100)])
def test_weight_gradient_with_the_input_stage_in_its_staging(rows, feats):
    from gnn_tail_generalization_amd import gemm
    torch.manual_seed(3)
    g = torch.randn(rows, 256, device=DEV)
    mfold = torch.randn(rows, 256, device=DEV) * 0.3
    x = torch.rand(rows, feats, device=DEV)
    active = torch.rand(rows, 256, device=DEV) < 0.55
    bits = _mask_words(active)
    p, sg, sx, row0 = 0.1, 1234567, 7654321, 5
    assert gemm.mm_tn_instage_supported(g, x, rows)
    dw, db = gemm.mm_tn_instage(g, mfold, bits, x, p, sg, p, sx, row0)
    gy = torch.where(active, g * _keep((rows, 256), p, sg, row0 * 256) + mfold, torch.zeros((), device=DEV))
    xd = x * _keep((rows, feats), p, sx, row0 * feats)
    want_dw = (gy.double().t() @ xd.double())
    want_db = gy.double().sum(0)
    # (fp32 accumulation over 3 * 10^5 rows: relative to the size of the sums' terms)
    scale_w = float((gy.abs().double().t() @ xd.abs().double()).max())
    assert float((dw.double() - want_dw).abs().max()) <= 2e-6 * scale_w
    assert float((db.double() - want_db).abs().max()) <= 2e-6 * float(gy.abs().double().sum(0).max())
    # the separate pass + the plain weight gradient give the same numbers up to the order of the sums
    ref = gemm.mm_tn_gdrop(gy, x, p, sx, row0)
    assert float((dw - ref).abs().max()) <= 2e-6 * scale_w
    assert not gemm.mm_tn_instage_supported(g, x[:, :64].contiguous(), rows)         # (the wide tile exists for 64 < F <= 128)
    assert not gemm.mm_tn_instage_supported(g[:1000], x[:1000], 1000)                # (and for >= 256 row slabs)


@pytest.mark.parametrize('n_mix', [0, 1, 2])
def test_reverse_aggregation_folds_the_mix_gradients(n_mix):
    from gnn_tail_generalization_amd import graph as G
    from gnn_tail_generalization_amd.data import synthetic_data
    data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=60000)
    g = G.build_graph(data.edge_index, 60000)
    n = g.N
    assert g._plan.n_hubs > 0                                                       # (the hub rows' epilogue runs in k_spmm_hub_finish)
    torch.manual_seed(5)
    h = torch.randn(n, 256, device=DEV)
    active = torch.rand(n, 256, device=DEV) < 0.5
    bits = _mask_words(active)
    p, seed, row0, alpha = 0.1, 424242, 3, 0.1
    ops_, pos_, seeds_ = [], [], []
    for q in range(n_mix):
        member = torch.rand(n, device=DEV) < (0.1 if q == 0 else 0.45)
        pos = torch.where(member, torch.cumsum(member, 0, dtype=torch.int32) - 1, torch.full((n,), -1, dtype=torch.int32, device=DEV))
        ops_.append(torch.randn(int(member.sum()), 256, device=DEV))
        pos_.append(pos.contiguous())
        seeds_.append(1000 + q)
    g_raw, gr = g.spmm_store_bwd(h, g.norm_out, bits, g.norm_in, 1 - alpha, p, seed, row0)
    m, gr2, db = g.spmm_store_bwd(h, g.norm_out, bits, g.norm_in, 1 - alpha, p, seed, row0, mix=(ops_, pos_, seeds_, alpha, True))
    assert torch.equal(gr, gr2)                                                     # the store backward itself is untouched
    want = g_raw * _keep((n, 256), p, seed, row0 * 256)
    for t, pos, sd in zip(ops_, pos_, seeds_):
        full = torch.zeros(n, 256, device=DEV)
        full[pos >= 0] = t
        want = want + full * _keep((n, 256), p, sd, row0 * 256)
    want = alpha * want
    assert float((m - want).abs().max()) <= 2e-6 * float(want.abs().max())
    want_db = (gr.double() / g.norm_in.double().unsqueeze(1)).sum(0)
    assert float((db.double() - want_db).abs().max()) <= 2e-5 * float((gr.abs().double() / g.norm_in.double().unsqueeze(1)).sum(0).max())
    m3, _, none = g.spmm_store_bwd(h, g.norm_out, bits, g.norm_in, 1 - alpha, p, seed, row0, mix=(ops_, pos_, seeds_, alpha, False))
    assert none is None and torch.equal(m3, m)


@pytest.mark.parametrize('dense_ops', [0, 1])
def test_store_backward_pass_folds_the_mix_gradients(dense_ops):
    """cb_trunk_layer_bwd_fold_f32 = cb_trunk_layer_bwd_f32 (same out, same column sums) + the folded mix gradients, with compact and dense operands."""
    from gnn_tail_generalization_amd import trunk
    torch.manual_seed(11)
    n, p, seed, row0, alpha = 200003, 0.1, 777, 4, 0.1
    g = torch.randn(n, 256, device=DEV)
    bits = _mask_words(torch.rand(n, 256, device=DEV) < 0.5)
    scale = torch.rand(n, device=DEV) + 0.5
    ops_, pos_, seeds_ = [], [], []
    for q in range(2):
        if q < dense_ops:
            ops_.append(torch.randn(n, 256, device=DEV)); pos_.append(None)
        else:
            member = torch.rand(n, device=DEV) < (0.1 if q == 0 else 0.45)
            pos_.append(torch.where(member, torch.cumsum(member, 0, dtype=torch.int32) - 1, torch.full((n,), -1, dtype=torch.int32, device=DEV)).contiguous())
            ops_.append(torch.randn(int(member.sum()), 256, device=DEV))
        seeds_.append(2000 + q)
    gr0, db0 = trunk._layer_bwd(g, bits, scale, None, False, p, seed, row0, 1 - alpha, alpha, True)
    gr1, db1, m = trunk._layer_bwd_fold(g, bits, scale, p, seed, row0, 1 - alpha, alpha, True, ops_, pos_, seeds_)
    assert torch.equal(gr0, gr1) and torch.equal(db0, db1)
    want = g * _keep((n, 256), p, seed, row0 * 256)
    for t, pos, sd in zip(ops_, pos_, seeds_):
        full = t if pos is None else torch.zeros(n, 256, device=DEV).index_copy_(0, torch.nonzero(pos >= 0).flatten(), t)
        want = want + full * _keep((n, 256), p, sd, row0 * 256)
    want = alpha * want
    assert float((m - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # the second column sum: operand 1 under its own dropout, through another store's mask words (what cb_trunk_input_bwd_multi_cs_f32 returns)
    active2 = torch.rand(n, 256, device=DEV) < 0.4
    res = trunk._layer_bwd_fold(g, bits, scale, p, seed, row0, 1 - alpha, alpha, True, ops_, pos_, seeds_, cs=(1, _mask_words(active2), 0.9))
    assert torch.equal(res[0], gr0) and torch.equal(res[2], m)
    full1 = ops_[1] if pos_[1] is None else torch.zeros(n, 256, device=DEV).index_copy_(0, torch.nonzero(pos_[1] >= 0).flatten(), ops_[1])
    term = torch.where(active2, 0.9 * full1 * _keep((n, 256), p, seeds_[1], row0 * 256), torch.zeros((), device=DEV))
    assert float((res[3].double() - term.double().sum(0)).abs().max()) <= 2e-6 * float(term.abs().double().sum(0).max())


@pytest.mark.parametrize('n_loss_rows,sum_first', [(None, True), (30000, True), (None, False)])
def test_training_step_with_the_folded_input_stage(n_loss_rows, sum_first, monkeypatch):
    """The trainer's step (rows-only forward, row-sparse backward, L = 3: level 1 writes all rows and carries layer 0's store backward) with the fold
    against the same step with the separate input-stage pass: the same loss bit for bit, every gradient above the input stage bit for bit, the three
    the fold touches (layer 0's bias, the input Linear's weight and bias) up to the order of their sums."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_rowsparse import _step_grads
    from gnn_tail_generalization_amd import gemm
    # sum_first: level 1 runs through its source rows' side and carries layer 0's store backward in its epilogue, which folds (cb_spmm_csr_store_bwd_mix_f32);
    # else (S-pl1M's own form: it sits below the break-even of the sum-first layer) the store backward is a pass, which folds (cb_trunk_layer_bwd_fold_f32)
    if sum_first:
        monkeypatch.setattr(tuning.T, 'sum_first_below_min_edges', 0)
    monkeypatch.setenv('CB_SPMM_STORE_BWD', '1')
    calls = []
    real = gemm.mm_tn_instage
    monkeypatch.setattr(gemm, 'mm_tn_instage', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setenv('CB_INSTAGE_FOLD', '1')
    loss_f, g_f, used_f = _step_grads('1', n_loss_rows=n_loss_rows, rows_only=True)
    assert calls == [1] and used_f
    monkeypatch.setenv('CB_INSTAGE_FOLD', '0')
    loss_p, g_p, used_p = _step_grads('1', n_loss_rows=n_loss_rows, rows_only=True)
    assert calls == [1] and used_p
    assert loss_f == loss_p and set(g_f) == set(g_p)
    touched = ('layers_MLP.0.weight', 'layers_MLP.0.bias', 'layers_GCN.0.bias')
    for k in g_p:
        if k.endswith(touched):
            assert float((g_f[k] - g_p[k]).abs().max()) <= 1e-5 * float(g_p[k].abs().max()) + 1e-9, k
        else:
            assert torch.equal(g_f[k], g_p[k]), k


def test_loss_on_the_compact_logits_of_the_rows_only_forward(monkeypatch):
    """The rows-only forward hands the train rows' logits over as the compact matrix they were computed as (res.emb4classi_rows == emb4classi_full[mask]:
    the reference's raw_logits, GNN_normalizations.py:45-47); the trainer's loss on it equals the masked loss over [N, C] and every gradient is the same
    bit for bit (the gradient reaches the backward's head compact instead of being gathered from an [N, C] matrix)."""
    import contextlib
    import io
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench
    from test_gpu_rowsparse import _step_grads
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    monkeypatch.setenv('CB_COMPACT_LOSS', '1')
    loss_c, g_c, used_c = _step_grads('1', rows_only=True)
    monkeypatch.setenv('CB_COMPACT_LOSS', '0')
    loss_f, g_f, used_f = _step_grads('1', rows_only=True)
    assert used_c and used_f and abs(loss_c - loss_f) <= 1e-6 * abs(loss_f) and set(g_c) == set(g_f)
    for k in g_f:
        assert torch.equal(g_c[k], g_f[k]), k
    # what the caller sees: the compact logits are the rows of the full output, and a mask= argument that IS the loss mask gets them without a gather
    args = bench.make_args('S-pl1M', ['--manual_assign_GPU=0'])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    t.teacherGNN.train()
    mask, n = t.data.train_mask, int(t.data.train_mask.sum())
    y_rows = t.data.y[mask].contiguous()
    grads = {}
    for both in (True, False):
        ops._seed_override[:] = [31, 32, 33, 34, 35]
        res = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, mask, loss_rows=(mask, n), rows_only=True)
        ops._seed_override[:] = []
        assert res.emb4classi_rows is not None and res.emb4classi_rows[1] is mask and res.emb4classi is res.emb4classi_rows[0]
        assert torch.equal(res.emb4classi.detach(), res.emb4classi_full.detach()[mask])
        # a loss that uses BOTH outputs of the forward: the two gradients add up on the loss rows (== twice the compact one alone)
        loss = ops.nll_logsoftmax(res.emb4classi, y_rows, None, n)
        loss = loss + (ops.nll_logsoftmax(res.emb4classi_full, t.data.y, mask, n) if both else loss)
        t.teacherGNN.zero_grad()
        loss.backward()
        grads[both] = {k: p.grad.detach().clone() for k, p in t.teacherGNN.named_parameters() if p.grad is not None}
    for k in grads[True]:
        assert float((grads[True][k] - grads[False][k]).abs().max()) <= 2e-6 * float(grads[False][k].abs().max()) + 1e-10, k
    res = t.teacherGNN.get_3_embs(t.data.x, t.data.edge_index, loss_rows=(mask, n), rows_only=False)
    assert res.emb4classi_rows is None      # (every row evaluated: nothing compact to hand over)


def test_hub_rows_come_back_in_ascending_order():
    """The hub plan lists its rows in ascending order (ordered compaction, no atomic cursor): which rows share a block of the hub-finish kernel — and with
    it the order of the column sums cb_spmm_csr_store_bwd_mix_f32 takes there — is the same in every process (found as a last-bit flicker of layer 0's
    bias gradient between runs)."""
    from gnn_tail_generalization_amd import graph as G
    from gnn_tail_generalization_amd.data import synthetic_data
    data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=200000)
    g = G.build_graph(data.edge_index, 200000)
    p = g._plan
    assert p.n_hubs > 1
    rows = p.hub_rows[:p.n_hubs].long()
    assert bool((rows[1:] > rows[:-1]).all())
    deg = g.rowptr[1:] - g.rowptr[:-1]
    assert torch.equal(rows, torch.nonzero(deg > g.hub_threshold).flatten())
    chunks = (deg[rows] + g.hub_threshold - 1) // g.hub_threshold
    assert torch.equal(p.hub_chunk_ptr.long(), torch.cat([chunks.new_zeros(1), torch.cumsum(chunks, 0)]).long()) and int(p.hub_chunk_ptr[-1]) == p.n_chunks
