"""GPU: the row-sparse first reverse aggregation of the backward (trunk.py, CSRGraph.filtered_t, ops.take_grad_rows / check_rows_zero).
The masked loss of trainer_node_classification.py:390-391 has a gradient that is zero in every row outside the train rows; the backward
of the last trunk layer gathers the train rows only.  Gradients equal the dense backward's up to the order in which a hub row's chunks are
summed, the claim is verified on the device, and a violation is reported instead of training on."""
import contextlib
import io
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step_grads(flag, dataset='S-pl1M'):
    import bench
    from gnn_tail_generalization_amd import _lib, ops
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    old = os.environ.get('CB_LOSS_ROWS')
    os.environ['CB_LOSS_ROWS'] = flag
    try:
        args = bench.make_args(dataset, ['--manual_assign_GPU=0'])
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            t = tnc.trainer(args, 0)
            t.setup_teacherGNN()
        t.teacherGNN.train()
        ops._seed_override[:] = [11, 12, 13, 14, 15]
        loss = t.training_loss()
        loss.backward()
        ops._seed_override[:] = []
        torch.cuda.synchronize()
        assert _lib.load().cb_device_status() == 0
        used = getattr(t.graph(), '_filtered', None) is not None
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in t.teacherGNN.named_parameters() if p.grad is not None}, used
    finally:
        ops._seed_override[:] = []
        if old is None:
            os.environ.pop('CB_LOSS_ROWS', None)
        else:
            os.environ['CB_LOSS_ROWS'] = old


@pytest.mark.parametrize('compact', ['1', '0'])
def test_row_sparse_backward_equals_the_dense_backward(compact, monkeypatch):
    """compact = 1: head, store backward, aggregation source and the input stage's operand on the loss rows alone ([n_loss, .] matrices);
    compact = 0: dense rows, only the aggregation's gather is restricted."""
    monkeypatch.setenv('CB_LOSS_ROWS_COMPACT', compact)
    loss_s, g_s, used_s = _step_grads('1')
    loss_d, g_d, used_d = _step_grads('0')
    assert used_s and not used_d                       # the 10 % train mask of the stand-in: the filtered orientation was built and used
    assert loss_s == loss_d
    assert set(g_s) == set(g_d)
    for k in g_d:
        scale = float(g_d[k].abs().max())
        # same addends; only the association of a hub row's chunk sums differs (fp32 rounding of 10^2-term sums)
        assert float((g_s[k] - g_d[k]).abs().max()) <= 2e-6 * scale, k


def test_filtered_orientation_is_the_reverse_graph_restricted_to_the_kept_rows():
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=70000)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    keep = data.train_mask
    sub = G.filtered_t(keep)
    assert sub is G.filtered_t(keep)                    # cached per mask
    rp, col = G.rowptr_t.long(), G.col_t[:G.E].long()
    rows = torch.repeat_interleave(torch.arange(G.N, device=DEV), rp[1:] - rp[:-1])
    m = keep[col]
    assert torch.equal(sub.col[:sub.E].long(), col[m])
    assert torch.equal(sub.rowptr.long(), torch.cat([rows.new_zeros(1), torch.cumsum(torch.bincount(rows[m], minlength=G.N), 0)]))
    # sums over it == sums of the full orientation over a matrix that is zero outside the kept rows: bit for bit where no hub chunking
    # is involved, to fp32 rounding of the chunk sums on hub rows
    h = torch.randn(G.N, 256, device=DEV) * keep.float().unsqueeze(1)
    full, part = G.spmm(h, transpose=True), sub.spmm(h)
    deg = (rp[1:] - rp[:-1])
    small = deg <= G.hub_threshold
    assert torch.equal(part[small], full[small])
    assert float((part - full).abs().max()) <= 2e-6 * float(full.abs().max())


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
    assert lib.cb_device_status() != 0
    assert f'row {row}' in lib.cb_last_error().decode()
    assert lib.cb_device_status() == 0                  # reported once, then cleared


def test_the_note_belongs_to_the_loss_gradient_buffer_only():
    from gnn_tail_generalization_amd import ops
    logits = torch.randn(3000, 7, device=DEV, requires_grad=True)
    y = torch.randint(0, 7, (3000,), device=DEV)
    mask = torch.rand(3000, device=DEV) < 0.3
    seen = []
    logits.register_hook(lambda gr: seen.append((ops.take_grad_rows(torch.empty_like(gr)), )))      # some other buffer: no hint, note consumed
    ops.nll_logsoftmax(logits, y, mask).backward()
    assert seen == [(None,)] and ops._GRAD_ROWS == []
    logits2 = torch.randn(3000, 7, device=DEV, requires_grad=True)
    got = []
    logits2.register_hook(lambda gr: got.append(ops.take_grad_rows(gr)))
    ops.nll_logsoftmax(logits2, y, mask).backward()
    assert got[0] is not None and got[0][0] is not None and got[0][1] == int(mask.sum())
    assert bool((logits2.grad[~mask] == 0).all())
