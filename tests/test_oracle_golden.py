"""CPU: the oracle (oracle/coldbrew_oracle.py) against the golden vectors produced by the
unmodified reference (tests/golden/make_golden.py).  Tolerances: fp32, 2e-5 abs+rel."""
import numpy as np
import pytest
import torch

import coldbrew_oracle as orc
from conftest import golden_cases, load_golden

ATOL = RTOL = 2e-5


def _cfg(g):
    c = dict(g['cfg'])
    return orc.make_cfg(**{k: c[k] for k in ['type_trick', 'num_layers', 'num_feats', 'dim_hidden', 'num_classes',
                                              'res_alpha', 'layer_agg', 'whetherHasSE', 'node_norm_type', 'num_groups',
                                              'skip_weight', 'se_reg', 'change_to_featureless', 'dim_learnable_input']})


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    cfg = _cfg(g)
    csr = orc.build_csr(g['edge_index'], g['cfg']['N_nodes'])
    if g.get('raises') == 'zero_in_degree':
        with pytest.raises(orc.ZeroInDegreeError):
            orc.teacher_forward(cfg, g['sd'], g['x'], csr)
        return
    sd = {k: v.clone() for k, v in g['sd'].items()}
    out, reg = orc.teacher_forward(cfg, sd, g['x'], csr, training=False)
    torch.testing.assert_close(out, g['eval_out'], atol=ATOL, rtol=RTOL)
    if g['se_reg_all'] is None:
        assert reg is None
    else:
        torch.testing.assert_close(reg, g['se_reg_all'], atol=1e-4, rtol=1e-5)
    # collect_SE (GCN.py:148-150)
    x_in = sd['embs'] if cfg.dim_learnable_input > 0 else (g['x'] * 0 if cfg.change_to_featureless else g['x'])
    _, _, les = orc.trickscomb_forward(cfg, orc.strip_prefix(sd), x_in, csr, training=False, want_les=True)
    torch.testing.assert_close(les, g['les'], atol=ATOL, rtol=RTOL)
    # train-mode loss and every parameter gradient (dropout = 0)
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    buffers = {}
    out, reg = orc.teacher_forward(cfg, sd, g['x'], csr, training=True, buffers_out=buffers)
    loss = orc.training_loss(cfg, out, reg, g['y'], g['train_mask'])
    torch.testing.assert_close(loss.detach(), g['train_loss'], atol=1e-4, rtol=1e-5)
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    got = {k: gr for k, gr in zip(names, grads) if gr is not None}
    assert set(got) == set(g['grads']), (sorted(set(got) ^ set(g['grads'])))
    for k, gr in g['grads'].items():
        torch.testing.assert_close(got[k], gr, atol=ATOL, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')
    for k, v in g['bn_after'].items():   # norm layers that never run keep their initial buffers
        got_buf = buffers.get(k[len('model.model.'):], g['sd'][k])
        torch.testing.assert_close(got_buf, v, atol=ATOL, rtol=RTOL)


def test_aggregate_against_dense_f64():
    """The DGL boundary is unpinned by the reference; pin the restated semantics to A^T.h."""
    g = load_golden('case_graph_asym_multi')
    csr = orc.build_csr(g['edge_index'])
    h = torch.randn(csr.N, 7, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    torch.testing.assert_close(orc.aggregate_sum(csr, h), orc.aggregate_sum_dense_f64(csr, h), atol=1e-12, rtol=1e-12)


def test_example_graph_known_answer():
    """utils.py:1096 example graph [[0,0,1,1,1,2],[0,1,0,1,2,2]]: hand-computed degrees and sums."""
    ei = torch.tensor([[0, 0, 1, 1, 1, 2], [0, 1, 0, 1, 2, 2]])
    csr = orc.build_csr(ei)
    assert csr.in_deg.tolist() == [2, 2, 2] and csr.out_deg.tolist() == [2, 3, 1]
    assert csr.rowptr.tolist() == [0, 2, 4, 6] and csr.col.tolist() == [0, 1, 0, 1, 1, 2]
    assert csr.rowptr_t.tolist() == [0, 2, 5, 6] and csr.col_t.tolist() == [0, 1, 0, 1, 2, 2]
    h = torch.tensor([[1.], [10.], [100.]])
    assert orc.aggregate_sum(csr, h).reshape(-1).tolist() == [11., 11., 110.]
    a, b = orc.degree_norms(csr)
    np.testing.assert_allclose(a.numpy(), np.array([2, 3, 1], dtype=np.float32) ** -0.5, rtol=1e-7)
    np.testing.assert_allclose(b.numpy(), np.array([2, 2, 2], dtype=np.float32) ** -0.5, rtol=1e-7)


@pytest.mark.parametrize('name', golden_cases('trainer_'))
def test_oracle_training_trajectory(name):
    """Rows a15-a17: K Adam steps reproduce the unmodified run_trainSet() losses."""
    g = load_golden(name)
    cfg = _cfg(g)
    csr = orc.build_csr(g['edge_index'], g['cfg']['N_nodes'])
    sd = {k: v.clone() for k, v in g['sd'].items()}
    losses = orc.train_steps(cfg, sd, g['x'], csr, g['y'], g['train_mask'], g['steps'], lr=0.01, weight_decay=5e-4)
    np.testing.assert_allclose(losses, g['trajectory'][:, 0].numpy(), rtol=1e-5)
    for k, v in g['sd_final'].items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(sd[k].detach(), v, atol=1e-5, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')
    out, _ = orc.teacher_forward(cfg, {k: v.detach() for k, v in sd.items()}, g['x'], csr, training=False)
    assert orc.evaluate(out, g['y'], g['train_mask']) == pytest.approx(float(g['trajectory'][-1, 1]))
    assert orc.evaluate(out, g['y'], ~g['train_mask']) == pytest.approx(float(g['trajectory'][-1, 2]))


def test_oracle_label_propagation_matches_reference():
    """§8f row 4: pure label propagation against the reference's outcome_correlation functions."""
    g = load_golden('lp_fixture')
    assert torch.equal(orc.to_undirected(g['edge_index'], g['y'].shape[0]), g['edge_index_undirected'])
    out, dis = orc.label_propagation(g['edge_index'], g['y'], g['train_mask'], g['num_classes'], g['alpha'], g['num_propagations'])
    torch.testing.assert_close(dis, g['deg_inv_sqrt'], atol=0, rtol=1e-6)
    torch.testing.assert_close(out, g['out'], atol=1e-5, rtol=1e-5)
    acc = [np.round(orc.evaluate(out, g['y'], m) * 100, 2) for m in (g['train_mask'], ~g['train_mask'])]
    assert acc == g['acc'].tolist()


def test_oracle_semlp_replacement_matches_reference():
    """§8f row 2: the SEMLP top-K "virtual neighbour" lookup against the unmodified reference method."""
    fx = load_golden('semlp_fixture')
    for name, g in fx.items():
        out, sel, w = orc.semlp_replacement(g['q'], g['teacher'], g['k'])
        torch.testing.assert_close(out, g['out'], atol=1e-6, rtol=1e-6, msg=lambda m, name=name: f'{name}: {m}')
        assert sel.shape == (g['q'].shape[0], g['k'])


def test_oracle_proj2class_head_matches_reference():
    """--has_proj2class=1: the oracle's trunk + head restatement against the unmodified reference (common embedding, logits, loss)."""
    from helpers import oracle_cfg
    fx = load_golden('proj2class_fixture')
    assert set(fx) == {'nr_se100', 'r_initial_se111'}
    for name, g in fx.items():
        cfg = dict(g['cfg'], num_classes=g['cfg']['dim_commonEmb'])        # the GNN's last layer emits the 128-wide common embedding
        csr = orc.build_csr(g['edge_index'], g['cfg']['N_nodes'])
        common, reg = orc.teacher_forward(oracle_cfg(cfg), g['sd'], g['x'], csr, training=False)
        torch.testing.assert_close(common, g['common'], atol=1e-5, rtol=1e-5, msg=lambda m: f'{name} common: {m}')
        logits = orc.proj2class_head(g['sd'], common)
        torch.testing.assert_close(logits, g['logits'], atol=1e-5, rtol=1e-5, msg=lambda m: f'{name} logits: {m}')
        loss = orc.training_loss(oracle_cfg(cfg), logits, reg, g['y'], g['train_mask'])
        torch.testing.assert_close(loss, g['loss'], atol=1e-5, rtol=1e-5)


def test_oracle_edge_weight_form_matches_reference():
    """GCNConv.forward(..., edge_weight=w) (GCN.py:199-202, u_mul_e): the oracle's restatement against the unmodified reference —
    output, regulariser and the gradients w.r.t. features, parameters and the edge weights themselves."""
    fx = load_golden('edge_weight_fixture')
    for name, g in fx.items():
        csr = orc.build_csr(g['edge_index'], g['n'])
        feat = g['feat'].clone().requires_grad_(True)
        w = g['edge_weight'].clone().requires_grad_(True)
        p = {k: v.clone().requires_grad_(True) for k, v in g['sd'].items()}
        out, reg = orc.gcnconv_forward(csr, feat, p['weight'], p['bias'], p.get('le'), edge_weight=w)
        torch.testing.assert_close(out, g['out'], atol=1e-5, rtol=1e-5, msg=lambda m: f'{name}: {m}')
        loss = (out * g['gout']).sum() + (0.5 * reg if reg is not None else 0.0)
        loss.backward()
        torch.testing.assert_close(feat.grad, g['d_feat'], atol=2e-5, rtol=1e-4)
        torch.testing.assert_close(w.grad, g['d_edge_weight'], atol=2e-5, rtol=1e-4)
        for k, v in g['grads'].items():
            torch.testing.assert_close(p[k].grad, v, atol=2e-5, rtol=1e-4, msg=lambda m, k=k: f'{name} {k}: {m}')
