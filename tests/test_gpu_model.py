"""GPU parity of the whole TeacherGNN path against the golden vectors of the unmodified reference:
eval logits within 1e-4 (north_star), se_reg_all, collect_SE, train-mode loss and every gradient."""
import pytest
import torch

from conftest import golden_cases, load_golden
from helpers import product_model

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('name', golden_cases())
def test_model_matches_reference_golden(name):
    from gnn_tail_generalization_amd.graph import ZeroInDegreeError
    g = load_golden(name)
    args, model = product_model(g['cfg'], g['sd'], DEV)
    x, ei = g['x'].to(DEV), g['edge_index'].to(DEV)
    if g.get('raises') == 'zero_in_degree':
        with pytest.raises(ZeroInDegreeError):
            model(x, ei)
        return
    model.eval()
    with torch.no_grad():
        out = model(x, ei)
        torch.testing.assert_close(out.cpu(), g['eval_out'], atol=1e-4, rtol=1e-4)
        if g['se_reg_all'] is None:
            assert model.se_reg_all is None
        else:
            torch.testing.assert_close(model.se_reg_all.cpu(), g['se_reg_all'], atol=1e-3, rtol=1e-5)
        xin = model.embs if args.dim_learnable_input > 0 else (x * 0 if args.TeacherGNN.change_to_featureless else x)
        les = model.model.model.collect_SE(xin, ei)
        torch.testing.assert_close(les.cpu(), g['les'], atol=1e-4, rtol=1e-4)
    model.train()
    mask, y = g['train_mask'].to(DEV), g['y'].to(DEV)
    res = model.get_3_embs(x, ei, mask)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(res.emb4classi, 1), y[mask])
    if model.se_reg_all is not None:
        loss = loss + args.se_reg * model.se_reg_all
    model.zero_grad()
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g['train_loss'], atol=1e-4, rtol=1e-5)
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(g['grads'])
    for k, ref in g['grads'].items():
        torch.testing.assert_close(got[k].cpu(), ref, atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')
    for k, v in g['bn_after'].items():
        torch.testing.assert_close(model.state_dict()[k].cpu(), v, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('name', ['case_nr_se111_L3', 'case_r_initialbn_se111_L3_powerlaw', 'case_r_residual_L3'])
def test_train_mode_with_dropout_matches_oracle(name):
    """Dropout cannot match torch's RNG stream; parity is defined by injecting the product's own
    keep-masks (pure functions of seed and index) into the oracle."""
    import coldbrew_oracle as orc
    from gnn_tail_generalization_amd import ops
    from helpers import oracle_cfg
    g = load_golden(name)
    cfg = dict(g['cfg'])
    cfg['dropout'] = 0.4
    args, model = product_model(cfg, g['sd'], DEV)
    x, ei = g['x'].to(DEV), g['edge_index'].to(DEV)
    n, L, H, C, F_ = cfg['N_nodes'], cfg['num_layers'], cfg['dim_hidden'], cfg['num_classes'], cfg['num_feats']
    residual = orc.has_residual_mlp(oracle_cfg(cfg))
    if residual:
        shapes = [(n, F_)] + [(n, H)] * L + [(n, H)]
    else:
        shapes = [(n, F_)] + [(n, H)] * (L - 1) + [(n, C)]
    seeds = [1000 + i for i in range(len(shapes))]
    ops._seed_override[:] = list(seeds)
    model.train()
    out = model(x, ei)
    assert not ops._seed_override
    masks = [ops.dropout_keep_mask(s, 0.4, sd, DEV).cpu() for s, sd in zip(shapes, seeds)]
    ocfg = oracle_cfg(cfg)
    ocfg.dropout = 0.4
    csr = orc.build_csr(g['edge_index'], n)
    ref, _ = orc.teacher_forward(ocfg, g['sd'], g['x'], csr, training=True, dropout_masks=masks)
    torch.testing.assert_close(out.detach().cpu(), ref, atol=1e-4, rtol=1e-4)


def _tiny_trunk_setup(L=3, se='111', n_override=None, extra=()):
    import contextlib
    import io
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    from gnn_tail_generalization_amd.utils import set_arch_configs
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-pl1M', '--use_special_split=0', f'--num_layers={L}', f'--whetherHasSE={se}',
                                            '--se_reg=0.5', '--manual_assign_GPU=0'] + list(extra))
    data = synthetic_data('S-pl1M', seed=2, device=DEV, n_override=n_override or 3000)
    args.N_nodes = data.x.shape[0]
    args.dropout = 0.3
    args.device = torch.device(DEV)
    set_arch_configs(args)
    torch.manual_seed(0)
    model = TeacherGNN(args).to(DEV)
    return args, model, data


@pytest.mark.parametrize('conn', ['InitialBatchNorm', 'Residual', 'ResidualPairNorm'])
@pytest.mark.parametrize('se', ['000', '111'])
@pytest.mark.parametrize('train', [True, False])
def test_fused_trunk_equals_modular_path(se, train, conn, monkeypatch):
    """The fused residual trunk (trunk.py) and the modular operator path compute the same logits, loss and
    gradients from the same parameters and the same dropout seeds (hidden = 256, hub rows present) — for the 'Initial' connection and for
    'Residual' (res_tricks.py:7-14: the mix source is the previous layer's ReLU output; WISCONSIN / CORNELL / TEXAS in the best-config table)."""
    from gnn_tail_generalization_amd import ops, trunk
    from gnn_tail_generalization_amd.GNN_model.GCN import TricksComb
    args, model, data = _tiny_trunk_setup(se=se, extra=() if conn == 'InitialBatchNorm' else ('--force_set_to_best_config=0', f'--type_trick={conn}'))
    assert model.model.model.type_trick == conn
    calls = []
    real = trunk._TrunkFn.apply
    monkeypatch.setattr(trunk._TrunkFn, 'apply', staticmethod(lambda *a, **k: (calls.append(a[1][7]), real(*a, **k))[1]))
    res = {}
    for fused in (True, False):
        TricksComb.use_fused_trunk = fused
        try:
            model.train(train)
            model.zero_grad()
            ops._seed_override[:] = list(range(900, 910)) if train else []
            out = model(data.x, data.edge_index)
            ops._seed_override[:] = []
            loss = ops.nll_logsoftmax(out, data.y, data.train_mask)
            if model.se_reg_all is not None:
                loss = loss + args.se_reg * model.se_reg_all
            loss.backward()
            res[fused] = (out.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            TricksComb.use_fused_trunk = True
    assert model.model.model.dglgraph._plan.n_hubs >= 0
    assert calls == [conn.startswith('Residual')]          # the fused node ran once (use_fused_trunk = True), with the connection asked for
    torch.testing.assert_close(res[True][0], res[False][0], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(res[True][1], res[False][1], atol=1e-6, rtol=1e-6)
    assert set(res[True][2]) == set(res[False][2])
    for k in res[True][2]:
        torch.testing.assert_close(res[True][2][k], res[False][2][k], atol=2e-6, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('L', [2, 3, 4])
@pytest.mark.parametrize('se', ['000', '111'])
@pytest.mark.parametrize('train', [True, False])
def test_fused_stack_equals_modular_path(se, train, L, monkeypatch):
    """The fused non-residual stack (stack.py: 'NoRes...' — Cora / Citeseer / ACTOR in the best-config table, base_options.py:416-421) and the
    modular operator path compute the same logits, loss and gradients from the same parameters and the same dropout seeds (hidden = 256,
    F -> H -> ... -> C, dropout on the logits, hub rows present; L = 4: two hidden layers in a row, i.e. the aggregation + GEMM kernel)."""
    from gnn_tail_generalization_amd import ops, stack
    from gnn_tail_generalization_amd.GNN_model.GCN import TricksComb
    args, model, data = _tiny_trunk_setup(L=L, se=se, extra=('--force_set_to_best_config=0', '--type_trick=NoResNodeNorm'))
    assert model.model.model.type_trick == 'NoResNodeNorm' and not model.model.model.has_residual_MLP
    calls = []
    real = stack._StackFn.apply
    monkeypatch.setattr(stack._StackFn, 'apply', staticmethod(lambda *a, **k: (calls.append(1), real(*a, **k))[1]))
    res = {}
    for fused in (True, False):
        TricksComb.use_fused_trunk = fused
        try:
            model.train(train)
            model.zero_grad()
            ops._seed_override[:] = list(range(900, 910)) if train else []
            out = model(data.x, data.edge_index)
            ops._seed_override[:] = []
            loss = ops.nll_logsoftmax(out, data.y, data.train_mask)
            if model.se_reg_all is not None:
                loss = loss + args.se_reg * model.se_reg_all
            loss.backward()
            res[fused] = (out.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            TricksComb.use_fused_trunk = True
    assert calls == [1]
    torch.testing.assert_close(res[True][0], res[False][0], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(res[True][1], res[False][1], atol=1e-6, rtol=1e-6)
    assert set(res[True][2]) == set(res[False][2])
    for k in res[True][2]:
        torch.testing.assert_close(res[True][2][k], res[False][2][k], atol=2e-6, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('name,which', [('case_r_residual_h256_L3_train10', 'trunk'), ('case_r_residual_h256_L3_train10_se111', 'trunk'),
                                        ('case_r_residual_h256_L2', 'trunk'), ('case_r_initialbn_h256_L3_train10', 'trunk'),
                                        ('case_nr_h256_L3_train10', 'stack'), ('case_nr_h256_L3_train10_se111', 'stack')])
def test_default_trunk_shapes_run_fused_against_the_reference(name, which, monkeypatch):
    """The reference's three default trunk shapes (base_options.py:416-421: Initial / Residual / NoRes) at hidden 256 go through the FUSED autograd
    nodes (trunk._TrunkFn / stack._StackFn), and what those compute is the unmodified reference's: logits, loss and every gradient of the goldens."""
    from gnn_tail_generalization_amd import ops, stack, trunk
    g = load_golden(name)
    args, model = product_model(g['cfg'], g['sd'], DEV)
    calls = []
    for mod, fn, tag in ((trunk, '_TrunkFn', 'trunk'), (stack, '_StackFn', 'stack')):
        real = getattr(mod, fn).apply
        monkeypatch.setattr(getattr(mod, fn), 'apply', staticmethod(lambda *a, _r=real, _t=tag, **k: (calls.append(_t), _r(*a, **k))[1]))
    x, ei, y, mask = g['x'].to(DEV), g['edge_index'].to(DEV), g['y'].to(DEV), g['train_mask'].to(DEV)
    model.eval()
    with torch.no_grad():
        torch.testing.assert_close(model(x, ei).cpu(), g['eval_out'], atol=1e-4, rtol=1e-4)
    model.train()
    out = model.get_3_embs(x, ei, mask).emb4classi_full
    loss = ops.nll_logsoftmax(out, y, mask)
    if model.se_reg_all is not None:
        loss = loss + args.se_reg * model.se_reg_all
    model.zero_grad()
    loss.backward()
    assert calls == [which, which]
    torch.testing.assert_close(out.detach().cpu(), g['train_out'], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), g['train_loss'], atol=1e-4, rtol=1e-5)
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(g['grads'])
    for k, ref in g['grads'].items():
        torch.testing.assert_close(got[k].cpu(), ref, atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('conn,L,se', [('InitialBatchNorm', 2, '111'), ('Residual', 3, '111'), ('Residual', 2, '000'), ('NoResNodeNorm', 3, '111'),
                                       ('NoResNodeNorm', 4, '000'), ('NoResNodeNorm', 2, '100')])
def test_fused_nodes_match_oracle_with_injected_masks(conn, L, se):
    """The fused nodes in TRAIN mode with dropout on against the ORACLE (not against the product's other path): the product's keep-masks
    (pure functions of seed and element index) are injected into the oracle's forward — 'Initial', 'Residual' (trunk.py) and the non-residual
    stack (stack.py: F -> H -> ... -> C, the last mask on the logits), hidden 256, with / without structural-embedding tables."""
    import coldbrew_oracle as orc
    from gnn_tail_generalization_amd import ops
    extra = () if conn == 'InitialBatchNorm' else ('--force_set_to_best_config=0', f'--type_trick={conn}')
    args, model, data = _tiny_trunk_setup(L=L, se=se, n_override=1500, extra=extra)
    n, H, F_, C = data.x.shape[0], 256, 128, 40
    if model.model.model.has_residual_MLP:
        shapes = [(n, F_)] + [(n, H)] * L + [(n, H)]
    else:
        shapes = [(n, F_)] + [(n, H)] * (L - 1) + [(n, C)]
    seeds = [4000 + i for i in range(len(shapes))]
    ops._seed_override[:] = list(seeds)
    model.train()
    out = model(data.x, data.edge_index)
    assert not ops._seed_override, 'the fused node must draw exactly one seed per dropout site, in the order of the reference\'s calls'
    masks = [ops.dropout_keep_mask(s_, 0.3, sd, DEV).cpu() for s_, sd in zip(shapes, seeds)]
    cfg = orc.make_cfg(type_trick=conn, num_layers=L, num_feats=F_, dim_hidden=H, num_classes=C, dropout=0.3,
                       res_alpha=args.res_alpha, whetherHasSE=tuple(int(c) for c in se), se_reg=0.5)
    csr = orc.build_csr(data.edge_index.cpu(), n)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref, reg = orc.teacher_forward(cfg, sd, data.x.cpu(), csr, training=True, dropout_masks=masks)
    torch.testing.assert_close(out.detach().cpu(), ref, atol=1e-4, rtol=1e-4)
    if reg is not None:
        torch.testing.assert_close(model.se_reg_all.detach().cpu(), reg, atol=1e-3, rtol=1e-5)
    else:
        assert model.se_reg_all is None


def test_bf16_aggregation_variant_config2():
    """BASELINE config 2 (Pubmed-shaped, whetherHasSE=111, 2 layers, hidden 256) with --agg_dtype=bf16.
    Parity targets (stated here, separate from the 1e-4 fp32 target): against the oracle that rounds the
    gathered rows to bf16 at the same two points: logits 2e-3 abs, loss 1e-4 rel, weight gradients 2e-2 of
    their max; against the pure fp32 oracle: logits 5e-2 abs."""
    import contextlib
    import io
    import coldbrew_oracle as orc
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    from gnn_tail_generalization_amd.utils import set_arch_configs
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-pubmed', '--whetherHasSE=111', '--num_layers=2', '--se_reg=0.5',
                                            '--agg_dtype=bf16', '--manual_assign_GPU=0'])
    data = synthetic_data('S-pubmed', seed=0, device=DEV, n_override=4000)
    args.N_nodes, args.dropout, args.device = data.x.shape[0], 0.0, torch.device(DEV)
    set_arch_configs(args)
    torch.manual_seed(1)
    model = TeacherGNN(args).to(DEV)
    assert model.model.model.type_trick == 'InitialBatchNorm' and model.model.model.dim_hidden == 256
    model.train()
    out = model(data.x, data.edge_index)
    loss = ops.nll_logsoftmax(out, data.y, data.train_mask) + args.se_reg * model.se_reg_all
    loss.backward()
    n = data.x.shape[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    csr = orc.build_csr(data.edge_index.cpu(), n)
    ref = {}
    for quant in (True, False):
        cfg = orc.make_cfg(type_trick='InitialBatchNorm', num_layers=2, num_feats=500, dim_hidden=256, num_classes=3,
                           res_alpha=args.res_alpha, whetherHasSE=(1, 1, 1), se_reg=0.5, quant_bf16=quant)
        sdr = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
        o, reg = orc.teacher_forward(cfg, sdr, data.x.cpu(), csr, training=True)
        l = orc.training_loss(cfg, o, reg, data.y.cpu(), data.train_mask.cpu())
        l.backward()
        ref[quant] = (o.detach(), l.detach(), {k: v.grad for k, v in sdr.items() if v.grad is not None})
    torch.testing.assert_close(out.detach().cpu(), ref[True][0], atol=2e-3, rtol=0)
    torch.testing.assert_close(loss.detach().cpu(), ref[True][1], atol=0, rtol=1e-4)
    torch.testing.assert_close(out.detach().cpu(), ref[False][0], atol=5e-2, rtol=0)
    for k, p in model.named_parameters():
        if p.grad is not None and k.endswith('weight'):
            r = ref[True][2][k]
            assert float((p.grad.cpu() - r).abs().max()) <= 2e-2 * float(r.abs().max()) + 1e-7, k


def test_fused_trunk_learnable_input_and_featureless():
    """TeacherGNN wrapper modes on the fused trunk: the gradient w.r.t. trainable node inputs (`embs`, d_x path of the
    trunk backward incl. the input dropout) and the featureless `x*0` mode equal the modular operator path."""
    import contextlib
    import io
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    from gnn_tail_generalization_amd.GNN_model.GCN import TricksComb
    from gnn_tail_generalization_amd.utils import set_arch_configs
    for extra in (['--dim_learnable_input=24'], ['--change_to_featureless=1']):
        with contextlib.redirect_stdout(io.StringIO()):
            args = BaseOptions().get_arguments(['--dataset=S-pubmed', '--num_layers=2', '--whetherHasSE=111', '--se_reg=0.5',
                                                '--manual_assign_GPU=0'] + extra)
        data = synthetic_data('S-pubmed', seed=4, device=DEV, n_override=2500)
        args.N_nodes, args.dropout, args.device = 2500, 0.25, torch.device(DEV)
        set_arch_configs(args)
        torch.manual_seed(3)
        model = TeacherGNN(args).to(DEV)
        if args.dim_learnable_input:
            with torch.no_grad():
                model.embs.mul_(300.0)          # the 0.001-scaled init would make the input gradient vanish in the comparison
        res = {}
        for fused in (True, False):
            TricksComb.use_fused_trunk = fused
            try:
                model.train()
                model.zero_grad()
                ops._seed_override[:] = list(range(300, 310))
                out = model(data.x, data.edge_index)
                ops._seed_override[:] = []
                loss = ops.nll_logsoftmax(out, data.y, data.train_mask) + args.se_reg * model.se_reg_all
                loss.backward()
                res[fused] = (out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
            finally:
                TricksComb.use_fused_trunk = True
        torch.testing.assert_close(res[True][0], res[False][0], atol=2e-5, rtol=1e-5)
        assert set(res[True][1]) == set(res[False][1]) and (('embs' in res[True][1]) == bool(args.dim_learnable_input))
        for k in res[True][1]:
            torch.testing.assert_close(res[True][1][k], res[False][1][k], atol=2e-6, rtol=2e-4, msg=lambda m, k=k: f'{extra} {k}: {m}')


@pytest.mark.parametrize('name', ['nr_se100', 'r_initial_se111'])
def test_proj2class_head_matches_reference(name):
    """--has_proj2class=1: 128-wide common embedding + getMLP([128, 20, C]) head whose Linear layers run on the
    MFMA GEMM (utils.HipLinear) — common embedding, logits 1e-4, loss, every gradient (trunk AND head) against the unmodified reference."""
    from helpers import product_args
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    from gnn_tail_generalization_amd.utils import HipLinear, getMLP
    g = load_golden('proj2class_fixture')[name]
    args = product_args(g['cfg'], extra=['--has_proj2class=1'])
    assert args.dim_commonEmb == 128 and args.TeacherGNN.neurons_proj2class == [128, 20, g['cfg']['num_classes']]
    args.device = torch.device(DEV)
    head = getMLP(args.TeacherGNN.neurons_proj2class)
    assert isinstance(head[0], HipLinear) and isinstance(head[4], HipLinear)
    model = TeacherGNN(args, head)
    model.load_state_dict(g['sd'], strict=True)
    model = model.to(DEV).eval()
    x, ei, y, mask = g['x'].to(DEV), g['edge_index'].to(DEV), g['y'].to(DEV), g['train_mask'].to(DEV)
    res = model.get_3_embs(x, ei, mask)
    torch.testing.assert_close(res.commonEmb.detach().cpu(), g['common'], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(res.emb4classi_full.detach().cpu(), g['logits'], atol=1e-4, rtol=1e-4)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(res.emb4classi, 1), y[mask])
    if model.se_reg_all is not None:
        loss = loss + args.se_reg * model.se_reg_all
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g['loss'], atol=1e-5, rtol=1e-5)
    for k, p in model.named_parameters():
        if k in g['grads']:
            torch.testing.assert_close(p.grad.cpu(), g['grads'][k], atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')
    assert sum(1 for k in g['grads'] if k.startswith('proj2class')) == 6
