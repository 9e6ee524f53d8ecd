"""Generates tests/golden/*.pt by running the UNMODIFIED reference (imported from
/root/reference through oracle/ref_import.py) on small seeded inputs.

Run in the build container only:   python tests/golden/make_golden.py
The fixtures are data (inputs, parameters, expected outputs); no reference source is
copied.  Each file: {'cfg': {...}, 'x', 'edge_index', 'y', 'train_mask', 'sd': state_dict,
'eval_out', 'se_reg_all', 'les', 'train_loss', 'grads': {...}, 'bn_after': {...}}.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_import  # noqa: E402


def ref_args(ns, dataset='Cora', extra=()):
    """The reference's own option pipeline (base_options.py:9-171) on a patched argv."""
    import base_options
    argv = sys.argv
    sys.argv = ['main.py', '--exp_mode=coldbrew', f'--dataset={dataset}', '--manual_assign_GPU=0'] + list(extra)
    try:
        with ref_import.in_scratch():
            args = base_options.BaseOptions().get_arguments()
    finally:
        sys.argv = argv
    args.cuda = False
    args.device = torch.device('cpu')
    return args


def make_graph(kind, n, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == 'example':                                  # utils.py:1096 example graph
        return torch.tensor([[0, 0, 1, 1, 1, 2], [0, 1, 0, 1, 2, 2]], dtype=torch.int64), 3
    if kind == 'sym_loops':                                # Planetoid-style: symmetric + self-loops appended last
        m = 3 * n
        s = torch.randint(0, n, (m,), generator=g)
        d = torch.randint(0, n, (m,), generator=g)
        keep = s != d
        s, d = s[keep], d[keep]
        key = torch.unique(torch.cat([s * n + d, d * n + s]))
        ei = torch.stack([key // n, key % n])
        loops = torch.arange(n)
        return torch.cat([ei, torch.stack([loops, loops])], dim=1), n
    if kind == 'powerlaw':                                 # heavy-tailed degrees, symmetric + loops
        w = (torch.arange(n, dtype=torch.float64) + 1.0) ** -0.9
        m = 4 * n
        s = torch.multinomial(w, m, replacement=True, generator=g)
        d = torch.multinomial(w, m, replacement=True, generator=g)
        keep = s != d
        s, d = s[keep], d[keep]
        perm = torch.randperm(n, generator=g)
        s, d = perm[s], perm[d]
        key = torch.unique(torch.cat([s * n + d, d * n + s]))
        ei = torch.stack([key // n, key % n])
        loops = torch.arange(n)
        return torch.cat([ei, torch.stack([loops, loops])], dim=1), n
    if kind == 'asym_multi':                               # asymmetric multigraph, every node has in-degree >= 1
        m = 4 * n
        s = torch.randint(0, n, (m,), generator=g)
        d = torch.randint(0, n, (m,), generator=g)
        ring_s = torch.arange(n)
        ring_d = (torch.arange(n) + 1) % n
        dup = torch.randint(0, m, (n,), generator=g)
        ei = torch.stack([torch.cat([s, ring_s, s[dup]]), torch.cat([d, ring_d, d[dup]])])
        ei = ei[:, torch.randperm(ei.shape[1], generator=g)]
        return ei, n
    if kind == 'zero_in':                                  # node n-1 has no incoming edge -> DGLError
        s = torch.arange(n)
        d = (torch.arange(n) + 1) % (n - 1)
        return torch.stack([s, d]), n
    raise ValueError(kind)


CASES = []


def case(name, **kw):
    d = dict(name=name, dataset='Cora', graph='sym_loops', n=96, f=24, h=16, c=5, layers=2, se='000',
             type_trick=None, force_best=1, extra=(), layer_agg='concat', learnable=0, featureless=0,
             node_norm_type='n', se_reg=0.5, seed=0)
    d.update(kw)
    CASES.append(d)


# non-residual mode with each structural-embedding flag (Cora best config -> 'NoResNodeNorm', norms skipped)
for _se in ['000', '100', '001', '111']:
    for _l in [2, 3]:
        case(f'nr_se{_se}_L{_l}', se=_se, layers=_l)
# residual mode; Pubmed best config -> 'InitialBatchNorm' (norm built, never run)
case('r_initialbn_se000_L2', dataset='Pubmed', se='000', h=32)
case('r_initialbn_se111_L2', dataset='Pubmed', se='111', h=32)
case('r_initialbn_se111_L3_powerlaw', dataset='Pubmed', se='111', h=32, layers=3, graph='powerlaw', n=160)
# the benchmark's shape in small: hidden 256 (the fused trunk and its aggregation + GEMM kernels), 3 layers, a 10 % train mask on a power-law graph —
# the masked loss's gradient is zero outside the train rows, which the product's row-sparse backward exploits (tests/test_gpu_rowsparse.py)
case('r_initialbn_h256_L3_train10', dataset='Pubmed', se='000', f=32, h=256, layers=3, graph='powerlaw', n=400, train_frac=0.1)
# the same with structural-embedding tables on every layer (the reference's own mode): a compact level of the row-sparse backward scatters the
# table gradient dL/dZ_l into all rows and keeps the first form of its level 0
case('r_initialbn_h256_L3_train10_se111', dataset='Pubmed', se='111', f=32, h=256, layers=3, graph='powerlaw', n=300, train_frac=0.1)
for _t in ['Residual', 'Initial']:
    case(f'r_{_t.lower()}_L3', force_best=0, type_trick=_t, layers=3, se='111')
# the reference's other two default trunk shapes at the fused path's width (round 5): 'Residual' (WISCONSIN / CORNELL / TEXAS in the best-config
# table, base_options.py:416-421; mix source = the previous layer's ReLU output, res_tricks.py:7-14) without and with structural-embedding
# tables, and the non-residual stack (Cora / Citeseer / ACTOR: F -> H -> H -> C, dropout on the logits, GCN.py:44-50,70-71,133)
case('r_residual_h256_L3_train10', force_best=0, type_trick='Residual', se='000', f=32, h=256, layers=3, graph='powerlaw', n=400, train_frac=0.1)
case('r_residual_h256_L3_train10_se111', force_best=0, type_trick='Residual', se='111', f=32, h=256, layers=3, graph='powerlaw', n=300, train_frac=0.1)
case('r_residual_h256_L2', force_best=0, type_trick='ResidualPairNorm', se='000', f=32, h=256, layers=2, graph='powerlaw', n=300, train_frac=0.3)
case('nr_h256_L3_train10', se='000', f=32, h=256, layers=3, graph='powerlaw', n=400, train_frac=0.1)
case('nr_h256_L3_train10_se111', se='111', f=32, h=256, layers=3, graph='powerlaw', n=300, train_frac=0.1)
for _agg in ['concat', 'maxpool', 'attention']:
    case(f'r_dense_{_agg}_L3', force_best=0, type_trick='Dense', layer_agg=_agg, layers=3)
    case(f'r_jumping_{_agg}_L2', force_best=0, type_trick='Jumping', layer_agg=_agg, layers=2, se='111')
# bare norm names (run_norm_if_any executes them)
case('norm_batchnorm', force_best=0, type_trick='BatchNorm', layers=3)
case('norm_pairnorm', force_best=0, type_trick='PairNorm', layers=3)
case('norm_meannorm', force_best=0, type_trick='MeanNorm', layers=3)
for _k in ['n', 'v', 'm', 'srv', 'pr']:
    case(f'norm_nodenorm_{_k}', force_best=0, type_trick='NodeNorm', node_norm_type=_k, layers=3)
case('norm_groupnorm', force_best=0, type_trick='GroupNorm', layers=3)
case('norm_groupnorm_pubmed', dataset='Pubmed', force_best=0, type_trick='GroupNorm', layers=2)
case('norm_combnorm', force_best=0, type_trick='CombNorm', layers=2)
# names proving the tricks are inert
case('inert_dropedge_name', force_best=0, type_trick='DropEdge', layers=2)
case('concat_initial_plus_batchnorm', force_best=0, type_trick='Initial+BatchNorm', layers=2, se='111')
# graph shapes
case('graph_asym_multi', graph='asym_multi', se='111', layers=3)
case('graph_example', graph='example', n=3, f=4, h=8, c=3, se='100')
case('graph_powerlaw_d7_d64', graph='powerlaw', n=200, f=30, h=64, c=7)
case('graph_zero_in_degree', graph='zero_in', n=12, f=4, h=8, c=3)
# TeacherGNN wrapper modes
case('teacher_learnable_input', learnable=12, se='100')
case('teacher_featureless', featureless=1, se='111')


def build(ns, c):
    extra = [f'--num_layers={c["layers"]}', f'--whetherHasSE={c["se"]}', f'--force_set_to_best_config={c["force_best"]}',
             f'--layer_agg={c["layer_agg"]}', f'--dim_learnable_input={c["learnable"]}',
             f'--change_to_featureless={c["featureless"]}', f'--node_norm_type={c["node_norm_type"]}',
             f'--se_reg={c["se_reg"]}'] + list(c['extra'])
    if c['type_trick'] is not None:
        extra.append(f'--type_trick={c["type_trick"]}')
    args = ref_args(ns, c['dataset'], extra)
    ei, n = make_graph(c['graph'], c['n'], c['seed'])
    args.N_nodes, args.num_feats, args.dim_hidden, args.num_classes = n, c['f'], c['h'], c['c']
    args.dropout = 0.0
    ns.utils.set_arch_configs(args)
    g = torch.Generator().manual_seed(1000 + c['seed'])
    x = torch.rand(n, c['f'], generator=g)
    y = torch.randint(0, c['c'], (n,), generator=g)
    train_mask = torch.rand(n, generator=g) < c.get('train_frac', 0.5)
    train_mask[0] = True
    torch.manual_seed(c['seed'])
    model = ns.GNN_normalizations.TeacherGNN(args)
    # perturb biases / affine params so that they matter in the comparison
    with torch.no_grad():
        gg = torch.Generator().manual_seed(77)
        for k, p in model.named_parameters():
            if k.endswith('bias') or 'layers_norm' in k:
                p.add_(0.1 * torch.randn(p.shape, generator=gg))
    return args, model, x, ei, y, train_mask


def cfg_of(args, c):
    return dict(type_trick=args.type_trick, num_layers=args.num_layers, num_feats=args.num_feats_bkup,
                dim_hidden=args.dim_hidden, num_classes=args.num_classes, dropout=0.0, res_alpha=args.res_alpha,
                layer_agg=args.layer_agg, whetherHasSE=tuple(args.TeacherGNN.whetherHasSE),
                node_norm_type=args.node_norm_type, num_groups=args.num_groups, skip_weight=args.skip_weight,
                se_reg=args.se_reg, change_to_featureless=int(args.TeacherGNN.change_to_featureless),
                dim_learnable_input=args.dim_learnable_input, N_nodes=args.N_nodes, dataset=args.dataset,
                weight_decay=args.weight_decay, lr=args.lr)


def run_case(ns, c):
    args, model, x, ei, y, train_mask = build(ns, c)
    out = dict(cfg=None, x=x, edge_index=ei, y=y, train_mask=train_mask)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out['sd'] = sd0
    if c['graph'] == 'zero_in':
        try:
            model.eval()
            model(x, ei)
            raise SystemExit('expected DGLError')
        except Exception as e:  # noqa: BLE001
            assert type(e).__name__ == 'DGLError', e
        out['cfg'] = cfg_of(args, c)
        out['raises'] = 'zero_in_degree'
        return out
    model.eval()
    with torch.no_grad():
        eval_out = model(x, ei).clone()
        out['eval_out'] = eval_out
        out['se_reg_all'] = None if model.se_reg_all is None else model.se_reg_all.detach().clone()
        xin = model.embs if args.dim_learnable_input > 0 else (x * 0 if args.TeacherGNN.change_to_featureless else x)
        out['les'] = model.model.model.collect_SE(xin, ei).clone()
    # train-mode (dropout = 0) loss and gradients; the reference's in-place `+=` (GCN.py:120)
    # needs allow_mutation_on_saved_tensors for >= 2 SE layers
    model.train()
    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        res = model.get_3_embs(x, ei, train_mask)
        logits = torch.nn.functional.log_softmax(res.emb4classi, 1)
        loss = torch.nn.functional.nll_loss(logits, y[train_mask])
        if model.se_reg_all is not None:
            loss = loss + args.se_reg * model.se_reg_all
        model.zero_grad()
        loss.backward()
    out['train_out'] = res.emb4classi_full.detach().clone()
    out['train_loss'] = loss.detach().clone()
    out['grads'] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    out['bn_after'] = {k: v.detach().clone() for k, v in model.state_dict().items() if 'running_' in k}
    out['cfg'] = cfg_of(args, c)
    return out


def run_trainer_case(ns, name, want_headtail, se, dataset, steps=5):
    """Rows a15-a17: the unmodified run_trainSet()/run_testSet() on hand-set trainer fields."""
    c = dict(name=name, dataset=dataset, graph='powerlaw', n=120, f=20, h=16, c=4, layers=2, se=se, type_trick=None,
             force_best=1, extra=(f'--want_headtail={want_headtail}',), layer_agg='concat', learnable=0, featureless=0,
             node_norm_type='n', se_reg=0.5, seed=3)
    args, model, x, ei, y, train_mask = build(ns, c)
    args.lr, args.weight_decay = 0.01, 5e-4
    Data = sys.modules['torch_geometric.data.data'].Data
    data = Data(x=x, y=y, edge_index=ei, train_mask=train_mask, test_mask=~train_mask)
    deg = torch.bincount(ei[1], minlength=x.shape[0])
    order = torch.argsort(deg, stable=True)
    data.zero_deg_idx = order[:10].numpy()
    data.small_deg_idx = order[10:40].numpy()
    data.large_deg_idx = order[-30:].numpy()
    t = ns.trainer.trainer.__new__(ns.trainer.trainer)
    t.args, t.data, t.bag = args, data, {}
    t.loss_fn = torch.nn.functional.nll_loss
    t.teacherGNN = model
    t.optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec, bags = [], []
    with ref_import.in_scratch(), torch.autograd.graph.allow_mutation_on_saved_tensors():
        for ep in range(steps):
            t.epoch = ep
            loss_train, _, _ = t.run_trainSet()
            acc_train, _, acc_test, _ = t.run_testSet()
            rec.append([loss_train, acc_train, acc_test])
            bags.append([float(v) for v in t.bag.get('head_tail_iso', [])])
    out = dict(cfg=cfg_of(args, c), x=x, edge_index=ei, y=y, train_mask=train_mask, sd=sd0,
               zero_deg_idx=torch.as_tensor(data.zero_deg_idx), small_deg_idx=torch.as_tensor(data.small_deg_idx),
               large_deg_idx=torch.as_tensor(data.large_deg_idx),
               trajectory=torch.tensor(rec, dtype=torch.float64), head_tail_iso=torch.tensor(bags, dtype=torch.float64),
               sd_final={k: v.detach().clone() for k, v in model.state_dict().items()},
               want_headtail=want_headtail, use_special_split=int(args.use_special_split), steps=steps)
    return out


def options_fixture(ns):
    """CLI contract (base_options.py): the namespace the reference derives for README-style command lines."""
    lines = {
        'cora_000': ['--dataset=Cora', '--train_which=TeacherGNN', '--whetherHasSE=000', '--want_headtail=1',
                     '--num_layers=2', '--use_special_split=1'],
        'cora_100': ['--dataset=Cora', '--train_which=TeacherGNN', '--whetherHasSE=100', '--se_reg=32',
                     '--want_headtail=1', '--num_layers=2', '--use_special_split=1'],
        'citeseer_100': ['--dataset=Citeseer', '--train_which=TeacherGNN', '--whetherHasSE=100', '--se_reg=0.5',
                         '--want_headtail=1', '--num_layers=2', '--use_special_split=1'],
        'pubmed_111': ['--dataset=Pubmed', '--train_which=TeacherGNN', '--whetherHasSE=111', '--se_reg=0.5',
                       '--want_headtail=1', '--num_layers=2', '--use_special_split=1'],
        'arxiv': ['--dataset=ogbn-arxiv', '--train_which=TeacherGNN', '--num_layers=3', '--use_special_split=0'],
        'texas_nobest': ['--dataset=TEXAS', '--force_set_to_best_config=0', '--type_trick=Residual'],
        'chameleon': ['--dataset=chameleon'],
    }
    out = {}
    for k, argv in lines.items():
        args = ref_args(ns, 'Cora', [])  # warm the parser
        saved = sys.argv
        sys.argv = ['main.py', '--exp_mode=coldbrew', '--manual_assign_GPU=0'] + argv
        try:
            import base_options
            with ref_import.in_scratch():
                args = base_options.BaseOptions().get_arguments()
                ns.utils.set_arch_configs(args)
        finally:
            sys.argv = saved
        d = {}
        for kk, vv in vars(args).items():
            if isinstance(vv, (int, float, str, bool, type(None))):
                d[kk] = vv
            elif isinstance(vv, (list, tuple)) and all(isinstance(e, (int, float, str)) for e in vv):
                d[kk] = list(vv)
        d['TeacherGNN.whetherHasSE'] = list(args.TeacherGNN.whetherHasSE)
        d['TeacherGNN.neurons_proj2class'] = list(args.TeacherGNN.neurons_proj2class)
        d['argv'] = argv
        out[k] = d
    return out


def utils_fixture(ns):
    """Caller-side helpers (utils.py): get_partial_sorted_idx, ensure_symmetric, save_graph_analyze +
    craft_isolation_v2 on a small power-law graph — the edge_index the path consumes."""
    out = {}
    g = torch.Generator().manual_seed(5)
    arr = torch.randint(0, 40, (500,), generator=g).numpy()
    out['degs'] = torch.from_numpy(arr)
    for mode in ['top50', 'top25', 'top12', 'top6', 'top3', 'bottom50', 'bottom25', 'bottom12', 'bottom6', 'bottom3']:
        out['idx_' + mode] = torch.from_numpy(ns.utils.get_partial_sorted_idx(arr, mode))
    ei = torch.randint(0, 60, (2, 300), generator=g)
    out['asym_edge_index'] = ei
    out['ensure_symmetric'] = ns.utils.ensure_symmetric(ei)
    Data = sys.modules['torch_geometric.data.data'].Data
    for special in (0, 1):
        ei2, n = make_graph('powerlaw', 150, 9)
        data = Data(x=torch.zeros(n, 3), edge_index=ei2.clone())
        with ref_import.in_scratch():
            ns.utils.save_graph_analyze(n, data, special)
        pre = f'sga{special}_'
        out[pre + 'edge_index_in'] = ei2
        out[pre + 'edge_index_out'] = data.edge_index.clone()
        for k in ['zero_deg_idx', 'small_deg_idx', 'large_deg_idx']:
            if hasattr(data, k):
                out[pre + k] = torch.as_tensor(getattr(data, k))
        for k in ['zero_deg_mask', 'small_deg_mask', 'large_deg_mask']:
            if hasattr(data, k):
                out[pre + k] = getattr(data, k).clone()
    return out


def lp_fixture(ns):
    """§8f row 4 — pure label propagation (trainer_node_classification.run_pureLP :33-63): the reference's own
    process_adj / gen_normalized_adjs / general_outcome_correlation (outcome_correlation.py:39-55,128-156) driven
    exactly as label_propagation does (:147-156), with device='cpu' passed explicitly (its default 'cuda' cannot
    run in this container)."""
    with ref_import.in_scratch():
        from Label_propagation_model import outcome_correlation as oc
    import torch.nn.functional as F
    Data = sys.modules['torch_geometric.data.data'].Data
    ei, n = make_graph('powerlaw', 180, 21)
    g = torch.Generator().manual_seed(21)
    c = 6
    y = torch.randint(0, c, (n,), generator=g)
    train_mask = torch.rand(n, generator=g) < 0.4
    data = Data(x=torch.zeros(n, 2), y=y, edge_index=ei.clone(), train_mask=train_mask)
    adj, d_isqrt = oc.process_adj(data)
    dad, _, _ = oc.gen_normalized_adjs(adj, d_isqrt)
    y0 = torch.zeros((n, c))
    idx = oc.get_labels_from_name(['train'], {'train': train_mask})
    y0[idx] = F.one_hot(y[idx], c).float().squeeze(1)
    out = oc.general_outcome_correlation(dad, y0, 0.5, 50, post_step=lambda t: torch.clamp(t, 0, 1), alpha_term=True,
                                         device='cpu', display=False)
    acc_train = ns.trainer.evaluate(out, y, train_mask)
    acc_test = ns.trainer.evaluate(out, y, ~train_mask)
    return dict(edge_index=ei, edge_index_undirected=data.edge_index.clone(), y=y, train_mask=train_mask, deg_inv_sqrt=d_isqrt,
                out=out, acc=torch.tensor([np.round(acc_train * 100, 2), np.round(acc_test * 100, 2)], dtype=torch.float64),
                alpha=0.5, num_propagations=50, num_classes=c)


def semlp_fixture(ns):
    """§8f row 2 — the unmodified SEMLP.replacement (MLP_model/__init__.py:143-156) on seeded embeddings."""
    with ref_import.in_scratch():
        import MLP_model
    out = {}
    for name, (b, n, d, k) in {'small': (40, 150, 24, 2), 'wide': (33, 300, 71, 3)}.items():
        g = torch.Generator().manual_seed(b + n)
        obj = MLP_model.SEMLP.__new__(MLP_model.SEMLP)
        torch.nn.Module.__init__(obj)
        obj.teacherSE = torch.randn(n, d, generator=g)
        obj.topK_2_replace = k
        q = torch.randn(b, d, generator=g)
        out[name] = dict(q=q, teacher=obj.teacherSE.clone(), k=k, out=obj.replacement(q).clone())
    return out


def proj2class_fixture(ns):
    """--has_proj2class=1 (GNN_normalizations.py:11-29, utils.py:613-624, trainer :304-305): the GNN emits a 128-wide common embedding
    and the getMLP([128, 20, C]) head turns it into logits.  Eval mode (the head's own Dropout(0.1) draws from torch's CPU stream
    in train mode): common embedding, logits, loss and every gradient, for a non-residual and a residual trunk."""
    out = {}
    for name, kw in {'nr_se100': dict(se='100', layers=2), 'r_initial_se111': dict(dataset='Pubmed', se='111', h=32, layers=2)}.items():
        c = dict(name=name, dataset='Cora', graph='powerlaw', n=140, f=24, h=16, c=5, layers=2, se='000', type_trick=None, force_best=1,
                 extra=('--has_proj2class=1',), layer_agg='concat', learnable=0, featureless=0, node_norm_type='n', se_reg=0.5, seed=11)
        c.update(kw)
        extra = [f'--num_layers={c["layers"]}', f'--whetherHasSE={c["se"]}', f'--force_set_to_best_config={c["force_best"]}',
                 f'--se_reg={c["se_reg"]}'] + list(c['extra'])
        args = ref_args(ns, c['dataset'], extra)
        ei, n = make_graph(c['graph'], c['n'], c['seed'])
        args.N_nodes, args.num_feats, args.dim_hidden, args.num_classes = n, c['f'], c['h'], c['c']
        args.dropout = 0.0
        ns.utils.set_arch_configs(args)
        assert args.dim_commonEmb == 128 and args.TeacherGNN.neurons_proj2class == [128, 20, c['c']]
        g = torch.Generator().manual_seed(1000 + c['seed'])
        x = torch.rand(n, c['f'], generator=g)
        y = torch.randint(0, c['c'], (n,), generator=g)
        train_mask = torch.rand(n, generator=g) < 0.5
        train_mask[0] = True
        torch.manual_seed(c['seed'])
        head = ns.utils.getMLP(args.TeacherGNN.neurons_proj2class)          # trainer :304-305
        model = ns.GNN_normalizations.TeacherGNN(args, head)
        with torch.no_grad():
            gg = torch.Generator().manual_seed(78)
            for k, p in model.named_parameters():
                if k.endswith('bias'):
                    p.add_(0.1 * torch.randn(p.shape, generator=gg))
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model.eval()
        with torch.autograd.graph.allow_mutation_on_saved_tensors():
            res = model.get_3_embs(x, ei, train_mask)
            loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(res.emb4classi, 1), y[train_mask])
            if model.se_reg_all is not None:
                loss = loss + args.se_reg * model.se_reg_all
            model.zero_grad()
            loss.backward()
        cfg = cfg_of(args, c)
        cfg['num_classes'] = c['c']              # cfg_of reads args.num_classes AFTER TeacherGNN replaced it by dim_commonEmb
        cfg['dim_commonEmb'] = 128
        out[name] = dict(cfg=cfg, x=x, edge_index=ei, y=y, train_mask=train_mask, sd=sd0, common=res.commonEmb.detach().clone(),
                         logits=res.emb4classi_full.detach().clone(), loss=loss.detach().clone(),
                         grads={k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    return out


def edge_weight_fixture(ns):
    """GCNConv.forward(graph, feat, edge_weight=w) of the unmodified reference (GCN.py:199-202: u_mul_e) on the directed multigraph and
    the power-law graph: output, regulariser and the gradients w.r.t. feat, weight, bias, le AND edge_weight."""
    import dgl
    out = {}
    for name, (kind, n, d_in, d_out, se) in {'asym_multi': ('asym_multi', 70, 12, 20, True), 'powerlaw': ('powerlaw', 120, 9, 33, False)}.items():
        args = ref_args(ns, 'Cora', [])
        ei, n = make_graph(kind, n, 13)
        args.N_nodes = n
        g = torch.Generator().manual_seed(500 + n)
        torch.manual_seed(7)
        conv = ns.GCN.GCNConv(d_in, d_out, args=args, whetherHasSE=se)
        with torch.no_grad():
            conv.bias.add_(0.1 * torch.randn(d_out, generator=g))
        feat = torch.randn(n, d_in, generator=g, requires_grad=True)
        w = (torch.rand(ei.shape[1], generator=g) + 0.25).requires_grad_(True)
        graph = dgl.graph((ei[0].tolist(), ei[1].tolist()))
        gout = torch.randn(n, d_out, generator=g)
        rst, se_reg = conv(graph, feat, edge_weight=w)
        loss = (rst * gout).sum() + (0.5 * se_reg if se_reg is not None else 0.0)
        loss.backward()
        out[name] = dict(edge_index=ei, n=n, feat=feat.detach().clone(), edge_weight=w.detach().clone(), gout=gout,
                         sd={k: v.detach().clone() for k, v in conv.state_dict().items()}, out=rst.detach().clone(),
                         se_reg=None if se_reg is None else se_reg.detach().clone(), d_feat=feat.grad.clone(), d_edge_weight=w.grad.clone(),
                         grads={k: p.grad.clone() for k, p in conv.named_parameters()})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    ns = ref_import.load_reference()
    torch.set_num_threads(1)
    for c in CASES:
        if a.only and (a.only in ('utils', 'options', 'trainer', 'lp', 'semlp', 'proj2class', 'edge_weight') or a.only not in c['name']):
            continue
        out = run_case(ns, c)
        torch.save(out, os.path.join(HERE, f'case_{c["name"]}.pt'))
        print('wrote', c['name'], out['cfg']['type_trick'])
    if not a.only or 'trainer' in a.only:
        for name, wh, se, ds in [('trainer_headtail1_se100_cora', 1, '100', 'Cora'),
                                 ('trainer_headtail0_se111_pubmed', 0, '111', 'Pubmed'),
                                 ('trainer_headtail1_se000_cora', 1, '000', 'Cora')]:
            out = run_trainer_case(ns, name, wh, se, ds)
            torch.save(out, os.path.join(HERE, f'{name}.pt'))
            print('wrote', name, out['trajectory'][-1].tolist())
    if not a.only or 'utils' in a.only:
        torch.save(utils_fixture(ns), os.path.join(HERE, 'utils_fixture.pt'))
        print('wrote utils_fixture.pt')
    if not a.only or 'semlp' in a.only:
        torch.save(semlp_fixture(ns), os.path.join(HERE, 'semlp_fixture.pt'))
        print('wrote semlp_fixture.pt')
    if not a.only or 'edge_weight' in a.only:
        torch.save(edge_weight_fixture(ns), os.path.join(HERE, 'edge_weight_fixture.pt'))
        print('wrote edge_weight_fixture.pt')
    if not a.only or 'proj2class' in a.only:
        torch.save(proj2class_fixture(ns), os.path.join(HERE, 'proj2class_fixture.pt'))
        print('wrote proj2class_fixture.pt')
    if not a.only or a.only == 'lp':
        torch.save(lp_fixture(ns), os.path.join(HERE, 'lp_fixture.pt'))
        print('wrote lp_fixture.pt')
    if not a.only or 'options' in a.only:
        import json
        with open(os.path.join(HERE, 'options.json'), 'w') as f:
            json.dump(options_fixture(ns), f, indent=1, sort_keys=True)
        print('wrote options.json')


if __name__ == '__main__':
    main()
