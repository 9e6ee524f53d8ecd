"""Normalisation tricks and their selection helpers — same names and arithmetic as the
reference's GNN_model/norm_tricks.py.

Reference quirk kept on purpose: appendNormLayer *builds* a layer on a
substring match (norm_tricks.py:131-143) but run_norm_if_any *runs* it only when
type_trick equals a bare name exactly (norm_tricks.py:147), so concatenated names such as
'InitialBatchNorm' register parameters that never execute.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

BARE_NORM_NAMES = ['BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm']


class comb_norm(nn.Module):                      # norm_tricks.py:9-17
    def __init__(self, norm_list):
        super().__init__()
        self.norm_list = nn.ModuleList(norm_list)

    def forward(self, x):
        for mod in self.norm_list:
            x = mod(x)
        return x


def _bn(layer, x):
    """nn.BatchNorm1d layers stay torch modules (parameter / buffer names), their arithmetic runs on the HIP reductions."""
    if x.is_cuda and x.dtype == torch.float32:
        from ..norms_hip import batch_norm
        return batch_norm(layer, x)
    return layer(x)


class pair_norm(nn.Module):                      # norm_tricks.py:20-30: centre columns, divide by the rms row norm
    def forward(self, x):
        from ..norms_hip import pair_norm as fn
        return fn(x)


class mean_norm(nn.Module):                      # norm_tricks.py:33-41: centre columns
    def forward(self, x):
        from ..norms_hip import mean_norm as fn
        return fn(x)


class node_norm(nn.Module):                      # norm_tricks.py:44-92
    def __init__(self, node_norm_type='n', unbiased=False, eps=1e-5, power_root=2, **kwargs):
        super().__init__()
        self.unbiased, self.eps = unbiased, eps
        self.node_norm_type = node_norm_type
        self.power = 1 / power_root

    def forward(self, x):
        if self.unbiased or self.node_norm_type not in ('n', 'v', 'm', 'srv', 'pr'):
            return x if self.node_norm_type not in ('n', 'v', 'm', 'srv', 'pr') else self._torch(x)
        from ..norms_hip import node_norm as fn
        return fn(x, self.node_norm_type, self.eps, self.power)

    def _torch(self, x):   # unbiased=True is never selected by the reference's options; kept for the ctor contract
        std = (torch.var(x, unbiased=self.unbiased, dim=1, keepdim=True) + self.eps).sqrt()
        mean = torch.mean(x, dim=1, keepdim=True)
        t = self.node_norm_type
        return {'n': (x - mean) / std, 'v': x / std, 'm': x - mean, 'srv': x / torch.sqrt(std),
                'pr': x / torch.pow(std, self.power)}[t]

    def extra_repr(self):
        return f'node_norm_type={self.node_norm_type}'


class group_norm(nn.Module):                     # norm_tricks.py:95-120
    def __init__(self, dim_to_norm=None, dim_hidden=16, num_groups=None, skip_weight=None, **w):
        super().__init__()
        self.num_groups, self.skip_weight = num_groups, skip_weight
        self.dim_hidden = dim_hidden if dim_to_norm is None else dim_to_norm
        self.bn = nn.BatchNorm1d(self.dim_hidden * self.num_groups, momentum=0.3)
        self.group_func = nn.Linear(self.dim_hidden, self.num_groups, bias=True)

    def forward(self, x):
        if self.num_groups == 1:
            x_temp = _bn(self.bn, x)
        else:
            from ..gemm import linear
            score = F.softmax(linear(x, self.group_func.weight, self.group_func.bias), dim=1)   # [N, G] gates (MFMA GEMM)
            x_temp = (score.unsqueeze(2) * x.unsqueeze(1)).reshape(x.shape[0], -1)  # G scaled copies, concatenated
            x_temp = _bn(self.bn, x_temp).view(-1, self.num_groups, self.dim_hidden).sum(dim=1)
        return x + x_temp * self.skip_weight


def AcontainsB(A, listB):                        # norm_tricks.py:123-127
    return any(s in A for s in listB)


def appendNormLayer(net, args, dim_to_norm=None):   # norm_tricks.py:130-143 (substring match)
    t = args.type_trick
    if 'BatchNorm' in t:
        net.layers_norm.append(nn.BatchNorm1d(net.dim_hidden if dim_to_norm is None else dim_to_norm))
    elif 'PairNorm' in t:
        net.layers_norm.append(pair_norm())
    elif 'NodeNorm' in t:
        net.layers_norm.append(node_norm(**vars(net.args)))
    elif 'MeanNorm' in t:
        net.layers_norm.append(mean_norm())
    elif 'GroupNorm' in t:
        net.layers_norm.append(group_norm(dim_to_norm, **vars(reset_weight_GroupNorm(args))))
    elif 'CombNorm' in t:
        net.layers_norm.append(comb_norm([group_norm(dim_to_norm, **vars(reset_weight_GroupNorm(args))),
                                          node_norm(**vars(net.args))]))


def run_norm_if_any(net, x, ilayer):             # norm_tricks.py:146-150 (exact match)
    if net.args.type_trick in BARE_NORM_NAMES:
        layer = net.layers_norm[ilayer]
        return _bn(layer, x) if isinstance(layer, nn.BatchNorm1d) else layer(x)
    return x


_SKIP_DEEP = {  # (shallow value, deep value, depth limit) per (dataset family, model family) — norm_tricks.py:159-195
    ('Citeseer', 'gnn'): (0.001, 0.005, 6), ('Citeseer', 'other'): (0.0005, 0.002, 60),
    ('ogbn-arxiv', 'gnn'): (0.001, 0.005, 6), ('ogbn-arxiv', 'other'): (0.0005, 0.002, 60),
    ('Cora', 'GCN'): (0.001, 0.03, 6), ('Cora', 'GAT'): (0.001, 0.01, 6), ('Cora', 'other'): (0.01, 0.005, 60),
    ('Pubmed', 'GCN'): (0.001, 0.01, 6), ('Pubmed', 'GAT'): (0.005, 0.01, 6),
    ('CoauthorCS', 'gnn'): (0.001, 0.03, 6),
}


def reset_weight_GroupNorm(args):                # norm_tricks.py:153-206
    if args.num_groups is not None:
        return args
    args.miss_rate = 0.
    ds, tm, L = args.dataset, args.type_model, args.num_layers

    def pick(key):
        lo, hi, lim = _SKIP_DEEP[key]
        return lo if L < lim else hi

    if ds == 'Citeseer' or 'CV' in ds:
        args.skip_weight = pick(('Citeseer', 'gnn' if tm in ['GAT', 'GCN'] else 'other'))
    elif ds == 'ogbn-arxiv':
        args.skip_weight = pick(('ogbn-arxiv', 'gnn' if tm in ['GAT', 'GCN'] else 'other'))
    elif ds == 'Pubmed':
        args.skip_weight = pick(('Pubmed', tm)) if tm in ['GCN', 'GAT'] else 0.05
    elif ds == 'Cora':
        args.skip_weight = pick(('Cora', tm if tm in ['GCN', 'GAT'] else 'other'))
    elif ds == 'CoauthorCS':
        if tm in ['GAT', 'GCN']:
            args.skip_weight = pick(('CoauthorCS', 'gnn'))
        else:
            args.epochs = 500
            args.skip_weight = 0.001 if L < 10 else .5
    elif ds in ['CoauthorPhysics', 'AmazonComputers', 'AmazonPhoto', 'TEXAS', 'WISCONSIN', 'CORNELL']:
        args.skip_weight = 0.005
    else:
        raise NotImplementedError
    args.num_groups = 5 if ds == 'Pubmed' else 10
    return args
