"""Callers' helpers of the TeacherGNN path — the subset of the reference's utils.py that the
node-classification trainer and TeacherGNN touch, with the same names and results.

Per-edge Python loops of the reference (graph_analyze utils.py:300-334, craft_isolation_v2
utils.py:731-752) are restated as vectorised tensor operations that run where the data
lives (GPU), producing identical outputs.
"""
import os

import numpy as np
import torch
import torch.nn as nn

join = os.path.join


class D:
    """Attribute bag (utils.py:857-876)."""

    def __repr__(self):
        lines = []

        def walk(obj, prefix=''):
            for att in dir(obj):
                if att.startswith('__'):
                    continue
                v = getattr(obj, att)
                if type(v) is D:
                    lines.append(prefix + f'{att:16} = \n')
                    walk(v, prefix='\t')
                else:
                    lines.append(prefix + f'{att:16} = {v}\n')
            lines.append('\n')

        walk(self)
        return '-' * 40 + '\n' + ''.join(lines)


C = D


def AcontainsB(A, listB):
    return any(s in A for s in listB)


def tonp(arr):                                    # utils.py:943-948
    if type(arr) is torch.Tensor:
        return arr.detach().cpu().data.numpy()
    return np.asarray(arr)


def toitem(arr, round=True):                      # utils.py:950-956
    arr1 = tonp(arr)
    value = arr1.reshape(-1)[0]
    if round:
        value = np.round(value, 3)
    assert arr1.size == 1
    return value


_SE_FLAGS = {'111': [1, 1, 1], '000': [0, 0, 0], '001': [0, 0, 1], '100': [1, 0, 0]}


def set_arch_configs(args):
    """Derived architecture options (utils.py:588-645): whetherHasSE string -> per-position flags
    (first / middle / last layer), common-embedding width, student-MLP options."""
    args.SEMLP__downgrade_to_MLP = args.SEMLP_topK_2_replace == -99
    args.activation = 'gelu'
    args.is_bipartite = False
    args.TeacherGNN = C()
    args.TeacherGNN.lossa_semantic = 1
    args.TeacherGNN.lossa_structure = 1
    args.TeacherGNN.change_to_featureless = args.change_to_featureless
    args.TeacherGNN.num_layers = args.num_layers
    if args.whetherHasSE not in _SE_FLAGS:
        raise NotImplementedError
    args.TeacherGNN.whetherHasSE = list(_SE_FLAGS[args.whetherHasSE])
    args.dim_commonEmb = 128 if args.has_proj2class else args.num_classes
    args.num_feats_bkup = args.num_feats
    args.embDim_linkp = 10
    args.num_classes_bkup = args.num_classes
    args.TeacherGNN.neurons_proj2class = [args.dim_commonEmb, 20, args.num_classes_bkup]
    args.TeacherGNN.neurons_proj2linkp = [args.dim_commonEmb, 32]
    args.StudentBaseMLP = C()
    if args.studentMLP__skip_conn_T_and_res_blks != '':
        skip_period, num_blocks = args.studentMLP__skip_conn_T_and_res_blks.split('&')
        args.StudentBaseMLP.skip_conn_period, args.StudentBaseMLP.num_blocks = int(skip_period), int(num_blocks)
    else:
        args.StudentBaseMLP.skip_conn_period, args.StudentBaseMLP.num_blocks = 2, 3
    args.StudentBaseMLP.dims_in_out = [args.num_feats_bkup, args.num_classes_bkup]
    args.StudentBaseMLP.dim_model = args.StudentMLP__dim_model
    args.StudentBaseMLP.lrn_from = 'label'
    if args.studentMLP__opt_lr != '':
        _opt, _lr = args.studentMLP__opt_lr.split('&')
        args.optfun = _opt
        args.lr = float(_lr)


def ensure_symmetric(edge_index):
    """Union with the transpose, coalesced, sorted by (row, col) (utils.py:667-674)."""
    n = int(edge_index.max()) + 1
    key = edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64)
    key_t = edge_index[1].to(torch.int64) * n + edge_index[0].to(torch.int64)
    key = torch.unique(torch.cat([key, key_t]))
    return torch.stack([key // n, key % n])


def remove_self_loops(edge_index):
    m = edge_index[0] != edge_index[1]
    return edge_index[:, m]


def add_self_loops(edge_index, num_nodes):
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1)


def to_undirected(edge_index, num_nodes=None):
    """PyG to_undirected semantics used by load_ogbn (trainer_node_classification.py:574): both
    directions, duplicates removed, sorted by (row, col)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    key = torch.cat([edge_index[0] * n + edge_index[1], edge_index[1] * n + edge_index[0]])
    key = torch.unique(key)
    return torch.stack([key // n, key % n])


def getMLP(neurons, activation=nn.GELU, bias=True, dropout=0.1, last_dropout=False, normfun='layernorm'):
    """utils.py:885-908: Linear/Norm/Act/Dropout stack; len<2 -> Identity, len==2 -> one Linear."""
    if len(neurons) in [0, 1]:
        return nn.Identity()
    if len(neurons) == 2:
        return nn.Linear(*neurons)
    layers = []
    n = len(neurons) - 1
    for i in range(n - 1):
        norm = nn.LayerNorm(neurons[i + 1]) if normfun == 'layernorm' else nn.BatchNorm1d(neurons[i + 1])
        layers.extend([nn.Linear(neurons[i], neurons[i + 1], bias=bias), norm, activation(), nn.Dropout(dropout)])
    layers.append(nn.Linear(neurons[n - 1], neurons[n], bias=bias))
    if last_dropout:
        layers.append(nn.Dropout(dropout))
    return nn.Sequential(*layers)


_DEPTH = {'25': 1, '12': 2, '6': 3, '3': 4}


def get_partial_sorted_idx(arr, mode='top25'):
    """Repeated median split (utils.py:910-941).  'top' = smaller values; topK/bottomK keeps halving
    (50 -> 25 -> 12 -> 6 -> 3 %), each cut taken against the median of the *current* subset but
    applied to the full array."""
    arr = tonp(arr).reshape(-1)
    top = 'top' in mode
    idx = np.where(arr <= np.median(arr))[0] if top else np.where(arr >= np.median(arr))[0]
    suffix = mode[3:] if top else mode[6:]
    for _ in range(_DEPTH.get(suffix, 0)):
        med = np.median(arr[idx])
        idx = np.where(arr <= med)[0] if top else np.where(arr >= med)[0]
    return idx


def graph_analyze(N_nodes, edge_index):
    """Out-/in-degree per node (utils.py:300-334), as two bincounts instead of a per-edge dict loop."""
    ei = edge_index.to(torch.int64)
    degs_ori = torch.bincount(ei[0], minlength=N_nodes)[:N_nodes]
    degs_dst = torch.bincount(ei[1], minlength=N_nodes)[:N_nodes]
    return tonp(degs_ori), tonp(degs_dst)


def save_graph_analyze(N_nodes, data, use_special_split, verbose=True):
    """Head/tail/isolated node sets (utils.py:680-729)."""
    data.N_nodes = N_nodes
    degs_ori, degs_dst = graph_analyze(N_nodes, data.edge_index)
    dev = data.x.device

    def mask_of(idx):
        m = torch.zeros(N_nodes, dtype=torch.bool, device=dev)
        m[torch.as_tensor(idx, device=dev)] = True
        return m

    if not use_special_split:
        data.small_deg_idx = get_partial_sorted_idx(degs_dst, 'top3')
        data.large_deg_idx = get_partial_sorted_idx(degs_dst, 'bottom3')
        data.small_deg_mask = mask_of(data.small_deg_idx)
        data.large_deg_mask = mask_of(data.large_deg_idx)
    else:
        _idx = get_partial_sorted_idx(degs_dst, 'top6')
        _idx = _idx[np.array(degs_dst)[_idx].argsort()]
        data.zero_deg_idx = _idx[:len(_idx) // 2]
        data.small_deg_idx = _idx[len(_idx) // 2:]
        data.large_deg_idx = get_partial_sorted_idx(degs_dst, 'bottom3')
        data.zero_deg_mask = mask_of(data.zero_deg_idx)
        data.small_deg_mask = mask_of(data.small_deg_idx)
        data.large_deg_mask = mask_of(data.large_deg_idx)
        if verbose:
            print(f'\n\n\n  isolation ratio is:   {len(data.zero_deg_idx) / N_nodes * 100:.2f} %')
        craft_isolation_v2(data, verbose)


def craft_isolation_v2(data, verbose=True):
    """Removes every non-self-loop edge that touches an 'isolated' node, keeping edge order
    (utils.py:731-752)."""
    ei = data.edge_index
    z = data.zero_deg_mask
    drop = (ei[0] != ei[1]) & (z[ei[0]] | z[ei[1]])
    crafted = ei[:, ~drop]
    if verbose:
        print(f'removed < {int(drop.sum())} > edge; shape change: {ei.shape} ›› {crafted.shape}')
    data.edge_index_bkup = ei
    data.edge_index = crafted


def save_model(net, cwd):                         # utils.py:958-960
    torch.save(net.state_dict(), cwd)
    print(f'‹‹‹‹‹‹‹---  Saved @ :{cwd}\n\n\n')


def load_model(net, cwd, verbose=True, strict=True, multiGPU=False):   # utils.py:962-986
    if not os.path.exists(cwd):
        if verbose:
            print(f'---››››  !!! FileNotFound when load_model: {cwd}\n\n\n')
        return
    sd = torch.load(cwd, map_location=lambda storage, loc: storage)
    if multiGPU:
        sd = {k[7:]: v for k, v in sd.items()}    # strip DataParallel's `module.`
        net.load_state_dict(sd)
    else:
        net.load_state_dict(sd, strict=strict)
    if verbose:
        print(f'---››››  LOAD success! from {cwd}\n\n\n')
