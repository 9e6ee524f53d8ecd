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


# ---------------------------------------------------------------------------------------------------------
# Graph analysis in front of the path.  Device tensors run on the hand-written HIP kernels of
# csrc/cb_ingest.hip (histogram / order-preserving compaction / sort-unique, through the C ABI); host tensors (the small
# fixtures of the CPU tests, host-side data preparation) use the equivalent tensor formulation below.  Both produce exactly
# what the reference's per-edge Python loops produce (tests/golden/utils_fixture.pt).
# ---------------------------------------------------------------------------------------------------------
def _dev_ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _symmetrize_device(edge_index, n):
    from . import _lib
    lib = _lib.load()
    src, dst = edge_index[0].to(torch.int64).contiguous(), edge_index[1].to(torch.int64).contiguous()
    E, dev = int(src.numel()), src.device
    out = torch.empty((2, max(2 * E, 1)), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.cb_symmetrize_workspace_bytes(E, n)
    ws = _dev_ws(wsb, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.cb_symmetrize_i64(_lib.ptr(src), _lib.ptr(dst), E, n, _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(cnt), _lib.ptr(bad),
                                         _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_symmetrize_i64')
        if int(bad.item()):
            raise ValueError(f'edge_index has {int(bad.item())} edges with an endpoint outside [0, {n})')
        return out[:, :int(cnt.item())].clone()


def ensure_symmetric(edge_index):
    """Union with the transpose, coalesced, sorted by (row, col) (utils.py:667-674)."""
    n = int(edge_index.max()) + 1
    if edge_index.is_cuda:
        return _symmetrize_device(edge_index, n)
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
    if edge_index.is_cuda:
        return _symmetrize_device(edge_index, n)
    key = torch.cat([edge_index[0] * n + edge_index[1], edge_index[1] * n + edge_index[0]])
    key = torch.unique(key)
    return torch.stack([key // n, key % n])


class HipLinear(nn.Linear):
    """nn.Linear (same parameters, initialisation and state_dict keys) whose product with a device matrix runs on the hand-written
    MFMA GEMM (gemm.linear: forward and both backward contractions) — the Linear layers of the proj2class head (trainer :304-305).
    CPU tensors (construction-time shape probes, host-side tests) fall through to torch."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            from . import gemm
            return gemm.linear(x, self.weight, self.bias)
        return super().forward(x)


def getMLP(neurons, activation=nn.GELU, bias=True, dropout=0.1, last_dropout=False, normfun='layernorm'):
    """utils.py:885-908: Linear/Norm/Act/Dropout stack; len<2 -> Identity, len==2 -> one Linear."""
    if len(neurons) in [0, 1]:
        return nn.Identity()
    if len(neurons) == 2:
        return HipLinear(*neurons)
    layers = []
    n = len(neurons) - 1
    for i in range(n - 1):
        norm = nn.LayerNorm(neurons[i + 1]) if normfun == 'layernorm' else nn.BatchNorm1d(neurons[i + 1])
        layers.extend([HipLinear(neurons[i], neurons[i + 1], bias=bias), norm, activation(), nn.Dropout(dropout)])
    layers.append(HipLinear(neurons[n - 1], neurons[n], bias=bias))
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


def _degrees_device(N_nodes, edge_index):
    """(out-degree, in-degree) int32 [N] on the device: cb_id_count_i64 over edge_index[0] / edge_index[1]."""
    from . import _lib
    lib = _lib.load()
    dev = edge_index.device
    out = []
    with torch.cuda.device(dev):
        for r in (0, 1):
            ids = edge_index[r].to(torch.int64).contiguous()
            cnt = torch.empty(N_nodes, dtype=torch.int32, device=dev)
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.cb_id_count_i64(_lib.ptr(ids), int(ids.numel()), N_nodes, _lib.ptr(cnt), _lib.ptr(bad), _lib.stream_ptr()),
                       'cb_id_count_i64')
            out.append((cnt, bad))
    n_bad = int(out[0][1].item()) + int(out[1][1].item())
    if n_bad:      # as _symmetrize_device: ids outside [0, N) are an input error, not something to drop silently
        raise ValueError(f'edge_index has {n_bad} endpoints outside [0, {N_nodes})')
    return out[0][0], out[1][0]


def graph_analyze(N_nodes, edge_index):
    """Out-/in-degree per node (utils.py:300-334) — two histograms instead of a per-edge dict loop."""
    if edge_index.is_cuda:
        d0, d1 = _degrees_device(N_nodes, edge_index)
        return tonp(d0).astype(np.int64), tonp(d1).astype(np.int64)
    ei = edge_index.to(torch.int64)
    degs_ori = torch.bincount(ei[0], minlength=N_nodes)[:N_nodes]
    degs_dst = torch.bincount(ei[1], minlength=N_nodes)[:N_nodes]
    return tonp(degs_ori), tonp(degs_dst)


def _median_of_range(hist, lo, hi):
    """np.median of the multiset {v : lo <= v <= hi} described by the value histogram (mean of the two middle order statistics)."""
    lo, hi = max(int(lo), 0), min(int(hi), len(hist) - 1)
    if hi < lo:
        return float('nan')
    c = np.cumsum(hist[lo:hi + 1])
    n = int(c[-1]) if len(c) else 0
    if n == 0:
        return float('nan')
    v1 = lo + int(np.searchsorted(c, (n - 1) // 2, side='right'))
    v2 = lo + int(np.searchsorted(c, n // 2, side='right'))
    return 0.5 * (v1 + v2)


def partial_sorted_select_device(vals, mode='top25'):
    """get_partial_sorted_idx (utils.py:910-941) for a non-negative int32 device vector: (ascending index tensor, bool mask).
    Every subset whose median the reference takes is {v <= t} ('top') or {v >= t} ('bottom'), so all medians are read off ONE
    value histogram (cb_value_hist_i32); the final np.where is one order-preserving compaction (cb_select_range_i32)."""
    from . import _lib
    lib = _lib.load()
    dev, n = vals.device, int(vals.numel())
    vals = vals.to(torch.int32).contiguous()
    top = 'top' in mode
    suffix = mode[3:] if top else mode[6:]
    with torch.cuda.device(dev):
        vmax = int(vals.max().item()) if n else 0
        hist_d = torch.empty(vmax + 1, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.cb_value_hist_i32(_lib.ptr(vals), n, vmax + 1, _lib.ptr(hist_d), _lib.ptr(bad), _lib.stream_ptr()), 'cb_value_hist_i32')
        hist = hist_d.cpu().numpy().astype(np.int64)
        if int(bad.item()):
            raise ValueError('partial_sorted_select_device expects non-negative values')
        lo, hi = 0, vmax
        for _ in range(1 + _DEPTH.get(suffix, 0)):
            med = _median_of_range(hist, lo, hi)
            if med != med:                                   # empty subset: the reference's np.where(arr <= nan) selects nothing
                lo, hi = 1, 0
                break
            if top:
                hi = int(np.floor(med))                      # arr <= med
            else:
                lo = int(np.ceil(med))                       # arr >= med
        idx = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        mask = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        wsb = lib.cb_compact_workspace_bytes(n)
        ws = _dev_ws(wsb, dev)
        _lib.check(lib.cb_select_range_i32(_lib.ptr(vals), n, lo, hi, _lib.ptr(idx), _lib.ptr(mask), _lib.ptr(cnt), _lib.ptr(ws), wsb,
                                           _lib.stream_ptr()), 'cb_select_range_i32')
        return idx[:int(cnt.item())].clone(), mask[:n].bool()


def save_graph_analyze(N_nodes, data, use_special_split, verbose=True):
    """Head/tail/isolated node sets (utils.py:680-729)."""
    data.N_nodes = N_nodes
    dev = data.x.device
    on_device = data.edge_index.is_cuda
    if on_device and not use_special_split:
        # large-graph branch (ogbn / synthetic power-law): everything stays on the device
        _, deg_dst = _degrees_device(N_nodes, data.edge_index)
        data.small_deg_idx, data.small_deg_mask = partial_sorted_select_device(deg_dst, 'top3')
        data.large_deg_idx, data.large_deg_mask = partial_sorted_select_device(deg_dst, 'bottom3')
        return
    degs_ori, degs_dst = graph_analyze(N_nodes, data.edge_index)

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
        # Planetoid-sized graphs: the tie order of the reference's np.argsort (utils.py:704) is part of the result, so the
        # N-sized index arithmetic stays on numpy; the per-edge work (degrees above, crafting below) runs on the device
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
    if ei.is_cuda:
        from . import _lib
        lib = _lib.load()
        src, dst = ei[0].to(torch.int64).contiguous(), ei[1].to(torch.int64).contiguous()
        E, dev = int(src.numel()), ei.device
        flag = z.to(device=dev, dtype=torch.uint8).contiguous()
        out = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        wsb = lib.cb_compact_workspace_bytes(E)
        ws = _dev_ws(wsb, dev)
        with torch.cuda.device(dev):
            _lib.check(lib.cb_craft_isolation_i64(_lib.ptr(src), _lib.ptr(dst), E, _lib.ptr(flag), int(flag.numel()), _lib.ptr(out[0]),
                                                  _lib.ptr(out[1]), _lib.ptr(cnt), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                       'cb_craft_isolation_i64')
            crafted = out[:, :int(cnt.item())].clone()
        n_drop = E - crafted.shape[1]
    else:
        drop = (ei[0] != ei[1]) & (z[ei[0]] | z[ei[1]])
        crafted = ei[:, ~drop]
        n_drop = int(drop.sum())
    if verbose:
        print(f'removed < {n_drop} > edge; shape change: {ei.shape} ›› {crafted.shape}')
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
