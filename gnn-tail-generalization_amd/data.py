"""Input contract of the path: a `torch_geometric.data.Data`-shaped namespace (x [N,F] float,
y [N] int64, edge_index [2,E] int64, train_mask/test_mask [N] bool, train_idx/test_idx), and
the seeded synthetic stand-ins of SURVEY.md §8(d) for the datasets that cannot be downloaded
(no network; torch_geometric / ogb are not installed).

A real PyG `Data` object works wherever this `Data` does (duck typing).  If
`data/<dataset>.pt` exists (a dict with x, y, edge_index, train_mask[, test_mask]) it is
loaded instead of the synthetic stand-in; so are the RAW files of the reference's own loaders
(Planetoid pickles, OGB csv files) when they lie under `data/` (datasets.py).
"""
import os

import torch

from .utils import add_self_loops, ensure_symmetric, remove_self_loops


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def keys(self):
        return list(vars(self).keys())

    def __repr__(self):
        parts = [f'{k}={list(v.shape)}' for k, v in vars(self).items() if isinstance(v, torch.Tensor)]
        return 'Data(' + ', '.join(parts) + ')'


# name -> (N, F, C, undirected pairs, self_loops, exponent gamma)
SYNTHETIC = {
    'S-tiny': (256, 16, 4, 700, True, 2.5),
    'S-cora': (2708, 1433, 7, 5278, True, 3.0),
    'S-pubmed': (19717, 500, 3, 44324, True, 3.0),
    'S-arxiv': (169343, 128, 40, 1157799, False, 2.5),
    'S-products': (2449029, 100, 47, 61859140, False, 2.5),
    'S-pl1M': (1000000, 128, 40, 4500000, True, 2.3),
    'S-pl10M': (10000000, 128, 40, 45000000, True, 2.3),
}
ALIASES = {'Cora': 'S-cora', 'Pubmed': 'S-pubmed', 'ogbn-arxiv': 'S-arxiv', 'ogbn-products': 'S-products'}


def _powerlaw_cdf(n, gamma, device):
    """Chung-Lu expected-degree weights w_i ~ (i + i0)^(-1/(gamma-1)); i0 caps the largest hub."""
    i0 = max(1.0, n ** 0.25)
    w = (torch.arange(n, dtype=torch.float64, device=device) + i0) ** (-1.0 / (gamma - 1.0))
    cdf = torch.cumsum(w, 0)
    return cdf / cdf[-1]


def _sample_nodes(cdf, k, gen):
    u = torch.rand(k, dtype=torch.float64, device=cdf.device, generator=gen)
    return torch.searchsorted(cdf, u).clamp_(max=cdf.numel() - 1)


def powerlaw_pairs(n, m, gamma, gen, device, cover=False):
    """Exactly m distinct undirected pairs (lo < hi) with power-law endpoint popularity and randomly
    permuted node ids ("no free locality").  cover=True additionally guarantees degree >= 1."""
    cdf = _powerlaw_cdf(n, gamma, device)
    perm = torch.randperm(n, device=device, generator=gen)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    if cover:   # one pair per node first, so that no node is isolated
        v = torch.arange(n, device=device)
        p = perm[_sample_nodes(cdf, n, gen)]
        p = torch.where(p == v, (v + 1) % n, p)
        keys = torch.unique(torch.minimum(v, p) * n + torch.maximum(v, p))
        if keys.numel() > m:
            raise ValueError('cover needs at least ~n/2 pairs')
    base = keys
    for _ in range(64):
        need = m - keys.numel()
        if need <= 0:
            break
        k = int(need * 1.2) + 1024
        a = perm[_sample_nodes(cdf, k, gen)]
        b = perm[_sample_nodes(cdf, k, gen)]
        ok = a != b
        a, b = a[ok], b[ok]
        new = torch.unique(torch.minimum(a, b) * n + torch.maximum(a, b))
        keys = torch.unique(torch.cat([keys, new]))
    if keys.numel() < m:
        raise RuntimeError('could not draw enough distinct pairs')
    if keys.numel() > m:    # drop a random surplus, never a cover pair
        is_base = torch.isin(keys, base) if base.numel() else torch.zeros_like(keys, dtype=torch.bool)
        extra = keys[~is_base]
        sel = torch.randperm(extra.numel(), device=device, generator=gen)[: m - int(is_base.sum())]
        keys = torch.cat([keys[is_base], extra[sel]])
    return torch.stack([keys // n, keys % n])


def synthetic_data(name, seed=0, device='cpu', n_override=None):
    """Seeded stand-in with the post-conditions of load_data / load_ogbn
    (trainer_node_classification.py:570-577,655-668): symmetric, coalesced, self-loops appended last
    for the Planetoid family; `to_undirected` without self-loops for the ogbn family."""
    name = ALIASES.get(name, name)
    n, f, c, pairs, loops, gamma = SYNTHETIC[name]
    if n_override:
        pairs = max(int(pairs * n_override / n), n_override)
        n = n_override
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(seed)
    p = powerlaw_pairs(n, pairs, gamma, gen, dev, cover=not loops)
    key = torch.cat([p[0] * n + p[1], p[1] * n + p[0]])
    key = torch.sort(key)[0]                       # coalesced order (row, col), as ensure_symmetric / to_undirected give
    edge_index = torch.stack([key // n, key % n])
    if loops:
        edge_index = add_self_loops(edge_index, n)
    if name in ('S-cora', 'S-pubmed', 'S-tiny'):   # NormalizeFeatures on a sparse bag-of-words-like matrix
        x = (torch.rand(n, f, device=dev, generator=gen) < max(0.0127, 4.0 / f)).float()
        x = x / x.sum(dim=1, keepdim=True).clamp(min=1)
    else:
        x = torch.rand(n, f, device=dev, generator=gen)
    y = torch.randint(0, c, (n,), device=dev, generator=gen)
    if name == 'S-cora':                            # trainer_node_classification.py:637-640: first 600 nodes train
        train_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        train_mask[:600] = True
    else:
        train_mask = torch.rand(n, device=dev, generator=gen) < 0.1
        train_mask[0] = True
    data = Data(x=x, y=y, edge_index=edge_index, train_mask=train_mask, test_mask=~train_mask, val_mask=None)
    data.train_idx = torch.where(train_mask)[0]
    data.test_idx = torch.where(~train_mask)[0]
    data.synthetic = True
    return data


def load_file(path, device):
    blob = torch.load(path, map_location='cpu', weights_only=True)     # a dict of tensors: no pickled code is executed
    d = Data(**{k: v for k, v in blob.items()})
    return d.to(device)


def _post_planetoid(data, num_nodes):
    """trainer_node_classification.py:655-662: symmetric, no self-loops, then one self-loop per node appended last."""
    ei = ensure_symmetric(data.edge_index)
    ei = remove_self_loops(ei)
    data.edge_index = add_self_loops(ei, num_nodes)
    return data


def load_data(dataset, which_run, trainer_self, root='data'):
    """Counterpart of trainer_node_classification.load_data (:616-670) + load_ogbn (:570-577) and the ogbn branch of trainer.__init__
    (:258-271).  Sources, in this order: `<root>/<dataset>.pt` (a dict of tensors), the RAW files the reference's own loaders keep under
    `<root>/` (Planetoid pickles, OGB csv files: datasets.py), else the seeded synthetic stand-in of the dataset's shape."""
    from . import datasets
    device = trainer_self.device
    path = os.path.join(root, f'{dataset}.pt')
    raw_pl = datasets.planetoid_raw_dir(root, dataset) if dataset in ('Cora', 'Citeseer', 'Pubmed') else None
    raw_ogb = datasets.ogb_dir(root, dataset) if dataset.startswith('ogbn') else None
    if os.path.isfile(path):
        data = load_file(path, device)
        if not dataset.startswith('ogbn'):
            data = _post_planetoid(data, data.x.shape[0])
    elif raw_pl is not None:
        print(f'[data] {dataset}: Planetoid raw files in {raw_pl}')
        data = Data(**datasets.read_planetoid(raw_pl, dataset)).to(device)
        if dataset == 'Cora':      # trainer_node_classification.py:637-640: the first 600 nodes train, all others test
            n = data.x.shape[0]
            data.train_mask = torch.arange(n, device=device) < 600
            data.test_mask = ~data.train_mask
        data = _post_planetoid(data, data.x.shape[0])
    elif raw_ogb is not None:
        print(f'[data] {dataset}: OGB raw files in {raw_ogb}')
        blob, split = datasets.read_ogbn(raw_ogb, device=device)
        data = Data(**blob).to(device)
        n = data.x.shape[0]
        data.train_mask = torch.zeros(n, dtype=torch.bool, device=device).index_fill_(0, split['train'].to(device), True)
        data.test_mask = torch.zeros(n, dtype=torch.bool, device=device).index_fill_(0, split['test'].to(device), True)
        data.val_mask = torch.zeros(n, dtype=torch.bool, device=device).index_fill_(0, split['valid'].to(device), True)
        data.train_idx = split['train'].to(device)      # (:261; test_idx = where(test_mask) below)
    else:
        print(f'[data] {dataset}: real files unavailable offline -> seeded synthetic stand-in '
              f'{ALIASES.get(dataset, dataset)} (SURVEY.md §8d)')
        data = synthetic_data(dataset, seed=0, device=device)
    if getattr(data, 'test_mask', None) is None:
        data.test_mask = ~data.train_mask
    if getattr(data, 'train_idx', None) is None:
        data.train_idx = torch.where(data.train_mask)[0]
    if getattr(data, 'test_idx', None) is None:
        data.test_idx = torch.where(data.test_mask)[0]
    return data
