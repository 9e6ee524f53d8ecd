"""TEST INFRASTRUCTURE ONLY — names imported by utils.py:7 / drop_tricks.py:8."""
import torch
from . import num_nodes  # noqa: F401


def remove_self_loops(edge_index, edge_attr=None):
    m = edge_index[0] != edge_index[1]
    return edge_index[:, m], (None if edge_attr is None else edge_attr[m])


def add_self_loops(edge_index, edge_weight=None, fill_value=1., num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1), edge_weight


def _absent(*a, **k):
    raise RuntimeError('torch_geometric.utils function not available (stub)')


def dropout_adj(edge_index, edge_attr=None, p=0.5, force_undirected=False, num_nodes=None, training=True):
    """Published PyG semantics: keep each edge with probability 1-p when training."""
    if not training or p == 0.0:
        return edge_index, edge_attr
    keep = torch.rand(edge_index.shape[1], device=edge_index.device) >= p
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    m = torch.zeros(n, dtype=torch.bool)
    m[subset] = True
    keep = m[edge_index[0]] & m[edge_index[1]]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def to_undirected(edge_index, num_nodes=None):
    """Published PyG semantics: union with the transpose, duplicates removed, sorted by (row, col)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    key = torch.unique(torch.cat([edge_index[0] * n + edge_index[1], edge_index[1] * n + edge_index[0]]))
    return torch.stack([key // n, key % n])


to_networkx = negative_sampling = _absent
