"""Graph-drop tricks — the API of the reference's GNN_model/drop_tricks.py (DropEdge :13, DropNode :26,
FastGCN :47, LADIES :71, DroppedEdges :120, DropoutTrick :127), kept so that every `--type_trick`
name still constructs and runs.

Exactly as in the reference, TricksComb.forward computes the dropped edge lists and then ignores
them (GCN.py:101,111,115: the aggregation always runs on the graph cached from the first call), so
these are inert for the results.  They are plain index arithmetic on torch tensors, without the
torch_geometric / torch_scatter dependency of the reference.
"""
import torch
from torch import nn


def _node_count(edge_index, num_nodes):
    return int(edge_index.max()) + 1 if num_nodes is None else num_nodes


def _induced_subgraph(keep_nodes, edge_index, edge_attr):
    """Edges whose two endpoints are both kept (PyG `subgraph` without relabelling)."""
    sel = keep_nodes[edge_index[0]] & keep_nodes[edge_index[1]]
    return edge_index[:, sel], (edge_attr[sel] if edge_attr is not None else None)


def _importance(edge_index, edge_weight, n):
    """q(u) ~ sum over incident edges of w^2 (FastGCN / LADIES sampling distribution)."""
    acc = torch.zeros(n, dtype=edge_weight.dtype, device=edge_weight.device)
    return acc.index_add(0, edge_index[1], edge_weight * edge_weight)


def _node_mask(n, picked, device):
    m = torch.zeros(n, dtype=torch.bool, device=device)
    m[picked.to(device)] = True
    return m


class _Sampler(nn.Module):
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate


class DropEdge(_Sampler):
    """Keeps every edge independently with probability 1 - drop_rate (training only)."""
    undirected = False

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training or self.drop_rate == 0.0:
            return edge_index, edge_attr
        sel = torch.rand(edge_index.shape[1], device=edge_index.device) >= self.drop_rate
        return edge_index[:, sel], (edge_attr[sel] if edge_attr is not None else None)


class DropNode(_Sampler):
    """Uniform Bernoulli node sample, induced subgraph."""

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return edge_index, edge_attr
        n = _node_count(edge_index, num_nodes)
        keep = torch.bernoulli(torch.full((n,), 1.0 - self.drop_rate)).bool().to(edge_index.device)
        return _induced_subgraph(keep, edge_index, edge_attr)


class FastGCN(_Sampler):
    """Importance-sampled nodes (without replacement), induced subgraph."""

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return edge_index, edge_attr
        n = _node_count(edge_index, num_nodes)
        w = torch.ones(edge_index.shape[1], device=edge_index.device) if edge_weight is None else edge_weight
        picked = torch.multinomial(_importance(edge_index, w, n), int(n * (1 - self.drop_rate)), replacement=False)
        return _induced_subgraph(_node_mask(n, picked, edge_index.device), edge_index, edge_attr)


class LADIES(_Sampler):
    """Layer-dependent importance sampling: each layer samples among the rows kept by the layer above."""

    def __init__(self, drop_rate, num_layers):
        super().__init__(drop_rate)
        self.num_layers = num_layers

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return [(edge_index, edge_attr)]
        n = _node_count(edge_index, num_nodes)
        w = torch.ones(edge_index.shape[1], device=edge_index.device) if edge_weight is None else edge_weight
        live = torch.ones_like(w, dtype=torch.bool)
        per_layer = []
        for _ in range(self.num_layers):
            q = _importance(edge_index, torch.where(live, w, torch.zeros_like(w)), n)
            picked = torch.multinomial(q, int(n * (1 - self.drop_rate)), replacement=False)
            keep = _node_mask(n, picked, edge_index.device)
            live = keep[edge_index[0]]
            per_layer.append(_induced_subgraph(keep, edge_index, edge_attr))
        return per_layer[::-1]


class DroppedEdges(list):
    """List of (edge_index, weight) per layer; a single entry serves every layer."""

    def __getitem__(self, i):
        return super().__getitem__(0 if len(self) == 1 else i)


_BY_NAME = (('DropEdge', DropEdge), ('DropNode', DropNode), ('FastGCN', FastGCN), ('LADIES', LADIES))


class DropoutTrick(nn.Module):
    """Picks the sampler named inside `type_trick` (first match in the order above) or none."""

    def __init__(self, args):
        super().__init__()
        self.type_trick, self.num_layers = args.type_trick, args.num_layers
        self.layerwise_drop = args.layerwise_dropout
        self.graph_dropout = None
        for name, cls in _BY_NAME:
            if name in self.type_trick:
                if cls is LADIES:
                    assert self.layerwise_drop, 'LADIES requires layer-wise dropout flag on'
                    self.graph_dropout = LADIES(args.graph_dropout, args.num_layers)
                else:
                    self.graph_dropout = cls(args.graph_dropout)
                break

    def forward(self, edge_index, edge_weight=None, adj_norm=False, num_nodes=-1):
        if adj_norm:
            raise NotImplementedError('adj_norm (PyG gcn_norm) is never requested by TricksComb (GCN.py:101)')
        if self.graph_dropout is None:
            return DroppedEdges([(edge_index, edge_weight)])
        draw = lambda: self.graph_dropout(edge_index, edge_attr=edge_weight, edge_weight=edge_weight)   # noqa: E731
        if isinstance(self.graph_dropout, LADIES):
            layers = draw()
            return DroppedEdges([layers[-1]]) if layers else DroppedEdges()   # the reference keeps the last entry (:152-156)
        return DroppedEdges([draw() for _ in range(self.num_layers if self.layerwise_drop else 1)])
