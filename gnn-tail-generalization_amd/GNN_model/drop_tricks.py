"""Graph-drop tricks — API of the reference's GNN_model/drop_tricks.py kept.

As in the reference, TricksComb.forward computes the dropped edge lists and then ignores
them (GCN.py:101,111,115: the aggregation always runs on the graph cached from the first
call), so these are inert for the results; they are plain index arithmetic in torch, with no
torch_geometric / torch_scatter dependency.
"""
import torch
from torch import nn


def _num_nodes(edge_index, num_nodes=None):
    return int(edge_index.max()) + 1 if num_nodes is None else num_nodes


def _induced(subnodes, edge_index, edge_attr, num_nodes):
    keep = torch.zeros(num_nodes, dtype=torch.bool, device=edge_index.device)
    keep[subnodes.to(edge_index.device)] = True
    m = keep[edge_index[0]] & keep[edge_index[1]]
    return edge_index[:, m], (None if edge_attr is None else edge_attr[m])


def _in_weight(edge_index, edge_weight, num_nodes):
    w = torch.zeros(num_nodes, dtype=edge_weight.dtype, device=edge_weight.device)
    return w.index_add(0, edge_index[1], edge_weight ** 2)      # q(u) ~ sum_v w^2(u,v)


class DropEdge(nn.Module):                   # drop_tricks.py:13-24: keep each edge with prob 1 - rate
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate, self.undirected = drop_rate, False

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training or self.drop_rate == 0.0:
            return edge_index, edge_attr
        keep = torch.rand(edge_index.shape[1], device=edge_index.device) >= self.drop_rate
        return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


class DropNode(nn.Module):                   # drop_tricks.py:26-45: uniform node sample, induced subgraph
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return edge_index, edge_attr
        n = _num_nodes(edge_index, num_nodes)
        keep = torch.bernoulli(torch.full((n,), 1 - self.drop_rate)).to(torch.bool)
        return _induced(keep.nonzero().reshape(-1), edge_index, edge_attr, n)


class FastGCN(nn.Module):                    # drop_tricks.py:47-69: importance-sampled nodes
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return edge_index, edge_attr
        n = _num_nodes(edge_index, num_nodes)
        if edge_weight is None:
            edge_weight = torch.ones((edge_index.shape[1],), device=edge_index.device)
        sub = torch.multinomial(_in_weight(edge_index, edge_weight, n), int(n * (1 - self.drop_rate)), replacement=False)
        return _induced(sub, edge_index, edge_attr, n)


class LADIES(nn.Module):                     # drop_tricks.py:71-118: layer-dependent importance sampling
    def __init__(self, drop_rate, num_layers):
        super().__init__()
        self.drop_rate, self.num_layers = drop_rate, num_layers

    def forward(self, edge_index, edge_attr=None, edge_weight=None, num_nodes=None):
        if not self.training:
            return [(edge_index, edge_attr)]
        n = _num_nodes(edge_index, num_nodes)
        if edge_weight is None:
            edge_weight = torch.ones((edge_index.shape[1],), device=edge_index.device)
        sampled = []
        row_mask = torch.ones(edge_weight.shape[0], dtype=torch.bool, device=edge_index.device)
        for _ in range(self.num_layers):
            w = torch.where(row_mask, edge_weight, torch.zeros_like(edge_weight))
            sub = torch.multinomial(_in_weight(edge_index, w, n), int(n * (1 - self.drop_rate)), replacement=False)
            keep = torch.zeros(n, dtype=torch.bool, device=edge_index.device)
            keep[sub] = True
            row_mask = keep[edge_index[0]]
            sampled.append(_induced(sub, edge_index, edge_attr, n))
        sampled.reverse()
        return sampled


class DroppedEdges(list):                    # drop_tricks.py:120-125: a single entry serves every layer
    def __getitem__(self, i):
        return super().__getitem__(0 if len(self) == 1 else i)


class DropoutTrick(nn.Module):               # drop_tricks.py:127-172
    def __init__(self, args):
        super().__init__()
        self.type_trick = args.type_trick
        self.num_layers = args.num_layers
        self.layerwise_drop = args.layerwise_dropout
        t = self.type_trick
        if 'DropEdge' in t:
            self.graph_dropout = DropEdge(args.graph_dropout)
        elif 'DropNode' in t:
            self.graph_dropout = DropNode(args.graph_dropout)
        elif 'FastGCN' in t:
            self.graph_dropout = FastGCN(args.graph_dropout)
        elif 'LADIES' in t:
            assert self.layerwise_drop, 'LADIES requires layer-wise dropout flag on'
            self.graph_dropout = LADIES(args.graph_dropout, args.num_layers)
        else:
            self.graph_dropout = None

    def forward(self, edge_index, edge_weight=None, adj_norm=False, num_nodes=-1):
        if adj_norm:
            raise NotImplementedError('adj_norm (PyG gcn_norm) is never requested by TricksComb (GCN.py:101)')
        if self.graph_dropout is None:
            return DroppedEdges([(edge_index, edge_weight)])
        if 'LADIES' in self.type_trick:
            new_adjs = DroppedEdges()
            for dp_edges, dp_weights in self.graph_dropout(edge_index, edge_attr=edge_weight, edge_weight=edge_weight):
                new_adjs = DroppedEdges([(dp_edges, dp_weights)])   # the reference keeps only the last entry (:152-156)
            return new_adjs
        new_adjs = DroppedEdges()
        for _ in range(self.num_layers if self.layerwise_drop else 1):
            new_adjs.append(self.graph_dropout(edge_index, edge_attr=edge_weight, edge_weight=edge_weight))
        return new_adjs
