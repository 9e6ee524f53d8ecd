"""Skip-connection tricks — same classes and semantics as the reference's
GNN_model/res_tricks.py:7-55 (ResidualConnection, InitialConnection, DenseConnection)."""
import torch
from torch import nn


class _AlphaMix(nn.Module):
    """(1 - alpha) * Xs[-1] + alpha * Xs[pick]; a single entry passes through."""
    _pick = -2

    def __init__(self, alpha=0.5):
        super().__init__()
        self.alpha = alpha

    def forward(self, Xs: list):
        assert len(Xs) >= 1
        if len(Xs) == 1:
            return Xs[-1]
        from ..ops import axpby
        return axpby(1 - self.alpha, Xs[-1], self.alpha, Xs[self._pick])


class ResidualConnection(_AlphaMix):     # res_tricks.py:7-14: mixes with the previous layer
    _pick = -2


class InitialConnection(_AlphaMix):      # res_tricks.py:16-23: mixes with the first entry (input MLP output)
    _pick = 0


class DenseConnection(nn.Module):        # res_tricks.py:25-55
    def __init__(self, in_dim, out_dim, aggregation='concat'):
        super().__init__()
        self.in_dim, self.out_dim, self.aggregation = in_dim, out_dim, aggregation
        if aggregation == 'concat':
            self.layer_transform = nn.Linear(in_dim, out_dim, bias=True)
        elif aggregation == 'attention':
            self.layer_att = nn.Linear(in_dim, 1, bias=True)

    def forward(self, Xs: list):
        assert len(Xs) >= 1
        if self.aggregation == 'concat':
            return self.layer_transform(torch.cat(Xs, dim=-1))
        if self.aggregation == 'maxpool':
            return torch.stack(Xs, dim=-1).max(dim=-1)[0]
        if self.aggregation == 'attention':          # DAGNN-style retain scores, n x (k+1) x c
            pps = torch.stack(Xs, dim=1)
            retain = torch.sigmoid(self.layer_att(pps).squeeze()).unsqueeze(1)
            return torch.matmul(retain, pps).squeeze()
        raise Exception('Unknown aggregation')
