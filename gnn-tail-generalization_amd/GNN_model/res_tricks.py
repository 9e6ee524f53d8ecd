"""Skip-connection tricks on the HIP path — the classes of the reference's GNN_model/res_tricks.py
(ResidualConnection :7, InitialConnection :16, DenseConnection :25) with the same constructor
arguments, parameter names (`layer_transform`, `layer_att`) and arithmetic."""
import torch
from torch import nn


class _AlphaMix(nn.Module):
    """(1 - alpha) * Xs[-1] + alpha * Xs[pick]; a single entry passes through unchanged."""
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


class ResidualConnection(_AlphaMix):
    """Mixes with the previous layer's activation."""
    _pick = -2


class InitialConnection(_AlphaMix):
    """Mixes with the first entry (the output of the input Linear + ReLU)."""
    _pick = 0


def _concat(mod, Xs):
    from ..gemm import linear
    return linear(torch.cat(Xs, dim=-1), mod.layer_transform.weight, mod.layer_transform.bias)   # Linear on the MFMA GEMM


def _maxpool(mod, Xs):
    return torch.stack(Xs, dim=-1).amax(dim=-1)


def _attention(mod, Xs):
    """DAGNN-style retain scores over the k+1 stacked layers (n x (k+1) x c)."""
    from ..gemm import linear
    stacked = torch.stack(Xs, dim=1)
    n, k1, c = stacked.shape
    logits = linear(stacked.reshape(n * k1, c), mod.layer_att.weight, mod.layer_att.bias).reshape(n, k1, 1)
    retain = torch.sigmoid(logits.squeeze()).unsqueeze(1)
    return torch.matmul(retain, stacked).squeeze()


class DenseConnection(nn.Module):
    _POOL = {'concat': _concat, 'maxpool': _maxpool, 'attention': _attention}

    def __init__(self, in_dim, out_dim, aggregation='concat'):
        super().__init__()
        self.in_dim, self.out_dim, self.aggregation = in_dim, out_dim, aggregation
        if aggregation == 'concat':
            self.layer_transform = nn.Linear(in_dim, out_dim, bias=True)
        elif aggregation == 'attention':
            self.layer_att = nn.Linear(in_dim, 1, bias=True)

    def forward(self, Xs: list):
        assert len(Xs) >= 1
        if self.aggregation not in self._POOL:
            raise Exception('Unknown aggregation')
        return self._POOL[self.aggregation](self, Xs)
