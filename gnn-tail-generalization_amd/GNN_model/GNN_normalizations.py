"""TeacherGNN wrapper — drop-in for the reference's GNN_model/GNN_normalizations.py:9-73."""
import torch
from torch import nn

from ..utils import D
from .GCN import TricksComb
from .norm_tricks import *  # noqa: F401,F403  (the reference re-exports these names)


class GNN_norm(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.model = TricksComb(args)

    def forward(self, x, edge_index):
        return self.model.forward(x, edge_index)


class TeacherGNN(nn.Module):
    """Teacher GCN with structural embeddings.  Mutates `args` like the reference does
    (num_classes := dim_commonEmb, num_feats := dim_learnable_input; :13-22)."""

    def __init__(self, args, proj2class=None):
        super().__init__()
        proj2class = proj2class or nn.Identity()
        args.num_classes_bkup = args.num_classes
        args.num_classes = args.dim_commonEmb
        self.args = args
        if self.args.dim_learnable_input > 0:
            self.embs = nn.Parameter(torch.randn(args.N_nodes, args.dim_learnable_input) * 0.001, requires_grad=True)
            self.args.num_feats_bkup = self.args.num_feats
            self.args.num_feats = self.args.dim_learnable_input
        self.model = GNN_norm(args)
        self.proj2linkp = nn.Identity()
        self.proj2class = proj2class
        self.dglgraph = None
        self.se_reg_all = None
        self.out = None

    def forward(self, x, edge_index):
        if self.args.TeacherGNN.change_to_featureless:
            x = x * 0
        if self.args.dim_learnable_input > 0:
            x = self.embs
        commonEmb, self.se_reg_all = self.model(x, edge_index)
        self.out = commonEmb
        return commonEmb

    def get_3_embs(self, x, edge_index, mask=None, want_heads=True):
        commonEmb = self.forward(x, edge_index)
        emb4classi_full = self.proj2class(commonEmb)
        if want_heads:
            emb4classi = emb4classi_full[mask] if mask is not None else emb4classi_full
            emb4linkp = self.proj2linkp(commonEmb)
        else:
            emb4linkp = emb4classi = None
        res = D()
        res.commonEmb, res.emb4classi, res.emb4classi_full, res.emb4linkp = commonEmb, emb4classi, emb4classi_full, emb4linkp
        return res

    def get_emb4linkp(self, x, edge_index, mask=None):
        # the reference unpacks the D namespace as a tuple here (:59) and so always raises; it is
        # unreachable in coldbrew mode.  This returns what that method was meant to return.
        return self.get_3_embs(x, edge_index, want_heads=True).emb4linkp

    def graph2commonEmb(self, x, edge_index, train_mask):
        commonEmb = self.forward(x, edge_index)
        return commonEmb[train_mask], commonEmb
