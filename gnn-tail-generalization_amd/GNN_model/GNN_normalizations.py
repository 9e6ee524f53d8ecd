"""TeacherGNN wrapper on the HIP path — same surface as the reference's
GNN_model/GNN_normalizations.py (TeacherGNN :9-65, GNN_norm :67-73): constructor `(args, proj2class)`,
`forward`, `get_3_embs`, `graph2commonEmb`, attributes `se_reg_all` / `out` / `model.model`.
"""
import torch
from torch import nn

from ..utils import D
from .GCN import TricksComb
from .norm_tricks import *  # noqa: F401,F403  (names re-exported by the reference module as well)


class GNN_norm(nn.Module):
    """Pass-through holder: `.model` is the TricksComb body (state_dict prefix `model.model.`)."""

    def __init__(self, args):
        super().__init__()
        self.model = TricksComb(args)

    def forward(self, x, edge_index, loss_rows=None, rows_only=False):
        return self.model.forward(x, edge_index, loss_rows=loss_rows, rows_only=rows_only)


class TeacherGNN(nn.Module):
    """Teacher GCN with structural embeddings (Cold Brew).

    Side effects on `args` are part of the contract (reference :13-22): `num_classes` becomes the common
    embedding width (`dim_commonEmb`), and in learnable-input mode `num_feats` becomes
    `dim_learnable_input` (the originals are kept as `*_bkup`)."""

    def __init__(self, args, proj2class=None):
        super().__init__()
        self.args = args
        args.num_classes_bkup, args.num_classes = args.num_classes, args.dim_commonEmb
        learn_dim = args.dim_learnable_input
        if learn_dim > 0:   # featureless mode: trainable node inputs replace x
            self.embs = nn.Parameter(torch.randn(args.N_nodes, learn_dim) * 0.001, requires_grad=True)
            args.num_feats_bkup, args.num_feats = args.num_feats, learn_dim
        self.model = GNN_norm(args)
        self.proj2linkp = nn.Identity()
        self.proj2class = proj2class if proj2class else nn.Identity()
        self.dglgraph = None
        self.se_reg_all, self.out = None, None

    def _input(self, x):
        if self.args.TeacherGNN.change_to_featureless:
            x = x * 0
        return self.embs if self.args.dim_learnable_input > 0 else x

    def forward(self, x, edge_index, loss_rows=None, rows_only=False):
        """loss_rows (extension, default None = the reference's call): (bool mask [N], count) — the caller's promise that the objective it
        builds on this forward's output puts gradient into the rows of the mask only (the masked loss of trainer…:390-391; any head between
        the output and the loss must be row-wise, as proj2class is).  The fused trunk's backward then skips the rows that stay zero; the
        promise is verified on the device every step (ops.check_rows_zero), a broken one raises and leaves the weights untouched.
        rows_only (extension, with loss_rows): the caller's second promise — it READS this forward's output (the return value, `self.out`) in the rows
        of the mask only.  A training forward of the fused trunk then evaluates its last layer and the output Linear on those rows; every other row
        of the output (and so of `self.out`, `res.commonEmb`, `res.emb4classi_full`) is returned as NaN (tuning.T.rows_only_poison): a reader that
        breaks the promise — e.g. the reference's edge-wise loss, which takes res.commonEmb of the same forward (trainer…:417-418) — fails loudly."""
        self.out, self.se_reg_all = self.model(self._input(x), edge_index, loss_rows=loss_rows, rows_only=rows_only)
        return self.out

    def get_3_embs(self, x, edge_index, mask=None, want_heads=True, loss_rows=None, rows_only=False):
        res = D()
        res.commonEmb = self.forward(x, edge_index, loss_rows=loss_rows, rows_only=rows_only)
        res.emb4classi_full = self.proj2class(res.commonEmb)
        res.emb4classi = res.emb4linkp = None
        # (extension) a rows-only forward of the fused trunk also hands back the logits of the loss rows as the compact matrix they were computed as,
        # with the mask they belong to: (== emb4classi_full[mask], mask).  A loss built on it (trainer_node_classification.training_loss) sends its
        # gradient back compact — no [N, C] loss pass, no row gather.  None otherwise.
        rows = getattr(res.commonEmb, '_cb_rows', None)
        res.emb4classi_rows = (self.proj2class(rows[0]), rows[1]) if rows is not None else None
        if want_heads:
            if mask is None:
                res.emb4classi = res.emb4classi_full
            elif rows is not None and mask is rows[1]:
                res.emb4classi = res.emb4classi_rows[0]      # the reference's raw_logits = emb4classi_full[mask] (:45-47), never gathered
            else:
                res.emb4classi = res.emb4classi_full[mask]
            res.emb4linkp = self.proj2linkp(res.commonEmb)
        return res

    def get_emb4linkp(self, x, edge_index, mask=None):
        # the reference unpacks the result namespace as a tuple here (:59) and therefore always raises; the
        # method is unreachable in coldbrew mode.  This returns what it was meant to return.
        return self.get_3_embs(x, edge_index, want_heads=True).emb4linkp

    def graph2commonEmb(self, x, edge_index, train_mask):
        emb = self.forward(x, edge_index)
        return emb[train_mask], emb
