"""TeacherGNN body on the MI355X HIP path — the drop-in for the reference's
GNN_model/GCN.py (TricksComb :18-150, GCNConv :152-277).

Same class names, constructor arguments, parameter names/shapes/registration order
(state_dict keys `layers_GCN.{i}.{weight,bias,le}`, `layers_MLP.{0,1}.*`,
`layers_norm.{i}.*`, `layers_res.{i}.*`) and the same forward contracts:
    TricksComb.forward(x, edge_index, want_les=False) -> (x, se_reg_all[, les])
    GCNConv.forward(graph, feat, weight=None, edge_weight=None) -> (rst, se_reg)
What changes is below the surface: the graph is a device CSR built once through the C ABI
(graph.CSRGraph instead of dgl.graph from Python lists, GCN.py:92-95), and the
transform / aggregate stages run as hand-written gfx950 kernels (ops.py).
"""
import os

import torch as th
import torch.nn.functional as F
from torch import nn
from torch.nn import init

from .. import gemm, ops, stack, trunk
from ..graph import CSRGraph, DGLError, build_graph  # noqa: F401
from .drop_tricks import DropoutTrick
from .norm_tricks import AcontainsB, appendNormLayer, run_norm_if_any
from .res_tricks import DenseConnection, InitialConnection, ResidualConnection


_BARE_NORMS = ('BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm')


class TricksComb(nn.Module):
    """Layer stack of the teacher: non-residual mode = GCNConv(F,H), (L-2) x GCNConv(H,H), GCNConv(H,C);
    residual mode (type_trick names Jumping/Initial/Residual/Dense) = Linear(F,H), L x GCNConv(H,H), Linear(H,C)."""

    # 'Initial' / 'Residual' connections (trunk.py) and the non-residual stack (stack.py) without a bare norm run as one fused autograd node at hidden 256
    # (CB_FUSED_TRUNK=0: one operator per stage everywhere — the path of every other configuration, for A/B runs)
    use_fused_trunk = os.environ.get('CB_FUSED_TRUNK', '1') != '0'

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.dglgraph = None            # name kept: the cached graph object (a CSRGraph here)
        self.alpha = args.res_alpha
        self.embedding_dropout = args.dropout
        for name, value in vars(args).items():   # every option is mirrored onto the module (reference GCN.py:26-27)
            setattr(self, name, value)
        self.cached = self.transductive = args.transductive
        if AcontainsB(self.type_trick, ['DropEdge', 'DropNode', 'FastGCN', 'LADIES']):
            self.cached = False
        self.has_residual_MLP = AcontainsB(self.type_trick, ['Jumping', 'Initial', 'Residual', 'Dense'])
        self.layers_GCN, self.layers_res = nn.ModuleList(), nn.ModuleList()
        self.layers_norm, self.layers_MLP = nn.ModuleList(), nn.ModuleList()
        # registration order below = the reference's, so equal seeds give bit-identical initial parameters
        se_first, se_mid, se_last = self.args.TeacherGNN.whetherHasSE
        H, L = self.dim_hidden, self.num_layers
        self.layers_MLP.append(nn.Linear(self.num_feats, H))
        if not self.has_residual_MLP:
            self.layers_GCN.append(self._conv(self.num_feats, H, se_first))
        for i in range(L):
            if self.has_residual_MLP or 0 < i < L - 1:   # hidden->hidden layers all take the "middle" SE flag
                self.layers_GCN.append(self._conv(H, H, se_mid))
            appendNormLayer(self, args, H if i < L - 1 else self.num_classes)
            res = self._res_layer(i)
            if res is not None:
                self.layers_res.append(res)
        self.graph_dropout = DropoutTrick(args)
        if not self.has_residual_MLP:
            self.layers_GCN.append(self._conv(H, self.num_classes, se_last))
        if AcontainsB(self.type_trick, ['Jumping']):
            head = self._dense((L + 1) * H, self.num_classes)
            if head is not None:
                self.layers_res.append(head)
        else:
            self.layers_MLP.append(nn.Linear(H, self.num_classes))
        if AcontainsB(self.type_trick, ['IdentityMapping']):
            self.lamda = args.lamda
        elif self.type_model in ('SGC', 'GCN'):
            self.lamda = 0. if self.type_model == 'SGC' else 1.

    def _conv(self, d_in, d_out, has_se):
        return GCNConv(d_in, d_out, cached=self.cached, args=self.args, whetherHasSE=has_se)

    def _dense(self, concat_dim, d_out):
        if self.layer_agg in ('concat', 'maxpool'):
            return DenseConnection(concat_dim, d_out, self.layer_agg)
        if self.layer_agg == 'attention':
            return DenseConnection(self.dim_hidden, d_out, self.layer_agg)
        return None

    def _res_layer(self, i):
        """Per-layer skip connection; precedence Residual > Initial > Dense as in the reference (GCN.py:57-67)."""
        if 'Residual' in self.type_trick:
            return ResidualConnection(alpha=self.alpha)
        if 'Initial' in self.type_trick:
            return InitialConnection(alpha=self.alpha)
        if 'Dense' in self.type_trick:
            return self._dense((i + 2) * self.dim_hidden, self.dim_hidden)
        return None

    def _graph(self, edge_index):
        """First call builds the device CSR and caches it forever; later edge_index arguments are
        ignored exactly as in the reference (GCN.py:92-95)."""
        if self.dglgraph is None:
            self.dglgraph = build_graph(edge_index)
        return self.dglgraph

    def forward(self, x, edge_index, want_les=False, loss_rows=None, rows_only=False):
        """loss_rows, rows_only: see TeacherGNN.forward (extensions of the reference's signature; the general path uses loss_rows for its last aggregation)."""
        graph = self._graph(edge_index)
        new_adjs = self.graph_dropout(edge_index)      # computed and discarded, as in the reference (GCN.py:101,111)
        if self.use_fused_trunk and not getattr(graph, 'segmented', False) and trunk.eligible(self, x, want_les):
            return trunk.forward(self, x, graph, loss_rows=loss_rows, rows_only=rows_only)
        if self.use_fused_trunk and not getattr(graph, 'segmented', False) and stack.eligible(self, x, graph, want_les):
            return stack.forward(self, x, graph, loss_rows=loss_rows, rows_only=rows_only)      # the non-residual stack (NoRes...) at hidden 256
        if getattr(self.args, 'agg_dtype', 'f32') != 'f32' and not want_les:
            raise NotImplementedError("--agg_dtype=bf16 (bf16-stored aggregation rows) is built for the fused 'Initial' trunk "
                                      '(hidden width a multiple of 256); this configuration runs the fp32 operator path')
        return self._forward_modular(x, graph, new_adjs, want_les, loss_rows)

    def _last_grad_rows(self, graph, want_les, loss_rows):
        """The mask the LAST aggregation's backward may restrict itself to (ops._AggregateFn), or None.  Every stage between that aggregation and
        the logits must act row by row: ReLU, the residual mixes, dropout and the output Linear do; of the norms only NodeNorm (per-row statistics,
        norm_tricks.py:53-84) — the others take column statistics, whose backward reaches every row."""
        from ..tuning import T
        if (loss_rows is None or want_les or not self.training or not ops.loss_rows_enabled() or hasattr(graph, 'part') or getattr(graph, 'segmented', False)
                or getattr(graph, 'rowptr_t', None) is None or (self.args.type_trick in _BARE_NORMS and self.args.type_trick != 'NodeNorm')):
            return None
        mask, count = loss_rows if isinstance(loss_rows, (tuple, list)) else (loss_rows, None)
        n = int(mask.shape[0])
        if mask.dtype != th.bool or n != graph.N or (n < T.rowsparse_min_nodes and not getattr(graph, 'rowsparse_small_ok', False)):
            return None
        if count is None:
            count = int(mask.sum())
        if not (0 < int(count) <= T.rowsparse_s0_limit * n) or not graph.support_plan_pays():
            return None
        return mask

    def _forward_modular(self, x, graph, new_adjs, want_les, loss_rows=None):
        """General path: one HIP operator per stage, any trick combination."""
        L, train = self.num_layers, self.training
        last_rows = self._last_grad_rows(graph, want_les, loss_rows)
        row0 = getattr(graph, 'row_offset', 0)     # first global row of this rank's shard (0 on one GPU)
        drop = lambda t, p: ops.dropout(t, p, train, offset=row0 * t.shape[1])   # noqa: E731
        x_list, les, se_reg_all = [], [], None
        if self.has_residual_MLP:                   # GCN.py:103-107
            lin = self.layers_MLP[0]
            x = gemm.linear(drop(x, self.embedding_dropout), lin.weight, lin.bias, relu=True)
            x_list.append(x)
        norms_run = self.args.type_trick in _BARE_NORMS
        mixes = AcontainsB(self.type_trick, ['Initial', 'Dense', 'Residual'])
        for i in range(L):                          # GCN.py:109-131
            _unused_edge_index, _ = new_adjs[i]
            act = self.has_residual_MLP or i < L - 1
            fuse_relu = act and not norms_run and not want_les   # ReLU rides in the aggregation epilogue
            x, se_reg = self.layers_GCN[i](graph, drop(x, self.dropout), _fused_relu=fuse_relu, _grad_rows=last_rows if i == L - 1 else None)
            if se_reg is not None:
                # intended semantics of GCN.py:116-120 (sum of the per-layer norms); the reference's in-place `+=`
                # on a tensor autograd saved breaks backward for >= 2 SE layers
                se_reg_all = se_reg if se_reg_all is None else se_reg_all + se_reg
            x = run_norm_if_any(self, x, i)
            if want_les:
                les.append(x.clone().detach())
            if act and not fuse_relu:
                x = F.relu(x)
            x_list.append(x)
            if mixes:
                x = self.layers_res[i](x_list)
        x = drop(x, self.args.dropout)              # on the logits in non-residual mode (GCN.py:133)
        if self.has_residual_MLP:
            if AcontainsB(self.type_trick, ['Jumping']):
                x = self.layers_res[0](x_list)      # index 0, as the reference writes it (GCN.py:136)
            else:
                lin = self.layers_MLP[-1]
                x = gemm.linear(x, lin.weight, lin.bias)
        if want_les:
            return x, se_reg_all, th.cat(les, dim=-1)
        return x, se_reg_all

    def get_se_dim(self, x, edge_index):
        return self.collect_SE(x, edge_index).shape[-1]

    def collect_SE(self, x, edge_index):
        """Per-layer pre-activation outputs concatenated on the feature axis (teacher -> SEMLP hand-off)."""
        return self.forward(x, edge_index, want_les=1)[2]


class GCNConv(nn.Module):
    """Cold Brew's graph convolution:  Y = D_in^-1/2 . A^T . ( (X . D_out^-1/2) W + E ) + bias,  with the
    optional structural embedding E (`le`, one row per node) and its Frobenius norm as regulariser.
    Always multiplies by W before aggregating.  Returns (Y, ||E||_F or None)   (reference GCN.py:152-258)."""

    def __init__(self, in_feats, out_feats, norm='both', weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False, cached=None, args=None, whetherHasSE=False):
        super().__init__()
        self.args = args
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        self._activation = activation
        self.whetherHasSE = whetherHasSE
        # registration order weight, bias, le and the initialisers match the reference (xavier / zeros / randn)
        self.weight = nn.Parameter(th.empty(in_feats, out_feats)) if weight else None
        self.bias = nn.Parameter(th.empty(out_feats)) if bias else None
        self.reset_parameters()
        if whetherHasSE:
            self.le = nn.Parameter(th.randn(args.N_nodes, out_feats), requires_grad=True)

    def reset_parameters(self):
        if self.weight is not None:
            init.xavier_uniform_(self.weight)
        if self.bias is not None:
            init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def _pick_weight(self, weight):
        if weight is not None and self.weight is not None:
            raise DGLError('External weight is provided while at the same time the module has defined its own '
                           'weight parameter. Please create the module with flag weight=False.')
        w = self.weight if weight is None else weight
        if w is None:
            raise NotImplementedError('GCNConv without a weight is not reachable from TricksComb')
        return w

    def forward(self, graph, feat, weight=None, edge_weight=None, _fused_relu=False, _grad_rows=None):
        if not self._allow_zero_in_degree:
            graph.check_zero_in_degree()                       # GCN.py:187-197
        if edge_weight is not None:                            # GCN.py:199-202: u_mul_e instead of copy_src (TricksComb never passes one)
            assert edge_weight.shape[0] == graph.number_of_edges()
        if self._norm != 'both':
            raise NotImplementedError("only norm='both' is reachable from TricksComb")
        w = self._pick_weight(weight)
        h, se_reg = ops.transform(feat, graph.norm_out, w, self.le if self.whetherHasSE else None, graph)   # :213,225,230-236
        self.se_norm = None if se_reg is None else se_reg.detach()      # last ||le||_F (ops.fold_se_reg)
        rst = ops.aggregate(graph, h, row_scale=graph.norm_in, bias=self.bias, relu=_fused_relu, edge_weight=edge_weight,
                            grad_rows=_grad_rows)                                                                         # :238,250,253
        return (rst if self._activation is None else self._activation(rst)), se_reg

    def extra_repr(self):
        return f'in={self._in_feats}, out={self._out_feats}, normalization={self._norm}'


def tonp(arr):
    import numpy as np
    if type(arr) is th.Tensor:
        return arr.detach().cpu().data.numpy()
    return np.asarray(arr)
