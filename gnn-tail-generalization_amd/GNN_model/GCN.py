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
import math

import torch as th
import torch.nn.functional as F
from torch import nn
from torch.nn import init

from .. import gemm, ops, trunk
from ..graph import CSRGraph, DGLError
from .drop_tricks import DropoutTrick
from .norm_tricks import AcontainsB, appendNormLayer, run_norm_if_any
from .res_tricks import DenseConnection, InitialConnection, ResidualConnection


class TricksComb(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.dglgraph = None            # name kept: cached graph object (a CSRGraph here)
        self.alpha = args.res_alpha
        self.embedding_dropout = args.dropout
        for k, v in vars(args).items():   # the reference mirrors every option onto the module (GCN.py:26-27)
            setattr(self, k, v)
        self.cached = self.transductive = args.transductive
        if AcontainsB(self.type_trick, ['DropEdge', 'DropNode', 'FastGCN', 'LADIES']):
            self.cached = False
        # residual-style tricks keep the hidden width constant behind an input/output Linear (GCN.py:34-36)
        self.has_residual_MLP = AcontainsB(self.type_trick, ['Jumping', 'Initial', 'Residual', 'Dense'])
        se = self.args.TeacherGNN.whetherHasSE

        self.layers_GCN = nn.ModuleList([])
        self.layers_res = nn.ModuleList([])
        self.layers_norm = nn.ModuleList([])
        self.layers_MLP = nn.ModuleList([])
        self.layers_MLP.append(nn.Linear(self.num_feats, self.dim_hidden))
        if not self.has_residual_MLP:
            self.layers_GCN.append(GCNConv(self.num_feats, self.dim_hidden, cached=self.cached, args=self.args, whetherHasSE=se[0]))
        for i in range(self.num_layers):
            if self.has_residual_MLP or 0 < i < self.num_layers - 1:
                # hidden -> hidden layers all take the "middle" SE flag (GCN.py:50,52)
                self.layers_GCN.append(GCNConv(self.dim_hidden, self.dim_hidden, cached=self.cached, args=self.args, whetherHasSE=se[1]))
            appendNormLayer(self, args, self.dim_hidden if i < self.num_layers - 1 else self.num_classes)
            if AcontainsB(self.type_trick, ['Residual']):
                self.layers_res.append(ResidualConnection(alpha=self.alpha))
            elif AcontainsB(self.type_trick, ['Initial']):
                self.layers_res.append(InitialConnection(alpha=self.alpha))
            elif AcontainsB(self.type_trick, ['Dense']):
                if self.layer_agg in ['concat', 'maxpool']:
                    self.layers_res.append(DenseConnection((i + 2) * self.dim_hidden, self.dim_hidden, self.layer_agg))
                elif self.layer_agg == 'attention':
                    self.layers_res.append(DenseConnection(self.dim_hidden, self.dim_hidden, self.layer_agg))
        self.graph_dropout = DropoutTrick(args)
        if not self.has_residual_MLP:
            self.layers_GCN.append(GCNConv(self.dim_hidden, self.num_classes, cached=self.cached, args=self.args, whetherHasSE=se[2]))
        if AcontainsB(self.type_trick, ['Jumping']):
            if self.layer_agg in ['concat', 'maxpool']:
                self.layers_res.append(DenseConnection((self.num_layers + 1) * self.dim_hidden, self.num_classes, self.layer_agg))
            elif self.layer_agg == 'attention':
                self.layers_res.append(DenseConnection(self.dim_hidden, self.num_classes, self.layer_agg))
        else:
            self.layers_MLP.append(nn.Linear(self.dim_hidden, self.num_classes))
        if AcontainsB(self.type_trick, ['IdentityMapping']):
            self.lamda = args.lamda
        elif self.type_model == 'SGC':
            self.lamda = 0.
        elif self.type_model == 'GCN':
            self.lamda = 1.

    def _graph(self, edge_index):
        """First call builds the device CSR and caches it forever; later edge_index arguments are
        ignored exactly as in the reference (GCN.py:92-95)."""
        if self.dglgraph is None:
            self.dglgraph = CSRGraph(edge_index)
        return self.dglgraph

    use_fused_trunk = True   # 'Initial' connection without a bare norm runs as one fused autograd node (trunk.py)

    def forward(self, x, edge_index, want_les=False):
        graph = self._graph(edge_index)
        x_list, le_collection, se_reg_all = [], [], None
        new_adjs = self.graph_dropout(edge_index)      # computed and discarded, as in the reference (GCN.py:101,111)
        if self.use_fused_trunk and trunk.eligible(self, x, want_les):
            return trunk.forward(self, x, graph)
        row0 = getattr(graph, 'row_offset', 0)     # first global row of this rank's shard (0 on one GPU)
        if self.has_residual_MLP:
            x = ops.dropout(x, self.embedding_dropout, self.training, offset=row0 * x.shape[1])
            x = gemm.linear(x, self.layers_MLP[0].weight, self.layers_MLP[0].bias, relu=True)   # Linear + ReLU, GCN.py:105-106
            x_list.append(x)
        norms_run = self.args.type_trick in ('BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm')
        for i in range(self.num_layers):
            x = ops.dropout(x, self.dropout, self.training, offset=row0 * x.shape[1])
            _unused_edge_index, _ = new_adjs[i]
            act = self.has_residual_MLP or i < self.num_layers - 1
            # the ReLU of GCN.py:127-128 rides in the aggregation epilogue when nothing sits in between
            fuse_relu = act and not norms_run and not want_les
            x, se_reg = self.layers_GCN[i](graph, x, _fused_relu=fuse_relu)
            if se_reg is not None:
                # intended semantics of GCN.py:116-120 (sum of per-layer norms); the reference's in-place
                # `+=` on a tensor autograd saved breaks backward for >= 2 SE layers
                se_reg_all = se_reg if se_reg_all is None else se_reg_all + se_reg
            x = run_norm_if_any(self, x, i)
            if want_les:
                le_collection.append(x.clone().detach())
            if act and not fuse_relu:
                x = F.relu(x)
            x_list.append(x)
            if AcontainsB(self.type_trick, ['Initial', 'Dense', 'Residual']):
                x = self.layers_res[i](x_list)
        x = ops.dropout(x, self.args.dropout, self.training, offset=row0 * x.shape[1])   # on the logits in non-residual mode (GCN.py:133)
        if self.has_residual_MLP:
            if AcontainsB(self.type_trick, ['Jumping']):
                x = self.layers_res[0](x_list)
            else:
                x = gemm.linear(x, self.layers_MLP[-1].weight, self.layers_MLP[-1].bias)
        if want_les:
            return x, se_reg_all, th.cat(le_collection, dim=-1)
        return x, se_reg_all

    def get_se_dim(self, x, edge_index):
        return self.forward(x, edge_index, want_les=1)[2].shape[-1]

    def collect_SE(self, x, edge_index):
        return self.forward(x, edge_index, want_les=1)[2]


class GCNConv(nn.Module):
    """Cold Brew's GraphConv: W first, + structural embedding `le`, sum-aggregate, symmetric
    degree normalisation, + bias; returns (rst, se_reg) (reference GCN.py:152-258)."""

    def __init__(self, in_feats, out_feats, norm='both', weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False, cached=None, args=None, whetherHasSE=False):
        super().__init__()
        self.args = args
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        if weight:
            self.weight = nn.Parameter(th.Tensor(in_feats, out_feats))
        else:
            self.register_parameter('weight', None)
        if bias:
            self.bias = nn.Parameter(th.Tensor(out_feats))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        self._activation = activation
        self.whetherHasSE = whetherHasSE
        if whetherHasSE:
            self.le = nn.Parameter(th.randn(args.N_nodes, self._out_feats), requires_grad=True)

    def forward(self, graph, feat, weight=None, edge_weight=None, _fused_relu=False):
        if not self._allow_zero_in_degree:
            graph.check_zero_in_degree()                       # GCN.py:187-197
        if edge_weight is not None:
            raise NotImplementedError('edge_weight (u_mul_e) is never passed by TricksComb (GCN.py:115,199-202)')
        if self._norm != 'both':
            raise NotImplementedError("only norm='both' is reachable from TricksComb")
        if weight is not None:
            if self.weight is not None:
                raise DGLError('External weight is provided while at the same time the module has defined its own '
                               'weight parameter. Please create the module with flag weight=False.')
        else:
            weight = self.weight
        if weight is None:
            raise NotImplementedError('GCNConv without a weight is not reachable from TricksComb')
        le = self.le if self.whetherHasSE else None
        h, se_reg = ops.transform(feat, graph.norm_out, weight, le, graph)          # GCN.py:213,225,230-236
        rst = ops.aggregate(graph, h, row_scale=graph.norm_in, bias=self.bias,      # GCN.py:238,250,253
                            relu=_fused_relu)
        if self._activation is not None:
            rst = self._activation(rst)
        return rst, se_reg

    def reset_parameters(self):
        if self.weight is not None:
            init.xavier_uniform_(self.weight)
        if self.bias is not None:
            init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def extra_repr(self):
        return f'in={self._in_feats}, out={self._out_feats}, normalization={self._norm}'


def tonp(arr):
    import numpy as np
    if type(arr) is th.Tensor:
        return arr.detach().cpu().data.numpy()
    return np.asarray(arr)
