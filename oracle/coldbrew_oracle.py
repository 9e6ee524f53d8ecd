"""TEST INFRASTRUCTURE ONLY — CPU oracle for Cold Brew's TeacherGNN hot path.

A plain torch/numpy restatement (CPU, fp32 or fp64) of the reference algorithm, each
function citing the reference file:line it follows (paths relative to /root/reference).
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product package (gnn-tail-generalization_amd/) never does and fails loudly
without its HIP extension.

Pinning status
  * Everything except the aggregation is pinned by importing the *unmodified* reference
    modules in the build container (oracle/ref_import.py) and comparing against them /
    against the golden vectors they produced (tests/golden/, tests/test_oracle_golden.py).
  * The aggregation `graph.update_all(copy_src, sum)` (GNN_model/GCN.py:198,238) is
    arithmetic of the third-party dgl==0.7.0 (requirements.txt:19), absent from the tree
    and from this image: PARITY UNPINNED at that boundary.  It is restated from DGL's
    published semantics (out[dst] = sum over edges of h[src], duplicates counted) and
    cross-checked against a dense fp64 A^T.h.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Graph: edge_index -> CSR (integer work, bit-exact contract; GNN_model/GCN.py:92-95)
# --------------------------------------------------------------------------------------
def build_csr(edge_index, num_nodes=None):
    """src = edge_index[0], dst = edge_index[1] (GCN.py:93-94); N = max id + 1 as DGL infers
    it unless given.  Returns the by-dst CSR (rows = dst, cols = src: the forward
    aggregation) and the by-src CSR (rows = src, cols = dst: the reverse graph used by
    the backward).  Canonical in-row order: ascending column id (duplicates adjacent)."""
    ei = edge_index.detach().cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    ei = ei.astype(np.int64, copy=False)
    src, dst = ei[0], ei[1]
    if ei.shape[1] and (src.min() < 0 or dst.min() < 0):
        raise ValueError('negative node id in edge_index')
    n = int(max(src.max(), dst.max())) + 1 if ei.shape[1] else 0
    if num_nodes is not None:
        if n > num_nodes:
            raise ValueError('node id >= num_nodes in edge_index')
        n = int(num_nodes)
    order = np.lexsort((src, dst))              # by dst, then src
    col = src[order].astype(np.int32)
    in_deg = np.bincount(dst, minlength=n).astype(np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(in_deg, out=rowptr[1:])
    order_t = np.lexsort((dst, src))            # by src, then dst
    col_t = dst[order_t].astype(np.int32)
    out_deg = np.bincount(src, minlength=n).astype(np.int64)
    rowptr_t = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(out_deg, out=rowptr_t[1:])
    return SimpleNamespace(N=n, E=int(ei.shape[1]), rowptr=rowptr, col=col, rowptr_t=rowptr_t, col_t=col_t,
                           in_deg=in_deg, out_deg=out_deg, src=src, dst=dst)


def degree_norms(csr, dtype=torch.float32):
    """a = clamp(out_deg,1)^-0.5 (GCN.py:206-208), b = clamp(in_deg,1)^-0.5 (GCN.py:243-245)."""
    a = torch.pow(torch.from_numpy(csr.out_deg).to(dtype).clamp(min=1), -0.5)
    b = torch.pow(torch.from_numpy(csr.in_deg).to(dtype).clamp(min=1), -0.5)
    return a, b


def check_no_zero_in_degree(csr):
    """GCN.py:187-197: a 0-in-degree node raises (allow_zero_in_degree is forced False, :167-168)."""
    if csr.N and (csr.in_deg == 0).any():
        raise ZeroInDegreeError('There are 0-in-degree nodes in the graph')


class ZeroInDegreeError(RuntimeError):
    pass


def aggregate_sum(csr, h):
    """rst[v,:] = sum_{e:(u->v)} h[u,:]  (GCN.py:198,238; DGL copy_u + sum).  Differentiable."""
    src = torch.from_numpy(csr.src)
    dst = torch.from_numpy(csr.dst)
    out = torch.zeros((csr.N,) + tuple(h.shape[1:]), dtype=h.dtype)
    return out.index_add(0, dst, h.index_select(0, src))


_AGGREGATE = [aggregate_sum]


def set_aggregate(fn=None):
    """bench.py's cpu_baseline leg swaps in the C/OpenMP SpMM (oracle/oracle_c.py) for graphs where the
    index_add message tensor [E, d] would not fit in host memory; None restores the default."""
    _AGGREGATE[0] = fn or aggregate_sum


def aggregate_sum_dense_f64(csr, h):
    """Independent check of aggregate_sum: dense A^T.h in fp64 (small graphs only)."""
    A = np.zeros((csr.N, csr.N), dtype=np.float64)
    np.add.at(A, (csr.src, csr.dst), 1.0)
    return torch.from_numpy(A.T @ h.detach().double().numpy())


# --------------------------------------------------------------------------------------
# GCNConv (GNN_model/GCN.py:184-258)
# --------------------------------------------------------------------------------------
class _QuantGradBf16(torch.autograd.Function):
    """Identity whose backward rounds the gradient to bf16 (storage of b*dY' in the bf16 aggregation variant)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def aggregate_sum_weighted(csr, h, edge_weight):
    """rst[v,:] = sum_{e:(u->v)} w_e * h[u,:]  (GCN.py:199-202; DGL u_mul_e + sum; w indexed like edge_index's columns)."""
    src = torch.from_numpy(csr.src)
    dst = torch.from_numpy(csr.dst)
    out = torch.zeros((csr.N,) + tuple(h.shape[1:]), dtype=h.dtype)
    return out.index_add(0, dst, h.index_select(0, src) * edge_weight.reshape(-1, 1))


def gcnconv_forward(csr, feat, weight, bias=None, le=None, a=None, b=None, quant_bf16=False, edge_weight=None):
    """quant_bf16 (build extension, not in the reference): the rows the aggregation gathers — Z in the forward,
    the scaled gradient in the backward — are rounded to bf16 (RNE), everything else stays fp32."""
    check_no_zero_in_degree(csr)
    if a is None or b is None:
        a, b = degree_norms(csr, feat.dtype)
    feat_src = feat * a.reshape(-1, 1)                       # :213
    feat_src = torch.matmul(feat_src, weight)                # :225 (always W first)
    if le is not None:                                       # :230-232
        h = feat_src + le
        se_reg = torch.norm(le)
    else:
        h = feat_src
        se_reg = None
    if quant_bf16:
        h = h + (h.detach().to(torch.bfloat16).to(h.dtype) - h.detach())    # value rounded, gradient passes straight through
    if edge_weight is not None:
        assert edge_weight.shape[0] == csr.E                 # :200
        rst = aggregate_sum_weighted(csr, h, edge_weight)    # :199-202,238
    else:
        rst = _AGGREGATE[0](csr, h)                          # :238
    if quant_bf16:
        rst = _QuantGradBf16.apply(rst)
    rst = rst * b.reshape(-1, 1)                             # :250
    if bias is not None:
        rst = rst + bias                                     # :253
    return rst, se_reg


# --------------------------------------------------------------------------------------
# Norm tricks (GNN_model/norm_tricks.py)
# --------------------------------------------------------------------------------------
def acontainsb(a, list_b):                                   # norm_tricks.py:123-127
    return any(s in a for s in list_b)


BARE_NORMS = ['BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm']   # norm_tricks.py:147


def norm_kind(type_trick):
    """Which norm layer appendNormLayer builds (substring match, norm_tricks.py:131-143)."""
    for k in ['BatchNorm', 'PairNorm', 'NodeNorm', 'MeanNorm', 'GroupNorm', 'CombNorm']:
        if k in type_trick:
            return k
    return None


def pair_norm(x):                                            # norm_tricks.py:25-30
    x = x - x.mean(dim=0)
    return x / (1e-6 + x.pow(2).sum(dim=1).mean()).sqrt()


def mean_norm(x):                                            # norm_tricks.py:38-41
    return x - x.mean(dim=0)


def node_norm(x, kind='n', eps=1e-5, power_root=2):          # norm_tricks.py:53-84
    if kind == 'm':
        return x - x.mean(dim=1, keepdim=True)
    std = (torch.var(x, unbiased=False, dim=1, keepdim=True) + eps).sqrt()
    if kind == 'n':
        return (x - x.mean(dim=1, keepdim=True)) / std
    if kind == 'v':
        return x / std
    if kind == 'srv':
        return x / torch.sqrt(std)
    if kind == 'pr':
        return x / torch.pow(std, 1.0 / power_root)
    return x                                                 # unknown type: identity (no branch taken)


def batch_norm(x, p, prefix, training, momentum=0.1, buffers_out=None):
    """torch.nn.BatchNorm1d (norm_tricks.py:132 / :106).  `p` holds weight,bias,running_*."""
    rm = p[prefix + 'running_mean'].clone()
    rv = p[prefix + 'running_var'].clone()
    y = F.batch_norm(x, rm, rv, p[prefix + 'weight'], p[prefix + 'bias'], training, momentum, 1e-5)
    if buffers_out is not None and training:
        buffers_out[prefix + 'running_mean'] = rm
        buffers_out[prefix + 'running_var'] = rv
    return y


def group_norm(x, p, prefix, num_groups, skip_weight, training, buffers_out=None):   # norm_tricks.py:110-120
    d = x.shape[1]
    if num_groups == 1:
        x_temp = batch_norm(x, p, prefix + 'bn.', training, 0.3, buffers_out)
    else:
        score = F.softmax(F.linear(x, p[prefix + 'group_func.weight'], p[prefix + 'group_func.bias']), dim=1)
        x_temp = torch.cat([score[:, g].unsqueeze(1) * x for g in range(num_groups)], dim=1)
        x_temp = batch_norm(x_temp, p, prefix + 'bn.', training, 0.3, buffers_out).view(-1, num_groups, d).sum(dim=1)
    return x + x_temp * skip_weight


def run_norm(cfg, p, x, i, training, buffers_out=None):
    """run_norm_if_any: runs only for an exact bare name (norm_tricks.py:146-150)."""
    if cfg.type_trick not in BARE_NORMS:
        return x
    pre = f'layers_norm.{i}.'
    k = cfg.type_trick
    if k == 'BatchNorm':
        return batch_norm(x, p, pre, training, 0.1, buffers_out)
    if k == 'PairNorm':
        return pair_norm(x)
    if k == 'MeanNorm':
        return mean_norm(x)
    if k == 'NodeNorm':
        return node_norm(x, cfg.node_norm_type)
    if k == 'GroupNorm':
        return group_norm(x, p, pre, cfg.num_groups, cfg.skip_weight, training, buffers_out)
    if k == 'CombNorm':                                      # norm_tricks.py:141-143
        x = group_norm(x, p, pre + 'norm_list.0.', cfg.num_groups, cfg.skip_weight, training, buffers_out)
        return node_norm(x, cfg.node_norm_type)
    raise AssertionError(k)


# --------------------------------------------------------------------------------------
# Res tricks (GNN_model/res_tricks.py)
# --------------------------------------------------------------------------------------
def dense_connection(xs, p, prefix, aggregation):            # res_tricks.py:36-53
    if aggregation == 'concat':
        return F.linear(torch.cat(xs, dim=-1), p[prefix + 'layer_transform.weight'], p[prefix + 'layer_transform.bias'])
    if aggregation == 'maxpool':
        return torch.stack(xs, dim=-1).max(dim=-1)[0]
    if aggregation == 'attention':
        pps = torch.stack(xs, dim=1)
        score = F.linear(pps, p[prefix + 'layer_att.weight'], p[prefix + 'layer_att.bias']).squeeze()
        score = torch.sigmoid(score).unsqueeze(1)
        return torch.matmul(score, pps).squeeze()
    raise Exception('Unknown aggregation')


def res_mix(cfg, p, xs, i):
    """layers_res[i](x_list) (GCN.py:130-131).  Construction precedence Residual > Initial >
    Dense (GCN.py:57-67)."""
    t = cfg.type_trick
    if 'Residual' in t:                                      # res_tricks.py:12-14
        return xs[-1] if len(xs) == 1 else (1 - cfg.res_alpha) * xs[-1] + cfg.res_alpha * xs[-2]
    if 'Initial' in t:                                       # res_tricks.py:21-23
        return xs[-1] if len(xs) == 1 else (1 - cfg.res_alpha) * xs[-1] + cfg.res_alpha * xs[0]
    if 'Dense' in t:
        return dense_connection(xs, p, f'layers_res.{i}.', cfg.layer_agg)
    raise AssertionError(t)


# --------------------------------------------------------------------------------------
# TricksComb.forward (GNN_model/GCN.py:91-142)
# --------------------------------------------------------------------------------------
def make_cfg(**kw):
    d = dict(type_trick='NoResNoNorm', num_layers=2, num_feats=None, dim_hidden=64, num_classes=None,
             dropout=0.0, res_alpha=0.1, layer_agg='concat', whetherHasSE=(0, 0, 0), node_norm_type='n',
             num_groups=None, skip_weight=None, se_reg=0.0, change_to_featureless=0, dim_learnable_input=0, quant_bf16=False)
    d.update(kw)
    return SimpleNamespace(**d)


def has_residual_mlp(cfg):                                   # GCN.py:34-36
    return acontainsb(cfg.type_trick, ['Jumping', 'Initial', 'Residual', 'Dense'])


def strip_prefix(state_dict, prefix='model.model.'):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


class _Dropper:
    """Dropout with injected keep-masks (already scaled semantics: y = x * mask / (1-p)).
    In eval mode or p == 0 it is the identity, like F.dropout."""

    def __init__(self, training, masks):
        self.training = training
        self.masks = list(masks) if masks is not None else None
        self.k = 0

    def __call__(self, x, p):
        if not self.training or p == 0.0:
            return x
        if self.masks is None:
            raise ValueError('oracle train-mode dropout with p>0 needs injected masks')
        m = self.masks[self.k]
        self.k += 1
        return x * m.to(x.dtype) / (1.0 - p)


def trickscomb_forward(cfg, p, x, csr, training=False, dropout_masks=None, want_les=False, buffers_out=None):
    """p: parameters keyed relative to TricksComb ('layers_GCN.0.weight', ...).
    Returns (x, se_reg_all[, les]).  se_reg_all is the *intended* sum of per-layer
    Frobenius norms (GCN.py:116-120; the in-place `+=` there is a latent autograd bug)."""
    drop = _Dropper(training, dropout_masks)
    a, b = degree_norms(csr, x.dtype)
    xs, les, se_reg_all = [], [], None
    residual = has_residual_mlp(cfg)
    if residual:                                             # GCN.py:103-107
        x = drop(x, cfg.dropout)
        x = F.linear(x, p['layers_MLP.0.weight'], p['layers_MLP.0.bias'])
        x = F.relu(x)
        xs.append(x)
    for i in range(cfg.num_layers):                          # GCN.py:109-131
        x = drop(x, cfg.dropout)
        x, se_reg = gcnconv_forward(csr, x, p[f'layers_GCN.{i}.weight'], p[f'layers_GCN.{i}.bias'],
                                    p.get(f'layers_GCN.{i}.le'), a, b, getattr(cfg, 'quant_bf16', False))
        if se_reg is not None:
            se_reg_all = se_reg if se_reg_all is None else se_reg_all + se_reg
        x = run_norm(cfg, p, x, i, training, buffers_out)
        if want_les:
            les.append(x.clone().detach())
        if residual or i < cfg.num_layers - 1:
            x = F.relu(x)
        xs.append(x)
        if acontainsb(cfg.type_trick, ['Initial', 'Dense', 'Residual']):
            x = res_mix(cfg, p, xs, i)
    x = drop(x, cfg.dropout)                                 # GCN.py:133
    if residual:                                             # GCN.py:134-138
        if 'Jumping' in cfg.type_trick:
            # GCN.py:136 indexes layers_res[0]; with Jumping alone that is the single Dense head
            x = dense_connection(xs, p, 'layers_res.0.', cfg.layer_agg)
        else:
            x = F.linear(x, p['layers_MLP.1.weight'], p['layers_MLP.1.bias'])
    if want_les:
        return x, se_reg_all, torch.cat(les, dim=-1)
    return x, se_reg_all


# --------------------------------------------------------------------------------------
# TeacherGNN (GNN_model/GNN_normalizations.py:31-55) + loss (trainer_node_classification.py:386-394)
# --------------------------------------------------------------------------------------
def teacher_forward(cfg, sd, x, csr, training=False, dropout_masks=None, buffers_out=None):
    """sd: full TeacherGNN state_dict ('model.model.*', optional 'embs').  proj2class = Identity."""
    if cfg.change_to_featureless:
        x = x * 0                                            # GNN_normalizations.py:32-33
    if cfg.dim_learnable_input > 0:
        x = sd['embs']                                       # :34-35
    return trickscomb_forward(cfg, strip_prefix(sd), x, csr, training, dropout_masks, False, buffers_out)


def proj2class_head(sd, common, training=False):
    """TeacherGNN.proj2class with --has_proj2class=1 (GNN_normalizations.py:13,28,42; built by utils.getMLP :885-908 from
    neurons_proj2class = [dim_commonEmb = 128, 20, C], utils.py:613-624): Linear -> LayerNorm -> GELU -> Dropout(0.1) -> Linear.
    Eval-mode restatement (the head's dropout draws from torch's CPU stream in train mode)."""
    if training:
        raise NotImplementedError('the oracle restates the proj2class head in eval mode only')
    h = F.linear(common, sd['proj2class.0.weight'], sd['proj2class.0.bias'])
    h = F.layer_norm(h, (h.shape[1],), sd['proj2class.1.weight'], sd['proj2class.1.bias'], 1e-5)
    h = F.gelu(h)
    return F.linear(h, sd['proj2class.4.weight'], sd['proj2class.4.bias'])


def training_loss(cfg, out, se_reg_all, y, train_mask):
    """nll_loss(log_softmax(out[mask])) + se_reg * se_reg_all (trainer_node_classification.py:390-394)."""
    logits = F.log_softmax(out[train_mask], 1)
    loss = F.nll_loss(logits, y[train_mask])
    if se_reg_all is not None:
        loss = loss + cfg.se_reg * se_reg_all
    return loss


def evaluate(output, labels, mask):                          # trainer_node_classification.py:672-681
    idx = output.max(dim=1)[1]
    if mask is None:
        return (idx == labels).sum().item() / len(idx)
    return (idx[mask] == labels[mask]).sum().item() * 1.0 / mask.sum().item()


def cal_acc_rounded100(output, labels):                      # trainer_node_classification.py:683-687 + utils.py:950-956
    idx = output.max(dim=1)[1]
    correct = (idx == labels).sum() / len(labels)
    return np.round((correct * 100).detach().cpu().numpy().reshape(-1)[0], 3)


def adam_step(params, grads, state, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (trainer_node_classification.py:310) restated: L2-style weight decay
    added to the gradient, bias-corrected moments.  params/grads/state are dicts of tensors."""
    b1, b2 = betas
    for k, w in params.items():
        g = grads.get(k)
        if g is None:
            continue
        if weight_decay != 0:
            g = g + weight_decay * w
        m = state.setdefault(k + '.m', torch.zeros_like(w))
        v = state.setdefault(k + '.v', torch.zeros_like(w))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)


def train_steps(cfg, sd, x, csr, y, train_mask, steps, lr, weight_decay, dropout_masks_per_step=None):
    """K optimisation steps of run_trainSet (trainer_node_classification.py:382-432, without the
    head/tail extra forward).  Returns per-step losses; sd tensors are updated in place."""
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k]
    for k in names:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    state, losses = {}, []
    for s in range(steps):
        masks = None if dropout_masks_per_step is None else dropout_masks_per_step[s]
        out, reg = teacher_forward(cfg, sd, x, csr, training=True, dropout_masks=masks)
        loss = training_loss(cfg, out, reg, y, train_mask)
        used = [k for k in names]
        grads = torch.autograd.grad(loss, [sd[k] for k in used], allow_unused=True)
        with torch.no_grad():
            adam_step({k: sd[k] for k in used}, {k: g for k, g in zip(used, grads)}, state, lr, weight_decay, s + 1)
        losses.append(float(loss.detach()))
    return losses


# --------------------------------------------------------------------------------------
# Pure label propagation: trainer_node_classification.run_pureLP :33-63 over
# Label_propagation_model/outcome_correlation.py — process_adj :39-49, gen_normalized_adjs :51-55 (DAD),
# label_propagation :147-156, general_outcome_correlation :128-145.
# The sparse algebra of the reference is torch_sparse==0.6.10 (absent): PARITY UNPINNED at that boundary,
# pinned against the reference functions run over the SparseTensor stand-in (tests/golden/lp_fixture.pt).
# --------------------------------------------------------------------------------------
def to_undirected(edge_index, num_nodes):
    """PyG to_undirected: union with the transpose, duplicates removed, sorted by (row, col) (:41)."""
    ei = edge_index.to(torch.int64)
    key = torch.unique(torch.cat([ei[0] * num_nodes + ei[1], ei[1] * num_nodes + ei[0]]))
    return torch.stack([key // num_nodes, key % num_nodes])


def label_propagation(edge_index, y, train_mask, num_classes, alpha=0.5, num_propagations=50):
    n = y.shape[0]
    ei = to_undirected(edge_index, n)
    csr = build_csr(ei, n)
    deg = torch.from_numpy(csr.in_deg).to(torch.float32)           # adj.sum(dim=1) (:46)
    dis = deg.pow(-0.5)
    dis[dis == float('inf')] = 0                                  # :47-48
    y0 = torch.zeros((n, num_classes))
    y0[train_mask] = F.one_hot(y[train_mask], num_classes).float()  # :151-153
    result = y0.clone()
    for _ in range(num_propagations):                               # :137-143 with alpha_term, post_step = clamp(0, 1)
        prop = aggregate_sum(csr, result * dis.unsqueeze(1)) * dis.unsqueeze(1)    # DAD @ result
        result = torch.clamp(alpha * prop + (1 - alpha) * y0, 0, 1)
    return result, dis


# --------------------------------------------------------------------------------------
# Teacher -> student hand-off: SEMLP.replacement, MLP_model/__init__.py:143-156
# --------------------------------------------------------------------------------------
def semlp_replacement(le_guess, teacher_se, k):
    """Per query: scores against all teacher rows (:150), the K largest by ascending argsort()[-K:] (:151-152), softmax over
    them (:153), weighted sum of the selected teacher rows (:154).  Returns (out [B,D], select [B,K] ascending, weights)."""
    outs, sels, wts = [], [], []
    t_T = teacher_se.transpose(0, 1)
    for i in range(le_guess.shape[0]):
        attn = torch.matmul(le_guess[[i]], t_T)
        select = attn.argsort()[0][-k:]
        w = F.softmax(attn[:, select], dim=1)
        outs.append(torch.matmul(w, teacher_se[select]))
        sels.append(select)
        wts.append(w[0])
    return torch.cat(outs, dim=0), torch.stack(sels), torch.stack(wts)
