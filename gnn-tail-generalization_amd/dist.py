"""Node-sharded TeacherGNN over the GPUs of one node (one process per GPU, torch.distributed
`nccl` = RCCL over xGMI).  New relative to the reference, which is single-device
(SURVEY.md §8e).

Partition: 1-D contiguous row blocks of equal size R = ceil(N / P) (node ids of the synthetic
graphs are randomly permuted, so equal rows ~ equal edges).  Rank p owns rows
[p*R, min((p+1)*R, N)) of x / y / masks / activations / structural embeddings and the matching row
slices of both CSR orientations, with GLOBAL column ids.

Exchange (the only data-path collective), two interchangeable forms:
  * 'halo' (default): each rank receives only the remote rows its edges reference.  Per orientation
    a HaloPlan is built once — unique remote column ids grouped by owner, the owners are told which
    of their rows to send (one all-to-all of counts + one of ids), local column ids are remapped to
    [0, n_local) and remote ones to n_local + position in the receive buffer.  Per aggregation: pack
    (row gather) -> all_to_all_single with uneven splits (RCCL grouped send/recv over the xGMI peer
    links) -> the local rows are reduced by the same SpMM kernel on [local rows | halo rows].  On the
    10M-node power-law graph a rank needs 0.41 N remote rows at P = 8 instead of the 0.875 N an
    all-gather delivers.
  * 'allgather': all-gather of the [R, d] shards into the full [P*R, d] matrix (simple baseline).
The backward of the aggregation is the same exchange on the gradient followed by the SpMM on the
by-src slice (its own plan; no symmetry assumption).  Everything else is row-local; tiny all-reduces cover the replicated
weights' gradients, the loss numerator and sum(E^2) of the structural-embedding regulariser.

The exchange and bookkeeping are device-agnostic torch.distributed code (tested on CPU with gloo,
world_size 2); the compute stays on the HIP path.
"""
import contextlib
import io

import torch
import torch.distributed as dist

from . import _lib
from .graph import CSRGraph, ZeroInDegreeError


def _staged(t, group):
    """gloo cannot move device tensors for every collective: stage through the host in that case (used to exercise the
    whole sharded HIP path with several processes on ONE GPU; the production backend is nccl = RCCL, no staging)."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _all_reduce(t, op=dist.ReduceOp.SUM, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_to_all_single(out, inp, out_splits=None, in_splits=None, group=None):
    if _staged(inp, group) or _staged(out, group):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(ho, hi, out_splits, in_splits, group=group)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather_into_tensor(out, inp, group=None):
    if _staged(inp, group):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(ho, hi, group=group)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


class Partition:
    """Equal row blocks: rank p owns [lo(p), hi(p))."""

    def __init__(self, n_nodes, world, rank):
        self.N, self.world, self.rank = int(n_nodes), int(world), int(rank)
        self.R = (self.N + self.world - 1) // self.world

    def lo(self, p=None):
        p = self.rank if p is None else p
        return min(p * self.R, self.N)

    def hi(self, p=None):
        p = self.rank if p is None else p
        return min((p + 1) * self.R, self.N)

    @property
    def n_local(self):
        return self.hi() - self.lo()

    @property
    def padded(self):
        return self.R * self.world

    def slice_rows(self, t):
        return t[self.lo():self.hi()]

    def owner(self, node_ids):
        return torch.div(node_ids, self.R, rounding_mode='floor')


def gather_rows(x_local, part, group=None, out=None):
    """All-gather of the row shards: [n_local, d] on each rank -> [P*R, d] (rows >= N are padding)."""
    d = x_local.shape[1]
    if out is None:
        out = torch.empty((part.padded, d), dtype=x_local.dtype, device=x_local.device)
    if x_local.shape[0] == part.R and x_local.is_contiguous():
        src = x_local
    else:   # last rank: pad its shard to R rows
        src = torch.zeros((part.R, d), dtype=x_local.dtype, device=x_local.device)
        src[:x_local.shape[0]] = x_local
    _all_gather_into_tensor(out, src, group=group)
    return out


class HaloPlan:
    """Who sends which rows to whom for one CSR orientation (built once per graph)."""

    def __init__(self, col_global, part, group=None):
        lo, hi, P, R = part.lo(), part.hi(), part.world, part.R
        col = col_global.to(torch.int64)
        dev = col.device
        self.n_local = hi - lo
        remote = (col < lo) | (col >= hi)
        uniq, inv = torch.unique(col[remote], return_inverse=True)      # ascending ids = grouped by owner
        self.n_halo = int(uniq.numel())
        new_col = col - lo
        new_col[remote] = self.n_local + inv
        self.col = new_col.to(torch.int32)
        recv_counts = torch.bincount(torch.div(uniq, R, rounding_mode='floor'), minlength=P)[:P]
        send_counts = torch.empty_like(recv_counts)
        if P > 1:
            _all_to_all_single(send_counts, recv_counts, group=group)    # how many rows each peer wants from me
        else:
            send_counts.copy_(recv_counts)
        self.recv_counts = [int(v) for v in recv_counts.tolist()]
        self.send_counts = [int(v) for v in send_counts.tolist()]
        wanted = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
        if P > 1:
            _all_to_all_single(wanted, uniq, self.send_counts, self.recv_counts, group=group)
        self.send_idx = (wanted - lo).contiguous()                         # my local rows, in per-destination order
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_local):
            raise RuntimeError('halo plan: a peer requested a row this rank does not own')


def _pack_rows(x, idx):
    if x.is_cuda and x.dtype == torch.float32:
        from .ops import gather_rows_by_index
        return gather_rows_by_index(x, idx)
    if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 2 == 0 and x.is_contiguous():
        from .ops import gather_rows_by_index      # bf16 rows move as packed 32-bit words
        return gather_rows_by_index(x.view(torch.float32), idx).view(torch.bfloat16)
    return x.index_select(0, idx)


class ShardedGraph:
    """Row slice [lo, hi) of both CSR orientations + local degree norms + the exchange plans.
    Quacks like graph.CSRGraph for GCNConv / ops.aggregate."""

    def __init__(self, full, part, group=None, spmm_fn=None, exchange='halo'):
        """full: an object with rowptr/col/rowptr_t/col_t/norm_in/norm_out/N/E (a CSRGraph built from
        the whole edge_index, or the numpy oracle CSR in the CPU tests)."""
        self.part, self.group = part, group
        self.exchange_kind = exchange
        self.N_global, self.E_global = full.N, full.E
        self.n_zero_in_degree = getattr(full, 'n_zero_in_degree', 0)
        self.row_offset = part.lo()
        lo, hi = part.lo(), part.hi()
        self.N = hi - lo
        self._spmm_fn = spmm_fn
        self.profile = None

        def cut(rowptr, col):
            rp = torch.as_tensor(rowptr)
            e0, e1 = int(rp[lo]), int(rp[hi])
            loc_rp = (rp[lo:hi + 1] - rp[lo]).clone()
            loc_col = torch.as_tensor(col)[e0:e1].clone()
            return loc_rp, loc_col

        rp, c = cut(full.rowptr, full.col)
        rpt, ct = cut(full.rowptr_t, full.col_t)
        self.E = int(c.numel())
        self.norm_in = torch.as_tensor(full.norm_in)[lo:hi].clone()
        self.norm_out = torch.as_tensor(full.norm_out)[lo:hi].clone()
        if exchange == 'halo':
            self.plan_fwd = HaloPlan(c, part, group)
            same = (rpt.shape == rp.shape and ct.shape == c.shape and bool(torch.equal(rpt, rp)) and bool(torch.equal(ct, c)))
            if part.world > 1:      # one global decision, or the ranks would disagree on which collectives follow
                flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=rp.device)
                _all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                same = bool(flag.item())
            self.plan_bwd = self.plan_fwd if same else HaloPlan(ct, part, group)
            c, ct = self.plan_fwd.col, self.plan_bwd.col
            ncols_f, ncols_b = self.N + self.plan_fwd.n_halo, self.N + self.plan_bwd.n_halo
        elif exchange == 'allgather':
            self.plan_fwd = self.plan_bwd = None
            ncols_f = ncols_b = part.padded
        else:
            raise ValueError(f'unknown exchange {exchange!r}')
        if spmm_fn is None:     # HIP path: wrap the slices as rectangular device CSRs
            self.fwd = CSRGraph.from_csr(rp, c, n_cols=ncols_f)
            self.bwd = CSRGraph.from_csr(rpt, ct, n_cols=ncols_b)
        else:
            self.fwd, self.bwd = (rp, c), (rpt, ct)

    def exchange(self, x_local, transpose=False):
        """[n_local, d] -> the matrix the local SpMM reads: [n_local + n_halo, d] (halo) or [P*R, d] (allgather)."""
        if self.exchange_kind == 'allgather':
            return gather_rows(x_local, self.part, self.group)
        plan = self.plan_bwd if transpose else self.plan_fwd
        if self.part.world == 1:
            return x_local
        d = x_local.shape[1]
        ext = torch.empty((plan.n_local + plan.n_halo, d), dtype=x_local.dtype, device=x_local.device)
        ext[:plan.n_local] = x_local
        send = _pack_rows(x_local, plan.send_idx)
        _all_to_all_single(ext[plan.n_local:], send, plan.recv_counts, plan.send_counts, group=self.group)
        return ext

    def check_zero_in_degree(self):
        if self.n_zero_in_degree:
            raise ZeroInDegreeError('There are 0-in-degree nodes in the graph')

    def number_of_nodes(self):
        return self.N_global

    def number_of_edges(self):
        return self.E_global

    def local_spmm(self, h_full, transpose=False, row_scale=None, bias=None, relu=False):
        g = self.bwd if transpose else self.fwd
        if self._spmm_fn is not None:
            return self._spmm_fn(g[0], g[1], h_full, row_scale, bias, relu)
        g.profile = self.profile
        return g.spmm(h_full, row_scale=row_scale, bias=bias, relu=relu)

    def algorithmic_bytes(self, d, **kw):
        return self.fwd.algorithmic_bytes(d, **kw)


class _ShardedAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, h_local, row_scale, bias, relu):
        h_full = graph.exchange(h_local, False)
        out = graph.local_spmm(h_full, False, row_scale, bias, relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, relu, bias is not None
        ctx.save_for_backward(out if relu else None, row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        out, row_scale = ctx.saved_tensors
        graph = ctx.graph
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        gs, dbias = _act_bwd(g, out if ctx.relu else None, row_scale, need_b)
        dh = None
        if ctx.needs_input_grad[1]:
            g_full = graph.exchange(gs, True)
            dh = graph.local_spmm(g_full, True)
        return None, dh, None, dbias, None


def _act_bwd(g, act, row_scale, need_b):
    if g.is_cuda:
        from .ops import act_bwd
        return act_bwd(g, act, row_scale, want_out=True, want_colsum=need_b)
    gm = g * (act > 0) if act is not None else g           # CPU tests of the exchange logic only
    return (gm * row_scale.unsqueeze(1) if row_scale is not None else gm), (gm.sum(0) if need_b else None)


def sharded_aggregate(graph, h_local, row_scale=None, bias=None, relu=False):
    return _ShardedAggregateFn.apply(graph, h_local, row_scale, bias, bool(relu))


def allreduce_grads(params, group=None):
    """Sum the gradients of the replicated parameters in one flat bucket (weights are a few hundred KB)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    _all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class _AllReduceSumFn(torch.autograd.Function):
    """y = sum over ranks of x; backward = all-reduce of the upstream gradients (each rank holds only its
    share of the loss, and all shares depend on x through y)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        _all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()                      # every rank's loss depends on x through y: sum the upstream gradients
        _all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def allreduce_sum(x, group=None):
    return _AllReduceSumFn.apply(x, group)


class ShardedTrainer:
    """bench.py / multi-GPU driver of the TeacherGNN step.  Mirrors trainer.train_step()."""

    def __init__(self, args, which_run, group=None):
        from . import optim as cb_optim
        from .data import synthetic_data
        from .utils import set_arch_configs
        self.args, self.group = args, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(f'cuda:{args.cuda_num}')
        args.device = self.device
        # every rank generates the same seeded graph, keeps its row block and drops the rest
        data = synthetic_data(args.dataset, seed=0, device=self.device)
        self._n, self._e = int(data.x.shape[0]), int(data.edge_index.shape[1])
        # all ranks must hold the same graph: compare a checksum before slicing
        chk = torch.stack([data.edge_index.sum(), (data.edge_index[0] * 31 + data.edge_index[1]).sum(),
                           data.train_mask.sum().to(torch.int64)]).to(torch.float64)
        lo_, hi_ = chk.clone(), chk.clone()
        _all_reduce(lo_, op=dist.ReduceOp.MIN, group=group)
        _all_reduce(hi_, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(lo_, hi_):
            raise RuntimeError('ranks generated different graphs (seeded generator mismatch)')
        self.part = Partition(self._n, self.world, self.rank)
        full = CSRGraph(data.edge_index, self._n)
        import os
        self.sgraph = ShardedGraph(full, self.part, group, exchange=os.environ.get('COLDBREW_EXCHANGE', 'halo'))
        self.n_train = int(data.train_mask.sum().item())
        p = self.part
        self.x = p.slice_rows(data.x).float().contiguous()
        self.y = p.slice_rows(data.y).contiguous()
        self.train_mask = p.slice_rows(data.train_mask).contiguous()
        self.edge_index = data.edge_index[:, :1]     # placeholder: the cached sharded graph is injected below
        del data, full
        torch.cuda.empty_cache()
        self.optfun = cb_optim.resolve(args.optfun)
        set_arch_configs(args)
        args.N_nodes_global = self._n
        args.N_nodes = self.part.n_local             # structural-embedding tables are row-sharded

    def setup_teacherGNN(self):
        from .GNN_model.GNN_normalizations import TeacherGNN
        if self.args.type_trick in ('BatchNorm', 'PairNorm', 'MeanNorm', 'GroupNorm', 'CombNorm'):
            raise NotImplementedError('column-statistic norm tricks need a cross-rank all-reduce; not built for the sharded path')
        torch.manual_seed(self.args.random_seed)
        with contextlib.redirect_stdout(io.StringIO()):
            self.teacherGNN = TeacherGNN(self.args, None).to(self.device)
        self.teacherGNN.model.model.dglgraph = self.sgraph
        self.replicated = [p for n, p in self.teacherGNN.named_parameters() if not n.endswith('.le') and n != 'embs']
        self.optimizer = self.optfun(self.teacherGNN.parameters(), lr=self.args.lr, weight_decay=self.args.weight_decay)

    def load_full_state_dict(self, sd_full):
        """Loads a single-GPU (unsharded) TeacherGNN state_dict: replicated tensors as they are, the per-node tables
        (structural embeddings `le`, learnable inputs `embs`) cut to this rank's row block."""
        lo, hi = self.part.lo(), self.part.hi()
        sd = {}
        for k, v in sd_full.items():
            per_node = (k.endswith('.le') or k == 'embs') and v.dim() == 2 and v.shape[0] == self._n
            sd[k] = v[lo:hi].clone() if per_node else v
        self.teacherGNN.load_state_dict(sd)

    def graph(self):
        return self.sgraph

    def global_nodes(self):
        return self._n

    def global_edges(self):
        return self._e

    def train_step(self):
        from . import ops
        m = self.teacherGNN
        m.train()
        out = m.get_3_embs(self.x, self.edge_index).emb4classi_full
        # local numerator / global count; the global loss is the sum over ranks
        loss = ops.nll_logsoftmax(out, self.y, self.train_mask, self.n_train)
        if m.se_reg_all is not None:
            loss = loss + self.args.se_reg * m.se_reg_all / self.world   # se_reg_all is already global; count it once
        self.optimizer.zero_grad()
        loss.backward()
        allreduce_grads(self.replicated, self.group)
        self.optimizer.step()
        total = loss.detach().clone()
        _all_reduce(total, group=self.group)
        return total
