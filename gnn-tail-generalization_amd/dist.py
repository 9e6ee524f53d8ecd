"""Node-sharded TeacherGNN over the GPUs of one node (one process per GPU, torch.distributed
`nccl` = RCCL over xGMI).  New relative to the reference, which is single-device.

Partition: 1-D contiguous row blocks.  'edges' (default): boundaries from the prefix sum of
in-degree + NODE_WEIGHT per node, so every rank gets the same share of the step's work (the aggregation
is per edge, the dense stages per node; NODE_WEIGHT = their cost ratio measured on one MI355X).  'rows':
equal row counts.  Rank p owns rows [lo(p), hi(p)) of x / y / masks / activations / structural embeddings.

Ingest: every rank keeps only the edges whose destination (forward CSR) resp. source (reverse CSR) it owns
and builds its row block from those — no rank ever builds the whole graph's CSR.

Exchange (the only data-path collective) before each aggregation, 'halo' form (default): a rank receives
exactly the remote rows its edges reference.  Per orientation a HaloPlan is built once — unique remote
column ids grouped by owner, owners learn which rows to send through one all-to-all of counts and one of
ids.  The row block is split by column owner into an INTERIOR CSR (columns = local rows) and a HALO CSR
(columns = slots of the receive buffer), and the aggregation runs as

    pack rows for the peers -> all_to_all_single(async)      [RCCL stream, xGMI]
    interior pass: raw sums over local columns                [compute stream, overlaps the exchange]
    wait -> halo pass: starts from the interior sums, applies the epilogue once (cb_spmm_csr_acc_f32)

COLDBREW_OVERLAP=0 keeps the single-pass form (one CSR over [local rows | halo rows] after a blocking
exchange); COLDBREW_EXCHANGE=allgather the all-gather baseline (equal-row partition only); COLDBREW_HALO_WIRE=bf16
(opt-in, outside the 1e-4 parity) halves the bytes on the links by sending the halo rows as bfloat16.
The backward of the aggregation is the same exchange on the gradient followed by the reverse-orientation
passes (own plan; aliasing the forward one when the edge multiset is symmetric on every rank).  Everything
else is row-local; small all-reduces cover the replicated weights' gradients, the loss numerator, sum(E^2)
of the structural-embedding regulariser and the column statistics of the norm tricks (norms_hip.py).

The bookkeeping here is device-agnostic torch / torch.distributed code; every pass over node data goes
through a compute object (HipCompute: the C-ABI kernels).  The CPU tests (gloo, world_size 2-3) plug
their own oracle-backed compute object in (tests/dist_cpu_compute.py) — there is no CPU arithmetic in
this module.
"""
import contextlib
import io
import os

import torch
import torch.distributed as dist

from . import _lib

NODE_WEIGHT = 12      # one node's dense work (GEMMs, elementwise) ~ 12 edges' aggregation work per step (S-pl10M profile)


# ---------------------------------------------------------------------------------------------------------
# collectives (gloo cannot move device tensors for every collective: staged through the host in that case —
# used to run the whole sharded HIP path with several processes on ONE GPU; production = nccl, no staging)
# ---------------------------------------------------------------------------------------------------------
def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _all_reduce(t, op=dist.ReduceOp.SUM, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


class _Done:
    def wait(self):
        return True


def _all_to_all_single(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    """Returns a work handle when async_op (already finished for the staged / gloo-CPU forms)."""
    if _staged(inp, group) or _staged(out, group):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(ho, hi, out_splits, in_splits, group=group)
        out.copy_(ho)
        return _Done()
    w = dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)
    return w if async_op else _Done()


def _all_gather_into_tensor(out, inp, group=None):
    if _staged(inp, group):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(ho, hi, group=group)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _broadcast(t, src=0, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


# ---------------------------------------------------------------------------------------------------------
# partition
# ---------------------------------------------------------------------------------------------------------
class Partition:
    """Contiguous row blocks: rank p owns [lo(p), hi(p)).  bounds = P+1 ascending ints (default: equal rows)."""

    def __init__(self, n_nodes, world, rank, bounds=None, kind='rows'):
        self.N, self.world, self.rank, self.kind = int(n_nodes), int(world), int(rank), kind
        if bounds is None:
            r = (self.N + self.world - 1) // self.world
            bounds = [min(p * r, self.N) for p in range(self.world + 1)]
        self.bounds = [int(b) for b in bounds]
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or self.bounds[-1] != self.N \
                or any(self.bounds[i] > self.bounds[i + 1] for i in range(self.world)):
            raise ValueError(f'bad partition bounds {self.bounds} for N={self.N}, world={self.world}')
        self.R = max(self.bounds[p + 1] - self.bounds[p] for p in range(self.world))     # largest block (all-gather slot size)
        self._bt = {}

    @classmethod
    def balanced(cls, in_degree, world, rank, node_weight=NODE_WEIGHT):
        """Boundaries at equal shares of sum_v (in_degree[v] + node_weight): SURVEY.md §8e 'balance on edges-per-rank
        (prefix sum over in-degree), not nodes'.  Deterministic in the degree vector, so all ranks agree."""
        n = int(in_degree.numel())
        cost = torch.cumsum(in_degree.to(torch.int64) + int(node_weight), 0)
        total = int(cost[-1]) if n else 0
        targets = torch.tensor([(total * p) // world for p in range(1, world)], dtype=torch.int64, device=cost.device)
        cuts = torch.searchsorted(cost, targets, right=False).tolist() if world > 1 and n else []
        bounds = [0] + [min(int(c) + 1, n) for c in cuts] + [n]
        for i in range(1, len(bounds)):
            bounds[i] = max(bounds[i], bounds[i - 1])
        return cls(n, world, rank, bounds, kind='edges')

    def lo(self, p=None):
        return self.bounds[self.rank if p is None else p]

    def hi(self, p=None):
        return self.bounds[(self.rank if p is None else p) + 1]

    @property
    def n_local(self):
        return self.hi() - self.lo()

    @property
    def padded(self):
        return self.R * self.world

    def slice_rows(self, t):
        return t[self.lo():self.hi()]

    def owner(self, node_ids):
        key = node_ids.device
        if key not in self._bt:
            self._bt[key] = torch.tensor(self.bounds[1:], dtype=torch.int64, device=key)
        return torch.bucketize(node_ids.to(torch.int64), self._bt[key], right=True)


# ---------------------------------------------------------------------------------------------------------
# compute object: every pass over node data (the product has exactly one implementation)
# ---------------------------------------------------------------------------------------------------------
class HipCompute:
    """The C-ABI kernels of libcoldbrew_hip.so behind the few operations the sharded path needs."""

    def csr(self, rows, cols, n_rows, n_cols):
        """Local row block from (row, col) pairs: CSR with ascending columns inside a row (graph.CSRGraph.from_pairs)."""
        from .graph import CSRGraph
        return CSRGraph.from_pairs(rows, cols, n_rows, n_cols)

    def spmm(self, g, h, row_scale=None, bias=None, relu=False, acc_init=None, profile=None):
        g.profile = profile
        return g.spmm(h, row_scale=row_scale, bias=bias, relu=relu, acc_init=acc_init)

    def pack_rows(self, x, idx):
        from .ops import gather_rows_by_index
        if x.dtype == torch.bfloat16 and x.shape[1] % 2 == 0 and x.is_contiguous():     # bf16 rows move as packed 32-bit words
            return gather_rows_by_index(x.view(torch.float32), idx).view(torch.bfloat16)
        return gather_rows_by_index(x, idx)

    def act_bwd(self, g, act, row_scale, need_b):
        from .ops import act_bwd
        return act_bwd(g, act, row_scale, want_out=True, want_colsum=need_b)

    def deg_norm(self, deg):
        return deg.to(torch.float32).clamp_(min=1).pow_(-0.5)        # [n_local] vector (GCN.py:206-208,243-245)


class HaloPlan:
    """Who sends which rows to whom for one CSR orientation (built once per graph)."""

    def __init__(self, uniq_remote, part, group=None):
        """uniq_remote: ascending unique remote column ids this rank's edges reference (= grouped by owner)."""
        P = part.world
        dev = uniq_remote.device
        self.n_local, self.n_halo = part.n_local, int(uniq_remote.numel())
        recv_counts = torch.bincount(part.owner(uniq_remote), minlength=P)[:P]
        send_counts = torch.empty_like(recv_counts)
        if P > 1:
            _all_to_all_single(send_counts, recv_counts, group=group)    # how many rows each peer wants from me
        else:
            send_counts.copy_(recv_counts)
        self.recv_counts = [int(v) for v in recv_counts.tolist()]
        self.send_counts = [int(v) for v in send_counts.tolist()]
        wanted = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
        if P > 1:
            _all_to_all_single(wanted, uniq_remote, self.send_counts, self.recv_counts, group=group)
        self.send_idx = (wanted - part.lo()).contiguous()                  # my local rows, in per-destination order
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_local):
            raise RuntimeError('halo plan: a peer requested a row this rank does not own')


class _WidenOnWait:
    """Work handle of a bf16-wire exchange: wait() = the collective's wait + widening of the received rows into the fp32 buffer
    the halo pass reads."""

    def __init__(self, inner, wire_in, out):
        self.inner, self.wire_in, self.out = inner, wire_in, out

    def wait(self):
        self.inner.wait()
        self.out.copy_(self.wire_in)


class _Orientation:
    """One CSR orientation of a rank's row block: the CSR(s) the local passes read and the plan that feeds them."""
    __slots__ = ('whole', 'interior', 'halo', 'plan', 'E', 'rowptr_key', 'col_key')


class ShardedGraph:
    """Row block [lo, hi) of both CSR orientations + local degree norms + the exchange plans.
    Quacks like graph.CSRGraph for GCNConv / ops.aggregate / the fused trunk."""

    def __init__(self, edge_index, n_nodes, part, group=None, exchange='halo', overlap=True, compute=None, wire='f32'):
        self.part, self.group = part, group
        self.exchange_kind = exchange
        if wire not in ('f32', 'bf16'):
            raise ValueError(f'unknown halo wire format {wire!r}')
        # 'bf16' (opt-in, COLDBREW_HALO_WIRE=bf16): halo rows travel rounded to bfloat16 (RNE) and are widened on arrival — half the
        # bytes on the xGMI links, the bound of the sharded step; the local rows stay fp32.  NOT within the 1e-4 logits parity
        # (2^-9 relative rounding of every remote neighbour row), hence never the default.
        self.wire = wire if exchange == 'halo' else 'f32'
        self.compute = compute if compute is not None else HipCompute()
        self.N_global, self.E_global = int(n_nodes), int(edge_index.shape[1])
        self.row_offset = part.lo()
        lo, hi = part.lo(), part.hi()
        self.N = hi - lo
        self.profile = None
        if exchange not in ('halo', 'allgather'):
            raise ValueError(f'unknown exchange {exchange!r}')
        if exchange == 'allgather' and part.kind != 'rows':
            raise ValueError("the all-gather baseline needs the equal-row partition (COLDBREW_PARTITION=rows)")
        self.overlap = bool(overlap) and exchange == 'halo' and part.world > 1
        src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
        bad = int(((src < 0) | (src >= n_nodes) | (dst < 0) | (dst >= n_nodes)).sum()) if src.numel() else 0
        if bad:
            raise ValueError(f'edge_index has {bad} edges with an endpoint outside [0, {n_nodes})')
        mf = (dst >= lo) & (dst < hi)                   # forward block: rows = destinations I own, columns = sources (GCN.py:238)
        mb = (src >= lo) & (src < hi)                   # reverse block: rows = sources I own, columns = destinations
        rf, cf = dst[mf] - lo, src[mf]
        rb, cb = src[mb] - lo, dst[mb]
        del mf, mb
        self.E = int(rf.numel())
        in_deg = torch.bincount(rf, minlength=self.N)[:self.N]
        out_deg = torch.bincount(rb, minlength=self.N)[:self.N]
        self.norm_in = self.compute.deg_norm(in_deg)
        self.norm_out = self.compute.deg_norm(out_deg)
        nz = torch.tensor([int((in_deg == 0).sum())], dtype=torch.int64, device=src.device)
        # one global decision each, or the ranks would disagree on which collectives follow
        same = bool(rf.numel() == rb.numel() and torch.equal(torch.sort(rf * n_nodes + cf)[0], torch.sort(rb * n_nodes + cb)[0]))
        flag = torch.tensor([1 if same else 0], dtype=torch.int64, device=src.device)
        if part.world > 1:
            _all_reduce(nz, group=group)
            _all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.n_zero_in_degree = int(nz.item())
        self.symmetric = bool(flag.item())
        self.f = self._orient(rf, cf)
        self.b = self.f if self.symmetric else self._orient(rb, cb)

    # -- build one orientation ---------------------------------------------------------------------------
    def _orient(self, rows, cols):
        part, lo, hi = self.part, self.part.lo(), self.part.hi()
        o = _Orientation()
        o.E = int(rows.numel())
        o.whole = o.interior = o.halo = o.plan = None
        if self.exchange_kind == 'allgather':       # columns index the gathered [P*R, d] matrix (equal blocks: identity)
            o.whole = self.compute.csr(rows, cols, self.N, part.padded)
            return o
        remote = (cols < lo) | (cols >= hi)
        uniq, inv = torch.unique(cols[remote], return_inverse=True)       # ascending ids = grouped by owner
        o.plan = HaloPlan(uniq, part, self.group)
        if self.overlap:
            o.interior = self.compute.csr(rows[~remote], cols[~remote] - lo, self.N, self.N)
            o.halo = self.compute.csr(rows[remote], inv, self.N, max(o.plan.n_halo, 1))
        else:
            new_col = cols - lo
            new_col[remote] = self.N + inv
            o.whole = self.compute.csr(rows, new_col, self.N, self.N + o.plan.n_halo)
        return o

    # -- CSRGraph-like surface ---------------------------------------------------------------------------
    def check_zero_in_degree(self):
        from .graph import ZeroInDegreeError
        if self.n_zero_in_degree:
            raise ZeroInDegreeError('There are 0-in-degree nodes in the graph')

    def number_of_nodes(self):
        return self.N_global

    def number_of_edges(self):
        return self.E_global

    def algorithmic_bytes(self, d, elem=4, row_scale=True, bias=True, src_elem=None):
        """SURVEY.md §8(d) bytes of THIS RANK's share of one aggregation (local edges, local rows)."""
        b = self.f.E * (d * (src_elem or elem) + 4) + self.N * (d * elem + 4)
        return b + (4 * self.N if row_scale else 0) + (d * elem if bias else 0)

    # -- exchange + aggregation --------------------------------------------------------------------------
    def exchange(self, x_local, transpose=False):
        """Blocking form: [n_local, d] -> the matrix the single-pass CSR reads ([local | halo] or the gathered [P*R, d])."""
        o = self.b if transpose else self.f
        if self.exchange_kind == 'allgather':
            return gather_rows(x_local, self.part, self.group)
        if self.part.world == 1:
            return x_local
        plan = o.plan
        ext = torch.empty((plan.n_local + plan.n_halo, x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
        ext[:plan.n_local] = x_local
        send = self.compute.pack_rows(x_local, plan.send_idx)
        if self.wire == 'bf16' and x_local.dtype == torch.float32:
            wire_in = torch.empty((plan.n_halo, x_local.shape[1]), dtype=torch.bfloat16, device=x_local.device)
            _all_to_all_single(wire_in.view(torch.uint8), send.to(torch.bfloat16).view(torch.uint8), plan.recv_counts, plan.send_counts,
                               group=self.group)
            ext[plan.n_local:] = wire_in
            return ext
        _all_to_all_single(ext[plan.n_local:], send, plan.recv_counts, plan.send_counts, group=self.group)
        return ext

    def start_halo(self, x_local, transpose=False):
        """Overlapped form, first half: pack + asynchronous all-to-all.  Returns (receive buffer, work handle, send buffer)."""
        plan = (self.b if transpose else self.f).plan
        send = self.compute.pack_rows(x_local, plan.send_idx)
        recv = torch.empty((max(plan.n_halo, 1), x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
        if self.wire == 'bf16' and x_local.dtype == torch.float32:
            send = send.to(torch.bfloat16)
            wire_in = torch.empty((plan.n_halo, x_local.shape[1]), dtype=torch.bfloat16, device=x_local.device)
            inner = _all_to_all_single(wire_in.view(torch.uint8), send.view(torch.uint8), plan.recv_counts, plan.send_counts,
                                       group=self.group, async_op=True)
            return recv, _WidenOnWait(inner, wire_in, recv[:plan.n_halo]), send
        work = _all_to_all_single(recv[:plan.n_halo], send, plan.recv_counts, plan.send_counts, group=self.group, async_op=True)
        return recv, work, send

    def aggregate_start(self, h_local, transpose=False):
        """First half of aggregate(): starts the exchange (overlapped form) and returns a handle for aggregate_finish().  Work
        issued between the two calls (e.g. the previous layer's weight-gradient GEMM in the trunk backward) runs under the exchange."""
        if not self.overlap or h_local.dtype != torch.float32:
            return (h_local, None, None, None)
        recv, work, send = self.start_halo(h_local, transpose)
        return (h_local, recv, work, send)

    def aggregate_finish(self, handle, transpose=False, row_scale=None, bias=None, relu=False):
        h_local, recv, work, send = handle
        o = self.b if transpose else self.f
        c = self.compute
        if work is None:
            return c.spmm(o.whole if o.whole is not None else self._whole(o), self.exchange(h_local, transpose), row_scale, bias, relu,
                          profile=self.profile)
        part = c.spmm(o.interior, h_local, profile=self.profile)             # raw sums over the local columns, overlaps the exchange
        work.wait()
        out = c.spmm(o.halo, recv, row_scale, bias, relu, acc_init=part, profile=self.profile)
        del send
        return out

    def aggregate(self, h_local, transpose=False, row_scale=None, bias=None, relu=False):
        """act(row_scale * (A_block . h) + bias) for this rank's rows; h_local = this rank's rows of h."""
        return self.aggregate_finish(self.aggregate_start(h_local, transpose), transpose, row_scale, bias, relu)

    def _whole(self, o):
        raise RuntimeError('single-pass aggregation requested on a graph built for the overlapped two-pass form '
                           '(bf16-stored rows need COLDBREW_OVERLAP=0)')


def gather_rows(x_local, part, group=None, out=None):
    """All-gather of the row shards: [n_local, d] on each rank -> [P*R, d] (rows beyond a block's end are padding)."""
    d = x_local.shape[1]
    if out is None:
        out = torch.empty((part.padded, d), dtype=x_local.dtype, device=x_local.device)
    if x_local.shape[0] == part.R and x_local.is_contiguous():
        src = x_local
    else:   # short block: pad to R rows
        src = torch.zeros((part.R, d), dtype=x_local.dtype, device=x_local.device)
        src[:x_local.shape[0]] = x_local
    _all_gather_into_tensor(out, src, group=group)
    return out


class _ShardedAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, h_local, row_scale, bias, relu):
        out = graph.aggregate(h_local, False, row_scale, bias, relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, relu, bias is not None
        ctx.save_for_backward(out if relu else None, row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        out, row_scale = ctx.saved_tensors
        graph = ctx.graph
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        gs, dbias = graph.compute.act_bwd(g, out if ctx.relu else None, row_scale, need_b)
        dh = graph.aggregate(gs, True) if ctx.needs_input_grad[1] else None
        return None, dh, None, dbias, None


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
        g = g.clone()
        _all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def allreduce_sum(x, group=None):
    return _AllReduceSumFn.apply(x, group)


def sync_initial_state(model, part, group=None, seed=0):
    """Makes a freshly constructed sharded model consistent across ranks: the per-node tables (`le`, `embs`)
    have part.n_local rows, so ranks with different block sizes consume different amounts of the CPU RNG stream before the
    replicated layers are drawn.  (1) every replicated parameter and buffer is broadcast from rank 0; (2) the per-node
    tables are re-drawn from a generator keyed by (seed, first global row), so ranks do not hold identical shards."""
    per_node = [(n, p) for n, p in model.named_parameters() if n.endswith('.le') or n == 'embs']
    names = {n for n, _ in per_node}
    with torch.no_grad():
        for n, t in list(model.named_parameters()) + list(model.named_buffers()):
            if n in names or t.numel() == 0:
                continue
            if part.world > 1:
                _broadcast(t.data, 0, group)
        for i, (n, p) in enumerate(per_node):
            gen = torch.Generator().manual_seed((int(seed) * 1000003 + part.lo()) * 64 + i)
            fresh = torch.randn(p.shape, generator=gen)
            if n == 'embs':
                fresh = fresh * 0.001                      # GNN_normalizations.py:18-22
            p.data.copy_(fresh.to(p.device))


class ShardedTrainer:
    """Multi-GPU driver of the TeacherGNN path: bench.py (train_step) and `torchrun ... main.py` (main -> train_teacherGNN: the
    reference's epoch loop with accuracy counts all-reduced).  Mirrors trainer.train_step / run_testSet / train_teacherGNN."""

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
        exchange = os.environ.get('COLDBREW_EXCHANGE', 'halo')
        kind = os.environ.get('COLDBREW_PARTITION', 'rows' if exchange == 'allgather' else 'edges')
        if kind == 'edges':
            self.part = Partition.balanced(torch.bincount(data.edge_index[1], minlength=self._n), self.world, self.rank)
        else:
            self.part = Partition(self._n, self.world, self.rank)
        self.sgraph = ShardedGraph(data.edge_index, self._n, self.part, group, exchange=exchange,
                                   overlap=os.environ.get('COLDBREW_OVERLAP', '1') != '0'
                                   and getattr(args, 'agg_dtype', 'f32') == 'f32',
                                   wire=os.environ.get('COLDBREW_HALO_WIRE', 'f32'))
        self.n_train = int(data.train_mask.sum().item())
        p = self.part
        self.x = p.slice_rows(data.x).float().contiguous()
        self.y = p.slice_rows(data.y).contiguous()
        self.train_mask = p.slice_rows(data.train_mask).contiguous()
        test_mask = data.test_mask if getattr(data, 'test_mask', None) is not None else ~data.train_mask
        self.test_mask = p.slice_rows(test_mask).contiguous()
        self.n_test = int(test_mask.sum().item())
        self.edge_index = data.edge_index[:, :1]     # placeholder: the cached sharded graph is injected below
        del data
        torch.cuda.empty_cache()
        self.optfun = cb_optim.resolve(args.optfun)
        set_arch_configs(args)
        args.N_nodes_global = self._n
        args.N_nodes = self.part.n_local             # structural-embedding tables are row-sharded

    def setup_teacherGNN(self):
        from .GNN_model.GNN_normalizations import TeacherGNN
        torch.manual_seed(self.args.random_seed)
        with contextlib.redirect_stdout(io.StringIO()):
            self.teacherGNN = TeacherGNN(self.args, None).to(self.device)
        sync_initial_state(self.teacherGNN, self.part, self.group, self.args.random_seed)
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
        from . import norms_hip, ops
        m = self.teacherGNN
        m.train()
        with norms_hip.row_sharding(self.group, self._n):      # column statistics of the norm tricks span all ranks
            out = m.get_3_embs(self.x, self.edge_index).emb4classi_full
            # local numerator / global count; the global loss is the sum over ranks
            loss = ops.nll_logsoftmax(out, self.y, self.train_mask, self.n_train)
            if m.se_reg_all is not None:
                folded = ops.fold_se_reg(m, self.optimizer, self.args.se_reg, m.se_reg_all)
                # se_reg_all is already global; count it once
                loss = loss + (folded if folded is not None else self.args.se_reg * m.se_reg_all) / self.world
            self.optimizer.zero_grad()
            loss.backward()
        allreduce_grads(self.replicated, self.group)
        self.optimizer.step()
        total = loss.detach().clone()
        _all_reduce(total, group=self.group)
        return total

    # -- the reference's epoch on row shards (trainer_node_classification.py:303-369,453-495,672-681) ----------------------
    def run_testSet(self):
        """Eval forward + argmax accuracy on the train / test masks: per-rank hit counts, all-reduced."""
        from . import norms_hip
        m = self.teacherGNN
        m.eval()
        with torch.no_grad(), norms_hip.row_sharding(self.group, self._n):
            out = m.get_3_embs(self.x, self.edge_index).emb4classi_full
        hit = out.argmax(dim=1) == self.y
        cnt = torch.stack([(hit & self.train_mask).sum(), (hit & self.test_mask).sum()]).to(torch.float64)
        _all_reduce(cnt, group=self.group)
        acc_train, acc_test = (cnt / torch.tensor([max(self.n_train, 1), max(self.n_test, 1)], dtype=torch.float64, device=cnt.device)).tolist()
        return acc_train, float('nan'), acc_test, float('nan')

    def train_teacherGNN(self):
        """Epoch loop with the record layout of trainer.train_teacherGNN (want_headtail = 0): returns [[acc_test * 100 per epoch]].
        Every rank returns the same array."""
        import numpy as np
        self.setup_teacherGNN()
        rows = []
        for epoch in range(self.args.epochs):
            loss = float(self.train_step())
            acc_train, _, acc_test, _ = self.run_testSet()
            rows.append([np.log(loss), acc_train * 100, acc_test * 100, 0, 0])
            if epoch % 20 == 0 and self.rank == 0:
                print(f'Ep{epoch:03d}, acc @ train/test: {acc_train * 100:.1f}, {acc_test * 100:.1f} ')
        return np.array(rows).T[[2]]

    def main(self):
        if self.args.train_which != 'TeacherGNN' or self.args.want_headtail:
            raise NotImplementedError('the node-sharded trainer runs --train_which=TeacherGNN with --want_headtail=0')
        return self.train_teacherGNN()
