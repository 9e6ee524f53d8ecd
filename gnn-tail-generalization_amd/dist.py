"""Node-sharded TeacherGNN over the GPUs of one node (one process per GPU, torch.distributed
`nccl` = RCCL over xGMI).  New relative to the reference, which is single-device.

Partition: 1-D contiguous row blocks.  'edges' (default): boundaries from the prefix sum of
in-degree + tuning.T.node_weight per node, so every rank gets the same share of the step's work (the aggregation
is per edge, the dense stages per node; node_weight = their cost ratio measured on one MI355X).  'rows':
equal row counts.  Rank p owns rows [lo(p), hi(p)) of x / y / masks / activations / structural embeddings.

Ingest: every rank keeps only the edges whose destination (forward CSR) resp. source (reverse CSR) it owns
and builds its row block from those — no rank ever builds the whole graph's CSR.

Exchange (the only data-path collective) before each aggregation, 'halo' form (default): a rank receives
exactly the remote rows its edges reference.  Per orientation a HaloPlan is built once — unique remote
column ids grouped by owner, owners learn which rows to send through one all-to-all of counts and one of
ids.  The row block is split by column owner into an INTERIOR CSR (columns = local rows) and HALO CSRs
(columns = slots of a receive buffer), and the aggregation runs as a pipeline of K time slices
(COLDBREW_HALO_SLICES; slice k = the halo rows that live in the k-th row chunk of their owner's block, all
peers at once, so every xGMI link carries 1/K of its traffic per slice — the links of a full mesh work in
parallel, which a peer-by-peer order would give up):

    for k: [producer of row chunk k, e.g. the layer GEMM] -> pack slice k -> all_to_all_single(async)   [RCCL stream, xGMI]
    interior pass: raw sums over local columns                [compute stream, under the exchange]
    for k: wait(k) -> halo pass k on top of the running sums (cb_spmm_csr_acc_f32); the last one applies the epilogue

so pack k+1, the GEMM rows of chunk k+1 and halo pass k-1 all run while slice k is on the links; what stays
exposed is the first chunk's producer + pack and the last slice's halo pass.  K = 1 is the two-pass form of
round 2 (bit for bit).  COLDBREW_OVERLAP=0 keeps the single-pass form (one CSR over [local rows | halo rows]
after a blocking exchange); COLDBREW_EXCHANGE=allgather the all-gather baseline (equal-row partition only);
COLDBREW_HALO_WIRE=bf16 (opt-in, outside the 1e-4 parity) halves the bytes on the links: the pack kernel
writes bf16 (cb_gather_rows_bf16_f32) and the halo passes read the wire buffer as it arrived
(cb_spmm_csr_acc_bf16_f32) — no conversion pass on either side.
Push / pull cover (default, COLDBREW_HALO_COVER=0 switches it off): a remote edge u -> v can be served by shipping the SOURCE row h[u]
(pull: what the plain halo does) or by shipping the owner-side PARTIAL SUM of the destination row, sum of h[u] over the owner's sources
of v (push).  Per ordered rank pair the requester picks a vertex cover of the pair's remote-edge bipartite graph (every edge to its
higher-degree endpoint, one absorb round, never worse than all-pull or all-push): hubs cover most edges of a power-law graph, and the
busiest link carries 26 - 32 % fewer rows on the permuted-id power-law benchmark graph (profiles/r04_halo_cover_study.md; = the pull
on the ogbn-products shape, where the optimum is the pull).  The owner's pack becomes one aggregation over a "send CSR" (a pulled row
= a row with one edge, a pushed row = a row with the edges it sums), the requester's halo CSRs hold one edge per pushed row.  Results
differ from the pull form only in summation order (inside the 1e-4 contract; not bit-identical to it).  With the cover on, a pair's
send list is cut into the K time slices by POSITION (pushed partial sums need all of the owner's rows, so slices cannot follow the
owner's row chunks — and need not: the aggregation + GEMM kernel of the previous stage delivers the whole matrix at once).

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

from .tuning import T

# (tuning.T: node_weight, cover_min_gain, slice_min_bytes, support_max_edge_frac — all read at call time)


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


def init_rccl(rank, world_size, device, **kw):
    """init_process_group('nccl' = RCCL) bound to `device`, with the collectives on HIGH-PRIORITY streams: the halo all-to-all of a
    slice is issued beside an aggregation pass that fills every CU, and its workgroups must be dispatched ahead of that pass's
    (tens of thousands of) pending ones, not behind them."""
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
    except (AttributeError, TypeError):        # a build without the option: default-priority streams
        opts = None
    if opts is not None:
        kw.setdefault('pg_options', opts)
    dist.init_process_group('nccl', rank=rank, world_size=world_size, device_id=torch.device(device), **kw)


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
    def balanced(cls, in_degree, world, rank, node_weight=None):
        """Boundaries at equal shares of sum_v (in_degree[v] + node_weight): SURVEY.md §8e 'balance on edges-per-rank
        (prefix sum over in-degree), not nodes'.  Deterministic in the degree vector, so all ranks agree."""
        n = int(in_degree.numel())
        cost = torch.cumsum(in_degree.to(torch.int64) + int(T.node_weight if node_weight is None else node_weight), 0)
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
        return torch.bucketize(node_ids.to(torch.int64).contiguous(), self._bt[key], right=True)


# ---------------------------------------------------------------------------------------------------------
# compute object: every pass over node data (the product has exactly one implementation)
# ---------------------------------------------------------------------------------------------------------
class HipCompute:
    """The C-ABI kernels of libcoldbrew_hip.so behind the few operations the sharded path needs."""

    def csr(self, rows, cols, n_rows, n_cols):
        """Local row block from (row, col) pairs: CSR with ascending columns inside a row (graph.CSRGraph.from_pairs)."""
        from .graph import CSRGraph
        return CSRGraph.from_pairs(rows, cols, n_rows, n_cols)

    def spmm(self, g, h, row_scale=None, bias=None, relu=False, acc_init=None, profile=None, out=None):
        g.profile = profile
        return g.spmm(h, row_scale=row_scale, bias=bias, relu=relu, acc_init=acc_init, out=out)

    def pack_rows(self, x, idx, wire='f32'):
        from .ops import gather_rows_by_index
        if wire == 'bf16' and x.dtype == torch.float32:                                  # narrowed by the pack kernel itself
            return gather_rows_by_index(x, idx, out_bf16=True)
        if x.dtype == torch.bfloat16 and x.shape[1] % 2 == 0 and x.is_contiguous():     # bf16 rows move as packed 32-bit words
            return gather_rows_by_index(x.view(torch.float32), idx).view(torch.bfloat16)
        return gather_rows_by_index(x, idx)

    def to_wire(self, x, wire):
        """fp32 send buffer -> the wire format (bf16 wire of a cover plan: its pack is an aggregation, which writes fp32)."""
        return x.to(torch.bfloat16) if wire == 'bf16' and x.dtype == torch.float32 else x

    def act_bwd(self, g, act, row_scale, need_b):
        from .ops import act_bwd
        return act_bwd(g, act, row_scale, want_out=True, want_colsum=need_b)

    def deg_norm(self, deg):
        return deg.to(torch.float32).clamp_(min=1).pow_(-0.5)        # [n_local] vector (GCN.py:206-208,243-245)


def slice_weights(k):
    """Relative sizes of the k time slices.  Default: the first and the last slice are half as large as the ones between
    (k >= 3) — the first slice's producer + pack is the pipeline's start-up and the last slice's halo pass its tail, both
    exposed.  COLDBREW_SLICE_WEIGHTS="1,2,2,1" overrides (k comma-separated positive numbers; every rank must see the same)."""
    env = os.environ.get('COLDBREW_SLICE_WEIGHTS')
    if env:
        w = [float(v) for v in env.split(',')]
        if len(w) == k and all(v > 0 for v in w):
            return w
    return [1.0] * k if k < 3 else [1.0] + [2.0] * (k - 2) + [1.0]


def chunk_bounds(lo, hi, k):
    """k+1 ascending global row ids cutting the block [lo, hi) into k row chunks with the relative sizes slice_weights(k)."""
    n = hi - lo
    w = slice_weights(k)
    tot, acc, out = sum(w), 0.0, [lo]
    for i in range(k - 1):
        acc += w[i]
        out.append(lo + min(n, int(n * acc / tot)))
    return out + [hi]


def default_slices(n_halo_rows, d=256):
    """Time slices of the exchange pipeline: COLDBREW_HALO_SLICES, else 4 once an aggregation moves >= 64 MiB of halo rows per rank
    (below that a slice's kernels are launch-sized and the split only adds passes), else 1."""
    env = os.environ.get('COLDBREW_HALO_SLICES')
    if env:
        return max(1, int(env))
    return 4 if int(n_halo_rows) * d * 4 >= T.slice_min_bytes else 1


class HaloPlan:
    """Who sends which rows to whom for one CSR orientation (built once per graph), cut into n_slices time slices: slice k
    holds the requested rows that live in the k-th row chunk of their OWNER's block (chunk_bounds), so an owner can ship
    slice k as soon as rows of chunk k exist and every peer link carries about 1/n_slices of its bytes per slice.

    Requester side: recv_counts[k][q] rows arrive from peer q in slice k (ascending ids); slice_of / slot_of map a position in
    `uniq_remote` to (slice, row of that slice's receive buffer).  Owner side: send_idx[k] = my local rows of slice k in
    per-destination order, send_counts[k][q] of them go to peer q."""

    def __init__(self, uniq_remote, part, group=None, n_slices=1, local_pos=None, n_src=None):
        """uniq_remote: ascending unique remote column ids this rank's edges reference (= grouped by owner).  local_pos (int32 [n_local], a
        compact level of the row-sparse backward): the matrix this rank SENDS from holds only the rows with local_pos >= 0, at those
        positions (n_src of them) — the send lists index it."""
        P, K = part.world, max(1, int(n_slices))
        dev = uniq_remote.device
        self.n_slices = K
        self.n_local, self.n_halo = (part.n_local if local_pos is None else int(n_src)), int(uniq_remote.numel())
        # segment (q, k) = ids in chunk k of owner q; boundaries ascending over (q major, k minor)
        bnd = [b for q in range(P) for b in chunk_bounds(part.lo(q), part.hi(q), K)[:-1]] + [part.N]
        pos = torch.searchsorted(uniq_remote, torch.tensor(bnd, dtype=torch.int64, device=dev))       # [P*K + 1] positions in uniq
        cnt = (pos[1:] - pos[:-1]).view(P, K)                                                          # cnt[q, k]
        snd = torch.empty_like(cnt)
        if P > 1:
            _all_to_all_single(snd.view(-1), cnt.contiguous().view(-1), group=group)                   # row q = what peer q wants from me
        else:
            snd.copy_(cnt)
        cnt_h, snd_h = cnt.tolist(), snd.tolist()
        self.recv_counts = [[int(cnt_h[q][k]) for q in range(P)] for k in range(K)]
        self.send_counts = [[int(snd_h[q][k]) for q in range(P)] for k in range(K)]
        self.n_halo_slice = [sum(c) for c in self.recv_counts]
        per_peer_out, per_peer_in = [sum(snd_h[q]) for q in range(P)], [sum(cnt_h[q]) for q in range(P)]
        wanted = torch.empty(sum(per_peer_out), dtype=torch.int64, device=dev)
        if P > 1:
            _all_to_all_single(wanted, uniq_remote, per_peer_out, per_peer_in, group=group)            # peer-major, ascending per peer
        local = wanted - part.lo()
        if local.numel() and (int(local.min()) < 0 or int(local.max()) >= part.n_local):
            raise RuntimeError('halo plan: a peer requested a row this rank does not own')
        # wanted is laid out (peer q, slice k); the send lists are (slice k, peer q)
        starts, off = {}, 0
        for q in range(P):
            for k in range(K):
                starts[(q, k)] = off
                off += snd_h[q][k]
        mine = chunk_bounds(part.lo(), part.hi(), K)
        self.chunks = [(mine[k] - part.lo(), mine[k + 1] - part.lo()) for k in range(K)]              # local row range of chunk k
        self.send_idx = []
        for k in range(K):
            pieces = [local[starts[(q, k)]: starts[(q, k)] + snd_h[q][k]] for q in range(P)]
            idx = torch.cat(pieces).contiguous() if pieces else local[:0]
            r0, r1 = self.chunks[k]
            if idx.numel() and (int(idx.min()) < r0 or int(idx.max()) >= r1):
                raise RuntimeError('halo plan: the peers cut their requests at other chunk boundaries than this rank')
            self.send_idx.append(self._to_pos(idx, local_pos))
        # requester side: position j of uniq_remote -> (slice, slot in that slice's receive buffer [peer-major, ascending])
        seg = torch.bucketize(torch.arange(self.n_halo, device=dev), pos[1:], right=True)              # segment q*K + k of position j
        roff = torch.cumsum(cnt, 0) - cnt                                                              # rows of peers < q in slice k
        base = (roff.reshape(-1) - pos[:-1])
        self.slice_of = seg % K
        self.slot_of = torch.arange(self.n_halo, device=dev) + base[seg] if self.n_halo else seg
        # K = 1 views (the blocking / single-pass forms)
        self.recv_counts_all, self.send_counts_all = per_peer_in, per_peer_out
        self.send_idx_all = self._to_pos(local.contiguous(), local_pos)

    @staticmethod
    def _to_pos(idx, local_pos):
        if local_pos is None:
            return idx
        pos = local_pos[idx].to(torch.int64)
        if pos.numel() and int(pos.min()) < 0:
            raise RuntimeError('halo plan: a peer requested a row outside the support this level sends from')
        return pos.contiguous()

    cover = False

    def pack(self, compute, x_local, k, wire):
        """Send buffer of slice k: the requested rows of x_local in per-destination order (cb_gather_rows_f32 / _bf16_f32)."""
        return compute.pack_rows(x_local, self.send_idx[k], wire)

    def halo_edges(self, rows, inv):
        """(rows, slice, slot) of the requester's halo edges: remote edge e reads slot slot_of[inv[e]] of slice slice_of[inv[e]]."""
        return rows, self.slice_of[inv], self.slot_of[inv]


def merged_first_pass_enabled():
    return os.environ.get('COLDBREW_MERGED_FIRST_PASS', '1') != '0'


def alloc_exchanged(graph, n_rows, d, dtype=torch.float32, device=None):
    """An [n_rows, d] matrix that an aggregation of `graph` will exchange, allocated with ROOM behind it for the first halo slice
    (ShardedGraph.halo_room rows): start_halo then receives slice 0 right behind the local rows and the interior pass and the first halo pass
    become one (o.first).  On anything but a node-sharded graph: a plain matrix."""
    room = int(getattr(graph, 'halo_room', 0) or 0)
    device = device if device is not None else graph.norm_in.device
    if room <= 0:
        return torch.empty((n_rows, d), dtype=dtype, device=device)
    buf = torch.empty((n_rows + room, d), dtype=dtype, device=device)
    v = buf[:n_rows]
    v._cb_room = room          # (an attribute of THIS tensor object: a matrix that merely happens to sit in a larger allocation is never taken for one)
    return v


def cover_slices_enabled():
    return os.environ.get('COLDBREW_HALO_COVER', '1') != '0'




def choose_cover(u_idx, v_idx, q_u, q_v, n_u, n_v, P, min_gain=None):
    """Push / pull assignment of the remote edges of one requester.  u_idx / v_idx: per edge, index of its source among the n_u distinct
    remote sources / of its (owner, destination) pair among the n_v distinct pairs; q_u / q_v: owner of each distinct source / pair.
    Returns pull [E] bool (True: the source row is shipped; False: the edge is summed by the owner into the destination's partial row).
    Heuristic vertex cover per owner: each edge goes to its higher-degree endpoint (ties: pull), sources that are shipped anyway absorb
    all their edges, then destinations that are pushed anyway absorb theirs; per owner the result competes with all-pull and all-push and
    the smallest row count wins — but the pull is kept unless it is beaten by min_gain (a pushed row costs its owner an aggregation over the
    edges it sums; on the ogbn-products shape a 0.4 % smaller all-push cover would double the local work).  Within 1 - 3 % of the
    Hopcroft-Karp optimum on the power-law graphs (profiles/r04_halo_cover_study.md)."""
    from . import tuning
    min_gain = tuning.T.cover_min_gain if min_gain is None else min_gain      # (read at call time; `T` below is a local)
    dev = u_idx.device
    du = torch.bincount(u_idx, minlength=n_u)
    dv = torch.bincount(v_idx, minlength=n_v)
    pull = du[u_idx] >= dv[v_idx]
    S = torch.zeros(n_u, dtype=torch.bool, device=dev)
    S[u_idx[pull]] = True
    pull = pull | S[u_idx]
    T = torch.zeros(n_v, dtype=torch.bool, device=dev)
    T[v_idx[~pull]] = True
    pull = pull & ~T[v_idx]
    S = torch.zeros(n_u, dtype=torch.bool, device=dev)
    S[u_idx[pull]] = True
    c_pull = torch.bincount(q_u, minlength=P)
    c_push = torch.bincount(q_v, minlength=P)
    c_mix = torch.bincount(q_u[S], minlength=P) + torch.bincount(q_v[T], minlength=P)
    best = torch.minimum(c_mix, c_push)
    all_pull = best.double() > (1.0 - float(min_gain)) * c_pull.double()      # not worth it (dense graphs: every source is referenced anyway)
    all_push = ~all_pull & (c_push <= c_mix)
    q_e = q_u[u_idx]
    return (pull | all_pull[q_e]) & ~all_push[q_e]


class CoverPlan:
    """Push / pull exchange plan of one CSR orientation (module docstring).  Per owner q the requester's list L_q of pulled source rows
    and pushed partial rows is cut into n_slices pieces by position (slice_weights); slice k's
    receive buffer is the concatenation over q of L_q's k-th piece (pulled and pushed rows interleaved in proportion inside L_q).

    Requester side: recv_counts[k][q]; halo_edges() = the edges of the halo CSRs (a pulled source keeps its edges, a pushed destination
    has ONE edge to its partial row).  Owner side: send_csr[k] = CSR over the rows of slice k's send buffer (columns = local rows: one
    per pulled row, the summed sources per pushed row), send_counts[k][p]; pack() is an aggregation over it."""
    cover = True

    @staticmethod
    def assign(rows, cols, part):
        """The push / pull assignment of this rank's remote edges (requester side, no communication): what __init__ builds the plan from.
        ['rows_pull'] / ['rows_cover'] = rows this rank would receive with the plain pull / with the cover."""
        nl = max(part.n_local, 1)
        dev = cols.device
        q_e = part.owner(cols)
        uu, ui = torch.unique(cols, return_inverse=True)                         # distinct remote sources (ascending: grouped by owner)
        vv, vi = torch.unique(q_e * nl + rows, return_inverse=True)              # distinct (owner, destination) pairs (owner-major)
        q_u, q_v = part.owner(uu), vv // nl
        pull = choose_cover(ui, vi, q_u, q_v, int(uu.numel()), int(vv.numel()), part.world)
        S = torch.zeros(uu.numel(), dtype=torch.bool, device=dev)
        S[ui[pull]] = True
        T = torch.zeros(vv.numel(), dtype=torch.bool, device=dev)
        T[vi[~pull]] = True
        return dict(q_e=q_e, uu=uu, ui=ui, vv=vv, vi=vi, q_u=q_u, q_v=q_v, pull=pull, S=S, T=T, rows_pull=int(uu.numel()),
                    rows_cover=int(S.sum()) + int(T.sum()))

    def __init__(self, rows, cols, part, group, n_slices, compute, assignment=None, local_pos=None, n_src=None):
        """rows / cols: local destination / global source of this rank's REMOTE edges (this orientation).  local_pos / n_src: see HaloPlan
        (the send CSRs' columns index the compact matrix this rank sends from)."""
        P, K = part.world, max(1, int(n_slices))
        dev = cols.device
        nl = max(part.n_local, 1)
        self.n_slices, self.n_local = K, part.n_local
        a = assignment if assignment is not None else CoverPlan.assign(rows, cols, part)
        q_e, uu, ui, vv, vi, q_u, q_v, pull, S, T = (a[k] for k in ('q_e', 'uu', 'ui', 'vv', 'vi', 'q_u', 'q_v', 'pull', 'S', 'T'))
        self.n_pull_only = a['rows_pull']                                        # rows the plain pull would move (diagnostics)
        su, tv = torch.nonzero(S).reshape(-1), torch.nonzero(T).reshape(-1)      # shipped sources / pushed pairs, owner-major ascending
        n_pull = torch.bincount(q_u[su], minlength=P)
        n_push = torch.bincount(q_v[tv], minlength=P)
        n_qt = n_pull + n_push
        n_q = n_qt.tolist()
        excl = lambda c: torch.cumsum(c, 0) - c      # noqa: E731
        # position of every item in its owner's list L_q: pulled and pushed rows INTERLEAVED in proportion (an item's key is its relative
        # rank inside its kind), so that every positional slice carries the same mix — a pushed row is cheap for the requester (one edge)
        # and dear for the owner (the edges it sums); kinds laid end to end would pile the owners' summing into the last, exposed slice
        q_it = torch.cat([q_u[su], q_v[tv]])
        rk = torch.cat([(torch.arange(su.numel(), device=dev) - excl(n_pull)[q_u[su]]).double() + 0.5,
                        (torch.arange(tv.numel(), device=dev) - excl(n_push)[q_v[tv]]).double() + 0.5])
        frac = rk / torch.cat([n_pull[q_u[su]], n_push[q_v[tv]]]).clamp(min=1).double()
        order_it = torch.sort(q_it.double() + frac, stable=True)[1]
        pos_it = torch.empty_like(order_it)
        pos_it[order_it] = torch.arange(order_it.numel(), device=dev) - excl(n_qt)[q_it[order_it]]
        pos_u, pos_t = pos_it[:su.numel()], pos_it[su.numel():]
        w = slice_weights(K)
        tot, acc, wfrac = sum(w), 0.0, [0.0]
        for k in range(K - 1):
            acc += w[k]
            wfrac.append(acc / tot)
        cuts_h = [[min(n, int(n * f)) for f in wfrac] + [n] for n in n_q]         # cuts_h[q][k]: first position of slice k in L_q
        cuts = torch.tensor(cuts_h, dtype=torch.int64, device=dev).view(P, K + 1)
        cnt = cuts[:, 1:] - cuts[:, :-1]                                          # cnt[q, k] rows of slice k from owner q
        base = (torch.cumsum(cnt, 0) - cnt).t().contiguous()                      # base[k, q]: first slot of owner q in slice k's buffer

        def place(qi, pos):      # (slice, slot) of items (owner qi, position pos)
            k_ = (pos.unsqueeze(1) >= cuts[qi][:, 1:K]).sum(1) if K > 1 else torch.zeros_like(pos)
            return k_, base[k_, qi] + pos - cuts[qi, k_], pos - cuts[qi, k_]
        k_u, slot_u, r_u = place(q_u[su], pos_u)
        k_t, slot_t, r_t = place(q_v[tv], pos_t)
        self.recv_counts = [[int(c) for c in row] for row in cnt.t().tolist()]
        self.n_halo_slice = [sum(c) for c in self.recv_counts]
        self.n_halo = sum(self.n_halo_slice)
        self.n_pulled, self.n_pushed = int(su.numel()), int(tv.numel())
        # requester's halo edges: pulled sources keep their edges, every pushed pair has one edge to its partial row
        u_item = torch.full((max(int(uu.numel()), 1),), -1, dtype=torch.int64, device=dev)
        u_item[su] = torch.arange(su.numel(), device=dev)
        it = u_item[ui[pull]]
        self._h_rows = torch.cat([rows[pull], vv[tv] % nl])
        self._h_slice = torch.cat([k_u[it], k_t])
        self._h_slot = torch.cat([slot_u[it], slot_t])
        # what the owners must know: per (slice, row of my segment in their send buffer) the local rows to sum — (k, r, column) triples
        t_item = torch.full((max(int(vv.numel()), 1),), -1, dtype=torch.int64, device=dev)
        t_item[tv] = torch.arange(tv.numel(), device=dev)
        pe = t_item[vi[~pull]]
        trip = torch.stack([torch.cat([k_u, k_t[pe]]), torch.cat([r_u, r_t[pe]]), torch.cat([uu[su], cols[~pull]])], 1)
        owner_of = torch.cat([q_u[su], q_e[~pull]])
        order = torch.sort(owner_of, stable=True)[1]
        trip = trip[order].contiguous()
        n_out = torch.bincount(owner_of, minlength=P)
        snd, n_in = torch.empty_like(cnt), torch.empty_like(n_out)
        if P > 1:
            _all_to_all_single(snd.view(-1), cnt.contiguous().view(-1), group=group)              # row p = what requester p expects from me per slice
            _all_to_all_single(n_in, n_out, group=group)
        else:
            snd.copy_(cnt)
            n_in.copy_(n_out)
        n_in_h, n_out_h = n_in.tolist(), n_out.tolist()
        got = torch.empty((sum(n_in_h), 3), dtype=torch.int64, device=dev)
        if P > 1:
            _all_to_all_single(got.view(-1), trip.view(-1), [3 * c for c in n_in_h], [3 * c for c in n_out_h], group=group)
        snd_h = snd.tolist()
        self.send_counts = [[int(snd_h[p_][k]) for p_ in range(P)] for k in range(K)]
        sbase = (torch.cumsum(snd, 0) - snd).t().contiguous()                                      # sbase[k, p]: first row of requester p in slice k's send buffer
        req = torch.repeat_interleave(torch.arange(P, device=dev), n_in)
        gk, gr, gc = got[:, 0], got[:, 1], got[:, 2] - part.lo()
        if got.numel() and (int(gc.min()) < 0 or int(gc.max()) >= part.n_local or int(gk.min()) < 0 or int(gk.max()) >= K):
            raise RuntimeError('cover plan: a peer requested a row this rank does not own')
        n_send_cols = nl
        if local_pos is not None:
            gc = HaloPlan._to_pos(gc, local_pos)
            n_send_cols = max(int(n_src), 1)
        self.send_csr, self.n_send_slice = [], []
        for k in range(K):
            m = gk == k if K > 1 else slice(None)
            n_rows_k = sum(self.send_counts[k])
            srow = sbase[k][req[m]] + gr[m]
            if srow.numel() and int(srow.max()) >= n_rows_k:
                raise RuntimeError('cover plan: a peer addressed a row beyond its segment of the send buffer')
            self.send_csr.append(compute.csr(srow, gc[m], max(n_rows_k, 1), n_send_cols))
            self.n_send_slice.append(n_rows_k)
        self.chunks = [(0, part.n_local)] + [(part.n_local, part.n_local)] * (K - 1)      # a producer, if any, delivers the whole matrix before slice 0
        self.recv_counts_all = [sum(self.recv_counts[k][q] for k in range(K)) for q in range(P)]
        self.send_counts_all = [sum(self.send_counts[k][q] for k in range(K)) for q in range(P)]

    def pack(self, compute, x_local, k, wire):
        """Send buffer of slice k = an aggregation over the send CSR (pulled rows: one edge, i.e. a copy; pushed rows: the partial sums)."""
        n_k = self.n_send_slice[k]
        out = compute.spmm(self.send_csr[k], x_local if x_local.shape[0] else x_local.new_zeros((1, x_local.shape[1])))
        return compute.to_wire(out[:n_k], wire)

    def halo_edges(self, rows=None, inv=None):
        return self._h_rows, self._h_slice, self._h_slot


class SupportLevel:
    """One level of the row-sparse backward on a rank: the level's orientation and the row spaces of what it reads (src) and writes (dst);
    None = all local rows."""
    __slots__ = ('orient', 'src', 'dst')

    def __init__(self, orient, src, dst):
        self.orient, self.src, self.dst = orient, src, dst


class _Orientation:
    """One CSR orientation of a rank's row block: the CSR(s) the local passes read and the plan that feeds them."""
    __slots__ = ('whole', 'interior', 'halo', 'plan', 'E', 'rowptr_key', 'col_key', 'first', 'n_cols_local')


class ShardedGraph:
    """Row block [lo, hi) of both CSR orientations + local degree norms + the exchange plans.
    Quacks like graph.CSRGraph for GCNConv / ops.aggregate / the fused trunk."""

    def __init__(self, edge_index, n_nodes, part, group=None, exchange='halo', overlap=True, compute=None, wire='f32', n_slices=None,
                 local_edges=None, cover=None):
        """edge_index: the whole [2, E] edge list (every rank filters its own blocks), or None with
        local_edges = (fwd [2, Ef], rev [2, Er]): the edges whose DESTINATION / SOURCE this rank owns, as scattered by the
        loading rank (ShardedTrainer: no rank but the loader ever holds the whole graph); E_global is then all-reduced."""
        self.part, self.group = part, group
        self.exchange_kind = exchange
        if wire not in ('f32', 'bf16'):
            raise ValueError(f'unknown halo wire format {wire!r}')
        # 'bf16' (opt-in, COLDBREW_HALO_WIRE=bf16): halo rows leave the pack kernel rounded to bfloat16 (RNE) and the halo passes
        # read them as they arrive — half the bytes on the xGMI links, the bound of the sharded step; the local rows stay fp32.
        # NOT within the 1e-4 logits parity (2^-9 relative rounding of every remote neighbour row), hence never the default.
        self.wire = wire if exchange == 'halo' else 'f32'
        self.compute = compute if compute is not None else HipCompute()
        self.N_global = int(n_nodes)
        self.row_offset = part.lo()
        lo, hi = part.lo(), part.hi()
        self.N = hi - lo
        self.profile = None
        self.exchange_log = None        # bench.py --gpus N sets a list: (event before the first send, event after the last wait) per aggregation
        if exchange not in ('halo', 'allgather'):
            raise ValueError(f'unknown exchange {exchange!r}')
        if exchange == 'allgather' and part.kind != 'rows':
            raise ValueError("the all-gather baseline needs the equal-row partition (COLDBREW_PARTITION=rows)")
        self.overlap = bool(overlap) and exchange == 'halo' and part.world > 1
        # push / pull cover of the remote edges (module docstring): overlapped halo form only; None = COLDBREW_HALO_COVER (default on)
        self.cover = self.overlap and (cover_slices_enabled() if cover is None else bool(cover))
        self._cover_forced = cover == 'force' or os.environ.get('COLDBREW_HALO_COVER') == 'force'      # tests: the cover plan whatever it saves
        self._n_slices_req = n_slices
        if local_edges is None:
            src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
            bad = int(((src < 0) | (src >= n_nodes) | (dst < 0) | (dst >= n_nodes)).sum()) if src.numel() else 0
            if bad:
                raise ValueError(f'edge_index has {bad} edges with an endpoint outside [0, {n_nodes})')
            mf = (dst >= lo) & (dst < hi)                   # forward block: rows = destinations I own, columns = sources (GCN.py:238)
            mb = (src >= lo) & (src < hi)                   # reverse block: rows = sources I own, columns = destinations
            rf, cf = dst[mf] - lo, src[mf]
            rb, cb = src[mb] - lo, dst[mb]
            del mf, mb
            self.E_global = int(edge_index.shape[1])
        else:
            ef, eb = local_edges
            rf, cf = ef[1].to(torch.int64) - lo, ef[0].to(torch.int64)
            rb, cb = eb[0].to(torch.int64) - lo, eb[1].to(torch.int64)
            for r_, c_ in ((rf, cf), (rb, cb)):
                if r_.numel() and (int(r_.min()) < 0 or int(r_.max()) >= self.N or int(c_.min()) < 0 or int(c_.max()) >= n_nodes):
                    raise ValueError('local_edges: an edge does not belong to this rank\'s row block / lies outside the graph')
            tot = torch.tensor([int(rf.numel())], dtype=torch.int64, device=rf.device)
            if part.world > 1:
                _all_reduce(tot, group=group)
            self.E_global = int(tot.item())
        self.E = int(rf.numel())
        # SURVEY.md 8(b) for E >= 2^31: the int32 guarantee is per rank — every rank's two blocks must stay below 2^31 edges (checked on
        # all ranks together, so that no rank walks into a collective the others skip)
        big = torch.tensor([int(max(rf.numel(), rb.numel()) >= 2 ** 31 - 1 or n_nodes >= 2 ** 31 - 1)], dtype=torch.int64, device=rf.device)
        if part.world > 1:
            _all_reduce(big, op=dist.ReduceOp.MAX, group=group)
        if int(big.item()):
            raise ValueError(f'a rank\'s row block holds >= 2^31 edges (this rank: {int(rf.numel())} of {self.E_global}) or the graph >= 2^31 nodes: '
                             f'int32 device indices need more ranks (world = {part.world})')
        in_deg = torch.bincount(rf, minlength=self.N)[:self.N]
        out_deg = torch.bincount(rb, minlength=self.N)[:self.N]
        self.in_deg = in_deg
        self.norm_in = self.compute.deg_norm(in_deg)
        self.norm_out = self.compute.deg_norm(out_deg)
        nz = torch.tensor([int((in_deg == 0).sum())], dtype=torch.int64, device=rf.device)
        # one global decision each, or the ranks would disagree on which collectives follow
        same = bool(rf.numel() == rb.numel() and torch.equal(torch.sort(rf * n_nodes + cf)[0], torch.sort(rb * n_nodes + cb)[0]))
        flag = torch.tensor([1 if same else 0], dtype=torch.int64, device=rf.device)
        if part.world > 1:
            _all_reduce(nz, group=group)
            _all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.n_zero_in_degree = int(nz.item())
        self.symmetric = bool(flag.item())
        self.f = self._orient(rf, cf)
        self.b = self.f if self.symmetric else self._orient(rb, cb)
        # the reverse orientation's edge list (local row, global column) stays: the row-sparse backward builds its level orientations from it
        self._rev_edges = (rf, cf) if self.symmetric else (rb, cb)
        self._fwd_edges = (rf, cf)      # ... and the forward one's: the rows-only forward restricts it to the edges that enter the loss rows (loss_rows_forward)
        self._support_cache = None
        self._fwd0_cache = None
        self.rows_only_forwards = 0     # how often a training forward evaluated its last layer on the loss rows (tests, bench)
        # rows of room behind an exchanged matrix for the first halo slice (alloc_exchanged): the largest first slice of the two orientations
        # (the level orientations of the row-sparse backward ask for subsets of the reverse one's rows)
        self.halo_room = 0
        self.merged_passes = self.interior_passes = 0      # how often an aggregation took the merged first pass / the separate interior pass (tests, bench)
        if self.overlap and self.f.first is not None:
            self.halo_room = max(self.f.plan.n_halo_slice[0], self.b.plan.n_halo_slice[0], 1)

    # -- build one orientation ---------------------------------------------------------------------------
    def _orient(self, rows, cols, like=None, dst=None, src=None):
        """like: an existing orientation whose KIND of plan (pull / cover) and slice count this one takes over — the callers' control flow is
        keyed to those (row-chunked producers, fused last pass), so a level orientation of the row-sparse backward mirrors the full one.
        dst / src (graph.RowSpace over this rank's rows, overlapped form only): a COMPACT level — the CSRs' rows are positions in dst, the
        interior columns and the send lists positions in src: the level reads a [src.n, d] matrix and writes a [dst.n, d] one."""
        part, lo, hi = self.part, self.part.lo(), self.part.hi()
        o = _Orientation()
        o.E = int(rows.numel())
        n_r = self.N if dst is None else max(dst.n, 1)
        n_c = self.N if src is None else max(src.n, 1)
        if dst is not None:
            rows = dst.pos[rows].to(torch.int64)
            if rows.numel() and int(rows.min()) < 0:
                raise RuntimeError('level orientation: an edge writes a row outside the destination support')
        o.whole = o.interior = o.halo = o.plan = o.first = None
        o.n_cols_local = n_c
        if self.exchange_kind == 'allgather':       # columns index the gathered [P*R, d] matrix (equal blocks: identity)
            o.whole = self.compute.csr(rows, cols, self.N, part.padded)
            return o
        remote = (cols < lo) | (cols >= hi)
        uniq, inv = torch.unique(cols[remote], return_inverse=True)       # ascending ids = grouped by owner
        K = 1
        if like is not None and like.plan is not None:
            K = like.plan.n_slices
        elif self.overlap:
            # every rank must cut its plans into the same number of slices: decide on the largest halo of the group
            if self._n_slices_req is not None:
                K = max(1, int(self._n_slices_req))
            else:
                nh = torch.tensor([int(uniq.numel())], dtype=torch.int64, device=uniq.device)
                if part.world > 1:
                    _all_reduce(nh, op=dist.ReduceOp.MAX, group=self.group)
                K = default_slices(int(nh.item()))
        use_cover = False
        if like is not None and like.plan is not None:
            use_cover = bool(like.plan.cover)
            asg = CoverPlan.assign(rows[remote], cols[remote], part) if use_cover else None
        elif self.cover:
            # one decision for the group (every rank must build the same kind of plan): the cover is taken when it spares the busiest
            # requester at least tuning.T.cover_min_gain of its rows (power-law graphs: 25 - 32 %; the ogbn-products shape: < 6 % -> plain pull,
            # whose pack is a row gather and whose slices follow the owners' row chunks)
            asg = CoverPlan.assign(rows[remote], cols[remote], part)
            # (the busiest requester under either plan — MAX of each count over the ranks —, not the best ratio of any one rank: a rank with
            # nothing to pull in this orientation has no say)
            busiest = torch.tensor([int(asg['rows_pull']), int(asg['rows_cover'])], dtype=torch.int64, device=rows.device)
            if part.world > 1:
                _all_reduce(busiest, op=dist.ReduceOp.MAX, group=self.group)
            n_pull, n_cover = (int(v) for v in busiest.tolist())
            use_cover = (n_pull > 0 and 1.0 - n_cover / n_pull >= T.cover_min_gain) or self._cover_forced
        lp_, ns_ = (src.pos, src.n) if src is not None else (None, None)
        o.plan = (CoverPlan(rows[remote], cols[remote], part, self.group, K, self.compute, asg, local_pos=lp_, n_src=ns_) if use_cover
                  else HaloPlan(uniq, part, self.group, K, local_pos=lp_, n_src=ns_))
        if self.overlap:
            ci = cols[~remote] - lo
            if src is not None:
                ci = HaloPlan._to_pos(ci, src.pos)
            o.interior = self.compute.csr(rows[~remote], ci, n_r, n_c)
            rr, sl, slot = o.plan.halo_edges(rows[remote], inv)
            o.halo = []
            for k in range(K):
                m = sl == k if K > 1 else slice(None)
                o.halo.append(self.compute.csr(rr[m], slot[m], n_r, max(o.plan.n_halo_slice[k], 1)))
            if merged_first_pass_enabled() and part.world > 1:
                # Round 5: the interior pass and the FIRST halo slice as ONE pass over [local rows | slice-0 receive buffer] — when the exchanged
                # matrix was allocated with room behind it (alloc_exchanged), slice 0 is received right there and one CSR with a single base
                # pointer reads both: one pass over the running sums less per aggregation (the rank's own HBM traffic is what bounds P = 8)
                m0 = sl == 0 if K > 1 else slice(None)
                o.first = self.compute.csr(torch.cat([rows[~remote], rr[m0]]), torch.cat([ci, n_c + slot[m0]]), n_r,
                                           n_c + max(o.plan.n_halo_slice[0], 1))
        elif dst is not None or src is not None:
            raise ValueError('compact level orientations need the overlapped exchange')
        else:
            new_col = cols - lo
            new_col[remote] = self.N + inv
            o.whole = self.compute.csr(rows, new_col, self.N, self.N + o.plan.n_halo)
        return o

    def support_orients(self, mask_local, n_aggr, max_edge_frac=None):
        """The level orientations alone (matrices keep all local rows): [lv.orient for lv in support_levels(..., compact=False)]."""
        return [lv.orient for lv in self.support_levels(mask_local, n_aggr, max_edge_frac, compact=False)]

    def support_levels(self, mask_local, n_aggr, max_edge_frac=None, compact=True, max_frac=None, cumulative=False):
        """Levels of a row-sparse backward on row shards (trunk.py; graph.CSRGraph.grad_support_plan is the one-GPU form): reverse aggregation
        j gathers only rows of the support S_j (S_0 = the loss rows `mask_local` of this rank, S_{j+1} = rows with a reverse-orientation
        neighbour in S_j) — all other rows of the gathered matrix are exact zeros.  Level j = the reverse orientation restricted to the edges
        whose gathered row is in S_j: its halo plan asks the peers for the support's rows only (the first backward exchange of the bench's
        graph ships a tenth of the rows, the second 45 %), its interior / halo passes read fewer edges.
        compact (round 5): while the GLOBAL support is at most max_frac (tuning.T.rowsparse_max_frac) of the nodes the level is also COMPACT
        in this rank's rows — SupportLevel.src / .dst are graph.RowSpace objects over the rank's block (S_j / S_{j+1} restricted to it), the
        level reads a [src.n, d] matrix and writes a [dst.n, d] one (dst None: all local rows; the last level always), so the rank's store
        backward, weight gradient and GEMM tail run on the support's rows only, as on one GPU.  One decision per level for the group; needs
        the overlapped exchange with a plan whose last pass can be the aggregation + GEMM kernel (cover, or one slice).
        cumulative (the 'Residual' trunk: a layer's store backward also takes the gradient of the layer above): S_{j+1} = N(S_j) ∪ S_j.
        Levels are built while they keep at most max_edge_frac of the edges; the supports travel as byte maps (all-gather of N bytes per
        level, once per mask).  Returns a list of SupportLevel, possibly empty."""
        from .graph import RowSpace
        max_edge_frac = T.support_max_edge_frac if max_edge_frac is None else max_edge_frac
        max_frac = T.rowsparse_max_frac if max_frac is None else max_frac
        key = (mask_local.data_ptr(), mask_local._version, int(n_aggr), bool(compact), float(max_frac), float(max_edge_frac), bool(cumulative))
        if self._support_cache is not None and self._support_cache[0] == key and self._support_cache[1] is mask_local:
            return self._support_cache[2]
        part, P = self.part, self.part.world
        rows, cols = self._rev_edges
        dev = rows.device
        owner = part.owner(cols)
        lo_t = torch.tensor(part.bounds[:-1], dtype=torch.int64, device=dev)
        slot = owner * part.R + (cols - lo_t[owner])          # position of a column's row in the all-gathered [P * R] map
        plan_b = self.b.plan
        compact = bool(compact) and self.overlap and plan_b is not None and (plan_b.cover or plan_b.n_slices == 1) and min(
            part.hi(q) - part.lo(q) for q in range(P)) > 0

        def space_of(mask):
            if not bool(mask.any()):      # (a superset of the support is as good: a row outside it carries exact zeros)
                mask = mask.clone()
                mask[0] = True
            pos = torch.cumsum(mask, 0, dtype=torch.int32) - 1
            idx = torch.nonzero(mask).flatten()
            return RowSpace(idx, torch.where(mask, pos, torch.full_like(pos, -1)), self.norm_out[idx].contiguous())

        def global_count(mask):
            c = torch.tensor([int(mask.sum())], dtype=torch.int64, device=dev)
            if P > 1:
                _all_reduce(c, group=self.group)
            return int(c.item())
        # pass 1: the supports and the edges each level keeps (collectives: one all-gather + one all-reduce per level, the same on every rank)
        found, s_local = [], mask_local.to(torch.bool)
        n0 = global_count(s_local) if compact else 0
        for _ in range(int(n_aggr)):
            m = torch.zeros(part.R, dtype=torch.uint8, device=dev)
            m[:self.N] = s_local.to(torch.uint8)
            allm = torch.empty(part.padded, dtype=torch.uint8, device=dev)
            if P > 1:
                _all_gather_into_tensor(allm, m, group=self.group)
            else:
                allm.copy_(m)
            keep = allm[slot] != 0
            cnt = torch.tensor([int(keep.sum())], dtype=torch.int64, device=dev)
            if P > 1:
                _all_reduce(cnt, group=self.group)
            if int(cnt.item()) > max_edge_frac * self.E_global:
                break
            r_k, c_k = rows[keep], cols[keep]
            s_next = torch.bincount(r_k, minlength=self.N)[:self.N] > 0
            if cumulative:
                s_next = s_next | s_local
            found.append((r_k, c_k, s_next, global_count(s_next) if compact else 0))
            s_local = s_next
        # pass 2: the orientations.  A level writes a COMPACT matrix only if the next level exists to read it (and its support is small)
        levels = []
        src = space_of(mask_local.to(torch.bool)) if (compact and found and n0 <= T.rowsparse_s0_limit * self.N_global) else None
        for j, (r_k, c_k, s_next, n_next) in enumerate(found):
            dst = None
            if src is not None and j + 1 < len(found) and j + 1 < int(n_aggr) and n_next <= max_frac * self.N_global:
                dst = space_of(s_next)
            levels.append(SupportLevel(self._orient(r_k, c_k, like=self.b, dst=dst, src=src), src, dst))
            src = dst
        self._support_cache = (key, mask_local, levels)
        return levels

    def loss_rows_forward(self, levels):
        """Rows-only forward on row shards (trunk.py): the FORWARD orientation restricted to the edges that enter this rank's loss rows — the compact
        row space levels[0].src of support_levels — as a level orientation: it writes a [src.n, d] matrix, reads all local rows of the gathered matrix,
        and its halo plan asks the peers only for the rows that are in-neighbours of those loss rows (a tenth of the forward exchange under a 10 % mask).
        Same kind of plan and slice count as the full forward orientation.  Built once per mask (every rank at the same point of the forward)."""
        if self._fwd0_cache is not None and self._fwd0_cache[0] is levels:
            return self._fwd0_cache[1]
        s0 = levels[0].src
        rf, cf = self._fwd_edges
        keep = s0.pos[rf] >= 0
        o = self._orient(rf[keep], cf[keep], like=self.f, dst=s0, src=None)
        self._fwd0_cache = (levels, o)
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
    def exchange(self, x_local, transpose=False, orient=None):
        """Blocking form: [n_local, d] -> the matrix the single-pass CSR reads ([local | halo] or the gathered [P*R, d])."""
        o = orient if orient is not None else (self.b if transpose else self.f)
        if self.exchange_kind == 'allgather':
            return gather_rows(x_local, self.part, self.group)
        if self.part.world == 1:
            return x_local
        plan = o.plan
        ext = torch.empty((plan.n_local + plan.n_halo, x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
        ext[:plan.n_local] = x_local
        if self.wire == 'bf16' and x_local.dtype == torch.float32:
            send = self.compute.pack_rows(x_local, plan.send_idx_all, 'bf16')
            wire_in = torch.empty((plan.n_halo, x_local.shape[1]), dtype=torch.bfloat16, device=x_local.device)
            _all_to_all_single(wire_in.view(torch.uint8), send.view(torch.uint8), plan.recv_counts_all, plan.send_counts_all, group=self.group)
            ext[plan.n_local:] = wire_in          # the single-pass CSR reads ONE fp32 matrix: widened here (the overlapped form never widens)
            return ext
        send = self.compute.pack_rows(x_local, plan.send_idx_all)
        _all_to_all_single(ext[plan.n_local:], send, plan.recv_counts_all, plan.send_counts_all, group=self.group)
        return ext

    def _send_slice(self, x_local, plan, k, recv=None):
        """pack slice k -> asynchronous all-to-all.  Returns (receive buffer, work handle, send buffer kept alive).  recv: receive into this
        [>= n_k, d] buffer (the room behind the exchanged matrix: merged first pass)."""
        bf16 = self.wire == 'bf16' and x_local.dtype == torch.float32
        send = plan.pack(self.compute, x_local, k, 'bf16' if bf16 else 'f32')
        n_k = plan.n_halo_slice[k]
        if recv is None or recv.dtype != send.dtype:
            recv = torch.empty((max(n_k, 1), x_local.shape[1]), dtype=send.dtype, device=x_local.device)
        if send.dtype == torch.bfloat16:       # moves as bytes: not every backend knows bfloat16
            work = _all_to_all_single(recv[:n_k].view(torch.uint8), send.view(torch.uint8), plan.recv_counts[k], plan.send_counts[k],
                                      group=self.group, async_op=True)
        else:
            work = _all_to_all_single(recv[:n_k], send, plan.recv_counts[k], plan.send_counts[k], group=self.group, async_op=True)
        return recv, work, send

    def start_halo(self, x_local, transpose=False, produce=None, orient=None):
        """Overlapped form, first half: for every slice k — produce(k, r0, r1) fills local rows [r0, r1) of x_local (if given: the
        layer GEMM / the trunk backward of row chunk k), then pack + asynchronous all-to-all of slice k, whose rows all lie in
        that chunk.  Returns the list of (receive buffer, work handle, send buffer) per slice."""
        o = orient if orient is not None else (self.b if transpose else self.f)
        plan = o.plan
        flights = []
        if self.exchange_log is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._xev0 = ev
        # merged first pass: slice 0 lands in the room behind x_local (alloc_exchanged), [local rows | slice 0] is then ONE matrix for o.first
        ext = None
        n0 = plan.n_halo_slice[0]
        if (o.first is not None and getattr(x_local, '_cb_room', 0) >= max(n0, 1) and x_local.dtype == torch.float32 and x_local.is_contiguous()
                and x_local.shape[0] == o.n_cols_local and not (self.wire == 'bf16')):
            d = x_local.shape[1]
            ext = torch.as_strided(x_local, (x_local.shape[0] + max(n0, 1), d), (d, 1), x_local.storage_offset())
        for k in range(plan.n_slices):
            if produce is not None:
                produce(k, *plan.chunks[k])
            flights.append(self._send_slice(x_local, plan, k, recv=ext[x_local.shape[0]:] if (ext is not None and k == 0) else None))
        if ext is not None:
            flights[0] = flights[0] + (ext,)          # (recv, work, send, [local | slice 0])
        return flights

    def aggregate_start(self, h_local, transpose=False, produce=None, orient=None):
        """First half of aggregate(): starts the exchange (overlapped form) and returns a handle for aggregate_finish().  Work
        issued between the two calls (e.g. the previous layer's weight-gradient GEMM in the trunk backward) runs under the
        exchange.  produce: see start_halo (h_local is then an allocated, not yet filled matrix).  orient: a level orientation of
        support_orients — the handle carries it to aggregate_finish."""
        if not self.overlap or h_local.dtype != torch.float32:
            if produce is not None:
                produce(0, 0, h_local.shape[0])
            return (h_local, None, orient)
        return (h_local, self.start_halo(h_local, transpose, produce, orient), orient)

    def finish_halo(self, flights, o, part_sums, last_pass, x_local=None):
        """Second half: the interior pass (part_sums None: computed here from x_local — or, when slice 0 was received behind x_local, the
        merged pass over [local | slice 0] once it has landed), then halo pass k (raw sums, in place) as slice k arrives;
        last_pass(csr, recv, acc) is the caller's final pass (it applies the epilogue and always runs, also over an empty last slice; acc is
        None when the merged pass IS the last one: a single-slice plan)."""
        c = self.compute
        K = len(flights)
        out = None
        merged = len(flights[0]) > 3 and part_sums is None
        if part_sums is None and not merged:
            self.interior_passes += 1
            part_sums = c.spmm(o.interior, x_local, profile=self.profile)       # raw sums over the local columns, overlaps the exchange
        for k, fl in enumerate(flights):
            recv, work = fl[0], fl[1]
            work.wait()
            if merged and k == 0:
                self.merged_passes += 1
                if self.exchange_log is not None and K == 1:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.exchange_log.append((self._xev0, ev))
                if K == 1:
                    return last_pass(o.first, fl[3], None)
                part_sums = c.spmm(o.first, fl[3], profile=self.profile)
                continue
            if k == K - 1:
                if self.exchange_log is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.exchange_log.append((self._xev0, ev))
                out = last_pass(o.halo[k], recv, part_sums)
            elif o.halo[k].E:
                c.spmm(o.halo[k], recv, acc_init=part_sums, out=part_sums, profile=self.profile)
        return out

    def aggregate_finish(self, handle, transpose=False, row_scale=None, bias=None, relu=False, last_pass=None):
        """last_pass(csr, recv, acc) (overlapped form only): the caller's own final pass over the last slice on top of the running sums
        (the fused trunk: aggregation + GEMM kernel, cb_spmm_gemm_f32 with acc_init) instead of the plain epilogue pass."""
        h_local, flights, orient = handle if len(handle) == 3 else (*handle, None)
        o = orient if orient is not None else (self.b if transpose else self.f)
        c = self.compute
        if flights is None:
            if last_pass is not None:
                raise ValueError('aggregate_finish: last_pass needs the overlapped exchange')
            return c.spmm(o.whole if o.whole is not None else self._whole(o), self.exchange(h_local, transpose, orient), row_scale, bias, relu,
                          profile=self.profile)
        if last_pass is None:
            last_pass = lambda g, recv, acc: c.spmm(g, recv, row_scale, bias, relu, acc_init=acc, profile=self.profile)      # noqa: E731
        return self.finish_halo(flights, o, None, last_pass, x_local=h_local)

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


def allreduce_grads(params, group=None, guard=None):
    """Sum the gradients of the replicated parameters in one flat bucket (weights are a few hundred KB).  guard (int32 [1] device word,
    _lib.grad_guard): rides in the same bucket, so that a failed gradient check on ONE rank stops the optimiser launch on EVERY rank."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads] + ([guard.to(grads[0].dtype)] if guard is not None else []))
    _all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    if guard is not None:
        guard.copy_((flat[off:off + 1] != 0).to(guard.dtype))


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
    per_node = [(n, p) for n, p in model.named_parameters() if _per_node(n)]
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


def scatter_rows(t_root, part, group=None, width=None, dtype=None, device=None):
    """Rank 0 holds a [N, ...] tensor ordered by row; every rank receives its row block [lo, hi).  One all-to-all whose only
    non-empty source is rank 0 (uneven splits), so no rank but the loader ever holds more than its block."""
    P, r = part.world, part.rank
    n_local = part.n_local
    if r == 0:
        inp = t_root.contiguous()
        shape_tail, dtype, device = tuple(inp.shape[1:]), inp.dtype, inp.device
        in_splits = [part.hi(q) - part.lo(q) for q in range(P)]
    else:
        shape_tail = tuple(width) if width is not None else ()
        inp = torch.empty((0,) + shape_tail, dtype=dtype, device=device)
        in_splits = [0] * P
    out = torch.empty((n_local,) + shape_tail, dtype=dtype, device=device)
    if P == 1:
        out.copy_(inp)
        return out
    _all_to_all_single(out, inp, [n_local] + [0] * (P - 1), in_splits, group=group)
    return out


def scatter_by_owner(pairs_root, owner_root, counts, part, group=None, device=None):
    """Rank 0 holds [E, 2] int64 pairs and the owning rank of each; rank q receives its `counts[q]` pairs (input order kept)."""
    P, r = part.world, part.rank
    if r == 0:
        order = torch.sort(owner_root, stable=True)[1]
        inp = pairs_root[order].contiguous()
        in_splits = [int(c) for c in counts]
        device = inp.device
    else:
        inp = torch.empty((0, 2), dtype=torch.int64, device=device)
        in_splits = [0] * P
    out = torch.empty((int(counts[r]), 2), dtype=torch.int64, device=device)
    if P == 1:
        out.copy_(inp)
        return out
    _all_to_all_single(out, inp, [int(counts[r])] + [0] * (P - 1), in_splits, group=group)
    return out


class ShardedTrainer:
    """Multi-GPU driver of the TeacherGNN path: bench.py (train_step) and `torchrun ... main.py` (main -> train_teacherGNN: the
    reference's epoch — run_trainSet incl. the head/tail metrics forward, run_testSet, records, checkpoints — on row shards,
    with hit counts / loss shares all-reduced).  Mirrors trainer.train_step / run_trainSet / run_testSet / train_teacherGNN
    (trainer_node_classification.py:303-369,382-432,453-495).

    Loading: rank 0 alone loads (data/<dataset>.pt) or generates the graph, runs the head/tail analysis and scatters every rank's
    row block of x / y / masks / metric sets and its two edge blocks (edges by destination owner and by source owner); the other
    ranks never hold more than their block.  `data` (rank 0; None elsewhere) injects a prepared Data object instead."""

    NODE_COLS = ('y', 'train', 'test', 'large', 'small', 'zero')

    def __init__(self, args, which_run, group=None, data=None):
        from . import optim as cb_optim
        from .utils import save_graph_analyze, set_arch_configs
        self.args, self.group, self.which_run = args, group, which_run
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(f'cuda:{args.cuda_num}')
        args.device = self.device
        self.bag = {}
        dev = self.device
        exchange = os.environ.get('COLDBREW_EXCHANGE', 'halo')
        kind = os.environ.get('COLDBREW_PARTITION', 'rows' if exchange == 'allgather' else 'edges')
        want_sets = bool(getattr(args, 'want_headtail', 0))
        head = [None]
        if self.rank == 0:
            if data is None:
                from .data import load_data
                with contextlib.redirect_stdout(io.StringIO()):
                    data = load_data(args.dataset, which_run, self)
                data.x = data.x.float()
                if want_sets or (getattr(args, 'do_deg_analyze', 0) and getattr(data, 'large_deg_idx', None) is None):
                    with contextlib.redirect_stdout(io.StringIO()):
                        save_graph_analyze(int(data.x.shape[0]), data, args.use_special_split, verbose=False)
            n = int(data.x.shape[0])
            ei = data.edge_index.to(torch.int64)
            bad = int(((ei < 0) | (ei >= n)).sum()) if ei.numel() else 0
            if bad:
                raise ValueError(f'edge_index has {bad} endpoints outside [0, {n})')
            test_mask = data.test_mask if getattr(data, 'test_mask', None) is not None else ~data.train_mask
            in_deg = torch.bincount(ei[1], minlength=n)
            part0 = Partition.balanced(in_deg, self.world, 0) if kind == 'edges' else Partition(n, self.world, 0)
            own_d, own_s = part0.owner(ei[1]), part0.owner(ei[0])
            cols = torch.zeros((n, len(self.NODE_COLS)), dtype=torch.int64, device=ei.device)
            cols[:, 0], cols[:, 1], cols[:, 2] = data.y.to(torch.int64), data.train_mask.to(torch.int64), test_mask.to(torch.int64)
            has = []
            for j, name in ((3, 'large_deg_idx'), (4, 'small_deg_idx'), (5, 'zero_deg_idx')):
                idx = getattr(data, name, None)
                if idx is not None:
                    idx = idx if torch.is_tensor(idx) else torch.as_tensor(__import__('numpy').asarray(idx))
                    cols[idx.to(device=ei.device, dtype=torch.long).reshape(-1), j] = 1
                    has.append(name)
            head[0] = {'n': n, 'e': int(ei.shape[1]), 'f': int(data.x.shape[1]), 'bounds': part0.bounds, 'kind': part0.kind,
                       'n_train': int(data.train_mask.sum()), 'n_test': int(test_mask.sum()), 'sets': has,
                       'cnt_d': torch.bincount(own_d, minlength=self.world).tolist(), 'cnt_s': torch.bincount(own_s, minlength=self.world).tolist()}
        if self.world > 1:
            dist.broadcast_object_list(head, src=0, group=group)
        h = head[0]
        self._n, self._e, self.n_train, self.n_test = h['n'], h['e'], h['n_train'], h['n_test']
        self.part = Partition(self._n, self.world, self.rank, h['bounds'], kind=h['kind'])
        root = self.rank == 0
        ef = scatter_by_owner(ei.t() if root else None, own_d if root else None, h['cnt_d'], self.part, group, dev)
        eb = scatter_by_owner(ei.t() if root else None, own_s if root else None, h['cnt_s'], self.part, group, dev)
        self.x = scatter_rows(data.x.float() if root else None, self.part, group, (h['f'],), torch.float32, dev)
        nc = scatter_rows(cols if root else None, self.part, group, (len(self.NODE_COLS),), torch.int64, dev)
        if root:
            del ei, own_d, own_s, cols, in_deg
        data = None
        self.y = nc[:, 0].contiguous()
        self.train_mask, self.test_mask = nc[:, 1].bool().contiguous(), nc[:, 2].bool().contiguous()
        # metric groups (large / small / zero degree): the local NON-training members, as eval_headtail__traintest_v2 keeps them
        self.sets = {name: torch.where(nc[:, 3 + j].bool() & ~self.train_mask)[0]
                     for j, name in enumerate(('large_deg_idx', 'small_deg_idx', 'zero_deg_idx')) if name in h['sets']}
        del nc
        self.sgraph = ShardedGraph(None, self._n, self.part, group, exchange=exchange,
                                   overlap=os.environ.get('COLDBREW_OVERLAP', '1') != '0' and getattr(args, 'agg_dtype', 'f32') == 'f32',
                                   wire=os.environ.get('COLDBREW_HALO_WIRE', 'f32'), local_edges=(ef.t(), eb.t()))
        del ef, eb
        self.edge_index = torch.zeros((2, 1), dtype=torch.int64, device=dev)     # placeholder: the cached sharded graph is injected below
        torch.cuda.empty_cache()
        self.optfun = cb_optim.resolve(args.optfun)
        set_arch_configs(args)
        args.N_nodes_global = self._n
        args.N_nodes = self.part.n_local             # structural-embedding tables are row-sharded
        self.modeldir = f'saved_models/{args.task}/{args.dataset}'
        self.resdir = f'{args.task}/{args.dataset}'
        if self.rank == 0:
            os.makedirs(self.modeldir, exist_ok=True)

    def setup_teacherGNN(self):
        from .GNN_model.GNN_normalizations import TeacherGNN
        from .utils import getMLP
        torch.manual_seed(self.args.random_seed)
        with contextlib.redirect_stdout(io.StringIO()):
            self.proj2class = getMLP(self.args.TeacherGNN.neurons_proj2class).to(self.device) if getattr(self.args, 'has_proj2class', 0) else None
            self.teacherGNN = TeacherGNN(self.args, self.proj2class).to(self.device)
        sync_initial_state(self.teacherGNN, self.part, self.group, self.args.random_seed)
        self.teacherGNN.model.model.dglgraph = self.sgraph
        self.replicated = [p for n, p in self.teacherGNN.named_parameters() if not _per_node(n)]
        self.optimizer = self.optfun(self.teacherGNN.parameters(), lr=self.args.lr, weight_decay=self.args.weight_decay)

    def load_full_state_dict(self, sd_full):
        """Loads a single-GPU (unsharded) TeacherGNN state_dict: replicated tensors as they are, the per-node tables
        (structural embeddings `le`, learnable inputs `embs`) cut to this rank's row block."""
        lo, hi = self.part.lo(), self.part.hi()
        sd = {}
        for k, v in sd_full.items():
            per_node = _per_node(k) and v.dim() == 2 and v.shape[0] == self._n
            sd[k] = v[lo:hi].clone() if per_node else v
        self.teacherGNN.load_state_dict(sd)

    def full_state_dict(self):
        """The unsharded state_dict on rank 0 (None elsewhere): per-node tables gathered block by block to rank 0's HOST memory
        (the tables of all ranks together may not fit one device next to its own shard), everything else as rank 0 holds it."""
        out = {} if self.rank == 0 else None
        for k, v in self.teacherGNN.state_dict().items():
            if not (_per_node(k) and v.dim() == 2):
                if self.rank == 0:
                    out[k] = v.detach().cpu()
                continue
            full = torch.empty((self._n, v.shape[1]), dtype=v.dtype) if self.rank == 0 else None
            for q in range(self.world):
                rows = self.part.hi(q) - self.part.lo(q)
                if q == 0:
                    if self.rank == 0:
                        full[:rows] = v.detach().cpu()
                    continue
                if self.rank == q:
                    _send_to(v.detach().contiguous(), 0, self.group)
                elif self.rank == 0:
                    full[self.part.lo(q):self.part.hi(q)] = _recv_from((rows, v.shape[1]), v.dtype, v.device, q, self.group).cpu()
            if self.rank == 0:
                out[k] = full
        return out

    def graph(self):
        return self.sgraph

    def global_nodes(self):
        return self._n

    def global_edges(self):
        return self._e

    # -- sharded resumable checkpoint ("sharded `le` tables"; utils.py:958-986 saves weights only) -------------
    def checkpoint_path(self, rank=None):
        r = self.rank if rank is None else rank
        return os.path.join(self.modeldir, f'teacherGNN-ckpt.shard{r}of{self.world}')

    def save_checkpoint(self, epoch, results, best_test_acc=0.):
        """One file per rank: its rows of the per-node tables with their Adam moments; rank 0's file also carries the replicated
        weights (once), their moments, the records and the RNG states.  Tensors and plain containers only (weights_only=True)."""
        import numpy as np
        names = {id(p): n for n, p in self.teacherGNN.named_parameters()}
        sd = self.teacherGNN.state_dict()
        keep = (lambda k: True) if self.rank == 0 else _per_node
        opt = {}
        for p_, st in self.optimizer.state.items():
            n = names.get(id(p_))
            if n is not None and keep(n):
                opt[n] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        blob = {'model': {k: v for k, v in sd.items() if keep(k)}, 'optimizer': opt, 'epoch': int(epoch), 'world': self.world,
                'bounds': list(self.part.bounds), 'torch_rng': torch.get_rng_state()}
        if self.rank == 0:
            np_state = np.random.get_state()
            blob.update({'results': [[float(v) for v in row] for row in results], 'best_test_acc': float(best_test_acc),
                         'numpy_rng': {'kind': str(np_state[0]), 'keys': torch.from_numpy(np_state[1].astype(np.int64)),
                                       'pos': int(np_state[2]), 'has_gauss': int(np_state[3]), 'cached_gaussian': float(np_state[4])}})
        torch.save(blob, self.checkpoint_path())
        if self.world > 1:
            dist.barrier(group=self.group)

    def load_checkpoint(self):
        """(last epoch, records, best accuracy) or (-1, [], 0.) without a complete set of shard files for THIS world / partition."""
        import numpy as np
        ok = torch.tensor([int(os.path.exists(self.checkpoint_path()) and os.path.exists(self.checkpoint_path(0)))], dtype=torch.int64)
        if self.world > 1:
            okd = ok.to(self.device)
            _all_reduce(okd, op=dist.ReduceOp.MIN, group=self.group)
            ok = okd.cpu()
        if not int(ok):
            return -1, [], 0.
        mine = torch.load(self.checkpoint_path(), map_location='cpu', weights_only=True)
        root = mine if self.rank == 0 else torch.load(self.checkpoint_path(0), map_location='cpu', weights_only=True)
        if mine['world'] != self.world or list(mine['bounds']) != list(self.part.bounds):
            raise RuntimeError(f"checkpoint was written for world {mine['world']} / bounds {mine['bounds']}, this run has world {self.world} / "
                               f'{self.part.bounds}: re-shard through the single-file model (teacherGNN) instead')
        sd = {k: v for k, v in root['model'].items() if not _per_node(k)}
        sd.update({k: v for k, v in mine['model'].items() if _per_node(k)})
        self.teacherGNN.load_state_dict(sd)
        by_name = dict(self.teacherGNN.named_parameters())
        for n, p_ in by_name.items():
            st = (mine if _per_node(n) else root)['optimizer'].get(n)
            if st is not None:
                self.optimizer.state[p_] = {k: (v.to(p_.device) if torch.is_tensor(v) else v) for k, v in st.items()}
        torch.set_rng_state(mine['torch_rng'])
        if self.rank == 0:
            r = root['numpy_rng']
            np.random.set_state((r['kind'], r['keys'].numpy().astype(np.uint32), r['pos'], r['has_gauss'], r['cached_gaussian']))
            print(f'---››››  RESUME from {self.checkpoint_path()} (+{self.world - 1} shard files) after epoch {root["epoch"]}')
        return root['epoch'], root['results'], root.get('best_test_acc', 0.)

    def save_model(self, name):
        """The reference's artifact (utils.py:958-960: one state_dict file) from the shards, written by rank 0 — loads into the
        single-GPU trainer / the student stage with utils.load_model."""
        sd = self.full_state_dict()
        if self.rank == 0:
            path = os.path.join(self.modeldir, name)
            torch.save(sd, path)
            print(f'‹‹‹‹‹‹‹---  Saved @ :{path}')
        if self.world > 1:
            dist.barrier(group=self.group)

    # -- one optimisation step ---------------------------------------------------------------------------------------------
    def training_loss(self):
        from . import ops
        m = self.teacherGNN
        # (loss_rows: this rank's rows of the train mask — the objective touches the logits there only; rows_only: and reads nothing else of this forward's
        # output — the other rows come back as NaN —, trainer_node_classification.training_loss; `rows_only_forward = False` withdraws the second promise)
        if getattr(self.args, 'has_loss_component_edgewise', False):
            raise NotImplementedError('edge-wise (link-prediction) loss belongs to the I2_GTL mode (out of scope); training_loss() promises the model '
                                      'that only the train rows of its output are read')
        rows_only = bool(getattr(self, 'rows_only_forward', getattr(self.args, 'rows_only_forward', True)))
        out = m.get_3_embs(self.x, self.edge_index, loss_rows=(self.train_mask, self.n_train), rows_only=rows_only).emb4classi_full
        # local numerator / global count; the global loss is the sum over ranks
        unit = float(self.args.TeacherGNN.lossa_semantic) == 1.0
        loss = ops.nll_logsoftmax(out, self.y, self.train_mask, self.n_train, unit_grad=unit)
        if not unit:
            loss = loss * self.args.TeacherGNN.lossa_semantic
        if m.se_reg_all is not None:
            folded = ops.fold_se_reg(m, self.optimizer, self.args.se_reg, m.se_reg_all)
            # se_reg_all is already global; count it once
            loss = loss + (folded if folded is not None else self.args.se_reg * m.se_reg_all) / self.world
        return loss

    def train_step(self, metrics=False):
        from . import norms_hip
        m = self.teacherGNN
        m.train()
        with norms_hip.row_sharding(self.group, self._n):      # column statistics of the norm tricks span all ranks
            loss = self.training_loss()
            if metrics:
                self._headtail_metrics()
            self.optimizer.zero_grad()
            loss.backward()
        allreduce_grads(self.replicated, self.group, guard=_lib.grad_guard(self.device))
        self.optimizer.step()
        total = loss.detach().clone()
        _all_reduce(total, group=self.group)
        return total

    # -- the reference's epoch on row shards (trainer_node_classification.py:303-369,382-432,453-495,672-687) -----------------
    def run_trainSet(self):
        """loss (global), 0, 0 — with --want_headtail=1 the second train-mode forward of :397-413 runs between the loss forward and
        the backward, as in the reference, and fills bag['head_tail_iso']."""
        loss = float(self.train_step(metrics=True))
        _lib.device_status()            # (the loss has just been read: a device-side error of this step is raised here, never trained through)
        return loss, 0, 0

    def _headtail_metrics(self):
        """bag['head_tail_iso'] (:397-413): accuracy x 100 of a second train-mode (dropout-active) forward on the non-training
        members of the large- / small- / zero-degree groups; per-rank hit counts and group sizes are all-reduced, the rounding is
        cal_acc_rounded100's (float32, 3 decimals)."""
        import numpy as np
        result = []
        if getattr(self.args, 'want_headtail', 0):
            names = ['large_deg_idx', 'small_deg_idx'] + (['zero_deg_idx'] if self.args.use_special_split else [])
            missing = [n for n in names if n not in self.sets]
            if missing:
                raise RuntimeError(f'--want_headtail=1 needs the degree groups {missing} (run with --do_deg_analyze=1 or provide them in the data)')
            with torch.no_grad():        # metrics only: the reference tracks this forward in autograd and never uses its graph
                logits = self.teacherGNN.get_3_embs(self.x, self.edge_index).emb4classi_full
            pred = torch.max(logits, dim=1)[1]
            cnt = torch.stack([torch.stack([(pred[self.sets[n]] == self.y[self.sets[n]]).sum(),
                                            torch.tensor(self.sets[n].numel(), device=pred.device)]) for n in names]).to(torch.float64)
            _all_reduce(cnt, group=self.group)
            with np.errstate(invalid='ignore', divide='ignore'):
                for hits, size in cnt.tolist():
                    result.append(np.round(np.float32(np.float32(hits) / np.float32(size)) * np.float32(100), 3))
        self.bag['head_tail_iso'] = result

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
        h_train, h_test = cnt.tolist()
        return h_train * 1.0 / max(self.n_train, 1), float('nan'), h_test * 1.0 / max(self.n_test, 1), float('nan')

    def train_teacherGNN(self):
        """Epoch loop with the record layout and return rows of trainer.train_teacherGNN (:303-369): per epoch
        [log(loss), acc_train*100, acc_test*100, 0, 0 (, head, tail, iso)]; returns rows [2] or [2,-3,-2,-1].  --resume / --ckpt_every
        use the sharded checkpoint; the final model is also written as the reference's single-file artifact.  Every rank returns the
        same array."""
        import numpy as np
        from . import norms_hip
        self.setup_teacherGNN()
        results, best_test_acc, first_epoch = [], 0., 0
        if getattr(self.args, 'resume', False):
            last, results, best_test_acc = self.load_checkpoint()
            first_epoch = last + 1
        ckpt_every = int(getattr(self.args, 'ckpt_every', 0) or 0)
        if int(getattr(self.args, 'hip_graph', 0) or 0) and self.rank == 0:
            print('--hip_graph=1 ignored by the node-sharded trainer: the collectives of a step are not captured (eager launches)')
        for epoch in range(first_epoch, self.args.epochs):
            self.epoch = epoch
            with norms_hip.row_sharding(self.group, self._n):
                loss, _, _ = self.run_trainSet()
            acc_train, _, acc_test, _ = self.run_testSet()
            if 'SEMLP' in self.args.train_which and acc_test > best_test_acc:
                best_test_acc = acc_test
                self.save_model('best-teacherGNN')
            results.append([float(np.log(loss)), acc_train * 100, acc_test * 100, 0, 0])
            if self.args.want_headtail:
                results[-1].extend(float(v) for v in self.bag['head_tail_iso'])
            if epoch % 20 == 0 and self.rank == 0:
                print(f'Ep{epoch:03d}, acc @ train/test: {acc_train * 100:.1f}, {acc_test * 100:.1f} ')
            if ckpt_every and (epoch + 1) % ckpt_every == 0:
                self.save_checkpoint(epoch, results, best_test_acc)
        if first_epoch < self.args.epochs:
            self.save_checkpoint(self.args.epochs - 1, results, best_test_acc)
        self.save_model('teacherGNN')
        arr = np.array(results).T
        if not self.args.want_headtail:
            return arr[[2]]
        return arr[[2, -3, -2, -1]]

    def main(self):
        if self.args.train_which != 'TeacherGNN':
            raise NotImplementedError('the node-sharded trainer runs --train_which=TeacherGNN')
        return self.train_teacherGNN()


def _per_node(name):
    return name.endswith('.le') or name == 'embs'


def _send_to(t, dst, group=None):
    if _staged(t, group):
        dist.send(t.cpu(), dst, group=group)
    else:
        dist.send(t, dst, group=group)


def _recv_from(shape, dtype, device, src, group=None):
    t = torch.empty(shape, dtype=dtype, device=device)
    if _staged(t, group):
        h = torch.empty(shape, dtype=dtype)
        dist.recv(h, src, group=group)
        t.copy_(h)
    else:
        dist.recv(t, src, group=group)
    return t
