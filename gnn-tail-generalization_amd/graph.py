"""Device-resident graph for the aggregation — the object TricksComb caches in place of the
reference's `dgl.graph((src_list, dst_list))` (GNN_model/GCN.py:92-95).

Holds both CSR orientations (int32), the two degree-norm vectors (GCN.py:206-208,243-245)
and the hub plan, all built once on the GPU through the C ABI (include/coldbrew_hip.h).
The DGL-like query surface that GCNConv.forward touches is kept: in_degrees(),
out_degrees(), number_of_edges(), number_of_nodes().
"""
import torch

from . import _lib
from .tuning import T

# (thresholds: tuning.T — hub_threshold, hot_bytes, fwd0_rows_per_edge, fwd0_min_edges — read at call time: hub_threshold=None means T.hub_threshold)
INT32_EDGE_LIMIT = 2 ** 31 - 1   # edge offsets (rowptr) and column ids are int32 on the device (include/coldbrew_hip.h); see CSRGraph.__init__


def prof_rec(ev0, ev1, g, kind, agg_bytes, store_bytes=0, tail_bytes=0):
    """One timed aggregation launch for bench.py's roofline: HIP events around it, the launch's SURVEY.md 8(d) bytes, the bytes of the
    streams a fused store / dense tail adds (kept apart), the edges the launch walked (g.E of the CSR it ran on) and the kernel family."""
    return {'ev': (ev0, ev1), 'agg': int(agg_bytes), 'store': int(store_bytes), 'tail': int(tail_bytes), 'edges': int(g.E), 'rows': int(g.N), 'kind': kind}


class ZeroInDegreeError(RuntimeError):
    """Stands where the reference raises dgl.base.DGLError (GCN.py:187-197)."""


DGLError = ZeroInDegreeError


class _Plan:
    __slots__ = ('n_hubs', 'n_chunks', 'hub_rows', 'hub_chunk_ptr')


class CSRGraph:
    def __init__(self, edge_index, num_nodes=None, hub_threshold=None, keep_edge_order=False):
        """keep_edge_order: keep the int64 edge list (16 bytes per edge) so that edge_perm() can map CSR positions back to columns of
        edge_index — needed only by the `edge_weight` form of GCNConv.forward (GCN.py:199-202), which TricksComb never uses; the cached graph
        of the training path is built without it (nothing but the CSR outlives the ingest)."""
        lib = _lib.load()
        _lib.require_device(edge_index)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f'edge_index must be [2, E], got {tuple(edge_index.shape)}')
        ei = edge_index.to(torch.int64).contiguous()      # may arrive as a transposed view (utils.py:745)
        dev = ei.device
        E = ei.shape[1]
        if E >= INT32_EDGE_LIMIT:
            # int32 indices, int64 row pointers from 2^31 edges on — that form is SegmentedCSRGraph (build_graph picks it)
            raise ValueError(f'edge_index has {E} columns: CSRGraph indexes edges with int32 (< 2^31); use graph.build_graph / '
                             f'SegmentedCSRGraph (int64 row pointers), or shard the graph over ranks (torchrun --nproc-per-node P main.py ...)')
        if num_nodes is None:                               # DGL infers max id + 1 (GCN.py:94)
            num_nodes = int(ei.max().item()) + 1 if E else 0
        N = int(num_nodes)
        self.N, self.E, self.device = N, E, dev
        self.n_cols, self.row_offset = N, 0
        self._ei, self._perm, self._perm_t = (ei.clone() if keep_edge_order and ei.data_ptr() == edge_index.data_ptr() else ei) if keep_edge_order else None, None, None
        self.hub_threshold = int(T.hub_threshold if hub_threshold is None else hub_threshold)
        self.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.rowptr_t = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.col_t = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        flags = torch.empty(4, dtype=torch.int32, device=dev)
        ws_bytes = lib.cb_csr_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.cb_csr_from_coo_i64(_lib.ptr(ei[0]), _lib.ptr(ei[1]), E, N, _lib.ptr(self.rowptr),
                                               _lib.ptr(self.col), _lib.ptr(self.rowptr_t), _lib.ptr(self.col_t),
                                               _lib.ptr(flags), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                       'cb_csr_from_coo_i64')
            self.norm_out = torch.empty(N, dtype=torch.float32, device=dev)   # a = clamp(out_deg,1)^-1/2
            self.norm_in = torch.empty(N, dtype=torch.float32, device=dev)    # b = clamp(in_deg,1)^-1/2
            _lib.check(lib.cb_deg_norm_f32(_lib.ptr(self.rowptr_t), N, _lib.ptr(self.norm_out), _lib.stream_ptr()),
                       'cb_deg_norm_f32')
            _lib.check(lib.cb_deg_norm_f32(_lib.ptr(self.rowptr), N, _lib.ptr(self.norm_in), _lib.stream_ptr()),
                       'cb_deg_norm_f32')
            f = flags.tolist()                               # the one host sync of the graph build
        del ws
        self.n_zero_in_degree, n_bad, sym, self.max_in_degree = f[0], f[1], f[2], f[3]
        if n_bad:
            raise ValueError(f'edge_index has {n_bad} edges with an endpoint outside [0, {N})')
        self.symmetric = bool(sym)
        if self.symmetric:                                   # A == A^T: one CSR serves forward and backward
            self.rowptr_t, self.col_t = self.rowptr, self.col
        self._plan = self._make_plan(self.rowptr)
        self._plan_t = self._plan if self.symmetric else self._make_plan(self.rowptr_t)
        self._ws = None
        self._hot_cols()
        self.profile = None      # bench.py sets a list: one prof_rec() per aggregation launch
        self.rows_only_forwards = 0      # training forwards that evaluated their last layer on the loss rows (trunk._last_layer_on_loss_rows, stack.py; tests, bench)

    def _hot_cols(self):
        """Kernel-side column arrays with the hot-source flag in bit 31 (include/coldbrew_hip.h, cb_spmm_csr_f32 col_flags):
        the most-referenced source rows of each orientation (referenced at least twice) — as many as fit the Infinity Cache at the
        row size being gathered, tuning.T.hot_bytes / (d * element bytes) — keep the default cache policy, every other gather streams.
        self.col / self.col_t stay the plain ids (the bit-exact CSR contract).  col_k / col_t_k hold the arrays for 1 KiB rows
        (d = 256 fp32, built eagerly); other row sizes (bf16-stored rows, d = 512) are flagged on first use (flagged_cols).
        CB_SPMM_GATHER=0 switches the flags off (every gather then uses the default cache policy)."""
        import os
        self.col_k = self.col_t_k = None
        self._hot_cache = {}
        if os.environ.get('CB_SPMM_GATHER', '2') != '2' or self.E == 0 or self.n_cols < 2 * (T.hot_bytes // 1024):
            return      # small graphs: the whole feature matrix is cache resident anyway
        self.col_k, self.col_t_k = self._flag_pair(self._hot_count(1024))

    def _hot_count(self, row_bytes):
        return max(1, min(T.hot_bytes // max(int(row_bytes), 1), self.n_cols))

    def _flag_pair(self, k):
        hit = self._hot_cache.get(k)
        if hit is not None:
            return hit

        def flag(col):
            # how often each source row is gathered per launch: the library's own histogram kernel (cb_value_hist_i32; no int64 copy of the ids)
            lib = _lib.load()
            refs = torch.empty(self.n_cols, dtype=torch.int32, device=self.device)
            bad = torch.zeros(1, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(lib.cb_value_hist_i32(_lib.ptr(col), self.E, self.n_cols, _lib.ptr(refs), _lib.ptr(bad), _lib.stream_ptr()), 'cb_value_hist_i32')
            thr = torch.clamp(torch.topk(refs, k).values[-1], min=2)
            hot = (refs >= thr)[col.long()]
            return torch.where(hot, col | (-2 ** 31), col).to(torch.int32)

        ck = flag(self.col)
        ctk = None
        if self.rowptr_t is not None:
            ctk = ck if self.symmetric else flag(self.col_t)
        self._hot_cache[k] = (ck, ctk)
        return ck, ctk

    def grad_support_plan(self, keep, n_aggr, max_frac=0.6, cumulative=False, count=True):
        """Row supports of a backward whose incoming gradient is non-zero on the rows `keep` only (the masked loss: the loss_rows promise of the forward, ops.py).
        Reverse aggregation j (j = 0 for the last layer) gathers rows of the support S_j and produces non-zero rows exactly on
        S_{j+1} = the rows with a (reverse-orientation) neighbour in S_j; supports are properties of the graph and the mask, not of the
        values, so they are built once (torch ops on the device) and cached.  Returns RowSupportPlan with
          .space0            the rows of S_0 as a compact space (ids ascending, position of every row or -1, row scale a restricted)
          .fwd[j]            the forward orientation restricted to the rows of S_j where the level pays to contract on that side (_support_fwd), else None
          .levels[j]         (csr, dst): csr = the reverse orientation restricted to gathered rows in S_j with its column ids renumbered to
                             positions in S_j; dst = the compact space of S_{j+1} when |S_{j+1}| <= max_frac * N (csr then has one row per
                             member of S_{j+1}), else None (csr has all N rows, the output is an ordinary dense matrix and the plan ends).
        At most n_aggr levels, and the last of them always has a dense destination (the stage below the first layer needs all rows).
        max_frac = 0: one level, S_0 -> all rows.  Sums equal the full orientation's up to the order in which a hub row's chunks are added.
        cumulative (the 'Residual' trunk): S_{j+1} = N(S_j) ∪ S_j — a superset that also holds the rows the previous level's gradient lives on
        (rows of S_j without a neighbour in S_j are rows without edges in the level's CSR).
        count=False: a second lookup of the same step (the rows-only forward asked already) — not counted as a use for support_plan_pays()."""
        # (the cache keeps the mask tensor itself alive: its address cannot be handed to another tensor while the plan is cached, and an
        # in-place change bumps its version)
        key = (keep.data_ptr(), keep._version, int(keep.shape[0]), int(n_aggr), float(max_frac), bool(cumulative))
        if getattr(self, '_support_key', None) == key and getattr(self, '_support_mask', None) is keep:
            if count:
                self._support_hits = getattr(self, '_support_hits', 0) + 1
            return self._support_plan
        self._support_builds = getattr(self, '_support_builds', 0) + 1
        if self.rowptr_t is None:
            raise ValueError('this graph holds the forward orientation only')
        if keep.dtype != torch.bool or keep.shape[0] != self.n_cols or self.N != self.n_cols:
            raise ValueError(f'grad_support_plan: bool mask over the {self.n_cols} rows of a square graph expected')
        rp, col = self.rowptr_t, self.col_t[:self.E]
        a = self.norm_out

        def space_of(mask):
            pos = torch.cumsum(mask, 0, dtype=torch.int32) - 1
            idx = torch.nonzero(mask).flatten()
            return RowSpace(idx, torch.where(mask, pos, torch.full_like(pos, -1)), a[idx].contiguous() if a is not None else None)
        src_mask, src = keep, space_of(keep)
        plan = RowSupportPlan(src, [])
        while len(plan.levels) < n_aggr:
            kept = torch.index_select(src_mask, 0, col)                   # edges whose gathered row is in the current support
            csum = torch.cumsum(kept, 0, dtype=torch.int32)
            csum = torch.cat([csum.new_zeros(1), csum])
            rp_new = torch.index_select(csum, 0, rp)
            col_new = torch.index_select(src.pos, 0, col[kept])
            del kept, csum
            cnt = rp_new[1:] - rp_new[:-1]
            dst_mask = (cnt > 0) | src_mask if cumulative else cnt > 0
            n_dst = int(dst_mask.sum())
            last = len(plan.levels) + 1 == n_aggr
            if not last and n_dst <= max_frac * self.N:
                dst = space_of(dst_mask)
                rp_c = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt[dst.idx], 0, dtype=torch.int32)])
                plan.levels.append((CSRGraph.from_csr(rp_c, col_new, src.n, hub_threshold=self.hub_threshold), dst))
                plan.fwd.append(self._support_fwd(src, dst.n))
                src_mask, src = dst_mask, dst
            else:
                plan.levels.append((CSRGraph.from_csr(rp_new, col_new, src.n, hub_threshold=self.hub_threshold), None))
                plan.fwd.append(self._support_fwd(src, self.N))      # (few rows that reach most of the graph — sparse labels — still contract on their side)
                break
        self._support_key, self._support_plan, self._support_mask = key, plan, keep
        return plan

    def support_plan_pays(self):
        """False once the plans of a SMALL graph keep being rebuilt instead of re-used (a caller that hands a new loss mask to every step):
        after four builds, fewer than three uses per build — the caller then takes the dense backward.  Building the supports is a few
        dozen torch ops and two host syncs: on a launch-bound graph that is more than one step's row-sparse backward saves, on a large
        one it is not (S-pl10M: 14 ms per build against 35 ms saved per step), so from 2^24 edges on a plan always pays."""
        if self.E >= (1 << 24):
            return True
        builds, hits = getattr(self, '_support_builds', 0), getattr(self, '_support_hits', 0)
        return builds < 4 or hits >= 3 * builds

    def spmm_store_bwd(self, h, row_scale, bits, bwd_rowscale, c_act, p, seed, row0, row_ids=None, mix=None):
        """(g, gr) of cb_spmm_csr_store_bwd_f32 over this (forward-orientation) CSR: g = row_scale * sum of the gathered rows, gr = the backward of the
        trunk's store applied to g (mask words `bits` of the written rows, dropout mask of `seed`, factor c_act, row factor bwd_rowscale) — the plain
        aggregation followed by cb_trunk_layer_bwd_f32 without the pass's read of g.  h float32 [n_cols, d], d % 256 == 0.  row_ids (int32 [N]): this CSR's rows
        are a subset of the node rows (a compact level) — bits / bwd_rowscale / the dropout mask at the node row, row_scale and the results compact.
        mix = (operands, positions, seeds, c_mix, want_colsum) (all node rows only; <= 2 compact operands [n_q, d] with int32 position maps [N]): the first
        result is the FOLDED mix gradient c_mix * (dropout_bwd(g) + sum_q dropout_bwd_q(operand_q)) instead of g, and a third result = the column sums of
        gr / bwd_rowscale (the store's bias gradient) or None (cb_spmm_csr_store_bwd_mix_f32)."""
        import ctypes
        from . import ops
        lib = _lib.load()
        _lib.require_device(h, row_scale, bits, bwd_rowscale)
        if h.dtype != torch.float32 or h.dim() != 2 or h.shape[0] != self.n_cols or h.shape[1] % 256:
            raise ValueError(f'spmm_store_bwd: float32 [{self.n_cols}, d] rows with d % 256 == 0 expected, got {tuple(h.shape)} {h.dtype}')
        if h.stride(1) != 1:
            h = h.contiguous()
        d = h.shape[1]
        g = torch.empty((self.N, d), dtype=torch.float32, device=h.device)
        gr = torch.empty((self.N, d), dtype=torch.float32, device=h.device)
        plan = self._plan
        col_k = self.flagged_cols(False, d * 4)
        flags = int(col_k is not None and h.data_ptr() % 16 == 0 and h.stride(0) % 4 == 0)
        wsb = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
        ws = self._workspace(wsb)
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if mix is not None:
            ops_, pos_, seeds_, c_mix, want_cs = mix
            if row_ids is not None or len(ops_) > 2 or any(q is None for q in pos_):
                raise ValueError('spmm_store_bwd(mix=...): all node rows, at most two compact operands with position maps')
            k = len(ops_)
            colsum = torch.empty(d, dtype=torch.float32, device=h.device) if want_cs else None
            ws2b = lib.cb_spmm_store_bwd_mix_workspace_bytes(self.N, plan.n_hubs, d) if want_cs else 0
            ws2 = torch.empty(max(ws2b, 16), dtype=torch.uint8, device=h.device) if want_cs else None
            with torch.cuda.device(h.device):
                _lib.check(lib.cb_spmm_csr_store_bwd_mix_f32(
                    _lib.ptr(self.rowptr), _lib.ptr(col_k if flags else self.col), flags, self.N, self.E, _lib.ptr(h), h.stride(0), d, _lib.ptr(row_scale), _lib.ptr(bits),
                    _lib.ptr(bwd_rowscale), float(c_act), float(p), ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(g), d, _lib.ptr(gr), d, self.hub_threshold,
                    plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), wsb, k,
                    (ctypes.c_void_p * max(k, 1))(*[t.data_ptr() for t in ops_]), (ctypes.c_void_p * max(k, 1))(*[q.data_ptr() for q in pos_]),
                    (ctypes.c_uint64 * max(k, 1))(*[int(s_) for s_ in seeds_]), float(c_mix), _lib.ptr(colsum), _lib.ptr(ws2), ws2b, _lib.stream_ptr()),
                    'cb_spmm_csr_store_bwd_mix_f32')
            if prof is not None:
                ev1.record()
                extra = sum(int(t.numel()) * 4 for t in ops_)
                prof.append(prof_rec(ev0, ev1, self, 'store_bwd', self.algorithmic_bytes(d, row_scale=row_scale is not None, bias=False), self.N * d * 4 + self.N * d // 8 + extra))
            return g, gr, colsum
        with torch.cuda.device(h.device):
            _lib.check(lib.cb_spmm_csr_store_bwd_f32(_lib.ptr(self.rowptr), _lib.ptr(col_k if flags else self.col), flags, self.N, self.E, _lib.ptr(h), h.stride(0), d,
                                                     _lib.ptr(row_scale), _lib.ptr(bits), _lib.ptr(bwd_rowscale), float(c_act), float(p), ctypes.c_uint64(seed),
                                                     ops.seed_dev_ptr(), int(row0), _lib.ptr(g), d, _lib.ptr(gr), d, self.hub_threshold, plan.n_hubs, plan.n_chunks,
                                                     _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), wsb, _lib.ptr(row_ids), _lib.stream_ptr()),
                       'cb_spmm_csr_store_bwd_f32')
        if prof is not None:
            ev1.record()
            prof.append(prof_rec(ev0, ev1, self, 'store_bwd', self.algorithmic_bytes(d, row_scale=row_scale is not None, bias=False), self.N * d * 4 + self.N * d // 8))
        return g, gr

    def loss_rows_fwd(self, plan):
        """The forward orientation on the rows of S_0 of a plan (plan.fwd[0]), built now if the thresholds of _support_fwd had left it out: the rows-only
        forward of trunk.py evaluates the last layer on those rows whatever the break-even of the backward's source-side form says."""
        if plan.fwd and plan.fwd[0] is not None:
            return plan.fwd[0]
        # (kept apart from plan.fwd: that list is what the break-evens of _support_fwd decided for the BACKWARD's source-side form, and a later
        # backward on this cached plan that did not follow a rows-only forward must still see their answer — ADVICE r05)
        if plan.fwd0_forced is None:
            dst = plan.levels[0][1]
            plan.fwd0_forced = self._support_fwd(plan.space0, dst.n if dst is not None else self.N, force=True)
        return plan.fwd0_forced

    def rows_only_fwd(self, plan):
        """Orientations of a rows-only forward that also evaluates the layer BELOW the last one on the rows the last layer reads — S_1, when the plan keeps
        that support compact (else None): (fwd1, fwd0c, ids1, b1, s1) with fwd1 = the forward orientation on the rows of S_1 (sources: all node rows),
        fwd0c = the one on the rows of S_0 with its sources renumbered to positions in S_1 (every in-neighbour of a loss row is a member of S_1),
        ids1 = int32 node ids of S_1's rows, b1 = norm_in on them, s1 = the RowSpace.  Built once per plan."""
        hit = getattr(plan, '_rows_fwd', None)
        if hit is not None:
            return hit or None
        s1 = plan.levels[0][1]
        if s1 is None:
            plan._rows_fwd = ()
            return None
        fwd0 = self.loss_rows_fwd(plan)
        fwd1 = self._support_fwd(s1, self.N, force=True)
        col_c = torch.index_select(s1.pos, 0, fwd0.col[:fwd0.E].long())
        fwd0c = CSRGraph.from_csr(fwd0.rowptr, col_c, s1.n, hub_threshold=self.hub_threshold)
        plan._rows_fwd = (fwd1, fwd0c, s1.idx.to(torch.int32).contiguous(), self.norm_in[s1.idx].contiguous(), s1)
        return plan._rows_fwd

    def _support_fwd(self, s0, n_out, force=False):
        """The FORWARD orientation restricted to the rows of a support S_j (one row per member; its in-neighbours — all of them members of
        S_{j+1} — keep their global ids): (A (a * X))[S_j] = fwd.spmm(X, col_scale=a).  With it the weight gradient of the level
        X^T (a * A^T dY) is taken as ((A (a * X))[S_j])^T dY[S_j] — a contraction over |S_j| rows instead of |S_{j+1}| — and
        a * (A^T dY) W^T as a * A^T (dY W^T): the GEMM on |S_j| rows in front of the aggregation (trunk.py).  n_out = the rows the level
        writes (|S_{j+1}|, or all rows when its destination is dense).  Built when the rows spared (n_out - |S_j|) outweigh the second pass
        over the level's edges (tuning.T.fwd0_rows_per_edge) and the level is large enough to be bound by bandwidth (fwd0_min_edges), else None."""
        rpf, colf = self.rowptr, self.col[:self.E]
        deg0 = torch.index_select(rpf[1:] - rpf[:-1], 0, s0.idx)
        rp_c = torch.cat([deg0.new_zeros(1), torch.cumsum(deg0, 0, dtype=torch.int32)])
        e0 = int(rp_c[-1])
        if not force and ((n_out - s0.n) < T.fwd0_rows_per_edge * e0 or e0 < T.fwd0_min_edges):
            return None
        shift = torch.index_select(rpf, 0, s0.idx).long() - rp_c[:-1].long()          # CSR position minus packed position, per S_0 row
        epos = torch.arange(e0, device=rpf.device) + torch.repeat_interleave(shift, deg0.long())
        return CSRGraph.from_csr(rp_c, torch.index_select(colf, 0, epos), self.n_cols, hub_threshold=self.hub_threshold)

    def flagged_cols(self, transpose, row_bytes):
        """Flagged column ids for source rows of `row_bytes` bytes (None: flags off / small graph)."""
        if self.col_k is None:
            return None
        if row_bytes == 1024:
            return self.col_t_k if transpose else self.col_k
        if self.n_cols < 2 * self._hot_count(row_bytes):
            return None
        pair = self._flag_pair(self._hot_count(row_bytes))
        return pair[1] if transpose else pair[0]

    @classmethod
    def from_csr(cls, rowptr, col, n_cols, hub_threshold=None):
        """Wraps an existing (possibly rectangular: len(rowptr)-1 rows x n_cols columns) device CSR, e.g.
        the row block a rank owns in the node-sharded path.  Forward orientation only."""
        _lib.require_device(rowptr, col)
        g = cls.__new__(cls)
        g.device = rowptr.device
        g.N, g.E, g.n_cols = int(rowptr.numel()) - 1, int(col.numel()), int(n_cols)
        g.hub_threshold = int(T.hub_threshold if hub_threshold is None else hub_threshold)
        g.rowptr = rowptr.to(torch.int32).contiguous()
        g.col = col.to(torch.int32).contiguous() if col.numel() else torch.zeros(1, dtype=torch.int32, device=g.device)
        g.rowptr_t = g.col_t = None
        g.symmetric, g.n_zero_in_degree, g.max_in_degree = False, 0, -1
        g._plan = g._make_plan(g.rowptr)
        g._plan_t = None
        g._ws, g.profile = None, None
        g.row_offset = 0
        g.n_cols = int(n_cols)
        g._hot_cols()                  # row blocks of the node-sharded path gather from [local | halo] matrices far beyond the caches too
        return g

    @classmethod
    def from_pairs(cls, rows, cols, n_rows, n_cols, hub_threshold=None):
        """Rectangular device CSR (n_rows x n_cols, ascending columns inside a row) from (row, col) pairs through the same
        C-ABI ingest as the square graph — the row block a rank owns in the node-sharded path (dist.py)."""
        lib = _lib.load()
        _lib.require_device(rows, cols)
        dev = rows.device
        rows, cols = rows.to(torch.int64).contiguous(), cols.to(torch.int64).contiguous()
        E, n = int(rows.numel()), max(int(n_rows), int(n_cols), 1)
        if E >= INT32_EDGE_LIMIT or n >= INT32_EDGE_LIMIT:
            raise ValueError(f'this row block holds {E} edges / {n} columns: a rank\'s CSR is int32-indexed (< 2^31); use more ranks')
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        rowptr_t = torch.empty(n + 1, dtype=torch.int32, device=dev)      # the ingest builds both orientations; the transposed
        col_t = torch.empty(max(E, 1), dtype=torch.int32, device=dev)     # one of a row block is not used
        flags = torch.empty(4, dtype=torch.int32, device=dev)
        wsb = lib.cb_csr_workspace_bytes(E, n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.cb_csr_from_coo_i64(_lib.ptr(cols), _lib.ptr(rows), E, n, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(rowptr_t),
                                               _lib.ptr(col_t), _lib.ptr(flags), _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_csr_from_coo_i64')
            n_bad = flags.tolist()[1]
        if n_bad or (E and (int(rows.max()) >= n_rows or int(cols.max()) >= n_cols)):
            raise ValueError('from_pairs: a (row, col) pair lies outside the n_rows x n_cols block')
        del rowptr_t, col_t, ws
        return cls.from_csr(rowptr[:int(n_rows) + 1].clone(), col[:E].clone(), n_cols, hub_threshold)

    # -- DGL-like surface (GCN.py:188,200,206,243) --------------------------------------
    def number_of_nodes(self):
        return self.N

    def number_of_edges(self):
        return self.E

    def in_degrees(self):
        return (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)

    def out_degrees(self):
        return (self.rowptr_t[1:] - self.rowptr_t[:-1]).to(torch.int64)

    def check_zero_in_degree(self):
        """GCN.py:187-197 — evaluated once at build time (the graph is immutable and cached)."""
        if self.n_zero_in_degree:
            raise ZeroInDegreeError('There are 0-in-degree nodes in the graph, output for those nodes will be invalid. '
                                    'Adding self-loop on the input graph will resolve the issue.')

    # -- plan -------------------------------------------------------------------------
    def _make_plan(self, rowptr):
        lib = _lib.load()
        p = _Plan()
        counts = torch.empty(2, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.cb_spmm_hub_count(_lib.ptr(rowptr), self.N, self.hub_threshold, _lib.ptr(counts),
                                             _lib.stream_ptr()), 'cb_spmm_hub_count')
            p.n_hubs, p.n_chunks = counts.tolist()
            p.hub_rows = torch.empty(max(p.n_hubs, 1), dtype=torch.int32, device=self.device)
            p.hub_chunk_ptr = torch.empty(p.n_hubs + 1, dtype=torch.int32, device=self.device)
            cursor = torch.empty(int(lib.cb_spmm_hub_fill_scratch_ints(self.N)), dtype=torch.int32, device=self.device)      # (scratch of the ordered compaction)
            _lib.check(lib.cb_spmm_hub_fill(_lib.ptr(rowptr), self.N, self.hub_threshold, p.n_hubs, _lib.ptr(p.hub_rows),
                                            _lib.ptr(p.hub_chunk_ptr), _lib.ptr(cursor), _lib.stream_ptr()),
                       'cb_spmm_hub_fill')
        return p

    def _workspace(self, nbytes):
        if nbytes == 0:
            return None
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    # -- the aggregation ----------------------------------------------------------------
    def spmm(self, h, transpose=False, row_scale=None, bias=None, relu=False, out=None, acc_init=None, col_scale=None):
        """out[v] = act(row_scale[v] * (acc_init[v] + sum_{u in row v} h[u]) + bias); by-dst CSR unless transpose.
        acc_init (optional, fp32 [N, d]): partial sums of an earlier pass over other columns (node-sharded path).
        col_scale (optional, fp32 [n_cols]; fp32 rows with d % 256 == 0, no bias / ReLU / acc_init): sum_u col_scale[u] * h[u]
        (cb_spmm_csr_colscale_f32)."""
        lib = _lib.load()
        _lib.require_device(h, row_scale, bias, out, acc_init, col_scale)
        if col_scale is not None and (h.dtype != torch.float32 or h.shape[1] % 256 or bias is not None or relu or acc_init is not None
                                      or col_scale.dtype != torch.float32 or col_scale.numel() != self.n_cols):
            raise ValueError('spmm(col_scale=...): float32 rows with d % 256 == 0, one float32 factor per column, no bias / ReLU / acc_init')
        if h.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f'aggregation expects float32 (or bf16-stored) features, got {h.dtype}')
        bf16 = h.dtype == torch.bfloat16
        if transpose and self.rowptr_t is None:
            raise ValueError('this graph holds the forward orientation only')
        if h.dim() != 2 or h.shape[0] != self.n_cols:
            raise ValueError(f'feature matrix must be [{self.n_cols}, d], got {tuple(h.shape)}')
        if h.stride(1) != 1 and h.shape[1] > 1:
            h = h.contiguous()
        d = h.shape[1]
        if out is None:
            out = torch.empty((self.N, d), dtype=torch.float32, device=h.device)
        rowptr, col, plan = (self.rowptr_t, self.col_t, self._plan_t) if transpose else (self.rowptr, self.col, self._plan)
        col_k = self.flagged_cols(transpose, d * h.element_size()) if d % 256 == 0 else None
        flags = int(col_k is not None and d % 256 == 0 and h.data_ptr() % 16 == 0
                    and out.data_ptr() % 16 == 0 and h.stride(0) % 4 == 0 and out.stride(0) % 4 == 0)
        if flags:
            col = col_k
        ws_bytes = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
        ws = self._workspace(ws_bytes)
        ld_h = h.stride(0) if h.shape[0] > 1 else d
        ld_o = out.stride(0) if out.shape[0] > 1 else d
        prof = self.profile
        if prof is not None:   # HIP events on the stream the kernels are launched on
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        fn = lib.cb_spmm_csr_bf16_f32 if bf16 else lib.cb_spmm_csr_f32
        with torch.cuda.device(h.device):
            if acc_init is not None:
                if acc_init.dtype != torch.float32 or acc_init.shape != (self.N, d) or acc_init.stride(1) != 1:
                    raise ValueError('acc_init must be a float32 [N, d] matrix with contiguous rows')
                _lib.check((lib.cb_spmm_csr_acc_bf16_f32 if bf16 else lib.cb_spmm_csr_acc_f32)(_lib.ptr(rowptr), _lib.ptr(col), flags, self.N, self.E, _lib.ptr(h), ld_h, d,
                                                   _lib.ptr(row_scale), _lib.ptr(bias), int(bool(relu)), _lib.ptr(acc_init),
                                                   acc_init.stride(0) if self.N > 1 else d, _lib.ptr(out), ld_o,
                                                   self.hub_threshold, plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows),
                                                   _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                           'cb_spmm_csr_acc_f32')
            elif col_scale is not None:
                _lib.check(lib.cb_spmm_csr_colscale_f32(_lib.ptr(rowptr), _lib.ptr(col), flags, self.N, self.E, _lib.ptr(h), ld_h, d,
                                                        _lib.ptr(col_scale.contiguous()), _lib.ptr(row_scale), _lib.ptr(out), ld_o,
                                                        self.hub_threshold, plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows),
                                                        _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                           'cb_spmm_csr_colscale_f32')
            else:
                _lib.check(fn(_lib.ptr(rowptr), _lib.ptr(col), flags, self.N, self.E, _lib.ptr(h), ld_h, d,
                              _lib.ptr(row_scale), _lib.ptr(bias), int(bool(relu)), _lib.ptr(out), ld_o,
                              self.hub_threshold, plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows),
                              _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                           'cb_spmm_csr')
        if prof is not None:
            ev1.record()
            prof.append(prof_rec(ev0, ev1, self, 'colscale' if col_scale is not None else 'plain',
                                 self.algorithmic_bytes(d, row_scale=row_scale is not None, bias=bias is not None, src_elem=2 if bf16 else 4)))
        return out

    def spmm_gemm_trunkbwd(self, h, image, g_rowscale, bits, c_act, p, seed, row0, rowscale2, want_colsum, transpose=True, acc_init=None):
        """(out, g, gr, colsum) of cb_spmm_gemm_trunkbwd_f32: out = A h (raw sums), g = g_rowscale * (out @ B) and, from the same epilogue, the
        trunk backward of the layer below: gr = c_act * dropout_bwd(g) * bits * rowscale2, colsum = column sums of the unscaled gr.
        acc_init: partial sums of the earlier passes of a node-sharded aggregation (the row sums start from them; overwritten by `out`)."""
        import ctypes
        from . import ops
        lib = _lib.load()
        _lib.require_device(h, image, g_rowscale, bits, rowscale2, acc_init)
        d = h.shape[1] if h.dim() == 2 else -1
        if h.dtype != torch.float32 or d != 256 or h.shape[0] != self.n_cols or h.stride(1) != 1:
            raise ValueError(f'spmm_gemm_trunkbwd: float32 [{self.n_cols}, 256] rows expected, got {tuple(h.shape)} {h.dtype}')
        if bits.dtype != torch.int64 or tuple(bits.shape) != (self.N, 1, 4) or not bits.is_contiguous():
            raise ValueError('spmm_gemm_trunkbwd: int64 [N, 1, 4] mask words expected')
        dev = h.device
        out = self._acc_out(acc_init, d, dev)
        g = torch.empty((self.N, 256), dtype=torch.float32, device=dev)
        gr = torch.empty((self.N, 256), dtype=torch.float32, device=dev)
        colsum = torch.empty(256, dtype=torch.float32, device=dev) if want_colsum else None
        rowptr, col, plan = (self.rowptr_t, self.col_t, self._plan_t) if transpose else (self.rowptr, self.col, self._plan)
        col_k = self.flagged_cols(transpose, d * 4)
        flags = int(col_k is not None and h.data_ptr() % 16 == 0 and h.stride(0) % 4 == 0)
        if flags:
            col = col_k
        ws_bytes = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
        ws = self._workspace(ws_bytes)
        ws2b = lib.cb_spmm_gemm_trunkbwd_workspace_bytes() if want_colsum else 0
        ws2 = torch.empty(max(ws2b, 16), dtype=torch.uint8, device=dev)
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        with torch.cuda.device(dev):
            _lib.check(lib.cb_spmm_gemm_trunkbwd_f32(_lib.ptr(rowptr), _lib.ptr(col), flags, self.N, self.E, _lib.ptr(h), h.stride(0), d,
                                                     _lib.ptr(acc_init), acc_init.stride(0) if acc_init is not None else 0, _lib.ptr(out), d,
                                                     self.hub_threshold, plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr),
                                                     _lib.ptr(ws), ws_bytes, _lib.ptr(image), _lib.ptr(g_rowscale), _lib.ptr(g), 256, _lib.ptr(bits),
                                                     float(c_act), float(p), ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(rowscale2),
                                                     _lib.ptr(gr), 256, _lib.ptr(colsum), _lib.ptr(ws2), ws2b, _lib.stream_ptr()),
                       'cb_spmm_gemm_trunkbwd_f32')
        if prof is not None:
            ev1.record()
            prof.append(prof_rec(ev0, ev1, self, 'agg_gemm_trunkbwd', self.algorithmic_bytes(d, row_scale=False, bias=False), 0,
                                 self.N * 256 * 4 * 2 + 4 * self.N + 32 * self.N))
        return out, g, gr, colsum

    def spmm_lp(self, h, row_scale, mix, c_mix, post_scale=None, out=None):
        """out = post_scale * clamp(row_scale * (A h) + c_mix * mix, 0, 1): one label-propagation step with both elementwise passes in the
        aggregation's store (cb_spmm_csr_lp_f32)."""
        lib = _lib.load()
        _lib.require_device(h, row_scale, mix, post_scale, out)
        if h.dtype != torch.float32 or h.dim() != 2 or h.shape[0] != self.n_cols or mix.shape != (self.N, h.shape[1]) or mix.dtype != torch.float32:
            raise ValueError('spmm_lp: float32 [n_cols, d] rows and a float32 [N, d] mix matrix expected')
        h = h if h.stride(1) == 1 else h.contiguous()
        mix = mix if mix.stride(1) == 1 else mix.contiguous()
        d = h.shape[1]
        if out is None:
            out = torch.empty((self.N, d), dtype=torch.float32, device=h.device)
        plan = self._plan
        ws_bytes = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
        ws = self._workspace(ws_bytes)
        with torch.cuda.device(h.device):
            _lib.check(lib.cb_spmm_csr_lp_f32(_lib.ptr(self.rowptr), _lib.ptr(self.col), self.N, self.E, _lib.ptr(h), h.stride(0) if h.shape[0] > 1 else d,
                                              d, _lib.ptr(row_scale), _lib.ptr(mix), mix.stride(0) if self.N > 1 else d, float(c_mix),
                                              _lib.ptr(post_scale), _lib.ptr(out), out.stride(0) if self.N > 1 else d, self.hub_threshold, plan.n_hubs,
                                              plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr()), 'cb_spmm_csr_lp_f32')
        return out

    def edge_perm(self, transpose=False):
        """CSR position -> column of edge_index (int64 [E]) for the by-dst (transpose=False) or by-src orientation: the stable sort of
        the (row, col) keys, i.e. the order cb_csr_from_coo_i64 lays the edges out in (duplicates of a multigraph keep their input
        order; which duplicate sits where is immaterial for sums over them).  Needed only by the edge_weight form (GCN.py:199-202)."""
        if getattr(self, '_ei', None) is None:
            raise ValueError('edge permutation requested on a graph built without keep_edge_order=True (CSRGraph(edge_index, n, keep_edge_order=True))')
        which = '_perm_t' if transpose else '_perm'
        if getattr(self, which) is None:
            src, dst = self._ei[0], self._ei[1]
            key = (src * self.N + dst) if transpose else (dst * self.N + src)
            setattr(self, which, torch.sort(key, stable=True)[1])
        return getattr(self, which)

    def spmm_weighted(self, h, w_csr, transpose=False, row_scale=None, bias=None, relu=False):
        """out[v] = act(row_scale[v] * sum_j w_csr[j] * h[col[j]] + bias) (cb_spmm_csr_weighted_f32); w_csr: fp32 [E] in the order of the
        chosen orientation's CSR (edge_weight[edge_perm(transpose)])."""
        lib = _lib.load()
        _lib.require_device(h, w_csr, row_scale, bias)
        if h.dtype != torch.float32 or h.dim() != 2 or h.shape[0] != self.n_cols:
            raise ValueError(f'spmm_weighted: float32 [{self.n_cols}, d] rows expected, got {tuple(h.shape)} {h.dtype}')
        if w_csr.dtype != torch.float32 or w_csr.numel() != self.E:
            raise ValueError(f'spmm_weighted: {self.E} float32 edge weights expected, got {tuple(w_csr.shape)} {w_csr.dtype}')
        if transpose and self.rowptr_t is None:
            raise ValueError('this graph holds the forward orientation only')
        if h.stride(1) != 1 and h.shape[1] > 1:
            h = h.contiguous()
        w_csr = w_csr.contiguous()
        d = h.shape[1]
        out = torch.empty((self.N, d), dtype=torch.float32, device=h.device)
        rowptr, col = (self.rowptr_t, self.col_t) if transpose else (self.rowptr, self.col)
        with torch.cuda.device(h.device):
            _lib.check(lib.cb_spmm_csr_weighted_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(w_csr), self.N, self.E, _lib.ptr(h),
                                                    h.stride(0) if h.shape[0] > 1 else d, d, _lib.ptr(row_scale), _lib.ptr(bias),
                                                    int(bool(relu)), _lib.ptr(out), d, _lib.stream_ptr()), 'cb_spmm_csr_weighted_f32')
        return out

    def edge_dot(self, h, g, transpose=False):
        """dw[j] = <h[col[j]], g[row of j]> over the chosen orientation's CSR (cb_spmm_edge_dot_f32): gradient of the edge weights."""
        lib = _lib.load()
        _lib.require_device(h, g)
        h, g = (t if t.stride(1) == 1 else t.contiguous() for t in (h, g))
        d = h.shape[1]
        dw = torch.empty(max(self.E, 1), dtype=torch.float32, device=h.device)
        rowptr, col = (self.rowptr_t, self.col_t) if transpose else (self.rowptr, self.col)
        with torch.cuda.device(h.device):
            _lib.check(lib.cb_spmm_edge_dot_f32(_lib.ptr(rowptr), _lib.ptr(col), self.N, self.E, _lib.ptr(h), h.stride(0) if h.shape[0] > 1 else d,
                                                _lib.ptr(g), g.stride(0) if g.shape[0] > 1 else d, d, _lib.ptr(dw), _lib.stream_ptr()),
                       'cb_spmm_edge_dot_f32')
        return dw[:self.E]

    def spmm_gemm(self, h, image, transpose=False, row_scale=None, bias=None, relu=False, g_rowscale=None, g_addend=None, acc_init=None):
        """(out, g_out): out = act(row_scale * (A h) + bias) as spmm() and, from the same kernel, g_out = g_rowscale * (out @ B) +
        g_addend with B the 256 x 256 matrix behind `image` (weight_image) — cb_spmm_gemm_f32: a block keeps its 64 aggregated rows
        in LDS and multiplies them on the matrix cores while other wavefronts gather.  acc_init (fp32 [N, 256], overwritten by `out`): the
        partial sums of the earlier passes of a node-sharded aggregation — the last halo pass then also yields the next dense transform."""
        lib = _lib.load()
        _lib.require_device(h, image, row_scale, bias, g_rowscale, g_addend, acc_init)
        d = h.shape[1] if h.dim() == 2 else -1
        if h.dtype != torch.float32 or d != 256 or h.shape[0] != self.n_cols or h.stride(1) != 1:
            raise ValueError(f'spmm_gemm: float32 [{self.n_cols}, 256] rows expected, got {tuple(h.shape)} {h.dtype}')
        if transpose and self.rowptr_t is None:
            raise ValueError('this graph holds the forward orientation only')
        out = self._acc_out(acc_init, d, h.device)
        g_out = torch.empty((self.N, 256), dtype=torch.float32, device=h.device)
        rowptr, col, plan = (self.rowptr_t, self.col_t, self._plan_t) if transpose else (self.rowptr, self.col, self._plan)
        col_k = self.flagged_cols(transpose, d * 4)
        flags = int(col_k is not None and h.data_ptr() % 16 == 0 and h.stride(0) % 4 == 0)
        if flags:
            col = col_k
        ws_bytes = lib.cb_spmm_workspace_bytes(plan.n_chunks, d)
        ws = self._workspace(ws_bytes)
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if g_addend is not None and g_addend.stride(1) != 1:
            g_addend = g_addend.contiguous()
        with torch.cuda.device(h.device):
            _lib.check(lib.cb_spmm_gemm_f32(_lib.ptr(rowptr), _lib.ptr(col), flags, self.N, self.E, _lib.ptr(h), h.stride(0), d,
                                            _lib.ptr(row_scale), _lib.ptr(bias), int(bool(relu)), _lib.ptr(acc_init),
                                            acc_init.stride(0) if acc_init is not None else 0, _lib.ptr(out), d, self.hub_threshold,
                                            plan.n_hubs, plan.n_chunks, _lib.ptr(plan.hub_rows), _lib.ptr(plan.hub_chunk_ptr), _lib.ptr(ws),
                                            ws_bytes, _lib.ptr(image), _lib.ptr(g_rowscale), _lib.ptr(g_addend),
                                            g_addend.stride(0) if g_addend is not None else 0, _lib.ptr(g_out), 256, _lib.stream_ptr()),
                       'cb_spmm_gemm_f32')
        if prof is not None:
            ev1.record()
            # SURVEY §8(d) bytes of the aggregation; the dense tail's own stream (its [N, 256] output) is kept apart
            prof.append(prof_rec(ev0, ev1, self, 'agg_gemm', self.algorithmic_bytes(d, row_scale=row_scale is not None, bias=bias is not None), 0,
                                 self.N * 256 * 4 * (2 if g_addend is not None else 1) + (4 * self.N if g_rowscale is not None else 0)))
        return out, g_out

    def _acc_out(self, acc_init, d, device):
        """Output matrix of an aggregation that starts from partial sums: the sums themselves (in place), else a fresh [N, d] matrix."""
        if acc_init is None:
            return torch.empty((self.N, d), dtype=torch.float32, device=device)
        if acc_init.dtype != torch.float32 or tuple(acc_init.shape) != (self.N, d) or acc_init.stride(1) != 1 or acc_init.stride(0) % 4:
            raise ValueError('acc_init must be a float32 [N, d] matrix with contiguous, 16-byte aligned rows')
        return acc_init

    def algorithmic_bytes(self, d, elem=4, row_scale=True, bias=True, src_elem=None):
        """E*(d*s+4) + N*(d*s+4) [+4N row scale] [+d*s bias]; src_elem = bytes per gathered element
        when the source rows are stored narrower than the output (bf16 variant)."""
        b = self.E * (d * (src_elem or elem) + 4) + self.N * (d * elem + 4)
        if row_scale:
            b += 4 * self.N
        if bias:
            b += d * elem
        return b


class SegmentedCSRGraph:
    """A graph with E >= 2^31 edge_index columns on ONE device ("indices int32 (int64 rowptr if E >= 2^31)"; the
    reference's DGL graph is int64 throughout, GCN.py:93-94).  The ingest writes int64 row pointers (cb_csr64_from_coo_i64), column ids
    stay int32.  The aggregation kernels index edges with 32 bits inside a launch, so the rows are cut into blocks of at most `max_edges`
    (< 2^31) edges; a block is an ordinary rectangular CSRGraph over a VIEW of the column ids with rebased row pointers
    (cb_csr_rebase_i64) — exactly the shape of a rank's row block in the node-sharded path — and an aggregation is one launch per block
    into the block's rows of the output.  Serves the operator path (ops.aggregate: forward and reverse orientation); the fused trunk keeps
    to graphs below 2^31 edges.  max_edges is a parameter so that the segment logic is testable on small graphs."""
    segmented = True

    def __init__(self, edge_index, num_nodes=None, hub_threshold=None, max_edges=INT32_EDGE_LIMIT - 1):
        lib = _lib.load()
        _lib.require_device(edge_index)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f'edge_index must be [2, E], got {tuple(edge_index.shape)}')
        ei = edge_index.to(torch.int64).contiguous()
        dev = ei.device
        E = int(ei.shape[1])
        N = int(num_nodes) if num_nodes is not None else (int(ei.max().item()) + 1 if E else 0)
        if N >= INT32_EDGE_LIMIT:
            raise ValueError(f'{N} nodes: column ids are int32 (N < 2^31)')
        self.N, self.E, self.device, self.n_cols, self.row_offset = N, E, dev, N, 0
        self.hub_threshold, self.max_edges = int(T.hub_threshold if hub_threshold is None else hub_threshold), int(max_edges)
        self.rowptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
        self.col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        self.rowptr_t = torch.empty(N + 1, dtype=torch.int64, device=dev)
        self.col_t = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        flags = torch.empty(4, dtype=torch.int32, device=dev)
        wsb = lib.cb_csr_workspace_bytes(E, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.cb_csr64_from_coo_i64(_lib.ptr(ei[0]), _lib.ptr(ei[1]), E, N, _lib.ptr(self.rowptr), _lib.ptr(self.col),
                                                 _lib.ptr(self.rowptr_t), _lib.ptr(self.col_t), _lib.ptr(flags), _lib.ptr(ws), wsb,
                                                 _lib.stream_ptr()), 'cb_csr64_from_coo_i64')
            self.norm_out = torch.empty(N, dtype=torch.float32, device=dev)
            self.norm_in = torch.empty(N, dtype=torch.float32, device=dev)
            _lib.check(lib.cb_deg_norm_i64ptr_f32(_lib.ptr(self.rowptr_t), N, _lib.ptr(self.norm_out), _lib.stream_ptr()), 'cb_deg_norm_i64ptr_f32')
            _lib.check(lib.cb_deg_norm_i64ptr_f32(_lib.ptr(self.rowptr), N, _lib.ptr(self.norm_in), _lib.stream_ptr()), 'cb_deg_norm_i64ptr_f32')
            f = flags.tolist()
        del ws, ei
        self.n_zero_in_degree, n_bad, sym, self.max_in_degree = f[0], f[1], f[2], f[3]
        if n_bad:
            raise ValueError(f'edge_index has {n_bad} edges with an endpoint outside [0, {N})')
        self.symmetric = bool(sym)
        if self.symmetric:
            self.rowptr_t, self.col_t = self.rowptr, self.col
        self.segments = self._cut(self.rowptr, self.col)
        self.segments_t = self.segments if self.symmetric else self._cut(self.rowptr_t, self.col_t)
        self.profile = None

    def _cut(self, rowptr, col):
        """[(first row, end row, CSRGraph of the block)]: greedy blocks of at most max_edges edges."""
        lib = _lib.load()
        out, r0 = [], 0
        while r0 < self.N:
            e0 = int(rowptr[r0])
            r1 = int(torch.searchsorted(rowptr, torch.tensor([e0 + self.max_edges], dtype=torch.int64, device=rowptr.device), right=True)) - 1
            r1 = min(r1, self.N)
            if r1 <= r0:
                raise ValueError(f'row {r0} alone holds more than {self.max_edges} edges: a destination row is reduced inside one launch')
            e1 = int(rowptr[r1])
            rp = torch.empty(r1 - r0 + 1, dtype=torch.int32, device=rowptr.device)
            with torch.cuda.device(rowptr.device):
                _lib.check(lib.cb_csr_rebase_i64(_lib.ptr(rowptr), r0, r1 - r0, _lib.ptr(rp), _lib.stream_ptr()), 'cb_csr_rebase_i64')
            out.append((r0, r1, CSRGraph.from_csr(rp, col[e0:e1], self.N, self.hub_threshold)))
            r0 = r1
        return out

    # -- the surface GCNConv / ops.aggregate touch ----------------------------------------------------------
    def number_of_nodes(self):
        return self.N

    def number_of_edges(self):
        return self.E

    def in_degrees(self):
        return self.rowptr[1:] - self.rowptr[:-1]

    def out_degrees(self):
        return self.rowptr_t[1:] - self.rowptr_t[:-1]

    def check_zero_in_degree(self):
        if self.n_zero_in_degree:
            raise ZeroInDegreeError('There are 0-in-degree nodes in the graph, output for those nodes will be invalid. '
                                    'Adding self-loop on the input graph will resolve the issue.')

    def algorithmic_bytes(self, d, elem=4, row_scale=True, bias=True, src_elem=None):
        return self.E * (d * (src_elem or elem) + 4) + self.N * (d * elem + 4) + (4 * self.N if row_scale else 0) + (d * elem if bias else 0)

    def spmm(self, h, transpose=False, row_scale=None, bias=None, relu=False, out=None, acc_init=None):
        """As CSRGraph.spmm: one launch per row block into that block's rows of `out`."""
        if h.dim() != 2 or h.shape[0] != self.N:
            raise ValueError(f'feature matrix must be [{self.N}, d], got {tuple(h.shape)}')
        if out is None:
            out = torch.empty((self.N, h.shape[1]), dtype=torch.float32, device=h.device)
        for r0, r1, seg in (self.segments_t if transpose else self.segments):
            seg.profile = self.profile
            seg.spmm(h, row_scale=row_scale[r0:r1] if row_scale is not None else None, bias=bias, relu=relu, out=out[r0:r1],
                     acc_init=acc_init[r0:r1] if acc_init is not None else None)
        return out


class RowSpace:
    """A subset of the node rows as a compact index space: idx int64 [n] (ascending global row ids), pos int32 [N] (position of every row
    in idx, -1 outside), a = norm_out restricted to the subset.  Matrices "over" the space are [n, d] and hold the rows idx."""

    def __init__(self, idx, pos, a):
        self.idx, self.pos, self.a, self.n = idx, pos, a, int(idx.numel())


class RowSupportPlan:
    def __init__(self, space0, levels):
        self.space0, self.levels = space0, levels
        self.fwd = []      # per level: the forward orientation on the level's source rows (CSRGraph._support_fwd) or None
        self.fwd0_forced = None      # level 0's, built for a rows-only FORWARD although the backward's break-evens left fwd[0] out (CSRGraph.loss_rows_fwd)

    @property
    def fwd0(self):
        return self.fwd[0] if self.fwd else None


def build_graph(edge_index, num_nodes=None):
    """The device graph TricksComb caches in place of the reference's dgl.graph (GCN.py:92-95): int32 CSR below 2^31 edge_index columns,
    int64 row pointers + row blocks from there on."""
    if int(edge_index.shape[1]) >= INT32_EDGE_LIMIT:
        return SegmentedCSRGraph(edge_index, num_nodes)
    return CSRGraph(edge_index, num_nodes)


def weight_image(w, transpose=False):
    """The 256 x 256 matrix B = w (or w^T) split once into bf16 limbs in MFMA fragment order (cb_agg_gemm_image_f32): the B operand
    of the dense tail of spmm_gemm / the fused trunk kernels.  384 KB, rebuilt every step (the weights change)."""
    lib = _lib.load()
    _lib.require_device(w)
    if w.dtype != torch.float32 or tuple(w.shape) != (256, 256):
        raise ValueError(f'weight_image: float32 [256, 256] expected, got {tuple(w.shape)} {w.dtype}')
    if w.stride(1) != 1:
        w = w.contiguous()
    nbytes = lib.cb_agg_gemm_image_bytes(256, 256)
    image = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.cb_agg_gemm_image_f32(_lib.ptr(w), w.stride(0), 256, 256, int(bool(transpose)), _lib.ptr(image), nbytes,
                                             _lib.stream_ptr()), 'cb_agg_gemm_image_f32')
    return image


def head_image(w_out):
    """The 256 x C (C <= 64) matrix B = w_out^T of the output nn.Linear (weight [C, 256]) as bf16 limbs in MFMA fragment order, zero columns
    beyond C (cb_agg_gemm_head_image_f32): the B operand of the narrow tail of the last layer's aggregation kernel.  None where no such tail
    exists (C > 64 or another input width)."""
    lib = _lib.load()
    _lib.require_device(w_out)
    if w_out.dtype != torch.float32 or w_out.dim() != 2 or w_out.shape[1] != 256:
        return None
    C = int(w_out.shape[0])
    nbytes = lib.cb_agg_gemm_head_image_bytes(256, C)
    if not nbytes:
        return None
    if w_out.stride(1) != 1:
        w_out = w_out.contiguous()
    image = torch.empty(nbytes, dtype=torch.uint8, device=w_out.device)
    with torch.cuda.device(w_out.device):
        _lib.check(lib.cb_agg_gemm_head_image_f32(_lib.ptr(w_out), w_out.stride(0), 256, C, 1, _lib.ptr(image), nbytes, _lib.stream_ptr()),
                   'cb_agg_gemm_head_image_f32')
    return image
