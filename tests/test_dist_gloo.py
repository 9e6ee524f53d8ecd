"""CPU, world_size 2-3, gloo: the node-sharded logic of dist.py (edge-balanced / equal-row partition, per-rank ingest of the
row blocks, halo plans, interior/halo two-pass aggregation with the asynchronous exchange, all-gather baseline, backward
exchange, gradient / regulariser all-reduces, cross-rank column statistics of the norm tricks, rank-consistent initial
state) against the unsharded oracle.  The passes over node data are served by tests/dist_cpu_compute.py (oracle arithmetic;
the product's HIP kernels need a GPU); everything else is the product's code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)


def _worker(rank, world, port, name, q, exchange, overlap, kind, wire='f32', n_slices=None, chunked=False, cover=False):
    _setup(rank, world, port)
    try:
        import coldbrew_oracle as orc
        from dist_cpu_compute import OracleCompute
        from gnn_tail_generalization_amd import dist as cbdist
        g = load_golden(name)
        n = g['cfg']['N_nodes']
        ei = g['edge_index']
        csr = orc.build_csr(ei, n)
        a, b = orc.degree_norms(csr)
        if kind == 'edges':
            part = cbdist.Partition.balanced(torch.from_numpy(csr.in_deg), world, rank, node_weight=2)
        else:
            part = cbdist.Partition(n, world, rank)
        sg = cbdist.ShardedGraph(ei, n, part, exchange=exchange, overlap=overlap, compute=OracleCompute(), wire=wire, n_slices=n_slices, cover=cover)
        assert sg.overlap == (overlap and exchange == 'halo' and world > 1)
        assert sg.cover == (bool(cover) and sg.overlap)
        if sg.cover:
            # push / pull cover: never more rows than the pull, every remote edge served exactly once (a pulled source keeps its edges on
            # the requester, a pushed destination has one edge to its partial row and its edges in the owner's send CSR)
            plan = sg.f.plan
            assert plan.cover and plan.n_slices == (n_slices or 1) and len(sg.f.halo) == plan.n_slices == len(plan.send_csr)
            assert 0 < plan.n_halo == plan.n_pulled + plan.n_pushed <= plan.n_pull_only and sum(plan.recv_counts_all) == plan.n_halo
            assert [sum(c) for c in plan.send_counts] == plan.n_send_slice and sum(plan.send_counts_all) == sum(plan.n_send_slice)
            served = torch.tensor([sum(h.E for h in sg.f.halo) - plan.n_pushed + sum(c.E for c in plan.send_csr) - sum(plan.n_send_slice),
                                   sg.E - sg.f.interior.E, plan.n_pushed, sum(plan.n_send_slice), plan.n_pulled])
            dist.all_reduce(served)
            # sum over ranks: (pulled edges kept by requesters) + (edges summed by owners beyond one per shipped row) + (rows shipped)
            #                 - (pulled rows, which are one-edge rows of the send CSRs) == remote edges
            assert int(served[0]) + int(served[3]) - int(served[4]) == int(served[1]), served.tolist()
        elif exchange == 'halo' and world > 1:
            plan = sg.f.plan
            assert plan.n_halo > 0 and sum(plan.recv_counts_all) == plan.n_halo == sum(plan.n_halo_slice)
            assert plan.n_halo <= csr.N - part.n_local
            assert plan.n_slices == ((n_slices or 1) if sg.overlap else 1)
            for k in range(plan.n_slices):      # slice k = the requested rows living in row chunk k of their owner
                r0, r1 = plan.chunks[k]
                assert plan.send_idx[k].numel() == sum(plan.send_counts[k])
                assert plan.send_idx[k].numel() == 0 or (int(plan.send_idx[k].min()) >= r0 and int(plan.send_idx[k].max()) < r1)
            assert [c[0] for c in plan.chunks] + [plan.chunks[-1][1]] == sorted({c for ch in plan.chunks for c in ch}) or part.n_local == 0
            if sg.overlap:
                assert sg.f.interior.E + sum(h.E for h in sg.f.halo) == sg.E and sg.f.whole is None and len(sg.f.halo) == plan.n_slices
        assert sg.N == part.n_local and sg.row_offset == part.lo()
        sym = np.array_equal(csr.rowptr, csr.rowptr_t) and np.array_equal(csr.col, csr.col_t)
        assert sg.symmetric == sym and (sg.b is sg.f) == sym
        torch.testing.assert_close(sg.norm_in, part.slice_rows(b))
        torch.testing.assert_close(sg.norm_out, part.slice_rows(a))
        # the row blocks re-assemble the global edge set
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([sg.E]))
        assert sum(int(c) for c in counts) == csr.E
        gen = torch.Generator().manual_seed(0)
        d_in, d_h, C = 6, 8, 3
        x = torch.randn(n, d_in, generator=gen)
        w1 = torch.randn(d_in, d_h, generator=gen, requires_grad=True)
        w2 = torch.randn(d_h, C, generator=gen, requires_grad=True)
        b1 = torch.randn(d_h, generator=gen, requires_grad=True)
        le = torch.randn(n, d_h, generator=gen)
        y = torch.randint(0, C, (n,), generator=gen)
        mask = torch.rand(n, generator=gen) < 0.5
        # ---- sharded: local rows only ----
        xl, yl, ml = part.slice_rows(x), part.slice_rows(y), part.slice_rows(mask)
        le_l = part.slice_rows(le).clone().requires_grad_(True)
        h = (xl * sg.norm_out.unsqueeze(1)) @ w1 + le_l
        if chunked and sg.overlap:
            # the producer form of the pipeline: row chunk k of the matrix is computed right before slice k is packed and sent
            hh = torch.full_like(h, float('nan'))
            filled = []

            def produce(k, r0, r1, src=h.detach()):
                hh[r0:r1] = src[r0:r1]
                filled.append((k, r0, r1))
            got = sg.aggregate_finish(sg.aggregate_start(hh, False, produce=produce), False, sg.norm_in, b1.detach(), True)
            assert [f[0] for f in filled] == list(range(sg.f.plan.n_slices)) and filled[0][1] == 0 and filled[-1][2] == part.n_local
            torch.testing.assert_close(got, cbdist.sharded_aggregate(sg, h.detach(), sg.norm_in, b1.detach(), relu=True), atol=0, rtol=0)
        h = cbdist.sharded_aggregate(sg, h, sg.norm_in, b1, relu=True)
        h = (h * sg.norm_out.unsqueeze(1)) @ w2
        out = cbdist.sharded_aggregate(sg, h, sg.norm_in, None, relu=False)
        reg_l = torch.norm(le_l)
        reg = cbdist.allreduce_sum(reg_l * reg_l).sqrt()
        n_train = int(mask.sum())
        logp = torch.log_softmax(out, 1)
        nll_local = -(logp[torch.arange(part.n_local), yl] * ml).sum() / n_train
        loss = nll_local + 0.5 * reg / world
        loss.backward()
        cbdist.allreduce_grads([w1, w2, b1])
        total = loss.detach().clone()
        dist.all_reduce(total)
        # ---- unsharded oracle on every rank ----
        w1r, w2r, b1r = (t.detach().clone().requires_grad_(True) for t in (w1, w2, b1))
        ler = le.clone().requires_grad_(True)
        hr, regr = orc.gcnconv_forward(csr, x, w1r, b1r, ler, a, b)
        hr = torch.relu(hr)
        outr, _ = orc.gcnconv_forward(csr, hr, w2r, None, None, a, b)
        lossr = torch.nn.functional.nll_loss(torch.log_softmax(outr[mask], 1), y[mask]) + 0.5 * regr
        lossr.backward()
        # bf16 wire (opt-in): every remote neighbour row is rounded to 8 significand bits on its way — a stated, looser bound
        tol = dict(atol=1e-5, rtol=1e-5) if wire == 'f32' else dict(atol=6e-2, rtol=2e-2)
        gtol = dict(atol=2e-5, rtol=1e-4) if wire == 'f32' else dict(atol=6e-2, rtol=5e-2)
        torch.testing.assert_close(out.detach(), part.slice_rows(outr.detach()), **tol)
        torch.testing.assert_close(total, lossr.detach(), **tol)
        for got, ref in [(w1.grad, w1r.grad), (w2.grad, w2r.grad), (b1.grad, b1r.grad), (le_l.grad, part.slice_rows(ler.grad))]:
            torch.testing.assert_close(got, ref, **gtol)
        if wire == 'bf16' and world > 1:
            assert float((out.detach() - part.slice_rows(outr.detach())).abs().max()) > 0      # the rounding is really on the wire
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def _run(target, world, *args):
    import oracle_c
    oracle_c.load()                      # build the C restatement before forking
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args[:1] + (q,) + args[1:]) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == 'ok', f'rank {r}: {msg}'


@pytest.mark.parametrize('name,world,exchange,overlap,kind', [
    ('case_graph_asym_multi', 2, 'halo', True, 'edges'),          # directed multigraph: own reverse plan
    ('case_graph_powerlaw_d7_d64', 2, 'halo', True, 'edges'),     # symmetric: the reverse orientation aliases the forward one
    ('case_graph_asym_multi', 3, 'halo', True, 'rows'),
    ('case_graph_powerlaw_d7_d64', 3, 'halo', False, 'edges'),    # single-pass form ([local | halo] matrix, blocking exchange)
    ('case_graph_powerlaw_d7_d64', 2, 'allgather', False, 'rows'),
])
def test_sharded_exchange_matches_unsharded_oracle(name, world, exchange, overlap, kind):
    _run(_worker, world, name, exchange, overlap, kind)


@pytest.mark.parametrize('name,world,n_slices', [('case_graph_powerlaw_d7_d64', 2, 2), ('case_graph_powerlaw_d7_d64', 3, 4),
                                                 ('case_graph_asym_multi', 2, 3), ('case_graph_asym_multi', 3, 7)])
def test_sliced_exchange_pipeline_matches_unsharded_oracle(name, world, n_slices):
    """The exchange cut into K time slices (slice k = halo rows of owner row chunk k, every peer at once; per-slice halo passes
    chained through the running sums; producer callbacks per row chunk) gives the unsharded result, forward and backward — also
    with more slices than a small block has rows (empty slices) and on the directed multigraph (own reverse plan)."""
    _run(_worker, world, name, 'halo', True, 'edges', 'f32', n_slices, True)


def test_sliced_exchange_bf16_wire():
    _run(_worker, 2, 'case_graph_powerlaw_d7_d64', 'halo', True, 'edges', 'bf16', 3, True)


@pytest.mark.parametrize('name,world,n_slices,kind,wire', [
    ('case_graph_powerlaw_d7_d64', 2, 1, 'edges', 'f32'), ('case_graph_powerlaw_d7_d64', 3, 4, 'edges', 'f32'),
    ('case_graph_asym_multi', 2, 3, 'edges', 'f32'),            # directed multigraph: own reverse plan, duplicate edges counted
    ('case_graph_asym_multi', 3, 7, 'rows', 'f32'),             # more slices than some lists have rows (empty slices)
    ('case_r_initialbn_se111_L3_powerlaw', 2, 2, 'edges', 'f32'), ('case_graph_powerlaw_d7_d64', 2, 2, 'edges', 'bf16')])
def test_push_pull_cover_exchange_matches_unsharded_oracle(name, world, n_slices, kind, wire):
    """per rank pair a vertex cover of the remote-edge bipartite graph decides which source rows are shipped (pull)
    and which destination rows arrive as owner-side partial sums (push); the send buffer is an aggregation over a send CSR, the time
    slices cut a pair's list by position.  Forward, backward (own reverse plan on the directed multigraph) and every gradient equal the
    unsharded oracle's within the fp32 contract; never more rows on a link than the pull; every remote edge served exactly once."""
    _run(_worker, world, name, 'halo', True, kind, wire, n_slices, True, 'force')


def _support_worker(rank, world, port, name, q, cover, n_slices):
    """Level orientations of the row-sparse backward (ShardedGraph.support_orients): aggregating a matrix that is zero outside the support
    S_j through level j equals the full reverse aggregation, level j + 1's support is what that aggregation can reach, and the level ships
    fewer rows than the full plan."""
    _setup(rank, world, port)
    try:
        import coldbrew_oracle as orc
        from dist_cpu_compute import OracleCompute
        from gnn_tail_generalization_amd import dist as cbdist
        g = load_golden(name)
        n, ei = g['cfg']['N_nodes'], g['edge_index']
        csr = orc.build_csr(ei, n)
        part = cbdist.Partition.balanced(torch.from_numpy(csr.in_deg), world, rank, node_weight=2)
        sg = cbdist.ShardedGraph(ei, n, part, exchange='halo', overlap=True, compute=OracleCompute(), n_slices=n_slices, cover=cover)
        gen = torch.Generator().manual_seed(3)
        mask = torch.rand(n, generator=gen) < 0.08
        mask[0] = True
        mask_l = part.slice_rows(mask).contiguous()
        levels = sg.support_orients(mask_l, 3, max_edge_frac=0.97)
        assert len(levels) >= 1
        # round 5: the same levels COMPACT in this rank's rows (SupportLevel.src / .dst) where the plan allows it (cover, or one slice)
        clevels = sg.support_levels(mask_l, 3, max_edge_frac=0.97, compact=True, max_frac=0.95)
        can_compact = bool(sg.b.plan.cover) or sg.b.plan.n_slices == 1
        assert len(clevels) == len(levels) and (clevels[0].src is not None) == can_compact
        assert clevels[-1].dst is None or len(clevels) < 3          # the last of n_aggr levels always writes all rows
        src, dst = ei[0], ei[1]
        S = mask.clone()
        for o, lv in zip(levels, clevels):
            assert o.plan.n_slices == sg.b.plan.n_slices and bool(o.plan.cover) == bool(sg.b.plan.cover)
            assert o.plan.n_halo <= sg.b.plan.n_halo and o.E <= sg.b.E
            h = torch.randn(n, 5, generator=gen) * S.float().unsqueeze(1)          # supported on S_j
            hl = part.slice_rows(h).contiguous()
            full = sg.aggregate(hl, True)
            lvl = sg.aggregate_finish(sg.aggregate_start(hl, True, orient=o), True)
            torch.testing.assert_close(lvl, full, atol=1e-5, rtol=1e-5)
            nxt = torch.zeros(n, dtype=torch.bool)
            nxt[src[S[dst]]] = True          # reverse aggregation: row u sums the rows dst(e) of its out-edges
            assert bool((full[~part.slice_rows(nxt)] == 0).all())
            # the compact level: reads the support's rows of this rank, writes the next support's (or all rows)
            if lv.src is not None:
                assert bool(part.slice_rows(S)[lv.src.idx].sum() == part.slice_rows(S).sum())          # src covers S_j on this rank
                assert lv.orient.plan.n_halo == o.plan.n_halo and lv.orient.E == o.E                    # same exchange, same edges
                inp = hl[lv.src.idx].contiguous()
                got = sg.aggregate_finish(sg.aggregate_start(inp, True, orient=lv.orient), True)
                if lv.dst is not None:
                    assert got.shape[0] == lv.dst.n and bool(part.slice_rows(nxt)[lv.dst.idx].sum() == part.slice_rows(nxt).sum())
                    torch.testing.assert_close(got, full[lv.dst.idx], atol=1e-5, rtol=1e-5)
                else:
                    torch.testing.assert_close(got, full, atol=1e-5, rtol=1e-5)
            S = nxt
        # rows-only forward (ShardedGraph.loss_rows_forward): the FORWARD orientation restricted to the edges that enter this rank's loss rows writes
        # the rows of the full forward aggregation on those rows, and asks the peers for fewer rows than the full forward plan
        if clevels[0].src is not None:
            s0 = clevels[0].src
            o_f = sg.loss_rows_forward(clevels)
            assert sg.loss_rows_forward(clevels) is o_f                                             # built once per mask
            assert o_f.plan.n_slices == sg.f.plan.n_slices and bool(o_f.plan.cover) == bool(sg.f.plan.cover)
            assert o_f.plan.n_halo <= sg.f.plan.n_halo and o_f.E <= sg.f.E
            h = torch.randn(n, 5, generator=gen)
            hl = part.slice_rows(h).contiguous()
            full = sg.aggregate(hl, False)
            got = sg.aggregate_finish(sg.aggregate_start(hl, False, orient=o_f), False)
            assert got.shape[0] == s0.n
            torch.testing.assert_close(got, full[s0.idx], atol=1e-5, rtol=1e-5)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,world,cover,n_slices', [('case_graph_powerlaw_d7_d64', 2, False, 2), ('case_graph_powerlaw_d7_d64', 3, 'force', 2),
                                                        ('case_graph_asym_multi', 2, False, 1)])
def test_row_sparse_level_orientations_match_the_full_reverse_exchange(name, world, cover, n_slices):
    _run(_support_worker, world, name, cover, n_slices)


def _merged_worker(rank, world, port, name, q, cover, n_slices):
    """Round 5: the interior pass and the first halo slice as ONE pass (dist._Orientation.first): a matrix allocated with room behind it
    (dist.alloc_exchanged) receives slice 0 right there; forward, reverse and compact-level aggregations equal the two-pass form's — whose
    matrix has no room — within summation order, with a fused-epilogue last pass (row scale, bias, ReLU) and when the merged pass IS the last one."""
    _setup(rank, world, port)
    try:
        import coldbrew_oracle as orc
        from dist_cpu_compute import OracleCompute
        from gnn_tail_generalization_amd import dist as cbdist
        g = load_golden(name)
        n, ei = g['cfg']['N_nodes'], g['edge_index']
        csr = orc.build_csr(ei, n)
        part = cbdist.Partition.balanced(torch.from_numpy(csr.in_deg), world, rank, node_weight=2)
        sg = cbdist.ShardedGraph(ei, n, part, exchange='halo', overlap=True, compute=OracleCompute(), n_slices=n_slices, cover=cover)
        assert sg.f.first is not None and sg.halo_room >= 1
        assert sg.f.first.E == sg.f.interior.E + sg.f.halo[0].E and sg.f.first.n_cols == part.n_local + max(sg.f.plan.n_halo_slice[0], 1)
        gen = torch.Generator().manual_seed(5)
        h = torch.randn(n, 6, generator=gen)
        hl = part.slice_rows(h).contiguous()
        bias = torch.randn(6, generator=gen)
        for transpose in (False, True):
            ref = sg.aggregate(hl, transpose, sg.norm_in, bias, True)                     # two passes: hl has no room behind it
            m = cbdist.alloc_exchanged(sg, part.n_local, 6, device='cpu')
            assert getattr(m, '_cb_room', 0) == sg.halo_room
            m.copy_(hl)
            flights = sg.start_halo(m, transpose)
            assert len(flights[0]) == 4 and flights[0][0].data_ptr() == m.data_ptr() + m.numel() * 4      # slice 0 lands right behind the local rows
            o = sg.b if transpose else sg.f
            c = sg.compute
            got = sg.finish_halo(flights, o, None, lambda gg, recv, acc: c.spmm(gg, recv, sg.norm_in, bias, True, acc_init=acc), x_local=m)
            torch.testing.assert_close(got, ref, atol=1e-5, rtol=1e-5)
            torch.testing.assert_close(m, hl, atol=0, rtol=0)                              # the local rows are untouched
        # a compact level of the row-sparse backward through the merged pass
        mask = torch.rand(n, generator=gen) < 0.1
        mask[0] = True
        mask_l = part.slice_rows(mask).contiguous()
        levels = sg.support_levels(mask_l, 3, max_edge_frac=0.97, compact=True, max_frac=0.95)
        if levels and levels[0].src is not None:
            lv = levels[0]
            hs = part.slice_rows(h * mask.float().unsqueeze(1)).contiguous()
            full = sg.aggregate(hs, True)
            m = cbdist.alloc_exchanged(sg, lv.src.n, 6, device='cpu')
            m.copy_(hs[lv.src.idx])
            got = sg.aggregate_finish(sg.aggregate_start(m, True, orient=lv.orient), True)
            torch.testing.assert_close(got, full[lv.dst.idx] if lv.dst is not None else full, atol=1e-5, rtol=1e-5)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,world,cover,n_slices', [('case_graph_powerlaw_d7_d64', 2, False, 3), ('case_graph_powerlaw_d7_d64', 3, 'force', 2),
                                                        ('case_graph_asym_multi', 2, False, 1), ('case_graph_asym_multi', 3, 'force', 1)])
def test_interior_and_first_slice_as_one_pass(name, world, cover, n_slices):
    _run(_merged_worker, world, name, cover, n_slices)


def _decision_worker(rank, world, port, _unused, q):
    _setup(rank, world, port)
    try:
        from dist_cpu_compute import OracleCompute
        from gnn_tail_generalization_amd import dist as cbdist
        gen = torch.Generator().manual_seed(0)
        n = 120
        # dense: every node has ~30 neighbours on the other rank -> every source row is referenced anyway: the cover cannot beat the pull by 10 %
        a = (torch.rand(n, n, generator=gen) < 0.5)
        a = a | a.t() | torch.eye(n, dtype=torch.bool)
        dense = a.nonzero().t().contiguous()
        # star-like: rank 0 holds two hubs that all of rank 1's nodes point to and are pointed to by -> push / pull covers save most rows
        hub = torch.tensor([0, 1])
        leaves = torch.arange(n // 2, n)
        src = torch.cat([hub.repeat_interleave(leaves.numel()), leaves.repeat(2), torch.arange(n)])
        dst = torch.cat([leaves.repeat(2), hub.repeat_interleave(leaves.numel()), torch.arange(n)])
        star = torch.stack([src, dst])
        part = cbdist.Partition(n, world, rank)
        kinds = []
        for ei in (dense, star):
            sg = cbdist.ShardedGraph(ei, n, part, compute=OracleCompute(), n_slices=2)      # cover allowed (default), not forced
            assert sg.cover
            kinds.append(bool(sg.f.plan.cover))
        assert kinds == [False, True], kinds
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def test_cover_is_taken_only_where_it_saves_rows():
    """One decision per orientation for the whole group (all-reduce): the push / pull cover where it spares the busiest requester >= 10 % of
    its rows (hubs), the plain pull plan (row-gather pack, slices by owner row chunk) on a dense graph where every source is referenced anyway."""
    _run(_decision_worker, 2, None)


def test_cover_heuristic_never_worse_than_pull_or_push():
    """choose_cover on hand-made pair graphs: a star (one hub destination: push one row instead of pulling its leaves), its mirror (one hub
    source: pull one row), a perfect matching (pull, by the tie rule) and a dense block (all-pull fallback: the heuristic alone would open
    both sides)."""
    from gnn_tail_generalization_amd.dist import choose_cover

    def rows_moved(u, v, pull):
        return len(set(u[pull].tolist())) + len(set(v[~pull].tolist()))
    z = lambda n: torch.zeros(n, dtype=torch.int64)      # noqa: E731  (one owner: index 0 of a world of 2)
    u, v = torch.arange(6), z(6)                                          # 6 sources -> 1 destination
    pull = choose_cover(u, v, z(6), z(1), 6, 1, 2)
    assert rows_moved(u, v, pull) == 1 and not pull.any()
    u, v = z(6), torch.arange(6)                                          # 1 source -> 6 destinations
    pull = choose_cover(u, v, z(1), z(6), 1, 6, 2)
    assert rows_moved(u, v, pull) == 1 and pull.all()
    u = v = torch.arange(5)                                               # matching
    pull = choose_cover(u, v, z(5), z(5), 5, 5, 2)
    assert rows_moved(u, v, pull) == 5 and pull.all()
    u, v = torch.arange(20), torch.arange(20) % 19                        # 20 sources -> 19 destinations: 5 % fewer rows are not worth a push
    assert choose_cover(u, v, z(20), z(19), 20, 19, 2).all() and not choose_cover(u, v, z(20), z(19), 20, 19, 2, min_gain=0.0).any()
    gen = torch.Generator().manual_seed(0)
    u, v = torch.randint(0, 40, (2000,), generator=gen), torch.randint(0, 50, (2000,), generator=gen)       # dense 40 x 50 block
    pull = choose_cover(u, v, z(40), z(50), 40, 50, 2)
    assert rows_moved(u, v, pull) <= 40
    # two hubs + leaves on both sides: the cover beats both pure forms
    u = torch.cat([torch.arange(10), z(10) + 10, torch.tensor([11])])
    v = torch.cat([z(10), torch.arange(1, 11), torch.tensor([11])])
    pull = choose_cover(u, v, z(12), z(12), 12, 12, 2)
    assert rows_moved(u, v, pull) == 3


@pytest.mark.parametrize('overlap', [True, False])
def test_sharded_exchange_bf16_wire_is_within_its_stated_bound(overlap):
    """COLDBREW_HALO_WIRE=bf16 (opt-in): halo rows cross the links as bfloat16; result within the bf16-rounding bound of the
    unsharded oracle (and measurably different from it: the rounding is on the wire, not a no-op)."""
    _run(_worker, 2, 'case_graph_powerlaw_d7_d64', 'halo', overlap, 'edges', 'bf16')


def test_partition_bookkeeping():
    from gnn_tail_generalization_amd.dist import Partition
    for n, w in [(10, 3), (8, 2), (7, 8), (1000001, 8)]:
        parts = [Partition(n, w, r) for r in range(w)]
        assert parts[0].lo() == 0 and parts[-1].hi() == n
        assert all(parts[i].hi() == parts[i + 1].lo() for i in range(w - 1))
        assert sum(p.n_local for p in parts) == n and parts[0].padded >= n
        ids = torch.arange(n)
        own = parts[0].owner(ids)
        for p in parts:
            assert (own[p.lo():p.hi()] == p.rank).all()


def test_edge_balanced_partition():
    """Boundaries from the prefix sum over in-degree (+ a per-node weight): on a power-law degree vector every rank's share of
    sum(deg + w) is within one row's cost of the mean, while equal-row blocks are far off; owners are consistent."""
    from gnn_tail_generalization_amd.dist import Partition
    gen = torch.Generator().manual_seed(0)
    deg = (torch.rand(200000, generator=gen) ** -0.8).to(torch.int64)          # heavy tail, ids NOT permuted:
    deg, _ = torch.sort(deg, descending=True)                                  # hubs first (what real ogbn ids can look like)
    for world in (2, 4, 8):
        parts = [Partition.balanced(deg, world, r, node_weight=12) for r in range(world)]
        assert parts[0].lo() == 0 and parts[-1].hi() == deg.numel() and parts[0].kind == 'edges'
        assert all(parts[i].hi() == parts[i + 1].lo() for i in range(world - 1))
        cost = deg + 12
        shares = torch.tensor([int(cost[p.lo():p.hi()].sum()) for p in parts], dtype=torch.float64)
        mean = float(cost.sum()) / world
        assert float((shares - mean).abs().max()) <= float(cost.max()) + 1
        eq = torch.tensor([int(cost[Partition(deg.numel(), world, r).lo():Partition(deg.numel(), world, r).hi()].sum())
                           for r in range(world)], dtype=torch.float64)
        assert float(eq.max()) > 1.15 * mean                                  # the equal-row partition is badly unbalanced here
        own = parts[0].owner(torch.arange(deg.numel()))
        for p in parts:
            assert (own[p.lo():p.hi()] == p.rank).all()
    empty = Partition.balanced(torch.zeros(3, dtype=torch.int64), 8, 0)        # more ranks than rows: empty blocks are legal
    assert empty.bounds[-1] == 3 and sum(empty.hi(p) - empty.lo(p) for p in range(8)) == 3


def _norm_worker(rank, world, port, _unused, q):
    _setup(rank, world, port)
    try:
        import coldbrew_oracle as orc
        from dist_cpu_compute import OraclePrims
        from gnn_tail_generalization_amd import norms_hip
        from gnn_tail_generalization_amd.dist import Partition, allreduce_grads
        norms_hip.PRIMS = OraclePrims()
        n, d = 101, 12
        gen = torch.Generator().manual_seed(3)
        x = torch.randn(n, d, generator=gen) * 2 + 0.5
        gout = torch.randn(n, d, generator=gen)
        part = Partition(n, world, rank, bounds=[0, 17, n] if world == 2 else None)      # uneven blocks
        bn = torch.nn.BatchNorm1d(d)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(d, generator=gen) + 0.5)
            bn.bias.copy_(torch.randn(d, generator=gen))
        bn_ref = torch.nn.BatchNorm1d(d)
        bn_ref.load_state_dict(bn.state_dict())
        for kind in ('batch', 'pair', 'mean'):
            xl = part.slice_rows(x).clone().requires_grad_(True)
            xr = x.clone().requires_grad_(True)
            with norms_hip.row_sharding(None, n):
                if kind == 'batch':
                    bn.train()
                    yl = norms_hip.batch_norm(bn, xl)
                elif kind == 'pair':
                    yl = norms_hip.pair_norm(xl)
                else:
                    yl = norms_hip.mean_norm(xl)
                (yl * part.slice_rows(gout)).sum().backward()
            if kind == 'batch':
                allreduce_grads([bn.weight, bn.bias])
                bn_ref.train()
                yr = bn_ref(xr)
            elif kind == 'pair':
                yr = orc.pair_norm(xr)
            else:
                yr = orc.mean_norm(xr)
            (yr * gout).sum().backward()
            torch.testing.assert_close(yl.detach(), part.slice_rows(yr.detach()), atol=2e-5, rtol=1e-5, msg=lambda m: f'{kind}: {m}')
            torch.testing.assert_close(xl.grad, part.slice_rows(xr.grad), atol=2e-5, rtol=1e-4, msg=lambda m: f'{kind} dx: {m}')
            if kind == 'batch':
                torch.testing.assert_close(bn.weight.grad, bn_ref.weight.grad, atol=2e-5, rtol=1e-4)
                torch.testing.assert_close(bn.bias.grad, bn_ref.bias.grad, atol=2e-5, rtol=1e-4)
                torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, atol=1e-6, rtol=1e-5)
                torch.testing.assert_close(bn.running_var, bn_ref.running_var, atol=1e-6, rtol=1e-5)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_column_statistic_norms_span_all_ranks(world):
    """BatchNorm1d / PairNorm / MeanNorm (norm_tricks.py:20-41,132) on row shards == the unsharded norm: outputs, input gradients,
    affine-parameter gradients (after the replicated-gradient all-reduce) and running statistics."""
    _run(_norm_worker, world, None)


def _init_worker(rank, world, port, _unused, q):
    _setup(rank, world, port)
    try:
        import contextlib
        import io
        from gnn_tail_generalization_amd.base_options import BaseOptions
        from gnn_tail_generalization_amd.dist import Partition, sync_initial_state
        from gnn_tail_generalization_amd.GNN_model import TeacherGNN
        from gnn_tail_generalization_amd.utils import set_arch_configs
        n = 1000                                                    # 1000 % 3 != 0: the last block is shorter
        part = Partition(n, world, rank)
        with contextlib.redirect_stdout(io.StringIO()):
            args = BaseOptions().get_arguments(['--dataset=S-pubmed', '--whetherHasSE=111', '--num_layers=2', '--dim_learnable_input=16',
                                                '--manual_assign_GPU=0'])
        set_arch_configs(args)
        args.N_nodes = part.n_local
        torch.manual_seed(args.random_seed)
        model = TeacherGNN(args)                                    # parameters only; no compute on CPU
        before = torch.cat([p.detach().reshape(-1) for k, p in model.named_parameters() if not k.endswith('.le') and k != 'embs'])
        sync_initial_state(model, part, None, args.random_seed)
        rep = torch.cat([p.detach().reshape(-1) for k, p in model.named_parameters() if not k.endswith('.le') and k != 'embs'])
        got = [torch.zeros_like(rep) for _ in range(world)]
        dist.all_gather(got, rep)
        assert all(torch.equal(got[0], t) for t in got), 'replicated weights differ across ranks after sync_initial_state'
        pre = [torch.zeros_like(before) for _ in range(world)]
        dist.all_gather(pre, before)
        differs = not all(torch.equal(pre[0], t) for t in pre)      # the hazard the sync removes (block sizes 334 / 334 / 332)
        le = model.model.model.layers_GCN[0].le.detach()
        assert le.shape[0] == part.n_local and model.embs.shape[0] == part.n_local
        head = [torch.zeros(4, le.shape[1]) for _ in range(world)]
        dist.all_gather(head, le[:4].contiguous())
        assert not torch.equal(head[0], head[1]), 'per-node tables must not repeat across ranks'
        assert float(model.embs.detach().abs().max()) < 0.01        # GNN_normalizations.py:18-22 scale kept
        q.put((rank, 'ok' if (differs or world == 1) else 'FAIL the unsynchronised init was already consistent: test lost its teeth'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL ' + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def test_fresh_sharded_init_is_consistent_across_ranks():
    """with whetherHasSE != 000 and N % world != 0 the ranks draw different amounts of RNG for their
    per-node tables, so the replicated weights drawn afterwards differ — sync_initial_state must make them identical."""
    _run(_init_worker, 3, None)
