"""What one rank of a P-way node-sharded run does per aggregation, measured on ONE MI355X (no multi-GPU box needed): builds rank r's
row block of the synthetic graph as dist.ShardedGraph does — edge-balanced partition, interior / halo column split, the exchange plan
INCLUDING the owner side (what every other rank asks of rank r) — times the LOCAL kernels of the sliced pipeline with HIP events at
d = 256 and turns the rows on the busiest peer link into predicted link times.

Two pipelines are probed side by side (--mode both):
  pull   round 3's: pull-only halo plan cut by owner row chunk, row-chunked producers (layer GEMM chunk k -> pack k -> send k), interior
         pass, per-slice halo passes (cb_spmm_csr_acc_f32), the last one with the epilogue; the layer GEMM is a kernel of its own.
  cover  this round's default: push / pull vertex cover per rank pair (dist.choose_cover; pack = aggregation over the send CSR),
         positional slices, and the LAST halo pass is the aggregation + GEMM kernel on top of the running sums (cb_spmm_gemm_f32 /
         cb_spmm_gemm_fused_f32 / cb_spmm_gemm_trunkbwd_f32 with acc_init) — the next stage's matrix exists when that kernel ends.

    python tools/shard_probe.py [--name S-pl10M] [--worlds 1,2,4,8] [--rank 0] [--slices 4] [--link-gbs 153] [--link-eff 0.8]
                                [--n1-ms <the driver's 1-GPU ms/step>] [--mode both|pull|cover] [--halo-only 1]

Prediction model (stated, not hidden): every ordered peer pair has its own xGMI link (7 links x 153 GB/s per GPU, full duplex), traffic
to different peers moves in parallel, a slice's link time = max over peers of (bytes on that link) / (link_gbs * link_eff), slices queue
on the links in order.  Timeline of one aggregation (dist.py):
    compute stream:  [producer chunk k ->] pack k for k < K, interior pass, then halo pass k as soon as slice k has landed
    links:           slice k starts when pack k is done and slice k-1 has left
Step = D_rest / P + sum over the 2L aggregations of (end of the last pass - producers' own time).  D_rest = what a 1-GPU step spends
outside its aggregation (+ fused GEMM) launches, taken from the 1-GPU bench line: `pull` keeps the L layer GEMMs + L-1 dX GEMMs + trunk
backward passes in it (--dense-pull-ms), `cover` has them inside the timed last passes (--dense-cover-ms).  Efficiency is computed against
--n1-ms, the DRIVER's 1-GPU step, not against the probe's own P = 1 model.  The tiny all-reduces are ignored."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import dist as cbdist  # noqa: E402
from gnn_tail_generalization_amd import _lib, gemm, trunk  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import weight_image  # noqa: E402


def timed(fn, iters=4):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))


def pipeline(prod, pack, link, interior, passes):
    """End time of the last halo pass for per-slice producer / pack / link / pass times (ms)."""
    K = len(pack)
    t, sent_at, link_free = 0.0, [], 0.0
    for k in range(K):
        t += prod[k] + pack[k]
        start = max(t, link_free)
        link_free = start + link[k]
        sent_at.append(link_free)
    t += interior
    for k in range(K):
        t = max(t, sent_at[k]) + passes[k]
    return t


def positional(n_items, K):
    """Slice of every position of a list of n_items cut by dist.slice_weights(K) (CoverPlan's cut)."""
    w = cbdist.slice_weights(K)
    tot, acc, cuts = sum(w), 0.0, [0]
    for k in range(K - 1):
        acc += w[k]
        cuts.append(min(n_items, int(n_items * acc / tot)))
    cuts.append(n_items)
    return cuts


def pair_cover(u, v):
    """choose_cover on ONE ordered rank pair: u / v = source / destination ids of its remote edges.  Returns (pull mask, pulled ids,
    pushed ids, item of every edge in the pair's interleaved list, list length)."""
    dev = u.device
    uu, ui = torch.unique(u, return_inverse=True)
    vv, vi = torch.unique(v, return_inverse=True)
    z = lambda n: torch.zeros(n, dtype=torch.int64, device=dev)      # noqa: E731
    pull = cbdist.choose_cover(ui, vi, z(uu.numel()), z(vv.numel()), int(uu.numel()), int(vv.numel()), 1)
    S = torch.zeros(uu.numel(), dtype=torch.bool, device=dev)
    S[ui[pull]] = True
    T = torch.zeros(vv.numel(), dtype=torch.bool, device=dev)
    T[vi[~pull]] = True
    su, tv = torch.nonzero(S).reshape(-1), torch.nonzero(T).reshape(-1)
    # interleave the two kinds in proportion (CoverPlan): position = rank of (relative rank inside the kind)
    frac = torch.cat([(torch.arange(su.numel(), device=dev).double() + 0.5) / max(su.numel(), 1),
                      (torch.arange(tv.numel(), device=dev).double() + 0.5) / max(tv.numel(), 1)])
    order = torch.sort(frac, stable=True)[1]
    pos = torch.empty_like(order)
    pos[order] = torch.arange(order.numel(), device=dev)
    u_item = torch.full((max(int(uu.numel()), 1),), -1, dtype=torch.int64, device=dev)
    u_item[su] = pos[:su.numel()]
    t_item = torch.full((max(int(vv.numel()), 1),), -1, dtype=torch.int64, device=dev)
    t_item[tv] = pos[su.numel():]
    item = torch.where(pull, u_item[ui], t_item[vi])
    return pull, int(uu.numel()), int(su.numel()), int(tv.numel()), item, int(order.numel())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='S-pl10M')
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--slices', type=int, default=4)
    ap.add_argument('--link-gbs', type=float, default=153.0)
    ap.add_argument('--link-eff', type=float, default=0.8)
    ap.add_argument('--n1-ms', type=float, default=195.7, help="the DRIVER's 1-GPU ms/step (BENCH_r03.json: 195.7): efficiency = n1 / (P * step)")
    ap.add_argument('--dense-pull-ms', type=float, default=113.0, help='1-GPU step outside the six plain aggregations (pull pipeline: GEMMs stay kernels of their own)')
    ap.add_argument('--dense-cover-ms', type=float, default=72.0,
                    help='1-GPU step outside the aggregation (+ GEMM) launches: 193.5 ms - 5 x 20.8 (fused, incl. hub kernels) - 17.3 (plain)')
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--mode', default='both', choices=['both', 'pull', 'cover'])
    ap.add_argument('--halo-only', type=int, default=0, help='1: only count rows per link (pull vs cover), no kernel timing')
    ap.add_argument('--support-level', type=int, default=-1,
                    help='j >= 0: probe level j of the row-sparse backward (dist.ShardedGraph.support_orients): only the edges whose gathered row lies in the '
                         'support S_j of the stand-in\'s train mask (S_0 = train rows, S_{j+1} = rows with a neighbour in S_j); the partition stays the full graph\'s')
    ap.add_argument('--compact', type=int, default=0,
                    help='with --support-level j: the level COMPACT in the rank\'s rows as dist.ShardedGraph.support_levels builds it (round 5) — the rank reads '
                         'a [|S_j in block|, d] matrix and writes [|S_{j+1} in block|, d] (all rows when S_{j+1} exceeds 70 %% of the nodes): CSR rows / interior '
                         'columns / send lists renumbered, last pass, store backward and packs timed at those shapes')
    ap.add_argument('--merged', type=int, default=1,
                    help='cover pipeline: 1 = the interior pass and the first halo slice as ONE pass over [local | slice 0] (dist._Orientation.first, round 5); '
                         '0 = two passes (round 4)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    data = synthetic_data(a.name, seed=0, device=dev)
    n, E = int(data.x.shape[0]), int(data.edge_index.shape[1])
    src, dst = data.edge_index[0], data.edge_index[1]
    in_deg = torch.bincount(dst, minlength=n)
    if a.support_level >= 0:
        S = data.train_mask.clone()
        for _ in range(a.support_level):
            nxt = torch.zeros_like(S)
            nxt[dst[S[src]]] = True
            S = nxt
        keep = S[src]
        print(f'support level {a.support_level}: |S| = {float(S.float().mean()):.3f} N, {float(keep.float().mean()):.3f} of the edges kept', flush=True)
        src, dst = src[keep], dst[keep]
        E = int(src.numel())
        S_src, S_dst = None, None
        if a.compact:
            S_src = S
            nxt = torch.zeros_like(S)
            nxt[dst] = True
            share = float(nxt.float().mean())
            S_dst = nxt if share <= 0.7 else None
            print(f'compact level: reads |S_j| = {float(S.float().mean()):.3f} N rows, writes ' + (f'|S_j+1| = {share:.3f} N rows' if S_dst is not None else f'all rows (|S_j+1| = {share:.3f} N)'), flush=True)
        del keep
    del data
    comp = cbdist.HipCompute()
    K, d, L = max(1, a.slices), 256, a.layers
    link = a.link_gbs * a.link_eff * 1e6        # bytes per ms
    bpr = d * 4
    out_rows = []
    for P in [int(w) for w in a.worlds.split(',')]:
        r = min(a.rank, P - 1)
        parts = [cbdist.Partition.balanced(in_deg, P, q) for q in range(P)]
        part = parts[r]
        lo, hi = part.lo(), part.hi()
        n_local = hi - lo
        m = (dst >= lo) & (dst < hi)
        rr, cc = dst[m] - lo, src[m]
        del m
        remote = (cc < lo) | (cc >= hi)
        # compact level (--compact): rows renumbered to positions in S_{j+1} of this block, local columns to positions in S_j of this block
        n_src_l, col_pos = n_local, None
        if a.support_level >= 0 and a.compact:
            col_pos = torch.cumsum(S_src[lo:hi], 0) - 1
            n_src_l = max(int(S_src[lo:hi].sum()), 1)
            if S_dst is not None:
                row_pos = torch.cumsum(S_dst[lo:hi], 0) - 1
                rr = row_pos[rr]
                n_full, n_local = n_local, max(int(S_dst[lo:hi].sum()), 1)      # (from here on n_local = the rows this rank WRITES)
        row = {'P': P, 'rank': r, 'rows_local': n_local, 'rows_read': n_src_l, 'edges_local': int(rr.numel()), 'edges_remote': int(remote.sum()), 'slices': K}
        h = torch.rand(n_src_l, d, device=dev)
        scale = torch.rand(n_local, device=dev)
        bias = torch.rand(d, device=dev)
        w = torch.rand(d, d, device=dev) * 0.06
        x0 = torch.rand(n_local, d, device=dev)
        ci_ = cc[~remote] - lo
        g_int = comp.csr(rr[~remote], col_pos[ci_] if col_pos is not None else ci_, n_local, n_src_l)
        row['edges_interior'] = g_int.E
        if P == 1:
            img = weight_image(w)
            g_int.norm_in = scale
            t_plain = timed(lambda: g_int.spmm(h, row_scale=scale, bias=bias, relu=True))
            t_fused = timed(lambda: trunk._fused_gemm_launch(g_int, h, bias, x0, 0.9, 0.1, 0.1, 7, img, scale, None))
            row.update({'aggregation_ms': t_plain, 'aggregation_plus_gemm_ms': t_fused})
            out_rows.append(row)
            print(json.dumps(row), flush=True)
            continue
        own_c = part.owner(cc)
        t_int = 0.0 if a.halo_only else timed(lambda: g_int.spmm(h))
        row['interior_ms'] = t_int
        # ============================ pull pipeline (round 3) ============================
        if a.mode in ('both', 'pull'):
            uniq, inv = torch.unique(cc[remote], return_inverse=True)
            bnd = [b for q in range(P) for b in cbdist.chunk_bounds(parts[q].lo(), parts[q].hi(), K)[:-1]] + [n]
            pos = torch.searchsorted(uniq, torch.tensor(bnd, dtype=torch.int64, device=dev))
            cnt = (pos[1:] - pos[:-1]).view(P, K)
            recv = cnt.t().tolist()
            mine = torch.tensor(cbdist.chunk_bounds(lo, hi, K), dtype=torch.int64, device=dev)
            send = [[0] * P for _ in range(K)]
            for q in range(P):
                if q == r:
                    continue
                mq = (dst >= parts[q].lo()) & (dst < parts[q].hi()) & (src >= lo) & (src < hi)
                p2 = torch.searchsorted(torch.unique(src[mq]), mine)
                for k in range(K):
                    send[k][q] = int(p2[k + 1] - p2[k])
                del mq
            link_rows = [max(max(recv[k]), max(send[k])) for k in range(K)]
            row['pull'] = {'halo_rows': int(uniq.numel()), 'send_rows': sum(sum(s) for s in send), 'max_link_rows': sum(link_rows)}
            if not a.halo_only:
                seg = torch.bucketize(torch.arange(uniq.numel(), device=dev), pos[1:], right=True)
                roff = torch.cumsum(cnt, 0) - cnt
                base = roff.reshape(-1) - pos[:-1]
                slice_of = (seg % K)[inv]
                slot = (torch.arange(uniq.numel(), device=dev) + base[seg])[inv]
                n_k = [sum(recv[k]) for k in range(K)]
                rrem = rr[remote]
                g_halo = [comp.csr(rrem[slice_of == k], slot[slice_of == k], n_local, max(n_k[k], 1)) for k in range(K)]
                chunks = [(int(mine[k]) - lo, int(mine[k + 1]) - lo) for k in range(K)]
                z = torch.empty(n_local, d, device=dev)
                t_gemm = timed(lambda: gemm.mm_nn(h, w, rowscale=scale, out=z))
                t_prod = [timed(lambda k=k: gemm.mm_nn(h[chunks[k][0]:chunks[k][1]], w, rowscale=scale[chunks[k][0]:chunks[k][1]],
                                                       out=z[chunks[k][0]:chunks[k][1]])) for k in range(K)]
                send_idx = [torch.randint(chunks[k][0], max(chunks[k][1], chunks[k][0] + 1), (sum(send[k]),), device=dev) for k in range(K)]
                t_pack = [timed(lambda k=k: comp.pack_rows(h, send_idx[k])) for k in range(K)]
                halos = [torch.rand(max(n_k[k], 1), d, device=dev) for k in range(K)]
                acc = g_int.spmm(h)
                t_pass = [timed(lambda k=k: g_halo[k].spmm(halos[k], row_scale=scale, bias=bias, relu=True, acc_init=acc)) if k == K - 1
                          else timed(lambda k=k: g_halo[k].spmm(halos[k], acc_init=acc, out=acc)) for k in range(K)]
                t_link = [link_rows[k] * bpr / link for k in range(K)]
                exposed = pipeline(t_prod, t_pack, t_link, t_int, t_pass) - sum(t_prod)
                pen = sum(t_prod) - t_gemm
                step = a.dense_pull_ms / P + 2 * L * (exposed + pen)
                row['pull'].update({'pack_ms': t_pack, 'halo_pass_ms': t_pass, 'gemm_chunk_ms': t_prod, 'gemm_whole_ms': t_gemm, 'link_ms': t_link,
                                    'exposed_ms': exposed, 'step_ms': step, 'steps_per_s': 1e3 / step, 'efficiency_vs_driver_n1': a.n1_ms / (P * step)})
                del g_halo, halos, acc, z, send_idx, seg, slot, slice_of, rrem
            del uniq, inv
            torch.cuda.empty_cache()
        # ============================ cover pipeline (this round) ============================
        if a.mode in ('both', 'cover'):
            # requester side: per owner q the pair graph (sources in q -> my destinations)
            e_rows, e_slot_local, e_slice, e_owner = [], [], [], []
            recv = [[0] * P for _ in range(K)]
            n_pull_only = n_pulled = n_pushed = 0
            rrem, crem, orem = rr[remote], cc[remote], own_c[remote]
            for q in range(P):
                if q == r:
                    continue
                mq = orem == q
                u, v = crem[mq], rrem[mq]
                pull, n_u, n_s, n_t, item, n_items = pair_cover(u, v)
                n_pull_only += n_u
                n_pulled += n_s
                n_pushed += n_t
                cuts = torch.tensor(positional(n_items, K), dtype=torch.int64, device=dev)
                # halo edges: every pulled edge, ONE edge per pushed destination
                keep = pull.clone()
                first = torch.zeros_like(pull)
                if (~pull).any():
                    pv = v[~pull]
                    _, inv_t = torch.unique(pv, return_inverse=True)
                    firsts = torch.zeros(int(inv_t.max()) + 1, dtype=torch.int64, device=dev).scatter_reduce_(
                        0, inv_t, torch.arange(pv.numel(), device=dev), 'amin', include_self=False)
                    idx_np = torch.nonzero(~pull).reshape(-1)
                    first[idx_np[firsts]] = True
                keep |= first
                it = item[keep]
                k_of = (it.unsqueeze(1) >= cuts[1:K]).sum(1) if K > 1 else torch.zeros_like(it)
                for k in range(K):
                    recv[k][q] = int(cuts[k + 1] - cuts[k])
                e_rows.append(v[keep])
                e_slot_local.append(it - cuts[k_of])
                e_slice.append(k_of)
                e_owner.append(torch.full_like(it, q))
            e_rows, e_slot_local, e_slice, e_owner = (torch.cat(t_) for t_ in (e_rows, e_slot_local, e_slice, e_owner))
            n_k = [sum(recv[k]) for k in range(K)]
            base = torch.tensor([[sum(recv[k][:q]) for q in range(P)] for k in range(K)], dtype=torch.int64, device=dev)
            e_slot = base[e_slice, e_owner] + e_slot_local
            # owner side: per requester q the pair graph (my sources -> q's destinations): rows of my send buffer
            s_rows, s_cols, s_slice, s_owner = [], [], [], []
            send = [[0] * P for _ in range(K)]
            for q in range(P):
                if q == r:
                    continue
                mq = (dst >= parts[q].lo()) & (dst < parts[q].hi()) & (src >= lo) & (src < hi)
                u, v = src[mq] - lo, dst[mq]
                del mq
                pull, n_u, n_s, n_t, item, n_items = pair_cover(u, v)
                cuts = torch.tensor(positional(n_items, K), dtype=torch.int64, device=dev)
                # send-CSR edges: ONE edge per pulled source, every pushed edge
                keep = ~pull
                if pull.any():
                    pu = u[pull]
                    _, inv_s = torch.unique(pu, return_inverse=True)
                    firsts = torch.zeros(int(inv_s.max()) + 1, dtype=torch.int64, device=dev).scatter_reduce_(
                        0, inv_s, torch.arange(pu.numel(), device=dev), 'amin', include_self=False)
                    idx_p = torch.nonzero(pull).reshape(-1)
                    keep[idx_p[firsts]] = True
                it = item[keep]
                k_of = (it.unsqueeze(1) >= cuts[1:K]).sum(1) if K > 1 else torch.zeros_like(it)
                for k in range(K):
                    send[k][q] = int(cuts[k + 1] - cuts[k])
                s_rows.append(it - cuts[k_of])
                s_cols.append(col_pos[u[keep]] if col_pos is not None else u[keep])
                s_slice.append(k_of)
                s_owner.append(torch.full_like(it, q))
            s_rows, s_cols, s_slice, s_owner = (torch.cat(t_) for t_ in (s_rows, s_cols, s_slice, s_owner))
            sbase = torch.tensor([[sum(send[k][:q]) for q in range(P)] for k in range(K)], dtype=torch.int64, device=dev)
            s_rows = sbase[s_slice, s_owner] + s_rows
            n_send = [sum(send[k]) for k in range(K)]
            link_rows = [max(max(recv[k]), max(send[k])) for k in range(K)]
            row['cover'] = {'rows_pull_only': n_pull_only, 'halo_rows': n_pulled + n_pushed, 'pulled': n_pulled, 'pushed': n_pushed,
                            'send_rows': sum(n_send), 'max_link_rows': sum(link_rows), 'halo_edges': int(e_rows.numel()), 'send_edges': int(s_rows.numel())}
            if not a.halo_only:
                g_halo = [comp.csr(e_rows[e_slice == k], e_slot[e_slice == k], n_local, max(n_k[k], 1)) for k in range(K)]
                g_first = None
                if a.merged and K > 1:
                    m0_ = e_slice == 0
                    g_first = comp.csr(torch.cat([rr[~remote], e_rows[m0_]]), torch.cat([col_pos[ci_] if col_pos is not None else ci_, n_src_l + e_slot[m0_]]),
                                       n_local, n_src_l + max(n_k[0], 1))
                g_send = [comp.csr(s_rows[s_slice == k], s_cols[s_slice == k], max(n_send[k], 1), n_src_l) for k in range(K)]
                del e_rows, e_slot, e_slice, e_owner, s_rows, s_cols, s_slice, s_owner, e_slot_local
                t_pack = [timed(lambda k=k: g_send[k].spmm(h)) for k in range(K)]
                halos = [torch.rand(max(n_k[k], 1), d, device=dev) for k in range(K)]
                acc = g_int.spmm(h)
                t_pass = [timed(lambda k=k: g_halo[k].spmm(halos[k], acc_init=acc, out=acc)) for k in range(K - 1)]
                t_int_cover = t_int
                if g_first is not None:      # merged: no interior pass of its own; the first pass waits for slice 0 and covers both
                    ext = torch.cat([h, halos[0]])
                    t_first = timed(lambda: g_first.spmm(ext, out=acc))
                    row['merged_first_pass_ms'], row['interior_plus_pass0_ms'] = t_first, t_int + t_pass[0]
                    t_pass[0], t_int_cover = t_first, 0.0
                    del ext
                gl = g_halo[K - 1]
                gl.norm_in, gl.row_offset = scale, 0
                img, img_t = weight_image(w), weight_image(w, transpose=True)
                bits = torch.randint(-2 ** 62, 2 ** 62, (n_local, 1, 4), dtype=torch.int64, device=dev)
                last = {
                    'plain_epilogue': timed(lambda: gl.spmm(halos[K - 1], row_scale=scale, bias=bias, relu=True, acc_init=acc.clone())),
                    'fused_store': timed(lambda: trunk._fused_launch(_lib.load(), gl, gl, halos[K - 1], acc.clone(), bias, x0, 0.9, 0.1, 0.1, 7, False)),
                    'fused_store_gemm': timed(lambda: trunk._fused_gemm_launch(gl, halos[K - 1], bias, x0, 0.9, 0.1, 0.1, 7, img, scale, None, g=gl,
                                                                                acc=acc.clone())),
                    'reverse_gemm': timed(lambda: gl.spmm_gemm(halos[K - 1], img_t, g_rowscale=scale, acc_init=acc.clone())),
                    'reverse_gemm_trunkbwd': timed(lambda: gl.spmm_gemm_trunkbwd(halos[K - 1], img_t, scale, bits, 0.9, 0.1, 7, 0, scale, True,
                                                                                  transpose=False, acc_init=acc.clone())),
                    'clone_only': timed(lambda: acc.clone()),
                }
                t_link = [link_rows[k] * bpr / link for k in range(K)]
                zero = [0.0] * K

                def exposed(last_ms):
                    return pipeline(zero, t_pack, t_link, t_int_cover, t_pass + [last_ms - last['clone_only']])
                # forward: layers 0 .. L-2 store + next GEMM, layer L-1 store only; backward: every layer the dX tail; the trunk backward of the
                # layer below stays a pass of its own (its time is part of --dense-cover-ms; in the tail's epilogue it measures slower:
                # 'reverse_gemm_trunkbwd' against 'reverse_gemm' + t_trunk_bwd)
                gb = torch.rand(n_local, d, device=dev)
                t_tb = timed(lambda: trunk._layer_bwd(gb, bits, scale, None, False, 0.1, 7, 0, 0.9, 0.1, True))
                last['trunk_bwd_pass'] = t_tb
                per = ([exposed(last['fused_store_gemm'])] * (L - 1) + [exposed(last['fused_store'])] + [exposed(last['reverse_gemm'])] * L)
                step = a.dense_cover_ms / P + sum(per)
                row['cover'].update({'pack_ms': t_pack, 'halo_pass_ms': t_pass, 'last_pass_ms': last, 'link_ms': t_link, 'exposed_ms_per_aggregation': per,
                                     'step_ms': step, 'steps_per_s': 1e3 / step, 'efficiency_vs_driver_n1': a.n1_ms / (P * step),
                                     'link_bound_step_ms': a.dense_cover_ms / P + 2 * L * sum(t_link)})
                del g_halo, g_send, halos, acc
            torch.cuda.empty_cache()
        if 'cover' in row and 'pull' in row:      # the plan dist.ShardedGraph adopts: the cover where it spares >= tuning.T.cover_min_gain of the rows
            row['adopted'] = 'cover' if row['cover']['halo_rows'] <= (1.0 - cbdist.T.cover_min_gain) * row['cover']['rows_pull_only'] else 'pull'
        out_rows.append(row)
        print(json.dumps(row), flush=True)
        del g_int, h, rr, cc, remote
        torch.cuda.empty_cache()
    print(f'\n{a.name} (N={n}, E={E}), d={d}, rank {a.rank} of each world, {K} slices, link {a.link_gbs} GB/s x {a.link_eff}, driver 1-GPU step {a.n1_ms} ms')
    print('| P | rows / rank | remote edges | pull: rows on the busiest link | cover: rows on the busiest link (pulled + pushed of all peers) | cover / pull |'
          + ('' if a.halo_only else ' pull: exposed / step / steps/s / eff. | cover: exposed (fwd+GEMM, fwd last, bwd+dX) / step / steps/s / eff. | link-bound step (cover) |'))
    print('|---|---|---|---|---|---|' + ('' if a.halo_only else '---|---|---|'))
    print('(adopted plan per world: ' + ', '.join(f"P={w_['P']}: {w_.get('adopted', '-')}" for w_ in out_rows if w_['P'] > 1) + ')')
    for w_ in out_rows:
        if w_['P'] == 1:
            print(f"| 1 | {w_['rows_local']} | 0 | - | - | - |" + ('' if a.halo_only else f" aggregation {w_['aggregation_ms']:.2f} ms, + GEMM {w_['aggregation_plus_gemm_ms']:.2f} ms | step {a.n1_ms} ms = {1e3 / a.n1_ms:.2f} steps/s (driver) | - |"))
            continue
        pl, cv = w_.get('pull'), w_.get('cover')
        line = (f"| {w_['P']} | {w_['rows_local']} | {w_['edges_remote']} | {pl['max_link_rows'] if pl else '-'} | "
                f"{cv['max_link_rows'] if cv else '-'} ({cv['pulled']} + {cv['pushed']}) | " + (f"{cv['max_link_rows'] / pl['max_link_rows']:.3f}" if pl and cv else '-') + ' |')
        if not a.halo_only:
            line += (f" {pl['exposed_ms']:.2f} / {pl['step_ms']:.1f} / {pl['steps_per_s']:.2f} / {pl['efficiency_vs_driver_n1']:.2f} |" if pl else ' - |')
            if cv:
                e = cv['exposed_ms_per_aggregation']
                line += (f" {e[0]:.2f}, {e[L - 1]:.2f}, {e[L]:.2f} / {cv['step_ms']:.1f} / {cv['steps_per_s']:.2f} / {cv['efficiency_vs_driver_n1']:.2f} | "
                         f"{cv['link_bound_step_ms']:.1f} |")
            else:
                line += ' - | - |'
        print(line)


if __name__ == '__main__':
    main()
