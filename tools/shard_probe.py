"""What one rank of a P-way node-sharded run does per aggregation, measured on ONE MI355X (no multi-GPU box needed):
builds rank r's row block of the synthetic graph exactly as dist.ShardedGraph does (edge-balanced partition, interior /
halo column split, halo plan cut into K time slices by owner row chunk, incl. the exact send lists derived from every other
rank's request list), times the LOCAL kernels of the sliced pipeline with HIP events at d = 256 (row-chunk layer GEMM, pack per
slice — fp32 and bf16 out —, interior pass, per-slice halo passes chained through the running sums, the last one with the
epilogue) and turns the halo byte counts into predicted link times.

    python tools/shard_probe.py [--name S-pl10M] [--d 256] [--worlds 1,2,4,8] [--rank 0] [--slices 4] [--link-gbs 153] [--link-eff 0.8]
                                [--dense-ms <ms of the non-aggregation part of the 1-GPU step>] [--partition edges|rows|degree|greedy]

Prediction model (stated, not hidden): every ordered peer pair has its own xGMI link (7 links x 153 GB/s per GPU, full duplex),
all-to-all traffic to different peers moves in parallel, so a slice's link time = max over peers of (bytes on that link) /
(link_gbs * link_eff); slices queue on the links in order.  Timeline of one aggregation (dist.py):
    compute stream:  [producer chunk k -> pack k] for k < K, interior pass, then halo pass k as soon as slice k has landed
    links:           slice k starts when pack k is done and slice k-1 has left
The aggregation's EXPOSED time = end of the last halo pass - (time the producers alone would have taken), i.e. what the step
pays on top of its dense part; step = dense part / P + 2L exposed aggregations; the tiny all-reduces are ignored.
K = 1 without producers is round 2's two-pass form (pack + max(interior, exchange) + halo pass)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import dist as cbdist  # noqa: E402
from gnn_tail_generalization_amd import gemm  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402


def timed(fn, iters=5):
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


def relabel(kind, src, dst, n, in_deg, P):
    """Internal relabelling (new id per node) evaluated for halo volume: 'degree' = ids sorted by in-degree (descending);
    'greedy' = streaming LDG-style assignment (nodes in descending-degree order go to the block holding most of their already
    placed neighbours, weighted by the remaining capacity), blocks then laid out contiguously.  Returns perm with new_id = perm[old]."""
    dev = src.device
    if kind == 'degree':
        order = torch.argsort(in_deg, descending=True, stable=True)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        perm[order] = torch.arange(n, device=dev)
        return perm
    # greedy, vectorised in rounds: nodes are processed in R batches of descending degree; a batch is scored against the labels
    # assigned so far (a streaming heuristic with a batch-sized look-back error)
    order = torch.argsort(in_deg, descending=True, stable=True)
    label = torch.full((n,), -1, dtype=torch.int64, device=dev)
    cap = float(int(in_deg.sum()) + 12 * n) / P * 1.02
    load = torch.zeros(P, dtype=torch.float64, device=dev)
    R = 64
    bsz = (n + R - 1) // R
    cost = (in_deg + 12).to(torch.float64)
    for b in range(R):
        batch = order[b * bsz:(b + 1) * bsz]
        if batch.numel() == 0:
            break
        inb = torch.zeros(n, dtype=torch.bool, device=dev)
        inb[batch] = True
        m = inb[dst] & (label[src] >= 0)
        loc = torch.full((n,), -1, dtype=torch.int64, device=dev)
        loc[batch] = torch.arange(batch.numel(), device=dev)
        score = torch.zeros((batch.numel(), P), dtype=torch.float32, device=dev)
        score.index_put_((loc[dst[m]], label[src[m]]), torch.ones(int(m.sum()), device=dev), accumulate=True)
        score = (score + 1e-3) * (1.0 - load / cap).clamp(min=0).to(torch.float32).unsqueeze(0)
        choice = score.argmax(1)
        label[batch] = choice
        load += torch.zeros(P, dtype=torch.float64, device=dev).index_add_(0, choice, cost[batch])
        del inb, m, loc, score
    order2 = torch.argsort(label * n + torch.arange(n, device=dev))          # blocks contiguous, original order inside a block
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    perm[order2] = torch.arange(n, device=dev)
    return perm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='S-pl10M')
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--slices', type=int, default=4)
    ap.add_argument('--link-gbs', type=float, default=153.0)
    ap.add_argument('--link-eff', type=float, default=0.8)
    ap.add_argument('--dense-ms', type=float, default=113.0, help='non-aggregation part of the 1-GPU step (GEMMs, elementwise, loss, Adam)')
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--partition', default='edges', choices=['edges', 'rows', 'degree', 'greedy'])
    ap.add_argument('--halo-only', type=int, default=0, help='1: only count halo rows / link bytes (partition study), no kernel timing')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    data = synthetic_data(a.name, seed=0, device=dev)
    n, E = int(data.x.shape[0]), int(data.edge_index.shape[1])
    src0, dst0 = data.edge_index[0], data.edge_index[1]
    del data
    in_deg0 = torch.bincount(dst0, minlength=n)
    comp = cbdist.HipCompute()
    K = max(1, a.slices)
    rows_out = []
    for P in [int(w) for w in a.worlds.split(',')]:
        r = min(a.rank, P - 1)
        if a.partition in ('degree', 'greedy') and P > 1:
            perm = relabel(a.partition, src0, dst0, n, in_deg0, P)
            src, dst = perm[src0], perm[dst0]
            in_deg = torch.bincount(dst, minlength=n)
            del perm
        else:
            src, dst, in_deg = src0, dst0, in_deg0
        balanced = a.partition != 'rows'
        parts = [cbdist.Partition.balanced(in_deg, P, q) if balanced else cbdist.Partition(n, P, q) for q in range(P)]
        part = parts[r]
        lo, hi = part.lo(), part.hi()
        m = (dst >= lo) & (dst < hi)
        rr, cc = dst[m] - lo, src[m]
        del m
        remote = (cc < lo) | (cc >= hi)
        uniq, inv = torch.unique(cc[remote], return_inverse=True)
        n_local, n_halo, e_local = hi - lo, int(uniq.numel()), int(rr.numel())
        # requester side: slices by owner row chunk
        bnd = [b for q in range(P) for b in cbdist.chunk_bounds(parts[q].lo(), parts[q].hi(), K)[:-1]] + [n]
        pos = torch.searchsorted(uniq, torch.tensor(bnd, dtype=torch.int64, device=dev))
        cnt = (pos[1:] - pos[:-1]).view(P, K)
        recv = cnt.t().tolist()                                                       # recv[k][q]
        # owner side: what every other rank asks of rank r, cut at rank r's chunk bounds
        mine = torch.tensor(cbdist.chunk_bounds(lo, hi, K), dtype=torch.int64, device=dev)
        send = [[0] * P for _ in range(K)]
        for q in range(P):
            if q == r:
                continue
            pq = parts[q]
            mq = (dst >= pq.lo()) & (dst < pq.hi()) & (src >= lo) & (src < hi)
            ids = torch.unique(src[mq])
            p2 = torch.searchsorted(ids, mine)
            for k in range(K):
                send[k][q] = int(p2[k + 1] - p2[k])
            del mq, ids
        link_rows = [max(max(recv[k]), max(send[k])) if P > 1 else 0 for k in range(K)]
        row = {'P': P, 'rank': r, 'partition': a.partition, 'slices': K, 'rows_local': n_local, 'edges_local': e_local,
               'halo_rows': n_halo, 'halo_frac_of_N': n_halo / n, 'send_rows': sum(sum(s) for s in send),
               'max_link_rows': sum(link_rows), 'edges_remote': int(remote.sum())}
        if a.halo_only:
            rows_out.append(row)
            print(json.dumps(row), flush=True)
            continue
        seg = torch.bucketize(torch.arange(n_halo, device=dev), pos[1:], right=True)
        roff = torch.cumsum(cnt, 0) - cnt
        base = roff.reshape(-1) - pos[:-1]
        slice_of = (seg % K)[inv]
        slot = (torch.arange(n_halo, device=dev) + base[seg])[inv] if n_halo else inv
        n_k = [sum(recv[k]) for k in range(K)]
        g_int = comp.csr(rr[~remote], cc[~remote] - lo, n_local, n_local)
        rrem = rr[remote]
        g_halo = [comp.csr(rrem[slice_of == k], slot[slice_of == k], n_local, max(n_k[k], 1)) for k in range(K)] if P > 1 else []
        row['edges_interior'] = g_int.E
        del rr, cc, remote, inv, seg, slot, slice_of, rrem
        d = a.d
        h = torch.rand(n_local, d, device=dev)
        scale = torch.rand(n_local, device=dev)
        bias = torch.rand(d, device=dev)
        w = torch.rand(d, d, device=dev)
        chunks = [(int(mine[k]) - lo, int(mine[k + 1]) - lo) for k in range(K)]
        t_int = timed(lambda: g_int.spmm(h))
        acc = g_int.spmm(h)
        if P == 1:
            t_whole = timed(lambda: g_int.spmm(h, row_scale=scale, bias=bias, relu=True))
            row.update({'single_gpu_aggregation_ms': t_whole, 'aggregation_exposed_ms': t_whole,
                        'step_ms_predicted': a.dense_ms + 2 * a.layers * t_whole, 'steps_per_s_predicted': 1e3 / (a.dense_ms + 2 * a.layers * t_whole)})
            row['steps_per_s_predicted_bf16_wire'] = row['steps_per_s_predicted']
            rows_out.append(row)
            print(json.dumps(row), flush=True)
            continue
        z = torch.empty(n_local, d, device=dev)
        t_gemm_whole = timed(lambda: gemm.mm_nn(h, w, rowscale=scale, out=z))
        t_prod = [timed(lambda k=k: gemm.mm_nn(h[chunks[k][0]:chunks[k][1]], w, rowscale=scale[chunks[k][0]:chunks[k][1]], out=z[chunks[k][0]:chunks[k][1]]))
                  for k in range(K)]
        send_idx = [torch.randint(chunks[k][0], max(chunks[k][1], chunks[k][0] + 1), (sum(send[k]),), device=dev) for k in range(K)]
        t_pack = [timed(lambda k=k: comp.pack_rows(h, send_idx[k])) for k in range(K)]
        t_pack16 = [timed(lambda k=k: comp.pack_rows(h, send_idx[k], 'bf16')) for k in range(K)]
        halos = [torch.rand(max(n_k[k], 1), d, device=dev) for k in range(K)]
        halos16 = [t.to(torch.bfloat16) for t in halos]

        def passes(bufs):
            out = []
            for k in range(K):
                if k == K - 1:
                    out.append(timed(lambda k=k: g_halo[k].spmm(bufs[k], row_scale=scale, bias=bias, relu=True, acc_init=acc)))
                else:
                    out.append(timed(lambda k=k: g_halo[k].spmm(bufs[k], acc_init=acc, out=acc)))
            return out
        t_pass, t_pass16 = passes(halos), passes(halos16)
        bpr = d * 4
        link = a.link_gbs * a.link_eff * 1e6        # bytes per ms
        t_link = [link_rows[k] * bpr / link for k in range(K)]
        t_link16 = [v / 2 for v in t_link]
        zero = [0.0] * K
        # exposed aggregation time = pipeline end - the producers' own time (already part of the dense share of the step)
        exp_f32 = pipeline(t_prod, t_pack, t_link, t_int, t_pass) - sum(t_prod)
        exp_f32_noprod = pipeline(zero, t_pack, t_link, t_int, t_pass)               # exchange starts only after the whole GEMM
        exp_bf16 = pipeline(t_prod, t_pack16, t_link16, t_int, t_pass16) - sum(t_prod)
        # chunking the GEMM costs something on its own: K launches instead of one
        gemm_penalty = sum(t_prod) - t_gemm_whole
        L2 = 2 * a.layers
        dense = a.dense_ms / P
        row.update({'pack_ms': t_pack, 'pack_bf16_ms': t_pack16, 'interior_ms': t_int, 'halo_pass_ms': t_pass, 'halo_pass_bf16_ms': t_pass16,
                    'gemm_chunk_ms': t_prod, 'gemm_whole_ms': t_gemm_whole, 'link_ms_predicted': t_link,
                    'aggregation_exposed_ms': exp_f32, 'aggregation_exposed_ms_without_chunked_producers': exp_f32_noprod,
                    'aggregation_exposed_ms_bf16_wire': exp_bf16, 'gemm_chunking_penalty_ms': gemm_penalty,
                    'step_ms_predicted': dense + L2 * (exp_f32 + gemm_penalty),
                    'steps_per_s_predicted': 1e3 / (dense + L2 * (exp_f32 + gemm_penalty)),
                    'steps_per_s_predicted_without_chunked_producers': 1e3 / (dense + L2 * exp_f32_noprod),
                    'steps_per_s_predicted_bf16_wire': 1e3 / (dense + L2 * (exp_bf16 + gemm_penalty))})
        rows_out.append(row)
        print(json.dumps(row), flush=True)
        del g_int, g_halo, h, halos, halos16, acc, z
        torch.cuda.empty_cache()
    if a.halo_only:
        print('\n| P | partition | rows/rank | edges/rank | remote edges | halo rows (x N) | rows on the busiest link |')
        print('|---|---|---|---|---|---|---|')
        for w_ in rows_out:
            print(f"| {w_['P']} | {w_['partition']} | {w_['rows_local']} | {w_['edges_local']} | {w_['edges_remote']} | {w_['halo_rows']} ({w_['halo_frac_of_N']:.3f}) | {w_['max_link_rows']} |")
        return
    base = rows_out[0]['steps_per_s_predicted'] if rows_out and rows_out[0]['P'] == 1 else None
    print(f'\n| P | rows / rank | edges / rank (interior) | halo rows (x N) | slices | pack (sum) | interior | link time (sum, pred.) | halo passes (sum; last) | '
          'GEMM chunks (sum; whole) | exposed per aggregation: fp32 wire / no chunked producers / bf16 wire | step (pred.) | steps/s (pred.) | efficiency | bf16 wire steps/s |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for w_ in rows_out:
        eff = f"{w_['steps_per_s_predicted'] / base / w_['P']:.2f}" if base else '-'
        if w_['P'] == 1:
            print(f"| 1 | {w_['rows_local']} | {w_['edges_local']} | - | - | - | - | - | - | - | {w_['aggregation_exposed_ms']:.2f} | {w_['step_ms_predicted']:.1f} | "
                  f"{w_['steps_per_s_predicted']:.2f} | 1.00 | - |")
            continue
        print(f"| {w_['P']} | {w_['rows_local']} | {w_['edges_local']} ({w_['edges_interior']}) | {w_['halo_rows']} ({w_['halo_frac_of_N']:.2f}) | {w_['slices']} | "
              f"{sum(w_['pack_ms']):.2f} | {w_['interior_ms']:.2f} | {sum(w_['link_ms_predicted']):.2f} | {sum(w_['halo_pass_ms']):.2f}; {w_['halo_pass_ms'][-1]:.2f} | "
              f"{sum(w_['gemm_chunk_ms']):.2f}; {w_['gemm_whole_ms']:.2f} | {w_['aggregation_exposed_ms']:.2f} / {w_['aggregation_exposed_ms_without_chunked_producers']:.2f} / "
              f"{w_['aggregation_exposed_ms_bf16_wire']:.2f} | {w_['step_ms_predicted']:.1f} | {w_['steps_per_s_predicted']:.2f} | {eff} | "
              f"{w_['steps_per_s_predicted_bf16_wire']:.2f} |")
    print(f'\nassumptions: {a.name} (N={n}, E={E}), d={a.d}, link {a.link_gbs} GB/s x {a.link_eff} efficiency per peer pair, '
          f'dense part {a.dense_ms} ms at P=1 scaled 1/P, {2 * a.layers} aggregations per step, rank {a.rank} of each world, partition {a.partition}')


if __name__ == '__main__':
    main()
