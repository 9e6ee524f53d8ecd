"""What one rank of a P-way node-sharded run does per aggregation, measured on ONE MI355X (no multi-GPU box needed):
builds rank r's row block of the synthetic graph exactly as dist.ShardedGraph does (edge-balanced partition, interior /
halo column split, halo plan incl. the exact send lists derived from every other rank's request list), times the local
kernels (row pack, interior pass, halo pass on top of the interior sums, single-pass form) with HIP events at d = 256 and
turns the halo byte counts into a predicted exchange time over the xGMI peer links.

    python tools/shard_probe.py [--name S-pl10M] [--d 256] [--worlds 1,2,4,8] [--rank 0] [--link-gbs 153] [--link-eff 0.8]
                                [--dense-ms <ms of the non-aggregation part of the 1-GPU step>]

Prediction model (stated, not hidden): every ordered peer pair has its own xGMI link (7 links x 153 GB/s per GPU, full duplex),
all-to-all traffic to different peers moves in parallel, so exchange_ms = max over peers of (bytes on that link) /
(link_gbs * link_eff).  Overlapped aggregation = pack + max(interior, exchange) + halo pass; single-pass = pack + exchange +
whole pass.  Step = dense part / P + 2L aggregations (+ 2L packs) ; the tiny all-reduces are ignored."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import dist as cbdist  # noqa: E402
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='S-pl10M')
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--link-gbs', type=float, default=153.0)
    ap.add_argument('--link-eff', type=float, default=0.8)
    ap.add_argument('--dense-ms', type=float, default=125.0, help='non-aggregation part of the 1-GPU step (GEMMs, elementwise, loss, Adam)')
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--partition', default='edges', choices=['edges', 'rows'])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    data = synthetic_data(a.name, seed=0, device=dev)
    n, E = int(data.x.shape[0]), int(data.edge_index.shape[1])
    src, dst = data.edge_index[0], data.edge_index[1]
    in_deg = torch.bincount(dst, minlength=n)
    comp = cbdist.HipCompute()
    rows_out = []
    for P in [int(w) for w in a.worlds.split(',')]:
        r = min(a.rank, P - 1)
        parts = [cbdist.Partition.balanced(in_deg, P, q) if a.partition == 'edges' else cbdist.Partition(n, P, q) for q in range(P)]
        part = parts[r]
        lo, hi = part.lo(), part.hi()
        m = (dst >= lo) & (dst < hi)
        rr, cc = dst[m] - lo, src[m]
        del m
        remote = (cc < lo) | (cc >= hi)
        uniq, inv = torch.unique(cc[remote], return_inverse=True)
        n_local, n_halo, e_local = hi - lo, int(uniq.numel()), int(rr.numel())
        recv_counts = torch.bincount(part.owner(uniq), minlength=P)[:P].tolist() if P > 1 else [0]
        # exact send lists: what every other rank asks of rank r
        send_counts = [0] * P
        for q in range(P):
            if q == r:
                continue
            pq = parts[q]
            mq = (dst >= pq.lo()) & (dst < pq.hi()) & (src >= lo) & (src < hi)
            send_counts[q] = int(torch.unique(src[mq]).numel())
            del mq
        g_int = comp.csr(rr[~remote], cc[~remote] - lo, n_local, n_local)
        g_halo = comp.csr(rr[remote], inv, n_local, max(n_halo, 1)) if P > 1 else None
        new_col = cc - lo
        new_col[remote] = n_local + inv
        g_whole = comp.csr(rr, new_col, n_local, n_local + n_halo)
        e_int = g_int.E
        del rr, cc, new_col, remote, inv
        h = torch.rand(n_local, a.d, device=dev)
        halo = torch.rand(max(n_halo, 1), a.d, device=dev)
        ext = torch.cat([h, halo[:n_halo]])
        scale = torch.rand(n_local, device=dev)
        bias = torch.rand(a.d, device=dev)
        send_idx = torch.randint(0, n_local, (sum(send_counts),), device=dev)
        t_pack = timed(lambda: comp.pack_rows(h, send_idx)) if P > 1 else 0.0
        t_int = timed(lambda: g_int.spmm(h))
        part_sums = g_int.spmm(h)
        t_halo = timed(lambda: g_halo.spmm(halo, row_scale=scale, bias=bias, relu=True, acc_init=part_sums)) if P > 1 else 0.0
        t_whole = timed(lambda: g_whole.spmm(ext, row_scale=scale, bias=bias, relu=True))
        # opt-in bf16 wire (COLDBREW_HALO_WIRE=bf16): narrowing of the packed rows + widening of the received ones, half the bytes
        if P > 1:
            packed = comp.pack_rows(h, send_idx)
            halo16 = halo[:max(n_halo, 1)].to(torch.bfloat16)
            t_cvt = timed(lambda: packed.to(torch.bfloat16)) + timed(lambda: halo16.float())
            del packed, halo16
        else:
            t_cvt = 0.0
        bpr = a.d * 4
        link = a.link_gbs * a.link_eff * 1e6        # bytes per ms
        t_xchg = max(max(recv_counts), max(send_counts)) * bpr / link if P > 1 else 0.0
        agg_overlap = t_pack + max(t_int, t_xchg) + t_halo if P > 1 else t_whole
        agg_single = t_pack + t_xchg + t_whole
        L2 = 2 * a.layers
        agg_bf16 = t_pack + t_cvt + max(t_int, t_xchg / 2) + t_halo if P > 1 else t_whole
        step_overlap = a.dense_ms / P + L2 * agg_overlap
        step_single = a.dense_ms / P + L2 * agg_single
        row = {'P': P, 'rank': r, 'partition': a.partition, 'rows_local': n_local, 'edges_local': e_local, 'edges_interior': e_int,
               'halo_rows': n_halo, 'halo_frac_of_N': n_halo / n, 'recv_bytes': n_halo * bpr, 'send_rows': sum(send_counts),
               'max_link_rows': max(max(recv_counts), max(send_counts)) if P > 1 else 0,
               'pack_ms': t_pack, 'interior_ms': t_int, 'halo_pass_ms': t_halo, 'single_pass_ms': t_whole,
               'exchange_ms_predicted': t_xchg, 'aggregation_ms_overlapped': agg_overlap, 'aggregation_ms_single_pass': agg_single,
               'step_ms_predicted_overlapped': step_overlap, 'step_ms_predicted_single_pass': step_single,
               'steps_per_s_predicted': 1e3 / min(step_overlap, step_single),
               'wire_convert_ms': t_cvt, 'aggregation_ms_bf16_wire': agg_bf16,
               'steps_per_s_predicted_bf16_wire': 1e3 / (a.dense_ms / P + L2 * agg_bf16)}
        rows_out.append(row)
        print(json.dumps(row), flush=True)
        del g_int, g_halo, g_whole, h, halo, ext, part_sums
        torch.cuda.empty_cache()
    base = rows_out[0]['steps_per_s_predicted'] if rows_out and rows_out[0]['P'] == 1 else None
    print('\n| P | rows/rank | edges/rank | interior edges | halo rows (x N) | pack ms | interior ms | exchange ms (pred.) | halo pass ms | '
          'single pass ms | aggregation ms overlapped / single | step ms (pred.) | steps/s (pred.) | efficiency |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for w in rows_out:
        eff = f"{w['steps_per_s_predicted'] / base / w['P']:.2f}" if base else '-'
        print(f"| {w['P']} | {w['rows_local']} | {w['edges_local']} | {w['edges_interior']} | {w['halo_rows']} ({w['halo_frac_of_N']:.2f}) | "
              f"{w['pack_ms']:.2f} | {w['interior_ms']:.2f} | {w['exchange_ms_predicted']:.2f} | {w['halo_pass_ms']:.2f} | {w['single_pass_ms']:.2f} | "
              f"{w['aggregation_ms_overlapped']:.2f} / {w['aggregation_ms_single_pass']:.2f} | "
              f"{min(w['step_ms_predicted_overlapped'], w['step_ms_predicted_single_pass']):.1f} | {w['steps_per_s_predicted']:.2f} | {eff} |")
    print('\nopt-in bf16 halo wire (outside the 1e-4 parity): ' + ', '.join(
        f"P={w['P']}: {w['steps_per_s_predicted_bf16_wire']:.2f} steps/s" + (f" ({w['steps_per_s_predicted_bf16_wire'] / base / w['P']:.2f})" if base else '')
        for w in rows_out))
    print(f'\nassumptions: {a.name} (N={n}, E={E}), d={a.d}, link {a.link_gbs} GB/s x {a.link_eff} efficiency per peer pair, '
          f'dense part {a.dense_ms} ms at P=1 scaled 1/P, {2 * a.layers} aggregations per step, rank {a.rank} of each world')


if __name__ == '__main__':
    main()
