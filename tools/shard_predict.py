"""Composite prediction of the node-sharded step from three tools/shard_probe.py runs (tools/probes/shard_r05.sh): the full graph (forward
aggregations and the dense last level of the backward), level 0 and level 1 of the row-sparse backward (compact in the rank's rows).

    step(P) = D_rest / P  +  (L - 1) x exposed(fwd store + GEMM) + exposed(fwd last)  +  exposed(level 0) + exposed(level 1) + exposed(full, bwd + dX)

D_rest = the 1-GPU step outside its aggregation launches (bench.py: ms_per_step - sum of roofline.per_family total_ms_per_step) — with the
row-sparse plan on one GPU AND, since round 5, on the shards (compact levels), so the same figure divides by P.  Efficiency against --n1-ms.

--rows-only 1 (the rows-only training forward on shards, dist.ShardedGraph.loss_rows_forward): the last layer's exchange is the TRANSPOSE of the backward's
level 0 (the same edges: loss rows <-> their neighbours) and is priced at that level's exposed time; the layer below it loses its dense tail:

    forward = (L - 2) x exposed(fwd store + GEMM) + exposed(fwd store) + exposed(level 0)

    python tools/shard_predict.py --dir profiles --prefix r05_shard_probe_S-pl10M --n1-ms 155.4 --drest-ms 52.8
    python tools/shard_predict.py --dir profiles --prefix r05_shard_probe_S-pl10M --n1-ms 128.8 --drest-ms 51.8 --rows-only 1"""
import argparse
import json
import os


def rows_of(path):
    out = {}
    for line in open(path):
        if line.startswith('{'):
            r = json.loads(line)
            out[r['P']] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default='profiles')
    ap.add_argument('--prefix', default='r05_shard_probe_S-pl10M')
    ap.add_argument('--n1-ms', type=float, default=155.4)
    ap.add_argument('--drest-ms', type=float, default=52.8)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--rows-only', type=int, default=0)
    a = ap.parse_args()
    full, l0, l1 = (rows_of(os.path.join(a.dir, f'{a.prefix}_{k}.txt')) for k in ('full', 'level0_compact', 'level1_compact'))
    L = a.layers
    fwd_head = f'{L - 2} x (store + GEMM) + store + last layer on the loss rows' if a.rows_only else f'{L - 1} x (store + GEMM) + last'
    print(f'| P | D_rest / P | forward: {fwd_head} | backward: level 0 + level 1 + full | step (pred.) | steps/s | efficiency vs {a.n1_ms} ms |')
    print('|---|---|---|---|---|---|---|')
    for P in sorted(p for p in full if p > 1):
        e = full[P]['cover']['exposed_ms_per_aggregation']
        b0, b1, b2 = l0[P]['cover']['exposed_ms_per_aggregation'][L], l1[P]['cover']['exposed_ms_per_aggregation'][L], e[L]
        if a.rows_only:
            fwd = (L - 2) * e[0] + e[L - 1] + b0
            fwd_txt = f'{L - 2} x {e[0]:.2f} + {e[L - 1]:.2f} + {b0:.2f} = {fwd:.1f}'
        else:
            fwd = (L - 1) * e[0] + e[L - 1]
            fwd_txt = f'{L - 1} x {e[0]:.2f} + {e[L - 1]:.2f} = {fwd:.1f}'
        step = a.drest_ms / P + fwd + b0 + b1 + b2
        print(f'| {P} | {a.drest_ms / P:.1f} | {fwd_txt} | {b0:.2f} + {b1:.2f} + {b2:.2f} = {b0 + b1 + b2:.1f} | {step:.1f} ms | '
              f'**{1e3 / step:.2f}** | {a.n1_ms / (P * step):.2f} |')


if __name__ == '__main__':
    main()
