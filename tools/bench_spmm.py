"""Times the aggregation kernel alone on a synthetic power-law graph (HIP events on the launch stream).
usage: python tools/bench_spmm.py [--n 10000000] [--d 256] [--iters 10] [--T 256]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import CSRGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=10_000_000)
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--T', type=int, default=256)
    ap.add_argument('--name', default='S-pl10M')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    t0 = time.time()
    data = synthetic_data(a.name, seed=0, device=dev, n_override=(a.n if a.n != 10_000_000 else None))
    torch.cuda.synchronize()
    t1 = time.time()
    G = CSRGraph(data.edge_index, data.x.shape[0], hub_threshold=a.T)
    torch.cuda.synchronize()
    t2 = time.time()
    print(f'graph: N={G.N} E={G.E} symmetric={G.symmetric} max_in_deg={G.max_in_degree} hubs={G._plan.n_hubs} '
          f'chunks={G._plan.n_chunks} gen={t1 - t0:.2f}s csr_build={t2 - t1:.3f}s', flush=True)
    h = torch.rand(G.N, a.d, device=dev)
    out = torch.empty_like(h)
    bias = torch.rand(a.d, device=dev)
    for _ in range(2):
        G.spmm(h, row_scale=G.norm_in, bias=bias, relu=True, out=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        G.spmm(h, row_scale=G.norm_in, bias=bias, relu=True, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]
    b = G.algorithmic_bytes(a.d)
    best, avg = min(ms), sum(ms) / len(ms)
    print(f'spmm d={a.d}: avg {avg:.3f} ms  best {best:.3f} ms  algorithmic {b / 1e9:.2f} GB  '
          f'-> {b / avg / 1e6:.1f} GB/s avg ({b / avg / 1e6 / 8000:.3f} of 8 TB/s), {b / best / 1e6:.1f} GB/s best', flush=True)


if __name__ == '__main__':
    main()
