"""Times pure label propagation (--train_which=LP: 50 steps of result <- clamp(0.5 D^-1/2 A D^-1/2 result + 0.5 y0, 0, 1),
Label_propagation_model/outcome_correlation.py:128-156) on the arxiv- and products-shaped synthetic graphs: every step is one
aggregation at width d = C on the same kernel as the teacher.  usage: python tools/bench_lp.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import ops  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import CSRGraph  # noqa: E402


def main():
    dev = 'cuda:0'
    for name in ('S-arxiv', 'S-products'):
        data = synthetic_data(name, seed=0, device=dev)
        n = data.x.shape[0]
        G = CSRGraph(data.edge_index, n)
        c = int(data.y.max()) + 1
        y0 = torch.zeros((n, c), device=dev)
        y0[data.train_mask] = torch.nn.functional.one_hot(data.y[data.train_mask], c).float()
        dis = G.in_degrees().float().pow(-0.5)
        dis[dis == float('inf')] = 0
        ops.label_propagation(G, y0, dis, 0.5, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.label_propagation(G, y0, dis, 0.5, 50)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        b = G.algorithmic_bytes(c, bias=False)
        print(f'{name}: N={n} E={G.E} d=C={c}: {dt * 1e3:.3f} ms per propagation step (aggregation + 2 elementwise passes), '
              f'aggregation bytes {b / 1e9:.2f} GB -> {b / dt / 1e9:.0f} GB/s over the whole step')


if __name__ == '__main__':
    main()
