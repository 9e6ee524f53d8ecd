"""Aggregation + dense tail in one kernel (cb_spmm_gemm_f32) against its two parts on the headline graph, HIP events.
usage: python tools/bench_agg_gemm.py [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import CSRGraph, weight_image  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return sum(ms) / len(ms), min(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--name', default='S-pl10M')
    ap.add_argument('--parts', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    data = synthetic_data(a.name, seed=0, device=dev)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    del data
    h = torch.rand(G.N, 256, device=dev)
    w = torch.rand(256, 256, device=dev) * 0.1
    img = weight_image(w)
    if a.parts:
        out = G.spmm(h)
        t_s = timed(lambda: G.spmm(h), a.iters)
        t_g = timed(lambda: gemm.mm_nn(out, w, rowscale=G.norm_out), a.iters)
        print(f'aggregation alone {t_s[0]:.2f} ms (best {t_s[1]:.2f}); GEMM alone {t_g[0]:.2f} ms (best {t_g[1]:.2f}); sum {t_s[0] + t_g[0]:.2f} ms', flush=True)
        del out
    t_f = timed(lambda: G.spmm_gemm(h, img, g_rowscale=G.norm_out), a.iters)
    print(f'aggregation + dense tail in one kernel (incl. hub kernels) {t_f[0]:.2f} ms (best {t_f[1]:.2f})', flush=True)


if __name__ == '__main__':
    main()
