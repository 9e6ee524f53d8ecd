"""Times the device-side graph analysis (csrc/cb_ingest.hip) on a synthetic graph next to the tensor formulation and prints
the reference's per-edge Python-loop cost extrapolated from a 200 000-edge sample of the same loop shape.
usage: python tools/bench_ingest.py [--name S-pl10M]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import utils  # noqa: E402
from gnn_tail_generalization_amd.data import Data, synthetic_data  # noqa: E402


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='S-pl10M')
    a = ap.parse_args()
    dev = 'cuda:0'
    data = synthetic_data(a.name, seed=0, device=dev)
    n, ei = data.x.shape[0], data.edge_index
    E = ei.shape[1]
    print(f'{a.name}: N={n} E={E}')
    print(f'degrees (graph_analyze)      HIP {t(lambda: utils._degrees_device(n, ei)):8.2f} ms   torch.bincount x2 '
          f'{t(lambda: (torch.bincount(ei[0], minlength=n), torch.bincount(ei[1], minlength=n))):8.2f} ms')
    _, d_in = utils._degrees_device(n, ei)
    arr = d_in.cpu().numpy()
    t0 = time.perf_counter()
    utils.get_partial_sorted_idx(arr, 'top3')
    host_ms = (time.perf_counter() - t0) * 1e3
    print(f'head/tail select (top3)      HIP {t(lambda: utils.partial_sorted_select_device(d_in, "top3")):8.2f} ms   numpy medians on the host {host_ms:8.2f} ms')
    _, z = utils.partial_sorted_select_device(d_in, 'top6')
    probe = Data(x=data.x[:, :1], edge_index=ei)
    probe.zero_deg_mask = z

    def craft():
        probe.edge_index = ei
        utils.craft_isolation_v2(probe, verbose=False)

    def craft_torch():
        drop = (ei[0] != ei[1]) & (z[ei[0]] | z[ei[1]])
        return ei[:, ~drop]

    print(f'craft_isolation_v2           HIP {t(craft):8.2f} ms   boolean indexing {t(craft_torch):8.2f} ms')
    half = ei[:, ei[0] <= ei[1]].contiguous()

    def sym_torch():
        key = torch.unique(torch.cat([half[0] * n + half[1], half[1] * n + half[0]]))
        return torch.stack([key // n, key % n])

    print(f'to_undirected ({half.shape[1]} in)  HIP {t(lambda: utils.to_undirected(half, n)):8.2f} ms   torch.unique {t(sym_torch):8.2f} ms')
    # the reference's loop shape (utils.py:300-334: dict get / update per edge) on a sample, extrapolated linearly
    m = 200_000
    sample = ei[:, :m].cpu().numpy()
    t0 = time.perf_counter()
    dd = {}
    for ie in range(m):
        o, d = sample[:, ie]
        if not dd.get(o):
            dd[o] = [0, 0]
        if not dd.get(d):
            dd[d] = [0, 0]
        dd[o][0] += 1
        dd[d][1] += 1
    per_edge = (time.perf_counter() - t0) / m
    print(f'reference-style per-edge Python loop: {per_edge * 1e6:.2f} us/edge on {m} edges -> {per_edge * E:.0f} s for graph_analyze alone at E={E} '
          f'(craft_isolation_v2 adds a second loop of the same shape)')


if __name__ == '__main__':
    main()
