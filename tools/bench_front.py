"""Forward front alone at the headline shape: cb_trunk_front_f32 (one kernel) against its two-kernel forms.
usage: python tools/bench_front.py [--rows 10000000] [--k 128] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return sum(ts) / len(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=10_000_000)
    ap.add_argument('--k', type=int, default=128)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only-front', type=int, default=0)
    a = ap.parse_args()
    dev = 'cuda:0'
    M, K, p = a.rows, a.k, 0.1
    x = torch.rand(M, K, device=dev)
    w_in = torch.randn(256, K, device=dev) * 0.1
    b_in = torch.randn(256, device=dev) * 0.1
    w0 = torch.randn(256, 256, device=dev) * 0.07
    rs = torch.rand(M, device=dev) + 0.5
    t = timed(lambda: gemm.trunk_front(x, w_in, b_in, w0, rs, None, p, 11, 12, 0, want_bits=True, want_drop=False), a.iters)
    print(f'front kernel (X0 + bits + Z0)            {t[0]:.2f} ms (best {t[1]:.2f})   dbg={os.environ.get("CB_FRONT_DBG", "0")}', flush=True)
    if a.only_front:
        return
    t = timed(lambda: gemm.trunk_front(x, w_in, b_in, w0, rs, None, p, 11, 12, 0, want_bits=True, want_drop=True), a.iters)
    print(f'front kernel + dropped copy stored        {t[0]:.2f} ms (best {t[1]:.2f})', flush=True)
    wt = w_in.t().contiguous()

    def two_copy():
        y, yd, _ = gemm.mm_nn_indrop_drop2(x, wt, p, 11, 12, 0, bias=b_in, relu=True, want_bits=True)
        return gemm.mm_nn(yd, w0, rowscale=rs)

    def two_nocopy():
        y, _ = gemm.mm_nn_indrop(x, wt, p, 11, 0, bias=b_in, relu=True, want_bits=True)
        return gemm.mm_nn_indrop(y, w0, p, 12, 0, rowscale=rs)
    t = timed(two_copy, a.iters)
    print(f'two kernels, dropped copy (round 3)       {t[0]:.2f} ms (best {t[1]:.2f})', flush=True)
    t = timed(two_nocopy, a.iters)
    print(f'two kernels, mask drawn while staging     {t[0]:.2f} ms (best {t[1]:.2f})', flush=True)


if __name__ == '__main__':
    main()
