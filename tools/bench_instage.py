"""Times the input Linear's weight gradient with the trunk's input stage computed in its staging (cb_gemm_tn_instage_f32) against the separate pass +
the plain weight gradient (cb_trunk_input_bwd_multi_f32 on ONE folded operand + cb_gemm_tn_gdrop_f32) at the headline size.
usage: python tools/bench_instage.py [--rows N] [--feats F]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gnn_tail_generalization_amd import gemm, trunk  # noqa: E402
from bench_trunk_bwd import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=10_000_000)
    ap.add_argument('--feats', type=int, default=128)
    a = ap.parse_args()
    dev, n, f = 'cuda:0', a.rows, a.feats
    g = torch.randn(n, 256, device=dev)
    m = torch.randn(n, 256, device=dev)
    x = torch.rand(n, f, device=dev)
    bits = torch.randint(-2 ** 62, 2 ** 62, (n, 1, 4), dtype=torch.int64, device=dev)
    t = timed(lambda: gemm.mm_tn_instage(g, m, bits, x, 0.1, 5, 0.1, 6, 0))
    byts = (2 * n * 256 + n * f) * 4 + n * 32
    print(f'gemm_tn_instage            {t:7.3f} ms   {byts / t / 1e6:7.1f} GB/s   {2 * n * 256 * f * 6 / t / 1e9:7.1f} TF/s (bf16 limb products)')
    x0 = torch.empty(0, device=dev)
    t1 = timed(lambda: trunk._input_bwd_multi(g, 5, [m], [7], 1.0, x0, 0.1, 0, act_bits=bits, mix_pos=[None]))
    gy = trunk._input_bwd_multi(g, 5, [m], [7], 1.0, x0, 0.1, 0, act_bits=bits, mix_pos=[None])[0]
    t2 = timed(lambda: gemm.mm_tn_gdrop(gy, x, 0.1, 6, 0))
    print(f'separate pass (1 operand)  {t1:7.3f} ms  + weight gradient {t2:7.3f} ms')


if __name__ == '__main__':
    main()
