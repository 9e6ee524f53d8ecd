"""A few launches of the step's dominant GEMM shapes, for rocprofv3 --pmc / --kernel-trace runs.
usage: python tools/gemm_one.py M K N"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402

M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = torch.rand(M, K, device='cuda:0') - 0.5
b = torch.rand(K, N, device='cuda:0') - 0.5
g = torch.rand(M, N, device='cuda:0') - 0.5
rs = torch.rand(M, device='cuda:0')
for _ in range(3):
    gemm.mm_nn(a, b, rowscale=rs)
for _ in range(3):
    gemm.mm_tn(a, g, rowscale=rs)
if '--torch' in sys.argv:
    for _ in range(2):
        torch.matmul(a, b)
torch.cuda.synchronize()
