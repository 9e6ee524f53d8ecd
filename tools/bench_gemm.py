"""Times the MFMA GEMM kernels on the shapes of the TeacherGNN step (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402


def timeit(fn, iters=5):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    with_torch = '--torch' in sys.argv
    dev = 'cuda:0'
    for K, N, tag in [(256, 256, 'layer fwd / dX'), (128, 256, 'input Linear'), (256, 40, 'output Linear'), (40, 256, 'dX of output Linear')]:
        a = torch.rand(M, K, device=dev) - 0.5
        b = torch.rand(K, N, device=dev) - 0.5
        rs = torch.rand(M, device=dev)
        add = torch.rand(M, N, device=dev)
        bias = torch.rand(N, device=dev)
        cases = [('plain', lambda: gemm.mm_nn(a, b)), ('rowscale', lambda: gemm.mm_nn(a, b, rowscale=rs)),
                 ('rowscale+addend', lambda: gemm.mm_nn(a, b, rowscale=rs, addend=add)),
                 ('bias+relu', lambda: gemm.mm_nn(a, b, bias=bias, relu=True))]
        if with_torch:
            cases.append(('torch.matmul', lambda: torch.matmul(a, b)))
        for name, fn in cases:
            ms = timeit(fn)
            print(f'NN M={M} K={K} N={N} [{tag}] {name:16s}: {ms:8.3f} ms  {2 * M * K * N / ms / 1e9:7.1f} TF/s', flush=True)
        del add
    for K1, K2, tag in [(256, 256, 'dW layer'), (256, 128, 'dW input (H x F)'), (40, 256, 'dW output')]:
        a = torch.rand(M, K1, device=dev) - 0.5
        g = torch.rand(M, K2, device=dev) - 0.5
        rs = torch.rand(M, device=dev)
        cases = [('tn rowscale', lambda: gemm.mm_tn(a, g, rowscale=rs))]
        if with_torch:
            cases.append(('torch a.T@g', lambda: torch.matmul(a.t(), g)))
        for name, fn in cases:
            ms = timeit(fn)
            print(f'TN M={M} K1={K1} K2={K2} [{tag}] {name:12s}: {ms:8.3f} ms  {2 * M * K1 * K2 / ms / 1e9:7.1f} TF/s', flush=True)


if __name__ == '__main__':
    main()
