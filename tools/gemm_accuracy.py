"""Error of the GEMM kernels against an fp64 reference (run once per mode: default three-limb bf16 path, and with
CB_GEMM_PLAIN_F32=1 for the fp32-input MFMA path); torch.matmul (hipBLASLt fp32) is printed beside them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402


def main():
    dev = 'cuda:0'
    torch.manual_seed(0)
    mode = 'plain fp32 MFMA' if os.environ.get('CB_GEMM_PLAIN_F32') else 'three-limb bf16 MFMA'
    for M, K, N in [(20000, 256, 256), (20000, 128, 256), (20000, 256, 40), (5000, 1024, 128)]:
        for dist in ('uniform', 'normal*exp'):
            if dist == 'uniform':
                a, b = torch.rand(M, K, device=dev) - 0.5, torch.rand(K, N, device=dev) - 0.5
            else:   # wide dynamic range
                a = torch.randn(M, K, device=dev) * torch.exp(3 * torch.randn(M, K, device=dev))
                b = torch.randn(K, N, device=dev) * torch.exp(3 * torch.randn(K, N, device=dev))
            ref = a.double() @ b.double()
            scale = (a.double().abs() @ b.double().abs())          # sum |a||b|: the natural error scale of a dot product
            for name, fn in [(mode, lambda: gemm.mm_nn(a, b)), ('torch.matmul', lambda: a @ b)]:
                err = ((fn().double() - ref).abs() / scale).max().item()
                print(f'NN {M}x{K}x{N} {dist:10s} {name:22s}: max |err| / sum|a||b| = {err:.3e}  ({err / 2 ** -24:.2f} ulp-equivalents)')
            g = torch.randn(M, N, device=dev)
            ref = a.double().t() @ g.double()
            scale = a.double().abs().t() @ g.double().abs()
            for name, fn in [(mode, lambda: gemm.mm_tn(a, g)), ('torch.matmul', lambda: a.t() @ g)]:
                err = ((fn().double() - ref).abs() / scale).max().item()
                print(f'TN {M}: {K}x{N} {dist:10s} {name:22s}: max |err| / sum|a||g| = {err:.3e}  ({err / 2 ** -24:.2f} ulp-equivalents)')


if __name__ == '__main__':
    main()
