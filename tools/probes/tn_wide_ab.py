import os, sys, subprocess, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gnn_tail_generalization_amd import gemm
M = 10_000_000
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(1)
a = torch.rand(M, 256, device=dev, generator=g) - 0.5
b = torch.rand(M, 256, device=dev, generator=g) - 0.5
rs = torch.rand(M, device=dev, generator=g)
def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(it + 1)]
    ev[0].record()
    for i in range(it):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(it))
out = gemm.mm_tn(a, b, rowscale=rs)
print(os.environ.get('CB_GEMM_TN_WIDE', '0'), 'tn 256x256 K=10M', round(timed(lambda: gemm.mm_tn(a, b, rowscale=rs)), 3), 'ms', 'checksum', float(out.double().sum()), float(out.abs().max()))
torch.save(out.cpu(), f'/tmp/tn_out_{os.environ.get("CB_GEMM_TN_WIDE", "0")}.pt')
