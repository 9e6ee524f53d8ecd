"""Times the SEMLP hand-off kernel (scores + top-K + combine) at ogbn-arxiv scale."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import ops
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 169343, int(sys.argv[2]) if len(sys.argv) > 2 else 768, 2
t = torch.randn(n, d, device='cuda:0'); q = torch.randn(n, d, device='cuda:0')
ops.se_topk_replace(q[:1024], t, k); torch.cuda.synchronize()
t0 = time.time(); out = ops.se_topk_replace(q, t, k); torch.cuda.synchronize(); dt = time.time() - t0
print(f'se_topk_replace B=N={n} D={d} K={k}: {dt*1e3:.1f} ms  {2*n*n*d/dt/1e12:.1f} TF/s')
