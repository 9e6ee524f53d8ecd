"""Does an MFMA-bound TN GEMM overlap with the HBM-bound aggregation when issued on two streams?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm
from gnn_tail_generalization_amd.data import synthetic_data
from gnn_tail_generalization_amd.graph import CSRGraph
dev = 'cuda:0'
data = synthetic_data('S-pl10M', seed=0, device=dev)
G = CSRGraph(data.edge_index, data.x.shape[0])
n = G.N
h = torch.rand(n, 256, device=dev); out = torch.empty_like(h)
a = torch.rand(n, 256, device=dev); g = torch.rand(n, 256, device=dev); rs = torch.rand(n, device=dev)
s2 = torch.cuda.Stream()
def t(fn, it=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def spmm(): G.spmm(h, out=out)
def tn(): gemm.mm_tn(a, g, rowscale=rs)
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        gemm.mm_tn(a, g, rowscale=rs)
        ev2 = torch.cuda.Event(); ev2.record()
    G.spmm(h, out=out)
    torch.cuda.current_stream().wait_event(ev2)
print('spmm alone %.2f ms   tn alone %.2f ms   both on two streams %.2f ms' % (t(spmm), t(tn), t(both)))
