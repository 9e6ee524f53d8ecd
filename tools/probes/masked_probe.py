"""Reverse aggregation with the store backward applied to the gathered rows (cb_spmm_csr_masked_f32) against the two-kernel form
(cb_trunk_layer_bwd_f32 + cb_spmm_csr_f32) on S-pl10M, d = 256: max difference and time per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnn_tail_generalization_amd import trunk
from gnn_tail_generalization_amd.data import synthetic_data
from gnn_tail_generalization_amd.graph import CSRGraph

dev = 'cuda:0'
data = synthetic_data('S-pl10M', seed=0, device=dev)
G = CSRGraph(data.edge_index, data.x.shape[0])
del data
n, d = G.N, 256
gen = torch.Generator(device=dev).manual_seed(3)
z = torch.randn(n, d, device=dev, generator=gen)
bias = torch.randn(d, device=dev, generator=gen)
bits, nxt, _ = trunk._fused_spmm(G, z, bias, None, 1.0, 0.0, 0.0, 0)      # p = 0: the words are the ReLU mask
del nxt, z
g = torch.randn(n, d, device=dev, generator=gen)
c_act = 0.9


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sum(ev[i].elapsed_time(ev[i + 1]) for i in range(iters)) / iters


def old():
    gr, _ = trunk._layer_bwd(g, bits, G.norm_in, None, False, 0.0, 0, 0, c_act, 0.1, False)
    return G.spmm(gr, transpose=True)


def new():
    return G.spmm_masked(g, bits, G.norm_in, c_act)


a, b = old(), new()
print('max |old - new| =', float((a - b).abs().max()), ' max |old| =', float(a.abs().max()))
del a, b
t_bwd = timed(lambda: trunk._layer_bwd(g, bits, G.norm_in, None, False, 0.0, 0, 0, c_act, 0.1, False))
t_bwd_cs = timed(lambda: trunk._layer_bwd(g, bits, G.norm_in, None, False, 0.0, 0, 0, c_act, 0.1, True))
gr, _ = trunk._layer_bwd(g, bits, G.norm_in, None, False, 0.0, 0, 0, c_act, 0.1, False)
t_spmm = timed(lambda: G.spmm(gr, transpose=True))
del gr
t_new = timed(new)
print(f'MASK_U={os.environ.get("CB_SPMM_MASK_U", "8")}: layer_bwd {t_bwd:.2f} ms (+colsum {t_bwd_cs:.2f}), plain reverse aggregation {t_spmm:.2f} ms, '
      f'sum {t_bwd + t_spmm:.2f} ms; masked reverse aggregation {t_new:.2f} ms')
