import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from gnn_tail_generalization_amd import trunk, ops
rec = {}
real = trunk._last_layer_on_loss_rows
def spy(graph, plan, cur, w, b, mix, alpha, p, seed, row0, residual, w_out, b_out):
    bits, x_l, out, h = real(graph, plan, cur, w, b, mix, alpha, p, seed, row0, residual, w_out, b_out)
    # the dense form of the same layer
    z = trunk.gemm.mm_nn(cur, w, rowscale=graph.norm_out)
    bits_d, cur_d, _ = trunk._fused_spmm(graph, z, b, mix, 1 - alpha, alpha, p, seed, want_bits=True, relu_only=residual)
    idx = plan.space0.idx
    print('x_l diff', float((x_l - cur_d[idx]).abs().max()), 'scale', float(cur_d[idx].abs().max()))
    print('bits equal on S_0:', bool((bits[idx] == bits_d[idx]).all()), int((bits[idx] != bits_d[idx]).sum()))
    print('zero pattern equal:', int(((x_l == 0) != (cur_d[idx] == 0)).sum()))
    return bits, x_l, out, h
trunk._last_layer_on_loss_rows = spy
from test_gpu_rowsparse import _step_grads
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ls, gs, _ = _step_grads('1', layers=layers, rows_only=True)
trunk._last_layer_on_loss_rows = real
lb, gb, _ = _step_grads('1', layers=layers, rows_only=False)
for k in gb:
    print(f'{k:40s} {float((gs[k]-gb[k]).norm())/float(gb[k].norm()):.2e}')
