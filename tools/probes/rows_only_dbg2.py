import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from gnn_tail_generalization_amd import trunk, ops, _lib
real = trunk._layer_on_rows
def spy(graph, space, fwd, col_scale, cur, w, b, mix, mix_index, alpha, p, seed, row0, residual, want_act=False):
    bits, x, act, h = real(graph, space, fwd, col_scale, cur, w, b, mix, mix_index, alpha, p, seed, row0, residual, want_act)
    if cur.shape[0] == graph.N and space.n > graph.N // 5:      # the layer below the last one (input on all rows)
        z = trunk.gemm.mm_nn(cur, w, rowscale=graph.norm_out)
        plan = graph._support_plan
        rb = graph.rows_only_fwd(plan)
        bits_z, x_z, _ = trunk._fused_launch(_lib.load(), graph, rb[0], z, None, b, mix, 1 - alpha, alpha, p, seed, False, want_bits=True, relu_only=residual,
                                             row_ids=rb[2], row_scale=rb[3])
        idx = space.idx
        nd = int((bits[idx] != bits_z[idx]).sum())
        print('layer below: x diff', float((x - x_z).abs().max()), 'scale', float(x_z.abs().max()), 'mask words differing', nd, 'of', idx.numel() * 4,
              'elements with different zero pattern', int(((x == 0) != (x_z == 0)).sum()))
    return bits, x, act, h
trunk._layer_on_rows = spy
from test_gpu_rowsparse import _step_grads
_step_grads('1', layers=3, rows_only=True)
