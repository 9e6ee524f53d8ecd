"""CSR ingest (edge_index -> both orientations, degree norms, hub plans, hot-source flags) at the headline size.  usage: python tools/bench_csr.py [name]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import _lib  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import CSRGraph  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'S-pl10M'
data = synthetic_data(name, seed=0, device='cuda:0')
ei, n = data.edge_index, int(data.x.shape[0])
del data
lib = _lib.load()
E = int(ei.shape[1])
rowptr = torch.empty(n + 1, dtype=torch.int32, device='cuda:0')
col = torch.empty(E, dtype=torch.int32, device='cuda:0')
rowptr_t, col_t = torch.empty_like(rowptr), torch.empty_like(col)
flags = torch.empty(4, dtype=torch.int32, device='cuda:0')
wsb = lib.cb_csr_workspace_bytes(E, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda:0')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
for i in range(5):
    ev[i].record()
    _lib.check(lib.cb_csr_from_coo_i64(_lib.ptr(ei[0]), _lib.ptr(ei[1]), E, n, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(rowptr_t), _lib.ptr(col_t),
                                       _lib.ptr(flags), _lib.ptr(ws), wsb, _lib.stream_ptr()), 'csr')
ev[5].record()
torch.cuda.synchronize()
print(f'{name}: N={n} E={E}: cb_csr_from_coo_i64 (both orientations) {min(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 5)):.2f} ms, workspace {wsb / 1e9:.2f} GB')
del ws
t0 = time.perf_counter()
G = CSRGraph(ei, n)
torch.cuda.synchronize()
print(f'CSRGraph(edge_index) incl. norms, hub plans, hot-source flags: {(time.perf_counter() - t0) * 1e3:.1f} ms wall')
