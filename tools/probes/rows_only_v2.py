import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_rowsparse import _step_grads
for layers in (3, 2, 4):
    l2, g2, _ = _step_grads('1', layers=layers, rows_only=2)
    l1, g1, _ = _step_grads('1', layers=layers, rows_only=1)
    ld, gd, _ = _step_grads('0', layers=layers)
    print('layers', layers, 'loss', l2, l1, ld)
    for k in gd:
        fr = float(gd[k].norm())
        print(f'  {k:40s} v2 vs dense {float((g2[k]-gd[k]).norm())/fr:.2e}   v1 vs dense {float((g1[k]-gd[k]).norm())/fr:.2e}')
