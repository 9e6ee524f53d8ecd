import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_rowsparse import _step_grads
for conn, layers, se in (('Initial', 3, '000'), ('Residual', 3, '000'), ('Initial', 4, '100'), ('Residual', 4, '000')):
    extra = () if conn == 'Initial' else ('--force_set_to_best_config=0', '--type_trick=Residual')
    os.environ['CB_ROWS_ONLY_BELOW'] = '2'
    l2, g2, _ = _step_grads('1', layers=layers, se=se, extra=extra, rows_only=True)
    os.environ['CB_ROWS_ONLY_BELOW'] = '1'
    l1, g1, _ = _step_grads('1', layers=layers, se=se, extra=extra, rows_only=True)
    ld, gd, _ = _step_grads('0', layers=layers, se=se, extra=extra)
    print(conn, layers, se, 'loss', l2, l1, ld)
    for k in gd:
        fr = float(gd[k].norm())
        print(f'  {k:40s} aggfirst vs dense {float((g2[k]-gd[k]).norm())/fr:.2e}   zfirst vs dense {float((g1[k]-gd[k]).norm())/fr:.2e}')
