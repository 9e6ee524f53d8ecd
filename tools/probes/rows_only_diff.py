"""Rows-only forward against the all-rows step (dense backward): per-parameter max and Frobenius differences of one step's gradients."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch
from test_gpu_rowsparse import _step_grads
for conn, se, layers in (('Initial', '000', 3), ('Residual', '000', 3), ('Initial', '000', 2), ('Initial', '100', 4)):
    extra = () if conn == 'Initial' else ('--force_set_to_best_config=0', '--type_trick=Residual')
    ls, gs, _ = _step_grads('1', se=se, layers=layers, extra=extra, rows_only=True)
    lb, gb, _ = _step_grads('1', se=se, layers=layers, extra=extra, rows_only=False)
    ld, gd, _ = _step_grads('0', se=se, layers=layers, extra=extra)
    print(conn, se, layers, 'loss', ls, lb, ld)
    for k in gd:
        sc, fr = float(gd[k].abs().max()), float(gd[k].norm())
        print(f'  {k:45s} rows-only vs dense: max {float((gs[k]-gd[k]).abs().max())/sc:.2e} fro {float((gs[k]-gd[k]).norm())/fr:.2e}   row-sparse bwd vs dense: max {float((gb[k]-gd[k]).abs().max())/sc:.2e} fro {float((gb[k]-gd[k]).norm())/fr:.2e}')
