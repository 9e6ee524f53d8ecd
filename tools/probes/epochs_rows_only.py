"""Ten epochs of the reference's epoch loop (main.py's trainer.main) on S-arxiv with and without the rows-only forward: the records (log loss, train / test accuracy)."""
import contextlib, io, os, sys, tempfile
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
import numpy as np, torch
from gnn_tail_generalization_amd.base_options import BaseOptions
from gnn_tail_generalization_amd.trainer_node_classification import trainer
for flag in ('1', '0'):
    os.environ['CB_ROWS_ONLY_FWD'] = flag
    os.chdir(tempfile.mkdtemp())
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-arxiv', '--epochs=10', '--manual_assign_GPU=0', '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0'])
        args.random_seed = 0
        torch.manual_seed(0); np.random.seed(0)
        t = trainer(args, 0)
        torch.manual_seed(0)
        rec = t.main()
    print('rows-only' if flag == '1' else 'all rows ', np.array2string(np.asarray(rec), precision=5, max_line_width=200))
