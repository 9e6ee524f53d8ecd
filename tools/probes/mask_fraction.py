"""Step time at the headline size for several train-mask fractions: rows-only forward (default) / all-rows forward + row-sparse backward / dense."""
import contextlib, io, os, sys, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
import torch
import bench
from gnn_tail_generalization_amd import trainer_node_classification as tnc
args = bench.make_args(sys.argv[1] if len(sys.argv) > 1 else 'S-pl10M', ['--manual_assign_GPU=0'])
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    t = tnc.trainer(args, 0)
    t.setup_teacherGNN()
g = torch.Generator(device='cpu').manual_seed(1)
n = t.data.train_mask.shape[0]
for frac in (0.02, 0.1, 0.3, 0.5, 0.7):
    m = (torch.rand(n, generator=g) < frac).to(t.data.train_mask.device)
    t.data.train_mask, t.data.test_mask, t._n_train = m, ~m, None
    row = []
    for env in ({'CB_ROWS_ONLY_FWD': '1', 'CB_LOSS_ROWS': '1'}, {'CB_ROWS_ONLY_FWD': '0', 'CB_LOSS_ROWS': '1'}, {'CB_ROWS_ONLY_FWD': '0', 'CB_LOSS_ROWS': '0'}):
        os.environ.update(env)
        for _ in range(2):
            t.train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t.train_step()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f'mask {frac:4.2f}: rows-only {row[0]:7.2f} ms   all rows + row-sparse backward {row[1]:7.2f} ms   dense {row[2]:7.2f} ms', flush=True)
