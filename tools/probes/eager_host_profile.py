"""cProfile of the host side of eager training steps on a launch-bound graph (S-cora / S-pubmed): where the ~20 us per launch go."""
import contextlib, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnn_tail_generalization_amd.base_options import BaseOptions
from gnn_tail_generalization_amd.trainer_node_classification import trainer

ds = sys.argv[1] if len(sys.argv) > 1 else 'S-cora'
with contextlib.redirect_stdout(io.StringIO()):
    args = BaseOptions().get_arguments([f'--dataset={ds}', '--manual_assign_GPU=0', '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0'] + sys.argv[2:])
    t = trainer(args, 0)
    t.setup_teacherGNN()
for _ in range(20):
    t.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    t.train_step()
torch.cuda.synchronize()
print(f'{ds}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step eager')
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    t.train_step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr).sort_stats('tottime')
st.print_stats(28)
