import sys, os, contextlib, io, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_tail_generalization_amd.base_options import BaseOptions
from gnn_tail_generalization_amd.trainer_node_classification import trainer
L, se, pre = sys.argv[1], sys.argv[2], int(sys.argv[3])
with contextlib.redirect_stdout(io.StringIO()):
    args = BaseOptions().get_arguments(['--dataset=S-pubmed', '--use_special_split=0', '--want_headtail=0', f'--whetherHasSE={se}',
                                        f'--num_layers={L}', '--manual_assign_GPU=0', '--do_deg_analyze=0'])
    t = trainer(args, 0); t.setup_teacherGNN()
for _ in range(pre):
    t.train_step()
torch.cuda.synchronize()
print('eager ok', flush=True)
t.enable_hip_graph(warmup=1)
print('capture ok', flush=True)
for _ in range(3):
    l = t.train_step()
torch.cuda.synchronize()
print('replay ok', float(l), flush=True)
