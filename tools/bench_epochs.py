"""Wall time per epoch of the reference's epoch loop (train_teacherGNN: run_trainSet + run_testSet + records) on the small BASELINE
shapes, eager vs --hip_graph=1.  usage: python tools/bench_epochs.py [--datasets S-cora,S-pubmed] [--epochs 200]"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--datasets', default='S-cora,S-pubmed,S-arxiv')
    ap.add_argument('--epochs', type=int, default=200)
    a = ap.parse_args()
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    os.chdir(tempfile.mkdtemp())
    for ds in a.datasets.split(','):
        for wh in (0, 1):
            for hg in (0, 1):
                with contextlib.redirect_stdout(io.StringIO()):
                    args = BaseOptions().get_arguments([f'--dataset={ds}', f'--epochs={a.epochs}', '--manual_assign_GPU=0', f'--hip_graph={hg}',
                                                        f'--want_headtail={wh}', '--use_special_split=1'])
                    torch.manual_seed(0)
                    t = trainer(args, 0)
                    if args.do_deg_analyze:
                        from gnn_tail_generalization_amd.utils import save_graph_analyze
                        save_graph_analyze(args.N_nodes, t.data, args.use_special_split)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rec = t.train_teacherGNN()
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                ops.set_graph_seed(None)
                print(f'{ds:10s} want_headtail={wh} hip_graph={hg}: {dt / a.epochs * 1e3:8.3f} ms/epoch   final test acc {rec[0][-1]:.2f}', flush=True)
                del t
                torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
