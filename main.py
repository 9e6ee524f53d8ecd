"""CLI entry of the TeacherGNN path — same contract as the reference's main.py:14-38:
    python main.py --dataset=Cora --train_which=TeacherGNN --whetherHasSE=000 --want_headtail=1 \
                   --num_layers=2 --use_special_split=1
parses the options, loops over seeds, builds the trainer and returns per-seed result arrays
[seeds, record_type, epochs].  `--exp_mode` defaults to 'coldbrew' here (documented deviation)."""
import gc
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gnn_tail_generalization_amd.base_options import BaseOptions  # noqa: E402


def main(argv=None):
    args = BaseOptions().get_arguments(argv)
    if args.exp_mode == 'coldbrew':
        from gnn_tail_generalization_amd.trainer_node_classification import trainer
    else:
        raise NotImplementedError(f"--exp_mode={args.exp_mode}: only 'coldbrew' (node classification, TeacherGNN) is built; "
                                  'the I2_GTL link-prediction trainer is out of scope (SURVEY.md §2 #13-14)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        # launched by torchrun (one process per GPU): the node-sharded trainer over RCCL (new; the reference is single-device)
        import torch.distributed as dist
        from gnn_tail_generalization_amd.dist import ShardedTrainer
        local_rank = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)   # (modulo: several gloo ranks on one GPU in the tests)
        args.cuda_num = local_rank
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group(os.environ.get('COLDBREW_DIST_BACKEND', 'nccl'))
        trainer = ShardedTrainer
    if args.prog:
        tensorRex(None, args.prog, args.rexName)
    full_recs_3D = []
    for seed in range(args.N_exp):
        print(f'seed (which_run) = <{seed}>')
        args.random_seed = seed
        set_seed(args)
        trnr = trainer(args, seed)
        results_arr2D = trnr.main()
        full_recs_3D.append(results_arr2D)
        del trnr
        torch.cuda.empty_cache()
        gc.collect()
    if args.prog:
        tensorRex(full_recs_3D, args.prog, args.rexName)
    return full_recs_3D


def set_seed(args):
    if args.cuda and not torch.cuda.is_available():
        args.cuda = False
    if args.cuda:
        torch.cuda.manual_seed_all(args.random_seed)
    torch.manual_seed(args.random_seed)
    np.random.seed(args.random_seed)
    random.seed(args.random_seed)


def tensorRex(dataND, prog, rexName):
    """Batch-sweep bookkeeping (main.py:54-124): prog = 'i_j_k__//__idx__//__s0*s1*s2'.  First call
    (dataND=None) aborts if the experiment is already recorded; second call stores the results."""
    indices, idx, shape = prog.split('__//__')
    indices = tuple(np.array(indices.split('_'), dtype=int))
    idx = int(idx)
    shape = list(np.array(shape.split('*'), dtype=int))
    try:
        rec = np.load(rexName, allow_pickle=True).item()
    except FileNotFoundError:
        assert idx == 0, '\n\n\nFatal Error! previous experiment file deleted!\n\n\n'
        rec = None
    if dataND is None:
        if rec is not None and rec['flag'][indices] == 1.:
            raise UserWarning('\n\n\nThis exp has completed already\n\n\n')
        return
    dataND = np.asarray(dataND)
    if rec is None:
        rec = {'data': np.zeros(shape + list(dataND.shape), dtype=float), 'flag': np.zeros(shape, dtype=float)}
    slot = rec['data'][indices]
    if slot.shape != dataND.shape:   # tolerant fill into the upper-front corner
        assert slot.ndim == dataND.ndim
        sl = tuple(slice(0, min(a, b)) for a, b in zip(dataND.shape, slot.shape))
        slot[sl] = dataND[sl]
    else:
        rec['data'][indices] = dataND
    rec['flag'][indices] = 1.
    np.save(rexName, rec)


if __name__ == '__main__':
    main()
