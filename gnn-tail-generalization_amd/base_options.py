"""Option system of the TeacherGNN path — same flag names, defaults and mutation passes as the
reference's base_options.py (get_arguments :9-171, reset_dataset_dependent_parameters
:186-304, force_set_to_best_config :404-438), so README command lines run unchanged.

Documented deviations:
  * `--exp_mode` defaults to 'coldbrew' (the reference hard-codes 'I2_GTL', :10, which routes
    to the out-of-scope link-prediction trainer).  The default `--lr` stays 0.001: the reference
    derives it from that hard-coded mode (:11-15,19), so 0.001 is what its coldbrew trainer
    actually runs with when `--exp_mode=coldbrew` is passed;
  * the GPU picker never shells out to nvidia-smi (reference bestGPU :332-350): without
    `--manual_assign_GPU` the device is LOCAL_RANK (one process per GPU) or 0;
  * `--dataset` additionally accepts 'ogbn-products' and the synthetic stand-ins of
    SURVEY.md §8(d) ('S-cora', 'S-pubmed', 'S-arxiv', 'S-products', 'S-pl10M', 'S-tiny');
  * extra flag `--agg_dtype {f32,bf16}` (build extension, BASELINE config 2): bf16 storage of the rows the
    aggregation gathers, fp32 accumulation; the reference is fp32-only.
"""
import argparse
import os

import numpy as np

DATASETS = ['Cora', 'Citeseer', 'Pubmed', 'ogbn-arxiv', 'chameleon', 'ACTOR', 'squirrel', 'WISCONSIN', 'CORNELL', 'TEXAS']
EXTRA_DATASETS = ['ogbn-products', 'S-cora', 'S-pubmed', 'S-arxiv', 'S-products', 'S-pl10M', 'S-pl1M', 'S-tiny']

# dataset -> (num_feats, num_classes, N_nodes, dropout, weight_decay, patience, dim_hidden, activation, res_alpha)
_PRESETS = {
    'Cora': (1433, 7, 2708, 0.6, 5e-4, 100, 64, 'relu', None),
    'Pubmed': (500, 3, 19717, 0.5, 5e-4, 100, 256, 'relu', None),
    'Citeseer': (3703, 6, 3327, 0.6, 5e-4, 100, 256, 'relu', 0.2),
    'ogbn-arxiv': (128, 40, 169343, 0.1, 0., 200, 256, None, None),
    'chameleon': (128, 6, 2277, 0.5, 5e-4, None, 256, 'relu', None),
    'squirrel': (128, 5, 5201, 0.5, 5e-4, None, 256, 'relu', None),
    'TEXAS': (1703, 5, 183, 0.6, 5e-4, 100, 256, 'relu', 0.9),
    'WISCONSIN': (1703, 5, 251, 0.6, 5e-4, 100, 256, 'relu', 0.9),
    'CORNELL': (1703, 5, 183, 0., 5e-4, 100, 256, 'relu', 0.9),
    'ACTOR': (932, 5, 7600, 0., 5e-4, 100, 256, 'relu', 0.9),
    # build extensions (not in the reference): shapes from SURVEY.md §8
    'ogbn-products': (100, 47, 2449029, 0.1, 0., 200, 256, None, None),
    'S-cora': (1433, 7, 2708, 0.6, 5e-4, 100, 64, 'relu', None),
    'S-pubmed': (500, 3, 19717, 0.5, 5e-4, 100, 256, 'relu', None),
    'S-arxiv': (128, 40, 169343, 0.1, 0., 200, 256, None, None),
    'S-products': (100, 47, 2449029, 0.1, 0., 200, 256, None, None),
    'S-pl10M': (128, 40, 10000000, 0.1, 0., 200, 256, None, None),
    'S-pl1M': (128, 40, 1000000, 0.1, 0., 200, 256, None, None),
    'S-tiny': (16, 4, 256, 0.1, 5e-4, 100, 32, 'relu', None),
}

# which reference dataset's "best config" a stand-in follows
_BEST_ALIAS = {'S-cora': 'Cora', 'S-pubmed': 'Pubmed', 'S-arxiv': 'ogbn-arxiv', 'ogbn-products': 'ogbn-arxiv',
               'S-products': 'ogbn-arxiv', 'S-pl10M': 'ogbn-arxiv', 'S-pl1M': 'ogbn-arxiv', 'S-tiny': 'Pubmed'}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def build_parser():
    _exp_mode = 'coldbrew'
    _lr = 0.001   # what the reference computes (its _exp_mode is hard-coded to I2_GTL, base_options.py:10-15)
    p = argparse.ArgumentParser(description='Tail and cold start generalization (MI355X TeacherGNN path)')
    a = p.add_argument
    a('--exp_mode', type=str, default=_exp_mode)
    a('--lr', type=float, default=_lr, help='learning rate')
    a('--dropout', type=float, default=0.2)
    a('--batch_size', type=int, default=64 * 1024)
    a('--epochs', type=int, default=1500)
    # node classification (Cold Brew)
    a('--samp_size_p', type=int, default=200)
    a('--samp_size_n_train', type=int, default=200)
    a('--samp_size_n_test_times_p', type=int, default=20)
    a('--dim_learnable_input', type=int, default=0)
    a('--unify_mlps', type=int, default=0)
    a('--force_set_to_best_config', type=int, default=1)
    a('--want_headtail', type=int, default=1)
    a('--num_layers', type=int, default=2)
    a('--studentMLP__skip_conn_T_and_res_blks', type=str, default='')
    a('--StudentMLP__dim_model', type=int, default=-1)
    a('--studentMLP__opt_lr', type=str, default='')
    a('--LP__which_corr_and_DAD', type=str, default='')
    a('--LP__num_propagations', type=int, default=-1)
    a('--LP__alpha', type=float, default=-1)
    a('--SEMLP_topK_2_replace', type=int, default=2)
    a('--SEMLP__include_part1out', type=int, default=1)
    a('--dropout_MLP', type=float, default=0.2)
    a('--SEMLP_part1_arch', type=str, default='2layer', choices=['residual', '2layer', '3layer', '4layer'])
    a('--has_proj2class', type=int, default=0)
    a('--whetherHasSE', type=str, default='000', choices=['100', '001', '111', '000'])
    a('--se_reg', type=float, default=10)
    a('--graphMLP_reg', type=float, default=0., choices=[0., 1., 10., 100.])
    a('--graphMLP_tau', type=float, default=2.0, choices=[0.5, 1.0, 2.0])
    a('--graphMLP_r', type=int, default=3, choices=[2, 3, 4])
    a('--change_to_featureless', type=int, default=0)
    a('--do_deg_analyze', type=int, default=1)
    a('--train_which', type=str, default='TeacherGNN',
      choices=['TeacherGNN', 'SEMLP', 'LP', 'StudentBaseMLP', 'GraphMLP', 'proj2class'])
    a('--task', type=str, default='nodeC')
    a('--dataset', type=str, default='', choices=DATASETS + EXTRA_DATASETS)
    a('--use_special_split', type=int, default=1)
    a('--optfun', type=str, default='torch.optim.Adam', choices=['torch.optim.Adam', 'torch.optim.SGD'])
    a('--manual_assign_GPU', type=int, default=-9999)
    a('--random_seed', type=int, default=100)
    a('--N_exp', type=int, default=1)
    a('--resume', action='store_true', default=False)
    a('--cuda', type=bool, default=True, required=False)    # type=bool as in the reference: any non-empty string is True
    a('--cuda_num', type=int, default=0)
    a('--records_desc', type=str, default='res_connection')
    a('--records_path', type=str, default='.')
    a('--compare_model', type=int, default=0)
    a('--type_model', type=str, default='GCN',
      choices=['GCN', 'GAT', 'SGC', 'GCNII', 'DAGNN', 'GPRGNN', 'APPNP', 'JKNet', 'DeeperGCN'])
    a('--type_trick', type=str, default='Initial+BatchNorm')
    a('--layer_agg', type=str, default='concat', choices=['concat', 'maxpool', 'attention', 'mean'])
    a('--res_alpha', type=float, default=0.1)
    a('--patience', type=int, default=100)
    a('--multi_label', type=bool, default=False)
    a('--weight_decay', type=float, default=5e-4)
    a('--dim_hidden', type=int, default=64)
    a('--transductive', type=bool, default=True)
    a('--float_or_double', type=str, default='float', required=False)
    a('--type_norm', type=str, default='None')
    a('--adj_dropout', type=float, default=0.5)
    a('--edge_dropout', type=float, default=0.2)
    a('--node_norm_type', type=str, default='n', choices=['n', 'v', 'm', 'srv', 'pr'])
    a('--skip_weight', type=float, default=None)
    a('--num_groups', type=int, default=None)
    a('--prog', type=str, default='')
    a('--rexName', type=str, default='res.npy')
    a('--graph_dropout', type=float, default=0.2)
    a('--layerwise_dropout', action='store_true', default=False)
    a('--ckpt_every', type=int, default=0)   # build extension: write the resumable checkpoint every k epochs (0 = only at the end)
    a('--hip_graph', type=int, default=0)    # build extension: 1 = the epoch loop replays the training step (and the eval forward) as hipGraphs
    a('--agg_dtype', type=str, default='f32', choices=['f32', 'bf16'])   # build extension: storage type of the aggregated rows
    a('--rows_only_forward', type=int, default=1)   # build extension: 1 = the trainer promises the model that its training forward's output is read in the train rows only (trunk.py "Rows-only forward"; the other rows come back as NaN); 0 = every row of every training forward
    # link-prediction (I2-GTL) flags: accepted for CLI compatibility, unused by this path
    a('--public_data_convert_overlapped_subgraph', type=bool, default=True)
    a('--transfer_setting', type=str, default='i2t', choices=['t2t', 'u2t', 'i2t', 'u', 'i', ''])
    a('--linkpred_baseline', type=str, default='', choices=['', 'EGI', 'DGI'])
    a('--edge_lp_mode', type=str, default='logit', choices=['emb', 'logit', 'xmc', ''])
    a('--ELP_alpha', type=str, default=0.995)
    a('--num_propagations', type=str, default=5)
    a('--LP_device', type=str, default='cuda:4', choices=['cpu', 'cuda:0', 'cuda:4'])
    a('--exp_on_cold_edge', type=bool, default=False)
    a('--encoder', type=str, default='SAGE', choices=['SAGE', 'MLP', 'CN', 'AA', 'PPR'])
    a('--predictor', type=str, default='DOT', choices=['MLP', 'DOT'])
    a('--optimizer', type=str, default='Adam')
    a('--loss_func', type=str, default='ce_loss', choices=['AUC', 'ce_loss', 'log_rank_loss', 'info_nce_loss'])
    a('--neg_sampler', type=str, default='global')
    a('--data_name', type=str, default='ogbl-citation2', choices=['ogbl-citation2', 'ogbl-collab'])
    a('--data_path', type=str, default='dataset')
    a('--eval_metric', type=str, default='recall_my@1.25',
      choices=['hits', 'mrr', 'recall_my@0.8', 'recall_my@1', 'recall_my@1.25', 'recall_my@0'])
    a('--res_dir', type=str, default='')
    a('--pretrain_emb', type=str, default='')
    a('--gnn_num_layers', type=int, default=2)
    a('--mlp_num_layers', type=int, default=2)
    a('--emb_hidden_channels', type=int, default=256)
    a('--gnn_hidden_channels', type=int, default=256)
    a('--mlp_hidden_channels', type=int, default=256)
    a('--grad_clip_norm', type=float, default=2.0)
    a('--num_neg', type=int, default=3)
    a('--log_steps', type=int, default=1)
    a('--eval_steps', type=int, default=5)
    a('--runs', type=int, default=10)
    a('--year', type=int, default=2010)
    a('--linkpred_device', type=int, default=1)
    a('--use_node_feats', type=str2bool, default=False)
    a('--use_coalesce', type=str2bool, default=False)
    a('--train_node_emb', type=str2bool, default=True)
    a('--train_on_subgraph', type=str2bool, default=True)
    a('--use_valedges_as_input', type=str2bool, default=True)
    a('--eval_last_best', type=str2bool, default=True)
    return p


class BaseOptions:
    def get_arguments(self, argv=None):
        args = build_parser().parse_args(argv)
        if args.unify_mlps:
            unify_mlps(args)
        args = self.reset_dataset_dependent_parameters(args)
        args = self.ini_records_saver(args)
        if args.manual_assign_GPU != -9999:
            args.cuda_num = args.manual_assign_GPU
        else:
            args.cuda_num = int(os.environ.get('LOCAL_RANK', args.cuda_num))
        set_labprop_configs(args)
        if args.force_set_to_best_config:
            force_set_to_best_config(args)
        print(f'\nConfigs: \n\tdataset = < {args.dataset} >\n\ttrain_which = < {args.train_which} >\n\t'
              f'type_trick = < {args.type_trick} >\n\tnum_layers = < {args.num_layers} >\n\t'
              f'dim_hidden = < {args.dim_hidden} >\n\tGPU actually use = < {args.cuda_num} >\n\n')
        if args.exp_mode == 'coldbrew':
            args.has_loss_component_nodewise = True
            args.has_loss_component_edgewise = False
        elif args.exp_mode == 'I2_GTL':
            args.has_loss_component_nodewise = False
            args.has_loss_component_edgewise = True
            if args.linkpred_baseline in ['EGI', 'DGI']:
                args.encoder = 'MLP'
                args.use_node_feats = True
        return args

    def ini_records_saver(self, args):
        records_file = os.path.join(args.records_path, args.records_desc)
        if os.path.exists(records_file):
            backup = os.path.join(args.records_path, args.records_desc + ' - backup')
            print(f'\n\n !!! Warning !!! assigned records_file < {records_file} > already exists, '
                  f'now re-name the previous one to < {backup} >\n\n')
            os.rename(records_file, backup)
        args.records_file = records_file
        return args

    def reset_dataset_dependent_parameters(self, args):
        pre = _PRESETS.get(args.dataset)
        if pre is None:
            return args
        nf, nc, n, dp, wd, pat, dh, act, ra = pre
        args.num_feats, args.num_classes, args.N_nodes = nf, nc, n
        args.dropout, args.weight_decay, args.dim_hidden = dp, wd, dh
        if pat is not None:
            args.patience = pat
        if act is not None:
            args.activation = act
        if ra is not None:
            args.res_alpha = ra
        return args


# Table of base_options.py:412-416: per dataset (depth index, residual kind, norm kind).
_D2I = {'Cora': 0, 'Citeseer': 1, 'Pubmed': 2, 'ogbn-arxiv': 3, 'chameleon': 4, 'ACTOR': 5, 'squirrel': 6,
        'WISCONSIN': 7, 'CORNELL': 8, 'TEXAS': 9}
_BEST_PERF = [86.9639468690702, 72.44, 75.96000000000001, 71.5367364154476, 68.50877192982458, 31.947368421052637,
              59.78866474543709, 65.09803921568627, 61.08108108108108, 81.62162162162163]
_RES = ('NoRes', 'Initial', 'Dense', 'Residual')
_NORM = ('NoNorm', 'GroupNorm', 'BatchNorm', 'PairNorm', 'NodeNorm')
_BEST_TEACHER = np.array([[0, 0, 4], [0, 0, 1], [4, 1, 2], [2, 1, 2], [1, 1, 3], [0, 0, 2], [0, 1, 4], [1, 3, 0], [2, 3, 3], [2, 3, 1]])
_BEST_STUDENT = np.array([[0, 1, 0], [0, 0, 0], [1, 0, 3], [1, 1, 0], [2, 0, 0], [0, 1, 2], [2, 1, 2], [0, 1, 0], [0, 1, 3], [0, 0, 2]])


def force_set_to_best_config(args):
    """type_trick := <residual kind><norm kind> from the per-dataset table (base_options.py:404-438).
    The result is a *concatenated* name ('NoResNodeNorm', 'InitialBatchNorm', ...)."""
    print('-' * 30, '\n\n\n   Now reseting configs !!! \n\n\n', '-' * 30)
    ds = _BEST_ALIAS.get(args.dataset, args.dataset)
    if ds not in _D2I:
        return
    if args.train_which in ['SEMLP', 'StudentBaseMLP', 'TeacherGNN']:
        args.best_config_performance = list(_BEST_PERF)
        _, ires, inorm = _BEST_TEACHER[_D2I[ds]]
        args.type_trick = _RES[ires] + _NORM[inorm]
    if args.train_which in ['SEMLP', 'StudentBaseMLP']:
        arr1 = ('2&1', '2&4', '2&16', '2&32', '4&2', '4&8')
        arr2 = (128, 256)
        i1, i2, _ = _BEST_STUDENT[_D2I[ds]]
        args.studentMLP__skip_conn_T_and_res_blks = arr1[i1]
        args.StudentMLP__dim_model = arr2[i2]
        args.studentMLP__opt_lr = 'torch.optim.Adam&0.005'


def unify_mlps(args):                              # base_options.py:450-471
    args.studentMLP__skip_conn_T_and_res_blks = '2&2'
    args.StudentMLP__dim_model = 128
    args.studentMLP__opt_lr = 'torch.optim.Adam&0.005'
    args.SEMLP__include_part1out = 1
    if args.train_which == 'SEMLP':
        args.SEMLP_topK_2_replace = 3
    elif args.train_which == 'GraphMLP':
        args.graphMLP_reg, args.graphMLP_tau, args.graphMLP_r = 10, 1, 3
    elif args.train_which in 'SEMLP_MLP':
        args.SEMLP_topK_2_replace = -99
        args.train_which = 'SEMLP'
    elif args.train_which == 'GraphMLP_MLP':
        args.graphMLP_reg = 0
        args.train_which = 'GraphMLP'


def set_labprop_configs(args):
    """Label-propagation options (base_options.py:352-402) belong to the out-of-scope LP trainer;
    only the attribute the TeacherGNN path could meet is defined."""
    args.lp_has_prep = 1
    args.lpStep = None
