"""Shared helpers for the parity tests: build the product's args/model from a golden cfg."""
import contextlib
import io

import torch


def product_args(cfg, extra=()):
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.utils import set_arch_configs
    se = ''.join(str(int(v)) for v in cfg['whetherHasSE'])
    argv = [f'--dataset={cfg["dataset"]}', '--manual_assign_GPU=0', '--force_set_to_best_config=0',
            f'--type_trick={cfg["type_trick"]}', f'--num_layers={cfg["num_layers"]}', f'--whetherHasSE={se}',
            f'--layer_agg={cfg["layer_agg"]}', f'--dim_learnable_input={cfg["dim_learnable_input"]}',
            f'--change_to_featureless={cfg["change_to_featureless"]}', f'--node_norm_type={cfg["node_norm_type"]}',
            f'--se_reg={cfg["se_reg"]}'] + list(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(argv)
    args.N_nodes, args.num_feats = cfg['N_nodes'], cfg['num_feats']
    args.dim_hidden, args.num_classes = cfg['dim_hidden'], cfg['num_classes']
    args.dropout = cfg.get('dropout', 0.0)
    args.res_alpha = cfg['res_alpha']
    args.num_groups, args.skip_weight = cfg['num_groups'], cfg['skip_weight']
    set_arch_configs(args)
    return args


def product_model(cfg, sd, device, extra=()):
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    args = product_args(cfg, extra)
    args.device = torch.device(device)
    model = TeacherGNN(args)
    model.load_state_dict(sd, strict=True)
    return args, model.to(device)


def oracle_cfg(cfg):
    import coldbrew_oracle as orc
    keys = ['type_trick', 'num_layers', 'num_feats', 'dim_hidden', 'num_classes', 'res_alpha', 'layer_agg',
            'whetherHasSE', 'node_norm_type', 'num_groups', 'skip_weight', 'se_reg', 'change_to_featureless',
            'dim_learnable_input']
    return orc.make_cfg(**{k: cfg[k] for k in keys})
