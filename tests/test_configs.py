"""BASELINE.json configs as test cases.  Config 1 (Cora-shaped, whetherHasSE=000, 2 layers) runs on the
oracle's CPU path here (plumbing, no GPU) and — on the GPU box — through the product CLI objects against the
oracle; configs 2/3 (Pubmed-shaped 111, arxiv-shaped 3-layer hidden 256) are checked against the oracle on a
node-subsampled instance of the same synthetic family."""
import contextlib
import io

import numpy as np
import pytest
import torch

import coldbrew_oracle as orc


def _args(argv):
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.utils import set_arch_configs
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(argv + ['--manual_assign_GPU=0'])
    return args, set_arch_configs


def _oracle_cfg(args):
    return orc.make_cfg(type_trick=args.type_trick, num_layers=args.num_layers, num_feats=args.num_feats, dim_hidden=args.dim_hidden,
                        num_classes=args.num_classes, res_alpha=args.res_alpha, whetherHasSE=tuple(args.TeacherGNN.whetherHasSE),
                        se_reg=args.se_reg, node_norm_type=args.node_norm_type)


def test_config1_cora_shaped_on_oracle_cpu_path():
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    args, set_arch = _args(['--dataset=S-cora', '--whetherHasSE=000', '--num_layers=2', '--use_special_split=1'])
    assert args.type_trick == 'NoResNodeNorm' and (args.num_feats, args.dim_hidden, args.num_classes) == (1433, 64, 7)
    data = synthetic_data('S-cora', seed=0)
    assert data.x.shape == (2708, 1433) and data.edge_index.shape[1] == 2 * 5278 + 2708
    set_arch(args)
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in TeacherGNN(args).state_dict().items()}    # parameters only; no compute on CPU
    csr = orc.build_csr(data.edge_index, 2708)
    losses = orc.train_steps(_oracle_cfg(args), sd, data.x, csr, data.y, data.train_mask, 4, lr=0.01, weight_decay=5e-4)
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize('argv,n', [(['--dataset=S-cora', '--whetherHasSE=000', '--num_layers=2'], None),
                                    (['--dataset=S-pubmed', '--whetherHasSE=111', '--num_layers=2', '--se_reg=0.5'], 3000),
                                    (['--dataset=S-arxiv', '--num_layers=3', '--use_special_split=0'], 6000)])
def test_configs_product_vs_oracle(argv, n):
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    dev = 'cuda:0'
    args, set_arch = _args(argv)
    data = synthetic_data(args.dataset, seed=0, device=dev, n_override=n)
    args.N_nodes, args.dropout, args.device = data.x.shape[0], 0.0, torch.device(dev)
    set_arch(args)
    torch.manual_seed(0)
    model = TeacherGNN(args).to(dev)
    model.train()
    out = model(data.x, data.edge_index)
    loss = ops.nll_logsoftmax(out, data.y, data.train_mask)
    if model.se_reg_all is not None:
        loss = loss + args.se_reg * model.se_reg_all
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    csr = orc.build_csr(data.edge_index.cpu(), data.x.shape[0])
    cfg = _oracle_cfg(args)
    o, reg = orc.teacher_forward(cfg, sd, data.x.cpu(), csr, training=True)
    l = orc.training_loss(cfg, o, reg, data.y.cpu(), data.train_mask.cpu())
    l.backward()
    torch.testing.assert_close(out.detach().cpu(), o.detach(), atol=1e-4, rtol=1e-4)        # north_star: logits within 1e-4
    torch.testing.assert_close(loss.detach().cpu(), l.detach(), atol=1e-5, rtol=1e-5)
    for k, p in model.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(p.grad.cpu(), sd[k].grad, atol=2e-5, rtol=5e-4, msg=lambda m, k=k: f'{k}: {m}')
