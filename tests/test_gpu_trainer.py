"""GPU: rows a15-a17 — the product's run_trainSet()/run_testSet() reproduce the trajectories the
unmodified reference trainer produced on the same inputs (loss 1e-5 rel, accuracies exact)."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from helpers import product_model

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('name', golden_cases('trainer_'))
def test_trainer_trajectory(name):
    from gnn_tail_generalization_amd import optim
    from gnn_tail_generalization_amd.data import Data
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    g = load_golden(name)
    args, model = product_model(g['cfg'], g['sd'], DEV, extra=[f'--want_headtail={g["want_headtail"]}',
                                                                  f'--use_special_split={g["use_special_split"]}'])
    args.lr, args.weight_decay = 0.01, 5e-4
    args.has_loss_component_nodewise, args.has_loss_component_edgewise = True, False
    data = Data(x=g['x'], y=g['y'], edge_index=g['edge_index'], train_mask=g['train_mask'], test_mask=~g['train_mask']).to(DEV)
    data.zero_deg_idx, data.small_deg_idx, data.large_deg_idx = (g[k].numpy() for k in ['zero_deg_idx', 'small_deg_idx', 'large_deg_idx'])
    t = trainer.__new__(trainer)
    t.args, t.data, t.bag = args, data, {}
    t.loss_fn = torch.nn.functional.nll_loss
    t.teacherGNN = model
    t.optimizer = optim.resolve(args.optfun)(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    rec, bags = [], []
    for ep in range(g['steps']):
        t.epoch = ep
        loss_train, _, _ = t.run_trainSet()
        acc_train, _, acc_test, _ = t.run_testSet()
        rec.append([loss_train, acc_train, acc_test])
        bags.append([float(v) for v in t.bag['head_tail_iso']])
    rec = np.array(rec)
    want = g['trajectory'].numpy()
    np.testing.assert_allclose(rec[:, 0], want[:, 0], rtol=1e-5)
    np.testing.assert_array_equal(rec[:, 1:], want[:, 1:])
    np.testing.assert_allclose(np.array(bags).reshape(want.shape[0], -1), g['head_tail_iso'].numpy().reshape(want.shape[0], -1), atol=1e-3)
    for k, v in g['sd_final'].items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(model.state_dict()[k].cpu(), v, atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')
    if g['want_headtail']:
        # the metric groups may arrive as numpy arrays (host analysis) or as device tensors (utils' device kernels, bench.py)
        model.eval()
        t._headtail_metrics()
        host = np.array(t.bag['head_tail_iso'], dtype=np.float64)
        for k in ['zero_deg_idx', 'small_deg_idx', 'large_deg_idx']:
            setattr(data, k, torch.as_tensor(getattr(data, k), device=DEV))
        t._headtail_metrics()
        np.testing.assert_array_equal(np.array(t.bag['head_tail_iso'], dtype=np.float64), host)


@pytest.mark.parametrize('name', [n for n in golden_cases('trainer_') if 'se000' not in n])
def test_se_regulariser_folded_into_adam_equals_autograd_path(name, monkeypatch):
    """ops.fold_se_reg (default: `se_reg * le / ||le||` added inside the fused Adam kernel) against CB_SE_REG_FOLD=0 (the term's
    gradient through autograd of the Frobenius norm, as the reference does): same losses, same final tables."""
    from gnn_tail_generalization_amd import optim
    from gnn_tail_generalization_amd.data import Data
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    g = load_golden(name)
    finals, losses = [], []
    cfg = dict(g['cfg'], dropout=0.0)         # deterministic steps: the two runs must see the same data gradients
    for fold in ('1', '0'):
        monkeypatch.setenv('CB_SE_REG_FOLD', fold)
        args, model = product_model(cfg, g['sd'], DEV, extra=['--want_headtail=0', f'--use_special_split={g["use_special_split"]}'])
        args.lr, args.weight_decay = 0.01, 5e-4
        data = Data(x=g['x'], y=g['y'], edge_index=g['edge_index'], train_mask=g['train_mask'], test_mask=~g['train_mask']).to(DEV)
        t = trainer.__new__(trainer)
        t.args, t.data, t.bag, t.device = args, data, {}, torch.device(DEV)
        t.teacherGNN = model
        t.optimizer = optim.resolve(args.optfun)(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        model.train()
        ls = []
        for _ in range(4):
            loss = t.training_loss()
            t.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            t.optimizer.step()
            ls.append(float(loss.detach()))
        les = {k: v.detach().clone() for k, v in model.state_dict().items() if k.endswith('.le')}
        assert les and len(t.optimizer._extra_decay) == (len(les) if fold == '1' else 0)
        finals.append(les)
        losses.append(ls)
    np.testing.assert_allclose(losses[0], losses[1], rtol=2e-6)
    for k in finals[0]:
        torch.testing.assert_close(finals[0][k], finals[1][k], atol=2e-6, rtol=1e-5)


def test_cli_end_to_end_tiny():
    """main.py CLI drives data -> head/tail split -> TeacherGNN training on the HIP path."""
    import os
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import main as cli
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())
    try:
        recs = cli.main(['--dataset=S-tiny', '--epochs=3', '--whetherHasSE=111', '--se_reg=0.5', '--want_headtail=1',
                         '--use_special_split=1', '--manual_assign_GPU=0'])
    finally:
        os.chdir(cwd)
    assert len(recs) == 1 and recs[0].shape == (4, 3) and np.isfinite(recs[0]).all()


@pytest.mark.parametrize('extra', [['--dataset=S-tiny', '--want_headtail=1', '--use_special_split=1', '--whetherHasSE=111', '--se_reg=0.5'],
                                   ['--dataset=S-pubmed', '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0'],
                                   # round 5: the other two default trunk shapes through their fused nodes (trunk 'Residual', stack.py NoRes) under replay
                                   ['--dataset=S-pubmed', '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0', '--force_set_to_best_config=0',
                                    '--type_trick=Residual'],
                                   ['--dataset=S-pubmed', '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0', '--force_set_to_best_config=0',
                                    '--type_trick=NoResNodeNorm']])
def test_epoch_loop_replayed_as_hip_graphs_matches_eager(extra):
    """--hip_graph=1: run_trainSet's step and run_testSet's eval forward replayed as hipGraphs give the records of the eager epoch
    loop (dropout 0 so that both draw the same data gradients): test accuracy and head/tail/isolation rows, final weights."""
    import contextlib
    import io
    import os
    import tempfile
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    cwd = os.getcwd()
    recs, sds = [], []
    try:
        for hg in (0, 1):
            os.chdir(tempfile.mkdtemp())
            with contextlib.redirect_stdout(io.StringIO()):
                args = BaseOptions().get_arguments(extra + ['--epochs=6', '--manual_assign_GPU=0', f'--hip_graph={hg}'])
                args.random_seed = 0
                torch.manual_seed(0)
                np.random.seed(0)
                t = trainer(args, 0)
                t.args.dropout = 0.0
                torch.manual_seed(0)
                recs.append(t.main())
            assert (getattr(t, '_hip_graph', None) is not None) == bool(hg) and (getattr(t, '_eval_graph', None) is not None) == bool(hg)
            sds.append({k: v.detach().clone() for k, v in t.teacherGNN.state_dict().items()})
            ops.set_graph_seed(None)
    finally:
        os.chdir(cwd)
    assert recs[0].shape == recs[1].shape and np.isfinite(recs[1]).all()
    np.testing.assert_allclose(recs[1], recs[0], atol=1e-3)
    for k, v in sds[0].items():
        if v.dtype.is_floating_point:
            # (under replay the trainer's graph takes the row-sparse plan and the rows-only forward at any size, the eager loop on these small graphs does not:
            # a pre-activation that is zero to rounding may fall on either side of the ReLU — tests/test_gpu_rowsparse.py::_close_up_to_relu_flips — and Adam
            # turns a gradient that moved by 1e-3 of its norm into a step that moved by about as much of lr = 0.01, six times)
            torch.testing.assert_close(sds[1][k], v, atol=2e-4, rtol=1e-3, msg=lambda m, k=k: f'{k}: {m}')


def test_sharded_trainer_world1_matches_plain_trainer():
    """The node-sharded code path (rectangular row-slice CSR, all-gather exchange, grad all-reduce) on one
    rank reproduces the plain trainer's losses on the same seeded data."""
    import contextlib
    import io
    import os
    import torch.distributed as dist
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.dist import ShardedTrainer
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    argv = ['--dataset=S-tiny', '--use_special_split=0', '--want_headtail=0', '--whetherHasSE=111', '--se_reg=0.5',
            '--num_layers=3', '--manual_assign_GPU=0', '--do_deg_analyze=0']
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if not dist.is_initialized():
        from gnn_tail_generalization_amd.dist import init_rccl
        init_rccl(0, 1, DEV)
    try:
        losses, sd0 = [], None
        for cls in (trainer, ShardedTrainer):
            with contextlib.redirect_stdout(io.StringIO()):
                args = BaseOptions().get_arguments(argv)
                args.random_seed = 0
                torch.manual_seed(0)
                t = cls(args, 0)
                torch.manual_seed(0)
                t.setup_teacherGNN()
            if sd0 is None:
                sd0 = {k: v.detach().clone() for k, v in t.teacherGNN.state_dict().items()}
            else:       # a fresh sharded model re-draws its per-node tables per rank (dist.sync_initial_state): start from the same state
                t.load_full_state_dict(sd0)
            ops._seed_override[:] = list(range(500, 530))
            losses.append([float(t.train_step()) for _ in range(4)])
            ops._seed_override[:] = []
        np.testing.assert_allclose(losses[0], losses[1], rtol=1e-5)
    finally:
        dist.destroy_process_group()


def test_rccl_carries_the_exchange_calls_in_the_forms_the_halo_plan_issues():
    """The multi-process tests on a one-GPU box stage the exchange through gloo; the collectives themselves — `all_to_all_single` on
    2-D device tensors with uneven (and zero) split lists, asynchronous work handles waited on later, bf16 rows moved as bytes, the
    small all-reduces / all-gathers — run here on RCCL (world size 1: every rank-pair list has one entry, the rows loop back)."""
    import os
    import torch.distributed as dist
    from gnn_tail_generalization_amd import dist as cbdist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    assert not dist.is_initialized()
    cbdist.init_rccl(0, 1, DEV)
    try:
        assert dist.get_backend() == 'nccl'
        g = torch.Generator(device='cpu').manual_seed(5)
        flights = []
        for n in (0, 1, 777, 40_001):                                   # several slices in flight, as start_halo leaves them
            send = torch.randn(n, 256, generator=g).to(DEV)
            recv = torch.empty((max(n, 1), 256), dtype=torch.float32, device=DEV)
            work = cbdist._all_to_all_single(recv[:n], send, [n], [n], async_op=True)
            flights.append((send, recv, n, work))
            torch.mm(torch.ones(512, 512, device=DEV), torch.ones(512, 512, device=DEV))      # compute issued under the exchange
        for send, recv, n, work in flights:
            work.wait()
            assert torch.equal(recv[:n], send)
        send = torch.randn(333, 256, generator=g).to(DEV).to(torch.bfloat16)                    # bf16 wire: moves as bytes
        recv = torch.empty_like(send)
        cbdist._all_to_all_single(recv.view(torch.uint8), send.view(torch.uint8), [333], [333], async_op=True).wait()
        assert torch.equal(recv, send)
        cnt = torch.tensor([[3, 1, 4]], dtype=torch.int64, device=DEV)                          # plan building: counts, then index lists
        got = torch.empty_like(cnt)
        cbdist._all_to_all_single(got.view(-1), cnt.view(-1))
        assert torch.equal(got, cnt)
        t = torch.arange(6, dtype=torch.float32, device=DEV)
        cbdist._all_reduce(t)
        assert torch.equal(t.cpu(), torch.arange(6, dtype=torch.float32))
        m = torch.tensor([7], dtype=torch.int64, device=DEV)
        cbdist._all_reduce(m, op=dist.ReduceOp.MAX)
        assert int(m) == 7
        out = torch.empty(6, dtype=torch.float32, device=DEV)
        cbdist._all_gather_into_tensor(out, t)
        assert torch.equal(out, t)
        cbdist._broadcast(t, src=0)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _graph_trainer(dropout):
    import contextlib
    import io
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-pubmed', '--use_special_split=0', '--want_headtail=0', '--whetherHasSE=111',
                                            '--se_reg=0.5', '--num_layers=2', '--manual_assign_GPU=0', '--do_deg_analyze=0'])
        args.random_seed = 0
        torch.manual_seed(0)
        t = trainer(args, 0)
        t.args.dropout = dropout
        torch.manual_seed(0)
        t.setup_teacherGNN()
    ops.set_graph_seed(None)
    return t


def test_hip_graph_step_matches_eager():
    """The training step captured as a hipGraph (device-resident Adam step count) reproduces the eager trajectory
    (dropout 0), and with dropout > 0 successive replays draw different masks (device-resident seed advances)."""
    from gnn_tail_generalization_amd import ops
    try:
        eager = _graph_trainer(0.0)
        ref = [float(eager.train_step()) for _ in range(6)]
        g = _graph_trainer(0.0)
        g.enable_hip_graph(warmup=2)                       # 2 eager warm-up steps, then replays
        got = [float(g.train_step()) for _ in range(4)]
        np.testing.assert_allclose(got, ref[2:], rtol=2e-5)
        for (k, a), (_, b) in zip(eager.teacherGNN.state_dict().items(), g.teacherGNN.state_dict().items()):
            if a.dtype.is_floating_point:
                torch.testing.assert_close(a, b, atol=1e-5, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')
        d = _graph_trainer(0.5)
        d.enable_hip_graph(warmup=1)
        losses = [float(d.train_step()) for _ in range(4)]
        assert np.isfinite(losses).all() and len(set(round(v, 6) for v in losses)) == 4
    finally:
        ops.set_graph_seed(None)


def test_label_propagation_matches_reference_fixture():
    """§8f row 4: --train_which=LP on the HIP aggregation reproduces the reference's outcome_correlation output."""
    import contextlib
    import io
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import Data
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    g = load_golden('lp_fixture')
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-tiny', '--train_which=LP', '--manual_assign_GPU=0'])
    n = g['y'].shape[0]
    t = trainer.__new__(trainer)
    t.args, t.device = args, torch.device(DEV)
    t.data = Data(x=torch.zeros(n, 2), y=g['y'], edge_index=g['edge_index'], train_mask=g['train_mask']).to(DEV)
    with contextlib.redirect_stdout(io.StringIO()):
        res = t.run_pureLP()
    assert torch.equal(t.data.edge_index.cpu(), g['edge_index_undirected'])
    torch.testing.assert_close(t.lp_out.cpu(), g['out'], atol=1e-5, rtol=1e-5)
    assert res.shape == (1, 2) and res[0].tolist() == g['acc'].tolist()


@pytest.mark.parametrize('hip_graph', [0, 1])
def test_resume_from_checkpoint_continues_bit_for_bit(hip_graph):
    """§8f row 3: weights + fused-Adam state + RNG are checkpointed; 3 epochs, resume, 2 more == 5 straight epochs — also when the
    epochs are replayed as hipGraphs (--hip_graph=1: the device-resident dropout seed word is part of the checkpoint)."""
    import os
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import main as cli
    common = ['--dataset=S-tiny', '--whetherHasSE=111', '--se_reg=0.5', '--want_headtail=0', '--use_special_split=0',
              '--do_deg_analyze=0', '--manual_assign_GPU=0', f'--hip_graph={hip_graph}']
    cwd = os.getcwd()
    try:
        os.chdir(tempfile.mkdtemp())
        straight = cli.main(common + ['--epochs=5'])[0]
        os.chdir(tempfile.mkdtemp())
        cli.main(common + ['--epochs=3'])
        resumed = cli.main(common + ['--epochs=5', '--resume'])[0]
    finally:
        os.chdir(cwd)
    assert straight.shape == resumed.shape == (1, 5)
    np.testing.assert_array_equal(straight, resumed)


def test_teacher_side_of_semlp_part1_handoff(tmp_path):
    """--train_which=SEMLP runs the teacher side of the reference's train_seMLP_part1 (:66-87) as one path —
    train_teacherGNN (best-test-accuracy weights saved, :331-334) -> load_teacherGNN('best checkpoint') -> collect_SE -> the
    replacement hand-off — and every link is checked: the best checkpoint is the state of the best epoch, teacherSE equals the
    ORACLE's TRAIN-mode collect_SE of those weights (the reference draws the targets with dropout active), replacement() equals the oracle's restatement of SEMLP.replacement (same selections)."""
    import contextlib
    import io
    import os
    import coldbrew_oracle as orc
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            args = BaseOptions().get_arguments(['--dataset=S-tiny', '--train_which=SEMLP', '--epochs=6', '--whetherHasSE=111', '--se_reg=0.5',
                                                '--want_headtail=0', '--use_special_split=0', '--do_deg_analyze=0', '--manual_assign_GPU=0'])
            torch.manual_seed(0)
            t = trainer(args, 0)
            # dropout seeds of the hand-off forward (drawn after training from torch's CPU stream) are recorded as they are drawn
            from gnn_tail_generalization_amd import ops
            drawn, real_next = [], ops.next_seed
            ops.next_seed = lambda: drawn.append(real_next()) or drawn[-1]
            try:
                rows = t.main()
            finally:
                ops.next_seed = real_next
        K = args.SEMLP_topK_2_replace          # 2 by default; 3 only under --unify_mlps (base_options.py:450-471)
        assert rows.shape == (1, 6) and K == 2 and t.topK_2_replace == K
        files = set(os.listdir(t.modeldir))
        assert {'best-teacherGNN', 'teacherGNN', 'teacherSE.pt'} <= files, files
        best = torch.load(os.path.join(t.modeldir, 'best-teacherGNN'), map_location='cpu')
        for k, v in t.teacherGNN.state_dict().items():          # the model in memory IS the reloaded best checkpoint
            assert torch.equal(v.cpu(), best[k]), k
        # teacherSE against the oracle's collect_SE (per-layer pre-activation outputs) of those weights
        n = int(t.data.x.shape[0])
        cfg = orc.make_cfg(type_trick=args.type_trick, num_layers=args.num_layers, num_feats=args.num_feats, dim_hidden=args.dim_hidden,
                           num_classes=args.num_classes, res_alpha=args.res_alpha, whetherHasSE=tuple(args.TeacherGNN.whetherHasSE))
        csr = orc.build_csr(t.data.edge_index.cpu(), n)
        # ... in TRAIN mode, as the reference draws them (its load_teacherGNN builds a new module and never calls eval(): ADVICE r03), with
        # the product's keep-masks of that forward injected into the oracle (the last L + 2 seeds drawn: x, X0, the layer outputs)
        assert t.teacherGNN.training and args.dropout > 0
        L, H, F_ = args.num_layers, args.dim_hidden, args.num_feats
        shapes = [(n, F_)] + [(n, H)] * L + [(n, H)]
        seeds = drawn[-len(shapes):]
        masks = [ops.dropout_keep_mask(sh, args.dropout, sd_, DEV).cpu() for sh, sd_ in zip(shapes, seeds)]
        cfg.dropout = args.dropout
        _, _, les = orc.trickscomb_forward(cfg, orc.strip_prefix(best), t.data.x.cpu(), csr, training=True, dropout_masks=masks, want_les=True)
        _, _, les_eval = orc.trickscomb_forward(cfg, orc.strip_prefix(best), t.data.x.cpu(), csr, training=False, want_les=True)
        assert float((les - les_eval).abs().max()) > 1e-3          # the dropout really is part of the targets
        assert tuple(t.teacherSE.shape) == tuple(les.shape) and t.teacherSE.shape[1] == t.teacherGNN.model.model.get_se_dim(t.data.x, t.data.edge_index)
        torch.testing.assert_close(t.teacherSE.cpu(), les, atol=1e-4, rtol=1e-4)
        saved = torch.load(os.path.join(t.modeldir, 'teacherSE.pt'), weights_only=True)
        assert torch.equal(saved['teacherSE'], t.teacherSE.cpu()) and saved['topK_2_replace'] == K
        # the student's part-1 output stand-in: noisy teacher rows -> virtual-neighbour replacement
        gen = torch.Generator().manual_seed(5)
        guess = (t.teacherSE.cpu()[:50] + 0.05 * torch.randn(50, les.shape[1], generator=gen))
        out, idx, wgt = t.replacement(guess.to(DEV), return_selection=True)
        want, sel, w = orc.semlp_replacement(guess, t.teacherSE.cpu(), K)
        assert torch.equal(torch.sort(idx.cpu().long(), 1)[0], torch.sort(sel, 1)[0])
        torch.testing.assert_close(out.cpu(), want, atol=1e-4, rtol=1e-4)
        sub = t.replacement(guess.to(DEV), node_idx=[3, 7])
        torch.testing.assert_close(sub.cpu(), want[[3, 7]], atol=1e-4, rtol=1e-4)
    finally:
        os.chdir(cwd)
