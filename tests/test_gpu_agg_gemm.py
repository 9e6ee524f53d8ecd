"""GPU: aggregation + next dense transform in one kernel (csrc/cb_agg_gemm.hip) against the two kernels it
replaces.  The aggregation part is k_spmm_rows' own edge-stream walk and the dense part repeats cb_gemm_limb.hip's limb products
in the same order, so everything is compared BIT FOR BIT with cb_spmm_csr_f32 / cb_spmm_csr_fused_f32 followed by cb_gemm_nn_f32
(which are pinned to the oracle in test_gpu_graph_spmm.py / test_gpu_kernels.py); the oracle itself is compared once more at the
end to end (logits, loss, gradients of a golden case through the fused trunk with the kernels on)."""
import contextlib
import io

import numpy as np
import pytest
import torch

import coldbrew_oracle as orc
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _powerlaw_graph(n, seed, T, sym=True):
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    data = synthetic_data('S-pl1M', seed=seed, device=DEV, n_override=n)
    ei = data.edge_index
    if not sym:      # directed multigraph: drop a third of the edges, duplicate some (own reverse orientation, duplicates counted)
        gen = torch.Generator(device=DEV).manual_seed(seed)
        keep = torch.rand(ei.shape[1], device=DEV, generator=gen) < 0.66
        ei = torch.cat([ei[:, keep], ei[:, :ei.shape[1] // 7], torch.arange(n, device=DEV).repeat(2, 1)], dim=1)
    return CSRGraph(ei, n, hub_threshold=T)


# 70001 rows = 1094 tiles on 256 persistent blocks: every block re-uses both LDS buffers (the freed / ready counters of the flag hand-over)
@pytest.mark.parametrize('n,T,sym', [(5003, 16, True), (64, 256, True), (20000, 256, True), (9001, 8, False), (1, 256, True), (70001, 64, True)])
@pytest.mark.parametrize('transpose', [False, True])
def test_spmm_gemm_equals_aggregation_then_gemm(n, T, sym, transpose):
    """cb_spmm_gemm_f32: out and g_out bit-identical to cb_spmm_csr_f32 followed by cb_gemm_nn_f32 — ragged last tile (N % 64 != 0),
    hub rows (tiny threshold: most blocks contain rows finished by the hub kernels), directed multigraph / reverse orientation."""
    from gnn_tail_generalization_amd import gemm
    from gnn_tail_generalization_amd.graph import CSRGraph, weight_image
    if n == 1:
        G = CSRGraph(torch.zeros((2, 1), dtype=torch.int64, device=DEV), 1)
    else:
        G = _powerlaw_graph(n, 3, T, sym)
    if T <= 16:
        assert G._plan.n_hubs > 0
    gen = torch.Generator(device=DEV).manual_seed(5)
    h = torch.randn(n, 256, device=DEV, generator=gen)
    w = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    rs = torch.rand(n, device=DEV, generator=gen) + 0.5
    bias = torch.randn(256, device=DEV, generator=gen)
    a = torch.rand(n, device=DEV, generator=gen) + 0.5
    le = torch.randn(n, 256, device=DEV, generator=gen)
    for tw in (False, True):
        B = w.t().contiguous() if tw else w
        img = weight_image(w, transpose=tw)
        out, g = G.spmm_gemm(h, img, transpose=transpose, row_scale=rs, bias=bias, relu=True, g_rowscale=a, g_addend=le)
        ref = G.spmm(h, transpose=transpose, row_scale=rs, bias=bias, relu=True)
        assert torch.equal(out, ref)
        assert torch.equal(g, gemm.mm_nn(ref, B, rowscale=a, addend=le))
    # the backward's form: raw sums, scale only
    out, g = G.spmm_gemm(h, weight_image(w, transpose=True), transpose=transpose, g_rowscale=a)
    ref = G.spmm(h, transpose=transpose)
    assert torch.equal(out, ref) and torch.equal(g, gemm.mm_nn(ref, w.t().contiguous(), rowscale=a))
    # and against fp64 (the dense tail is an fp32-grade GEMM)
    want = (ref.double() @ w.t().double()) * a.double().unsqueeze(1)
    torch.testing.assert_close(g.double(), want, atol=2e-5 * float(want.abs().max()) + 1e-6, rtol=1e-5)


def test_tile_hand_over_is_race_free_under_repetition():
    """The LDS tiles of the persistent kernel change hands through two counters per buffer instead of a block barrier.  200 003 rows = 3126 tiles
    = 12 per block: every launch makes ~37 000 hand-overs; 40 launches (plain and reverse orientation alternating) all reproduce the two-kernel
    result bit for bit — a lost or early hand-over would show as a wrong 64-row tile."""
    from gnn_tail_generalization_amd import gemm
    from gnn_tail_generalization_amd.graph import weight_image
    n = 200003
    G = _powerlaw_graph(n, 9, 256)
    gen = torch.Generator(device=DEV).manual_seed(21)
    h = torch.randn(n, 256, device=DEV, generator=gen)
    w = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    a = torch.rand(n, device=DEV, generator=gen) + 0.5
    img = weight_image(w)
    refs = {}
    for tr in (False, True):
        ro = G.spmm(h, transpose=tr)
        refs[tr] = (ro, gemm.mm_nn(ro, w, rowscale=a))
    for i in range(40):
        tr = bool(i & 1)
        out, g = G.spmm_gemm(h, img, transpose=tr, g_rowscale=a)
        assert torch.equal(out, refs[tr][0]) and torch.equal(g, refs[tr][1]), f'launch {i}'


@pytest.mark.parametrize('n,T,p', [(5003, 16, 0.1), (20000, 256, 0.0), (777, 8, 0.3), (70001, 64, 0.2)])
def test_fused_store_gemm_equals_fused_store_then_gemm(n, T, p):
    """cb_spmm_gemm_fused_f32: mask words, out_next and Z_next bit-identical to cb_spmm_csr_fused_f32 followed by cb_gemm_nn_f32."""
    from gnn_tail_generalization_amd import gemm, trunk
    from gnn_tail_generalization_amd.graph import weight_image
    G = _powerlaw_graph(n, 9, T)
    gen = torch.Generator(device=DEV).manual_seed(6)
    z = torch.randn(n, 256, device=DEV, generator=gen)
    x0 = torch.randn(n, 256, device=DEV, generator=gen)
    w = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    bias = torch.randn(256, device=DEV, generator=gen)
    le = torch.randn(n, 256, device=DEV, generator=gen)
    a = G.norm_out
    for mix, addend in ((x0, le), (None, None)):
        bits, nxt, zn = trunk._fused_gemm_launch(G, z, bias, mix, 0.9, 0.1, p, 4242, weight_image(w), a, addend)
        bits_r, nxt_r, _ = trunk._fused_spmm(G, z, bias, mix, 0.9, 0.1, p, 4242)
        assert torch.equal(bits, bits_r) and torch.equal(nxt, nxt_r)
        assert torch.equal(zn, gemm.mm_nn(nxt_r, w, rowscale=a, addend=addend))
    # a forward that no backward follows (cb_spmm_gemm_fused_eval_f32): no mask words, the stored activations are not written, Z_next the same bits
    b2, n2, z2 = trunk._fused_gemm_launch(G, z, bias, x0, 0.9, 0.1, p, 4242, weight_image(w), a, le, want_bits=False)
    nxt_mix = trunk._fused_spmm(G, z, bias, x0, 0.9, 0.1, p, 4242)[1]
    assert b2 is None and n2 is None and torch.equal(z2, gemm.mm_nn(nxt_mix, w, rowscale=a, addend=le))


@pytest.mark.parametrize('C', [40, 47, 3, 64, 7])
@pytest.mark.parametrize('n,T,p', [(5003, 16, 0.1), (70001, 64, 0.2), (777, 256, 0.0)])
def test_head_linear_as_the_tail_of_the_last_aggregation(n, T, p, C):
    """cb_spmm_gemm_fused_head_f32 (round 5): the LAST layer's trunk store and the output Linear (GCN.py:133-138) from one
    kernel — mask words and the stored activations bit-identical to cb_spmm_csr_fused_f32, the logits bit-identical to cb_gemm_nn_f32 on those
    activations (same limb products in the same order), for class counts that are / are not multiples of 4, with hub rows, on top of running sums
    (node-sharded last pass), and in the evaluation form, which writes no activations at all."""
    from gnn_tail_generalization_amd import _lib, gemm, trunk
    from gnn_tail_generalization_amd.graph import head_image
    G = _powerlaw_graph(n, 9, T)
    gen = torch.Generator(device=DEV).manual_seed(8 + C)
    z = torch.randn(n, 256, device=DEV, generator=gen)
    x0 = torch.randn(n, 256, device=DEV, generator=gen)
    bias = torch.randn(256, device=DEV, generator=gen)
    w_out = torch.randn(C, 256, device=DEV, generator=gen) * 0.07          # nn.Linear layout
    b_out = torch.randn(C, device=DEV, generator=gen)
    img = head_image(w_out)
    assert img is not None and head_image(torch.randn(65, 256, device=DEV)) is None
    bits_r, nxt_r, _ = trunk._fused_spmm(G, z, bias, x0, 0.9, 0.1, p, 4242)
    want = gemm.mm_nn(nxt_r, w_out.t().contiguous(), bias=b_out)
    bits, nxt, logits = trunk._fused_gemm_launch(G, z, bias, x0, 0.9, 0.1, p, 4242, img, None, None, head=(b_out, C))
    assert logits.shape == (n, C) and torch.equal(bits, bits_r) and torch.equal(nxt, nxt_r)
    ref64 = nxt_r.double() @ w_out.t().double() + b_out.double()

    def same(got, ref):
        # class counts that are multiples of 4: cb_gemm_nn_f32 runs the same limb kernel -> the same bits; others send it to its fp32-input fallback
        # (rows of C floats are not 16-byte aligned), so the two agree as two fp32-grade GEMMs do — and the tail is the closer one to fp64
        if C % 4 == 0:
            return torch.equal(got, ref)
        tol = 2e-5 * float(ref64.abs().max())
        return float((got.double() - ref64).abs().max()) <= tol and float((got - ref).abs().max()) <= 2 * tol
    assert same(logits, want), float((logits - want).abs().max())
    b2, n2, l2 = trunk._fused_gemm_launch(G, z, bias, x0, 0.9, 0.1, p, 4242, img, None, None, want_bits=False, head=(b_out, C))
    assert b2 is None and n2 is None and torch.equal(l2, logits)
    # on top of the partial sums of earlier passes (the last halo pass of a node-sharded aggregation)
    acc = torch.randn(n, 256, device=DEV, generator=gen)
    rb, rn, _ = trunk._fused_launch(_lib.load(), G, G, z, acc.clone(), bias, x0, 0.9, 0.1, p, 4242, False)
    b3, n3, l3 = trunk._fused_gemm_launch(G, z, bias, x0, 0.9, 0.1, p, 4242, img, None, None, g=G, acc=acc.clone(), head=(b_out, C))
    ref64 = rn.double() @ w_out.t().double() + b_out.double()
    assert torch.equal(b3, rb) and torch.equal(n3, rn) and same(l3, gemm.mm_nn(rn, w_out.t().contiguous(), bias=b_out))


@pytest.mark.parametrize('n,T,p', [(5003, 16, 0.1), (20000, 256, 0.0)])
def test_reverse_aggregation_gemm_trunk_backward_equals_three_kernels(n, T, p):
    """cb_spmm_gemm_trunkbwd_f32: dL/dZ (reverse aggregation), dL/dx = a * (dL/dZ @ W^T) and the trunk backward of the layer below
    (dropout backward, ReLU / mix mask bits, c_act, degree norm) from ONE kernel: dL/dZ, dL/dx and the next aggregation's input are
    bit-identical to cb_spmm_csr_f32 + cb_gemm_nn_f32 + cb_trunk_layer_bwd_f32; the bias column sums agree to summation order."""
    from gnn_tail_generalization_amd import gemm, trunk
    from gnn_tail_generalization_amd.graph import weight_image
    G = _powerlaw_graph(n, 4, T)
    gen = torch.Generator(device=DEV).manual_seed(8)
    z = torch.randn(n, 256, device=DEV, generator=gen)
    bias = torch.randn(256, device=DEV, generator=gen)
    bits, _, _ = trunk._fused_spmm(G, z, bias, None, 1.0, 0.0, p, 777)            # mask words of a forward store
    gr_in = torch.randn(n, 256, device=DEV, generator=gen)
    w = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    a, b = G.norm_out, G.norm_in
    out, g, gr, cs = G.spmm_gemm_trunkbwd(gr_in, weight_image(w, transpose=True), a, bits, 0.9, p, 777, 0, b, True)
    ref_out = G.spmm(gr_in, transpose=True)
    ref_g = gemm.mm_nn(ref_out, w.t().contiguous(), rowscale=a)
    ref_gr, ref_cs = trunk._layer_bwd(ref_g, bits, b, None, False, p, 777, 0, 0.9, 0.1, True)
    assert torch.equal(out, ref_out) and torch.equal(g, ref_g) and torch.equal(gr, ref_gr)
    torch.testing.assert_close(cs, ref_cs, atol=1e-4 * float(ref_cs.abs().max()) + 1e-5, rtol=1e-5)
    _, _, _, none = G.spmm_gemm_trunkbwd(gr_in, weight_image(w, transpose=True), a, bits, 0.9, p, 777, 0, b, False)
    assert none is None


def test_lost_tile_hand_over_is_reported_not_silent():
    """VERDICT r03 item 1b / ADVICE r03: a wavefront that gives up the bounded wait for an LDS tile hand-over records it in the device error
    word; cb_device_status() (and every later cb_spmm_gemm_* launch) returns CB_E_DEVICE with the reason, then the word is clear again.
    Fault injection: cb_agg_gemm_handover_selftest waits, with a short spin bound, for a counter nobody increments."""
    from gnn_tail_generalization_amd import _lib, gemm
    from gnn_tail_generalization_amd.graph import weight_image
    lib = _lib.load()
    torch.cuda.synchronize()
    _lib.device_status()                                       # clear
    G = _powerlaw_graph(5003, 3, 256)
    h = torch.randn(5003, 256, device=DEV)
    img = weight_image(torch.randn(256, 256, device=DEV))
    ref = G.spmm_gemm(h, img)
    with torch.cuda.device(DEV):
        _lib.check(lib.cb_agg_gemm_handover_selftest(_lib.stream_ptr()), 'cb_agg_gemm_handover_selftest')
    torch.cuda.synchronize()
    with pytest.raises(_lib.HipExtensionError, match='device-side error'):      # the NEXT launch refuses to run on top of invalid results
        G.spmm_gemm(h, img)
    with pytest.raises(_lib.HipExtensionError, match='hand-over timed out'):
        _lib.device_status()
    _lib.device_status()                                       # cleared by the report
    out = G.spmm_gemm(h, img)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])


@pytest.mark.parametrize('n,T,p,fused', [(5003, 16, 0.1, True), (20000, 256, 0.0, True), (70001, 64, 0.2, False), (777, 8, 0.3, False)])
def test_acc_init_forms_equal_the_two_kernel_forms(n, T, p, fused):
    """The ACC forms (node-sharded path: the last halo pass of a rank's aggregation is the aggregation + GEMM kernel on top of the running
    sums of the earlier passes): cb_spmm_gemm_f32 / cb_spmm_gemm_fused_f32 / cb_spmm_gemm_trunkbwd_f32 with acc_init are bit-identical to
    cb_spmm_csr_acc_f32 / cb_spmm_csr_fused_acc_f32 followed by cb_gemm_nn_f32 (and cb_trunk_layer_bwd_f32) — hub rows included (the hub
    finish kernel adds the partial sums of its rows)."""
    from gnn_tail_generalization_amd import _lib, gemm, trunk
    from gnn_tail_generalization_amd.graph import weight_image
    G = _powerlaw_graph(n, 5, T)
    gen = torch.Generator(device=DEV).manual_seed(17)
    h = torch.randn(n, 256, device=DEV, generator=gen)
    acc = torch.randn(n, 256, device=DEV, generator=gen)
    w = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    bias = torch.randn(256, device=DEV, generator=gen)
    le = torch.randn(n, 256, device=DEV, generator=gen)
    x0 = torch.randn(n, 256, device=DEV, generator=gen)
    a, b = G.norm_out, G.norm_in
    if fused:
        lib = _lib.load()
        ref_bits, ref_next, _ = trunk._fused_launch(lib, G, G, h, acc.clone(), bias, x0, 0.9, 0.1, p, 4242, False)
        bits, nxt, zn = trunk._fused_gemm_launch(G, h, bias, x0, 0.9, 0.1, p, 4242, weight_image(w), a, le, g=G, acc=acc.clone())
        assert torch.equal(bits, ref_bits) and torch.equal(nxt, ref_next)
        assert torch.equal(zn, gemm.mm_nn(ref_next, w, rowscale=a, addend=le))
    else:
        ref = G.spmm(h, row_scale=b, bias=bias, relu=True, acc_init=acc.clone())
        buf = acc.clone()
        out, g = G.spmm_gemm(h, weight_image(w), row_scale=b, bias=bias, relu=True, g_rowscale=a, g_addend=le, acc_init=buf)
        assert out.data_ptr() == buf.data_ptr()               # the running sums are finished in place
        assert torch.equal(out, ref) and torch.equal(g, gemm.mm_nn(ref, w, rowscale=a, addend=le))
        # + the trunk backward from the tail's epilogue
        bits, _, _ = trunk._fused_spmm(G, h, bias, None, 1.0, 0.0, p, 777)
        ref_out = G.spmm(h, acc_init=acc.clone())
        ref_g = gemm.mm_nn(ref_out, w.t().contiguous(), rowscale=a)
        ref_gr, ref_cs = trunk._layer_bwd(ref_g, bits, b, None, False, p, 777, 0, 0.9, 0.1, True)
        out, g, gr, cs = G.spmm_gemm_trunkbwd(h, weight_image(w, transpose=True), a, bits, 0.9, p, 777, 0, b, True, transpose=False,
                                              acc_init=acc.clone())
        assert torch.equal(out, ref_out) and torch.equal(g, ref_g) and torch.equal(gr, ref_gr)
        torch.testing.assert_close(cs, ref_cs, atol=1e-4 * float(ref_cs.abs().max()) + 1e-5, rtol=1e-5)


def _step_losses(monkeypatch, flag, steps=3, n=30000):
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.trainer_node_classification import trainer
    monkeypatch.setenv('CB_AGG_GEMM', flag)
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(['--dataset=S-pl1M', '--num_layers=3', '--use_special_split=0', '--want_headtail=0',
                                            '--manual_assign_GPU=0', '--do_deg_analyze=0', '--whetherHasSE=111', '--se_reg=0.3'])
        t = trainer.__new__(trainer)
        t.args, t.bag, t.device = args, {}, torch.device(DEV)
        args.device = t.device
        t.data = synthetic_data('S-pl1M', seed=0, device=DEV, n_override=n)
        args.N_nodes = n
        from gnn_tail_generalization_amd.utils import set_arch_configs
        from gnn_tail_generalization_amd import optim
        set_arch_configs(args)
        torch.manual_seed(0)
        from gnn_tail_generalization_amd.GNN_model import TeacherGNN
        t.teacherGNN = TeacherGNN(args).to(DEV)
        t.optimizer = optim.resolve(args.optfun)(t.teacherGNN.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    ops._seed_override[:] = list(range(9100, 9100 + 8 * steps))
    losses = [float(t.train_step()) for _ in range(steps)]
    ops._seed_override[:] = []
    sd = {k: v.detach().clone() for k, v in t.teacherGNN.state_dict().items()}
    return losses, sd


def test_training_step_with_fused_kernels_equals_two_kernel_form(monkeypatch):
    """Three optimisation steps of the 3-layer 'Initial' trunk (hidden 256, structural embeddings on, dropout on) with the
    aggregation + GEMM kernels (CB_AGG_GEMM=1, default) and with the separate kernels (=0): same losses and parameters, bit for bit —
    with the trunk backward kept as a pass of its own (one-GPU default); with it in the fused kernel's epilogue (CB_AGG_GEMM_TRUNKBWD=1, the
    node-sharded default) the bias gradients are summed in another order, so parameters agree to rounding instead."""
    l0, sd0 = _step_losses(monkeypatch, '0')
    monkeypatch.setenv('CB_AGG_GEMM_TRUNKBWD', '0')
    l1, sd1 = _step_losses(monkeypatch, '1')
    assert l1 == l0
    for k in sd0:
        assert torch.equal(sd1[k], sd0[k]), k
    monkeypatch.delenv('CB_AGG_GEMM_TRUNKBWD')
    l1b, sd1b = _step_losses(monkeypatch, '1')                 # default == '0' on one GPU
    assert l1b == l0 and all(torch.equal(sd1b[k], sd0[k]) for k in sd0)
    monkeypatch.setenv('CB_AGG_GEMM_TRUNKBWD', '1')
    l2, sd2 = _step_losses(monkeypatch, '1')
    np.testing.assert_allclose(l2, l0, rtol=1e-6)
    for k in sd0:
        torch.testing.assert_close(sd2[k], sd0[k], atol=1e-5, rtol=1e-4, msg=lambda m, k=k: f'{k}: {m}')


def test_fused_kernels_against_the_oracle(monkeypatch):
    """End to end against the ORACLE (not against the product's other path): the benchmark configuration (3-layer 'InitialBatchNorm'
    trunk, hidden 256, structural embeddings on) on a 3000-node power-law graph with the aggregation + GEMM kernels on — logits 1e-4,
    loss 1e-5, every parameter gradient to the golden-case tolerances."""
    from gnn_tail_generalization_amd import ops, trunk
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    from gnn_tail_generalization_amd.utils import set_arch_configs
    monkeypatch.setenv('CB_AGG_GEMM', '1')
    monkeypatch.setenv('CB_SE_REG_FOLD', '0')           # the regulariser's gradient through autograd, so that le.grad is complete
    n = 3000
    with contextlib.redirect_stdout(io.StringIO()):
        a = BaseOptions().get_arguments(['--dataset=S-pl1M', '--num_layers=3', '--use_special_split=0', '--manual_assign_GPU=0',
                                         '--whetherHasSE=111', '--se_reg=0.5'])
    data = synthetic_data('S-pl1M', seed=1, device=DEV, n_override=n)
    a.N_nodes, a.dropout, a.device = n, 0.0, torch.device(DEV)
    set_arch_configs(a)
    torch.manual_seed(0)
    m = TeacherGNN(a).to(DEV)
    m.train()
    out = m(data.x, data.edge_index)
    graph = m.model.model._graph(data.edge_index)
    assert trunk.agg_gemm_eligible(graph, 256, False) and not trunk.agg_gemm_eligible(graph, 64, False)
    loss = ops.nll_logsoftmax(out, data.y, data.train_mask) + a.se_reg * m.se_reg_all
    loss.backward()
    cfg = orc.make_cfg(type_trick=a.type_trick, num_layers=3, num_feats=a.num_feats, dim_hidden=a.dim_hidden, num_classes=a.num_classes,
                       res_alpha=a.res_alpha, whetherHasSE=(1, 1, 1), se_reg=a.se_reg)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    csr = orc.build_csr(data.edge_index.cpu(), n)
    o, reg = orc.teacher_forward(cfg, sd, data.x.cpu(), csr, training=True)
    ref = orc.training_loss(cfg, o, reg, data.y.cpu(), data.train_mask.cpu())
    ref.backward()
    torch.testing.assert_close(out.detach().cpu(), o.detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), atol=1e-5, rtol=1e-5)
    checked = 0
    for k, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        want = sd[k].grad
        assert want is not None, k
        torch.testing.assert_close(p_.grad.cpu(), want, atol=2e-5, rtol=2e-4, msg=lambda s_, k=k: f'{k}: {s_}')
        checked += 1
    assert checked >= 10
