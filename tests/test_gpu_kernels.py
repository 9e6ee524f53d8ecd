"""GPU parity of the dense / elementwise / optimiser kernels (through the C ABI) against fp64 torch
references and the oracle's Adam.  fp32 tolerances are written per test."""
import numpy as np
import pytest
import torch

import coldbrew_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize('M,K,N', [(1, 1, 1), (5, 3, 7), (128, 16, 128), (130, 17, 129), (257, 1433, 64), (1000, 500, 256),
                                   (333, 256, 40), (64, 100, 47), (4096, 256, 256), (300, 64, 7)])
def test_gemm_nn_epilogues(M, K, N):
    from gnn_tail_generalization_amd import gemm
    a, b = _rand(M, K, seed=1), _rand(K, N, seed=2)
    rs, add, bias = torch.rand(M) + 0.5, _rand(M, N, seed=3), _rand(N, seed=4)
    ref = a.double() @ b.double()
    tol = dict(atol=2e-5 * max(1, K ** 0.5), rtol=2e-5)
    out = gemm.mm_nn(a.to(DEV), b.to(DEV))
    torch.testing.assert_close(out.cpu().double(), ref, **tol)
    out = gemm.mm_nn(a.to(DEV), b.to(DEV), rowscale=rs.to(DEV), addend=add.to(DEV), bias=bias.to(DEV), relu=True)
    ref2 = torch.relu(ref * rs.double().unsqueeze(1) + add.double() + bias.double())
    torch.testing.assert_close(out.cpu().double(), ref2, **tol)


@pytest.mark.parametrize('M,K,N', [(2708, 1433, 64), (3327, 3703, 256), (500, 2048, 128)])
def test_gemm_nn_split_k_small_m_long_k(M, K, N):
    """Input Linear of a Cora / Citeseer-sized graph (GCN.py:105): few output tiles, long contraction -> the fp32-input kernel cuts
    K over blocks (cb_gemm_nn_splitk_workspace_bytes > 0) and applies the epilogue after the fixed-order sum of the planes."""
    from gnn_tail_generalization_amd import _lib, gemm
    lib = _lib.load()
    a, b = _rand(M, K, seed=1).to(DEV), _rand(K, N, seed=2).to(DEV)
    if K % 4 == 0:
        a = torch.nn.functional.pad(a, (0, 1))[:, :K]          # odd leading dimension: keeps the shape on the fp32-input kernel
    assert lib.cb_gemm_nn_splitk_workspace_bytes(M, N, K) > 0
    rs, add, bias = (torch.rand(M) + 0.5).to(DEV), _rand(M, N, seed=3).to(DEV), _rand(N, seed=4).to(DEV)
    ref = torch.relu((a.double() @ b.double()) * rs.double().unsqueeze(1) + add.double() + bias.double())
    out = gemm.mm_nn(a, b, rowscale=rs, addend=add, bias=bias, relu=True)
    torch.testing.assert_close(out.double(), ref, atol=2e-5 * K ** 0.5, rtol=2e-5)
    assert torch.equal(out, gemm.mm_nn(a, b, rowscale=rs, addend=add, bias=bias, relu=True))      # fixed summation order
    # same entry point without the workspace: one block per tile walks the whole K range
    one = torch.empty_like(out)
    _lib.check(lib.cb_gemm_nn_f32(_lib.ptr(a), a.stride(0), _lib.ptr(b), N, _lib.ptr(one), N, M, N, K, _lib.ptr(rs), _lib.ptr(add), N,
                                  _lib.ptr(bias), 1, None, 0, _lib.stream_ptr()), 'cb_gemm_nn_f32')
    torch.testing.assert_close(out, one, atol=2e-5 * K ** 0.5, rtol=2e-5)
    assert lib.cb_gemm_nn_splitk_workspace_bytes(100000, N, K) == 0 and lib.cb_gemm_nn_splitk_workspace_bytes(M, N, 256) == 0


def test_gemm_nn_asymmetric_identity_and_strides():
    """A = I against an asymmetric B catches a transposed C write; strided (non-16B) operands take the generic path."""
    from gnn_tail_generalization_amd import gemm
    n = 160
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 13.0
    out = gemm.mm_nn(torch.eye(n, device=DEV), b.to(DEV))
    assert torch.equal(out.cpu(), b)
    big = _rand(200, 301, seed=5).to(DEV)
    a = big[:, 3:203]                      # ld = 301, misaligned start
    w = _rand(200, 33, seed=6).to(DEV)
    torch.testing.assert_close(gemm.mm_nn(a, w).cpu().double(), a.cpu().double() @ w.cpu().double(), atol=3e-4, rtol=2e-5)


@pytest.mark.parametrize('M,K1,K2', [(1, 1, 1), (77, 5, 9), (1000, 128, 256), (5000, 256, 256), (3001, 1433, 64), (20000, 40, 256),
                                     (999, 130, 7)])
def test_gemm_tn(M, K1, K2):
    from gnn_tail_generalization_amd import gemm
    a, g, rs = _rand(M, K1, seed=1), _rand(M, K2, seed=2), torch.rand(M) + 0.5
    ref = a.double().t() @ (g.double() * rs.double().unsqueeze(1))
    out = gemm.mm_tn(a.to(DEV), g.to(DEV), rowscale=rs.to(DEV))
    torch.testing.assert_close(out.cpu().double(), ref, atol=3e-5 * max(1, M ** 0.5), rtol=3e-5)
    out2 = gemm.mm_tn(a.to(DEV), g.to(DEV), rowscale=rs.to(DEV))
    assert torch.equal(out, out2)          # fixed reduction order


def test_linear_autograd_matches_torch():
    from gnn_tail_generalization_amd import gemm
    x, w, b = _rand(300, 50, seed=1), _rand(20, 50, seed=2), _rand(20, seed=3)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.relu(torch.nn.functional.linear(xr, wr, br))
    go = _rand(300, 20, seed=4)
    yr.backward(go.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = gemm.linear(xd, wd, bd, relu=True)
    y.backward(go.to(DEV))
    for got, ref in [(y, yr), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)]:
        torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), atol=1e-4, rtol=1e-4)
    rs, e = torch.rand(300) + 0.5, _rand(300, 30, seed=7)
    w2 = _rand(50, 30, seed=8)
    xr, w2r, er = (t.double().requires_grad_(True) for t in (x, w2, e))
    zr = (xr * rs.double().unsqueeze(1)) @ w2r + er
    g2 = _rand(300, 30, seed=9)
    zr.backward(g2.double())
    xd, w2d, ed = (t.to(DEV).requires_grad_(True) for t in (x, w2, e))
    z = gemm.linear_rowscale(xd, w2d, rs.to(DEV), ed)
    z.backward(g2.to(DEV))
    for got, ref in [(z, zr), (xd.grad, xr.grad), (w2d.grad, w2r.grad), (ed.grad, er.grad)]:
        torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('n', [1, 3, 4, 1023, 4096 * 37 + 5])
def test_dropout_properties(n):
    from gnn_tail_generalization_amd import ops
    x = (torch.rand(n) + 0.5).to(DEV).requires_grad_(True)
    p, seed = 0.3, 123456789
    y = ops.dropout(x, p, True, seed=seed)
    keep = ops.dropout_keep_mask((n,), p, seed, DEV)
    torch.testing.assert_close(y.detach(), torch.where(keep, x.detach() / (1 - p), torch.zeros_like(x)), rtol=1e-6, atol=0)
    assert torch.equal(y, ops.dropout(x, p, True, seed=seed))                 # pure function of (seed, index)
    if n > 1000:
        assert abs(float(keep.float().mean()) - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5
        assert not torch.equal(keep, ops.dropout_keep_mask((n,), p, seed + 1, DEV))
    g = torch.rand(n, device=DEV)
    y.backward(g)
    torch.testing.assert_close(x.grad, torch.where(keep, g / (1 - p), torch.zeros_like(g)), rtol=1e-6, atol=0)
    assert ops.dropout(x, p, False) is x and ops.dropout(x, 0.0, True) is x
    # row shards reproduce the full mask (offset = flat index of the shard's first element)
    for off in (0, 1, 2, 3, 8):
        if off < n:
            part = ops.dropout_keep_mask((n - off,), p, seed, DEV, offset=off)
            assert torch.equal(part, keep[off:])


def test_axpby_act_bwd_frobenius():
    from gnn_tail_generalization_amd import ops
    for shape in [(7, 3), (100, 40), (513, 256), (64, 260), (33, 7)]:
        x, y = _rand(*shape, seed=1).to(DEV).requires_grad_(True), _rand(*shape, seed=2).to(DEV).requires_grad_(True)
        out = ops.axpby(0.9, x, 0.1, y)
        torch.testing.assert_close(out, 0.9 * x + 0.1 * y, rtol=1e-6, atol=1e-7)
        out.backward(torch.ones_like(out))
        assert torch.allclose(x.grad, torch.full_like(x, 0.9)) and torch.allclose(y.grad, torch.full_like(y, 0.1))
        g, act, rs = _rand(*shape, seed=3), _rand(*shape, seed=4), torch.rand(shape[0]) + 0.5
        gs, cs = ops.act_bwd(g.to(DEV), act.to(DEV), rs.to(DEV), want_out=True, want_colsum=True)
        gm = g.double() * (act > 0)
        torch.testing.assert_close(gs.cpu().double(), gm * rs.double().unsqueeze(1), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(cs.cpu().double(), gm.sum(0), rtol=1e-5, atol=1e-4)
        gs2, cs2 = ops.act_bwd(g.to(DEV), None, None, want_out=False, want_colsum=True)
        assert gs2 is None
        torch.testing.assert_close(cs2.cpu().double(), g.double().sum(0), rtol=1e-5, atol=1e-4)
        le = _rand(*shape, seed=5).to(DEV).requires_grad_(True)
        nrm = ops.frobenius_norm(le)
        torch.testing.assert_close(nrm.cpu(), torch.norm(le.detach().cpu()), rtol=1e-6, atol=1e-6)
        (3.0 * nrm).backward()
        torch.testing.assert_close(le.grad.cpu(), 3.0 * le.detach().cpu() / torch.norm(le.detach().cpu()), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('rows,C', [(1, 2), (50, 3), (1000, 7), (5000, 40), (777, 47), (300, 128), (4099, 48), (513, 8), (2000, 64), (70000, 4)])
def test_nll_logsoftmax_fused(rows, C):
    from gnn_tail_generalization_amd import ops
    z = (3 * _rand(rows, C, seed=rows)).requires_grad_(True)
    y = torch.randint(0, C, (rows,), generator=torch.Generator().manual_seed(C))
    mask = torch.rand(rows, generator=torch.Generator().manual_seed(1)) < 0.4
    mask[0] = True
    ref = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(z.double()[mask], 1), y[mask])
    ref.backward()
    zd = z.detach().to(DEV).requires_grad_(True)
    loss = ops.nll_logsoftmax(zd, y.to(DEV), mask.to(DEV))
    (2.0 * loss).backward()
    torch.testing.assert_close(loss.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(zd.grad.cpu().double(), 2.0 * z.grad.double(), rtol=1e-4, atol=1e-7)
    loss_all = ops.nll_logsoftmax(zd.detach(), y.to(DEV))
    ref_all = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(z.detach().double(), 1), y)
    torch.testing.assert_close(loss_all.cpu().double(), ref_all, rtol=1e-5, atol=1e-6)


def test_fused_adam_matches_oracle_and_torch():
    from gnn_tail_generalization_amd.optim import Adam
    shapes = [(5,), (33, 7), (256, 256), (1001,)]
    ps = [_rand(*s, seed=i) for i, s in enumerate(shapes)]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    orc_p = {str(i): p.clone() for i, p in enumerate(ps)}
    opt_m = Adam(mine, lr=0.01, weight_decay=5e-4)
    opt_r = torch.optim.Adam(ref, lr=0.01, weight_decay=5e-4)
    state = {}
    for step in range(1, 6):
        grads = [_rand(*s, seed=100 * step + i) for i, s in enumerate(shapes)]
        for p, g in zip(mine, grads):
            p.grad = g.to(DEV)
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        opt_m.step()
        opt_r.step()
        orc.adam_step(orc_p, {str(i): g for i, g in enumerate(grads)}, state, 0.01, 5e-4, step)
    for i, (m, r) in enumerate(zip(mine, ref)):
        torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m.detach().cpu(), orc_p[str(i)], rtol=1e-5, atol=1e-6)


def test_fused_adam_extra_decay_is_the_frobenius_regulariser_gradient():
    """optim.Adam.extra_decay_buffer(p) <- c / ||p||_F : the fused update equals torch.optim.Adam on the gradient g + c * p / ||p||_F
    (what autograd produces for the loss term c * th.norm(p), GCN.py:232 / trainer_node_classification.py:393); tensors without a
    buffer are untouched by it."""
    from gnn_tail_generalization_amd.optim import Adam
    shapes = [(300, 64), (77,), (1000, 256)]
    ps = [_rand(*s, seed=40 + i) for i, s in enumerate(shapes)]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt_m = Adam(mine, lr=0.01, weight_decay=5e-4)
    opt_r = torch.optim.Adam(ref, lr=0.01, weight_decay=5e-4)
    c = 0.3
    for step in range(1, 5):
        grads = [_rand(*s, seed=900 * step + i) for i, s in enumerate(shapes)]
        for i, (p, g) in enumerate(zip(mine, grads)):
            p.grad = g.to(DEV)
            if i != 1:
                opt_m.extra_decay_buffer(p).fill_(c / float(p.detach().norm()))
        for i, (p, g) in enumerate(zip(ref, grads)):
            p.grad = g.clone() if i == 1 else g + c * p.detach() / p.detach().norm()
        opt_m.step()
        opt_r.step()
    for m, r in zip(mine, ref):
        torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)


def test_fused_adam_leaves_the_norm_of_what_it_wrote():
    """A parameter whose Frobenius norm a forward has asked for (ops.frobenius_norm on a leaf Parameter: GCN.py:232's th.norm(self.le)) gets the
    norm of its UPDATED values from the Adam kernel itself (cb_adam_multi_norm_f32); the next ops.frobenius_norm reads it instead of the table:
    the same bits as the stand-alone reduction for the largest tensor of the launch, the same update as without it, and it is not
    trusted once torch has written the parameter."""
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd.optim import Adam
    shapes = [(3000, 256), (77,), (3000, 256), (513, 256)]
    mk = lambda: [torch.nn.Parameter(_rand(*s, seed=70 + i).to(DEV)) for i, s in enumerate(shapes)]      # noqa: E731
    with_norm, without = mk(), mk()
    opt_a, opt_b = Adam(with_norm, lr=0.01, weight_decay=5e-4), Adam(without, lr=0.01, weight_decay=5e-4)
    for p in (with_norm[0], with_norm[2], with_norm[3]):
        first = ops.frobenius_norm(p)                                   # computed: nothing known yet
        torch.testing.assert_close(first, p.detach().norm(), rtol=1e-6, atol=0)
    calls = []
    lib_norm = ops._lib.load().cb_frobenius_norm_f32
    for step in range(1, 4):
        for i, s in enumerate(shapes):
            g = _rand(*s, seed=500 * step + i).to(DEV)
            with_norm[i].grad, without[i].grad = g, g.clone()
        opt_a.step()
        opt_b.step()
        for a, b in zip(with_norm, without):
            assert torch.equal(a.detach(), b.detach())                   # the update does not depend on the by-product
        assert getattr(with_norm[1], '_cb_norm', None) is None          # never asked for
        for i in (0, 2, 3):
            p = with_norm[i]
            known = ops.known_norm(p)
            assert known is not None
            alone = torch.empty(2, device=DEV)
            ws = ops._ws(ops._lib.load().cb_reduce_workspace_bytes(), p.device)
            ops._lib.check(lib_norm(ops._lib.ptr(p.detach()), p.numel(), ops._lib.ptr(alone), ops._lib.ptr(ws), ws.numel(), ops._lib.stream_ptr()), 'norm')
            if i != 3:                                                   # largest tensors of the launch: k_sumsq's own thread map, bit for bit
                assert torch.equal(known, alone)
            torch.testing.assert_close(known, alone, rtol=1e-6, atol=0)
            assert torch.equal(ops.frobenius_norm(p).detach(), known[0])
    p = with_norm[0]
    with torch.no_grad():
        p.mul_(2.0)                                                      # torch wrote it: the known norm is of other values
    assert ops.known_norm(p) is None
    torch.testing.assert_close(ops.frobenius_norm(p).detach(), p.detach().norm(), rtol=1e-6, atol=0)
    # autograd through the known norm: d||p||/dp = p / ||p||
    q = with_norm[2]
    assert ops.known_norm(q) is not None
    q.grad = None
    ops.frobenius_norm(q).backward()
    torch.testing.assert_close(q.grad, q.detach() / q.detach().norm(), rtol=1e-5, atol=1e-7)
    ops.forget_norms([q])
    assert ops.known_norm(q) is None


def test_gather_rows_by_index():
    from gnn_tail_generalization_amd import ops
    for shape in [(100, 256), (57, 40), (9, 7)]:
        x = _rand(*shape, seed=1).to(DEV)
        idx = torch.randint(0, shape[0], (333,), generator=torch.Generator().manual_seed(2)).to(DEV)
        assert torch.equal(ops.gather_rows_by_index(x, idx), x[idx])
    assert ops.gather_rows_by_index(x, idx[:0]).shape == (0, 7)


def test_bf16_storage_kernels():
    """bf16-stored aggregation rows (build extension): GEMM writes RNE-rounded bf16, SpMM widens and accumulates in fp32."""
    from gnn_tail_generalization_amd import gemm
    from gnn_tail_generalization_amd.graph import CSRGraph
    from conftest import load_golden
    a, b = _rand(300, 64, seed=1), _rand(64, 256, seed=2)
    rs, add = torch.rand(300, generator=torch.Generator().manual_seed(5)) + 0.5, _rand(300, 256, seed=3)
    z16 = gemm.mm_nn(a.to(DEV), b.to(DEV), rowscale=rs.to(DEV), addend=add.to(DEV), out_bf16=True)
    assert z16.dtype == torch.bfloat16
    ref = ((a.double() @ b.double()) * rs.double().unsqueeze(1) + add.double())
    # within one bf16 ulp of the exactly rounded value (fp32 summation order may flip a tie)
    err = (z16.cpu().double() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 1e-6).all()
    exact = ref.float().to(torch.bfloat16)
    assert (z16.cpu() == exact).float().mean() > 0.98
    g = load_golden('case_graph_powerlaw_d7_d64')
    n = g['cfg']['N_nodes']
    for T in (256, 4):
        G = CSRGraph(g['edge_index'].to(DEV), n, hub_threshold=T)
        for d in (256, 128, 40):
            h16 = _rand(n, d, seed=d).to(torch.bfloat16)
            csr = orc.build_csr(g['edge_index'], n)
            want = orc.aggregate_sum_dense_f64(csr, h16.double())
            got = G.spmm(h16.to(DEV))
            torch.testing.assert_close(got.cpu().double(), want, atol=1e-4, rtol=1e-5)


def test_se_topk_replace_matches_reference_and_oracle():
    """§8f row 2: fused scores + running top-K + softmax-combine (cb_topk_replace_f32) vs the reference fixture and,
    on larger ragged shapes, vs the oracle's per-node restatement."""
    from gnn_tail_generalization_amd import ops
    from conftest import load_golden
    fx = load_golden('semlp_fixture')
    for name, g in fx.items():
        out = ops.se_topk_replace(g['q'].to(DEV), g['teacher'].to(DEV), g['k'])
        torch.testing.assert_close(out.cpu(), g['out'], atol=2e-5, rtol=2e-5, msg=lambda m, name=name: f'{name}: {m}')
    for b, n, d, k, seed in [(1, 5, 3, 1, 0), (130, 1000, 71, 2, 1), (257, 3333, 256, 3, 2), (64, 129, 768, 8, 3), (500, 128, 16, 4, 4)]:
        gen = torch.Generator().manual_seed(seed)
        q, t = torch.randn(b, d, generator=gen), torch.randn(n, d, generator=gen)
        want, sel, w = orc.semlp_replacement(q, t, k)
        out, idx, wgt = ops.se_topk_replace(q.to(DEV), t.to(DEV), k, return_selection=True)
        assert torch.equal(idx.cpu().to(torch.int64), sel), (b, n, d, k)
        torch.testing.assert_close(wgt.cpu(), w, atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(out.cpu(), want, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('M,K,N', [(1000, 256, 256), (777, 36, 132), (130, 20, 40), (4099, 128, 64), (65, 8, 4), (300, 260, 516)])
def test_gemm_three_limb_error_is_fp32_level(M, K, N):
    """The default GEMM path decomposes fp32 operands into three bf16 limbs (csrc/cb_gemm_limb.hip).  Its error against an
    fp64 product, measured in units of sum|a||b| per output, must stay at the level of a true fp32 GEMM (torch.matmul),
    on ragged shapes (partial K steps, tile overhang in M and N) and on operands spanning ~12 decades."""
    from gnn_tail_generalization_amd import gemm
    g = torch.Generator().manual_seed(M + K + N)
    for wide in (False, True):
        a = torch.randn(M, K, generator=g)
        b = torch.randn(K, N, generator=g)
        gr = torch.randn(M, N, generator=g)
        if wide:
            a = a * torch.exp(3 * torch.randn(M, K, generator=g))
            b = b * torch.exp(3 * torch.randn(K, N, generator=g))
        ad, bd, gd = a.to(DEV), b.to(DEV), gr.to(DEV)
        ref = a.double() @ b.double()
        scale = a.double().abs() @ b.double().abs() + 1e-300
        err = ((gemm.mm_nn(ad, bd).cpu().double() - ref).abs() / scale).max().item()
        err32 = (((ad @ bd).cpu().double() - ref).abs() / scale).max().item()
        assert err <= max(2 * err32, 8 * 2.0 ** -24), ('nn', wide, err / 2.0 ** -24, err32 / 2.0 ** -24)
        ref = a.double().t() @ gr.double()
        scale = a.double().abs().t() @ gr.double().abs() + 1e-300
        err = ((gemm.mm_tn(ad, gd).cpu().double() - ref).abs() / scale).max().item()
        err32 = (((ad.t() @ gd).cpu().double() - ref).abs() / scale).max().item()
        assert err <= max(2 * err32, 8 * 2.0 ** -24), ('tn', wide, err / 2.0 ** -24, err32 / 2.0 ** -24)


def test_gemm_three_limb_split_is_exact_on_special_values():
    """hi + mid + lo reproduces the operand exactly: products with one-hot operands return the other operand bit for bit
    (zeros and huge values included).  Only where a limb falls below the smallest normal fp32 (|a| < ~2^-110) is it flushed:
    the absolute error then stays below 2^-126."""
    from gnn_tail_generalization_amd import gemm
    vals = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.14159274, 1e-30, -7.5e-20, 16777215.0, 1.0000001, 3.4e37, -2.9e-37, 0.33333334])
    K = 64
    a = vals.repeat(8)[:K].repeat(40, 1).contiguous()       # [40, 64], row = the value list
    eye = torch.eye(K)
    big = a.abs() >= 1e-30
    out = gemm.mm_nn(a.to(DEV), eye.to(DEV)).cpu()
    assert torch.equal(out[big], a[big]) and (out - a).abs().max().item() <= 2.0 ** -126
    at = a.t().contiguous()[:, :40].contiguous()
    out_t = gemm.mm_tn(eye.to(DEV), at.to(DEV)).cpu()       # I^T @ A^T
    assert torch.equal(out_t[big.t()[:, :40]], at[big.t()[:, :40]]) and (out_t - at).abs().max().item() <= 2.0 ** -126


def test_gemm_full_size_properties():
    """BASELINE full size (10^7 rows, 256 x 256 weights): size-independent properties of the dense contractions.
    NN with the identity returns the operand bit for bit (the three-limb split is exact and every other limb product is an
    exact zero), on every row block incl. the ragged last one; TN against a one-hot row selector returns the selected rows;
    TN of ones gives exact column sums of small integers; both are deterministic across launches."""
    from gnn_tail_generalization_amd import gemm
    M, D = 10_000_003, 256
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.randn(M, D, device=DEV, generator=g)
    eye = torch.eye(D, device=DEV)
    out = gemm.mm_nn(a, eye)
    assert torch.equal(out, a)
    del out
    # TN: sel^T @ a with sel[r_j, j] = 1 picks rows r_j of a exactly
    rows = torch.tensor([0, 1, 127, 128, 5_000_000, M - 2, M - 1, 777_777], device=DEV)
    sel = torch.zeros(M, 8, device=DEV)
    sel[rows, torch.arange(8, device=DEV)] = 1.0
    picked = gemm.mm_tn(sel, a)
    assert torch.equal(picked, a[rows])
    # exact integer column sums (all partial sums < 2^24) and launch-to-launch determinism
    ints = torch.randint(0, 2, (M, D), device=DEV, generator=g).float()
    ones = torch.ones(M, 4, device=DEV)
    s1 = gemm.mm_tn(ones, ints)
    assert torch.equal(s1[0].double(), ints.sum(0, dtype=torch.float64)) and torch.equal(s1[0], s1[3])
    w = torch.randn(D, D, device=DEV, generator=g)
    z1, z2 = gemm.mm_nn(a, w), gemm.mm_nn(a, w)
    assert torch.equal(z1, z2)
    d1, d2 = gemm.mm_tn(a, z1), gemm.mm_tn(a, z1)
    assert torch.equal(d1, d2)


def test_gemm_presplit_lds_dma_variant_matches_fp64():
    """CB_LIMB_PRESPLIT=1: the weight operand split once per launch (k_presplit_cols) and staged by LDS-DMA
    (global_load_lds_dwordx4) — off by default (measured neutral), kept correct: same fp32-level error as the default path,
    ragged K / N / M, with and without the epilogue terms, incl. the dual-output (dropout copy) epilogue."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from gnn_tail_generalization_amd import gemm, ops
torch.manual_seed(0)
dev = 'cuda:0'
for M, K, N in [(4096, 256, 256), (5000, 128, 256), (3001, 40, 256), (2500, 100, 132), (70000, 256, 260)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    rs = torch.rand(M, device=dev) + 0.5; add = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
    got = gemm.mm_nn(a, b, rowscale=rs, addend=add, bias=bias, relu=True).double().cpu()
    ref = torch.relu((a.double().cpu() @ b.double().cpu()) * rs.double().cpu().unsqueeze(1) + add.double().cpu() + bias.double().cpu())
    scale = float((a.double().abs().cpu() @ b.double().abs().cpu()).max()) + 10
    err = float((got - ref).abs().max())
    assert err <= 3e-6 * scale, (M, K, N, err, scale)
y, yd = gemm.mm_nn_drop2(torch.randn(40000, 128, device=dev), torch.randn(128, 256, device=dev), 0.3, 4242, row0=17, bias=None, relu=True)
keep = ops.dropout_keep_mask((40000, 256), 0.3, 4242, dev, offset=17 * 256)
torch.testing.assert_close(yd, torch.where(keep, y / 0.7, torch.zeros_like(y)), atol=1e-6, rtol=1e-6)
print('PRESPLIT_OK')
''' % root
    env = dict(os.environ, CB_LIMB_PRESPLIT='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert 'PRESPLIT_OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize('M,K,p', [(40001, 128, 0.3), (33000, 1000, 0.0), (5000, 128, 0.5)])
def test_gemm_dropout_copy_epilogue_matches_two_kernels(M, K, p):
    """cb_gemm_nn_drop2_f32: y = relu(a @ b + bias) and dropout_p(y) from one epilogue (>= 256 wide tiles: the dual-output kernel;
    below that the entry point itself runs the two kernels) == cb_gemm_nn_f32 followed by cb_dropout_f32 with the same seed / offset."""
    from gnn_tail_generalization_amd import gemm, ops
    torch.manual_seed(M)
    a, b, bias = torch.randn(M, K, device=DEV), torch.randn(K, 256, device=DEV), torch.randn(256, device=DEV)
    y, yd = gemm.mm_nn_drop2(a, b, p, 777, row0=31, bias=bias, relu=True)
    y_ref = gemm.mm_nn(a, b, bias=bias, relu=True)
    assert torch.equal(y, y_ref)
    if p > 0:
        keep = ops.dropout_keep_mask((M, 256), p, 777, DEV, offset=31 * 256)
        torch.testing.assert_close(yd, torch.where(keep, y / (1 - p), torch.zeros_like(y)), atol=1e-6, rtol=1e-6)
    else:
        assert torch.equal(yd, y)


@pytest.mark.parametrize('M,K,N,row0', [(40000, 128, 256, 0), (33001, 100, 256, 77)])
def test_operand_dropout_gemms_equal_dropout_then_gemm(M, K, N, row0):
    """the dropout of the input features applied by the input Linear's GEMM while it stages x
    (cb_gemm_nn_indrop_drop2_f32) and regenerated by the weight-gradient GEMM (cb_gemm_tn_gdrop_f32) — both bit-identical to
    cb_dropout_f32 followed by the plain GEMMs (same keep-mask: a pure function of seed and flat index, also for a row-sharded x)."""
    from gnn_tail_generalization_amd import gemm, ops
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(M, K, device=DEV, generator=gen)
    w = torch.randn(K, N, device=DEV, generator=gen) * 0.1
    b = torch.randn(N, device=DEV, generator=gen)
    g = torch.randn(M, N, device=DEV, generator=gen)
    p, s_in, s_out = 0.1, 0x1234ABCD5, 0x77
    fused = gemm.mm_nn_indrop_drop2(x, w, p, s_in, s_out, row0, bias=b, relu=True)
    assert fused is not None
    xd = ops._dropout_raw(x, p, s_in, row0 * K)
    y, yd = gemm.mm_nn_drop2(xd, w, p, s_out, row0, bias=b, relu=True)
    assert torch.equal(fused[0], y) and torch.equal(fused[1], yd)
    dw = gemm.mm_tn_gdrop(g, x, p, s_in, row0)
    assert dw is not None and torch.equal(dw, gemm.mm_tn(g, xd))
    assert gemm.mm_nn_indrop_drop2(x[:100], w, p, s_in, s_out, 0, bias=b, relu=True) is None       # too few tiles: no fused form, caller falls back
    # the mask words of (y > 0) from the same epilogue (N == 256), in the layout of the aggregation's fused store; the input stage of the trunk
    # backward gives the same result from them as from y itself
    if N == 256:
        from gnn_tail_generalization_amd import trunk
        y3, yd3, bits = gemm.mm_nn_indrop_drop2(x, w, p, s_in, s_out, row0, bias=b, relu=True, want_bits=True)
        assert torch.equal(y3, y) and torch.equal(yd3, yd)
        pos = y > 0
        for kk in range(4):
            ref_w = (pos[:, kk::4].to(torch.int64) << torch.arange(64, device=DEV, dtype=torch.int64)).sum(dim=1)
            assert torch.equal(bits[:, 0, kk], ref_w), kk
        g1 = torch.randn(M, N, device=DEV, generator=gen)
        o_a, c_a = trunk._input_bwd_multi(g, 5, [g1], [6], 0.1, y, p, row0)
        o_b, c_b = trunk._input_bwd_multi(g, 5, [g1], [6], 0.1, y, p, row0, act_bits=bits)
        assert torch.equal(o_a, o_b) and torch.equal(c_a, c_b)
        # round 4: the single-output form (no dropped copy of the OUTPUT either) + mask words
        y4, bits4 = gemm.mm_nn_indrop(x, w, p, s_in, row0, bias=b, relu=True, want_bits=True)
        assert torch.equal(y4, y) and torch.equal(bits4, bits)
        # ... and the next GEMM that applies the dropout of y while IT stages y (layer 0's transform: row scale + addend epilogue),
        # with the weight gradient that regenerates that mask from y
        w2 = torch.randn(N, 256, device=DEV, generator=gen) * 0.1
        rs = torch.rand(M, device=DEV, generator=gen) + 0.5
        le = torch.randn(M, 256, device=DEV, generator=gen)
        z = gemm.mm_nn_indrop(y, w2, p, s_out, row0, rowscale=rs, addend=le)
        assert z is not None and torch.equal(z, gemm.mm_nn(yd, w2, rowscale=rs, addend=le))
        gz = torch.randn(M, 256, device=DEV, generator=gen)
        dw2 = gemm.mm_tn_adrop(y, gz, p, s_out, row0, rowscale=rs)
        assert dw2 is not None and torch.equal(dw2, gemm.mm_tn(yd, gz, rowscale=rs))
        dw3 = gemm.mm_tn_adrop(y, gz, p, s_out, row0)
        assert torch.equal(dw3, gemm.mm_tn(yd, gz))
        assert gemm.mm_nn_indrop(y[:100], w2, p, s_out, 0, rowscale=rs[:100]) is None                # too few tiles: the caller materialises the copy


@pytest.mark.parametrize('M,K,p,row0', [(40000, 128, 0.1, 0), (33001, 128, 0.3, 77), (513, 64, 0.2, 5), (70, 128, 0.0, 0), (100003, 64, 0.0, 0)])
def test_forward_front_kernel_equals_the_two_gemms(M, K, p, row0):
    """cb_trunk_front_f32 — dropout(x), input Linear, ReLU, dropout(X0) and layer 0's transform in one kernel, the 64-row
    tile of dropout(X0) never leaving the chip — is BIT-identical to dropout + cb_gemm_nn_f32 (bias, ReLU) + dropout + cb_gemm_nn_f32 (row
    scale, addend): X0, its mask words, the optional dropped copy and Z0; ragged last tile, fewer tiles than persistent blocks, row-sharded
    Philox offsets, p = 0 (eval forwards)."""
    from gnn_tail_generalization_amd import gemm, ops
    gen = torch.Generator(device=DEV).manual_seed(M)
    x = torch.rand(M, K, device=DEV, generator=gen)
    w_in = torch.randn(256, K, device=DEV, generator=gen) * 0.1
    b_in = torch.randn(256, device=DEV, generator=gen) * 0.1
    w0 = torch.randn(256, 256, device=DEV, generator=gen) * 0.07
    a = torch.rand(M, device=DEV, generator=gen) + 0.5
    le = torch.randn(M, 256, device=DEV, generator=gen)
    sx, s0 = 0x1234ABCD5, 0x77
    xd = ops._dropout_raw(x, p, sx, row0 * K) if p > 0 else x
    x0_ref = gemm.mm_nn(xd, w_in.t().contiguous(), bias=b_in, relu=True)
    x0d_ref = ops._dropout_raw(x0_ref, p, s0, row0 * 256) if p > 0 else x0_ref
    for addend, rs, want_drop in ((le, a, True), (None, None, False)):
        fr = gemm.trunk_front(x, w_in, b_in, w0, rs, addend, p, sx, s0, row0, want_bits=True, want_drop=want_drop)
        assert fr is not None
        x0, bits, x0d, z0 = fr
        assert torch.equal(x0, x0_ref)
        assert torch.equal(z0, gemm.mm_nn(x0d_ref, w0, rowscale=rs, addend=addend))
        assert (x0d is None) == (not want_drop) and (x0d is None or torch.equal(x0d, x0d_ref))
        pos = x0_ref > 0
        for kk in range(4):
            ref_w = (pos[:, kk::4].to(torch.int64) << torch.arange(64, device=DEV, dtype=torch.int64)).sum(dim=1)
            assert torch.equal(bits[:, 0, kk], ref_w), kk
    assert gemm.trunk_front(x[:, :60].contiguous(), w_in[:, :60].contiguous(), b_in, w0, a, le, p, sx, s0) is None      # no kernel for this input width


@pytest.mark.parametrize('M,p,relu_only,with_mix,with_index,want_act', [(40_000, 0.1, False, True, False, False), (3_001, 0.3, True, True, True, True),
                                                                        (70_000, 0.0, False, False, False, False), (257, 0.1, False, True, False, True)])
def test_gemm_with_the_trunk_store_on_a_subset_of_rows_equals_the_two_kernels(M, p, relu_only, with_mix, with_index, want_act):
    """cb_gemm_nn_store_rows_f32 (the rows-only forward's sum-first layer: transform + ReLU / mask words / mix / dropout on compact rows in one kernel) against
    cb_gemm_nn_f32 followed by cb_trunk_store_rows_f32: stored rows, ReLU output and the mask words of the written node rows, bit for bit."""
    from gnn_tail_generalization_amd import gemm, trunk
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(M)
    n_nodes = 3 * M + 7
    idx = torch.sort(torch.randperm(n_nodes, generator=g)[:M]).values.to(dev)
    a = (torch.rand(M, 256, generator=g) - 0.5).to(dev)
    w = ((torch.rand(256, 256, generator=g) - 0.5) * 0.2).to(dev)
    rs = (torch.rand(M, generator=g) + 0.5).to(dev)
    addend = (torch.rand(M, 256, generator=g) - 0.5).to(dev)
    bias = (torch.rand(256, generator=g) - 0.5).to(dev)
    mix = mix_index = None
    if with_mix:
        n_mix = 2 * M if with_index else n_nodes
        mix = (torch.rand(n_mix, 256, generator=g) - 0.5).to(dev)
        mix_index = torch.randint(0, n_mix, (M,), generator=g).to(dev) if with_index else None
    bits_f = torch.zeros((n_nodes, 1, 4), dtype=torch.int64, device=dev)
    bits_t = torch.zeros_like(bits_f)
    fused = gemm.mm_nn_store_rows(a, w, rs, addend, bias, idx, mix, mix_index, 0.9, 0.1, p, 4242, 5, bits_f, relu_only, want_act)
    assert fused is not None
    y = gemm.mm_nn(a, w, rowscale=rs, addend=addend, bias=bias)
    out_t, act_t = trunk._store_rows(y, idx, mix, 0.9, 0.1, p, 4242, 5, bits_t, relu_only, mix_index, want_act)
    assert torch.equal(fused[0], out_t) and torch.equal(bits_f, bits_t)
    assert (fused[1] is None) == (not want_act) and (not want_act or torch.equal(fused[1], act_t))
    assert int((bits_f[idx] != 0).sum()) > 0


@pytest.mark.parametrize('p', [0.0, 0.2])
def test_reverse_aggregation_with_the_store_backward_in_its_epilogue(p):
    """cb_spmm_csr_store_bwd_f32 (+ the second column sum of cb_trunk_input_bwd_multi_cs_f32) against cb_spmm_csr_f32 followed by cb_trunk_layer_bwd_f32:
    the raw gradient, the masked / scaled gradient and the bias gradient, bit for bit — on a power-law graph with hub rows."""
    from gnn_tail_generalization_amd import trunk
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    dev = 'cuda:0'
    data = synthetic_data('S-pl1M', seed=0, device=dev, n_override=30_000)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    assert G._plan.n_hubs > 0
    n = G.N
    gen = torch.Generator(device='cpu').manual_seed(7)
    h = (torch.rand(n, 256, generator=gen) - 0.5).to(dev)
    bits = torch.randint(-2 ** 62, 2 ** 62, (n, 1, 4), generator=gen, dtype=torch.int64).to(dev)
    seed, row0, c_act = 991, 3, 0.9
    g_ref = G.spmm(h, row_scale=G.norm_out)
    gr_ref, db_ref = trunk._layer_bwd(g_ref, bits, G.norm_in, None, False, p, seed, row0, c_act, 0.1, True)
    g, gr = G.spmm_store_bwd(h, G.norm_out, bits, G.norm_in, c_act, p, seed, row0)
    assert torch.equal(g, g_ref) and torch.equal(gr, gr_ref)
    # the bias gradient from the input stage's pass over g
    main = (torch.rand(n, 256, generator=gen) - 0.5).to(dev)
    x0_bits = torch.randint(-2 ** 62, 2 ** 62, (n, 1, 4), generator=gen, dtype=torch.int64).to(dev)
    other = (torch.rand(n, 256, generator=gen) - 0.5).to(dev)
    plain = trunk._input_bwd_multi(main, 5, [other, g], [17, seed], 0.1, None, p, row0, act_bits=x0_bits)
    with_cs = trunk._input_bwd_multi(main, 5, [other, g], [17, seed], 0.1, None, p, row0, act_bits=x0_bits, cs=[(1, bits, c_act)])
    assert torch.equal(with_cs[0], plain[0]) and torch.equal(with_cs[1], plain[1]) and torch.equal(with_cs[2][0], db_ref)
