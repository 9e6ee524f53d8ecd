"""GPU parity (through the C ABI): edge_index -> CSR bit-exact vs the oracle, degree norms, and
the sum-aggregation kernel vs the oracle on the same seeded inputs.  fp32 tolerance 1e-5 rel
(north_star: logits within 1e-4)."""
import numpy as np
import pytest
import torch

import coldbrew_oracle as orc
from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _graph(ei, n=None, T=256):
    from gnn_tail_generalization_amd.graph import CSRGraph
    return CSRGraph(ei.to(DEV), n, hub_threshold=T)


GRAPH_CASES = ['case_graph_asym_multi', 'case_graph_example', 'case_graph_powerlaw_d7_d64', 'case_nr_se000_L2',
               'case_r_initialbn_se111_L3_powerlaw', 'case_graph_zero_in_degree']


@pytest.mark.parametrize('name', GRAPH_CASES)
def test_csr_bit_exact(name):
    g = load_golden(name)
    n = g['cfg']['N_nodes']
    csr = orc.build_csr(g['edge_index'], n)
    G = _graph(g['edge_index'], n)
    assert G.N == csr.N and G.E == csr.E
    assert np.array_equal(G.rowptr.cpu().numpy().astype(np.int64), csr.rowptr)
    assert np.array_equal(G.col.cpu().numpy()[:csr.E], csr.col)
    assert np.array_equal(G.rowptr_t.cpu().numpy().astype(np.int64), csr.rowptr_t)
    assert np.array_equal(G.col_t.cpu().numpy()[:csr.E], csr.col_t)
    assert G.n_zero_in_degree == int((csr.in_deg == 0).sum())
    assert G.max_in_degree == int(csr.in_deg.max())
    sym = np.array_equal(csr.rowptr, csr.rowptr_t) and np.array_equal(csr.col, csr.col_t)
    assert G.symmetric == sym
    a, b = orc.degree_norms(csr)
    np.testing.assert_allclose(G.norm_out.cpu().numpy(), a.numpy(), rtol=1.2e-7)
    np.testing.assert_allclose(G.norm_in.cpu().numpy(), b.numpy(), rtol=1.2e-7)
    assert torch.equal(G.in_degrees().cpu(), torch.from_numpy(csr.in_deg))
    assert torch.equal(G.out_degrees().cpu(), torch.from_numpy(csr.out_deg))


def test_edge_order_invariance_and_noncontiguous_view():
    g = load_golden('case_graph_asym_multi')
    ei = g['edge_index']
    G1 = _graph(ei)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    G2 = _graph(ei[:, perm].t().contiguous().t())   # permuted + transposed (non-contiguous) view, cf. utils.py:745
    assert torch.equal(G1.rowptr, G2.rowptr) and torch.equal(G1.col, G2.col)
    assert torch.equal(G1.rowptr_t, G2.rowptr_t) and torch.equal(G1.col_t, G2.col_t)


def test_out_of_range_and_empty():
    from gnn_tail_generalization_amd.graph import CSRGraph
    with pytest.raises(ValueError):
        CSRGraph(torch.tensor([[0, 5], [1, 2]], device=DEV), 4)
    G = CSRGraph(torch.zeros((2, 0), dtype=torch.int64, device=DEV), 5)
    assert G.E == 0 and G.n_zero_in_degree == 5 and G.rowptr.tolist() == [0] * 6
    out = G.spmm(torch.randn(5, 8, device=DEV), bias=torch.ones(8, device=DEV))
    assert torch.equal(out.cpu(), torch.ones(5, 8))


def test_zero_in_degree_raises():
    from gnn_tail_generalization_amd.graph import ZeroInDegreeError
    g = load_golden('case_graph_zero_in_degree')
    G = _graph(g['edge_index'], g['cfg']['N_nodes'])
    with pytest.raises(ZeroInDegreeError):
        G.check_zero_in_degree()


@pytest.mark.parametrize('name', ['case_graph_asym_multi', 'case_graph_powerlaw_d7_d64', 'case_graph_example'])
@pytest.mark.parametrize('d', [1, 3, 7, 40, 64, 100, 128, 130, 256, 260, 512])
@pytest.mark.parametrize('T', [256, 4])
def test_spmm_vs_oracle(name, d, T):
    g = load_golden(name)
    n = g['cfg']['N_nodes']
    csr = orc.build_csr(g['edge_index'], n)
    G = _graph(g['edge_index'], n, T=T)
    gen = torch.Generator().manual_seed(d)
    h = torch.randn(n, d, generator=gen)
    bias = torch.randn(d, generator=gen)
    a, b = orc.degree_norms(csr)
    ref = orc.aggregate_sum_dense_f64(csr, h)
    # plain sum, forward orientation
    out = G.spmm(h.to(DEV))
    torch.testing.assert_close(out.cpu().double(), ref, atol=1e-5, rtol=1e-5)
    # fused epilogue: * norm_in + bias, relu
    out = G.spmm(h.to(DEV), row_scale=G.norm_in, bias=bias.to(DEV), relu=True)
    ref2 = torch.relu(ref * b.double().unsqueeze(1) + bias.double())
    torch.testing.assert_close(out.cpu().double(), ref2, atol=1e-5, rtol=1e-5)
    # reverse orientation (the backward of the aggregation): A.h
    A = np.zeros((n, n))
    np.add.at(A, (csr.src, csr.dst), 1.0)
    out_t = G.spmm(h.to(DEV), transpose=True)
    torch.testing.assert_close(out_t.cpu().double(), torch.from_numpy(A @ h.double().numpy()), atol=1e-5, rtol=1e-5)


def test_spmm_strided_rows_and_determinism():
    g = load_golden('case_graph_powerlaw_d7_d64')
    n = g['cfg']['N_nodes']
    G = _graph(g['edge_index'], n, T=8)
    big = torch.randn(n, 300, device=DEV)
    h = big[:, 4:260]                     # ld = 300, 16-byte aligned start, d = 256
    o1 = G.spmm(h)
    o2 = G.spmm(h.contiguous())
    assert torch.equal(o1, o2)
    for _ in range(3):
        assert torch.equal(G.spmm(h), o1)  # fixed reduction order: bitwise reproducible


def test_spmm_linearity_large():
    """Size-independent property at a scale the dense oracle cannot reach: A^T(x + 2y) = A^T x + 2 A^T y,
    row sums of A^T.1 equal the in-degrees (checksum of checksums)."""
    from gnn_tail_generalization_amd.data import synthetic_data
    d = synthetic_data('S-pl1M', seed=1, device=DEV, n_override=200000)
    G = _graph(d.edge_index, 200000)
    x = torch.randn(200000, 256, device=DEV)
    y = torch.randn(200000, 256, device=DEV)
    lhs = G.spmm(x + 2 * y)
    rhs = G.spmm(x) + 2 * G.spmm(y)
    torch.testing.assert_close(lhs, rhs, atol=2e-3, rtol=1e-4)
    ones = torch.ones(200000, 4, device=DEV)
    deg = G.spmm(ones)[:, 0]
    assert torch.equal(deg.to(torch.int64), G.in_degrees())
    assert G._plan.n_hubs > 0


def test_full_size_properties_s_pl10m():
    """BASELINE full size (10^7 nodes, 10^8 edge_index columns, d = 256): size-independent properties of the
    ingest + aggregation that need no dense oracle."""
    from gnn_tail_generalization_amd.data import synthetic_data
    data = synthetic_data('S-pl10M', seed=0, device=DEV)
    n, E = data.x.shape[0], data.edge_index.shape[1]
    assert (n, E) == (10_000_000, 100_000_000)
    G = _graph(data.edge_index, n)
    ei = data.edge_index
    # CSR == edge multiset: row lengths are the in-degrees, the (row, col) pairs re-sorted equal the sorted edge keys
    assert torch.equal(G.in_degrees(), torch.bincount(ei[1], minlength=n))
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), G.in_degrees())
    key_csr = rows * n + G.col.to(torch.int64)
    assert bool((key_csr[1:] >= key_csr[:-1]).all())                          # sorted by (dst, src)
    assert torch.equal(key_csr, torch.sort(ei[1] * n + ei[0])[0])             # same multiset, bit-exact
    del rows, key_csr
    assert G.symmetric and G.n_zero_in_degree == 0 and G._plan.n_hubs > 0
    # aggregation of ones = in-degree (exact in fp32: max degree < 2^24), through main + hub kernels
    deg = G.spmm(torch.ones(n, 256, device=DEV))
    assert torch.equal(deg[:, 0].to(torch.int64), G.in_degrees()) and torch.equal(deg[:, 255], deg[:, 0])
    del deg
    # column-sum identity: sum_v out[v] = sum_u outdeg(u) * h[u]   (checksum of checksums, fp64)
    h = torch.rand(n, 256, device=DEV)
    out = G.spmm(h)
    lhs = out.sum(dim=0, dtype=torch.float64)
    rhs = (h.double() * G.out_degrees().double().unsqueeze(1)).sum(dim=0)
    torch.testing.assert_close(lhs, rhs, rtol=1e-6, atol=0)
    # run-to-run bit reproducibility (no atomics)
    assert torch.equal(out, G.spmm(h))


@pytest.mark.parametrize('d', [2, 4, 5, 8, 10, 12, 16, 18, 20, 31, 32, 36, 47, 60])
@pytest.mark.parametrize('T', [256, 4])
def test_spmm_narrow_widths_every_group_shape(d, T):
    """The narrow-feature kernel (cb_spmm_small.hip: a wavefront cut into groups of 4 / 8 / 16 / 32 lanes, scalar or
    float4 lanes) on every group shape, contiguous and strided rows, hub rows present (T = 4) — vs dense fp64."""
    g = load_golden('case_graph_powerlaw_d7_d64')
    n = g['cfg']['N_nodes']
    csr = orc.build_csr(g['edge_index'], n)
    G = _graph(g['edge_index'], n, T=T)
    gen = torch.Generator().manual_seed(1000 + d)
    h = torch.randn(n, d, generator=gen)
    bias = torch.randn(d, generator=gen)
    _, b = orc.degree_norms(csr)
    ref = torch.relu(orc.aggregate_sum_dense_f64(csr, h) * b.double().unsqueeze(1) + bias.double())
    out = G.spmm(h.to(DEV), row_scale=G.norm_in, bias=bias.to(DEV), relu=True)
    torch.testing.assert_close(out.cpu().double(), ref, atol=1e-5, rtol=1e-5)
    wide = torch.zeros(n, d + 8, device=DEV)                # strided source and destination rows (ld = d + 8, offset 4)
    wide[:, 4:4 + d] = h.to(DEV)
    dst = torch.full((n, d + 8), 7.0, device=DEV)
    G.spmm(wide[:, 4:4 + d], row_scale=G.norm_in, bias=bias.to(DEV), relu=True, out=dst[:, 4:4 + d])
    torch.testing.assert_close(dst[:, 4:4 + d].cpu().double(), ref, atol=1e-5, rtol=1e-5)
    assert bool((dst[:, :4] == 7.0).all()) and bool((dst[:, 4 + d:] == 7.0).all())     # nothing written outside the d columns
    assert torch.equal(G.spmm(h.to(DEV)), G.spmm(h.to(DEV)))                             # no atomics: bit-reproducible


@pytest.mark.parametrize('d', [3, 7, 16, 40, 64])
def test_spmm_narrow_large_graph_vs_c_oracle(d):
    """Narrow widths on a 200 000-node power-law graph (hub rows, long streams, windows that straddle many rows) against
    the C restatement of the aggregation; also an asymmetric graph with empty rows."""
    import oracle_c
    from gnn_tail_generalization_amd.data import synthetic_data
    data = synthetic_data('S-pl1M', seed=3, device=DEV, n_override=200000)
    n = 200000
    G = _graph(data.edge_index, n)
    assert G._plan.n_hubs > 0
    h = torch.randn(n, d, device=DEV)
    bias = torch.randn(d, device=DEV)
    got = G.spmm(h, row_scale=G.norm_in, bias=bias, relu=True)
    rowptr = G.rowptr.cpu().numpy().astype(np.int64)
    col = G.col.cpu().numpy()[:G.E]
    ref = oracle_c.spmm(rowptr, col, h.cpu().numpy(), scale=G.norm_in.cpu().numpy(), bias=bias.cpu().numpy(), relu=True)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(ref), atol=2e-5, rtol=1e-5)
    # directed random graph: many empty rows, reverse orientation
    gen = torch.Generator(device=DEV).manual_seed(5)
    ei = torch.stack([torch.randint(0, n, (300000,), device=DEV, generator=gen), torch.randint(0, n // 3, (300000,), device=DEV, generator=gen)])
    G2 = _graph(ei, n)
    assert G2.n_zero_in_degree > 0
    got2 = G2.spmm(h, bias=bias)
    ref2 = oracle_c.spmm(G2.rowptr.cpu().numpy().astype(np.int64), G2.col.cpu().numpy()[:G2.E], h.cpu().numpy(), bias=bias.cpu().numpy())
    torch.testing.assert_close(got2.cpu(), torch.from_numpy(ref2), atol=2e-5, rtol=1e-5)
    got3 = G2.spmm(h, transpose=True)
    ref3 = oracle_c.spmm(G2.rowptr_t.cpu().numpy().astype(np.int64), G2.col_t.cpu().numpy()[:G2.E], h.cpu().numpy())
    torch.testing.assert_close(got3.cpu(), torch.from_numpy(ref3), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('name', ['asym_multi', 'powerlaw'])
def test_edge_weight_form_matches_reference(name):
    """GCNConv.forward(graph, feat, edge_weight=w) — the u_mul_e variant of the aggregation (GCN.py:199-202) on the
    HIP kernels (cb_spmm_csr_weighted_f32 forward, the same on the reverse CSR + cb_spmm_edge_dot_f32 backward) against the
    unmodified reference: output 1e-4, gradients w.r.t. features, weight, bias, le and the edge weights."""
    from types import SimpleNamespace
    from gnn_tail_generalization_amd.GNN_model.GCN import GCNConv
    g = load_golden('edge_weight_fixture')[name]
    n = g['n']
    d_in, d_out = g['sd']['weight'].shape
    conv = GCNConv(d_in, d_out, args=SimpleNamespace(N_nodes=n), whetherHasSE='le' in g['sd'])
    conv.load_state_dict(g['sd'])
    conv = conv.to(DEV)
    from gnn_tail_generalization_amd.graph import CSRGraph
    with pytest.raises(ValueError, match='keep_edge_order'):   # the training path's cached graph does not keep the edge list
        _graph(g['edge_index'], n).edge_perm()
    G = CSRGraph(g['edge_index'].to(DEV), n, keep_edge_order=True)
    feat = g['feat'].to(DEV).requires_grad_(True)
    w = g['edge_weight'].to(DEV).requires_grad_(True)
    out, reg = conv(G, feat, edge_weight=w)
    torch.testing.assert_close(out.detach().cpu(), g['out'], atol=1e-4, rtol=1e-4)
    loss = (out * g['gout'].to(DEV)).sum() + (0.5 * reg if reg is not None else 0.0)
    loss.backward()
    torch.testing.assert_close(feat.grad.cpu(), g['d_feat'], atol=2e-5, rtol=2e-4)
    torch.testing.assert_close(w.grad.cpu(), g['d_edge_weight'], atol=2e-5, rtol=2e-4)
    for k, v in g['grads'].items():
        torch.testing.assert_close(getattr(conv, k).grad.cpu(), v, atol=5e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')
    with pytest.raises(AssertionError):                       # :200 — the reference asserts the length
        conv(G, feat, edge_weight=w[:-1])
    # wide rows (two column sweeps) + unit weights == the unweighted kernel
    h = torch.randn(n, 600, device=DEV)
    ones = torch.ones(G.E, device=DEV)
    torch.testing.assert_close(G.spmm_weighted(h, ones, False, row_scale=G.norm_in), G.spmm(h, row_scale=G.norm_in), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('d,dtype', [(256, torch.float32), (512, torch.float32), (40, torch.float32), (7, torch.float32), (256, torch.bfloat16)])
def test_in_place_accumulation_skips_rows_without_edges(d, dtype):
    """Intermediate halo passes of the node-sharded aggregation: cb_spmm_csr_acc_f32 with out == acc_init and no epilogue neither reads nor
    writes rows that have no edge in the slice's CSR.  A rectangular CSR in which most rows are empty (runs of empty rows at the start, in the
    middle and at the end of wavefront row blocks, a hub row, an empty row right after a hub row) gives, in place, exactly the out-of-place
    sums — and the same through the bf16-stored source rows of the bf16 wire."""
    from gnn_tail_generalization_amd.graph import CSRGraph
    n_rows, n_cols = 3001, 777
    gen = torch.Generator(device=DEV).manual_seed(d)
    rows = torch.randint(0, n_rows, (2500,), device=DEV, generator=gen)
    rows = rows[(rows % 16 != 3) & (rows % 16 != 4) & (rows < 2900) & ((rows < 1000) | (rows >= 1100))]      # holes: rows 1000..1099, 2900.. and two per block
    rows = torch.cat([rows, torch.full((700,), 41, device=DEV)])                                              # a hub row (T = 256); row 42 has few or no edges
    cols = torch.randint(0, n_cols, (rows.numel(),), device=DEV, generator=gen)
    G = CSRGraph.from_pairs(rows, cols, n_rows, n_cols)
    assert G._plan.n_hubs >= 1 and int((G.in_degrees() == 0).sum()) > n_rows // 3
    h = torch.randn(n_cols, d, device=DEV, generator=gen).to(dtype)
    acc = torch.randn(n_rows, d, device=DEV, generator=gen)
    want = G.spmm(h, acc_init=acc.clone(), out=torch.empty_like(acc))       # out-of-place: every row written
    assert torch.equal(want[G.in_degrees() == 0], acc[G.in_degrees() == 0])
    buf = acc.clone()
    got = G.spmm(h, acc_init=buf, out=buf)
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, want)
    # with an epilogue nothing is skipped (every row gets its scale / bias)
    rs = torch.rand(n_rows, device=DEV, generator=gen) + 0.5
    buf2 = acc.clone()
    got2 = G.spmm(h, row_scale=rs, acc_init=buf2, out=buf2)
    torch.testing.assert_close(got2, want * rs.unsqueeze(1), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('name,max_edges', [('case_graph_asym_multi', 7), ('case_graph_powerlaw_d7_d64', 100), ('case_r_initialbn_se111_L3_powerlaw', 64)])
def test_segmented_graph_with_int64_row_pointers_small(name, max_edges):
    """SURVEY.md 8(b) "int64 rowptr if E >= 2^31": SegmentedCSRGraph (int64 row pointers from cb_csr64_from_coo_i64, row blocks of at most
    max_edges edges with rebased int32 pointers) on the golden graphs with a tiny block size: int64 row pointers / columns / degree norms
    equal the int32 graph's bit for bit, and the aggregation (both orientations, epilogue, narrow and wide rows, running sums) equals it —
    a row is reduced inside one launch either way."""
    from gnn_tail_generalization_amd.graph import CSRGraph, SegmentedCSRGraph
    g = load_golden(name)
    n = int(g['cfg']['N_nodes']) if 'cfg' in g else int(g['edge_index'].max()) + 1
    ei = g['edge_index'].to(DEV)
    G = CSRGraph(ei, n)
    big = max(max_edges, int(G.in_degrees().max()), int(G.out_degrees().max()))
    S = SegmentedCSRGraph(ei, n, max_edges=big)
    assert len(S.segments) > 1 and S.rowptr.dtype == torch.int64
    assert torch.equal(S.rowptr, G.rowptr.long()) and torch.equal(S.col[:S.E], G.col[:G.E]) and S.symmetric == G.symmetric
    assert torch.equal(S.rowptr_t, G.rowptr_t.long()) and torch.equal(S.col_t[:S.E], G.col_t[:G.E])
    assert torch.equal(S.norm_in, G.norm_in) and torch.equal(S.norm_out, G.norm_out) and S.n_zero_in_degree == G.n_zero_in_degree
    assert sum(seg.E for _, _, seg in S.segments) == S.E and all(seg.E <= big for _, _, seg in S.segments)
    gen = torch.Generator(device=DEV).manual_seed(0)
    for d in (7, 40, 256):
        h = torch.randn(n, d, device=DEV, generator=gen)
        bias = torch.randn(d, device=DEV, generator=gen)
        # d > 16: one wavefront adds a row's edges in CSR order whatever block the row sits in -> bit-identical; d <= 16 (grouped-stream
        # kernel): the segmented scan's partial sums depend on where a row falls in its wavefront's edge window -> equal to rounding
        same = torch.equal if d > 16 else (lambda a_, b_: bool(torch.allclose(a_, b_, atol=1e-5, rtol=1e-5)))
        for tr in (False, True):
            assert same(S.spmm(h, transpose=tr), G.spmm(h, transpose=tr))
        assert same(S.spmm(h, row_scale=S.norm_in, bias=bias, relu=True), G.spmm(h, row_scale=G.norm_in, bias=bias, relu=True))
        acc = torch.randn(n, d, device=DEV, generator=gen)
        assert same(S.spmm(h, acc_init=acc), G.spmm(h, acc_init=acc))
    with pytest.raises(ValueError, match='alone holds more'):
        SegmentedCSRGraph(ei, n, max_edges=max(int(G.in_degrees().max()) - 1, 0))


def test_teacher_runs_on_a_segmented_graph():
    """GCNConv / TricksComb on a SegmentedCSRGraph (what build_graph returns from 2^31 edges on; here forced on a golden case): the
    operator path gives the reference's logits, loss and gradients."""
    from helpers import product_model
    from gnn_tail_generalization_amd.graph import SegmentedCSRGraph
    g = load_golden('case_r_initialbn_se111_L3_powerlaw')
    args, model = product_model(g['cfg'], g['sd'], DEV)
    x, ei, y, mask = g['x'].to(DEV), g['edge_index'].to(DEV), g['y'].to(DEV), g['train_mask'].to(DEV)
    tc = model.model.model
    tc.dglgraph = SegmentedCSRGraph(ei, x.shape[0], max_edges=200)
    assert len(tc.dglgraph.segments) > 2
    model.train()
    res = model.get_3_embs(x, ei, mask)
    loss = torch.nn.functional.nll_loss(torch.nn.functional.log_softmax(res.emb4classi, 1), y[mask]) + args.se_reg * model.se_reg_all
    loss.backward()
    torch.testing.assert_close(res.emb4classi_full.detach().cpu(), g['train_out'], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), g['train_loss'], atol=1e-4, rtol=1e-5)
    for k, v in g['grads'].items():
        got = dict(model.named_parameters())[k].grad
        torch.testing.assert_close(got.cpu(), v, atol=2e-5, rtol=2e-4, msg=lambda m, k=k: f'{k}: {m}')
