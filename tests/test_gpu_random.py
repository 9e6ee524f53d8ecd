"""GPU: randomised parity sweep of the ingest + aggregation against the oracle — many small random graphs with
ragged shapes (N not a multiple of the 16-row wave block, empty rows, duplicate edges, rows above and below the
hub threshold, tiny and odd feature widths), fused epilogue included."""
import numpy as np
import pytest
import torch

import coldbrew_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _random_graph(rng, n, e, hubby):
    if hubby:   # a few destination rows collect most edges
        w = rng.random(n) ** 6 + 1e-3
        dst = rng.choice(n, size=e, p=w / w.sum())
    else:
        dst = rng.integers(0, n, size=e)
    src = rng.integers(0, n, size=e)
    return torch.from_numpy(np.stack([src, dst]).astype(np.int64))


@pytest.mark.parametrize('seed', range(24))
def test_random_graph_spmm_matches_oracle(seed):
    from gnn_tail_generalization_amd.graph import CSRGraph
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 15, 16, 17, 33, 100, 257, 1000]))
    e = int(rng.choice([0, 1, n, 5 * n, 40 * n]))
    d = int(rng.choice([1, 2, 5, 64, 127, 128, 256, 300, 512]))
    T = int(rng.choice([1, 3, 16, 256]))
    ei = _random_graph(rng, n, e, hubby=bool(seed % 2))
    csr = orc.build_csr(ei, n)
    G = CSRGraph(ei.to(DEV), n, hub_threshold=T)
    assert np.array_equal(G.rowptr.cpu().numpy().astype(np.int64), csr.rowptr)
    assert e == 0 or np.array_equal(G.col.cpu().numpy()[:e], csr.col)
    h = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(d).astype(np.float32))
    a, b = orc.degree_norms(csr)
    want = orc.aggregate_sum_dense_f64(csr, h) if n <= 1000 else None
    got = G.spmm(h.to(DEV), row_scale=G.norm_in, bias=bias.to(DEV), relu=bool(seed % 3 == 0))
    ref = want * b.double().unsqueeze(1) + bias.double()
    if seed % 3 == 0:
        ref = torch.relu(ref)
    torch.testing.assert_close(got.cpu().double(), ref, atol=2e-5 * max(1.0, float(csr.in_deg.max()) ** 0.5), rtol=1e-5)
    A = np.zeros((n, n))
    np.add.at(A, (csr.src, csr.dst), 1.0)
    got_t = G.spmm(h.to(DEV), transpose=True)
    torch.testing.assert_close(got_t.cpu().double(), torch.from_numpy(A @ h.double().numpy()),
                               atol=2e-5 * max(1.0, float(csr.out_deg.max()) ** 0.5), rtol=1e-5)


@pytest.mark.parametrize('seed', range(8))
def test_random_fused_epilogue_matches_composition(seed):
    """cb_spmm_csr_fused_f32 == plain aggregation + relu + mix + dropout composed from the separately tested ops."""
    from gnn_tail_generalization_amd import ops, trunk
    from gnn_tail_generalization_amd.graph import CSRGraph
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([5, 16, 100, 700]))
    d = int(rng.choice([256, 512]))
    T = int(rng.choice([2, 256]))
    ei = _random_graph(rng, n, 12 * n, hubby=True)
    G = CSRGraph(ei.to(DEV), n, hub_threshold=T)
    z = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    x0 = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(d).astype(np.float32)).to(DEV)
    p, sd, alpha = float(rng.choice([0.0, 0.3])), 777 + seed, 0.1
    bits, nxt, act = trunk._fused_spmm(G, z, bias, x0, 1 - alpha, alpha, p, sd, want_act=True)
    ref_act = G.spmm(z, row_scale=G.norm_in, bias=bias, relu=True)
    torch.testing.assert_close(act, ref_act, atol=1e-6, rtol=1e-6)
    mix = (1 - alpha) * ref_act + alpha * x0
    if p > 0:
        mix = torch.where(ops.dropout_keep_mask((n, d), p, sd, DEV), mix / (1 - p), torch.zeros_like(mix))
    torch.testing.assert_close(nxt, mix, atol=2e-6, rtol=1e-5)
    # mask bits: word k of (row, tile), bit l  <->  column 256*tile + 4*l + k
    b = bits.cpu().numpy().astype(np.uint64).reshape(n, d // 256, 4)
    cols = np.arange(d)
    tile, lane, k = cols // 256, (cols % 256) // 4, cols % 4
    got_mask = ((b[:, tile, k] >> lane.astype(np.uint64)) & np.uint64(1)).astype(bool)
    want_mask = ref_act > 0                 # gradient passes where the activation is positive AND the dropout keeps the element
    if p > 0:
        want_mask &= ops.dropout_keep_mask((n, d), p, sd, DEV)
    assert np.array_equal(got_mask, want_mask.cpu().numpy())


@pytest.mark.parametrize('seed', range(16))
def test_random_gemm_shapes_strides_and_epilogues(seed):
    """Dense contractions on random ragged shapes: every size around the tile / K-step boundaries, operands that are
    column slices of wider buffers (leading dimension > width, pointer offsets), shapes that qualify for the three-limb
    path (multiples of 4, 16-byte aligned) and shapes that fall back to the fp32-input MFMA path, with the fused epilogue
    terms switched on at random.  Reference: fp64 on the host."""
    from gnn_tail_generalization_amd import gemm
    rng = np.random.default_rng(500 + seed)
    M = int(rng.choice([1, 3, 63, 64, 65, 127, 128, 129, 255, 257, 640, 1031]))
    K = int(rng.choice([4, 8, 12, 16, 20, 36, 64, 100, 128, 260, 7, 33]))
    N = int(rng.choice([4, 8, 40, 60, 64, 68, 128, 132, 256, 260, 300, 5, 130]))
    pad_a, pad_b = int(rng.choice([0, 4, 8, 3])), int(rng.choice([0, 4, 12, 1]))
    off_a, off_b = int(rng.choice([0, 4, 1])), int(rng.choice([0, 4]))
    abuf = torch.from_numpy(rng.standard_normal((M, K + pad_a + off_a)).astype(np.float32)).to(DEV)
    bbuf = torch.from_numpy(rng.standard_normal((K, N + pad_b + off_b)).astype(np.float32)).to(DEV)
    a, b = abuf[:, off_a:off_a + K], bbuf[:, off_b:off_b + N]
    rs = torch.from_numpy(rng.random(M).astype(np.float32) + 0.5).to(DEV) if rng.random() < 0.6 else None
    add = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV) if rng.random() < 0.5 else None
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV) if rng.random() < 0.5 else None
    relu = bool(rng.random() < 0.5)
    got = gemm.mm_nn(a, b, rowscale=rs, addend=add, bias=bias, relu=relu).cpu().double()
    ref = a.cpu().double() @ b.cpu().double()
    if rs is not None:
        ref = ref * rs.cpu().double().unsqueeze(1)
    if add is not None:
        ref = ref + add.cpu().double()
    if bias is not None:
        ref = ref + bias.cpu().double()
    if relu:
        ref = torch.relu(ref)
    scale = (a.cpu().double().abs() @ b.cpu().double().abs()).max().item() + 10.0
    assert (got - ref).abs().max().item() <= 3e-6 * scale, (M, K, N, pad_a, off_a, pad_b, off_b)
    # weight-gradient form on the same operands: a^T @ (rs * g)
    gbuf = torch.from_numpy(rng.standard_normal((M, N + pad_b + off_b)).astype(np.float32)).to(DEV)
    g = gbuf[:, off_b:off_b + N]
    got_t = gemm.mm_tn(a, g, rowscale=rs).cpu().double()
    gs = g.cpu().double() * (rs.cpu().double().unsqueeze(1) if rs is not None else 1.0)
    ref_t = a.cpu().double().t() @ gs
    scale_t = (a.cpu().double().abs().t() @ gs.abs()).max().item() + 10.0
    assert (got_t - ref_t).abs().max().item() <= 3e-6 * scale_t, (M, K, N)
