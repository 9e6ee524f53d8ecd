"""CPU: the oracle's C restatement against the numpy/torch oracle (integer work bit-exact)."""
import numpy as np
import pytest
import torch

import coldbrew_oracle as orc
import oracle_c
from conftest import load_golden


@pytest.mark.parametrize('name', ['case_graph_asym_multi', 'case_graph_powerlaw_d7_d64', 'case_graph_example',
                                  'case_graph_zero_in_degree'])
def test_c_csr_and_spmm_match_numpy_oracle(name):
    g = load_golden(name)
    n = g['cfg']['N_nodes']
    csr = orc.build_csr(g['edge_index'], n)
    rowptr, col = oracle_c.csr_from_coo(csr.dst, csr.src, n)
    assert np.array_equal(rowptr, csr.rowptr) and np.array_equal(col, csr.col)
    rowptr_t, col_t = oracle_c.csr_from_coo(csr.src, csr.dst, n)
    assert np.array_equal(rowptr_t, csr.rowptr_t) and np.array_equal(col_t, csr.col_t)
    a, b = orc.degree_norms(csr)
    assert np.array_equal(oracle_c.deg_norm(csr.rowptr_t), a.numpy()) and np.array_equal(oracle_c.deg_norm(csr.rowptr), b.numpy())
    h = torch.randn(n, 19, generator=torch.Generator().manual_seed(0), requires_grad=True)
    ref = orc.aggregate_sum(csr, h)
    got = oracle_c.aggregate_sum(csr, h)
    torch.testing.assert_close(got, ref, atol=1e-5, rtol=1e-5)
    w = torch.randn(n, 19, generator=torch.Generator().manual_seed(1))
    g1, = torch.autograd.grad((ref * w).sum(), h)
    g2, = torch.autograd.grad((got * w).sum(), h)
    torch.testing.assert_close(g2, g1, atol=1e-5, rtol=1e-5)


def test_c_csr_rejects_out_of_range():
    with pytest.raises(ValueError):
        oracle_c.csr_from_coo(np.array([0, 9]), np.array([1, 2]), 4)
