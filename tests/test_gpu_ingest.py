"""GPU: the device-side graph analysis in front of the path (csrc/cb_ingest.hip through the C ABI) —
degrees, repeated-median head/tail selection, isolation crafting, symmetrisation — against the fixture produced by the
unmodified reference's per-edge Python loops (tests/golden/utils_fixture.pt), against numpy on random inputs with the edge
cases the reference's np.median / np.where define, and at the benchmark's full size (10^8 edges) against the tensor
formulation."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MODES = ['top50', 'top25', 'top12', 'top6', 'top3', 'bottom50', 'bottom25', 'bottom12', 'bottom6', 'bottom3']


def test_device_analysis_matches_reference_fixture():
    from gnn_tail_generalization_amd import utils
    from gnn_tail_generalization_amd.data import Data
    fx = torch.load(os.path.join(GOLDEN, 'utils_fixture.pt'), weights_only=False)
    degs = fx['degs'].to(DEV)
    for mode in MODES:
        idx, mask = utils.partial_sorted_select_device(degs, mode)
        want = fx['idx_' + mode]
        assert torch.equal(idx.cpu(), want), mode
        m = torch.zeros(degs.numel(), dtype=torch.bool)
        m[want] = True
        assert torch.equal(mask.cpu(), m), mode
    assert torch.equal(utils.ensure_symmetric(fx['asym_edge_index'].to(DEV)).cpu(), fx['ensure_symmetric'])
    assert torch.equal(utils.to_undirected(fx['asym_edge_index'].to(DEV), 60).cpu(), fx['ensure_symmetric'])
    for special in (0, 1):
        pre = f'sga{special}_'
        ei = fx[pre + 'edge_index_in']
        data = Data(x=torch.zeros(150, 3, device=DEV), edge_index=ei.clone().to(DEV))
        utils.save_graph_analyze(150, data, special, verbose=False)
        assert torch.equal(data.edge_index.cpu(), fx[pre + 'edge_index_out'])
        for k in ['zero_deg_idx', 'small_deg_idx', 'large_deg_idx', 'zero_deg_mask', 'small_deg_mask', 'large_deg_mask']:
            if pre + k in fx:
                assert torch.equal(torch.as_tensor(getattr(data, k)).cpu(), fx[pre + k]), (special, k)
    d0, d1 = utils.graph_analyze(150, fx['sga0_edge_index_in'].to(DEV))
    r0, r1 = utils.graph_analyze(150, fx['sga0_edge_index_in'])
    assert np.array_equal(d0, r0) and np.array_equal(d1, r1)


@pytest.mark.parametrize('case', ['random', 'heavy_tail', 'all_equal', 'all_zero', 'single', 'two_values', 'ragged_4099'])
def test_partial_sorted_select_vs_numpy(case):
    from gnn_tail_generalization_amd import utils
    rng = np.random.default_rng(11)
    arr = {'random': rng.integers(0, 50, 20000), 'heavy_tail': (rng.random(50000) ** -1.3).astype(np.int64),
           'all_equal': np.full(777, 5), 'all_zero': np.zeros(1000, dtype=np.int64), 'single': np.array([3]),
           'two_values': np.array([0, 9] * 501 + [9]), 'ragged_4099': rng.integers(0, 3, 4099)}[case]
    dev = torch.from_numpy(arr.astype(np.int32)).to(DEV)
    for mode in MODES:
        want = utils.get_partial_sorted_idx(arr, mode)
        idx, mask = utils.partial_sorted_select_device(dev, mode)
        assert np.array_equal(idx.cpu().numpy(), want), (case, mode)
        assert int(mask.sum()) == len(want)


def test_compaction_kernels_edge_cases():
    from gnn_tail_generalization_amd import utils
    from gnn_tail_generalization_amd.data import Data
    # empty edge list, nothing flagged, everything flagged, self-loops survive
    n = 10
    data = Data(x=torch.zeros(n, 1, device=DEV), edge_index=torch.zeros((2, 0), dtype=torch.int64, device=DEV))
    data.zero_deg_mask = torch.zeros(n, dtype=torch.bool, device=DEV)
    utils.craft_isolation_v2(data, verbose=False)
    assert data.edge_index.shape == (2, 0)
    ei = torch.tensor([[0, 1, 2, 3, 3, 4], [1, 1, 2, 4, 3, 0]], device=DEV)
    for flagged, keep in [([], [0, 1, 2, 3, 4, 5]), (list(range(n)), [1, 2, 4]), ([3], [0, 1, 2, 4, 5])]:
        data = Data(x=torch.zeros(n, 1, device=DEV), edge_index=ei.clone())
        z = torch.zeros(n, dtype=torch.bool, device=DEV)
        z[flagged] = True
        data.zero_deg_mask = z
        utils.craft_isolation_v2(data, verbose=False)
        assert torch.equal(data.edge_index, ei[:, keep]), flagged
    # transposed (non-contiguous) edge_index view, as utils.py:745 produces
    eit = ei.t().contiguous().t()
    assert torch.equal(utils.ensure_symmetric(eit), utils.ensure_symmetric(ei))
    with pytest.raises(ValueError):
        utils.to_undirected(torch.tensor([[0, 7], [1, 2]], device=DEV), 4)


def test_full_size_analysis_s_pl10m():
    """10^7 nodes / 10^8 edge_index columns: the device kernels against the tensor formulation (bincount / boolean indexing /
    unique), plus order preservation of the crafted edge list."""
    from gnn_tail_generalization_amd import utils
    from gnn_tail_generalization_amd.data import Data, synthetic_data
    data = synthetic_data('S-pl10M', seed=0, device=DEV)
    n, ei = data.x.shape[0], data.edge_index
    d_out, d_in = utils._degrees_device(n, ei)
    assert torch.equal(d_in.long(), torch.bincount(ei[1], minlength=n)) and torch.equal(d_out.long(), torch.bincount(ei[0], minlength=n))
    arr = d_in.cpu().numpy()
    for mode in ('top3', 'bottom3', 'top6'):
        idx, mask = utils.partial_sorted_select_device(d_in, mode)
        want = utils.get_partial_sorted_idx(arr, mode)
        assert np.array_equal(idx.cpu().numpy(), want), mode
        assert int(mask.sum()) == len(want)
    # craft: flag the 'top6' nodes (what the special split isolates) and compare with the boolean-index formulation
    z = mask
    probe = Data(x=data.x[:, :1], edge_index=ei)
    probe.zero_deg_mask = z
    utils.craft_isolation_v2(probe, verbose=False)
    keep = ~((ei[0] != ei[1]) & (z[ei[0]] | z[ei[1]]))
    assert torch.equal(probe.edge_index, ei[:, keep])
    del probe, keep
    # symmetrisation of a directed half (row < col) recovers the undirected part of the graph, self-loops once
    half = ei[:, ei[0] <= ei[1]]
    sym = utils.to_undirected(half, n)
    key = torch.unique(torch.cat([half[0] * n + half[1], half[1] * n + half[0]]))
    assert torch.equal(sym[0] * n + sym[1], key)
