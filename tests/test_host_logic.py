"""CPU tests of the host logic and the boundary: option pipeline vs the reference's namespace,
state_dict contract, caller-side utils vs reference fixtures, C-ABI symbol export, loud failure
without a GPU.  No compute call is made through the HIP library here."""
import contextlib
import ctypes
import io
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden_cases, load_golden
from helpers import product_args


def _opts(argv):
    from gnn_tail_generalization_amd.base_options import BaseOptions
    from gnn_tail_generalization_amd.utils import set_arch_configs
    with contextlib.redirect_stdout(io.StringIO()):
        args = BaseOptions().get_arguments(argv)
        set_arch_configs(args)
    return args


def test_options_match_reference_namespace():
    """tests/golden/options.json = the reference's own get_arguments()+set_arch_configs() for README lines."""
    ref = json.load(open(os.path.join(GOLDEN, 'options.json')))
    skip = {'argv', 'exp_mode', 'cuda', 'records_file'}
    for name, want in ref.items():
        args = _opts(['--manual_assign_GPU=0'] + want['argv'])
        got = vars(args)
        for k, v in want.items():
            if k in skip or '.' in k:
                continue
            assert k in got, f'{name}: option {k} missing'
            assert got[k] == v, f'{name}: {k}: {got[k]!r} != reference {v!r}'
        assert list(args.TeacherGNN.whetherHasSE) == want['TeacherGNN.whetherHasSE']
        assert list(args.TeacherGNN.neurons_proj2class) == want['TeacherGNN.neurons_proj2class']


def test_best_config_concatenated_names():
    assert _opts(['--dataset=Cora']).type_trick == 'NoResNodeNorm'
    assert _opts(['--dataset=Pubmed']).type_trick == 'InitialBatchNorm'
    assert _opts(['--dataset=ogbn-arxiv']).type_trick == 'InitialBatchNorm'
    assert _opts(['--dataset=S-pl10M']).type_trick == 'InitialBatchNorm'
    a = _opts(['--dataset=Cora', '--force_set_to_best_config=0'])
    assert a.type_trick == 'Initial+BatchNorm' and a.lr == 0.001 and a.exp_mode == 'coldbrew'
    with pytest.raises(SystemExit):
        _opts(['--dataset=Cora', '--whetherHasSE=010'])


@pytest.mark.parametrize('name', golden_cases())
def test_state_dict_contract(name):
    """Parameter names, shapes and order equal the reference's (checkpoint interchange)."""
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    g = load_golden(name)
    args = product_args(g['cfg'])
    model = TeacherGNN(args)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g['sd'].keys())
    for k, v in g['sd'].items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict(g['sd'], strict=True)


def test_same_seed_same_init_as_reference():
    """Construction order / initialisers match: with the same torch seed the fresh parameters are
    bit-identical to the reference's (needs /root/reference; skipped on the GPU box)."""
    import ref_import
    if not ref_import.reference_available():
        pytest.skip('reference tree not present')
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_golden
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    ns = ref_import.load_reference()
    for c in make_golden.CASES:
        if c['name'] not in ('nr_se111_L3', 'r_initialbn_se111_L2', 'r_dense_concat_L3', 'norm_groupnorm',
                             'r_jumping_attention_L2', 'teacher_learnable_input'):
            continue
        with contextlib.redirect_stdout(io.StringIO()):
            rargs, _, _, _, _, _ = make_golden.build(ns, c)
            torch.manual_seed(123)
            ref_model = ns.GNN_normalizations.TeacherGNN(rargs)
        args = product_args(make_golden.cfg_of(rargs, c))
        torch.manual_seed(123)
        mine = TeacherGNN(args)
        ref_sd = ref_model.state_dict()
        for k, v in mine.state_dict().items():
            assert torch.equal(v, ref_sd[k]), f'{c["name"]}: {k}'


def test_utils_match_reference_fixture():
    from gnn_tail_generalization_amd import utils
    from gnn_tail_generalization_amd.data import Data
    fx = torch.load(os.path.join(GOLDEN, 'utils_fixture.pt'), weights_only=False)
    arr = fx['degs'].numpy()
    for k, v in fx.items():
        if k.startswith('idx_'):
            assert np.array_equal(utils.get_partial_sorted_idx(arr, k[4:]), v.numpy()), k
    assert torch.equal(utils.ensure_symmetric(fx['asym_edge_index']), fx['ensure_symmetric'])
    for special in (0, 1):
        pre = f'sga{special}_'
        ei = fx[pre + 'edge_index_in']
        data = Data(x=torch.zeros(150, 3), edge_index=ei.clone())
        utils.save_graph_analyze(150, data, special, verbose=False)
        assert torch.equal(data.edge_index, fx[pre + 'edge_index_out'])
        for k in ['zero_deg_idx', 'small_deg_idx', 'large_deg_idx', 'zero_deg_mask', 'small_deg_mask', 'large_deg_mask']:
            if pre + k in fx:
                assert torch.equal(torch.as_tensor(getattr(data, k)), fx[pre + k]), k


def test_synthetic_data_postconditions():
    """load_data post-conditions (trainer_node_classification.py:655-662): symmetric, coalesced,
    self-loops appended last; ogbn family: to_undirected, no self-loops, every degree >= 1."""
    from gnn_tail_generalization_amd.data import synthetic_data
    d = synthetic_data('S-tiny', seed=3)
    n = d.x.shape[0]
    ei = d.edge_index
    body, loops = ei[:, :-n], ei[:, -n:]
    assert torch.equal(loops[0], torch.arange(n)) and torch.equal(loops[1], torch.arange(n))
    key = body[0] * n + body[1]
    assert torch.equal(key, torch.unique(key)) and (body[0] != body[1]).all()
    assert torch.equal(torch.sort(body[1] * n + body[0])[0], key)
    assert torch.equal(synthetic_data('S-tiny', seed=3).edge_index, ei)          # seeded
    d2 = synthetic_data('S-arxiv', seed=0, n_override=3000)
    e2 = d2.edge_index
    assert (e2[0] != e2[1]).all() and torch.bincount(e2[1], minlength=3000).min() >= 1
    assert d2.x.shape == (3000, 128) and int(d2.y.max()) < 40


def test_c_abi_exports_every_declared_symbol():
    """Every function declared in include/coldbrew_hip.h is exported by the in-tree .so and bound in _lib."""
    from gnn_tail_generalization_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'coldbrew_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(cb_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert os.path.isfile(_lib.LIB_PATH), 'build the extension first: python __graft_entry__.py'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().cb_version() >= 1


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected by the product path."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gnn_tail_generalization_amd import _lib
    from gnn_tail_generalization_amd.GNN_model import TeacherGNN
    g = load_golden('case_nr_se000_L2')
    model = TeacherGNN(product_args(g['cfg']))
    with pytest.raises(_lib.HipExtensionError):
        model(g['x'], g['edge_index'])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'gnn-tail-generalization_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not _IMPORTS_ORACLE.search(src), f
    assert not _IMPORTS_ORACLE.search(open(os.path.join(ROOT, 'main.py')).read())


_IMPORTS_ORACLE = re.compile(r"^\s*(import|from)\s+\S*(coldbrew_oracle|ref_import|oracle)|sys\.path\S*oracle|'oracle'|\"oracle\"", re.M)


def test_three_limb_split_arithmetic_numpy():
    """The arithmetic the default GEMM path relies on (csrc/cb_limb_core.h `split3x2`), restated in numpy: three successive
    round-to-nearest-even bf16 conversions of what is left give a = hi + mid + lo exactly, with |mid| <= 2^-8 |a| and
    |lo| <= 2^-17 |a|, and the six limb products the kernels issue reproduce a*b to within 2^-24 |a*b| (half an fp32 ulp)."""
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.exp(rng.uniform(-20, 20, 200000)).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.1415927, 16777215.0, 1.0000001, 1.9999999, 1.00390625, 3.4e37, 1e-30],
                                 dtype=np.float32)])

    def rne_bf16(x):                                     # v_cvt_pk_bf16_f32 on finite values
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    def split(x):
        hi = rne_bf16(x)
        r1 = x - hi                                     # exact
        mid = rne_bf16(r1)
        r2 = r1 - mid                                   # exact
        return hi, mid, rne_bf16(r2), r2

    f = lambda x: x.astype(np.float64)
    hi, mid, lo, r2 = split(a)
    assert np.array_equal(lo, r2)                       # the last residual is a bf16 already
    assert np.array_equal(f(hi) + f(mid) + f(lo), f(a))
    nz = a != 0
    assert (np.abs(f(mid))[nz] <= 2.0 ** -8 * np.abs(f(a))[nz]).all() and (np.abs(f(lo))[nz] <= 2.0 ** -16.9 * np.abs(f(a))[nz]).all()
    b = np.roll(a, 7)
    bh, bm, bl, _ = split(b)
    six = f(hi) * f(bh) + f(hi) * f(bm) + f(mid) * f(bh) + f(hi) * f(bl) + f(lo) * f(bh) + f(mid) * f(bm)
    exact = f(a) * f(b)
    ok = np.isfinite(exact) & (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    rel = np.abs(six - exact)[ok] / np.abs(exact)[ok]
    assert rel.max() <= 2.0 ** -24 and rel.mean() <= 2.0 ** -27


def test_public_header_is_plain_c(tmp_path):
    """include/coldbrew_hip.h is the drop-in boundary: it must compile as C99 (extern "C" guards, no C++ / torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    src = tmp_path / 'h.c'
    src.write_text('#include "coldbrew_hip.h"\nint main(void) { return cb_version() < 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-fsyntax-only', '-I' + inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_int32_edge_contract_is_a_per_rank_guarantee():
    """SURVEY.md 8(b) 'int64 rowptr if E >= 2^31': the int32-indexed CSRGraph refuses 2^31 or more edge_index columns and names the two
    ways on (graph.build_graph -> SegmentedCSRGraph with int64 row pointers on one device: tests/test_gpu_graph_spmm.py; or sharding), and
    the edge-balanced partition keeps every rank's block at E / P (+ one row) edges, i.e. below the limit from P = 2 on up to 2^32."""
    import torch
    from gnn_tail_generalization_amd import graph as cbgraph
    from gnn_tail_generalization_amd.dist import Partition
    assert cbgraph.INT32_EDGE_LIMIT == 2 ** 31 - 1

    class FakeEdges:          # a [2, 2^31] edge_index without allocating 32 GB: only what __init__ touches before the size check
        shape = (2, 2 ** 31)
        is_cuda = True

        def dim(self):
            return 2

        def to(self, *_a, **_k):
            return self

        def contiguous(self):
            return self

        device = 'cuda:0'
    import pytest
    from gnn_tail_generalization_amd import _lib
    real = (_lib.load, _lib.require_device)
    _lib.load, _lib.require_device = (lambda: None), (lambda *a: None)
    try:
        with pytest.raises(ValueError, match='SegmentedCSRGraph.*shard the graph'):
            cbgraph.CSRGraph(FakeEdges(), 10)
    finally:
        _lib.load, _lib.require_device = real
    # per-rank edge counts of the balanced partition on a heavy-tailed degree vector scaled to 3 * 2^30 edges
    gen = torch.Generator().manual_seed(1)
    deg = (torch.rand(1_000_000, generator=gen) ** -0.7).to(torch.int64)
    scale = (3 * 2 ** 30) // int(deg.sum()) + 1
    deg = deg * scale
    total = int(deg.sum())
    assert total >= 3 * 2 ** 30
    for world in (2, 4, 8):
        shares = [int(deg[Partition.balanced(deg, world, r).lo():Partition.balanced(deg, world, r).hi()].sum()) for r in range(world)]
        assert sum(shares) == total and max(shares) <= total // world + int(deg.max()) + 12 * deg.numel()
        assert max(shares) < 2 ** 31 - 1
