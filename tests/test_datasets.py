"""Readers of the raw dataset formats (gnn-tail-generalization_amd/datasets.py) and load_data's post-conditions
(trainer_node_classification.py:570-577,616-670), on files written here in those formats — the real files cannot be downloaded."""
import collections
import gzip
import os
import pickle
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from gnn_tail_generalization_amd import data as cb_data
from gnn_tail_generalization_amd import datasets


def _write_planetoid(raw, name, X, Y, n_train, n_all, test_index, n_classes, graph):
    os.makedirs(raw, exist_ok=True)
    onehot = lambda y: np.eye(n_classes, dtype=np.int32)[y]
    parts = {'x': sp.csr_matrix(X[:n_train]), 'allx': sp.csr_matrix(X[:n_all]), 'tx': sp.csr_matrix(X[test_index]),
             'y': onehot(Y[:n_train]), 'ally': onehot(Y[:n_all]), 'ty': onehot(Y[test_index]), 'graph': graph}
    for k, v in parts.items():
        with open(os.path.join(raw, f'ind.{name}.{k}'), 'wb') as f:
            pickle.dump(v, f, protocol=2)
    with open(os.path.join(raw, f'ind.{name}.test.index'), 'w') as f:
        f.write('\n'.join(str(int(i)) for i in test_index) + '\n')


def _graph(rng, n, present):
    g = collections.defaultdict(list)
    nodes = np.asarray(sorted(present))
    for u in nodes:
        for v in rng.choice(nodes, size=3):
            g[int(u)].append(int(v))           # directed entries, duplicates and a few self-loops included
    g[int(nodes[0])].append(int(nodes[0]))
    g[int(nodes[1])] += [int(nodes[2]), int(nodes[2])]
    return g


def _expected_edges(g, n):
    e = {(u, v) for u, vs in g.items() for v in vs if u != v}
    return torch.tensor(sorted(e), dtype=torch.int64).t()


@pytest.mark.parametrize('name', ['cora', 'citeseer'])
def test_planetoid_reader(tmp_path, name):
    rng = np.random.default_rng(3)
    n_train, n_all, n_cls, f = 10, 40, 3, 12
    if name == 'cora':
        n = 60
        test_index = rng.permutation(np.arange(n_all, n))
        present = range(n)
    else:                                   # Citeseer: the test range has holes (isolated nodes that appear in no file)
        n = 65
        test_index = rng.permutation(rng.choice(np.arange(n_all, n), size=20, replace=False))
        present = sorted(set(range(n_all)) | set(int(i) for i in test_index))
    X = (rng.random((n, f)) < 0.3).astype(np.float32) * rng.integers(1, 4, size=(n, f))
    X[5] = 0                                # a row without features: NormalizeFeatures clamps its sum at 1
    Y = rng.integers(0, n_cls, size=n)
    g = _graph(rng, n, present)
    raw = str(tmp_path / 'Data' / 'raw')
    _write_planetoid(raw, name, X, Y, n_train, n_all, test_index, n_cls, g)
    out = datasets.read_planetoid(raw, name.capitalize())
    holes = sorted(set(range(n)) - set(present))
    Xe, Ye = X.copy(), Y.copy()
    Xe[holes] = 0
    Ye[holes] = 0
    Xe = Xe / np.maximum(Xe.sum(1, keepdims=True), 1.0)
    np.testing.assert_allclose(out['x'].numpy(), Xe, rtol=1e-6)
    assert out['y'].tolist() == Ye.tolist()
    assert torch.equal(out['edge_index'], _expected_edges(g, n))
    assert out['train_mask'].nonzero().flatten().tolist() == list(range(n_train))
    assert sorted(out['test_mask'].nonzero().flatten().tolist()) == sorted(int(i) for i in test_index)
    assert int(out['val_mask'].sum()) == min(500, n - n_train)


def _fake_trainer():
    return types.SimpleNamespace(device=torch.device('cpu'))


def test_load_data_from_planetoid_raw_files(tmp_path, capsys):
    rng = np.random.default_rng(5)
    n, n_train, n_all, n_cls, f = 700, 20, 600, 4, 9
    test_index = rng.permutation(np.arange(n_all, n))
    X = rng.random((n, f)).astype(np.float32)
    Y = rng.integers(0, n_cls, size=n)
    g = _graph(rng, n, range(n))
    _write_planetoid(str(tmp_path / 'Cora' / 'Cora' / 'raw'), 'cora', X, Y, n_train, n_all, test_index, n_cls, g)
    d = cb_data.load_data('Cora', 0, _fake_trainer(), root=str(tmp_path))
    assert 'Planetoid raw files' in capsys.readouterr().out
    # the reference's Cora split (:637-640) and edge post-conditions (:655-662): symmetric, self-loops appended last, one per node
    assert d.train_mask.nonzero().flatten().tolist() == list(range(600)) and torch.equal(d.test_mask, ~d.train_mask)
    ei = d.edge_index
    body, loops = ei[:, :-n], ei[:, -n:]
    assert torch.equal(loops[0], torch.arange(n)) and torch.equal(loops[1], torch.arange(n))
    assert bool((body[0] != body[1]).all())
    und = {(u, v) for u, vs in g.items() for v in vs if u != v}
    und |= {(v, u) for u, v in und}
    assert torch.equal(body, torch.tensor(sorted(und), dtype=torch.int64).t())
    assert torch.equal(d.train_idx, torch.arange(600)) and torch.equal(d.test_idx, torch.arange(600, n))


def test_ogbn_reader_and_load_data(tmp_path, capsys):
    rng = np.random.default_rng(7)
    n, e, f, c = 50, 200, 6, 5
    root = tmp_path / 'ogbn_arxiv'
    os.makedirs(root / 'raw')
    os.makedirs(root / 'split' / 'time')
    edges = rng.integers(0, n, size=(e, 2))
    X = rng.standard_normal((n, f)).astype(np.float32)
    Y = rng.integers(0, c, size=(n, 1))
    perm = rng.permutation(n)
    split = {'train': perm[:30], 'valid': perm[30:40], 'test': perm[40:]}

    def dump(path, a, fmt):
        with gzip.open(path, 'wt') as fh:
            np.savetxt(fh, a, fmt=fmt, delimiter=',')
    dump(root / 'raw' / 'edge.csv.gz', edges, '%d')
    dump(root / 'raw' / 'node-feat.csv.gz', X, '%.9g')
    dump(root / 'raw' / 'node-label.csv.gz', Y, '%d')
    for k, v in split.items():
        dump(root / 'split' / 'time' / f'{k}.csv.gz', v.reshape(-1, 1), '%d')
    blob, sp_idx = datasets.read_ogbn(str(root))
    np.testing.assert_allclose(blob['x'].numpy(), X, rtol=1e-6)
    assert blob['y'].tolist() == Y.reshape(-1).tolist()
    und = {(int(u), int(v)) for u, v in edges} | {(int(v), int(u)) for u, v in edges}      # to_undirected keeps self-loops (:574)
    assert torch.equal(blob['edge_index'], torch.tensor(sorted(und), dtype=torch.int64).t())
    assert all(sp_idx[k].tolist() == split[k].tolist() for k in split)
    d = cb_data.load_data('ogbn-arxiv', 0, _fake_trainer(), root=str(tmp_path))
    assert 'OGB raw files' in capsys.readouterr().out
    assert sorted(d.train_mask.nonzero().flatten().tolist()) == sorted(split['train'].tolist())
    assert sorted(d.test_mask.nonzero().flatten().tolist()) == sorted(split['test'].tolist())
    assert torch.equal(d.edge_index, blob['edge_index'])            # no self-loops added, no normalisation for the ogbn family


def test_load_data_falls_back_to_the_synthetic_stand_in(tmp_path, capsys):
    d = cb_data.load_data('Pubmed', 0, _fake_trainer(), root=str(tmp_path))
    assert 'synthetic stand-in' in capsys.readouterr().out and d.x.shape == (19717, 500)
