"""Readers for the RAW on-disk formats of the datasets the reference loads through PyG / OGB (trainer_node_classification.py:570-577,
616-670): neither package is installed here and nothing can be downloaded, but a user who has run the reference once has the raw files
under `data/`, exactly where `Planetoid(path, dataset)` / `PygNodePropPredDataset(name, root='data')` left them.  These functions read those
files directly and reproduce what the reference's loaders hand to the trainer:

    data/<Dataset>/<Dataset>/raw/ind.<dataset>.{x,tx,allx,y,ty,ally,graph,test.index}      Planetoid (Cora / Citeseer / Pubmed)
    data/ogbn_arxiv/{raw/{edge,node-feat,node-label}.csv.gz, split/time/{train,valid,test}.csv.gz}
    data/ogbn_products/{raw/..., split/sales_ranking/...}

The Planetoid format is the public one of Yang et al. / Kipf & Welling's `gcn` repository (pickled scipy CSR matrices, one-hot label
arrays, an adjacency dict, a text file of test indices); the assembly below follows its published description — torch_geometric 1.7.2's
`read_planetoid_data` (the reference pins that version, requirements.txt:99) does the same steps: test rows re-ordered to their indices,
Citeseer's isolated test nodes as zero rows, labels = argmax, 'public' split = first len(y) rows train / next 500 validation / test
indices test, adjacency dict -> edge list without self-loops, coalesced; then `T.NormalizeFeatures()` (rows divided by their sum, sums
clamped at 1).  The unpickling executes what the files contain, as PyG's does: point it at files you trust.

What the trainer's own post-processing then does (ensure_symmetric, self-loops, the Cora 600-node split, masks) is in data.load_data.
"""
import gzip
import os
import pickle

import numpy as np
import torch

from .utils import to_undirected


def planetoid_raw_dir(root, dataset):
    """Where `Planetoid(os.path.join(root, dataset), dataset)` keeps its raw files (trainer_node_classification.py:628,631), or None."""
    for d in (os.path.join(root, dataset, dataset, 'raw'), os.path.join(root, dataset, 'raw'), os.path.join(root, dataset)):
        if os.path.isfile(os.path.join(d, f'ind.{dataset.lower()}.x')):
            return d
    return None


def _unpickle(path):
    with open(path, 'rb') as f:
        out = pickle.load(f, encoding='latin1')      # (written by Python 2)
    if hasattr(out, 'todense'):                       # scipy sparse matrix
        out = np.asarray(out.todense())
    return out


def read_planetoid(raw_dir, dataset):
    """dict(x float32 [N, F] row-normalised, y int64 [N], edge_index int64 [2, E] (no self-loops, coalesced, both directions as the files
    list them), train_mask / val_mask / test_mask bool [N]) of the 'public' split."""
    name = dataset.lower()
    part = {k: _unpickle(os.path.join(raw_dir, f'ind.{name}.{k}')) for k in ('x', 'tx', 'allx', 'y', 'ty', 'ally', 'graph')}
    with open(os.path.join(raw_dir, f'ind.{name}.test.index')) as f:
        test_index = torch.tensor([int(line) for line in f.read().split()], dtype=torch.int64)
    tx, allx = torch.from_numpy(np.asarray(part['tx'], dtype=np.float32)), torch.from_numpy(np.asarray(part['allx'], dtype=np.float32))
    ty, ally = torch.from_numpy(np.asarray(part['ty'], dtype=np.float32)), torch.from_numpy(np.asarray(part['ally'], dtype=np.float32))
    n_train = int(np.asarray(part['y']).shape[0])
    sorted_test = torch.sort(test_index)[0]
    if name == 'citeseer':
        # some test nodes of Citeseer are isolated and absent from tx: they become zero rows (and class 0)
        lo, span = int(test_index.min()), int(test_index.max() - test_index.min()) + 1
        tx_ext, ty_ext = torch.zeros(span, tx.shape[1]), torch.zeros(span, ty.shape[1])
        tx_ext[sorted_test - lo] = tx
        ty_ext[sorted_test - lo] = ty
        tx, ty = tx_ext, ty_ext
    x = torch.cat([allx, tx], 0)
    y = torch.cat([ally, ty], 0).argmax(dim=1)
    x[test_index] = x[sorted_test]
    y[test_index] = y[sorted_test]
    n = int(y.shape[0])
    rows, cols = [], []
    for key, nbrs in part['graph'].items():
        rows += [int(key)] * len(nbrs)
        cols += [int(v) for v in nbrs]
    ei = torch.tensor([rows, cols], dtype=torch.int64)
    ei = ei[:, ei[0] != ei[1]]
    key = torch.unique(ei[0] * n + ei[1])                     # coalesce: duplicates out, sorted by (row, col)
    ei = torch.stack([key // n, key % n])
    x = x / x.sum(1, keepdim=True).clamp(min=1.0)             # T.NormalizeFeatures()
    mask = lambda idx: torch.zeros(n, dtype=torch.bool).index_fill_(0, idx, True)
    return dict(x=x, y=y, edge_index=ei, train_mask=mask(torch.arange(n_train)), val_mask=mask(torch.arange(n_train, n_train + 500).clamp(max=n - 1)),
                test_mask=mask(test_index))


def ogb_dir(root, dataset):
    """Where `PygNodePropPredDataset(name=dataset, root=root)` keeps the dataset (trainer_node_classification.py:571), or None."""
    d = os.path.join(root, dataset.replace('-', '_'))
    return d if os.path.isfile(os.path.join(d, 'raw', 'edge.csv.gz')) else None


def _csv(path, dtype):
    import pandas as pd
    with gzip.open(path, 'rt') as f:
        return pd.read_csv(f, header=None).to_numpy(dtype=dtype)


def read_ogbn(d, device='cpu'):
    """(dict(x float32 [N, F], y int64 [N], edge_index int64 [2, E] made undirected as load_ogbn does (:574-575)), split_idx dict of
    int64 index tensors) from OGB's raw csv files; the undirected edge list is built on `device` (utils.to_undirected: the device kernel
    for 1.2 * 10^8 product edges)."""
    raw = os.path.join(d, 'raw')
    ei = torch.from_numpy(_csv(os.path.join(raw, 'edge.csv.gz'), np.int64).T.copy())
    x = torch.from_numpy(_csv(os.path.join(raw, 'node-feat.csv.gz'), np.float32))
    y = torch.from_numpy(_csv(os.path.join(raw, 'node-label.csv.gz'), np.int64)).reshape(-1)
    n = int(x.shape[0])
    ei = to_undirected(ei.to(device), n)
    split_dir = None
    for cand in sorted(os.listdir(os.path.join(d, 'split'))) if os.path.isdir(os.path.join(d, 'split')) else []:
        if os.path.isfile(os.path.join(d, 'split', cand, 'train.csv.gz')):
            split_dir = os.path.join(d, 'split', cand)      # 'time' (arxiv), 'sales_ranking' (products)
            break
    if split_dir is None:
        raise FileNotFoundError(f'{d}: no split/<scheme>/train.csv.gz')
    split = {k: torch.from_numpy(_csv(os.path.join(split_dir, f'{k}.csv.gz'), np.int64)).reshape(-1) for k in ('train', 'valid', 'test')}
    return dict(x=x, y=y, edge_index=ei), split
