"""TEST INFRASTRUCTURE ONLY — stand-in for the absent third-party `dgl==0.7.0`.

The reference's aggregation arithmetic lives in DGL (requirements.txt:19), which is
not installed in this image and cannot be installed (no network).  This module gives
the *unmodified* reference files (GNN_model/GCN.py:4,12-14,94,186-206,238-243) just
enough of DGL's surface to import and run on CPU so golden vectors can be generated
in the build container.  It never travels into the product path.

Semantics restated from DGL's published API:
  * dgl.graph((src, dst))          -> homogeneous multigraph, num_nodes = max id + 1
  * g.in_degrees()/out_degrees()   -> int64 counts (duplicates counted)
  * g.update_all(copy_src('h','m'), sum('m','h'))
                                   -> dstdata['h'][v] = sum_{e:(u->v)} srcdata['h'][u]
  * g.update_all(u_mul_e('h','w','m'), sum('m','h')) -> same with per-edge weights
PARITY UNPINNED at this boundary: no reference test pins DGL's result; the stand-in is
cross-checked against a dense fp64 A^T.h in tests/test_oracle_golden.py.
"""
import contextlib

import torch

from . import function  # noqa: F401
from . import base  # noqa: F401
from . import utils  # noqa: F401


class _Graph:
    def __init__(self, src, dst, num_nodes=None):
        self._src = torch.as_tensor(src, dtype=torch.int64).reshape(-1)
        self._dst = torch.as_tensor(dst, dtype=torch.int64).reshape(-1)
        if num_nodes is None:
            num_nodes = int(max(self._src.max().item(), self._dst.max().item())) + 1 if self._src.numel() else 0
        self._n = num_nodes
        self.srcdata = {}
        self.dstdata = self.srcdata  # homogeneous graph: one node frame
        self.ndata = self.srcdata
        self.edata = {}

    def to(self, device):
        self._src = self._src.to(device)
        self._dst = self._dst.to(device)
        return self

    @contextlib.contextmanager
    def local_scope(self):
        saved_n, saved_e = dict(self.srcdata), dict(self.edata)
        try:
            yield
        finally:
            self.srcdata.clear()
            self.srcdata.update(saved_n)
            self.edata.clear()
            self.edata.update(saved_e)

    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self._src.numel())

    num_edges = number_of_edges

    def in_degrees(self):
        return torch.bincount(self._dst, minlength=self._n)

    def out_degrees(self):
        return torch.bincount(self._src, minlength=self._n)

    def is_block(self):
        return False

    def update_all(self, message_func, reduce_func):
        kind = message_func[0]
        assert reduce_func[0] == 'sum'
        if kind == 'copy_u':
            _, u_field, m_field = message_func
            msg = self.srcdata[u_field].index_select(0, self._src)
        elif kind == 'u_mul_e':
            _, u_field, e_field, m_field = message_func
            w = self.edata[e_field]
            msg = self.srcdata[u_field].index_select(0, self._src)
            msg = msg * w.reshape((-1,) + (1,) * (msg.dim() - 1))
        else:
            raise NotImplementedError(kind)
        assert reduce_func[1] == m_field
        out = torch.zeros((self._n,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out = out.index_add(0, self._dst, msg)
        self.dstdata[reduce_func[2]] = out


def graph(data, num_nodes=None, **_):
    src, dst = data
    return _Graph(src, dst, num_nodes)


DGLGraph = _Graph
