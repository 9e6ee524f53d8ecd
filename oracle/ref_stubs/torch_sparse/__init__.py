"""TEST INFRASTRUCTURE ONLY — stand-in for the absent torch_sparse==0.6.10 (requirements.txt:101), just enough of
`SparseTensor` for Label_propagation_model/outcome_correlation.py:39-55,128-145 (process_adj, gen_normalized_adjs,
general_outcome_correlation): construction from (row, col[, value]), `.sum(dim=1)`, broadcasting `*` with a
column / row vector, `@` with a dense matrix, `.to(device)`.  PARITY UNPINNED at this boundary (published
semantics restated: entries are kept as given, missing values count as 1)."""
import torch


class SparseTensor:
    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, is_sorted=False, **_):
        self.row, self.col = row.to(torch.int64), col.to(torch.int64)
        self.value = value
        self.sizes = tuple(int(s) for s in sparse_sizes)

    def _val(self):
        return self.value if self.value is not None else torch.ones(self.row.numel(), dtype=torch.float32, device=self.row.device)

    def has_value(self):
        return self.value is not None

    def to(self, device):
        return SparseTensor(self.row.to(device), self.col.to(device), None if self.value is None else self.value.to(device), self.sizes)

    def sum(self, dim):
        idx = self.row if dim == 1 else self.col
        out = torch.zeros(self.sizes[0 if dim == 1 else 1], dtype=self._val().dtype, device=idx.device)
        return out.index_add(0, idx, self._val())

    def _scale(self, other):
        other = torch.as_tensor(other)
        v = self._val()
        if other.dim() == 2 and other.shape[1] == 1:        # column vector: scales rows
            v = v * other[self.row, 0]
        elif other.dim() == 2 and other.shape[0] == 1:      # row vector: scales columns
            v = v * other[0, self.col]
        else:
            raise NotImplementedError('SparseTensor stand-in: only [N,1] / [1,N] broadcasts')
        return SparseTensor(self.row, self.col, v, self.sizes)

    __mul__ = _scale
    __rmul__ = _scale

    def __matmul__(self, dense):
        out = torch.zeros((self.sizes[0],) + tuple(dense.shape[1:]), dtype=dense.dtype, device=dense.device)
        return out.index_add(0, self.row, dense[self.col] * self._val().to(dense.dtype).unsqueeze(1))


def matmul(a, b):
    return a @ b
