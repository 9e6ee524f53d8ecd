"""TEST INFRASTRUCTURE ONLY — stub for the absent torch_scatter (drop_tricks.py:5)."""
import torch


def scatter_add(src, index, dim=0, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add(0, index, src)
