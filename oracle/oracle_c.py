"""TEST INFRASTRUCTURE ONLY — ctypes wrapper of oracle/coldbrew_oracle.c (CPU, OpenMP) plus a
differentiable aggregation built on it, used by the CPU tests and by bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libcoldbrew_oracle.so')
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO) and build:
            subprocess.check_call(['make', '-C', _HERE], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(_SO)
        P, I64 = ctypes.c_void_p, ctypes.c_int64
        lib.orc_csr_from_coo.restype = ctypes.c_int
        lib.orc_csr_from_coo.argtypes = [P, P, I64, I64, P, P]
        lib.orc_deg_norm.restype = None
        lib.orc_deg_norm.argtypes = [P, I64, P]
        lib.orc_spmm_csr.restype = None
        lib.orc_spmm_csr.argtypes = [P, P, I64, P, I64, P, P, ctypes.c_int, P]
        lib.orc_set_num_threads.restype = None
        lib.orc_set_num_threads.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def set_num_threads(n):
    load().orc_set_num_threads(int(n))


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def csr_from_coo(major, minor, n):
    lib = load()
    major = np.ascontiguousarray(major, dtype=np.int64)
    minor = np.ascontiguousarray(minor, dtype=np.int64)
    rowptr = np.empty(n + 1, dtype=np.int64)
    col = np.empty(max(len(major), 1), dtype=np.int32)
    rc = lib.orc_csr_from_coo(_p(major), _p(minor), len(major), n, _p(rowptr), _p(col))
    if rc:
        raise ValueError(f'orc_csr_from_coo failed ({rc})')
    return rowptr, col[:len(major)]


def deg_norm(rowptr):
    lib = load()
    out = np.empty(len(rowptr) - 1, dtype=np.float32)
    lib.orc_deg_norm(_p(rowptr), len(rowptr) - 1, _p(out))
    return out


def spmm(rowptr, col, h, scale=None, bias=None, relu=False):
    lib = load()
    h = np.ascontiguousarray(h, dtype=np.float32)
    n = len(rowptr) - 1
    out = np.empty((n, h.shape[1]), dtype=np.float32)
    lib.orc_spmm_csr(_p(rowptr), _p(col), n, _p(h), h.shape[1], _p(scale), _p(bias), int(relu), _p(out))
    return out


class _AggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, csr):
        ctx.csr = csr
        return torch.from_numpy(spmm(csr.rowptr, csr.col, h.detach().numpy()))

    @staticmethod
    def backward(ctx, g):
        c = ctx.csr
        return torch.from_numpy(spmm(c.rowptr_t, c.col_t, g.contiguous().numpy())), None


def aggregate_sum(csr, h):
    """Differentiable A^T.h on the C/OpenMP SpMM (fp32) — drop-in for coldbrew_oracle.aggregate_sum at scale."""
    return _AggFn.apply(h, csr)
