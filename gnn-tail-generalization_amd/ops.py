"""Autograd-visible operators of the TeacherGNN hot path, each backed by the C ABI
(include/coldbrew_hip.h).  One torch.autograd.Function per fused stage; tensors must live
on the MI355X (no CPU fallback — _lib.require_device raises otherwise).

Stage map (reference lines are GNN_model/GCN.py):
  transform   Z = (X * a) @ W + E            :213,:225,:230-231   (dense, MFMA-bound)
  aggregate   Y = act(b * (A^T Z) + bias)    :238,:250,:253(+:128) (sparse, HBM-bound)
"""
import torch

from . import _lib


class _AggregateFn(torch.autograd.Function):
    """Forward: by-dst CSR SpMM with fused `* norm_in`, `+ bias`, optional ReLU.
    Backward (autograd of DGL's gspmm = SpMM on the reverse graph):
        dY' = dY * (Y > 0) if relu;  dbias = colsum(dY');  dZ = A (b * dY')."""

    @staticmethod
    def forward(ctx, graph, h, row_scale, bias, relu):
        out = graph.spmm(h, transpose=False, row_scale=row_scale, bias=bias, relu=relu)
        ctx.graph, ctx.relu = graph, relu
        ctx.has_bias = bias is not None
        ctx.save_for_backward(out if relu else None, row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        out, row_scale = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            g = g * (out > 0)
        dbias = g.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        dh = None
        if ctx.needs_input_grad[1]:
            gs = g * row_scale.unsqueeze(1) if row_scale is not None else g
            dh = ctx.graph.spmm(gs, transpose=True)
        return None, dh, None, dbias, None


def aggregate(graph, h, row_scale=None, bias=None, relu=False):
    _lib.require_device(h)
    return _AggregateFn.apply(graph, h, row_scale, bias, bool(relu))


def transform(feat, norm_out, weight, le=None):
    """Z = (feat * a[:,None]) @ W (+ le) and se_reg = ||le||_F (GCN.py:213,225,230-236)."""
    _lib.require_device(feat, weight)
    z = torch.matmul(feat * norm_out.unsqueeze(1), weight)
    if le is not None:
        return z + le, torch.norm(le)
    return z, None
