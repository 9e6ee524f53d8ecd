"""Normalisation tricks of GNN_model/norm_tricks.py on hand-written reductions (csrc/cb_elementwise.hip):
node_norm (row-wise), mean_norm / pair_norm / BatchNorm1d (column statistics), each with its backward.
Only [d]-sized vectors are touched by torch arithmetic; every pass over an [N, d] matrix is a HIP kernel."""
import contextlib

import torch

from . import _lib

# Node-sharded runs (dist.py): rows of x are spread over the ranks of `_SHARD[0]`; the column statistics below are then
# all-reduced and the row count is the global one, so every rank normalises with the statistics of the WHOLE node set
# (norm_tricks.py:25-41,106,132 see all rows).  None = single device.
_SHARD = None


@contextlib.contextmanager
def row_sharding(group, n_global):
    """Column statistics inside this context span all ranks of `group` (ShardedTrainer.train_step)."""
    global _SHARD
    import torch.distributed as dist
    prev = _SHARD
    _SHARD = (group, int(n_global)) if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 else None
    try:
        yield
    finally:
        _SHARD = prev


def _rows(x):
    return _SHARD[1] if _SHARD is not None else x.shape[0]


def _global(*vecs):
    """Sum of the per-rank statistic vectors over the ranks (one small all-reduce for all of them)."""
    if _SHARD is None:
        return vecs
    from .dist import _all_reduce
    flat = torch.cat([v.reshape(-1) for v in vecs])
    _all_reduce(flat, group=_SHARD[0])
    out, off = [], 0
    for v in vecs:
        out.append(flat[off:off + v.numel()].view_as(v))
        off += v.numel()
    return tuple(out)


_NODE = {'n': (1.0, 1.0), 'v': (0.0, 1.0), 'm': (1.0, 0.0), 'srv': (0.0, 0.5), 'pr': (0.0, None)}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# The three passes over [N, d] data every column norm is made of.  PRIMS = None: the HIP kernels below (the product has no
# other implementation); the CPU tests of the cross-rank statistics (tests/test_dist_gloo.py) install an oracle-backed
# object with the same three methods, exactly as they do for dist.HipCompute.
PRIMS = None


def _colstats(x, w=None):
    if PRIMS is not None:
        return PRIMS.colstats(x, w)
    lib = _lib.load()
    rows, d = x.shape
    s1 = torch.empty(d, dtype=torch.float32, device=x.device)
    s2 = torch.empty(d, dtype=torch.float32, device=x.device)
    wsb = lib.cb_colstats_workspace_bytes(max(rows, 1), d)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.cb_colstats_f32(_lib.ptr(x), _lib.ptr(w), rows, d, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(ws), wsb,
                                       _lib.stream_ptr()), 'cb_colstats_f32')
    return s1, s2


def _affine(x, shift=None, scale=None, bias=None, gscale=1.0):
    if PRIMS is not None:
        return PRIMS.affine(x, shift, scale, bias, gscale)
    lib = _lib.load()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.cb_col_affine_f32(_lib.ptr(x), _lib.ptr(shift), _lib.ptr(scale), _lib.ptr(bias), float(gscale), _lib.ptr(y),
                                         x.shape[0], x.shape[1], _lib.stream_ptr()), 'cb_col_affine_f32')
    return y


def _combine(g, xh=None, a=None, b=None, e=None):
    if PRIMS is not None:
        return PRIMS.combine(g, xh, a, b, e)
    lib = _lib.load()
    dx = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_col_bwd_combine_f32(_lib.ptr(g), _lib.ptr(xh), _lib.ptr(a), _lib.ptr(b), _lib.ptr(e), 1.0, 1.0, _lib.ptr(dx),
                                              g.shape[0], g.shape[1], _lib.stream_ptr()), 'cb_col_bwd_combine_f32')
    return dx


class _NodeNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, q, eps):
        lib = _lib.load()
        x = _c(x)
        y = torch.empty_like(x)
        stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.cb_node_norm_fwd_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(stats), x.shape[0], x.shape[1], c, q, eps,
                                                _lib.stream_ptr()), 'cb_node_norm_fwd_f32')
        ctx.c, ctx.q = c, q
        ctx.save_for_backward(x, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, stats = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.cb_node_norm_bwd_f32(_lib.ptr(x), _lib.ptr(g), _lib.ptr(stats), _lib.ptr(dx), x.shape[0], x.shape[1],
                                                ctx.c, ctx.q, _lib.stream_ptr()), 'cb_node_norm_bwd_f32')
        return dx, None, None, None


def node_norm(x, kind='n', eps=1e-5, power=0.5):
    if kind not in _NODE:
        return x
    c, q = _NODE[kind]
    if q is None:
        q = power
    if q not in (0.0, 0.5, 1.0):
        raise NotImplementedError('node_norm pr: only power_root = 2 (the reference default) is built')
    return _NodeNormFn.apply(x, float(c), float(q), float(eps))


class _MeanNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        (s1,) = _global(_colstats(x)[0])
        ctx.n = _rows(x)
        return _affine(x, shift=s1 / ctx.n)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        (s1,) = _global(_colstats(g)[0])
        return _combine(g, e=-s1 / ctx.n)


def mean_norm(x):
    return _MeanNormFn.apply(x)


class _PairNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n = _rows(x)
        s1, s2 = _global(*_colstats(x))
        mu = s1 / n
        r = torch.sqrt(1e-6 + ((s2 - n * mu * mu).sum() / n))          # sqrt(1e-6 + mean_rows sum_c (x - mu)^2)
        inv = (1.0 / r).expand(x.shape[1]).contiguous()
        y = _affine(x, shift=mu, scale=inv)
        ctx.save_for_backward(y, r)
        ctx.n = n
        return y

    @staticmethod
    def backward(ctx, g):
        y, r = ctx.saved_tensors
        g = _c(g)
        n, d = ctx.n, g.shape[1]
        sg, sgy = _global(*_colstats(g, w=y))
        kappa = sgy.sum() / (n * r)
        a = (1.0 / r).expand(d).contiguous()
        b = (-kappa).expand(d).contiguous()
        return _combine(g, xh=y, a=a, b=b, e=-sg / (n * r))


def pair_norm(x):
    return _PairNormFn.apply(x)


class _BatchNormTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = _c(x)
        n = _rows(x)
        s1, s2 = _global(*_colstats(x))
        mu = s1 / n
        var = (s2 / n - mu * mu).clamp_(min=0)
        rstd = torch.rsqrt(var + eps)
        scale = rstd * weight if weight is not None else rstd
        y = _affine(x, shift=mu, scale=scale, bias=bias)
        ctx.save_for_backward(x, weight, mu, rstd)
        ctx.has_bias, ctx.n = bias is not None, n
        ctx.mark_non_differentiable(mu, var)
        return y, mu, var

    @staticmethod
    def backward(ctx, g, _gmu, _gvar):
        x, weight, mu, rstd = ctx.saved_tensors
        g = _c(g)
        n = ctx.n
        sg_l, sgx_l = _colstats(g, w=x)                  # this rank's rows
        sg, sgx = _global(sg_l, sgx_l)                   # all rows
        dgamma = rstd * (sgx - mu * sg)                  # sum_r g * xhat
        gam = weight if weight is not None else torch.ones_like(rstd)
        a = gam * rstd                                   # dx = a*g + b*xhat + e, with xhat = (x - mu) * rstd
        b_hat = -a * dgamma / n
        e = -a * sg / n - b_hat * rstd * mu
        dx = _combine(g, xh=x, a=a, b=b_hat * rstd, e=e)
        # the affine parameters are replicated and their gradients are summed over the ranks afterwards (dist.allreduce_grads):
        # hand back this rank's share (linear in the sums, so the shares add up to dgamma / sg)
        return dx, (rstd * (sgx_l - mu * sg_l) if weight is not None else None), (sg_l if ctx.has_bias else None), None


def batch_norm(layer, x):
    """torch.nn.BatchNorm1d semantics (affine, running statistics, momentum, unbiased running_var) on the HIP kernels."""
    if layer.training or not layer.track_running_stats:
        y, mu, var = _BatchNormTrainFn.apply(x, layer.weight, layer.bias, layer.eps)
        if layer.training and layer.track_running_stats:
            with torch.no_grad():
                n = _rows(x)
                layer.num_batches_tracked += 1
                m = layer.momentum if layer.momentum is not None else 1.0 / float(layer.num_batches_tracked)
                layer.running_mean.mul_(1 - m).add_(mu, alpha=m)
                layer.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
        return y
    scale = torch.rsqrt(layer.running_var + layer.eps)
    if layer.weight is not None:
        scale = scale * layer.weight
    return _AffineEvalFn.apply(_c(x), layer.running_mean, scale, layer.bias)


class _AffineEvalFn(torch.autograd.Function):
    """y = (x - shift) * scale + bias with frozen statistics; differentiable in x, scale and bias (torch.nn.BatchNorm1d in eval
    mode still trains its affine parameters: scale = weight * rsqrt(running_var + eps) is built by torch ops outside)."""

    @staticmethod
    def forward(ctx, x, shift, scale, bias):
        ctx.save_for_backward(x, shift, scale)
        ctx.has_bias = bias is not None
        return _affine(x, shift=shift, scale=scale.contiguous(), bias=bias)

    @staticmethod
    def backward(ctx, g):
        x, shift, scale = ctx.saved_tensors
        g = _c(g)
        dscale = dbias = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            sg, sgx = _colstats(g, w=x)                  # sum_r g, sum_r g * x  (this rank's rows: replicated-parameter share)
            dscale = sgx - shift * sg if ctx.needs_input_grad[2] else None
            dbias = sg if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        dx = _combine(g, a=scale.contiguous()) if ctx.needs_input_grad[0] else None
        return dx, None, dscale, dbias
