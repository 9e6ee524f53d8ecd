"""Normalisation tricks of GNN_model/norm_tricks.py on hand-written reductions (csrc/cb_elementwise.hip):
node_norm (row-wise), mean_norm / pair_norm / BatchNorm1d (column statistics), each with its backward.
Only [d]-sized vectors are touched by torch arithmetic; every pass over an [N, d] matrix is a HIP kernel."""
import torch

from . import _lib

_NODE = {'n': (1.0, 1.0), 'v': (0.0, 1.0), 'm': (1.0, 0.0), 'srv': (0.0, 0.5), 'pr': (0.0, None)}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _colstats(x, w=None):
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
    lib = _lib.load()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.cb_col_affine_f32(_lib.ptr(x), _lib.ptr(shift), _lib.ptr(scale), _lib.ptr(bias), float(gscale), _lib.ptr(y),
                                         x.shape[0], x.shape[1], _lib.stream_ptr()), 'cb_col_affine_f32')
    return y


def _combine(g, xh=None, a=None, b=None, e=None):
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
        s1, _ = _colstats(x)
        ctx.n = x.shape[0]
        return _affine(x, shift=s1 / x.shape[0])

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        s1, _ = _colstats(g)
        return _combine(g, e=-s1 / ctx.n)


def mean_norm(x):
    return _MeanNormFn.apply(x)


class _PairNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n = x.shape[0]
        s1, s2 = _colstats(x)
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
        sg, sgy = _colstats(g, w=y)
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
        n = x.shape[0]
        s1, s2 = _colstats(x)
        mu = s1 / n
        var = (s2 / n - mu * mu).clamp_(min=0)
        rstd = torch.rsqrt(var + eps)
        scale = rstd * weight if weight is not None else rstd
        y = _affine(x, shift=mu, scale=scale, bias=bias)
        ctx.save_for_backward(x, weight, mu, rstd)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(mu, var)
        return y, mu, var

    @staticmethod
    def backward(ctx, g, _gmu, _gvar):
        x, weight, mu, rstd = ctx.saved_tensors
        g = _c(g)
        n = x.shape[0]
        sg, sgx = _colstats(g, w=x)
        dgamma = rstd * (sgx - mu * sg)                  # sum_r g * xhat
        gam = weight if weight is not None else torch.ones_like(rstd)
        a = gam * rstd                                   # dx = a*g + b*xhat + e, with xhat = (x - mu) * rstd
        b_hat = -a * dgamma / n
        e = -a * sg / n - b_hat * rstd * mu
        dx = _combine(g, xh=x, a=a, b=b_hat * rstd, e=e)
        return dx, (dgamma if weight is not None else None), (sg if ctx.has_bias else None), None


def batch_norm(layer, x):
    """torch.nn.BatchNorm1d semantics (affine, running statistics, momentum, unbiased running_var) on the HIP kernels."""
    if layer.training or not layer.track_running_stats:
        y, mu, var = _BatchNormTrainFn.apply(x, layer.weight, layer.bias, layer.eps)
        if layer.training and layer.track_running_stats:
            with torch.no_grad():
                n = x.shape[0]
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
    @staticmethod
    def forward(ctx, x, shift, scale, bias):
        ctx.save_for_backward(scale)
        return _affine(x, shift=shift, scale=scale.contiguous(), bias=bias)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return _combine(_c(g), a=scale.contiguous()), None, None, None
