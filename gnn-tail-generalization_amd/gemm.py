"""Dense stages of the path on the hand-written MFMA kernels (csrc/cb_gemm_limb.hip: fp32 operands as three exact bf16
limbs on the bf16 matrix cores, fp32 accumulate; csrc/cb_gemm.hip: fp32-input MFMA fallback) with their
autograd.  Raw entry points: mm_nn / mm_tn; autograd stages: linear_rowscale (GCNConv transform,
GNN_model/GCN.py:213,225,231) and linear (nn.Linear + optional ReLU, GCN.py:105-106,138)."""
import torch

from . import _lib


def _rowmajor(t):
    """A 2-D tensor whose rows are contiguous (stride(1) == 1); copies otherwise."""
    if t.dim() != 2:
        raise ValueError('expected a matrix')
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    elif t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _b_workspace(lib, N, K, M, device):
    """Workspace of an NN launch: split-K partial planes for small-M / long-K shapes on the fp32-input fallback kernel
    (cb_gemm_nn_splitk_workspace_bytes: x @ W_0 of a Cora-sized graph); nothing otherwise."""
    wsb = lib.cb_gemm_nn_splitk_workspace_bytes(M, N, K) if K >= 512 else 0
    return (torch.empty(wsb, dtype=torch.uint8, device=device), wsb) if wsb else (None, 0)


def mm_nn(a, b, rowscale=None, addend=None, bias=None, relu=False, out_bf16=False, out=None):
    """act(rowscale[:,None] * (a @ b) + addend + bias) in one kernel; a [M,K], b [K,N] float32 on device.
    out_bf16: store the result as bfloat16 (round-to-nearest-even) — the bf16 aggregation variant.
    out: optional [M, N] destination with contiguous rows (a row chunk of a larger matrix: the chunked layer GEMM of the
    node-sharded pipeline, dist.py)."""
    lib = _lib.load()
    _lib.require_device(a, b, rowscale, addend, bias)
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = a.shape
    K2, N = b.shape
    if K != K2:
        raise ValueError(f'shape mismatch: {tuple(a.shape)} @ {tuple(b.shape)}')
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError('mm_nn expects float32')
    if addend is not None:
        addend = _rowmajor(addend)
    odt = torch.bfloat16 if out_bf16 else torch.float32
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    elif tuple(out.shape) != (M, N) or out.dtype != odt or (N > 1 and out.stride(1) != 1) or out.device != a.device:
        raise ValueError(f'mm_nn: out must be a [{M}, {N}] {odt} matrix with contiguous rows on the operands\' device')
    fn = lib.cb_gemm_nn_bf16out_f32 if out_bf16 else lib.cb_gemm_nn_f32
    ws, wsb = _b_workspace(lib, N, K, M, a.device)
    with torch.cuda.device(a.device):
        _lib.check(fn(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(out), _ld(out), M, N, K, _lib.ptr(rowscale),
                      _lib.ptr(addend), _ld(addend) if addend is not None else 0, _lib.ptr(bias),
                      int(bool(relu)), _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_gemm_nn')
    return out


def mm_nn_store_rows(a, b, rowscale, addend, bias, row_index, mix, mix_index, c_act, c_mix, p, seed, row0, bits, relu_only, want_act=False):
    """The trunk's store on a subset of the node rows as the epilogue of the transform in front of it (cb_gemm_nn_store_rows_f32):
    act = relu(rowscale * (a @ b) + addend + bias);  out = dropout(c_act * act + c_mix * mix[mix_index | row_index]) — mask words (bits: the full
    [N, 1, 4] array) and dropout mask at the node rows row_index.  Returns (out, act | None), or None where the fused form does not exist for the
    shape (the caller then runs mm_nn and the elementwise pass) or CB_GEMM_STORE_ROWS=0."""
    import ctypes
    import os
    from . import ops
    lib = _lib.load()
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = a.shape
    N = b.shape[1]
    if os.environ.get('CB_GEMM_STORE_ROWS', '1') == '0' or N != 256 or M == 0:
        return None
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if not lib.cb_gemm_nn_store_rows_supported(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(out), _ld(out), M, N, K):
        return None
    if addend is not None:
        addend = _rowmajor(addend)
    act = torch.empty_like(out) if want_act else None
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_nn_store_rows_f32(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(out), _ld(out), M, N, K, _lib.ptr(rowscale), _lib.ptr(addend),
                                                 _ld(addend) if addend is not None else 0, _lib.ptr(bias), _lib.ptr(row_index), _lib.ptr(mix),
                                                 mix.stride(0) if mix is not None else 0, _lib.ptr(mix_index), float(c_act), float(c_mix), float(p),
                                                 ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(bits), int(bool(relu_only)), _lib.ptr(act), N, None, 0,
                                                 _lib.stream_ptr()), 'cb_gemm_nn_store_rows_f32')
    return out, act


def mm_nn_drop2(a, b, p, seed, row0=0, bias=None, relu=False):
    """(y, dropout_p(y)) with y = act(a @ b + bias), both written by one GEMM epilogue (cb_gemm_nn_drop2_f32); the dropped copy
    uses the keep-mask ops._dropout_raw(y, p, seed, row0 * N) would draw."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(a, b, bias)
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = a.shape
    K2, N = b.shape
    if K != K2 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError(f'mm_nn_drop2: bad operands {tuple(a.shape)} @ {tuple(b.shape)}')
    y = torch.empty((M, N), dtype=torch.float32, device=a.device)
    yd = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws, wsb = _b_workspace(lib, N, K, M, a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_nn_drop2_f32(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(y), N, _lib.ptr(yd), N, M, N, K, None, None, 0,
                                            _lib.ptr(bias), int(bool(relu)), float(p), ctypes.c_uint64(seed), ops.seed_dev_ptr(), int(row0),
                                            _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_gemm_nn_drop2_f32')
    return y, yd


def mm_nn_indrop_drop2(a, b, p, a_seed, seed, row0=0, bias=None, relu=False, want_bits=False):
    """(y, dropout_seed(y)) with y = act(dropout_{a_seed}(a) @ b + bias): the dropout in front of the input Linear (GCN.py:104) is
    applied to `a` while the GEMM stages it (cb_gemm_nn_indrop_drop2_f32) — bit-identical to ops._dropout_raw(a, p, a_seed, row0 * K)
    followed by mm_nn_drop2.  Returns None where the fused form does not exist for the shape (the caller keeps the two-kernel form).
    want_bits (N == 256, relu): a third result, the int64 [M, 1, 4] mask words of (y > 0) in the layout of the aggregation's fused store."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(a, b, bias)
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = a.shape
    K2, N = b.shape
    if K != K2 or a.dtype != torch.float32 or b.dtype != torch.float32 or not (0.0 < p < 1.0):
        return None
    y = torch.empty((M, N), dtype=torch.float32, device=a.device)
    yd = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if not lib.cb_gemm_nn_indrop_supported(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(y), N, _lib.ptr(yd), N, M, N, K):
        return None
    bits = torch.empty((M, 1, 4), dtype=torch.int64, device=a.device) if (want_bits and N == 256 and relu) else None
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_nn_indrop_drop2_f32(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(y), N, _lib.ptr(yd), N, M, N, K, _lib.ptr(bias),
                                                   int(bool(relu)), float(p), ctypes.c_uint64(a_seed), float(p), ctypes.c_uint64(seed),
                                                   ops.seed_dev_ptr(), int(row0), _lib.ptr(bits), _lib.stream_ptr()), 'cb_gemm_nn_indrop_drop2_f32')
    return (y, yd, bits) if want_bits else (y, yd)


def mm_nn_indrop(a, b, p, a_seed, row0=0, rowscale=None, addend=None, bias=None, relu=False, want_bits=False, out=None):
    """act(rowscale * (dropout_{a_seed}(a) @ b) + addend + bias) with the dropout applied to `a` while the GEMM stages it
    (cb_gemm_nn_indrop_f32): no dropped copy of `a` exists — bit-identical to ops._dropout_raw(a, p, a_seed, row0 * K) followed by mm_nn.
    Returns y, or (y, bits) with want_bits (N == 256, relu: the int64 [M, 1, 4] mask words of y > 0); None where the fused form does not
    exist for the shape (few rows, narrow outputs, misaligned operands)."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(a, b, rowscale, addend, bias, out)
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = a.shape
    K2, N = b.shape
    if K != K2 or a.dtype != torch.float32 or b.dtype != torch.float32 or not (0.0 < p < 1.0):
        return None
    if addend is not None:
        addend = _rowmajor(addend)
        if addend.data_ptr() % 16 or _ld(addend) % 4:
            return None
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if not lib.cb_gemm_nn_indrop_supported(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(y), _ld(y), _lib.ptr(y), _ld(y), M, N, K):
        return None
    bits = torch.empty((M, 1, 4), dtype=torch.int64, device=a.device) if (want_bits and N == 256 and relu) else None
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_nn_indrop_f32(_lib.ptr(a), _ld(a), _lib.ptr(b), _ld(b), _lib.ptr(y), _ld(y), M, N, K, _lib.ptr(rowscale), _lib.ptr(addend),
                                             _ld(addend) if addend is not None else 0, _lib.ptr(bias), int(bool(relu)), float(p),
                                             ctypes.c_uint64(a_seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(bits), _lib.stream_ptr()),
                   'cb_gemm_nn_indrop_f32')
    return (y, bits) if want_bits else y


def trunk_front(x, w_in, b_in, w0, rowscale, addend, p, seed_x, seed_x0, row0=0, want_bits=False, want_drop=False, z_out=None):
    """The forward front of the residual trunk in one kernel (cb_trunk_front_f32): X0 = relu(dropout_{seed_x}(x) @ w_in^T + b_in) and
    Z0 = rowscale * (dropout_{seed_x0}(X0) @ w0) + addend; dropout(X0) stays on chip unless want_drop.  Returns (x0, bits | None,
    x0_drop | None, z0), or None where the kernel does not exist for the shape (input width not 64 / 128, hidden width not 256,
    misaligned operands).  w_in: nn.Linear layout [256, K]; w0: [256, 256]."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(x, w_in, b_in, w0, rowscale, addend)
    if x.dim() != 2 or x.dtype != torch.float32 or tuple(w_in.shape) != (256, x.shape[1]) or tuple(w0.shape) != (256, 256):
        return None
    M, K = x.shape
    nb_in = lib.cb_front_image_bytes(K)
    if not nb_in or w_in.dtype != torch.float32 or w0.dtype != torch.float32:
        return None
    x, w_in, w0 = _rowmajor(x), _rowmajor(w_in), _rowmajor(w0)
    if addend is not None:
        addend = _rowmajor(addend)
    if x.data_ptr() % 16 or _ld(x) % 4 or (addend is not None and (addend.data_ptr() % 16 or _ld(addend) % 4)):
        return None
    dev = x.device
    img_in = torch.empty(nb_in, dtype=torch.uint8, device=dev)
    nb0 = lib.cb_agg_gemm_image_bytes(256, 256)
    img0 = torch.empty(nb0, dtype=torch.uint8, device=dev)
    x0 = torch.empty((M, 256), dtype=torch.float32, device=dev)
    z0 = z_out if z_out is not None else torch.empty((M, 256), dtype=torch.float32, device=dev)      # (z_out: the caller's [M, 256] matrix, e.g. dist.alloc_exchanged)
    bits = torch.empty((M, 1, 4), dtype=torch.int64, device=dev) if want_bits else None
    xd = torch.empty((M, 256), dtype=torch.float32, device=dev) if want_drop else None
    with torch.cuda.device(dev):
        _lib.check(lib.cb_front_image_f32(_lib.ptr(w_in), _ld(w_in), K, 1, _lib.ptr(img_in), nb_in, _lib.stream_ptr()), 'cb_front_image_f32')
        _lib.check(lib.cb_agg_gemm_image_f32(_lib.ptr(w0), _ld(w0), 256, 256, 0, _lib.ptr(img0), nb0, _lib.stream_ptr()), 'cb_agg_gemm_image_f32')
        _lib.check(lib.cb_trunk_front_f32(_lib.ptr(x), _ld(x), M, K, _lib.ptr(img_in), _lib.ptr(b_in), _lib.ptr(img0), _lib.ptr(rowscale),
                                          _lib.ptr(addend), _ld(addend) if addend is not None else 0, _lib.ptr(x0), 256, _lib.ptr(bits),
                                          _lib.ptr(xd), 256, _lib.ptr(z0), 256, float(p), ctypes.c_uint64(seed_x), ctypes.c_uint64(seed_x0),
                                          ops.seed_dev_ptr(), int(row0), _lib.stream_ptr()), 'cb_trunk_front_f32')
    return x0, bits, xd, z0


def mm_tn_adrop(a, g, p, a_seed, row0=0, rowscale=None):
    """dropout_{a_seed}(a)^T @ (rowscale * g) with the keep-mask regenerated while a is staged (cb_gemm_tn_adrop_f32); None where unsupported."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(a, g, rowscale)
    a, g = _rowmajor(a), _rowmajor(g)
    M, K1 = a.shape
    M2, K2 = g.shape
    if M != M2 or not (0.0 < p < 1.0) or not lib.cb_gemm_tn_adrop_supported(_lib.ptr(a), _ld(a), _lib.ptr(g), _ld(g), K1, K2):
        return None
    out = torch.empty((K1, K2), dtype=torch.float32, device=a.device)
    wsb = lib.cb_gemm_tn_workspace_bytes(M, K1, K2)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_tn_adrop_f32(_lib.ptr(a), _ld(a), _lib.ptr(g), _ld(g), _lib.ptr(rowscale), _lib.ptr(out), M, K1, K2, float(p),
                                            ctypes.c_uint64(a_seed), ops.seed_dev_ptr(), int(row0), _lib.ptr(ws), wsb, _lib.stream_ptr()),
                   'cb_gemm_tn_adrop_f32')
    return out


def mm_tn_gdrop(a, g, p, g_seed, row0=0):
    """a^T @ dropout_{g_seed}(g) with the keep-mask regenerated while g is staged (cb_gemm_tn_gdrop_f32); None where unsupported."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(a, g)
    a, g = _rowmajor(a), _rowmajor(g)
    M, K1 = a.shape
    M2, K2 = g.shape
    if M != M2 or not (0.0 < p < 1.0) or not lib.cb_gemm_tn_gdrop_supported(_lib.ptr(a), _ld(a), _lib.ptr(g), _ld(g), K1, K2):
        return None
    out = torch.empty((K1, K2), dtype=torch.float32, device=a.device)
    wsb = lib.cb_gemm_tn_workspace_bytes(M, K1, K2)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_tn_gdrop_f32(_lib.ptr(a), _ld(a), _lib.ptr(g), _ld(g), _lib.ptr(out), M, K1, K2, float(p), ctypes.c_uint64(g_seed),
                                            ops.seed_dev_ptr(), int(row0), _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_gemm_tn_gdrop_f32')
    return out


def mm_tn_instage_supported(g_like, x, M):
    """The input stage inside the input Linear's weight gradient exists for this shape (cb_gemm_tn_instage_supported): hidden 256, 64 < F <= 128, enough rows."""
    lib = _lib.load()
    return (g_like.dim() == 2 and g_like.shape[1] == 256 and g_like.is_contiguous() and x.dim() == 2 and x.stride(1) == 1
            and bool(lib.cb_gemm_tn_instage_supported(_lib.ptr(g_like), _lib.ptr(g_like), _lib.ptr(x), _ld(x), int(M), int(x.shape[1]))))


def mm_tn_instage(g, mfold, x0_bits, x, p_g, g_seed, p_x, x_seed, row0=0):
    """(layers_MLP[0].weight.grad [256, F], layers_MLP[0].bias.grad [256]) of the fused trunk with its input stage computed while the GEMM stages it
    (cb_gemm_tn_instage_f32): gy = (X0 > 0) * (dropout_bwd_{g_seed}(g) + mfold) is never written; the results are gy^T @ dropout_{x_seed}(x) and gy's
    column sums.  g: dL/d dropout(X0) [M, 256]; mfold: the folded mix gradients (graph.CSRGraph.spmm_store_bwd(mix=...)); x0_bits: int64 [M, 1, 4]."""
    import ctypes
    from . import ops
    lib = _lib.load()
    _lib.require_device(g, mfold, x0_bits, x)
    M, K2 = x.shape
    if (tuple(g.shape) != (M, 256) or tuple(mfold.shape) != (M, 256) or not g.is_contiguous() or not mfold.is_contiguous() or x0_bits.numel() != 4 * M
            or not x0_bits.is_contiguous() or x0_bits.dtype != torch.int64):
        raise ValueError('mm_tn_instage: contiguous [M, 256] gradients, [M, 1, 4] int64 mask words and [M, F] features expected')
    out = torch.empty((256, K2), dtype=torch.float32, device=g.device)
    colsum = torch.empty(256, dtype=torch.float32, device=g.device)
    wsb = lib.cb_gemm_tn_instage_workspace_bytes(M, K2)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.cb_gemm_tn_instage_f32(_lib.ptr(g), _lib.ptr(mfold), _lib.ptr(x0_bits), _lib.ptr(x), _ld(x), _lib.ptr(out), _lib.ptr(colsum), M, K2,
                                              float(p_g), ctypes.c_uint64(g_seed), float(p_x), ctypes.c_uint64(x_seed), ops.seed_dev_ptr(), int(row0),
                                              _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_gemm_tn_instage_f32')
    return out, colsum


def mm_tn(a, g, rowscale=None):
    """a^T @ (rowscale[:,None] * g): a [M,K1], g [M,K2] -> [K1,K2]; deterministic split reduction over M."""
    lib = _lib.load()
    _lib.require_device(a, g, rowscale)
    a, g = _rowmajor(a), _rowmajor(g)
    M, K1 = a.shape
    M2, K2 = g.shape
    if M != M2:
        raise ValueError(f'shape mismatch: {tuple(a.shape)}^T @ {tuple(g.shape)}')
    out = torch.empty((K1, K2), dtype=torch.float32, device=a.device)
    wsb = lib.cb_gemm_tn_workspace_bytes(M, K1, K2)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.cb_gemm_tn_f32(_lib.ptr(a), _ld(a), _lib.ptr(g), _ld(g), _lib.ptr(rowscale), _lib.ptr(out), M, K1, K2,
                                      _lib.ptr(ws), wsb, _lib.stream_ptr()), 'cb_gemm_tn_f32')
    return out


class _LinearRowscaleFn(torch.autograd.Function):
    """Z = rowscale * (X @ W) + E.   dX = rowscale * (dZ @ W^T);  dW = X^T (rowscale * dZ);  dE = dZ."""

    @staticmethod
    def forward(ctx, x, w, rowscale, addend):
        ctx.save_for_backward(x, w, rowscale)
        ctx.has_addend = addend is not None
        return mm_nn(x, w, rowscale=rowscale, addend=addend)

    @staticmethod
    def backward(ctx, g):
        x, w, rowscale = ctx.saved_tensors
        g = _rowmajor(g)
        dx = mm_nn(g, w.t().contiguous(), rowscale=rowscale) if ctx.needs_input_grad[0] else None
        dw = mm_tn(x, g, rowscale=rowscale) if ctx.needs_input_grad[1] else None
        de = g if (ctx.has_addend and ctx.needs_input_grad[3]) else None
        return dx, dw, None, de


def linear_rowscale(x, w, rowscale=None, addend=None):
    return _LinearRowscaleFn.apply(x, w, rowscale, addend)


class _LinearFn(torch.autograd.Function):
    """Y = act(X @ W^T + b) with nn.Linear's weight layout [out, in]."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        y = mm_nn(x, weight.t().contiguous(), bias=bias, relu=relu)
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        from .ops import act_bwd
        x, weight, y = ctx.saved_tensors
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        g = _rowmajor(g)
        if ctx.relu or need_b:
            gm, db = act_bwd(g, y if ctx.relu else None, None, want_out=ctx.relu, want_colsum=need_b)
            if gm is None:
                gm = g
        else:
            gm, db = g, None
        dx = mm_nn(gm, weight) if ctx.needs_input_grad[0] else None       # [M,out] @ [out,in]
        dw = mm_tn(gm, x) if ctx.needs_input_grad[1] else None            # [out,in]
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    _lib.require_device(x, weight)
    return _LinearFn.apply(x, weight, bias, bool(relu))
