"""Optimisers of the path (trainer_node_classification.py:293-296,310): `--optfun` names map to
classes with torch.optim's constructor signature.  Adam runs as ONE fused HIP launch for all parameter
tensors of a group (cb_adam_multi_f32) with torch.optim.Adam's update rule."""
import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = None      # hipGraph mode: int64 device tensor holding the step count
        self._skipped_dev = {}     # device index -> int64 [1]: launches the gradient-check guard kept from writing since the last roll-back (eager mode)
        _lib.register_guard_listener(self)
        self._extra_decay = {}     # parameter -> [1] float32 device tensor added to weight_decay for that tensor (see extra_decay_buffer)

    def extra_decay_buffer(self, p):
        """Persistent device scalar whose value the next step() adds to `weight_decay` for parameter `p` (the kernel reads it at
        launch / hipGraph-replay time).  The trainer writes se_reg / ||le||_F into it every step: the structural-embedding
        regulariser's gradient `se_reg * le / ||le||` then enters the update inside the fused Adam kernel instead of through
        autograd (ops.fold_se_reg).  clear_extra_decay() removes every buffer."""
        buf = self._extra_decay.get(p)
        if buf is None:
            buf = self._extra_decay[p] = torch.zeros(1, dtype=torch.float32, device=p.device)
        return buf

    def clear_extra_decay(self):
        self._extra_decay.clear()

    def make_capturable(self, device):
        """Moves the step count to device memory so that step() can be captured in a hipGraph: the kernels read
        the count (and derive the bias corrections) at replay time.  All parameters must share one step count."""
        steps = {self.state[p]['step'] for g in self.param_groups for p in g['params'] if self.state.get(p)}
        if len(steps) > 1:
            raise RuntimeError('capturable Adam needs one common step count')
        self._step_dev = torch.tensor([steps.pop() if steps else 0], dtype=torch.int64, device=device)
        return self._step_dev

    def state_dict(self):
        """hipGraph replays advance the device-resident step count only: bring the per-tensor Python counts up to date first."""
        if self._step_dev is not None:
            n = int(self._step_dev.item())
            for st in self.state.values():
                if st:
                    st['step'] = n
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # A launch that the gradient-check guard keeps from writing (_lib.grad_guard) is not a step: the device-resident count advances only while
        # the guard word is clear, and in eager mode — where the bias corrections come from the host's counts, which cannot see the word — the
        # skipped launches are counted on the device and taken back off the host's counts when _lib.device_status() reports the failure
        # (roll_back_skipped): an update that "leaves the weights and moments untouched" leaves the step counts untouched too.
        devs = {p.device for g_ in self.param_groups for p in g_['params'] if p.grad is not None and p.is_cuda}
        for dev in devs:
            guard = _lib.grad_guard(dev, create=False)
            if self._step_dev is not None and self._step_dev.device == dev:
                continue
            if guard is not None:
                sk = self._skipped_dev.get(dev.index)
                if sk is None:
                    sk = self._skipped_dev[dev.index] = torch.zeros(1, dtype=torch.int64, device=dev)
                sk.add_(guard.ne(0))
        if self._step_dev is not None:
            guard = _lib.grad_guard(self._step_dev.device, create=False)
            # captured: every replay advances the device-resident count (unless its launch is skipped)
            self._step_dev.add_(1) if guard is None else self._step_dev.add_(guard.eq(0))
        import ctypes
        from . import ops
        capturing = torch.cuda.is_current_stream_capturing()
        for group in self.param_groups:
            b1, b2 = group['betas']
            ps, gs, ms, vs, ns, cs, keep, step = [], [], [], [], [], [], [], None
            qs, wanted = [], []        # per tensor: the [2] device buffer that receives ||p||_F of the updated tensor (None: not asked for)
            for p in group['params']:
                if p.grad is None:
                    continue
                _lib.require_device(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError('fused Adam expects contiguous float32 parameters')
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if step is None:
                    step = st['step']
                extra = self._extra_decay.get(p)
                p._cb_norm = None                      # p changes below: whatever norm was known is of the old values
                if capturing:
                    p._cb_norm_off = True              # replays of this graph will update p without Python seeing it
                if st['step'] != step or p.device != group['params'][0].device:
                    # tensors that joined later (own bias correction) or live elsewhere: one launch of their own
                    one = lambda t, ty=ctypes.c_void_p: (ty * 1)(t)       # noqa: E731
                    with torch.cuda.device(p.device):
                        _lib.check(lib.cb_adam_multi_f32(1, one(p.data_ptr()), one(g.data_ptr()), one(st['exp_avg'].data_ptr()),
                                                         one(st['exp_avg_sq'].data_ptr()), one(p.numel(), ctypes.c_int64),
                                                         one(extra.data_ptr() if extra is not None else None), group['lr'], b1, b2,
                                                         group['eps'], group['weight_decay'], st['step'], _lib.ptr(self._step_dev),
                                                         _lib.ptr(_lib.grad_guard(p.device, create=False)), _lib.stream_ptr()), 'cb_adam_multi_f32')
                    continue
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st['exp_avg'].data_ptr())
                vs.append(st['exp_avg_sq'].data_ptr()); ns.append(p.numel())
                cs.append(extra.data_ptr() if extra is not None else None)
                norm = None
                if getattr(p, '_cb_want_norm', False) and not capturing and not getattr(p, '_cb_norm_off', False):
                    norm = torch.empty(2, dtype=torch.float32, device=p.device)
                    wanted.append((p, norm))
                qs.append(norm.data_ptr() if norm is not None else None)
                keep.append(g)                               # contiguous copies stay alive until the launch below
            if ps:
                n = len(ps)
                arr = lambda vals, ty: (ty * n)(*vals)       # noqa: E731
                wsb = lib.cb_adam_norm_workspace_bytes(len(wanted))
                ws = ops._ws(wsb, group['params'][0].device) if wsb else None
                with torch.cuda.device(group['params'][0].device):
                    _lib.check(lib.cb_adam_multi_norm_f32(n, arr(ps, ctypes.c_void_p), arr(gs, ctypes.c_void_p), arr(ms, ctypes.c_void_p),
                                                          arr(vs, ctypes.c_void_p), arr(ns, ctypes.c_int64),
                                                          arr(cs, ctypes.c_void_p) if any(c is not None for c in cs) else None,
                                                          arr(qs, ctypes.c_void_p) if wanted else None,
                                                          group['lr'], b1, b2, group['eps'],
                                                          group['weight_decay'], step, _lib.ptr(self._step_dev),
                                                          # (a failed gradient check of this step's backward: the launch writes nothing, _lib.grad_guard)
                                                          _lib.ptr(_lib.grad_guard(group['params'][0].device, create=False)), _lib.ptr(ws), wsb,
                                                          _lib.stream_ptr()),
                               'cb_adam_multi_norm_f32')
                for p, norm in wanted:
                    p._cb_norm = (p._version, norm)
                del keep
        # one-shot: a coefficient belongs to the step whose forward wrote it (ops.fold_se_reg rewrites it every step); a later step()
        # after a different loss must not apply a stale one
        for buf in self._extra_decay.values():
            buf.zero_()
        return loss


    def roll_back_skipped(self):
        """Called by _lib.device_status() when it reports a failed gradient check (a host synchronisation has just happened): the launches the
        guard skipped since then did not update anything, so the per-tensor step counts that step() advanced for them are taken back."""
        for idx, sk in self._skipped_dev.items():
            n = int(sk.item())
            if n:
                sk.zero_()
                for g_ in self.param_groups:
                    for p in g_['params']:
                        st = self.state.get(p)
                        if st and p.is_cuda and p.device.index == idx:
                            st['step'] = max(int(st['step']) - n, 0)


def resolve(name):
    if name == 'torch.optim.Adam':
        return Adam
    if name == 'torch.optim.SGD':
        return torch.optim.SGD
    raise ValueError(f'unknown --optfun {name}')
