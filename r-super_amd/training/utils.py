"""Optimiser / schedule / EMA with the interfaces of rsuper_train/training/utils.py, backed by the fused
multi-tensor gfx950 kernels (csrc/optim.hip)."""
import ctypes
import math

import torch

from ..hip import lib as _l
from ..hip.ops import _stream


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class FusedAdamWEMA(torch.optim.Optimizer):
    """AdamW(eps=1e-5) as built by get_optimizer (training/utils.py:46-51), fused with the global-norm clip of
    train_ddp.py:352 and the EMA update of update_ema_variables (training/utils.py:154-161): one HBM pass over
    (g, p, m, v, ema).  `step()` alone == torch.optim.AdamW.step(); clipping/EMA are opt-in via `fused_step`."""

    def __init__(self, params, lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._total_sq = None

    def _state_for(self, p):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not isinstance(st['step'], int):
            # resumed from a torch.optim.AdamW / reference optimizer_state_dict (train_ddp.py:184-189): torch keeps `step`
            # as a per-parameter tensor
            st['step'] = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
        return st

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st and not isinstance(st['step'], int):
                st['step'] = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
            for k in ('exp_avg', 'exp_avg_sq'):
                if k in st:
                    st[k] = st[k].float().contiguous()

    @torch.no_grad()
    def grad_sqnorm(self):
        """Device f64 scalar: sum of squared gradient elements over all parameters (no host sync)."""
        from ..hip import ops as _ops
        _ops.join_side()      # weight gradients may still be in flight on the side stream
        ps = [p for g in self.param_groups for p in g['params'] if p.grad is not None]
        if self._total_sq is None:
            self._total_sq = torch.zeros(1, device=ps[0].device, dtype=torch.float64)
        gs = [p.grad for p in ps]
        numel = (ctypes.c_size_t * len(gs))(*[g.numel() for g in gs])
        _l.check(_l.lib().rsuper_grad_sqnorm(len(gs), _ptr_array(gs), numel, self._total_sq.data_ptr(), _stream()), 'grad_sqnorm')
        return self._total_sq

    @torch.no_grad()
    def fused_step(self, max_norm=None, ema_params=None, ema_alpha=0.0):
        """clip_grad_norm_(max_norm) + AdamW + EMA in one pass.  Returns the pre-clip gradient norm (device tensor)
        when max_norm is given."""
        from ..hip import ops as _ops
        _ops.join_side()
        total = self.grad_sqnorm() if max_norm is not None else None
        ema_of = {}
        if ema_params is not None:
            all_p = [p for g in self.param_groups for p in g['params']]
            ema_of = {id(p): e for p, e in zip(all_p, ema_params)}
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                assert p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32
            sts = [self._state_for(p) for p in ps]
            steps = {st['step'] for st in sts}
            assert len(steps) == 1, 'parameters of one group must share the step count'
            step = steps.pop() + 1
            for st in sts:
                st['step'] = step
            b1, b2 = group['betas']
            emas = [ema_of[id(p)] for p in ps] if ema_params is not None else None
            numel = (ctypes.c_size_t * len(ps))(*[p.numel() for p in ps])
            dyn = getattr(self, 'dyn', None)
            if dyn is not None:      # captured step (rsuper_amd.graph): lr / bias corrections / EMA alpha are read from device memory
                assert len(self.param_groups) == 1 and max_norm is not None
                _l.check(_l.lib().rsuper_adamw_ema_step_dyn(
                    len(ps), _ptr_array(ps), _ptr_array([p.grad for p in ps]), _ptr_array([st['exp_avg'] for st in sts]),
                    _ptr_array([st['exp_avg_sq'] for st in sts]), _ptr_array(emas) if emas is not None else None, numel,
                    b1, b2, group['eps'], group['weight_decay'], float(max_norm), total.data_ptr(), dyn.data_ptr(), _stream()), 'adamw_ema_step_dyn')
                continue
            _l.check(_l.lib().rsuper_adamw_ema_step(
                len(ps), _ptr_array(ps), _ptr_array([p.grad for p in ps]), _ptr_array([st['exp_avg'] for st in sts]),
                _ptr_array([st['exp_avg_sq'] for st in sts]), _ptr_array(emas) if emas is not None else None, numel,
                group['lr'], b1, b2, group['eps'], group['weight_decay'], step, float(ema_alpha),
                float(max_norm if max_norm is not None else 0.0), total.data_ptr() if total is not None else None, _stream()),
                'adamw_ema_step')
        _ops.WEIGHTS_EPOCH += 1            # parameters (and EMA) were rewritten through raw pointers: drop cached fragments
        return total.sqrt().float() if total is not None else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.fused_step()
        return loss


def clip_grad_norm_(parameters, max_norm, optimizer=None):
    """torch.nn.utils.clip_grad_norm_ semantics (train_ddp.py:352) on the gfx950 kernels: scales grads in place,
    returns the total norm as a device tensor."""
    ps = [p for p in parameters if p.grad is not None]
    gs = [p.grad for p in ps]
    total = torch.zeros(1, device=gs[0].device, dtype=torch.float64)
    numel = (ctypes.c_size_t * len(gs))(*[g.numel() for g in gs])
    arr = _ptr_array(gs)
    _l.check(_l.lib().rsuper_grad_sqnorm(len(gs), arr, numel, total.data_ptr(), _stream()), 'grad_sqnorm')
    _l.check(_l.lib().rsuper_clip_scale(len(gs), arr, numel, float(max_norm), total.data_ptr(), _stream()), 'clip_scale')
    return total.sqrt().float()


def get_optimizer(args, net):
    """get_optimizer (training/utils.py:10-56): one param group, lr=base_lr, weight_decay; AdamW uses eps=1e-5."""
    root = net.module if hasattr(net, 'module') else net
    params = [p for p in root.parameters()]
    name = args.optimizer.lower()
    if name == 'adamw':
        return FusedAdamWEMA(params, lr=args.base_lr, betas=tuple(args.betas), eps=1e-5, weight_decay=args.weight_decay)
    if name == 'sgd':
        return torch.optim.SGD(params, lr=args.base_lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if name == 'adam':
        return torch.optim.Adam(params, lr=args.base_lr, betas=tuple(args.betas), weight_decay=args.weight_decay)
    raise ValueError(f'Unknown optimizer: {args.optimizer}')


def exp_lr_scheduler_with_warmup(optimizer, epoch, warmup_epoch, max_epoch):
    """(training/utils.py:119-151) exp warm-up then polynomial decay, keeps per-group relative LRs."""
    for g in optimizer.param_groups:
        g.setdefault('base_lr', g['lr'])
    if warmup_epoch and 0 <= epoch <= warmup_epoch:
        lr_mult = math.exp(10.0 * (float(epoch) / float(warmup_epoch) - 1.0))
        if epoch == warmup_epoch:
            lr_mult = 1.0
    else:
        lr_mult = (1.0 - epoch / max_epoch) ** 0.9
    for g in optimizer.param_groups:
        g['lr'] = g['base_lr'] * lr_mult
    return optimizer.param_groups[0]['lr']


def ema_alpha_for_step(alpha, global_step):
    """alpha schedule of update_ema_variables (training/utils.py:156)."""
    return min((1 - 1 / (global_step + 1)), alpha)


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha, global_step):
    """Stand-alone EMA update (training/utils.py:154-161); the training step normally uses the fused path."""
    from ..hip import ops as _ops
    a = ema_alpha_for_step(alpha, global_step)
    for e, p in zip(ema_model.parameters(), model.parameters()):
        e.mul_(a).add_(p, alpha=1 - a)       # in place on the parameter itself (under no_grad): bumps e._version
    for eb, mb in zip(ema_model.buffers(), model.buffers()):
        eb.copy_(mb)
    _ops.WEIGHTS_EPOCH += 1                  # cached MFMA weight fragments of the EMA network are stale now


def unwrap_model_checkpoint(net, ema_net, args):
    """(training/utils.py:71-82) -- but always plain state_dicts, which the reference loaders accept
    (rsuper_train/utils.py:53-56)."""
    root = net.module if hasattr(net, 'module') else net
    sd = root.state_dict()
    esd = None
    if ema_net is not None:
        eroot = ema_net.module if hasattr(ema_net, 'module') else ema_net
        esd = eroot.state_dict()
    return sd, esd
