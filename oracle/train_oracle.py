"""ORACLE -- test infrastructure only.  CPU restatement of the optimiser side of the
reference training step (rsuper_train/train_ddp.py:308-357, rsuper_train/training/utils.py).
"""
import math
import torch


def clip_grad_norm_(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ as called at train_ddp.py:352: global L2 norm,
    coef = max_norm / (norm + 1e-6) clamped to <= 1, grads scaled in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class AdamW:
    """torch.optim.AdamW(lr, betas, eps=1e-5, weight_decay) as built by get_optimizer
    (training/utils.py:46-51): decoupled decay, bias-corrected moments."""

    def __init__(self, params, lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05):
        self.params, self.lr, self.betas, self.eps, self.wd = list(params), lr, betas, eps, weight_decay
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            p.mul_(1 - self.lr * self.wd)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


def update_ema(params, ema_params, alpha, global_step):
    """update_ema_variables (training/utils.py:154-161)."""
    a = min(1 - 1 / (global_step + 1), alpha)
    for e, p in zip(ema_params, params):
        e.mul_(a).add_(p, alpha=1 - a)


def lr_multiplier(epoch, warmup_epoch, max_epoch):
    """exp_lr_scheduler_with_warmup (training/utils.py:119-151)."""
    if warmup_epoch and 0 <= epoch <= warmup_epoch:
        return 1.0 if epoch == warmup_epoch else math.exp(10.0 * (float(epoch) / float(warmup_epoch) - 1.0))
    return (1.0 - epoch / max_epoch) ** 0.9
