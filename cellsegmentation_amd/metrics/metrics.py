"""weighted_mse / dice_coef (metrics/metrics.py:23-53) on the HIP kernels.

Same call signatures, assertion messages and return shapes as the reference; differentiable.
"""
import torch

from .. import functional as HF
from .. import kernels as K


def weighted_mse(inputs, targets, reduction="mean"):
    """sum|mean of w*(x-t)^2 with w = ln(t) where t >= 20, else t itself (the reference clones the
    targets as initial weights, metrics/metrics.py:27-31)."""
    assert reduction in ('mean', 'sum'), "\'reduction\' must be one of (\'mean\', \'sum\'). "
    return HF.mse_loss(inputs, targets, weighted=True, reduction=reduction)


class _DiceCoef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t, eps):
        _, sums = K.dice_fwd(p, t, eps, True)
        s = K.dice_sums_values(sums, p.shape[0]).float()
        d = (2 * s[:, 0] + eps) / (s[:, 1] + s[:, 2] + eps)
        ctx.save_for_backward(p, t, s)
        ctx.eps = eps
        return d

    @staticmethod
    def backward(ctx, g):
        p, t, s = ctx.saved_tensors
        num = (2 * s[:, 0] + ctx.eps).unsqueeze(1)
        den = (s[:, 1] + s[:, 2] + ctx.eps).unsqueeze(1)
        return g.unsqueeze(1) * (2 * t * den - num * 2 * p) / (den * den), None, None


def dice_coef(batch_inputs, batch_targets, epsilon=1e-6):
    assert batch_inputs.dtype == batch_targets.dtype, "Input & target vectors should have same dtype. "
    if batch_inputs.ndim == 2 and batch_targets.ndim == 2:
        p = batch_inputs.contiguous().view(1, -1)
        t = batch_targets.contiguous().view(1, -1).float()
        return _DiceCoef.apply(p, t, epsilon)[0]
    p = batch_inputs.contiguous().view(batch_inputs.size()[0], -1)
    t = batch_targets.contiguous().view(batch_targets.size()[0], -1).float()
    return _DiceCoef.apply(p, t, epsilon)
