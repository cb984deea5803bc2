"""Differentiable host-side wrappers (torch.autograd.Function) around the HIP kernels used outside
the trunk plans: layout changes at the module boundary, pooled head, Linear, BatchNorm1d, losses."""
import torch

from . import kernels as K


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, Cp):
        ctx.C = x.shape[1]
        return K.to_nhwc(x.contiguous(), dtype, Cp)

    @staticmethod
    def backward(ctx, g):
        return K.to_nchw(g.contiguous(), ctx.C), None, None


def to_nhwc(x, dtype, Cp=None):
    return _ToNHWC.apply(x, dtype, Cp or K.pad_channels(x.shape[1]))


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, C):
        ctx.dtype, ctx.Cp = y.dtype, y.shape[-1]
        return K.to_nchw(y, C)

    @staticmethod
    def backward(ctx, g):
        return K.to_nhwc(g.contiguous(), ctx.dtype, ctx.Cp), None


def to_nchw(y, C):
    return _ToNCHW.apply(y, C)


class _GapAvgMax(torch.autograd.Function):
    """AdaptiveAvgPool2d(1)(x) + AdaptiveMaxPool2d(1)(x), flattened (model/resnet.py:266,274)."""

    @staticmethod
    def forward(ctx, x, C, relu_input):
        feat, am = K.gap_fwd(x)
        ctx.save_for_backward(x, am)
        ctx.C, ctx.relu_input = C, relu_input
        return feat[:, :C] if C != feat.shape[1] else feat

    @staticmethod
    def backward(ctx, g):
        x, am = ctx.saved_tensors
        Cp = x.shape[-1]
        if g.shape[1] != Cp:
            gp = torch.zeros((g.shape[0], Cp), dtype=g.dtype, device=g.device)
            gp[:, : g.shape[1]] = g
            g = gp
        return K.gap_bwd(g.contiguous(), am, x, ctx.relu_input), None, None


def gap_avgmax(x_nhwc, C, relu_input=True):
    return _GapAvgMax.apply(x_nhwc, C, relu_input)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x.contiguous()
        y = K.linear_fwd(x, w.detach().contiguous(), b.detach().contiguous() if b is not None else None, act)
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.has_b = act, b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_b and ctx.needs_input_grad[2]
        dx, dw, db = K.linear_bwd(x, w.detach().contiguous(), g.contiguous(), y, ctx.act, need_dx, need_dw or need_db, need_db)
        return dx, (dw if need_dw else None), (db if need_db else None), None


def linear(x, w, b=None, act=K.CS_ACT_NONE):
    return _Linear.apply(x, w, b, act)


class _BatchNormRows(torch.autograd.Function):
    """BatchNorm1d on [M, C] rows with batch statistics (+ optional fused ReLU)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act):
        x = x.contiguous()
        M, C = x.shape
        stats = K.bn_stats(x)
        mean, rstd = K.bn_finalize(stats, M, eps, momentum, running_mean, running_var)
        y = K.bn_apply(x, mean, rstd, gamma.detach(), beta.detach(), None, act)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        flags = K.CS_BN_BWD_OWN_RELU if ctx.act == K.CS_ACT_RELU else 0       # the ReLU mask is applied inside the HIP passes
        dz, dgamma, dbeta = K.bn_bwd(g.contiguous(), x, mean, rstd, gamma.detach(), beta=beta.detach(), act=flags)
        return dz, dgamma, dbeta, None, None, None, None, None


class _AffineRows(torch.autograd.Function):
    """BatchNorm1d in eval mode: y = act(x*scale + shift) from running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, act):
        x = x.contiguous()
        _, _, rstd = K.bn_fold(gamma.detach(), beta.detach(), running_mean, running_var, eps)
        y = K.bn_apply(x, running_mean, rstd, gamma.detach(), beta.detach(), None, act)
        ctx.save_for_backward(x, rstd, gamma, beta, running_mean)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        x, rstd, gamma, beta, mean = ctx.saved_tensors
        # dz = gamma * rstd * g, dgamma = sum g * xhat, dbeta = sum g (g masked by the layer's own ReLU): the train-mode passes with
        # the batch terms switched off
        flags = K.CS_BN_BWD_FROZEN | (K.CS_BN_BWD_OWN_RELU if ctx.act == K.CS_ACT_RELU else 0)
        dz, dgamma, dbeta = K.bn_bwd(g.contiguous(), x, mean, rstd, gamma.detach(), beta=beta.detach(), act=flags)
        return dz, dgamma, dbeta, None, None, None, None


def batch_norm_rows(x, bn, act=K.CS_ACT_NONE):
    """nn.BatchNorm1d forward on GPU rows; honours bn.training."""
    if bn.training:
        if x.shape[0] == 1:
            # nn.BatchNorm1d (resnet.py:134,138) refuses a single row in train mode; a ragged last batch must fail as loudly here
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.shape}")
        momentum = bn.momentum if bn.momentum is not None else 0.1
        y = _BatchNormRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, act)
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        return y
    return _AffineRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, act)


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, gamma):
        loss, dl = K.softmax_ce(logits.contiguous(), labels.contiguous(), gamma, want_grad=True)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def cross_entropy(logits, labels, gamma=1.0):
    """nn.CrossEntropyLoss()(logits, labels) * gamma  (train/train.py:34,80)."""
    return _SoftmaxCE.apply(logits, labels, float(gamma))


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, weighted, mean):
        shape = x.shape
        loss, dx = K.mse(x.contiguous().view(-1), t.contiguous().view(-1).float(), weighted, mean, want_grad=True)
        ctx.save_for_backward(dx)
        ctx.shape = shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (dx * g).view(ctx.shape), None, None, None


def mse_loss(x, t, weighted=False, reduction="mean"):
    return _MSE.apply(x, t, weighted, reduction == "mean")


class _Dice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t, eps, mean):
        if p.ndim == 2 and t.ndim == 2:          # metrics/metrics.py:39-44: one global dice
            p2, t2 = p.contiguous().view(1, -1), t.contiguous().view(1, -1).float()
        else:
            p2 = p.contiguous().view(p.shape[0], -1)
            t2 = t.contiguous().view(t.shape[0], -1).float()
        loss, sums = K.dice_fwd(p2, t2, eps, mean)
        ctx.save_for_backward(p2, t2, sums)
        ctx.eps, ctx.mean, ctx.shape = eps, mean, p.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        p2, t2, sums = ctx.saved_tensors
        dp = K.dice_bwd(p2, t2, sums, ctx.eps, ctx.mean)
        return (dp * g).view(ctx.shape), None, None, None


def dice_loss(p, t, eps=1e-6, reduction="mean"):
    return _Dice.apply(p, t, eps, reduction == "mean")


class _SoftmaxChannel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, ch):
        logits = logits.contiguous()
        ctx.save_for_backward(logits)
        ctx.ch = ch
        return K.softmax_channel_fwd(logits, ch)

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        return K.softmax_channel_bwd(logits, g.contiguous(), ctx.ch), None


def softmax_channel(logits_nchw, ch=1):
    """F.softmax(logits, dim=1)[:, ch] for NCHW fp32 logits (train/train.py:189)."""
    return _SoftmaxChannel.apply(logits_nchw, ch)
