"""MSELoss / WeightedMSELoss / DiceLoss with the reference's API (train/losses.py:5-62), computed by
the HIP loss kernels.  Inputs must be GPU tensors (no CPU fallback)."""
import torch.nn as nn

from .. import functional as HF
from ..metrics import dice_coef, weighted_mse  # noqa: F401  (re-exported like train/__init__.py:2-3)

_MSG = "\'reduction\' must be one of (\'mean\', \'sum\'). "


class MSELoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        assert reduction in ('mean', 'sum'), _MSG
        self.reduction = reduction

    def forward(self, inputs, targets):
        return HF.mse_loss(inputs, targets, weighted=False, reduction=self.reduction)


class WeightedMSELoss(nn.Module):
    def __init__(self, reduction='mean'):
        super().__init__()
        assert reduction in ('mean', 'sum'), _MSG
        self.reduction = reduction

    def forward(self, inputs, targets):
        return weighted_mse(inputs, targets, reduction=self.reduction)


class DiceLoss(nn.Module):
    def __init__(self, epsilon=1e-6, reduction='mean'):
        super().__init__()
        assert reduction in ('mean', 'sum'), _MSG
        self.epsilon = epsilon
        self.reduction = reduction

    def forward(self, inputs, targets):
        return HF.dice_loss(inputs, targets, self.epsilon, self.reduction)
