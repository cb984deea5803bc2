"""Mirror of the reference's ``train`` package surface (train/__init__.py:1-3)."""
from .train import train_alternative, train_image, train_image_cls, train_image_reg, train_seg, train_tile, use_graphed_steps  # noqa: F401
from .losses import DiceLoss, MSELoss, WeightedMSELoss  # noqa: F401
from .losses import dice_coef, weighted_mse  # noqa: F401
