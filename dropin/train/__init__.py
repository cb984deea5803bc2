"""Drop-in for the reference's ``train`` package (train/__init__.py:1-3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellsegmentation_amd.train import *  # noqa: F401,F403,E402
from cellsegmentation_amd.train import (DiceLoss, MSELoss, WeightedMSELoss, dice_coef, train_alternative, train_image,  # noqa: F401,E402
                                        train_image_cls, train_image_reg, train_seg, train_tile, weighted_mse)
