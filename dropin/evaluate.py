"""Drop-in for the reference's ``evaluate`` module.  ``evaluate_tile`` (evaluate.py:8-27) runs on the device; every other name
(``evaluate_image``: quadratic weighted kappa, a scalar host metric outside the hot path) falls through to the reference's own
module further down sys.path, so ``from evaluate import evaluate_image`` (train_image.py:24) keeps working."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cellsegmentation_amd.stage import evaluate_tile  # noqa: F401,E402
from _delegate import fallthrough  # noqa: E402
__getattr__ = fallthrough("evaluate", __file__)
