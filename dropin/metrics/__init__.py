"""Drop-in for the reference's ``metrics`` package: the hot-path names (metrics/metrics.py:23-53) come from the MI355X
implementation, everything else (``calc_err``, ``calc_map``, ``qwk`` ...: metrics/__init__.py:1-4, used by evaluate.py:5) falls
through to the reference's own package further down sys.path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd.metrics import dice_coef, weighted_mse  # noqa: F401,E402
from _delegate import fallthrough  # noqa: E402
__getattr__ = fallthrough("metrics", __file__)
