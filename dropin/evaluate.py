"""Drop-in for the reference's ``evaluate`` module (evaluate.py:8-27).  ``evaluate_image`` scores with the quadratic weighted
kappa (metrics/quadratic_weighted_kappa.py), a scalar CPU metric outside the hot path: not provided."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd.stage import evaluate_tile  # noqa: F401,E402
