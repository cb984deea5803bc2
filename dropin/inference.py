"""Drop-in for the reference's ``inference`` module (inference.py:9-153)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd.inference import (inference_image, inference_image_cls, inference_image_reg, inference_seg,  # noqa: F401,E402
                                            inference_tiles, sample, select_topk)
