"""Sliding-window tiling on the device: the host computes the tile grid exactly like the reference's
``get_tiles`` (dataset/dataset.py:718-742, including the border-aligned last row/column); the pixels are gathered, scaled
(ToTensor) and normalised (ImageNet mean/std, dataset/dataset.py:78-83) by one HIP kernel straight into the NHWC compute
layout, so overlapping tiles never cross PCIe more than once."""
import ctypes

import numpy as np
import torch

from . import _lib
from . import kernels as K
from .synth import IMAGENET_MEAN, IMAGENET_STD


def _axis_origins(length, interval, size):
    o = list(range(0, length - size + 1, interval))
    if not o:
        raise ValueError("tile larger than the image")
    if o[-1] + size != length:
        o.append(length - size)
    return o


def get_tiles(shape_hw, interval, size):
    """[(row, col)] upper-left corners, same order as the reference's get_tiles(image, interval, size)."""
    h, w = shape_hw
    return [(x, y) for x in _axis_origins(h, interval, size) for y in _axis_origins(w, interval, size)]


def tile_index(n_images, shape_hw, interval, size):
    """tileIDX (image id per tile) and the (row, col) grid for n_images equally sized images (dataset/dataset.py:118-140)."""
    grid = np.asarray(get_tiles(shape_hw, interval, size), dtype=np.int32)
    tile_img = np.repeat(np.arange(n_images, dtype=np.int32), len(grid))
    tile_rc = np.tile(grid, (n_images, 1))
    return tile_img, tile_rc


def gather_tiles(images_u8, tile_img, tile_rc, size, dtype=torch.bfloat16, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """images_u8: uint8 [n, H, W, 3] on the GPU; tile_img int32 [T]; tile_rc int32 [T, 2] -> NHWC [T, size, size, 8] dtype."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise TypeError("gather_tiles expects uint8 images shaped [n, H, W, 3]")
    n, H, W, _ = images_u8.shape
    dev = images_u8.device
    ti = torch.as_tensor(tile_img, dtype=torch.int32, device=dev).contiguous()
    rc = torch.as_tensor(tile_rc, dtype=torch.int32, device=dev).contiguous()
    T = ti.numel()
    out = torch.empty((T, size, size, 8), dtype=dtype, device=dev)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().cs_tile_gather(K._p(images_u8), n, H, W, K._p(ti), K._p(rc), T, size, m, s, K._code(dtype), K._p(out), K._stream()),
               "tile_gather")
    return out
