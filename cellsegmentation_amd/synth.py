"""Deterministic synthetic parameters and inputs (no torch RNG, no downloads).

* ``fill_state_dict`` fills any state_dict in place from a name-keyed counter-based generator
  (FNV-1a of the key -> splitmix64 stream), so the HIP model, the CPU oracle and the imported
  reference all get bit-identical weights from the key names alone.
* ``ihc_tiles`` makes IHC-like uint8 RGB tiles (haematoxylin-blue / DAB-brown blobs on a near-white
  background) and ``normalise`` applies the reference's ToTensor + ImageNet Normalize
  (dataset/dataset.py:78-83).
"""
import numpy as np
import torch

_MASK = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(name: str, n: int, lo: float, hi: float, salt: int = 0) -> np.ndarray:
    """n floats in [lo, hi) determined only by (name, salt)."""
    base = np.uint64((_fnv1a64(name) + salt * 0x9E3779B1) & _MASK)
    with np.errstate(over="ignore"):
        bits = _splitmix64(base + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # 24-bit mantissa: exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32)


def fill_state_dict(sd, salt: int = 0):
    """In-place deterministic fill. Rules (by key suffix / tensor rank):
    conv weight  : U(+-sqrt(6/fan_in))           (variance 2/fan_in, keeps ReLU nets O(1))
    linear weight: U(+-1/sqrt(fan_in)); biases U(+-0.1)
    BN weight    : U(0.8,1.2), except the last BN of a residual branch / MBConv project: U(0.2,0.4)
    BN bias, running_mean: U(+-0.1); running_var: U(0.8,1.2); num_batches_tracked: 0
    """
    for key, t in sd.items():
        if key.endswith("num_batches_tracked"):
            t.zero_()
            continue
        n = t.numel()
        if key.endswith("running_var"):
            v = uniform(key, n, 0.8, 1.2, salt)
        elif key.endswith("running_mean"):
            v = uniform(key, n, -0.1, 0.1, salt)
        elif t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            b = (6.0 / fan_in) ** 0.5
            v = uniform(key, n, -b, b, salt)
        elif t.dim() == 2:
            b = 1.0 / (t.shape[1] ** 0.5)
            v = uniform(key, n, -b, b, salt)
        elif key.endswith("weight"):          # 1-D weight = a norm layer's gamma
            parts = key.split(".")
            last_of_branch = False
            if len(parts) >= 2 and parts[0].startswith("layer"):
                # bottleneck: bn3; basic block: bn2 (no bn3 in the block) -- decided by caller's key set
                blk = ".".join(parts[:2])
                has_bn3 = (blk + ".bn3.weight") in sd
                last_of_branch = parts[2] == ("bn3" if has_bn3 else "bn2")
            if key.startswith("features.") and ".block." in key:
                # MBConv project conv's BN is the last Conv-BN pair of the block
                blk = key.split(".block.")[0]
                idx = int(key.split(".block.")[1].split(".")[0])
                later = [k for k in sd if k.startswith(blk + ".block.") and k.endswith(".1.weight")
                         and int(k.split(".block.")[1].split(".")[0]) > idx]
                last_of_branch = len(later) == 0 and key.endswith(".1.weight")
            v = uniform(key, n, 0.2, 0.4, salt) if last_of_branch else uniform(key, n, 0.8, 1.2, salt)
        else:                                  # biases
            v = uniform(key, n, -0.1, 0.1, salt)
        t.copy_(torch.from_numpy(v).view(t.shape))
    return sd


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def ihc_tiles(n: int, size: int = 299, seed: int = 1234) -> np.ndarray:
    """uint8 [n, size, size, 3] IHC-like synthetic tiles."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = np.empty((n, size, size, 3), dtype=np.uint8)
    blue = np.array([70, 90, 160], dtype=np.float32)
    brown = np.array([120, 80, 50], dtype=np.float32)
    for i in range(n):
        img = np.full((size, size, 3), 235.0, dtype=np.float32)
        n_blobs = int(rng.integers(4, 28))
        for _ in range(n_blobs):
            cy, cx = rng.uniform(0, size, 2)
            r = rng.uniform(max(2.0, size / 60), max(4.0, size / 14))
            a = np.clip(1.2 - ((yy - cy) ** 2 + (xx - cx) ** 2) / (r * r), 0, 1)[..., None]
            col = brown if rng.random() < 0.4 else blue
            img = img * (1 - a) + col * a
        img += rng.normal(0, 8, img.shape).astype(np.float32)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def normalise(tiles_u8: np.ndarray) -> torch.Tensor:
    """uint8 NHWC -> fp32 NCHW, /255 then ImageNet mean/std (dataset/dataset.py:78-83)."""
    x = tiles_u8.astype(np.float32) / 255.0
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
