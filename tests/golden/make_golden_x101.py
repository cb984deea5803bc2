"""ResNeXt-101 32x8d (model/resnext.py:431-442) golden vectors: the REAL reference factory run in the build container, the
oracle pinned against it in the same pass (tile mode, n = 2, 64 x 64), written to tests/golden/reference_vectors_x101.npz.
A separate file so that the vectors of make_golden.py stay byte-identical.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_x101.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (imports the reference modules from /root/reference)

ARCH = "resnext101_32x8d"
MG.FACTORY[ARCH] = MG.ref_resnext.MILresnext101_32x8d
# 8 channels per group (ResNeXt-50 has 4): first / middle / last grouped 3x3, the widest 1x1, a strided shortcut
MG.DIGEST_KEYS[ARCH] = ["conv1.weight", "layer1.0.conv2.weight", "layer2.0.conv2.weight", "layer2.0.downsample.0.weight",
                        "layer3.11.conv2.weight", "layer3.22.conv3.weight", "layer4.2.conv2.weight", "layer4.2.conv3.weight",
                        "layer4.2.bn3.weight"]


def main():
    out = {}
    MG.tile_case(ARCH, 2, 64, 17, out)
    path = os.path.join(HERE, "reference_vectors_x101.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
