"""GPU parity of the second-generation 3x3 weight gradient (csrc/wgrad_v2.hip, reached through cs_conv2d_wgrad_batched) against torch's
CPU fp32 weight gradient on the same bf16-rounded operands (autograd backward of model/resnet.py:51-53).  The raw split-K slabs are
summed here; tolerance 1e-2 of max|ref| as for the other bf16 convolution tests (fp32 accumulation, different summation order)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import _lib, kernels as K  # noqa: E402

BF = torch.bfloat16

#        N   H   W   C    K   items
SHAPES = [
    (2, 19, 19, 256, 256, 1),      # 128-position stages, Wp = 20
    (3, 10, 10, 128, 64, 2),       # 128-position stages, Wp = 11; two layers in one launch
    (2, 38, 38, 128, 128, 1),      # 64-position stages, Wp = 39
    (2, 75, 75, 64, 64, 3),        # 64-position stages, Wp = 76; three layers (layer1 of ResNet-50)
    (5, 7, 9, 64, 128, 1),         # non-square, tiny images: several images per stage
    (1, 21, 17, 192, 64, 1),       # three source chunks, one image
    (64, 10, 10, 512, 512, 1),     # layer4 at bench size: 64 tile pairs, few splits
    (1, 150, 150, 128, 64, 1),     # the widest decoder layer (upconv8, resnet.py:163): Wp = 151, the 384-row x region (one workgroup per CU)
    (2, 97, 131, 64, 128, 2),      # Wp = 132, non-square, two layers
]


@pytest.mark.parametrize("shape", SHAPES)
def test_wgrad_v2_matches_torch(shape, dev):
    N, H, W, C, Kc, n = shape
    g = torch.Generator().manual_seed(5 + H + C + Kc)
    geom = K.make_geom(N, H, W, C, Kc, 3, 3, 1, 1)
    lib = _lib.load()
    assert lib.cs_conv2d_wgrad_batched_splits(ctypes.byref(geom), K._code(BF), n) >= 1
    xs = [torch.randn((N, C, H, W), generator=g).to(BF).float() for _ in range(n)]
    dys = [torch.randn((N, Kc, H, W), generator=g).to(BF).float() for _ in range(n)]
    xd = [x.permute(0, 2, 3, 1).contiguous().to(BF).to(dev) for x in xs]
    dyd = [d.permute(0, 2, 3, 1).contiguous().to(BF).to(dev) for d in dys]
    slabs = K.wgrad_batched(geom, xd, dyd)
    torch.cuda.synchronize()
    assert (lib.cs_last_conv_variant() or b"").decode().startswith("wgrad2_kernel"), "the second-generation kernel must have served this shape"
    for i in range(n):
        ref = torch.nn.grad.conv2d_weight(xs[i], (Kc, C, 3, 3), dys[i], stride=1, padding=1)          # [K][C][3][3]
        got = slabs[i].sum(dim=0).cpu().permute(0, 3, 1, 2)                                          # [K][3][3][C] -> [K][C][3][3]
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-2, (shape, i, err)


def test_wgrad_v2_declines(dev):
    lib = _lib.load()
    code = K._code(BF)
    # served by the first-generation kernel: its own split count, never zero
    for geom in (K.make_geom(2, 19, 19, 64, 64, 3, 3, 2, 1), K.make_geom(2, 19, 19, 64, 64, 1, 1, 1, 0), K.make_geom(2, 19, 19, 24, 64, 3, 3, 1, 1)):
        assert lib.cs_conv2d_wgrad_batched_splits(ctypes.byref(geom), code, 1) >= 1
    # fp32 never takes the bf16 kernel
    geom = K.make_geom(2, 19, 19, 64, 64, 3, 3, 1, 1)
    x = torch.randn((2, 19, 19, 64), device=dev)
    dy = torch.randn((2, 19, 19, 64), device=dev)
    K.wgrad_batched(geom, [x], [dy])
    torch.cuda.synchronize()
    assert not (lib.cs_last_conv_variant() or b"").decode().startswith("wgrad2_kernel")
