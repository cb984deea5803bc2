#!/usr/bin/env python3
"""Yardstick only (never a product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16, no epilogue) needs for the
plain products behind ResNet-50's deep 1x1 convolutions at bag 64, next to this library's kernels on the same operands
(forward with shift + ReLU + bits).  python tools/gemm_yardstick.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

SHAPES = [("2048->512 @10", 64, 10, 2048, 512), ("512->2048 @10", 64, 10, 512, 2048), ("1024->256 @19", 64, 19, 1024, 256),
          ("256->1024 @19", 64, 19, 256, 1024), ("1024->512 @19", 64, 19, 1024, 512), ("512->128 @38", 64, 38, 512, 128),
          ("128->512 @38", 64, 38, 128, 512), ("256->64 @75", 64, 75, 256, 64), ("64->256 @75", 64, 75, 64, 256)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':16s} {'M':>7s} {'vendor us':>10s} {'TF/s':>7s} | {'ours fwd us':>11s} {'TF/s':>7s}  kernel")
    for name, N, H, C, Kc in SHAPES:
        M = N * H * H
        x = torch.randn((M, C), device=dev).to(torch.bfloat16)
        w = (torch.randn((Kc, C), device=dev) / C ** 0.5).to(torch.bfloat16)
        out = torch.empty((M, Kc), device=dev, dtype=torch.bfloat16)
        t_v = timeit(lambda: torch.matmul(x, w.t(), out=out))
        g = K.make_geom(N, H, H, C, Kc, 1, 1, 1, 0)
        wk, _ = K.weight_prep(w.float().view(Kc, C, 1, 1), None, torch.bfloat16, C, Kc, True, False)
        x4 = x.view(N, H, H, C)
        shift = torch.zeros((Kc,), device=dev)
        if K.packed_supported(g, torch.bfloat16, False):
            wpk = K.pack_conv_weights(g, wk, False)
            t_o = timeit(lambda: K.conv_fwd_packed(g, x4, wpk, shift, None, K.CS_ACT_RELU, want_bits=True))
        else:
            t_o = timeit(lambda: K.conv_fwd(g, x4, wk, None, shift, None, K.CS_ACT_RELU))
        var = (K._lib.load().cs_last_conv_variant() or b"").decode()
        fl = 2.0 * M * C * Kc
        print(f"{name:16s} {M:7d} {t_v:10.1f} {fl / t_v / 1e6:7.0f} | {t_o:11.1f} {fl / t_o / 1e6:7.0f}  {var}", flush=True)


if __name__ == "__main__":
    main()
