#!/usr/bin/env python3
"""The squeeze-excitation element-wise passes on EfficientNet-B3's own tensors (bag 64 x 299 x 299; GPU time from graph replays):
squeeze (per-sample channel sums, 1 read), scale (x * s[n, c]: read + write), backward ds (sum dy * x: 2 reads), backward dx
(dy * s + davg / HW: read + write), residual add (2 reads + write).   python tools/se_microbench.py [bag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

bag = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
SHAPES = [(150, 40, 2), (75, 144, 1), (75, 192, 2), (38, 192, 1), (38, 288, 2), (19, 288, 1), (19, 576, 5), (19, 816, 4), (10, 816, 1),
          (10, 1392, 6), (10, 2304, 1)]


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n)


tot = [0.0] * 5
print(f"{'H':>4} {'C':>5} {'n':>2} {'MB':>7} | squeeze us GB/s | scale us GB/s | bwd ds us GB/s | bwd dx us GB/s | add us GB/s")
for h, c, nb in SHAPES:
    x = torch.randn(bag, h, h, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(bag, h, h, c, device=dev).to(torch.bfloat16)
    s = torch.rand(bag, c, device=dev)
    davg = torch.randn(bag, c, device=dev)
    b = x.numel() * 2
    r = [t(lambda: K.sample_sum(x, None, 1.0 / (h * h))), t(lambda: K.se_scale(x, s)), t(lambda: K.se_scale_bwd_ds(dy, x)),
         t(lambda: K.se_scale_bwd_dx(dy, s, davg)), t(lambda: K.rowscale_add(x, None, dy))]
    by = [b, 2 * b, 2 * b, 2 * b, 3 * b]
    for i in range(5):
        tot[i] += r[i] * nb
    print(f"{h:4d} {c:5d} {nb:2d} {b / 1e6:7.1f} | " + " | ".join(f"{r[i] * 1e3:6.1f} {by[i] / r[i] / 1e6:5.0f}" for i in range(5)), flush=True)
    del x, dy
print("per step ms: squeeze %.3f scale %.3f bwd_ds %.3f bwd_dx %.3f add %.3f" % tuple(tot))
