#!/usr/bin/env python3
"""HBM bandwidth probe: torch device copy vs this library's elementwise kernels on a layer1-sized tensor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
N, H, W, C = 64, 75, 75, 256
x = torch.randn((N, H, W, C), device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
mean = torch.zeros(C, device=dev); rstd = torch.ones(C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
nbytes = x.numel() * 2


def timeit(name, fn, traffic):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{name:28s} {ms * 1e3:8.1f} us  {traffic / ms / 1e9:7.2f} TB/s", flush=True)


timeit("torch copy_ (r+w)", lambda: y.copy_(x), 2 * nbytes)
timeit("torch relu (r+w)", lambda: torch.relu(x), 2 * nbytes)
timeit("cs bn_apply relu (r+w)", lambda: K.bn_apply(x, mean, rstd, g, b, None, K.CS_ACT_RELU, out=y), 2 * nbytes)
timeit("cs bn_apply +res (2r+w)", lambda: K.bn_apply(x, mean, rstd, g, b, x, K.CS_ACT_RELU, out=y), 3 * nbytes)
timeit("cs colsum (r)", lambda: K.colsum(x), nbytes)
x64 = torch.randn((N, 150, 150, 64), device=dev).to(torch.bfloat16)
timeit("cs maxpool fwd (r + w/4 + idx)", lambda: K.maxpool_fwd(x64), x64.numel() * 2 * 1.25 + x64.numel() / 4)
timeit("torch fill_ (w)", lambda: y.zero_(), nbytes)
timeit("torch sum (r)", lambda: x.sum(), nbytes)
x4 = torch.randn((N, H, W, 64), device=dev).to(torch.bfloat16)
timeit("torch repeat 1->4 (r/4 + w)", lambda: torch.cat([x4, x4, x4, x4], dim=-1, out=y), 1.25 * nbytes)
timeit("cs colsum_partial + fold (r)", lambda: K.colsum_partial(x).vector(), nbytes)
timeit("cs bn_stats (r)", lambda: K.bn_stats(x), nbytes)
