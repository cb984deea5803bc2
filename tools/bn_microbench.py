#!/usr/bin/env python3
"""The train-mode BatchNorm passes on EfficientNet-B3's own tensors (bag 64 x 299 x 299): per (pixels, channels) the time and the
algorithmic GB/s of bn_stats (1 read), bn_apply_stats + SiLU (read + write), bn_bwd_reduce (2 reads), bn_bwd_apply (2 reads + write).
python tools/bn_microbench.py [bag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import _lib, kernels as K  # noqa: E402
from cellsegmentation_amd.model import efficientnet as EN  # noqa: E402

bag = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
# (H, C) of every BatchNorm of B3 at 299 x 299: stem, per MBConv block expand (at the input resolution) / depthwise / project, head
shapes = {}
def add(h, c):
    shapes[(h, c)] = shapes.get((h, c), 0) + 1
h = 150
add(h, 40)
cin = 40
for (expand, k, stride, _cin, cout, n) in EN.mbconv_table(1.2, 1.4):
    for i in range(n):
        s = stride if i == 0 else 1
        ce = cin * expand
        if expand != 1:
            add(h, ce)
        h2 = (h + s - 1) // s
        add(h2, ce)
        add(h2, cout)
        h, cin = h2, cout
add(h, 1536)
def t_ms(fn, n=10):
    """GPU time per call: n calls captured into one HIP graph and replayed (the Python wrappers cost 10-20 us per call, more than the
    small tensors' kernels)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n)
tot = [0.0, 0.0, 0.0, 0.0]
print(f"{'H':>4} {'C':>5} {'n':>2} {'MB':>7} | stats us GB/s | apply us GB/s | bwd_reduce us GB/s | bwd_apply us GB/s")
for (h, c), n in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][0] * kv[0][1]):
    M = bag * h * h
    z = torch.randn(M, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, c, device=dev).to(torch.bfloat16)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    nb = z.numel() * 2
    stats = K.bn_stats(z)
    y, mean, rstd = K.bn_apply_stats(z, stats, 1e-5, 0.1, None, None, gamma, beta, act=K.CS_ACT_SILU)
    lib = _lib.load()
    sums = K.new_stats(c, dev)
    ws = K._bn_ws(M, c, dev)
    dz = torch.empty_like(z)
    dg = torch.empty((2, c), dtype=torch.float32, device=dev)
    a = t_ms(lambda: K.bn_stats(z, stats))
    b = t_ms(lambda: K.bn_apply_stats(z, stats, 1e-5, 0.1, None, None, gamma, beta, act=K.CS_ACT_SILU))
    def red():
        sums.zero_()
        lib.cs_bn_bwd_reduce(K._p(dy), K._p(z), K._code(z.dtype), K._p(mean), K._p(rstd), K._p(gamma), K._p(beta), K.CS_ACT_SILU, M, c, K._p(sums), K._p(ws), K._stream())
    def app():
        lib.cs_bn_bwd_apply(K._p(dy), K._p(z), K._code(z.dtype), K._p(mean), K._p(rstd), K._p(gamma), K._p(beta), K.CS_ACT_SILU, K._p(sums), M, c, K._p(dz), K._p(dg[0]), K._p(dg[1]), K._stream())
    zt = t_ms(lambda: sums.zero_())
    r = t_ms(red) - zt
    p = t_ms(app)
    tot[0] += a * n; tot[1] += b * n; tot[2] += r * n; tot[3] += p * n
    print(f"{h:4d} {c:5d} {n:2d} {nb / 1e6:7.1f} | {a * 1e3:6.1f} {nb / a / 1e6:5.0f} | {b * 1e3:6.1f} {2 * nb / b / 1e6:5.0f} | {r * 1e3:6.1f} {2 * nb / r / 1e6:5.0f} | {p * 1e3:6.1f} {3 * nb / p / 1e6:5.0f}", flush=True)
    del z, dy, y, dz
print("per step ms (all BatchNorms, isolated launches): stats %.3f apply %.3f bwd_reduce %.3f bwd_apply %.3f" % tuple(tot))
