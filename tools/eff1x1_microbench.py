#!/usr/bin/env python3
"""The 1x1 products of EfficientNet-B3 (bag 64 x 299 x 299) on the kernels the engine uses for them -- forward with the BN statistics
epilogue, data gradient -- one line per distinct layer: GPU time from graph replays, algorithmic GB/s (x + y once) and TFLOP/s.
python tools/eff1x1_microbench.py [bag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

bag = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
#  (H, Cin, Cout, layers)
LAYERS = [(150, 40, 24, 1), (150, 24, 24, 1), (150, 24, 144, 1), (75, 144, 32, 1), (75, 32, 192, 3), (75, 192, 32, 2), (38, 192, 48, 1),
          (38, 48, 288, 3), (38, 288, 48, 2), (19, 288, 96, 1), (19, 96, 576, 5), (19, 576, 96, 4), (19, 576, 136, 1), (19, 136, 816, 5),
          (19, 816, 136, 4), (10, 816, 232, 1), (10, 232, 1392, 6), (10, 1392, 232, 5), (10, 1392, 384, 1), (10, 384, 2304, 1),
          (10, 2304, 384, 1), (10, 384, 1536, 1)]


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


tot = [0.0, 0.0, 0.0, 0.0]
print(f"{'H':>4} {'Cin':>5} {'Cout':>5} {'n':>2} {'MB':>7} | fwd+stats us GB/s TF/s | dgrad us GB/s TF/s | kernels")
for h, c, k, nl in LAYERS:
    g = K.make_geom(bag, h, h, c, k, 1, 1, 1, 0)
    x = torch.randn((bag, h, h, c), device=dev).to(torch.bfloat16)
    dy = torch.randn((bag, h, h, k), device=dev).to(torch.bfloat16)
    w = torch.randn((k, c, 1, 1), device=dev) * (1.0 / c ** 0.5)
    wk, wc = K.weight_prep(w, None, torch.bfloat16, c, k, True, True)
    stats = K.new_stats(k, dev)
    by = (x.numel() + dy.numel()) * 2
    fl = 2.0 * bag * h * h * c * k
    a = t(lambda: K.conv_fwd(g, x, wk, None, None, None, K.CS_ACT_NONE, stats=stats))
    va = (K._lib.load().cs_last_conv_variant() or b"").decode()
    a0 = t(lambda: K.conv_fwd(g, x, wk, None, None, None, K.CS_ACT_NONE))
    a1 = t(lambda: K.bn_stats(dy, stats))
    b = t(lambda: K.conv_dgrad(g, dy, wc))
    vb = (K._lib.load().cs_last_conv_variant() or b"").decode()
    tot[0] += a * nl
    tot[1] += b * nl
    tot[2] += a0 * nl
    tot[3] += a1 * nl
    print(f"{h:4d} {c:5d} {k:5d} {nl:2d} {by / 1e6:7.1f} | {a * 1e3:6.1f} {by / a / 1e6:5.0f} {fl / a / 1e9:5.0f} | plain {a0 * 1e3:6.1f} + bn_stats {a1 * 1e3:5.1f} | {b * 1e3:6.1f} {by / b / 1e6:5.0f} {fl / b / 1e9:5.0f} | {va[17:31]} / {vb[17:31]}",
          flush=True)
    del x, dy
print("per step ms: fwd with statistics %.3f dgrad %.3f | plain fwd %.3f + bn_stats %.3f" % tuple(tot))
