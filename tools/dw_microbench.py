#!/usr/bin/env python3
"""Per-layer timing of the depthwise kernels on EfficientNet-B3's 26 MBConv geometries (bag 64, 299x299 input): forward + batch
statistics, data gradient, weight gradient; algorithmic GB/s = (input + output bytes) / time.  CELLSEG_DW_UNTILED=1 selects the
element-per-thread kernels for A/B."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402


def b3_layers():
    base = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4),
            (6, 3, 1, 192, 320, 1)]

    def adj(c, w=1.2):
        v = c * w
        n = max(8, int(v + 4) // 8 * 8)
        return n + 8 if n < 0.9 * v else n
    H, cin, out = 150, adj(32), []
    for (e, k, s, ci, co, l) in base:
        co, l = adj(co), int(math.ceil(l * 1.4))
        for i in range(l):
            st = s if i == 0 else 1
            out.append((cin * e, k, st, H))
            H = (H - 1) // 2 + 1 if st == 2 else H
            cin = co
    return out


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", "10"))
    seen = {}
    for cfg in b3_layers():
        seen[cfg] = seen.get(cfg, 0) + 1
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for (C, k, s, H), cnt in seen.items():
        g = K.make_geom(64, H, H, C, C, k, k, s, (k - 1) // 2)
        x = torch.randn((64, H, H, C), device=dev).to(torch.bfloat16)
        dy = torch.randn((64, g.P, g.Q, C), device=dev).to(torch.bfloat16)
        w = torch.randn((k, k, C), device=dev)
        res = {}
        for name, fn in (("fwd", lambda: K.dwconv_fwd_stats(g, x, w)), ("dgrad", lambda: K.dwconv_dgrad(g, dy, w)), ("wgrad", lambda: K.dwconv_wgrad(g, x, dy))):
            # GPU time: `iters` calls captured into one HIP graph and replayed (event-timed Python calls measure the wrapper on small maps)
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(iters):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / (3 * iters)
            tot[name] += res[name] * cnt
        by = (x.numel() + dy.numel()) * 2
        print(f"C{C:5d} k{k} s{s} H{H:4d} x{cnt}  " + "  ".join(f"{n}: {ms * 1e3:7.1f} us {by / ms / 1e6:6.0f} GB/s" for n, ms in res.items()) + f"   {by / 1e6:6.1f} MB", flush=True)
    print("per step (ms):", {k_: round(v, 3) for k_, v in tot.items()})


if __name__ == "__main__":
    main()
