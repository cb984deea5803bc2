"""Per-call timing of the batched weight-gradient finalize (cs_wgrad_finalize_batched) with the arguments of the ResNet-50 tile step.

One bench step is run with K.wgrad_finalize_batched wrapped to record every call (slab shape, K, Cin, partial-row counts); each recorded
call is then repeated REPS times on its own between two events.  Usage (GPU box): python tools/finalize_microbench.py [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from cellsegmentation_amd import kernels as K
from cellsegmentation_amd import engine as E
from cellsegmentation_amd import functional as HF
from cellsegmentation_amd import synth

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def main():
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, torch.bfloat16)
    base = synth.normalise(synth.ihc_tiles(8, bench.SIZE, 1234))
    x = base.repeat(bench.BAG // 8, 1, 1, 1).contiguous().to(dev)
    labels = torch.tensor([(i * 7 + 1) % 2 for i in range(bench.BAG)], device=dev)
    calls = []
    orig = K.wgrad_finalize_batched

    def rec(slabs, ws, scales, rstds, means, gsums, dws, dgammas, dbetas, Cin):
        calls.append((slabs, ws, scales, rstds, means, gsums, dws, dgammas, dbetas, Cin))
        return orig(slabs, ws, scales, rstds, means, gsums, dws, dgammas, dbetas, Cin)

    for target in (K, E):
        if hasattr(target, "wgrad_finalize_batched"):
            setattr(target, "wgrad_finalize_batched", rec)
    for _ in range(2):
        calls.clear()
        for p in model.parameters():
            p.grad = None
        loss = HF.cross_entropy(model(x, freeze_bn=True), labels, 1.0)
        loss.backward()
    torch.cuda.synchronize()
    total = 0.0
    print(f"{'n':>2} {'nsplit':>6} {'K':>5} {'Cin':>5} {'RS':>2} {'grows':>6}  {'us':>8}  {'slab MB':>8}  {'GB/s':>7}")
    for c in calls:
        slabs, ws, scales, rstds, means, gsums, dws, dgammas, dbetas, Cin = c
        n, nsplit, Kp, R, S, Cp = slabs.shape
        grows = [g.rows if isinstance(g, K.PartialColsum) and g._vec is None else 0 for g in gsums]
        # GPU time per call: ten calls captured into one HIP graph and replayed (the Python wrapper costs more than the small launches)
        for _ in range(2):
            orig(*c)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(10):
                orig(*c)
        g_.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g_.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        mb = slabs.numel() * 4 / 1e6
        total += us
        print(f"{n:2d} {nsplit:6d} {dws[0].shape[0]:5d} {Cin:5d} {R * S:2d} {max(grows):6d}  {us:8.1f}  {mb:8.1f}  {mb / us * 1e3:7.0f}")
    print(f"total {total:.1f} us over {len(calls)} calls")


if __name__ == "__main__":
    t0 = time.time()
    main()
