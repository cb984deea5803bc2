"""Timing of the Linear kernels on the squeeze-excitation shapes of EfficientNet-B3 (batch 64): fc1 C -> Csq (SiLU), fc2 Csq -> C (sigmoid),
forward and backward (dx + dw + db).  rocprofv3 --kernel-trace --stats around it gives the per-kernel GPU time without the host's launch rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
# (C, squeeze channels, blocks of EfficientNet-B3 with this pair): 26 MBConv blocks
SHAPES = [(40, 10, 2), (144, 6, 1), (192, 8, 3), (288, 12, 3), (576, 24, 5), (816, 34, 5), (1392, 58, 6), (2304, 96, 1)]
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def t(fn, n=10):
    """GPU microseconds per call: n calls captured into one HIP graph and replayed (the Python wrapper costs more than these kernels)."""
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
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


tot = [0.0] * 4
for C, Q, NB in SHAPES:
    x = torch.randn((BATCH, C), device=dev)
    w1, b1 = torch.randn((Q, C), device=dev) * 0.05, torch.zeros((Q,), device=dev)
    w2, b2 = torch.randn((C, Q), device=dev) * 0.05, torch.zeros((C,), device=dev)
    h, pre = K.linear_fwd(x, w1, b1, K.CS_ACT_SILU, want_preact=True)
    s = K.linear_fwd(h, w2, b2, K.CS_ACT_SIGMOID)
    g2, g1 = torch.randn_like(s), torch.randn_like(h)
    r = [t(lambda: K.linear_fwd(x, w1, b1, K.CS_ACT_SILU, want_preact=True)), t(lambda: K.linear_fwd(h, w2, b2, K.CS_ACT_SIGMOID)),
         t(lambda: K.linear_bwd(h, w2, g2, s, K.CS_ACT_SIGMOID)), t(lambda: K.linear_bwd(x, w1, g1, pre, K.CS_ACT_SILU))]
    for i in range(4):
        tot[i] += r[i] * NB
    print(f"C{C:5d} sq{Q:3d} x{NB}  fc1 fwd {r[0]:6.1f}  fc2 fwd {r[1]:6.1f}  fc2 bwd {r[2]:6.1f}  fc1 bwd {r[3]:6.1f} us", flush=True)
print("per step, 26 blocks (us): fc1 fwd, fc2 fwd, fc2 bwd, fc1 bwd =", [round(v, 1) for v in tot], "total", round(sum(tot), 1))
