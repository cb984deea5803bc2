"""Timing of the Linear kernels on the squeeze-excitation shapes of EfficientNet-B3 (batch 64): fc1 C -> Csq (SiLU), fc2 Csq -> C (sigmoid),
forward and backward (dx + dw + db).  rocprofv3 --kernel-trace --stats around it gives the per-kernel GPU time without the host's launch rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(40, 10), (144, 6), (192, 8), (192, 8), (288, 12), (288, 12), (576, 24), (576, 24), (816, 34), (816, 34), (1392, 58), (1392, 58), (2304, 96)]


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = [0.0] * 4
for C, Q in SHAPES:
    x = torch.randn((64, C), device=dev)
    w1, b1 = torch.randn((Q, C), device=dev) * 0.05, torch.zeros((Q,), device=dev)
    w2, b2 = torch.randn((C, Q), device=dev) * 0.05, torch.zeros((C,), device=dev)
    h, pre = K.linear_fwd(x, w1, b1, K.CS_ACT_SILU, want_preact=True)
    s = K.linear_fwd(h, w2, b2, K.CS_ACT_SIGMOID)
    g2, g1 = torch.randn_like(s), torch.randn_like(h)
    r = [t(lambda: K.linear_fwd(x, w1, b1, K.CS_ACT_SILU, want_preact=True)), t(lambda: K.linear_fwd(h, w2, b2, K.CS_ACT_SIGMOID)),
         t(lambda: K.linear_bwd(h, w2, g2, s, K.CS_ACT_SIGMOID)), t(lambda: K.linear_bwd(x, w1, g1, pre, K.CS_ACT_SILU))]
    for i in range(4):
        tot[i] += r[i]
    print(f"C{C:5d} sq{Q:3d}  fc1 fwd {r[0]:6.1f}  fc2 fwd {r[1]:6.1f}  fc2 bwd {r[2]:6.1f}  fc1 bwd {r[3]:6.1f} us", flush=True)
print("sum over the listed blocks (us):", [round(v, 1) for v in tot])
