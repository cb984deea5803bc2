"""Timing of cs_weight_prep (fp32 [K][C][R][S] -> bf16 [K][R][S][C] and [C][R][S][K]) on the segmentation decoder's layers.
CELLSEG_WPREP_UNTILED=1 selects the element-per-thread kernel for A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
tot = 0.0
for (k, c, r) in [(1024, 2048, 3), (1024, 2048, 3), (512, 1024, 3), (512, 1024, 3), (256, 512, 3), (256, 512, 3), (128, 256, 3), (64, 128, 3),
                  (256, 64, 1), (512, 2048, 1)]:
    w = torch.randn((k, c, r, r), device=dev)
    for _ in range(3):
        K.weight_prep(w, None, torch.bfloat16, c, k, True, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.weight_prep(w, None, torch.bfloat16, c, k, True, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = w.numel() * 8 / 1e6
    tot += us
    print(f"K{k:5d} C{c:5d} R{r}  {us:8.1f} us  {mb / us * 1e3:7.0f} GB/s")
print(f"total {tot:.1f} us")
