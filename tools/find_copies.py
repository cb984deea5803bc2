#!/usr/bin/env python3
"""Which host operations of the benched step issue device copies (`__amd_rocclr_copyBuffer`, Memcpy DtoD/HtoD) or fills?
torch.profiler over 3 steps of bench.py's step; prints the aten ops / python frames that own such launches."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cellsegmentation_amd import functional as HF, synth  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, torch.bfloat16)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=5e-4, weight_decay=1e-4, fused=True)
x = synth.normalise(synth.ihc_tiles(8, 299, 1234)).repeat(8, 1, 1, 1).contiguous().to(dev)
labels = torch.tensor([(i * 7 + 1) % 2 for i in range(64)], device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = HF.cross_entropy(model(x, freeze_bn=True), labels, 1.0)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
evs = prof.events()
hits = {}
for e in evs:
    n = e.name
    if "copyBuffer" in n or "Memcpy" in n or "Memset" in n or "fillBuffer" in n:
        hits.setdefault(n, []).append(e)
for n, lst in hits.items():
    print(f"{n}: {len(lst)} in 3 steps, total {sum(x.device_time for x in lst):.1f} us")
# CPU-side ops that launched them: aten::copy_ / aten::to / aten::clone / aten::fill_ / aten::zero_ with shapes and the python frame
interesting = ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::cat")
agg = {}
for e in evs:
    if e.name in interesting and e.device_type == torch.autograd.DeviceType.CPU:
        st = [f for f in (e.stack or []) if "cellsegmentation_amd" in f or "bench" in f or "optim" in f][:2]
        key = (e.name, str(e.input_shapes)[:80], " <- ".join(s.strip()[-70:] for s in st))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e.device_time_total if hasattr(e, "device_time_total") else 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{v[0]:4d}x {k[0]:18s} {k[1]:82s} {k[2]}")
