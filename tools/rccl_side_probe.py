#!/usr/bin/env python3
"""The headline step (bench.py's model) with the gradient exchange ON over a one-rank RCCL communicator or OFF, eager, for per-kernel
comparison under rocprofv3:   REDUCER=1|0 [NCCL_MAX_NCHANNELS=n] python3 tools/rccl_side_probe.py [steps]"""
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cellsegmentation_amd import functional as HF, synth  # noqa: E402
from cellsegmentation_amd.optim import Adam  # noqa: E402
from cellsegmentation_amd.parallel import GradReducer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
on = os.environ.get("REDUCER", "1") != "0"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model = bench.build_model(dev, torch.bfloat16)
params = [p for p in model.parameters() if p.requires_grad]
opt = Adam(params, lr=5e-4, weight_decay=1e-4)
x = synth.normalise(synth.ihc_tiles(bench.BAG, bench.SIZE, 1234)).contiguous().to(dev)
labels = torch.tensor([(i * 7 + 1) % 2 for i in range(bench.BAG)], device=dev)
red = None
if on:
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    red = GradReducer(params, force_collectives=True).attach()
    red.broadcast_parameters(model)


def step():
    opt.zero_grad(set_to_none=True)
    HF.cross_entropy(model(x, freeze_bn=True), labels, 1.0).backward()
    if red is not None:
        red.reduce()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
th = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"reducer={'on' if on else 'off'} nchannels={os.environ.get('NCCL_MAX_NCHANNELS', 'default')} {dt * 1e3:.3f} ms/step host {th * 1e3:.3f}", flush=True)
if red is not None:
    red.detach()
    dist.destroy_process_group()
