#!/usr/bin/env python3
"""Host-side profile of the headline step (bench.py's model and step): cProfile over N steps, top functions by own time, and the
host time to ENQUEUE a step against its GPU time.  [CFG=c1] [REDUCER=1] python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cellsegmentation_amd import functional as HF, synth  # noqa: E402
from cellsegmentation_amd.optim import Adam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
if os.environ.get("CFG", "c2") == "c1":
    # BASELINE configs[0]: the ResNet-18 image counter, batch 8, batch-statistics BN -- ~1000 small launches, host-bound when eager
    from cellsegmentation_amd.model import resnet as R
    model = R.MILresnet18()
    sd = model.state_dict()
    synth.fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(dev).set_compute_dtype(torch.bfloat16)
    model.setmode("image")
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=8e-5, weight_decay=1e-4)
    x = synth.normalise(synth.ihc_tiles(8, 299, 1234)).contiguous().to(dev)
    cls = torch.tensor([0, 1, 3, 4, 1, 2, 4, 6], device=dev)
    cnt = torch.tensor([0.0, 3.0, 12.0, 40.0, 1.0, 7.0, 25.0, 230.0], device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        oc, orr = model(x)
        (HF.cross_entropy(oc, cls) + HF.mse_loss(orr.squeeze(), cnt)).backward()
        opt.step()
elif os.environ.get("CFG") == "c4":
    # BASELINE configs[3]: EfficientNet-B3 tile classifier, bag 64, BN train -- ~1900 launches per step
    from cellsegmentation_amd.model import efficientnet as EN
    model = EN.MILefficientnetB3(num_classes=2)
    sd = model.state_dict()
    synth.fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(dev).set_compute_dtype(torch.bfloat16)
    model.setmode("tile")
    model.set_encoder_grads(True)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=5e-4, weight_decay=1e-4)
    x = synth.normalise(synth.ihc_tiles(8, 299, 1234)).repeat(8, 1, 1, 1).contiguous().to(dev)
    labels = torch.tensor([i % 2 for i in range(64)], device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        HF.cross_entropy(model(x, freeze_bn=True), labels, 1.0).backward()
        opt.step()
else:
    model = bench.build_model(dev, torch.bfloat16)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=5e-4, weight_decay=1e-4)
    x = synth.normalise(synth.ihc_tiles(bench.BAG, bench.SIZE, 1234)).contiguous().to(dev)
    labels = torch.tensor([(i * 7 + 1) % 2 for i in range(bench.BAG)], device=dev)

    red = None
    if os.environ.get("REDUCER", "0") != "0":
        # the one-rank RCCL gradient exchange switched on (what the N > 1 step adds on the host)
        import socket
        import torch.distributed as dist
        from cellsegmentation_amd.parallel import GradReducer
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        red = GradReducer(params, force_collectives=True).attach()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = HF.cross_entropy(model(x, freeze_bn=True), labels, 1.0)
        loss.backward()
        if red is not None:
            red.reduce()
        opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / steps
print(f"host enqueue {t_host * 1e3:.3f} ms/step, wall {t_all * 1e3:.3f} ms/step")
# with the GPU idle in between (host time alone): synchronise before every step
ts = []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    ts.append(time.perf_counter() - t0)
print(f"host enqueue with an idle queue: median {sorted(ts)[len(ts) // 2] * 1e3:.3f} ms/step")
# the HIP backward runs on autograd's device thread, which cProfile (per thread) does not see: profile it from inside
from cellsegmentation_amd import engine as E  # noqa: E402
_orig_bwd = E.backward
bpr = cProfile.Profile()


def _profiled_backward(*a, **k):
    bpr.enable()
    try:
        return _orig_bwd(*a, **k)
    finally:
        bpr.disable()


E.backward = _profiled_backward
for _ in range(steps):
    step()
torch.cuda.synchronize()
E.backward = _orig_bwd
s = io.StringIO()
pstats.Stats(bpr, stream=s).sort_stats("tottime").print_stats(30)
print("---- engine.backward (autograd thread), %d steps" % steps)
print(s.getvalue()[:6000])
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
