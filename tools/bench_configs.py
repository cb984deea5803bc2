#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (bench.py itself stays on configs[1]):
  c1  ResNet-18 image-wise counter, batch 8, CE+MSE, BN train, Adam            (configs[0])
  c1api the same step through train.train_image with the driver's torch.optim.Adam: eager / train.use_graphed_steps(True)
  c2f ResNet-50 tile classifier, reference-default frozen encoder (fwd + fc bwd)
  c2s selection pass: eval fwd + softmax + adaptive top-k on 64-tile bags
  c2i the inference pass at the reference's own tile size and batch: 40 960 tiles of 32 x 32 per forward
  c2t the tile-training step at that operating point (batch 40 960 of 32 x 32)
  c4t the same for EfficientNet-B0 / -B3
  c4  EfficientNet-B3 tile classifier, bag 64, BN train                          (configs[3])
  c5  ResNet-50 encoder-decoder, batch 8 at 299x299, Dice, decoder training      (configs[4], per GPU)
  c4g / c5g the c4 / c5 step replayed as one HIP graph
  c5x the same at 512x512, batch 4 (the reference's segmentation resolution, dataset/datasets.py MaskSet)
  c1cpu the c1 step on the host cores through the oracle (torch CPU fp32): the reference's own CPU-runnable case timed beside c1
One JSON line per config: images-or-tiles per second and ms/step (inputs resident in HBM)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import functional as HF, inference as I, synth  # noqa: E402
from cellsegmentation_amd.model import efficientnet as EN, resnet as R  # noqa: E402

dev = torch.device("cuda:0")
STEPS, WARM = int(os.environ.get("STEPS", "10")), 3
FUSED = os.environ.get("FUSED_ADAM", "1") != "0"      # graphed configs: torch's fused implementation (capturable) of the Adam update
HIP_ADAM = os.environ.get("HIP_ADAM", "1") != "0"     # eager configs: cellsegmentation_amd.optim.Adam (one launch), as bench.py


def make_adam(params, lr, wd, capturable=False):
    """eager configs: cellsegmentation_amd.optim.Adam (one launch); graphed configs: the same class with its step counts on the
    device (capturable=True, round 5) -- HIP_ADAM=0 falls back to torch's fused implementation for A/B runs."""
    if HIP_ADAM:
        from cellsegmentation_amd.optim import Adam
        return Adam(params, lr=lr, weight_decay=wd, capturable=capturable)
    return torch.optim.Adam(params, lr=lr, weight_decay=wd, fused=FUSED, capturable=capturable)


def fill(m):
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    return m.to(dev)


def run(name, step, n, unit, roofline_cfg=None):
    for _ in range(WARM):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    line = {"config": name, "value": round(n / dt, 1), "unit": unit, "ms_per_step": round(dt * 1e3, 3)}
    if roofline_cfg and os.environ.get("ROOFLINE"):
        # the dominant conv-family kernel of this config: per-launch HIP events over three more (eager) steps, priced exactly as
        # bench.py prices the headline's; `traffic` from this config's own PMC passes (profiles/round*_traffic_<cfg>.json)
        from bench import roofline_from
        from cellsegmentation_amd import kernels as KK
        timer = KK.LaunchTimer()
        with timer:
            timer.enabled = True
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        roof = roofline_from(timer.results(), 3, torch.bfloat16, traffic_cfg=roofline_cfg)
        if roof:
            line["roofline"] = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "share_of_conv_time")}
            line["conv3x3_family_frac_of_mfma_peak"] = roof.get("conv3x3_family_frac_of_mfma_peak")
    print(json.dumps(line), flush=True)


def tiles(n, size=299, seed=1234):
    base = synth.normalise(synth.ihc_tiles(min(n, 8), size, seed))
    return base.repeat((n + base.shape[0] - 1) // base.shape[0], 1, 1, 1)[:n].contiguous().to(dev)


which = sys.argv[1:] or ["c1", "c1api", "c2f", "c2s", "c2i", "c2t", "c4", "c4g", "c4t", "c5", "c5g", "c5x", "c1cpu"]
if "c1" in which:
    m = fill(R.MILresnet18()); m.setmode("image"); m.train()
    x = tiles(8); counts = torch.tensor([0, 3, 12, 40, 1, 7, 25, 230], device=dev); cls = torch.tensor([0, 1, 3, 4, 1, 2, 4, 6], device=dev)
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 8e-5, 1e-4)

    def s1():
        opt.zero_grad(set_to_none=True)
        oc, orr = m(x)
        (HF.cross_entropy(oc, cls) + HF.mse_loss(orr.squeeze(), counts.float())).backward()
        opt.step()
    run("c1 resnet18 image counter B=8 bf16 (fwd+bwd+Adam, BN train)", s1, 8, "images/s")
    # the same step replayed as one HIP graph (cellsegmentation_amd.graphed): the eager step is host-bound (~1000 launches)
    from cellsegmentation_amd.graphed import GraphedStep
    m = fill(R.MILresnet18()); m.setmode("image"); m.train()
    optg = make_adam([p for p in m.parameters() if p.requires_grad], 8e-5, 1e-4, capturable=True)

    def s1g_body(xb, cb, nb):
        optg.zero_grad(set_to_none=True)
        oc, orr = m(xb)
        loss = HF.cross_entropy(oc, cb) + HF.mse_loss(orr.squeeze(), nb)
        loss.backward()
        optg.step()
        return loss.detach()
    gstep = GraphedStep(s1g_body, (x, cls, counts.float()))
    cf = counts.float()
    run("c1g same step as one HIP graph (GraphedStep)", lambda: gstep(x, cls, cf), 8, "images/s")
if "c1api" in which:
    # the reference-shaped loop itself (cellsegmentation_amd.train.train_image, train/train.py:51-105) with the driver's own optimizer
    # (torch.optim.Adam, host step counts): eager, and with train.use_graphed_steps(True) -- zero_grad..backward replayed as one graph
    from cellsegmentation_amd import train as T

    class _L(list):
        def __init__(self, b, n):
            super().__init__(b)
            self.dataset = range(n)
    x = tiles(8); counts = torch.tensor([0, 3, 12, 40, 1, 7, 25, 230], device=dev); cls = torch.tensor([0, 1, 3, 4, 1, 2, 4, 6], device=dev)
    nb = 60
    for graphed in (False, True):
        m = fill(R.MILresnet18()); m.setmode("image"); m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=8e-5, weight_decay=1e-4)
        prev = T.use_graphed_steps(graphed)
        ld = _L([(x, cls, counts)] * nb, 8 * nb)
        T.train_image(ld, 0, 2, m, dev, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), opt, None, 1.0, 1.0)      # warm (and capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.train_image(ld, 1, 2, m, dev, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), opt, None, 1.0, 1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb
        T.use_graphed_steps(prev)
        print(json.dumps({"config": "c1api train.train_image(...) epoch of 60 batches of 8, torch.optim.Adam, "
                                    + ("train.use_graphed_steps(True)" if graphed else "eager (default)"),
                          "value": round(8 / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3)}), flush=True)
if "c2g" in which:
    # the headline step (bench.py: ResNet-50 tile, trainable trunk, freeze_bn) eager and as one HIP graph: how much of the step is launch gaps / host
    from cellsegmentation_amd.graphed import GraphedStep
    m = fill(R.MILresnet50()); m.setmode("tile"); m.set_encoder_grads(True); m.train()
    x = tiles(64); y = torch.tensor([(i * 7 + 1) % 2 for i in range(64)], device=dev)
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

    def s2e():
        opt.zero_grad(set_to_none=True)
        HF.cross_entropy(m(x, freeze_bn=True), y).backward()
        opt.step()
    run("c2 resnet50 tile bag=64 bf16 --scratch (the headline step), eager", s2e, 64, "tiles/s")
    m = fill(R.MILresnet50()); m.setmode("tile"); m.set_encoder_grads(True); m.train()
    optg = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4, capturable=True)

    def s2g_body(xb, yb):
        optg.zero_grad(set_to_none=True)
        loss = HF.cross_entropy(m(xb, freeze_bn=True), yb)
        loss.backward()
        optg.step()
        return loss.detach()
    g2 = GraphedStep(s2g_body, (x, y))
    run("c2g the same step as one HIP graph (torch fused capturable Adam)", lambda: g2(x, y), 64, "tiles/s")
if "c2f" in which:
    m = fill(R.MILresnet50()); m.setmode("tile"); m.train()
    x = tiles(64); y = torch.tensor([i % 2 for i in range(64)], device=dev)
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

    def s2():
        opt.zero_grad(set_to_none=True)
        HF.cross_entropy(m(x, freeze_bn=True), y).backward()
        opt.step()
    run("c2f resnet50 tile bag=64 bf16, frozen encoder (reference default)", s2, 64, "tiles/s")
if "c2s" in which:
    m = fill(R.MILresnet50()); m.setmode("tile"); m.eval()
    x = tiles(256)
    groups = np.repeat(np.arange(4), 64); labels = [0, 3, 0, 12]

    def s3():
        with torch.no_grad():
            p = HF.K.softmax_prob1(m(x))
        I.select_topk(p, groups, labels, 1, 30, dev)
    run("c2s selection pass: eval fwd + softmax + adaptive top-k, 4 bags x 64 tiles", s3, 256, "tiles/s")
if "c2i" in which:
    # inference_tiles at the REFERENCE's own batch: 40 960 tiles of 32 x 32 per forward (train_tile.py: tile_size 32, interval 20 -> 225
    # tiles per 299 x 299 image, ~4 M tiles per epoch; inference.py:9-28)
    for name, ctor in (("resnet18", R.MILresnet18), ("resnet50", R.MILresnet50)):
        m = fill(ctor()); m.setmode("tile"); m.eval()
        x = tiles(40960, 32)

        def s3i():
            with torch.no_grad():
                HF.K.softmax_prob1(m(x))
        run(f"c2i {name} inference pass, batch 40960 tiles of 32x32 (the reference's tile size and batch), eval fwd + softmax", s3i, 40960, "tiles/s")
if "c2t" in which:
    # the tile-training step at the reference's own operating point: train_tile.py's loader hands out batches of 40 960 tiles of 32 x 32
    # (-b 40960, -t 32); --scratch semantics (whole trunk trains, BN frozen) and the default frozen encoder
    for name, ctor, scratch in (("resnet18", R.MILresnet18, True), ("resnet50", R.MILresnet50, True), ("resnet50", R.MILresnet50, False)):
        m = fill(ctor()); m.setmode("tile"); m.train()
        if scratch:
            m.set_encoder_grads(True)
        x = tiles(40960, 32); y = torch.tensor([(i * 7 + 1) % 2 for i in range(40960)], device=dev)
        opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

        def s2t():
            opt.zero_grad(set_to_none=True)
            HF.cross_entropy(m(x, freeze_bn=True), y).backward()
            opt.step()
        run(f"c2t {name} tile training step, batch 40960 tiles of 32x32 (the reference's -b / -t defaults), " + ("--scratch" if scratch else "frozen encoder"),
            s2t, 40960, "tiles/s")
        del m, opt, x, y
        torch.cuda.empty_cache()
if "c4t" in which:
    # EfficientNet tile training at the reference's own operating point (train_tile.py -b 40960 -t 32): the squeeze-excite Linear layers see a
    # batch axis of 40 960 rows, whose weight gradient runs over row slices (cs_linear_bwd + workspace)
    for name, ctor in (("efficientnet_b0", EN.MILefficientnetB0), ("efficientnet_b3", EN.MILefficientnetB3)):
        m = fill(ctor(num_classes=2)); m.setmode("tile"); m.set_encoder_grads(True); m.train()
        x = tiles(40960, 32); y = torch.tensor([(i * 7 + 1) % 2 for i in range(40960)], device=dev)
        opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

        def s4t():
            opt.zero_grad(set_to_none=True)
            HF.cross_entropy(m(x, freeze_bn=True), y).backward()
            opt.step()
        run(f"c4t {name} tile training step, batch 40960 tiles of 32x32 (the reference's -b / -t defaults), whole trunk trains", s4t, 40960, "tiles/s")
        del m, opt, x, y
        torch.cuda.empty_cache()
if "c4" in which:
    m = fill(EN.MILefficientnetB3(num_classes=2)); m.setmode("tile"); m.set_encoder_grads(True); m.train()
    x = tiles(64); y = torch.tensor([i % 2 for i in range(64)], device=dev)
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

    def s4():
        opt.zero_grad(set_to_none=True)
        HF.cross_entropy(m(x, freeze_bn=True), y).backward()
        opt.step()
    run("c4 efficientnet_b3 tile bag=64 bf16 (fwd+bwd+Adam, BN train), enqueued from Python (~1900 launches per step: host-bound on slow hosts, see c4g)", s4, 64, "tiles/s", roofline_cfg="c4")
    # roofline entry of C4's dominant kernel family (profiles/round3_efficientnet_b3_kernel_stats.md: bn_bwd_reduce_kernel + bn_bwd_apply_kernel,
    # 78 train-mode BatchNorms): HIP events around every BN backward of three more steps; algorithmic bytes = the reduction reads dy and z,
    # the apply pass reads them again and writes dz: 5 passes over an M x C bf16 tensor; HBM-bound by construction
    from cellsegmentation_amd import kernels as KK
    recs, orig = [], KK.bn_bwd

    def timed_bn(dy_, z_, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(dy_, z_, *a, **k)
        e1.record()
        recs.append((5 * z_.numel() * z_.element_size(), e0, e1))
        return out
    KK.bn_bwd = timed_bn
    for _ in range(3):
        s4()
    torch.cuda.synchronize()
    KK.bn_bwd = orig
    by, ms = sum(r[0] for r in recs), sum(r[1].elapsed_time(r[2]) for r in recs)
    print(json.dumps({"config": "c4 roofline: bn_bwd_reduce_kernel + bn_partial_fold_kernel + bn_bwd_apply_kernel (78 BatchNorm backward passes/step, "
                                "the largest rows of the C4 profile)",
                      "roofline": {"bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(by / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None},
                      "algorithmic_mbytes_per_step": round(by / 3 / 1e6, 1), "ms_per_step": round(ms / 3, 3), "launches_per_step": len(recs) // 3}),
          flush=True)
if "c4g" in which:
    # the C4 step as ONE HIP graph (cellsegmentation_amd.graphed.GraphedStep): ~1900 launches per eager step keep the host as busy as the GPU
    from cellsegmentation_amd.graphed import GraphedStep
    m = fill(EN.MILefficientnetB3(num_classes=2)); m.setmode("tile"); m.set_encoder_grads(True); m.train()
    x = tiles(64); y = torch.tensor([i % 2 for i in range(64)], device=dev)
    optg = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4, capturable=True)

    def s4g_body(xb, yb):
        optg.zero_grad(set_to_none=True)
        loss = HF.cross_entropy(m(xb, freeze_bn=True), yb)
        loss.backward()
        optg.step()
        return loss.detach()
    g4 = GraphedStep(s4g_body, (x, y))
    run("c4g efficientnet_b3 tile bag=64 bf16, the same step as one HIP graph", lambda: g4(x, y), 64, "tiles/s")
if "c5g" in which:
    from cellsegmentation_amd.graphed import GraphedStep
    m = fill(R.MILresnet50()); m.setmode("segment"); m.train()
    x = tiles(8); mask = (torch.rand(8, 299, 299, device=dev) > 0.8).float()
    optg = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4, capturable=True)

    def s5g_body(xb, mb):
        optg.zero_grad(set_to_none=True)
        loss = HF.dice_loss(HF.softmax_channel(m(xb), 1), mb)
        loss.backward()
        optg.step()
        return loss.detach()
    g5 = GraphedStep(s5g_body, (x, mask))
    run("c5g resnet50 segment B=8 299x299 bf16, the same step as one HIP graph", lambda: g5(x, mask), 8, "images/s")
if "c5" in which:
    m = fill(R.MILresnet50()); m.setmode("segment"); m.train()
    x = tiles(8); mask = (torch.rand(8, 299, 299, device=dev) > 0.8).float()
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

    def s5():
        opt.zero_grad(set_to_none=True)
        HF.dice_loss(HF.softmax_channel(m(x), 1), mask).backward()
        opt.step()
    run("c5 resnet50 segment B=8 299x299 bf16 (decoder training, Dice)", s5, 8, "images/s", roofline_cfg="c5")
if "c5x" in which:
    m = fill(R.MILresnet50()); m.setmode("segment"); m.train()
    x = tiles(4, 512); mask = (torch.rand(4, 512, 512, device=dev) > 0.8).float()
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 5e-4, 1e-4)

    def s5x():
        opt.zero_grad(set_to_none=True)
        HF.dice_loss(HF.softmax_channel(m(x), 1), mask).backward()
        opt.step()
    run("c5x resnet50 segment B=4 512x512 bf16 (decoder training, Dice)", s5x, 4, "images/s", roofline_cfg="c5x")
if "c1cpu" in which:
    # the oracle is test infrastructure: it runs inside bench.py's cpu_baseline leg only
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import cpu_baseline_c1
    r = cpu_baseline_c1()
    print(json.dumps({"config": f"c1cpu resnet18 image counter B=8 fp32 on the host ({r['cpu_model']}, {r['physical_cores']} cores), oracle port: best of a "
                                f"thread sweep ({r['threads']} threads), median of 5",
                      "value": r["value"], "unit": "images/s", "ms_per_step": round(r["s_per_step"] * 1e3, 1),
                      "thread_sweep_images_per_s": r["thread_sweep_images_per_s"]}), flush=True)
