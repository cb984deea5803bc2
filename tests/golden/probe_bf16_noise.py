#!/usr/bin/env python3
"""How far are bf16 training gradients from the reference's fp32 ones for an INDEPENDENT bf16 implementation (torch autocast on
the GPU, ATen/MIOpen kernels driven by the oracle's functional restatement)?  Prints per-tensor cosine / relative L2 of
(a) torch-bf16 and (b) this library's bf16 path against the golden fp32 gradients (ResNet-50 tile, 299x299, n = 2)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellsegmentation_amd import synth, functional as HF
from cellsegmentation_amd.model import resnet as R
from oracle import cellseg_oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz"))
dev = torch.device("cuda:0")
tag = "resnet50/tile299"
n, seed = int(GOLD[f"{tag}/n"]), int(GOLD[f"{tag}/seed"])
x = synth.normalise(synth.ihc_tiles(n, 299, seed)).to(dev)
labels = torch.from_numpy(GOLD[f"{tag}/labels"]).to(dev)

def stats(grads):
    rows = []
    for key in GOLD.files:
        pre = f"{tag}/gradfull/"
        if key.startswith(pre):
            name = key[len(pre):]
            got = grads[name].detach().flatten()[:4096].double().cpu().numpy()
            want = GOLD[key].flatten().astype(np.float64)
            if np.abs(want).max() == 0:
                continue
            rows.append((name, float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-300)),
                         float(np.linalg.norm(got - want) / np.linalg.norm(want))))
    return rows

m = R.MILresnet50()
sd0 = m.state_dict(); synth.fill_state_dict(sd0); m.load_state_dict(sd0)
m = m.to(dev).set_compute_dtype(torch.bfloat16)
m.setmode("tile"); m.train(); m.set_encoder_grads(True)
loss = HF.cross_entropy(m(x, freeze_bn=True), labels, 1.0); loss.backward(); torch.cuda.synchronize()
ours = stats({k: p.grad for k, p in m.named_parameters()})

t0 = time.time()
sd = {k: v.detach().clone().to(dev).requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd0.items()}
for dt_name, ctx in (("torch fp32", torch.autocast("cuda", enabled=False)), ("torch bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
    for v in sd.values():
        v.grad = None
    with ctx:
        l2 = orc.tile_step_loss(sd, x, labels, "resnet50")
    l2.backward(); torch.cuda.synchronize()
    ref = stats({k: v.grad for k, v in sd.items() if v.grad is not None})
    print(f"---- {dt_name}: loss {l2.item():.6f} (golden {float(GOLD[tag + '/loss']):.6f}, ours bf16 {loss.item():.6f}), {time.time() - t0:.1f}s")
    for (name, c, r), (_, c2, r2) in zip(ref, ours):
        print(f"   {name:34s} {dt_name}: cos {c:.5f} rel {r:.4f}   | this library bf16: cos {c2:.5f} rel {r2:.4f}")
