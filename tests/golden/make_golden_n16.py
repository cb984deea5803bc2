"""A second golden case for the BENCHED dtype: ResNet-50 tile mode at 299x299 with n = 16 tiles, where the
gradient of a training step (train/train.py:32-37, --scratch semantics) averages enough ReLU / max-pool decisions that a bf16
implementation can be held to a tight band (the n = 2 case of reference_vectors.npz cannot: cos >= 0.90).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_n16.py

Runs the REAL reference (imported from /root/reference by make_golden.py's loader, build container only), checks the oracle against it
and writes tests/golden/reference_vectors_n16.npz: data only (logits, loss, gradient digests and the first 4096 elements of 24 tensors)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference modules)

KEYS = ["conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
        "layer1.2.conv3.weight", "layer2.0.conv1.weight", "layer2.0.conv2.weight", "layer2.0.downsample.0.weight", "layer2.0.bn2.weight",
        "layer2.3.conv2.weight", "layer3.0.conv2.weight", "layer3.0.downsample.0.weight", "layer3.3.conv1.weight", "layer3.5.conv2.weight",
        "layer3.5.conv3.weight", "layer4.0.conv2.weight", "layer4.0.downsample.0.weight", "layer4.1.conv1.weight", "layer4.2.conv3.weight",
        "layer4.2.bn3.weight", "layer4.2.bn3.bias", "fc_tile.1.weight", "fc_tile.1.bias"]


def main():
    arch, n, size, seed = "resnet50", 16, 299, 17
    net = G.build_ref(arch)
    x = G.inputs(n, size, seed)
    labels = torch.tensor([(i * 7 + 1) % 2 for i in range(n)])
    net.setmode("tile")
    net.train()
    net.set_encoder_grads(True)
    net.zero_grad()
    logits = net(x, freeze_bn=True)
    loss = torch.nn.functional.cross_entropy(logits, labels) * 1.0
    loss.backward()
    sd = G.oracle_sd(arch)
    o_loss = G.orc.tile_step_loss(sd, x, labels, arch)
    o_loss.backward()
    G.close(o_loss, loss, what="resnet50 n16 tile loss")
    params = dict(net.named_parameters())
    worst = 0.0
    for k, p in params.items():
        if p.grad is not None:
            worst = max(worst, G.close(sd[k].grad, p.grad, 2e-5, f"n16 grad {k}"))
    print(f"  resnet50 tile n=16 size=299: oracle==reference, worst grad rel err {worst:.2e}")
    tag = "resnet50/tile299n16"
    out = {f"{tag}/n": np.array(n), f"{tag}/seed": np.array(seed), f"{tag}/x_digest": G.digest(x), f"{tag}/labels": labels.numpy(),
           f"{tag}/logits_train": logits.detach().numpy(), f"{tag}/loss": np.array(loss.item())}
    for k in KEYS:
        out[f"{tag}/grad/{k}"] = G.digest(params[k].grad)
        out[f"{tag}/gradfull/{k}"] = params[k].grad.flatten()[:4096].numpy()
    path = os.path.join(HERE, "reference_vectors_n16.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
