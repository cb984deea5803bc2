"""SURVEY 8(f) rank 4 on the GPU ('no -m gpu test that loads a reference-format .pth into the HIP model').

A pt2 file in the reference's on-disk format (train_tile.py:161-176: {'mode','epoch','state_dict' (prefix-filtered),'encoder',
'optimizer','scheduler'}) is written from a CPU model holding the synthetic weights, then loaded the way the reference's
train_tile.py:242-270 does (`nets[cp['encoder']]`, filtered `load_state_dict(strict=False)`) into a FRESH HIP ResNet-50 on cuda,
which must reproduce the reference-generated golden logits of `resnet50/tile299` at 1e-4 (and the stage hand-off pt1 -> tile)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import checkpoint as C  # noqa: E402
from cellsegmentation_amd import synth  # noqa: E402
from cellsegmentation_amd.model import resnet as R  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"), allow_pickle=False)
RTOL = 1e-4


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _source_model():
    m = R.MILresnet50()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    return m


def _scrambled_nets():
    """A factory whose resnet50 holds DIFFERENT weights: every golden-matching number must come out of the file."""
    m = R.MILresnet50()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(0.5).add_(0.01)
        for n, b in m.named_buffers():
            if "running_var" in n:
                b.fill_(3.0)
    return {"resnet50": m}


def _golden_eval_check(model, dev, tag="resnet50/tile299"):
    n, seed = int(GOLD[f"{tag}/n"]), int(GOLD[f"{tag}/seed"])
    x = synth.normalise(synth.ihc_tiles(n, 299, seed)).to(dev)
    model.set_compute_dtype(torch.float32)
    model.setmode("tile")
    model.eval()
    with torch.no_grad():
        logits = model(x)
        probs = torch.softmax(logits, 1)[:, 1]
    torch.cuda.synchronize()
    return rel(logits.cpu(), GOLD[f"{tag}/logits_eval"]), rel(probs.cpu(), GOLD[f"{tag}/probs"])


def test_pt2_resume_into_the_hip_model_reproduces_the_golden_logits(dev, tmp_path):
    src = _source_model()
    src.setmode("tile")
    opt = torch.optim.Adam([p for p in src.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-4)
    path = C.save_model(7, src, opt, None, str(tmp_path), stage="tile")
    assert os.path.basename(path) == "pt2_7epochs.pth"
    cp = torch.load(path, map_location="cpu")
    assert set(cp) == {"mode", "epoch", "state_dict", "encoder", "optimizer", "scheduler"} and cp["mode"] == "tile"
    assert not any(k.startswith(("upconv", "seg_out")) for k in cp["state_dict"])          # the pt2 name filter

    model, last_epoch, last_sched, cp2 = C.load_for_stage("tile", _scrambled_nets(), path, dev, resume=True)
    assert last_epoch == 7 and last_sched == -1 and cp2["encoder"] == "resnet50"
    assert next(model.parameters()).is_cuda
    e_logits, e_probs = _golden_eval_check(model, dev)
    assert e_logits <= RTOL and e_probs <= RTOL, (e_logits, e_probs)

    # the optimizer state round-trips onto the device-resident parameters (train_tile.py:308-311)
    model.setmode("tile")
    opt2 = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-4)
    C.restore_optimizer(cp2, opt2)


def test_a_scrambled_model_does_not_match_without_the_file(dev):
    """Guards the test above: the scrambled factory on its own is far from the golden logits."""
    m = _scrambled_nets()["resnet50"].to(dev)
    e_logits, _ = _golden_eval_check(m, dev)
    assert e_logits > 1e-2


def test_pt1_hand_off_to_the_tile_stage(dev, tmp_path):
    """`train_tile.py -m pt1.pth` (:262-268): encoder + image heads come from the pt1 file, fc_tile stays the factory's."""
    src = _source_model()
    src.setmode("image")
    opt = torch.optim.Adam([p for p in src.parameters() if p.requires_grad], lr=8e-5)
    path = C.save_model(3, src, opt, None, str(tmp_path), stage="image")
    assert os.path.basename(path) == "pt1_3epochs.pth"
    nets = _scrambled_nets()
    fc_before = {k: v.clone() for k, v in nets["resnet50"].state_dict().items() if k.startswith("fc_tile")}
    model, last_epoch, _, _ = C.load_for_stage("tile", nets, path, dev, resume=False)
    assert last_epoch == 0
    sd = model.state_dict()
    for k, v in fc_before.items():
        assert torch.equal(sd[k].cpu(), v)                                  # not in a pt1 file
    want = src.state_dict()
    for k in ("conv1.weight", "layer3.2.bn2.running_var", "fc_image_cls.4.weight"):
        assert torch.equal(sd[k].cpu(), want[k])
    # with the golden fc_tile put back, the trunk loaded from pt1 reproduces the golden logits
    model.load_state_dict({k: v for k, v in want.items() if k.startswith("fc_tile")}, strict=False)
    e_logits, e_probs = _golden_eval_check(model, dev)
    assert e_logits <= RTOL and e_probs <= RTOL, (e_logits, e_probs)
