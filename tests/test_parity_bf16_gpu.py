"""Parity of the BENCHED dtype and of the bit-mask entry points (round-1 review items):

* bf16 training-step gradients of ResNet-50 tile mode at 299x299 against the reference's fp32 gradients (golden vectors,
  train/train.py:32-37 semantics): per-tensor cosine and relative error, bounds measured on MI355X (see BOUND_*);
* cs_conv2d_fwd_bits: the bit tensor equals (stored y > 0) exactly; cs_conv2d_dgrad_bits equals cs_conv2d_dgrad with the bf16 /
  fp32 mask tensor bit for bit -- on launches with and without the prefetched epilogue (multi / single K-step), both dtypes;
* the device top-k against the reference-generated sample/* fixtures (inference.py:31-43);
* the "mode unset" error text (model/resnet.py:305-306).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import functional as HF  # noqa: E402
from cellsegmentation_amd import inference as I  # noqa: E402
from cellsegmentation_amd import kernels as K  # noqa: E402
from cellsegmentation_amd import synth  # noqa: E402
from cellsegmentation_amd.model import resnet as R  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"), allow_pickle=False)

# What "close" means for bf16 gradients is MEASURED, not assumed (tests/golden/probe_bf16_noise.py, MI355X): with n = 2 tiles the early
# layers' gradients hinge on ReLU / max-pool decisions that bf16 rounding flips, so ANY bf16 implementation drifts from the
# reference's fp32 gradients -- torch's own bf16 autocast (ATen / MIOpen kernels driven by the oracle's functional restatement, an
# independent implementation) reaches cosine 0.918 / relative L2 0.40 on layer1.0.conv1.weight and 0.96-0.99 / 0.20-0.28 elsewhere
# in the trunk, while the same oracle in fp32 on the GPU reproduces the golden gradients to <= 8e-4.  This library's bf16 path is
# closer to fp32 than torch-bf16 on 11 of 13 tensors (0.921 / 0.39 on the worst).  This library's figures repeat to the digit
# between boxes; the yardstick's do not (MIOpen picks algorithms per process: layer1.0.conv1.weight 0.918 <-> 0.947,
# layer2.0.conv2.weight 0.934 <-> 0.983 were all observed), so a per-tensor comparison against it is a coin toss.  The test
# pins (a) every tensor to absolute floors and (b) the MEAN relative error / cosine over the probed tensors to the independent
# bf16 implementation's means (+10 % + 0.01 / -0.01), which move by < 0.01 between runs.
ABS_COS_FLOOR = 0.90
ABS_REL_CEIL = 0.45


def _build(dtype, dev):
    torch.manual_seed(0)
    m = R.MILresnet50()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    return m.to(dev).set_compute_dtype(dtype), sd


def _distances(tag, grads):
    out = {}
    for key in GOLD.files:
        pre = f"{tag}/gradfull/"
        if not key.startswith(pre):
            continue
        name = key[len(pre):]
        want = GOLD[key].flatten().astype(np.float64)
        if np.abs(want).max() == 0 or grads.get(name) is None:
            continue
        got = grads[name].detach().flatten()[:4096].double().cpu().numpy()
        out[name] = (float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-300)),
                     float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-300)))
    return out


def test_bf16_training_gradients_against_reference_fp32(dev):
    from oracle import cellseg_oracle as orc
    tag = "resnet50/tile299"
    n, seed = int(GOLD[f"{tag}/n"]), int(GOLD[f"{tag}/seed"])
    x = synth.normalise(synth.ihc_tiles(n, 299, seed)).to(dev)
    labels = torch.from_numpy(GOLD[f"{tag}/labels"]).to(dev)
    m, sd0 = _build(torch.bfloat16, dev)
    m.setmode("tile")
    m.train()
    m.set_encoder_grads(True)
    loss = HF.cross_entropy(m(x, freeze_bn=True), labels, 1.0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(GOLD[f"{tag}/loss"])) < 3e-2 * abs(float(GOLD[f"{tag}/loss"]))
    ours = _distances(tag, {k: p.grad for k, p in m.named_parameters()})
    # the yardsticks: the oracle on the GPU in fp32 (must reproduce the golden vectors) and under bf16 autocast
    sd = {k: v.detach().clone().to(dev).requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd0.items()}
    orc.tile_step_loss(sd, x, labels, "resnet50").backward()
    fp32 = _distances(tag, {k: v.grad for k, v in sd.items()})
    assert max(r for _, r in fp32.values()) < 5e-3, "the fp32 yardstick itself must sit on the golden gradients"
    for v in sd.values():
        v.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l_bf = orc.tile_step_loss(sd, x, labels, "resnet50")
    l_bf.backward()
    torch.cuda.synchronize()
    yard = _distances(tag, {k: v.grad for k, v in sd.items()})
    assert len(ours) >= 10 and set(ours) == set(yard)
    errs = []
    for name, (cos, relerr) in ours.items():
        ycos, yrel = yard[name]
        print(f"{name:34s} this library bf16: cos {cos:.5f} rel {relerr:.4f}   torch bf16 autocast: cos {ycos:.5f} rel {yrel:.4f}")
        if cos < ABS_COS_FLOOR or relerr > ABS_REL_CEIL:
            errs.append(f"{name}: cos {cos:.4f} rel {relerr:.3f} outside the absolute band")
    names = sorted(ours)
    m_rel, m_cos = sum(ours[n][1] for n in names) / len(names), sum(ours[n][0] for n in names) / len(names)
    y_rel, y_cos = sum(yard[n][1] for n in names) / len(names), sum(yard[n][0] for n in names) / len(names)
    print(f"mean over {len(names)} tensors: this library rel {m_rel:.4f} cos {m_cos:.5f}   torch bf16 autocast rel {y_rel:.4f} cos {y_cos:.5f}")
    if m_rel > 1.10 * y_rel + 0.01 or m_cos < y_cos - 0.01:
        errs.append(f"mean rel {m_rel:.4f} / cos {m_cos:.4f} worse than the independent bf16 implementation ({y_rel:.4f} / {y_cos:.4f})")
    assert not errs, "\n".join(errs)


#        N   H   W  Cin Cout R  s  p      (first-generation kernels: cs_conv2d_fwd_bits / cs_conv2d_dgrad_bits)
BIT_SHAPES = [
    (2, 19, 19, 64, 128, 1, 1, 0),     # single K-step forward; data gradient contracts 128 channels (two K-steps: prefetched epilogue)
    (2, 19, 19, 128, 64, 1, 1, 0),     # two K-steps forward; single K-step data gradient
    (2, 19, 19, 64, 64, 3, 1, 1),      # tap-walking launches
    (8, 40, 40, 32, 128, 1, 1, 0),     # 128x128 tiles
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", BIT_SHAPES)
def test_bit_masks_equal_the_tensor_masks(shape, dtype, dev, monkeypatch):
    N, H, W, Cin, Cout, R_, s, p = shape
    g = torch.Generator().manual_seed(3 + Cin + Cout)
    geom = K.make_geom(N, H, W, Cin, Cout, R_, R_, s, p)
    x = torch.randn((N, H, W, Cin), generator=g).to(dtype).to(dev)
    w = (torch.randn((Cout, Cin, R_, R_), generator=g) / (Cin * R_ * R_) ** 0.5).to(dev)
    w_khwc, w_chwk = K.weight_prep(w, None, dtype, Cin, Cout, True, True)
    shift = (torch.randn((Cout,), generator=g) * 0.1).to(dev)
    res = torch.randn((N, geom.P, geom.Q, Cout), generator=g).to(dtype).to(dev)
    for residual in (None, res):
        y, bits = K.conv_fwd(geom, x, w_khwc, None, shift, residual, K.CS_ACT_RELU, want_bits=True)
        y_plain = K.conv_fwd(geom, x, w_khwc, None, shift, residual, K.CS_ACT_RELU)
        torch.cuda.synchronize()
        assert torch.equal(y, y_plain), "the bit-writing launch must store the same tensor"
        assert torch.equal(K.unpack_bits(bits, Cout), y > 0)
    # data gradient: mask given as bits == mask given as the activation tensor, bit for bit
    dy = torch.randn((N, geom.P, geom.Q, Cout), generator=g).to(dtype).to(dev)
    act = torch.relu(torch.randn((N, H, W, Cin), generator=g)).to(dtype).to(dev)       # the conv's input, a post-ReLU tensor
    mask_bits = K.pack_bits(act > 0)
    add = torch.randn((N, H, W, Cin), generator=g).to(dtype).to(dev)
    for a in (None, add):
        cs1 = torch.zeros((Cin,), dtype=torch.float32, device=dev)
        cs2 = torch.zeros((Cin,), dtype=torch.float32, device=dev)
        dx_t = K.conv_dgrad(geom, dy, w_chwk, add=a, mask=act, colsum=cs1)
        dx_b = K.conv_dgrad(geom, dy, w_chwk, add=a, mask_bits=mask_bits, colsum=cs2)
        torch.cuda.synchronize()
        assert torch.equal(dx_t, dx_b)
        assert torch.allclose(cs1, cs2, rtol=1e-5, atol=1e-4)


def test_device_topk_on_the_reference_sample_fixtures(dev):
    """sample() index lists produced by the reference itself (tests/golden/make_golden.py) through the device kernel."""
    n_cases = int(GOLD["sample/n_cases"])
    assert n_cases >= 3
    for ci in range(n_cases):
        groups = GOLD[f"sample/{ci}/groups"]
        labels = dict(zip(GOLD[f"sample/{ci}/label_keys"].tolist(), GOLD[f"sample/{ci}/labels"].tolist()))
        kp, kn = GOLD[f"sample/{ci}/kp_kn"].tolist()
        probs = torch.from_numpy(GOLD[f"sample/{ci}/probs"]).to(dev)
        got = I.select_topk(probs, groups, labels, kp, kn, device=dev)
        assert got == GOLD[f"sample/{ci}/selected"].tolist(), f"case {ci}"


def test_forward_with_mode_unset_raises_the_reference_message(dev):
    m = R.MILresnet18().to(dev)
    m.mode = None
    m.eval()
    with pytest.raises(Exception) as e:
        m(torch.zeros((2, 3, 32, 32), device=dev))
    assert str(e.value) == str(GOLD["forward/unset_msg"]) == "Something wrong in setmode."


GOLD16 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors_n16.npz"), allow_pickle=False)
# n = 16 (tests/golden/make_golden_n16.py): what the benched bf16 kernels are held to, measured on MI355X (round 3, table printed by
# the test): on the 23 well-conditioned tensors cosine 0.981-1.000 / relative L2 0.0001-0.20 -- uniformly ~200x the fp32 parity mode's
# own distance to the reference (4e-4), i.e. bf16 rounding and nothing else.  Two probed heads are ill-conditioned: the first 4096
# elements of layer3.3.conv1.weight (51 % exact zeros: dead channels of the synthetic fill) and of layer1.0.conv2.weight have an rms
# 13x / 3.6x below their tensor's, and the fp32 mode itself is 10x / 4x further from the reference there.  The error is therefore
# normalised by max(rms of the head, rms of the whole reference gradient) (the digest holds the latter), and the cosine is asserted
# where the head carries at least half of the tensor's rms.  A wrong tap on one of 2048 channels, a dropped split-K slab or a stale
# staged weight moves a tensor by far more than this band leaves.
N16_COS_FLOOR = 0.975
N16_NORM_ERR_CEIL = 0.22


def test_bf16_training_gradients_n16_against_reference_fp32(dev):
    """The benched kernels (conv2_halo / conv2_ring / wgrad2, bf16) against the REFERENCE's fp32 gradients of a 16-tile step
    (train/train.py:32-37, --scratch semantics), 25 tensors from the stem to fc_tile, plus the fp32 parity mode on the same case at
    the north-star tolerance."""
    tag = "resnet50/tile299n16"
    n, seed = int(GOLD16[f"{tag}/n"]), int(GOLD16[f"{tag}/seed"])
    xh = synth.normalise(synth.ihc_tiles(n, 299, seed))
    want_digest = GOLD16[f"{tag}/x_digest"]
    t = xh.double().flatten()
    assert np.allclose([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], want_digest[:3], rtol=1e-6)
    x = xh.to(dev)
    labels = torch.from_numpy(GOLD16[f"{tag}/labels"]).to(dev)
    gold_loss = float(GOLD16[f"{tag}/loss"])

    def run(dtype):
        m, _ = _build(dtype, dev)
        m.setmode("tile")
        m.train()
        m.set_encoder_grads(True)
        out = m(x, freeze_bn=True)
        loss = HF.cross_entropy(out, labels, 1.0)
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().float().cpu().numpy(), float(loss), {k: p.grad for k, p in m.named_parameters()}

    def distances(grads):
        d = {}
        for key in GOLD16.files:
            pre = f"{tag}/gradfull/"
            if key.startswith(pre):
                name = key[len(pre):]
                want = GOLD16[key].flatten().astype(np.float64)
                got = grads[name].detach().flatten()[:4096].double().cpu().numpy()
                rms_full = float(np.sqrt(GOLD16[f"{tag}/grad/{name}"][2] / grads[name].numel()))
                rms_head = float(np.sqrt((want * want).mean()))
                d[name] = (float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-300)),
                           float(np.linalg.norm(got - want) / (np.sqrt(want.size) * max(rms_head, rms_full) + 1e-300)),
                           rms_head / (rms_full + 1e-300))
        return d

    # fp32 parity mode: north-star tolerance on logits / loss, gradients within the reference's own fp32 noise band
    lo32, l32, g32 = run(torch.float32)
    ref_logits = GOLD16[f"{tag}/logits_train"]
    assert np.abs(lo32 - ref_logits).max() <= 1e-4 * np.abs(ref_logits).max()
    assert abs(l32 - gold_loss) <= 1e-4 * abs(gold_loss)
    d32 = distances(g32)
    assert len(d32) == 25
    assert max(r for _, r, _ in d32.values()) < 5e-3, d32

    lo16, l16, g16 = run(torch.bfloat16)
    assert np.abs(lo16 - ref_logits).max() <= 5e-2 * np.abs(ref_logits).max()
    assert abs(l16 - gold_loss) < 3e-2 * abs(gold_loss)
    d16 = distances(g16)
    errs = []
    for name, (cos, nerr, cond) in sorted(d16.items()):
        print(f"n16 {name:34s} bf16: cos {cos:.5f} normalised err {nerr:.4f} (head/tensor rms {cond:.2f})   "
              f"fp32 mode: cos {d32[name][0]:.7f} err {d32[name][1]:.2e}")
        if nerr > N16_NORM_ERR_CEIL or (cond >= 0.5 and cos < N16_COS_FLOOR):
            errs.append(f"{name}: cos {cos:.4f} normalised err {nerr:.3f} (head/tensor rms {cond:.2f})")
    assert not errs, "\n".join(errs)
