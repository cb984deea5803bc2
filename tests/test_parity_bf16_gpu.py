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
        sh = torch.arange(8, dtype=torch.uint8, device=dev)
        unpacked = ((bits.unsqueeze(-1) >> sh) & 1).bool().reshape(N, geom.P, geom.Q, Cout)
        assert torch.equal(unpacked, y > 0)
    # data gradient: mask given as bits == mask given as the activation tensor, bit for bit
    dy = torch.randn((N, geom.P, geom.Q, Cout), generator=g).to(dtype).to(dev)
    act = torch.relu(torch.randn((N, H, W, Cin), generator=g)).to(dtype).to(dev)       # the conv's input, a post-ReLU tensor
    m8 = (act > 0).reshape(N, H, W, Cin // 8, 8).to(torch.uint8)
    mask_bits = (m8 << torch.arange(8, dtype=torch.uint8, device=dev)).sum(-1).to(torch.uint8).contiguous()
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
