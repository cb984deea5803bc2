"""GPU: two training steps from the same state give the same bits.

The reference's CPU path (fp32 ATen on the host) repeats bit for bit; rounds 1-4 of this library did not wherever a reduction
added per-workgroup partial sums with floating-point atomics in arrival order -- the batch statistics of every train-mode
BatchNorm (model/resnet.py:21,52,112,184,199; model/efficientnet.py:97-103), their backward sums, the column sums of strided data
gradients, the Dice sums (train/losses.py:52-62).  Round 5: exact order-independent accumulators (csrc/cs_common.h: ex_add) and
fixed-order folds.  Checked on the loop bodies of BASELINE.json's configs 1, 2, 4 and 5: losses, every parameter gradient and the
BN running statistics of two runs from one state are compared with torch.equal."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import functional as HF  # noqa: E402
from cellsegmentation_amd import synth  # noqa: E402
from cellsegmentation_amd.model import efficientnet as EN, resnet as R  # noqa: E402


def _fill(m, dev, dtype):
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                              # (torch's RNG stream is not what is tested here)
        if hasattr(mod, "p") and type(mod).__name__ == "StochasticDepth":
            mod.p = 0.0
    return m.to(dev).set_compute_dtype(dtype)


def _twice(m, loss_fn):
    """Run zero_grad -> forward -> loss -> backward twice from the same parameters AND buffers; return both (loss, grads, buffers)."""
    state = {k: v.clone() for k, v in m.state_dict().items()}
    out = []
    for _ in range(2):
        m.load_state_dict(state)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(4321)                      # (any stochastic-depth / dropout draw repeats)
        loss = loss_fn()
        loss.backward()
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                    {k: b.clone() for k, b in m.named_buffers()}))
    return out


def _assert_same(a, b):
    assert torch.equal(a[0], b[0]), (a[0], b[0])
    assert a[1].keys() == b[1].keys() and len(a[1]) > 10
    bad = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
    assert not bad, bad[:8]
    bad = [k for k in a[2] if not torch.equal(a[2][k], b[2][k])]
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_c1_resnet18_image_counter_step_repeats_bit_for_bit(dtype, dev):
    m = _fill(R.MILresnet18(), dev, dtype)
    m.setmode("image")
    m.train()
    x = synth.normalise(synth.ihc_tiles(8, 299, 1234)).to(dev)                  # 8 x 150 x 150 rows: > 512 pixel tiles AND <= 512 ones
    cls = torch.tensor([0, 1, 3, 4, 1, 2, 4, 6], device=dev)
    cnt = torch.tensor([0.0, 3.0, 12.0, 40.0, 1.0, 7.0, 25.0, 230.0], device=dev)

    def loss_fn():
        oc, orr = m(x)
        return HF.cross_entropy(oc, cls) + HF.mse_loss(orr.squeeze(), cnt)
    _assert_same(*_twice(m, loss_fn))


def test_c2_resnet50_tile_scratch_step_repeats_bit_for_bit(dev):
    """Eval-mode BN folded into the convolutions: the sums that used atomics here were the column sums of the three stride-2 3x3 data
    gradients (d beta of layer2/3/4.0.bn1)."""
    m = _fill(R.MILresnet50(), dev, torch.bfloat16)
    m.setmode("tile")
    m.set_encoder_grads(True)
    m.train()
    x = synth.normalise(synth.ihc_tiles(16, 299, 77)).to(dev)
    y = torch.tensor([(i * 7 + 1) % 2 for i in range(16)], device=dev)
    _assert_same(*_twice(m, lambda: HF.cross_entropy(m(x, freeze_bn=True), y)))


@pytest.mark.parametrize("arch,n,size", [("b0", 4, 96), ("b3", 16, 299)])
def test_c4_efficientnet_tile_step_repeats_bit_for_bit(arch, n, size, dev):
    m = _fill({"b0": EN.MILefficientnetB0, "b3": EN.MILefficientnetB3}[arch](num_classes=2), dev, torch.bfloat16)
    m.setmode("tile")
    m.set_encoder_grads(True)
    m.train()
    x = synth.normalise(synth.ihc_tiles(n, size, 5)).to(dev)
    y = torch.tensor([i % 2 for i in range(n)], device=dev)
    _assert_same(*_twice(m, lambda: HF.cross_entropy(m(x, freeze_bn=True), y)))


@pytest.mark.parametrize("arch,n,size", [("resnet18", 2, 128), ("resnet50", 4, 299)])
def test_c5_segmentation_step_repeats_bit_for_bit(arch, n, size, dev):
    m = _fill({"resnet18": R.MILresnet18, "resnet50": R.MILresnet50}[arch](), dev, torch.bfloat16)
    m.setmode("segment")
    m.train()
    x = synth.normalise(synth.ihc_tiles(n, size, 9)).to(dev)
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand((n, size, size), generator=g) > 0.8).float().to(dev)
    _assert_same(*_twice(m, lambda: HF.dice_loss(HF.softmax_channel(m(x), 1), mask)))


def test_resnext50_tile_step_repeats_bit_for_bit(dev):
    """Grouped 3x3 convolutions (model/resnext.py:76-91): float column sums of the grouped data gradients and the per-row dot products
    of the stand-alone finalize were atomic sums in rounds 1-4."""
    m = _fill(R.MILresnext50_32x4d(), dev, torch.bfloat16)
    m.setmode("tile")
    m.set_encoder_grads(True)
    m.train()
    x = synth.normalise(synth.ihc_tiles(8, 128, 31)).to(dev)
    y = torch.tensor([i % 2 for i in range(8)], device=dev)
    _assert_same(*_twice(m, lambda: HF.cross_entropy(m(x, freeze_bn=True), y)))


def test_loss_values_of_many_workgroups_repeat_and_are_exact(dev):
    """The VALUE of a CE / MSE launch with more than one workgroup (the logged per-pixel CE of train/train.py:186: ~700 k rows) is the
    exactly accumulated sum of the workgroups' partial sums: equal across runs, and within fp32 rounding of an fp64 reference."""
    from cellsegmentation_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    M = 300_007
    logits = (torch.randn((M, 2), generator=g) * 3).to(dev)
    labels = torch.randint(0, 2, (M,), generator=g).to(dev)
    a, _ = K.softmax_ce(logits, labels, 1.0, want_grad=False)
    b, _ = K.softmax_ce(logits, labels, 1.0, want_grad=False)
    ref = torch.nn.functional.cross_entropy(logits.double(), labels)
    x, t = torch.randn((M,), generator=g).to(dev), (torch.rand((M,), generator=g) * 40).to(dev)
    c, _ = K.mse(x, t, weighted=True, mean=True, want_grad=False)
    d, _ = K.mse(x, t, weighted=True, mean=True, want_grad=False)
    w = torch.where(t >= 20, torch.log(t.double()), t.double())
    ref2 = (w * (x.double() - t.double()) ** 2).mean()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(c, d)
    assert abs(float(a) - float(ref)) <= 2e-6 * abs(float(ref)) and abs(float(c) - float(ref2)) <= 2e-6 * abs(float(ref2))
