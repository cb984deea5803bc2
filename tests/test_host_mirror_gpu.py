"""GPU: the reference-shaped host API (train.train_*, inference.inference_tiles/sample, train.losses)
driven end to end like the reference drivers drive it, against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import inference as I  # noqa: E402
from cellsegmentation_amd import synth, train as T  # noqa: E402
from cellsegmentation_amd.model import resnet as R  # noqa: E402
from oracle import cellseg_oracle as orc  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"), allow_pickle=False)


class _Loader(list):
    """Just enough of a DataLoader: iterable of batches + .dataset + .batch_size."""

    def __init__(self, batches, n, batch_size):
        super().__init__(batches)
        self.dataset = range(n)
        self.batch_size = batch_size


def _model(arch, dev):
    m = {"resnet18": R.MILresnet18, "resnet50": R.MILresnet50}[arch]()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.to(dev).set_compute_dtype(torch.float32)


def _oracle_sd(arch, trainable):
    sd = orc.empty_state_dict(arch)
    synth.fill_state_dict(sd)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and trainable(k):
            v.requires_grad_()
    return sd


def test_inference_tiles_and_sample_like_train_tile_py(dev):
    arch, n_img, tiles_per = "resnet18", 6, 9
    x = synth.normalise(synth.ihc_tiles(n_img * tiles_per, 32, 41))
    tile_idx = np.repeat(np.arange(n_img), tiles_per).tolist()
    labels = [0, 2, 0, 1, 5, 0]
    bs = 16
    batches = [(x[i:i + bs], torch.zeros(min(bs, len(x) - i))) for i in range(0, len(x), bs)]
    m = _model(arch, dev)
    m.setmode("tile")
    probs = I.inference_tiles(_Loader(batches, len(x), bs), m, dev)
    ref = orc.tile_probs(_oracle_sd(arch, lambda k: False), x, arch)
    assert probs.dtype == np.float32 and probs.shape == (len(x),)
    assert float(np.abs(probs - ref).max()) < 1e-5

    class _Set:
        tileIDX, got = tile_idx, None

        def __init__(self):
            self.labels = labels

        def __len__(self):
            return len(self.tileIDX)

        def make_train_data(self, idxs, ratio):
            self.got = list(idxs)
            return 1, 2

    ds = _Set()
    I.sample(ds, ref, 1, 3, 0.5)          # identical probs into both selections (SURVEY hard part iii)
    assert ds.got == orc.sample_indices(ref, tile_idx, labels, 1, 3).tolist()


def test_train_tile_epoch_matches_oracle(dev):
    arch, n = "resnet18", 8
    x = synth.normalise(synth.ihc_tiles(n, 32, 43))
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    batches = [(x[:4], y[:4]), (x[4:], y[4:])]
    m = _model(arch, dev)
    m.setmode("tile")                                        # reference default: encoder frozen
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=5e-4, weight_decay=1e-4)
    loss = T.train_tile(_Loader(batches, n, 4), 1, 1, m, dev, torch.nn.CrossEntropyLoss(), opt, None, 1.0)
    sd = _oracle_sd(arch, lambda k: k.startswith("fc_tile"))
    oopt = torch.optim.Adam([sd["fc_tile.1.weight"], sd["fc_tile.1.bias"]], lr=5e-4, weight_decay=1e-4)
    tot = 0.0
    for xb, yb in batches:
        oopt.zero_grad()
        l = orc.tile_step_loss(sd, xb, yb, arch)
        l.backward()
        oopt.step()
        tot += l.item() * len(xb)
    assert abs(loss - tot / n) < 1e-4 * abs(tot / n)
    assert float((m.fc_tile[1].weight.detach().cpu() - sd["fc_tile.1.weight"].detach()).abs().max()) < 1e-5


def test_train_image_step_matches_oracle(dev):
    arch, n = "resnet18", 4
    x = synth.normalise(synth.ihc_tiles(n, 96, 45))
    counts = torch.tensor([0, 3, 12, 40])
    cls = torch.tensor([orc.categorize(int(c)) for c in counts])
    m = _model(arch, dev)
    m.setmode("image")
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, m.parameters()), lr=0.0)
    c, r, t = T.train_image(_Loader([(x, cls, counts)], n, n), 1, 1, m, dev, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), opt, None, 1.0, 1.0)
    sd = _oracle_sd(arch, lambda k: True)
    oc, orr, ot = orc.image_step_loss(sd, x, cls, counts, arch)
    for got, want in ((c, oc.item()), (r, orr.item()), (t, ot.item())):
        assert abs(got - want) < 2e-4 * abs(want)


def test_train_seg_step_matches_oracle(dev):
    arch, n, size = "resnet18", 2, 75
    x = synth.normalise(synth.ihc_tiles(n, size, 47))
    mask = (torch.rand(n, size, size, generator=torch.Generator().manual_seed(1)) > 0.7).to(torch.uint8) * 255
    m = _model(arch, dev)
    m.setmode("segment")
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, m.parameters()), lr=0.0)
    loss = T.train_seg(_Loader([(x, mask, torch.zeros(n))], n, n), 1, 1, m, dev, opt, None)
    want = orc.seg_step_loss(_oracle_sd(arch, lambda k: False), x, (mask / 255).float(), arch).item()
    assert abs(loss - want) < 2e-4 * abs(want)
    # inference_seg 'test' mode returns softmax channel 1
    m2 = _model(arch, dev)
    m2.setmode("segment")
    out = I.inference_seg([x], m2, dev, mode="test")
    ref = torch.softmax(orc.forward(_oracle_sd(arch, lambda k: False), x, arch, "segment", training=False), 1)[:, 1]
    assert out.shape == (n, size, size) and float(np.abs(out - ref.numpy()).max()) < 1e-4


def test_train_alternative_matches_oracle(dev):
    """train/train.py:210-300: a tile step and an image step per batch on ONE optimizer.  Losses against the oracle (train-mode BN in both
    steps: the reference passes no freeze_bn there), and the per-parameter step counts of the one-launch HIP Adam after the epoch: the
    tile head and the image heads step once per batch, the encoder only in image steps (setmode("tile") freezes it)."""
    import torch.nn.functional as F
    from cellsegmentation_amd.optim import Adam
    arch, n = "resnet18", 4
    xi = synth.normalise(synth.ihc_tiles(n, 64, 61))
    xt = synth.normalise(synth.ihc_tiles(2 * n, 32, 62))
    counts = torch.tensor([0, 3, 12, 40])
    cls = torch.tensor([orc.categorize(int(c)) for c in counts])
    yt = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    m = _model(arch, dev)
    m.setmode("image")
    m.set_tile_module_grads(True)
    opt = Adam([p for p in m.parameters() if p.requires_grad], lr=0.0)        # lr 0: the second batch sees the same weights
    batches = [((xi, xt), (cls, counts, yt))] * 2
    tl, cl, rl, sl, il = T.train_alternative(_Loader(batches, 2 * n, n), 1, 1, m, dev, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), opt, None,
                                             0.5, 1.0, 2.0, 0.7, 0.0)
    sd = _oracle_sd(arch, lambda k: False)
    want_t = 0.7 * F.cross_entropy(orc.forward(sd, xt, arch, "tile", training=True), yt).item()
    oc, orr, _ = orc.image_step_loss(_oracle_sd(arch, lambda k: False), xi, cls, counts, arch)
    assert abs(tl - want_t) < 2e-4 * abs(want_t)
    assert abs(cl - oc.item()) < 2e-4 * abs(oc.item()) and abs(rl - orr.item()) < 2e-4 * abs(orr.item())
    assert sl == 0.0 and abs(il - (1.0 * oc.item() + 2.0 * orr.item())) < 2e-4 * abs(il)
    steps = {k: float(opt.state[p]["step"]) for k, p in m.named_parameters() if p in opt.state and len(opt.state[p])}
    assert steps["fc_tile.1.weight"] == 2.0 and steps["fc_image_cls.7.weight"] == 2.0 and steps["layer1.0.conv1.weight"] == 2.0
    assert m.mode == "image"


def test_loss_modules_match_reference_vectors(dev):
    a, b = torch.from_numpy(GOLD["loss/dice_in"]).to(dev), torch.from_numpy(GOLD["loss/dice_tg"]).to(dev)
    assert abs(T.DiceLoss()(a, b).item() - float(GOLD["loss/dice_mean"])) < 1e-5
    assert abs(T.DiceLoss(reduction="sum")(a, b).item() - float(GOLD["loss/dice_sum"])) < 1e-5
    assert abs(T.DiceLoss()(a[0], b[0]).item() - float(GOLD["loss/dice_2d"])) < 1e-5
    x, t = torch.from_numpy(GOLD["loss/wmse_in"]).to(dev), torch.from_numpy(GOLD["loss/wmse_tg"]).to(dev)
    assert abs(T.WeightedMSELoss()(x, t).item() - float(GOLD["loss/wmse_mean"])) < 1e-4 * float(GOLD["loss/wmse_mean"])
    assert abs(T.WeightedMSELoss(reduction="sum")(x, t).item() - float(GOLD["loss/wmse_sum"])) < 1e-4 * float(GOLD["loss/wmse_sum"])
    assert abs(T.MSELoss()(x, t).item() - float(GOLD["loss/mse_mean"])) < 1e-4 * float(GOLD["loss/mse_mean"])
    # gradients of the dice modules against autograd on the oracle formula
    p = a.clone().requires_grad_()
    T.DiceLoss()(p, b).backward()
    pc = a.cpu().clone().requires_grad_()
    orc.dice_loss(pc, b.cpu()).backward()
    assert float((p.grad.cpu() - pc.grad).abs().max()) < 1e-6
    d = T.dice_coef(a, b)
    assert float((d.cpu() - orc.dice_coef(a.cpu(), b.cpu())).abs().max()) < 1e-5


@pytest.mark.parametrize("fused", [False, True])
def test_staged_weights_follow_the_optimizer(dev, fused):
    """The folded/staged weight cache must never outlive a parameter update.  torch's fused optimizers write the
    parameters without bumping `_version`, so a version-keyed cache alone would keep serving the old weights."""
    x = synth.normalise(synth.ihc_tiles(8, 32, 77)).to(dev)
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1], device=dev)
    m = _model("resnet18", dev)
    m.setmode("tile")
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=fused)
    m.eval()
    with torch.no_grad():
        before = m(x).clone()              # fills the cache
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x), y).backward()
        opt.step()
    with torch.no_grad():
        after = m(x).clone()
    fresh = _model("resnet18", dev)        # same architecture, empty cache, the trained parameters
    fresh.load_state_dict(m.state_dict())
    fresh.setmode("tile")
    fresh.eval()
    with torch.no_grad():
        expect = fresh(x)
    assert float((after - before).abs().max()) > 1e-3          # the update is visible at all
    assert torch.equal(after, expect)


@pytest.mark.parametrize("graphed", [False, True], ids=["eager", "graph-replay"])
def test_staged_weights_follow_an_unfrozen_encoder(dev, graphed):
    """eval -> 2 steps -> eval -> 2 steps -> eval with the WHOLE network trainable (train_tile.py --scratch) and a fused
    optimizer: from the second training pass on the folded Conv+BN units are staged by the one-launch pack, which must not
    leave the no-grad cache of the first eval alive; the same after HIP-graph replays, which run no host code at all."""
    from cellsegmentation_amd.graphed import GraphedStep
    x = synth.normalise(synth.ihc_tiles(8, 32, 78)).to(dev)
    y = torch.tensor([1, 0, 1, 0, 0, 1, 1, 0], device=dev)
    m = _model("resnet18", dev)
    m.setmode("tile")
    m.set_encoder_grads(True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=not graphed, capturable=graphed)

    def step(xb, yb):
        opt.zero_grad(set_to_none=False) if graphed else opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(xb, freeze_bn=True), yb)
        loss.backward()
        opt.step()
        return loss

    def check(prev):
        m.eval()
        with torch.no_grad():
            got = m(x).clone()
        fresh = _model("resnet18", dev)
        fresh.load_state_dict(m.state_dict())
        fresh.setmode("tile")
        fresh.eval()
        with torch.no_grad():
            expect = fresh(x)
        assert torch.equal(got, expect)
        if prev is not None:
            assert float((got - prev).abs().max()) > 1e-4
        m.train()
        return got

    out = check(None)
    runner = GraphedStep(step, (x, y)) if graphed else step
    for _ in range(2):
        for _ in range(2):
            runner(x, y)
        out = check(out)


def test_graphed_step_equals_eager_steps(dev):
    """A training step captured into a HIP graph (graphed.GraphedStep) leaves the same parameters, BN running statistics and
    loss as the same number of eager steps: image-mode ResNet-18 with batch-statistics BN, Adam."""
    from cellsegmentation_amd import functional as HF
    from cellsegmentation_amd.graphed import GraphedStep

    def build():
        m = _model("resnet18", dev)
        m.setmode("image")
        m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, capturable=True)
        return m, opt

    x = synth.normalise(synth.ihc_tiles(4, 64, 21)).to(dev)
    cls = torch.tensor([0, 2, 1, 3], device=dev)
    cnt = torch.tensor([0.0, 7.0, 2.0, 12.0], device=dev)

    def make_step(m, opt):
        def step(xb, cb, nb):
            opt.zero_grad(set_to_none=True)
            oc, orr = m(xb)
            loss = HF.cross_entropy(oc, cb) + HF.mse_loss(orr.squeeze(1), nb)
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    m1, o1 = build()
    eager = make_step(m1, o1)
    for _ in range(2 + 3):                       # GraphedStep: 2 warm-up steps + 3 replays
        l1 = eager(x, cls, cnt)
    m2, o2 = build()
    graphed = GraphedStep(make_step(m2, o2), (x, cls, cnt), warmup=2)
    for _ in range(3):
        (l2) = graphed(x, cls, cnt)
    torch.cuda.synchronize()
    assert abs(float(l1) - float(l2)) <= 1e-5 * max(1.0, abs(float(l1)))
    worst = 0.0
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if a.is_floating_point():
            worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-12)))
        else:
            assert torch.equal(a, b), k           # num_batches_tracked
    assert worst < 1e-4, worst                    # fp32 atomics in a few reductions: summation order only
    with pytest.raises(ValueError):
        graphed(x[:2], cls[:2], cnt[:2])


def test_training_reduces_the_loss_bf16(dev):
    """End to end through the HIP path in the throughput dtype: a ResNet-18 tile classifier with a trainable trunk learns a
    separable synthetic task (bright vs dark tiles) in a few dozen fused-Adam steps."""
    torch.manual_seed(0)
    m = R.MILresnet18().to(dev).set_compute_dtype(torch.bfloat16)        # torch's own initialisation (kaiming / BN constants)
    m.setmode("tile")
    m.set_encoder_grads(True)
    m.train()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
    g = torch.Generator().manual_seed(3)
    y = torch.tensor([i % 2 for i in range(32)])
    x = torch.randn((32, 3, 64, 64), generator=g) * 0.5 + (y.float() * 1.5 - 0.75).view(-1, 1, 1, 1)
    x, y = x.to(dev), y.to(dev)
    losses = []
    for _ in range(40):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x, freeze_bn=True).float(), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert min(losses[-5:]) < 0.3 * losses[0], losses[::5]
    with torch.no_grad():
        m.eval()
        acc = float((m(x).argmax(1) == y).float().mean())
    assert acc >= 0.9, acc


def test_mixed_frozen_and_train_mode_batchnorm_trains(dev):
    """a strided shortcut whose own BN is frozen (folded, packed, compact gradient) next to a conv1 whose BN uses batch
    statistics (not packed) must fall back to the dense strided data gradient instead of raising 'a compact gradient reached a
    data gradient that cannot add it'.  Gradients equal the all-first-generation path (CELLSEG_PACKED=0 semantics)."""
    from cellsegmentation_amd import engine as E
    torch.manual_seed(0)

    def run(packed):
        old, old_t = E.PACKED, E.PACKED_TRAIN_BN
        E.PACKED = packed
        E.PACKED_TRAIN_BN = False        # (the 3x3 route under batch statistics has its own test; here only the shortcut logic differs)
        try:
            m = _model("resnet50", dev).set_compute_dtype(torch.bfloat16)
            m.setmode("image")
            m.train()
            for blk in (m.layer2[0], m.layer3[0], m.layer4[0]):
                blk.downsample[1].eval()                     # frozen statistics on the shortcut only
            x = synth.normalise(synth.ihc_tiles(4, 96, 77)).to(dev)
            cls, reg = m(x)
            (cls.float().square().mean() + reg.float().square().mean()).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            E.PACKED, E.PACKED_TRAIN_BN = old, old_t

    got, ref = run(True), run(False)
    assert set(got) == set(ref) and "layer2.0.conv1.weight" in got
    for k in ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer2.0.downsample.0.weight", "layer3.0.conv1.weight"):
        a, b = got[k], ref[k]
        assert torch.isfinite(a).all()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.98, (k, cos)


@pytest.mark.parametrize("loop", ["image", "tile", "seg"])
def test_graphed_train_loops_equal_the_eager_loops_bit_for_bit(loop, dev):
    """train.use_graphed_steps(True): the loops of train/train.py replay zero_grad -> forward -> loss -> backward as one HIP graph once a
    batch shape has been seen twice, the driver's own (non-capturable) optimizer steps eagerly in between, the ragged last batch runs
    eagerly: per-epoch losses, parameters and BN buffers must equal the all-eager loops BIT FOR BIT over two epochs."""
    arch = "resnet18"
    g = torch.Generator().manual_seed(5)
    if loop == "seg":
        full, ragged, size = 2, 1, 64
    else:
        full, ragged, size = 4, 3, 64
    sizes = [full] * 5 + [ragged]
    batches = []
    for i, n in enumerate(sizes):
        x = synth.normalise(synth.ihc_tiles(n, size, 300 + i))
        if loop == "image":
            counts = torch.randint(0, 60, (n,), generator=g)
            batches.append((x, torch.tensor([orc.categorize(int(c)) for c in counts]), counts))
        elif loop == "tile":
            batches.append((x, torch.randint(0, 2, (n,), generator=g)))
        else:
            batches.append((x, (torch.rand(n, size, size, generator=g) > 0.7).to(torch.uint8) * 255, torch.zeros(n)))
    total = sum(sizes)

    def run(graphed):
        m = _model(arch, dev)
        m.setmode({"image": "image", "tile": "tile", "seg": "segment"}[loop])
        if loop == "tile":
            m.set_encoder_grads(True)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        prev = T.use_graphed_steps(graphed)
        out = []
        try:
            for epoch in range(2):
                ld = _Loader(batches, total, full)
                if loop == "image":
                    out.append(T.train_image(ld, epoch, 2, m, dev, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), opt, None, 1.0, 0.5))
                elif loop == "tile":
                    out.append(T.train_tile(ld, epoch, 2, m, dev, torch.nn.CrossEntropyLoss(), opt, None, 1.0))
                else:
                    out.append(T.train_seg(ld, epoch, 2, m, dev, opt, None))
        finally:
            T.use_graphed_steps(prev)
        torch.cuda.synchronize()
        return out, {k: v.clone() for k, v in m.state_dict().items()}, m

    eager_out, eager_sd, _ = run(False)
    graph_out, graph_sd, gm = run(True)
    assert eager_out == graph_out, (eager_out, graph_out)
    for k in eager_sd:
        assert torch.equal(eager_sd[k], graph_sd[k]), k
    from cellsegmentation_amd.train import train as TT
    runners = TT._RUNNERS.get(gm)
    assert runners and any(r.graphs for r in runners.values())         # a graph was captured and replayed (not a silent eager run)
    import copy
    copy.deepcopy(gm)                                                  # train_ensemble.py:202 deep-copies models: the graphs are not on the model
