"""GPU: MBConv kernels (depthwise conv, SE, stochastic-depth scale) and the EfficientNet models against the
CPU oracle.  The oracle's EfficientNet branch is NOT pinned by the reference (torchvision 0.11.2 absent):
these tests prove HIP == restated semantics, per-op against torch functional ops and end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import functional as HF  # noqa: E402
from cellsegmentation_amd import kernels as K  # noqa: E402
from cellsegmentation_amd import synth  # noqa: E402
from cellsegmentation_amd.model import efficientnet as EN  # noqa: E402
from oracle import cellseg_oracle as orc  # noqa: E402


def _nhwc(t, dtype, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(24, 3, 1, 15, 14), (40, 5, 2, 19, 19), (16, 3, 2, 10, 11), (48, 5, 1, 9, 9),
                                 (528, 5, 1, 21, 19), (1032, 3, 2, 12, 13), (264, 5, 2, 12, 11), (72, 3, 1, 1, 7), (8, 5, 2, 2, 3),
                                 (32, 7, 1, 9, 9)])
def test_depthwise_conv(cfg, dtype, dev):
    C, k, s, H, W = cfg
    torch.manual_seed(C + k)
    x = torch.randn(2, C, H, W).to(dtype).float().requires_grad_()
    w = (torch.randn(C, 1, k, k) / k).requires_grad_()
    y = F.conv2d(x, w, None, s, (k - 1) // 2, 1, C)
    dy = torch.randn_like(y).to(dtype).float()
    y.backward(dy)
    g = K.make_geom(2, H, W, C, C, k, k, s, (k - 1) // 2)
    w_hwc = w.detach()[:, 0].permute(1, 2, 0).contiguous().to(dev)
    xd, dyd = _nhwc(x.detach(), dtype, dev), _nhwc(dy, dtype, dev)
    yd = K.dwconv_fwd(g, xd, w_hwc)
    dxd = K.dwconv_dgrad(g, dyd, w_hwc)
    dwd = K.dwconv_wgrad(g, xd, dyd)
    sc, sh = torch.rand(C) + 0.5, torch.randn(C)
    yf = K.dwconv_fwd(g, xd, w_hwc, sc.to(dev), sh.to(dev), K.CS_ACT_SILU)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert _rel(_nchw(yd), y.detach()) < tol
    assert _rel(_nchw(dxd), x.grad) < tol
    assert _rel(dwd.permute(2, 0, 1).unsqueeze(1).cpu(), w.grad) < 1e-4
    assert _rel(_nchw(yf), F.silu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))) < tol
    # the train-mode forward: the same z plus the batch statistics of the STORED values (fp64 sum / sum of squares per channel)
    zs, stats = K.dwconv_fwd_stats(g, xd, w_hwc)
    torch.cuda.synchronize()
    assert torch.equal(zs, yd)
    zf = zs.double().cpu().reshape(-1, C)
    sv = K.stats_values(stats)
    assert _rel(sv[0].cpu(), zf.sum(0)) < 1e-5 and _rel(sv[1].cpu(), (zf * zf).sum(0)) < 1e-5


def test_se_and_rowscale(dev):
    torch.manual_seed(2)
    N, C, Cs, H = 3, 48, 6, 7
    x = torch.randn(N, C, H, H, requires_grad=True)
    w1, b1 = torch.randn(Cs, C, 1, 1, requires_grad=True), torch.randn(Cs, requires_grad=True)
    w2, b2 = torch.randn(C, Cs, 1, 1, requires_grad=True), torch.randn(C, requires_grad=True)
    sc = torch.sigmoid(F.conv2d(F.silu(F.conv2d(F.adaptive_avg_pool2d(x, 1), w1, b1)), w2, b2))
    y = sc * x
    dy = torch.randn_like(y)
    y.backward(dy)
    xd, dyd = _nhwc(x.detach(), torch.float32, dev), _nhwc(dy, torch.float32, dev)
    avg, _ = K.gap_fwd(xd, with_max=False)
    W1, W2 = w1.detach().view(Cs, C).to(dev), w2.detach().view(C, Cs).to(dev)
    h1, u1 = K.linear_fwd(avg, W1, b1.detach().to(dev), K.CS_ACT_SILU, want_preact=True)
    s = K.linear_fwd(h1, W2, b2.detach().to(dev), K.CS_ACT_SIGMOID)
    yd = K.se_scale(xd, s)
    ds = K.se_scale_bwd_ds(dyd, xd)
    dh1, dw2, db2 = K.linear_bwd(h1, W2, ds, s, K.CS_ACT_SIGMOID)
    davg, dw1, db1 = K.linear_bwd(avg, W1, dh1, u1, K.CS_ACT_SILU)
    dxd = K.se_scale_bwd_dx(dyd, s, davg)
    torch.cuda.synchronize()
    assert _rel(_nchw(yd), y.detach()) < 1e-5
    assert _rel(_nchw(dxd), x.grad) < 1e-4
    assert _rel(dw1.cpu(), w1.grad.view(Cs, C)) < 1e-4 and _rel(db1.cpu(), b1.grad) < 1e-4
    assert _rel(dw2.cpu(), w2.grad.view(C, Cs)) < 1e-4 and _rel(db2.cpu(), b2.grad) < 1e-4
    # StochasticDepth row scale + residual
    a, b = torch.randn(N, H, H, C).to(dev), torch.randn(N, H, H, C).to(dev)
    rs = torch.tensor([0.0, 1.25, 1.25], device=dev)
    out = K.rowscale_add(a, rs, b)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), (a * rs.view(-1, 1, 1, 1) + b).cpu()) < 1e-6


def _pair(arch, dev, dtype=torch.float32, sd_prob=0.0):
    m = {"efficientnet_b0": EN.MILefficientnetB0, "efficientnet_b2": EN.MILefficientnetB2, "efficientnet_b3": EN.MILefficientnetB3}[arch](
        stochastic_depth_prob=sd_prob, num_classes=2)
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    osd = {k: v.clone() for k, v in sd.items()}
    for k, v in osd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    return m.to(dev).set_compute_dtype(dtype), osd


# B0 and B2 are the reference's two factories (model/efficientnet.py:417-440); B3 is the same rule at w = 1.2, d = 1.4 (BASELINE config 4)
@pytest.mark.parametrize("arch,size,n", [("efficientnet_b0", 64, 4), ("efficientnet_b2", 64, 2), ("efficientnet_b3", 96, 2)])
def test_efficientnet_tile_and_image_vs_oracle(arch, size, n, dev):
    x = synth.normalise(synth.ihc_tiles(n, size, 51))
    labels = torch.tensor([i % 2 for i in range(n)])
    m, osd = _pair(arch, dev)
    # eval inference
    m.setmode("tile")
    m.eval()
    with torch.no_grad():
        out_eval = m(x.to(dev))
    ref_eval = orc.eff_forward(osd, x, arch, "tile", training=False).detach()
    assert _rel(out_eval.cpu(), ref_eval) < 1e-4
    # --scratch style training step (BN batch statistics: freeze_bn is a no-op for this family)
    m.train()
    m.set_encoder_grads(True)
    loss = HF.cross_entropy(m(x.to(dev), freeze_bn=True), labels.to(dev))
    loss.backward()
    ref = F.cross_entropy(orc.eff_forward(osd, x, arch, "tile", training=True), labels)
    ref.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    # Error per tensor relative to max(|ref tensor|, 1e-3 * largest gradient of the net): the bias of a BN that feeds
    # another train-mode BN has an analytically ZERO gradient (a per-channel constant is normalised away); the reference
    # value is ~1e-7 rounding noise there, so a purely per-tensor relative error would be meaningless.
    gmax = max(float(v.grad.abs().max()) for v in osd.values() if v.grad is not None)
    worst = []
    for k, p in m.named_parameters():
        if p.grad is None or osd[k].grad is None:
            continue
        g, r = p.grad.cpu(), osd[k].grad
        worst.append((float((g - r).abs().max() / max(float(r.abs().max()), 1e-3 * gmax)), k))
    worst.sort(reverse=True)
    # (round 4 had loosened B3 to 2e-2: its batch statistics were added up with fp64 atomics in arrival order, a one-ulp difference is
    # amplified ~100x per 18-row BatchNorm of the last stages and one run in ten landed above 5e-3.  Round 5: the statistics are exact,
    # order-independent sums -- tests/test_determinism_gpu.py -- and the bound is 5e-3 for every network again.)
    assert worst[0][0] < 5e-3, worst[:5]
    assert float(np.median([w for w, _ in worst])) < 1e-3
    bufs = dict(m.named_buffers())
    assert _rel(bufs["features.0.1.running_var"].cpu(), osd["features.0.1.running_var"]) < 1e-5


def test_efficientnet_bf16_and_api(dev):
    m, osd = _pair("efficientnet_b0", dev, torch.bfloat16)
    x = synth.normalise(synth.ihc_tiles(2, 64, 53))
    m.setmode("image")
    m.eval()
    with torch.no_grad():
        c, r = m(x.to(dev))
    rc, rr = orc.eff_forward(osd, x, "efficientnet_b0", "image", training=False)
    assert tuple(c.shape) == (2, 7) and tuple(r.shape) == (2, 1)
    assert _rel(c.float().cpu(), rc.detach()) < 6e-2
    with pytest.raises(Exception, match="Invalid mode"):
        m.setmode("nope")
    m.setmode("segment")
    with pytest.raises(Exception):
        m(x.to(dev))
    # segment mode freezes every group setmode() knows; `classifier` is never toggled (neither does the reference)
    assert sorted(k for k, p in m.named_parameters() if p.requires_grad) == ["classifier.1.bias", "classifier.1.weight"]


def test_stochastic_depth_row_mode_vs_oracle(dev):
    """EfficientNet-B0 with the reference's stochastic_depth_prob = 0.2 (model/efficientnet.py:101-114 on torchvision's
    StochasticDepth(p, "row")): the training forward draws one bernoulli(1 - p) / (1 - p) factor per sample and residual block from
    torch's CUDA generator; re-drawing the same sequence after re-seeding feeds the oracle, so loss and gradients must agree --
    and they must DIFFER from the p = 0 network (some block was actually dropped)."""
    n, arch = 8, "efficientnet_b0"
    x = synth.normalise(synth.ihc_tiles(n, 64, 52))
    labels = torch.tensor([i % 2 for i in range(n)])
    m, osd = _pair(arch, dev, sd_prob=0.2)
    m.setmode("tile")
    m.train()
    m.set_encoder_grads(True)
    torch.manual_seed(1234)
    loss = HF.cross_entropy(m(x.to(dev), freeze_bn=True), labels.to(dev))
    loss.backward()
    # the same draws, in plan order: one per residual block with p > 0
    torch.manual_seed(1234)
    noise, dropped = [], 0
    for mod in m.modules():
        if isinstance(mod, EN.MBConv) and mod.use_res_connect:
            p = mod.stochastic_depth.p
            if p > 0:
                v = torch.empty((n,), dtype=torch.float32, device=dev).bernoulli_(1.0 - p).div_(1.0 - p)
                dropped += int((v == 0).sum())
                noise.append(v.cpu())
            else:
                noise.append(None)
    assert dropped > 0, "the seed must drop at least one (sample, block) pair for the test to mean anything"
    ref = F.cross_entropy(orc.eff_forward(osd, x, arch, "tile", training=True, sd_noise=iter(noise)), labels)
    ref.backward()
    plain = F.cross_entropy(orc.eff_forward({k: v.detach() for k, v in osd.items()}, x, arch, "tile", training=True), labels)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert abs(ref.item() - plain.item()) > 1e-3 * abs(ref.item())
    gmax = max(float(v.grad.abs().max()) for v in osd.values() if v.grad is not None)
    worst = []
    for k, p_ in m.named_parameters():
        if p_.grad is None or osd[k].grad is None:
            continue
        g, r = p_.grad.cpu(), osd[k].grad
        worst.append((float((g - r).abs().max() / max(float(r.abs().max()), 1e-3 * gmax)), k))
    worst.sort(reverse=True)
    assert worst[0][0] < 5e-3, worst[:5]


def test_depthwise_tiled_kernels_everywhere_the_geometry_allows(dev):
    """Three generations of depthwise kernels share the entry points: the strip kernels serve every bf16 3x3 / 5x5 launch, the launch
    rule (csrc/dwse.hip: dw_tiled_geometry) sends part of what is left to the channel-tiled kernels, the rest goes to the
    element-per-thread ones.  CELLSEG_DW_NOSTRIP=1 switches the strips off; CELLSEG_DW_UNTILED=2 then forces the tiled kernels wherever
    the geometry allows, =1 the element-per-thread kernels.  Every combination must pass the same per-op parity test (the knobs are
    read once per process: child interpreters)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("2", "1", "0"):
        env = dict(os.environ, CELLSEG_DW_UNTILED=mode, CELLSEG_DW_NOSTRIP="1", CELLSEG_LIB_FLAVOUR="ab")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_efficientnet_gpu.py"), "-m", "gpu", "-x", "-q",
                            "-k", "test_depthwise_conv"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, (mode, (r.stdout + r.stderr)[-3000:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(3, 19, 19, 144), (2, 10, 10, 2304), (4, 75, 38, 40), (2, 8, 8, 24), (1, 30, 31, 2136)])
def test_per_sample_reductions_of_the_se_block(cfg, dtype, dev):
    """cs_sample_rowsum_ behind cs_gap_avgmax_fwd(with_max = 0) and cs_se_scale_bwd(phase 0): average pool and ds = sum_p dy * x per
    (sample, channel), for channel-group counts below / above one workgroup row (18, 288 > 256, 5, 3, 267)."""
    N, H, W, C = cfg
    torch.manual_seed(C + H)
    x = torch.randn(N, H, W, C).to(dtype)
    dy = torch.randn(N, H, W, C).to(dtype)
    avg, _ = K.gap_fwd(x.to(dev), with_max=False)
    ds = K.se_scale_bwd_ds(dy.to(dev), x.to(dev))
    torch.cuda.synchronize()
    ref_avg = x.double().mean(dim=(1, 2))
    ref_ds = (x.double() * dy.double()).sum(dim=(1, 2))
    assert float((avg.cpu().double() - ref_avg).abs().max()) < 1e-5 * max(1.0, float(ref_avg.abs().max())) + 1e-6
    assert float((ds.cpu().double() - ref_ds).abs().max()) < 2e-5 * float(ref_ds.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4096, 144), (23104, 576), (6400, 2304), (77, 40)])
@pytest.mark.parametrize("act", [K.CS_ACT_NONE, K.CS_ACT_SILU, K.CS_ACT_RELU])
def test_bn_apply_stats_equals_finalize_then_apply(shape, act):
    """cs_bn_apply_stats (finalize folded into the apply pass) gives the bits of cs_bn_finalize + cs_bn_apply: y, the saved mean / rstd
    and the running statistics."""
    dev = torch.device("cuda:0")
    M, C = shape
    g = torch.Generator().manual_seed(M + C)
    z = (torch.randn((M, C), generator=g) * 1.7 + 0.3).to(torch.bfloat16).to(dev)
    res = torch.randn((M, C), generator=g).to(torch.bfloat16).to(dev) if act == K.CS_ACT_RELU else None
    gamma = (torch.rand((C,), generator=g) + 0.5).to(dev)
    beta = (torch.randn((C,), generator=g) * 0.1).to(dev)
    stats = K.bn_stats(z)
    rm0, rv0 = torch.randn((C,), generator=g).to(dev), (torch.rand((C,), generator=g) + 0.5).to(dev)
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    mean, rstd = K.bn_finalize(stats, M, 1e-3, 0.1, rm_a, rv_a)
    y_ref = K.bn_apply(z, mean, rstd, gamma, beta, res, act)
    y, mean2, rstd2 = K.bn_apply_stats(z, stats, 1e-3, 0.1, rm_b, rv_b, gamma, beta, res, act)
    torch.cuda.synchronize()
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))


def test_depthwise_filter_staging_and_parameter_layout_gradient(dev):
    """kernels.DwStagePack (cs_dw_weights_hwc_multi): the filters of several layers, parameter layout [C, 1, R, S] -> [R, S, C], in one launch,
    bit for bit the permute + copy it replaces; cs_dwconv_wgrad_oihw: the weight gradient in the parameter's layout equals the [R, S, C]
    one permuted (model/efficientnet.py:97-103: nn.Conv2d(groups = C))."""
    g = torch.Generator().manual_seed(11)
    ws = [torch.randn((c, 1, r, r), generator=g).to(dev) for c, r in ((40, 3), (144, 3), (288, 5), (8, 5), (2304, 3))]
    pack = K.DwStagePack(ws)
    pack.run()
    torch.cuda.synchronize()
    for w, hwc in zip(ws, pack.hwc):
        assert torch.equal(hwc, w[:, 0].permute(1, 2, 0).contiguous())
    ws[1].mul_(2.0)                      # an optimizer update in place: the next launch restages it
    pack.run()
    assert torch.equal(pack.hwc[1], ws[1][:, 0].permute(1, 2, 0).contiguous())
    for (n, h, c, r, s) in ((4, 19, 144, 5, 1), (3, 20, 48, 3, 2), (2, 10, 816, 5, 1)):
        geom = K.make_geom(n, h, h, c, c, r, r, s, (r - 1) // 2)
        x = torch.randn((n, h, h, c), generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn((n, geom.P, geom.Q, c), generator=g).to(torch.bfloat16).to(dev)
        a = K.dwconv_wgrad(geom, x, dy)
        b = K.dwconv_wgrad(geom, x, dy, param_layout=True)
        assert tuple(b.shape) == (c, 1, r, r)
        assert torch.equal(b, a.permute(2, 0, 1).unsqueeze(1).contiguous())
