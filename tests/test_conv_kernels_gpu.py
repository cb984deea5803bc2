"""GPU parity of the implicit-GEMM convolution family (fwd / dgrad / wgrad) through the C ABI,
against torch's CPU fp32 conv on the same inputs.

Tolerances: fp32 mode 2e-5 relative to max|ref| (exact-f32 MFMA, different summation order);
bf16 mode: inputs are pre-rounded to bf16 so the only differences are fp32 accumulation order and
the final bf16 rounding of the stored result: 1e-2 relative to max|ref|.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import kernels as K  # noqa: E402

TOL = {torch.float32: 2e-5, torch.bfloat16: 1e-2}

#        N   H   W  Cin Cout R  s  p
SHAPES = [
    (2, 19, 19, 64, 256, 1, 1, 0),
    (2, 19, 19, 64, 128, 1, 2, 0),
    (2, 19, 19, 64, 64, 3, 1, 1),
    (3, 10, 10, 128, 128, 3, 1, 1),
    (2, 19, 19, 64, 128, 3, 2, 1),
    (2, 37, 37, 3, 64, 7, 2, 3),
    (2, 13, 13, 24, 40, 3, 1, 1),
    (2, 11, 11, 40, 24, 5, 2, 2),
    (8, 80, 80, 32, 128, 1, 1, 0),    # M=51200: 128x128 tile path
    (8, 80, 80, 16, 64, 3, 1, 1),     # 128x64 tile path
]


def _q(t, dtype):
    return t.to(dtype).float()


def _nhwc(t_nchw, dtype, dev, Cp=None):
    n, c, h, w = t_nchw.shape
    Cp = Cp or K.pad_channels(c)
    out = torch.zeros((n, h, w, Cp), dtype=dtype)
    out[..., :c] = t_nchw.permute(0, 2, 3, 1).to(dtype)
    return out.to(dev)


def _from_nhwc(t, c):
    return t[..., :c].float().cpu().permute(0, 3, 1, 2).contiguous()


def _relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# The register-staged twin (the production rule for operands >= 2 GiB) and the experimental streaming kernel are reachable at test
# sizes only through cs_set_igemm_path, which exists in the A/B flavour of the library (`make AB=1`): the children started by
# test_wave_specialised_weight_gradient_on_every_shape run this file with CELLSEG_LIB_FLAVOUR=ab and get all three paths.
_PATHS = ([(0, "lds_dma"), (1, "reg_staged"), (3, "lds_dma+stream")]
          if os.environ.get("CELLSEG_LIB_FLAVOUR") == "ab" and os.environ.get("CELLSEG_TEST_IGEMM_PATHS", "all") == "all" else [(0, "lds_dma")])


@pytest.fixture(params=[p for p, _ in _PATHS], ids=[i for _, i in _PATHS])
def igemm_path(request):
    if len(_PATHS) == 1:
        yield request.param
        return
    old = K.set_igemm_path(request.param)
    yield request.param
    K.set_igemm_path(old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_fwd_dgrad_wgrad(shape, dtype, dev, igemm_path):
    N, H, W, Cin, Cout, R, s, p = shape
    g = torch.Generator().manual_seed(1234 + Cin * 7 + Cout)
    x = _q(torch.randn((N, Cin, H, W), generator=g), dtype)
    w = _q(torch.randn((Cout, Cin, R, R), generator=g) / (Cin * R * R) ** 0.5, dtype)
    scale = torch.rand((Cout,), generator=g) + 0.5
    shift = torch.randn((Cout,), generator=g) * 0.1
    Cp, Kp = K.pad_channels(Cin), K.pad_channels(Cout)
    geom = K.make_geom(N, H, W, Cp, Kp, R, R, s, p)
    P, Q = geom.P, geom.Q
    res = _q(torch.randn((N, Cout, P, Q), generator=g), dtype)

    # ---------- forward: relu(scale*conv + shift + res)
    ref = F.conv2d(x, w, stride=s, padding=p)
    ref_full = torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd = _nhwc(x, dtype, dev)
    w_khwc, w_chwk = K.weight_prep(w.to(dev), None, dtype, Cp, Kp, want_fwd=True, want_bwd=True)
    sc = torch.ones(Kp); sc[:Cout] = scale
    sh = torch.zeros(Kp); sh[:Cout] = shift
    y_plain = K.conv_fwd(geom, xd, w_khwc)
    y_full = K.conv_fwd(geom, xd, w_khwc, sc.to(dev), sh.to(dev), _nhwc(res, dtype, dev, Kp), K.CS_ACT_RELU)
    torch.cuda.synchronize()
    e1 = _relerr(_from_nhwc(y_plain, Cout), ref)
    e2 = _relerr(_from_nhwc(y_full, Cout), ref_full)
    assert e1 < TOL[dtype], f"fwd plain relerr {e1}"
    assert e2 < TOL[dtype], f"fwd fused relerr {e2}"
    if Kp != Cout:
        assert float(y_plain[..., Cout:].float().abs().max()) == 0.0

    # ---------- dgrad (needs Cin chunk-aligned rows in w_chwk)
    dy = _q(torch.randn((N, Cout, P, Q), generator=g), dtype)
    dyd = _nhwc(dy, dtype, dev, Kp)
    if Cin % 8 == 0:
        ref_dx = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, stride=s, padding=p)
        add = _q(torch.randn((N, Cin, H, W), generator=g), dtype)
        mask = _q(torch.randn((N, Cin, H, W), generator=g), dtype)
        ref_dx2 = (ref_dx + add) * (mask > 0)
        dx = K.conv_dgrad(geom, dyd, w_chwk)
        cs = torch.zeros((Cp,), dtype=torch.float32, device=dev)
        dx2 = K.conv_dgrad(geom, dyd, w_chwk, _nhwc(add, dtype, dev), _nhwc(mask, dtype, dev), cs)
        torch.cuda.synchronize()
        e3 = _relerr(_from_nhwc(dx, Cin), ref_dx)
        e4 = _relerr(_from_nhwc(dx2, Cin), ref_dx2)
        assert e3 < TOL[dtype], f"dgrad relerr {e3}"
        assert e4 < TOL[dtype], f"dgrad fused relerr {e4}"
        ref_cs = dx2[..., :Cin].float().sum(dim=(0, 1, 2)).cpu()
        e5 = _relerr(cs[:Cin].cpu(), ref_cs)
        assert e5 < 1e-3, f"dgrad colsum relerr {e5}"

    # ---------- wgrad (both LDS operand paths for bf16)
    ref_dw = torch.nn.grad.conv2d_weight(x, (Cout, Cin, R, R), dy, stride=s, padding=p)
    for use_tr in ([False, True] if dtype == torch.bfloat16 else [False]):
        raw = K.new_wgrad_buffer(geom, dev)
        raw.fill_(float("nan"))                      # every slab element must be overwritten by the kernel
        K.conv_wgrad(geom, xd, dyd, raw, use_tr_read=use_tr)
        dw = torch.empty((Cout, Cin, R, R), dtype=torch.float32, device=dev)
        gsum = K.colsum(dyd)
        dbias = torch.empty((Cout,), dtype=torch.float32, device=dev)
        K.wgrad_finalize(raw, None, None, None, None, gsum, Cin, dw, dbias=dbias)
        torch.cuda.synchronize()
        e6 = _relerr(dw.cpu(), ref_dw)
        assert e6 < (1e-4 if dtype == torch.float32 else 2e-3), f"wgrad(tr={use_tr}) relerr {e6}"
        e7 = _relerr(dbias.cpu(), dy.sum(dim=(0, 2, 3)))
        assert e7 < 1e-4, f"colsum/dbias relerr {e7}"


def test_bn_fold_and_finalize_bn_grads(dev):
    """eval-BN folded conv: dgamma/dbeta from the raw weight gradient (no saved conv output)."""
    torch.manual_seed(7)
    N, C, Kc, H = 2, 16, 24, 9
    x = torch.randn(N, C, H, H)
    w = torch.randn(Kc, C, 3, 3, requires_grad=True)
    gamma = (torch.rand(Kc) + 0.5).requires_grad_()
    beta = torch.randn(Kc).requires_grad_()
    mean, var, eps = torch.randn(Kc) * 0.1, torch.rand(Kc) + 0.5, 1e-5
    z = F.conv2d(x, w, padding=1)
    u = F.batch_norm(z, mean, var, gamma, beta, False, 0.0, eps)
    g = torch.randn_like(u)
    u.backward(g)
    scale, shift, rstd = K.bn_fold(gamma.detach().to(dev), beta.detach().to(dev), mean.to(dev), var.to(dev), eps)
    torch.cuda.synchronize()
    ref_scale = gamma.detach() / torch.sqrt(var + eps)
    assert _relerr(scale.cpu(), ref_scale) < 1e-6
    assert _relerr(shift.cpu(), beta.detach() - mean * ref_scale) < 1e-6
    geom = K.make_geom(N, H, H, C, Kc, 3, 3, 1, 1)
    xd = _nhwc(x, torch.float32, dev)
    gd = _nhwc(g, torch.float32, dev)
    raw = K.new_wgrad_buffer(geom, dev)
    K.conv_wgrad(geom, xd, gd, raw)
    gsum = K.colsum(gd)
    dw = torch.empty((Kc, C, 3, 3), device=dev)
    dgamma = torch.empty((Kc,), device=dev)
    dbeta = torch.empty((Kc,), device=dev)
    K.wgrad_finalize(raw, w.detach().to(dev), scale, rstd, mean.to(dev), gsum, C, dw, dgamma=dgamma, dbeta=dbeta)
    torch.cuda.synchronize()
    assert _relerr(dw.cpu(), w.grad) < 1e-4
    assert _relerr(dgamma.cpu(), gamma.grad) < 1e-4
    assert _relerr(dbeta.cpu(), beta.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 19, 19, 128, 32, 1), (2, 19, 19, 128, 32, 2), (1, 10, 10, 256, 32, 1), (1, 9, 9, 256, 8, 2)])
def test_grouped_conv_slab_dense(cfg, dtype, dev, igemm_path):
    """ResNeXt grouped 3x3 (groups=32 / widths 128, 256; and a Cg=32 case) vs F.conv2d(groups=...)."""
    N, H, W, C, G, s = cfg
    g = torch.Generator().manual_seed(C + G + s)
    x = _q(torch.randn((N, C, H, W), generator=g), dtype).requires_grad_()
    w = _q(torch.randn((C, C // G, 3, 3), generator=g) / (9 * C / G) ** 0.5, dtype).requires_grad_()
    y = F.conv2d(x, w, None, s, 1, 1, G)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    geom = K.make_geom(N, H, W, C, C, 3, 3, s, 1)
    xd, dyd = _nhwc(x.detach(), dtype, dev), _nhwc(dy, dtype, dev)
    wk, wc = K.weight_prep_grouped(w.detach().to(dev), None, dtype, True, True)
    yd = K.conv_fwd(geom, xd, wk, grouped=True)
    dxd = K.conv_dgrad(geom, dyd, wc, grouped=True)
    raw = K.new_wgrad_buffer(geom, dev, grouped=True)
    K.conv_wgrad(geom, xd, dyd, raw, grouped=True)
    dw = torch.empty_like(w.detach()).to(dev)
    K.wgrad_finalize_grouped(raw, None, None, None, None, None, dw)
    torch.cuda.synchronize()
    assert _relerr(_from_nhwc(yd, C), y.detach()) < TOL[dtype]
    assert _relerr(_from_nhwc(dxd, C), x.grad) < TOL[dtype]
    assert _relerr(dw.cpu(), w.grad) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(37, 41), (32, 32), (75, 64)])
def test_paired_stem_equals_the_generic_7x7_path(dev, dtype, hw):
    """cs_stem_* (two pixels x 4 channels per chunk, 7x4 taps) against cs_conv2d_fwd / _wgrad on the same NHWC8 image: odd and
    even widths, the last pixel of an odd row paired with a zero."""
    from cellsegmentation_amd import kernels as K
    H, W = hw
    N, Kc = 3, 64
    g = K.make_geom(N, H, W, 8, Kc, 7, 7, 2, 3)
    assert K.is_stem_geom(g)
    gen = torch.Generator().manual_seed(5)
    x = torch.zeros((N, H, W, 8))
    x[..., :3] = torch.randn((N, H, W, 3), generator=gen)
    x = x.to(dev).to(dtype)
    w = (torch.randn((Kc, 3, 7, 7), generator=gen) * 0.1).to(dev)
    shift = torch.randn((Kc,), generator=gen).to(dev)
    wk, _ = K.weight_prep(w, None, dtype, 8, Kc, True, False)
    y_ref = K.conv_fwd(g, x, wk, None, shift, None, K.CS_ACT_RELU)
    xp, wp = K.stem_pair_input(x), K.stem_pair_weights(wk)
    assert tuple(xp.shape) == (N, H, (W + 1) // 2, 8) and tuple(wp.shape) == (Kc, 7, 4, 8)
    y = K.stem_fwd(g, xp, wp, None, shift, K.CS_ACT_RELU)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((y.float() - y_ref.float()).abs().max()) <= tol * max(1.0, float(y_ref.float().abs().max()))
    if dtype == torch.bfloat16:
        # the same forward on the ring kernel of conv_v2.hip (gathered operand rows, packed [K][256] weights), with sign bits
        assert K.stem_fwd_packed_supported(g, dtype)
        y2, bits = K.stem_fwd_packed(g, xp, K.stem_pack_weights(wp), shift, K.CS_ACT_RELU, want_bits=True)
        torch.cuda.synchronize()
        assert float((y2.float() - y_ref.float()).abs().max()) <= tol * max(1.0, float(y_ref.float().abs().max()))
        assert torch.equal(K.unpack_bits(bits, Kc), y2 > 0)
    # batch statistics through the same entry point
    st_ref, st = K.new_stats(Kc, dev), K.new_stats(Kc, dev)
    K.conv_fwd(g, x, wk, None, shift, None, K.CS_ACT_NONE, stats=st_ref)
    K.stem_fwd(g, xp, wp, None, shift, K.CS_ACT_NONE, stats=st)
    assert torch.allclose(K.stats_values(st), K.stats_values(st_ref), rtol=1e-3 if dtype == torch.bfloat16 else 1e-6, atol=1e-3)
    # weight gradient
    dy = torch.randn((N, g.P, g.Q, Kc), generator=gen).to(dev).to(dtype)
    raw_ref = K.new_wgrad_buffer(g, dev)
    K.conv_wgrad(g, x, dy, raw_ref)
    ref = raw_ref.sum(0)
    raw = K.stem_wgrad(g, xp, dy)
    assert tuple(raw.shape) == (1, Kc, 7, 7, 8)
    scale = float(ref.abs().max())
    assert float((raw[0] - ref).abs().max()) <= (1e-4 if dtype == torch.float32 else 1e-3) * scale
    assert float(raw[0][..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("K_,Cin,R,Cp,nsplit,n", [
    (64, 64, 1, 64, 512, 1),        # layer1 64->64 1x1: one tile, split ~500 ways (16 float4 items per row, 16 split lanes)
    (256, 64, 1, 64, 64, 4),        # layer1 64->256 x4
    (64, 64, 3, 64, 126, 3),        # layer1 3x3 x3: 1024-thread workgroups, 144 items
    (64, 3, 7, 8, 37, 1),           # stem: Cin 3 padded to 8, 49 taps
    (512, 2048, 1, 2048, 4, 2),     # layer4 2048->512: 512 items per row
    (256, 256, 3, 256, 5, 5),       # layer3 3x3 x5: 256-thread workgroups, two channel tiles
    (40, 24, 3, 24, 9, 2),          # odd sizes (Cin not a multiple of 4 is not served by the packed path, 24 is)
    (24, 6, 1, 8, 3, 1),            # Cin % 4 != 0: scalar tail of the 1x1 path
    (1024, 256, 1, 256, 6, 6),      # layer3 256->1024 x6
    (130, 96, 1, 96, 3, 2),         # K, Cin not multiples of 64
    (66, 72, 3, 72, 2, 2),          # 3x3: a full and an 8-channel tile
])
def test_wgrad_finalize_batched_matches_torch(K_, Cin, R, Cp, nsplit, n, dev):
    """cs_wgrad_finalize_batched (one launch per group of identical layers): dw[k][c][r][s] = scale[k] * sum over the split-K slabs,
    dgamma = rstd * (<w, raw> - mean * gsum), dbeta = gsum, for every access shape of the float4 / split-lane rewrite of round 3."""
    torch.manual_seed(K_ + Cin + R)
    slabs = torch.randn(n, nsplit, K_, R, R, Cp)
    slabs[..., Cin:] = 7.0                                   # padding channels hold garbage: must never reach dw
    ws = [torch.randn(K_, Cin, R, R) for _ in range(n)]
    scales = [torch.rand(K_) + 0.5 for _ in range(n)]
    rstds = [torch.rand(K_) + 0.5 for _ in range(n)]
    means = [torch.randn(K_) for _ in range(n)]
    gsums = [torch.randn(K_) for _ in range(n)]
    d = lambda lst: [t.to(dev) for t in lst]
    dws = [torch.full((K_, Cin, R, R), float("nan"), device=dev) for _ in range(n)]
    dgs = [torch.full((K_,), float("nan"), device=dev) for _ in range(n)]
    dbs = [torch.full((K_,), float("nan"), device=dev) for _ in range(n)]
    K.wgrad_finalize_batched(slabs.to(dev), d(ws), d(scales), d(rstds), d(means), d(gsums), dws, dgs, dbs, Cin)
    torch.cuda.synchronize()
    for i in range(n):
        raw = slabs[i].double().sum(0)[..., :Cin].permute(0, 3, 1, 2)        # [K][Cin][R][S]
        ref_dw = scales[i].double().view(-1, 1, 1, 1) * raw
        ref_dg = rstds[i].double() * ((ws[i].double() * raw).sum((1, 2, 3)) - means[i].double() * gsums[i].double())
        tol = 2e-5 * (1 + nsplit ** 0.5)
        assert float((dws[i].cpu().double() - ref_dw).abs().max()) < tol * float(ref_dw.abs().max())
        assert float((dgs[i].cpu().double() - ref_dg).abs().max()) < tol * float(ref_dg.abs().max()) + 1e-4
        assert torch.equal(dbs[i].cpu(), gsums[i])
    # without BN parameters (plain conv weights): dw only
    dws2 = [torch.full((K_, Cin, R, R), float("nan"), device=dev) for _ in range(n)]
    K.wgrad_finalize_batched(slabs.to(dev), None, None, None, None, None, dws2, None, None, Cin)
    torch.cuda.synchronize()
    for i in range(n):
        raw = slabs[i].double().sum(0)[..., :Cin].permute(0, 3, 1, 2)
        assert float((dws2[i].cpu().double() - raw).abs().max()) < 2e-5 * (1 + nsplit ** 0.5) * float(raw.abs().max())


@pytest.mark.parametrize("K_,Cin,R,nsplit,n,rows", [(256, 64, 1, 64, 4, 2813), (1024, 256, 1, 6, 6, 181), (256, 256, 3, 5, 5, 241),
                                                   (64, 64, 3, 126, 3, 5626), (130, 96, 1, 3, 2, 77)])
def test_wgrad_finalize_batched_folds_partial_column_sums(K_, Cin, R, nsplit, n, rows, dev):
    """The batched finalize takes the column sums of a data gradient as per-workgroup partial rows [rows][2][n_out] (first half of a
    row) and folds them itself."""
    torch.manual_seed(K_ + Cin + rows)
    slabs = torch.randn(n, nsplit, K_, R, R, Cin)
    ws = [torch.randn(K_, Cin, R, R) for _ in range(n)]
    scales = [torch.rand(K_) + 0.5 for _ in range(n)]
    rstds = [torch.rand(K_) + 0.5 for _ in range(n)]
    means = [torch.randn(K_) for _ in range(n)]
    n_out = (K_ + 7) // 8 * 8
    parts = [torch.randn(rows, 2, n_out) for _ in range(n)]
    d = lambda lst: [t.to(dev) for t in lst]
    gs = [K.PartialColsum(p.to(dev), rows, n_out) for p in parts]
    dws = [torch.empty((K_, Cin, R, R), device=dev) for _ in range(n)]
    dgs = [torch.empty((K_,), device=dev) for _ in range(n)]
    dbs = [torch.empty((K_,), device=dev) for _ in range(n)]
    K.wgrad_finalize_batched(slabs.to(dev), d(ws), d(scales), d(rstds), d(means), gs, dws, dgs, dbs, Cin)
    torch.cuda.synchronize()
    for i in range(n):
        gsum = parts[i][:, 0, :K_].double().sum(0)
        raw = slabs[i].double().sum(0).permute(0, 3, 1, 2)
        ref_dg = rstds[i].double() * ((ws[i].double() * raw).sum((1, 2, 3)) - means[i].double() * gsum)
        tol = 2e-5 * (1 + nsplit ** 0.5)
        assert float((dbs[i].cpu().double() - gsum).abs().max()) < 1e-5 * rows ** 0.5 * (1 + float(gsum.abs().max()))
        assert float((dgs[i].cpu().double() - ref_dg).abs().max()) < tol * float(ref_dg.abs().max()) + 1e-3 * rows ** 0.5


def test_wave_specialised_weight_gradient_on_every_shape(dev):
    """wgrad_spec_kernel (4 loader + 4 consumer waves) is selected by rule for the deep layers only; CELLSEG_WGRAD_SPEC=1 forces it for
    every LDS-DMA weight gradient, =2 forces the four-wave kernel.  Both extremes must pass this file's parity tests and the exact
    integer-data test of the packed kernels' file (the knob is read once per process: child interpreters)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("1", "2"):
        # (the first child also walks the three staging paths of the first-generation kernels; the grandchildren of the packed file's own
        # forced-mode test and its bag-size sweep are left to the parent run)
        env = dict(os.environ, CELLSEG_WGRAD_SPEC=mode, CELLSEG_LIB_FLAVOUR="ab", CELLSEG_TEST_IGEMM_PATHS="all" if mode == "1" else "dma")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_conv_kernels_gpu.py"),
                            os.path.join(root, "tests", "test_conv_packed_gpu.py"), "-m", "gpu", "-x", "-q",
                            "-k", "not wave_specialised and not wide_kernel_forced and not first_generation_over"],
                           capture_output=True, text=True, timeout=1200, env=env, cwd=root)
        assert r.returncode == 0, (mode, (r.stdout + r.stderr)[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(70, 100, 3, 72, 104), (600, 520, 3, 608, 520), (512, 1024, 3, 512, 1024), (128, 96, 1, 128, 96), (40, 328, 5, 40, 328), (2048, 1024, 1, 2048, 1024), (2050, 1030, 1, 2056, 1032)])
def test_weight_prep_tiled_layouts(shape, dev):
    """The LDS-tiled bf16 weight staging (prep.hip weight_prep_tiled_kernel) against a permute: both layouts, padded channels zero,
    per-filter scale applied before the rounding."""
    Kc, Cin, R, Kp, Cp = shape
    g = torch.Generator().manual_seed(Kc * 13 + Cin)
    w = torch.randn((Kc, Cin, R, R), generator=g)
    scale = torch.rand((Kc,), generator=g) + 0.5
    for sc in (None, scale):
        wk, wc = K.weight_prep(w.to(dev), None if sc is None else sc.to(dev), torch.bfloat16, Cp, Kp, want_fwd=True, want_bwd=True)
        ws = w if sc is None else w * sc.view(-1, 1, 1, 1)
        ref_k = torch.zeros((Kp, R, R, Cp), dtype=torch.bfloat16)
        ref_k[:Kc, :, :, :Cin] = ws.permute(0, 2, 3, 1).to(torch.bfloat16)
        ref_c = torch.zeros((Cp, R, R, Kp), dtype=torch.bfloat16)
        ref_c[:Cin, :, :, :Kc] = ws.permute(1, 2, 3, 0).to(torch.bfloat16)
        assert torch.equal(wk.cpu().view(Kp, R, R, Cp), ref_k)
        assert torch.equal(wc.cpu().view(Cp, R, R, Kp), ref_c)


@pytest.mark.parametrize("shape", [(2, 19, 19, 64, 128, 1), (8, 38, 38, 128, 144, 1), (64, 38, 38, 32, 192, 1), (3, 21, 17, 64, 64, 3)])
def test_conv_fwd_batch_statistics_atomic_and_slab_paths(shape, dev):
    """cs_conv2d_fwd with `stats`: per-channel sum / sum of squares of the STORED bf16 output.  Launches with <= 512 pixel tiles add them
    with fp64 atomics from the epilogue, larger ones leave partial rows for slab_reduce (the third shape: 722 tiles); both must match the
    sums of the stored tensor."""
    N, H, W, Cin, Cout, R = shape
    g_ = torch.Generator().manual_seed(H + Cin + Cout)
    x = torch.randn((N, H, W, Cin), generator=g_).to(torch.bfloat16).to(dev)
    w = (torch.randn((Cout, Cin, R, R), generator=g_) / (Cin * R * R) ** 0.5).to(dev)
    geom = K.make_geom(N, H, W, Cin, Cout, R, R, 1, R // 2)
    wk, _ = K.weight_prep(w, None, torch.bfloat16, Cin, Cout, True, False)
    stats = K.new_stats(Cout, dev)
    y = K.conv_fwd(geom, x, wk, None, None, None, K.CS_ACT_NONE, stats=stats)
    torch.cuda.synchronize()
    yd = y.double().view(-1, Cout)
    sv = K.stats_values(stats)
    assert torch.allclose(sv[0].cpu(), yd.sum(0).cpu(), rtol=1e-6, atol=1e-4)
    assert torch.allclose(sv[1].cpu(), (yd * yd).sum(0).cpu(), rtol=1e-6, atol=1e-4)
