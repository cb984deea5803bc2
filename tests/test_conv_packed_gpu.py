"""GPU parity of the packed-operand bf16 convolutions (csrc/conv_v2.hip: halo-tile 3x3 forward / data gradient) through the C
ABI, against torch's CPU fp32 conv on the same bf16-rounded inputs (model/resnet.py:20,23,53 are the layers served).

Tolerance: inputs are pre-rounded to bf16, so the differences are fp32 accumulation order and the final bf16 rounding of the stored
result: 1e-2 relative to max|ref| (the bound of tests/test_conv_kernels_gpu.py).  Bit tensors and masks are exact.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
TOL = 1e-2

#        N   H   W  Cin Cout R  s  pad
SHAPES = [
    (2, 19, 19, 64, 128, 3, 1, 1),      # one chunk, window blocks: cold start + last tap only
    (3, 10, 10, 128, 128, 3, 1, 1),     # two chunks, tiles straddle images (100 px per image)
    (2, 38, 38, 128, 256, 3, 1, 1),     # two N tiles
    (1, 19, 19, 256, 128, 3, 1, 1),     # four chunks: both stages, odd/even chunk bodies
    (2, 75, 75, 64, 64, 3, 1, 1),       # 256 x 64 tile configuration (2 x 2 waves), M tail
    (5, 7, 9, 64, 128, 3, 1, 1),        # non-square, tiny images
    (2, 12, 12, 64, 128, 3, 1, 0),      # valid convolution: the data gradient's source is smaller than its destination
    (2, 9, 9, 64, 128, 3, 1, 2),        # pad 2: output larger than input
    (3, 21, 17, 192, 128, 3, 1, 1),     # three chunks
    (1, 150, 150, 128, 128, 3, 1, 1),   # wide image (the 150 x 150 decoder layers): 6-7 window blocks per wave, two stages, one workgroup per CU
    (1, 131, 97, 256, 128, 3, 1, 1),    # the same with four chunks, non-square
    # the wide kernel (one wave per SIMD, round 4) is picked where all of its tiles are resident at once; the small shapes above take
    # its 128-pixel form, these two its 192- and 256-pixel forms (the forced-mode test below runs every shape on every form)
    (64, 19, 19, 256, 256, 3, 1, 1),    # ResNet-50 layer3 at the bench's size: 242 tiles of 192 px, 5 window blocks per wave
    (40, 38, 38, 128, 128, 3, 1, 1),    # 226 tiles of 256 px, 7 window blocks per wave
    # 1x1: the persistent ring kernel (any contraction depth): (pixel tile, 64-channel chunk) steps over three LDS slots
    (2, 19, 19, 64, 256, 1, 1, 0),      # one chunk (an epilogue every step); data gradient: four chunks into 64 channels (2 x 2 waves)
    (3, 21, 17, 256, 64, 1, 1, 0),      # four chunks into 64 channels; data gradient: one chunk, 1 x 4 waves
    (2, 38, 38, 128, 512, 1, 1, 0),     # two chunks, four N tiles; data gradient: eight chunks
    (2, 19, 19, 192, 128, 1, 1, 0),     # three chunks
    (2, 75, 75, 64, 64, 1, 1, 0),       # 64-channel tiles, many pixel tiles per workgroup
    (2, 38, 38, 128, 64, 1, 1, 0),      # two chunks, 64-channel tiles
    (2, 37, 41, 256, 128, 1, 2, 0),     # the strided down-sampling 1x1 (its data gradient is compact: test_compact_strided_gradient)
    (3, 10, 10, 2048, 512, 1, 1, 0),    # 32 chunks (layer4 conv1); data gradient: 8 chunks into 2048 channels
    (2, 19, 19, 1024, 256, 1, 1, 0),    # 16 chunks
    (64, 19, 19, 64, 128, 1, 1, 0),     # 181 pixel tiles: several per workgroup, tail tile, re-armed operands
    (40, 23, 23, 128, 64, 1, 1, 0),     # the same for the 2 x 2 wave layout, two chunks
    (1, 5, 5, 320, 128, 1, 1, 0),       # a single, partial pixel tile; five chunks
]
if os.environ.get("CELLSEG_TEST_ONLY_3X3"):        # the forced-mode children of test_wide_kernel_forced_on_every_shape: what the wide kernel can serve
    SHAPES = [s for s in SHAPES if s[5] == 3 or (s[5] == 1 and s[6] == 1 and (s[3] >= 512 or s[4] >= 512))]


def _q(t):
    return t.to(BF).float()


def _nhwc(t_nchw, dev):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(BF).to(dev)


def _from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _unpack_bits(bits, C):
    """a bit plane of the library (channel-block-major, kernels.unpack_bits) -> bool [N,C,H,W]"""
    return K.unpack_bits(bits, C).cpu().permute(0, 3, 1, 2)


def _pack_bits(mask_nchw, dev):
    """bool [N,C,H,W] -> a bit plane in the library's layout (kernels.pack_bits)"""
    return K.pack_bits(mask_nchw.permute(0, 2, 3, 1).contiguous().to(dev))


@pytest.mark.parametrize("shape", SHAPES)
def test_packed_fwd_and_dgrad(shape, dev):
    N, H, W, Cin, Cout, R, stride, pad = shape
    g = torch.Generator().manual_seed(77 + H * 3 + Cin + Cout + R)
    x = _q(torch.randn((N, Cin, H, W), generator=g))
    w = _q(torch.randn((Cout, Cin, R, R), generator=g) / (Cin * R * R) ** 0.5)
    geom = K.make_geom(N, H, W, Cin, Cout, R, R, stride, pad)
    P, Q = geom.P, geom.Q
    w_khwc, w_chwk = K.weight_prep(w.to(dev), None, BF, Cin, Cout, want_fwd=True, want_bwd=True)

    # ---------- forward
    if K.packed_supported(geom, BF, dgrad=False):
        shift = torch.randn((Cout,), generator=g) * 0.1
        res = _q(torch.randn((N, Cout, P, Q), generator=g))
        ref = F.conv2d(x, w, stride=stride, padding=pad)
        ref_full = torch.relu(ref + shift.view(1, -1, 1, 1) + res)
        xd = _nhwc(x, dev)
        wp = K.pack_conv_weights(geom, w_khwc, dgrad=False)
        y_plain = K.conv_fwd_packed(geom, xd, wp)
        y_full, bits = K.conv_fwd_packed(geom, xd, wp, shift.to(dev), _nhwc(res, dev), K.CS_ACT_RELU, want_bits=True)
        torch.cuda.synchronize()
        assert _relerr(_from_nhwc(y_plain), ref) < TOL
        assert _relerr(_from_nhwc(y_full), ref_full) < TOL
        # the bit tensor is exactly "stored value > 0"
        assert torch.equal(_unpack_bits(bits, Cout), _from_nhwc(y_full) > 0)
        # ... and the legacy kernel agrees to rounding (same operands, different summation order)
        y_old = K.conv_fwd(geom, xd, w_khwc)
        torch.cuda.synchronize()
        assert _relerr(_from_nhwc(y_plain), _from_nhwc(y_old)) < TOL
    else:
        assert Cout % 64 != 0 or Cin % 64 != 0 or (Cout == 64 and Cin > 64 and R == 3) , f"{shape}: forward unexpectedly not served"

    # ---------- data gradient
    if stride == 1 and K.packed_supported(geom, BF, dgrad=True):
        dy = _q(torch.randn((N, Cout, P, Q), generator=g))
        add = _q(torch.randn((N, Cin, H, W), generator=g))
        mask = torch.rand((N, Cin, H, W), generator=g) > 0.4
        ref_dx = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, stride=stride, padding=pad)
        ref_dx2 = (ref_dx + add) * mask
        dyd = _nhwc(dy, dev)
        wpd = K.pack_conv_weights(geom, w_chwk, dgrad=True)
        dx = K.conv_dgrad_packed(geom, dyd, wpd)
        dx2, pc = K.conv_dgrad_packed(geom, dyd, wpd, add=_nhwc(add, dev), mask_bits=_pack_bits(mask, dev), want_colsum=True)
        cs = pc.vector()
        torch.cuda.synchronize()
        assert _relerr(_from_nhwc(dx), ref_dx) < TOL
        assert _relerr(_from_nhwc(dx2), ref_dx2) < TOL
        assert bool((_from_nhwc(dx2)[~mask] == 0).all()), "masked elements must be stored as exact zeros"
        ref_cs = dx2.float().sum(dim=(0, 1, 2)).cpu()
        assert _relerr(cs.cpu(), ref_cs) < 1e-3


def test_packed_declines_what_it_does_not_serve(dev):
    assert not K.packed_supported(K.make_geom(2, 19, 19, 64, 128, 3, 3, 2, 1), BF)          # stride 2
    assert not K.packed_supported(K.make_geom(2, 19, 19, 128, 64, 1, 1, 3, 0), BF, dgrad=True)      # 1x1 data gradient with stride 3
    assert not K.packed_supported(K.make_geom(2, 19, 19, 24, 128, 3, 3, 1, 1), BF)          # channels not a multiple of 64
    assert not K.packed_supported(K.make_geom(2, 19, 19, 64, 128, 3, 3, 1, 1), torch.float32)
    assert not K.packed_supported(K.make_geom(2, 1, 1, 64, 128, 3, 3, 1, 1), BF)            # 1x1 images (32x32 tiles at layer4)
    geom = K.make_geom(2, 19, 19, 24, 128, 3, 3, 1, 1)
    x = torch.zeros((2, 19, 19, 24), dtype=BF, device=dev)
    with pytest.raises(RuntimeError, match="not served"):
        K.conv_fwd_packed(geom, x, torch.zeros((16,), dtype=BF, device=dev))


# (the last shape: conv1 of layer4.0 -- 8 chunks into 1024 channels, served by the wide kernel's 1x1 form; it must take the compact operand too)
@pytest.mark.parametrize("shape", [(2, 37, 41, 256, 128, 64), (3, 20, 20, 128, 256, 128), (64, 10, 10, 512, 1024, 256), (1, 7, 9, 64, 64, 64),
                                   (2, 19, 19, 1024, 2048, 512)])
def test_compact_strided_gradient(shape, dev):
    """The block input of a down-sampling bottleneck (model/resnet.py:183 + :51) receives two gradients: the stride-2 1x1 shortcut's
    (non-zero at even pixels only) and the stride-1 conv1's.  The shortcut's data gradient is written in compact [N][P][Q][C] form and
    conv1's data gradient adds it at the even pixels (add_stride = 2), masks and sums: the result must equal the dense computation."""
    N, H, W, C, Ks, Kv = shape
    g = torch.Generator().manual_seed(3 + H + C)
    ws = _q(torch.randn((Ks, C, 1, 1), generator=g) / C ** 0.5)          # shortcut, stride 2
    wv = _q(torch.randn((Kv, C, 1, 1), generator=g) / C ** 0.5)          # conv1, stride 1
    gs = K.make_geom(N, H, W, C, Ks, 1, 1, 2, 0)
    gv = K.make_geom(N, H, W, C, Kv, 1, 1, 1, 0)
    assert K.packed_supported(gs, BF, dgrad=True) and K.packed_supported(gv, BF, dgrad=True)
    dys = _q(torch.randn((N, Ks, gs.P, gs.Q), generator=g))
    dyv = _q(torch.randn((N, Kv, H, W), generator=g))
    mask = torch.rand((N, C, H, W), generator=g) > 0.4
    ref_s = torch.nn.grad.conv2d_input((N, C, H, W), ws, dys, stride=2, padding=0)
    ref_v = torch.nn.grad.conv2d_input((N, C, H, W), wv, dyv, stride=1, padding=0)
    _, ws_chwk = K.weight_prep(ws.to(dev), None, BF, C, Ks, want_fwd=False, want_bwd=True)
    _, wv_chwk = K.weight_prep(wv.to(dev), None, BF, C, Kv, want_fwd=False, want_bwd=True)
    comp = K.conv_dgrad_packed(gs, _nhwc(dys, dev), K.pack_conv_weights(gs, ws_chwk, dgrad=True))
    assert isinstance(comp, K.CompactGrad) and tuple(comp.t.shape) == (N, gs.P, gs.Q, C)
    torch.cuda.synchronize()
    assert _relerr(_from_nhwc(comp.t), ref_s[:, :, ::2, ::2]) < TOL
    assert float(ref_s[:, :, 1::2, :].abs().max()) == 0.0 and float(ref_s[:, :, :, 1::2].abs().max()) == 0.0
    dx, pc = K.conv_dgrad_packed(gv, _nhwc(dyv, dev), K.pack_conv_weights(gv, wv_chwk, dgrad=True), add=comp, mask_bits=_pack_bits(mask, dev),
                                 want_colsum=True)
    cs = pc.vector()
    torch.cuda.synchronize()
    # the compact operand was rounded to bf16 once more than the dense fp32 sum: same tolerance, relative to max|ref|
    ref = (ref_v + _q(ref_s)) * mask
    got = _from_nhwc(dx)
    assert _relerr(got, ref) < TOL
    assert bool((got[~mask] == 0).all())
    assert _relerr(cs.cpu(), dx.float().sum(dim=(0, 1, 2)).cpu()) < 1e-3
    with pytest.raises(ValueError, match="compact"):
        K.conv_dgrad_packed(gs, _nhwc(dys, dev), K.pack_conv_weights(gs, ws_chwk, dgrad=True), mask_bits=_pack_bits(mask, dev))


@pytest.mark.parametrize("shape", [(128, 64, 3), (64, 256, 1), (256, 128, 3), (512, 2048, 1), (64, 64, 3)])
def test_one_launch_staging_writes_the_packed_operands_bit_exactly(shape, dev):
    """cs_stage_conv_bn_multi (LDS-tiled path for unpadded packed layers) against the two-step route: fold + stage with
    cs_stage_conv_bn, then cs_pack_conv_weights -- the same arithmetic, so the packed operands must be identical bit for bit."""
    Kc, C, R = shape
    torch.manual_seed(Kc + C + R)
    conv = torch.nn.Conv2d(C, Kc, R, 1, R // 2, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(Kc).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    pack = K.StagePack([(conv, bn, C, Kc, True, True, True), (conv, bn, C, Kc, True, False, False)], BF)
    pack.launch()
    w_f, w_b, scale, shift, rstd = pack.staged[0]
    u_f, u_b = pack.staged[1][:2]                                      # the plain [k][tap][c] / [c][tap][k] operands of the same layer
    geom = K.make_geom(2, 19, 19, C, Kc, R, R, 1, R // 2)
    scale_ref = bn.weight.detach() * (1.0 / torch.sqrt(bn.running_var + bn.eps))          # the kernels' own expression
    w_khwc, w_chwk = K.weight_prep(conv.weight.detach(), scale_ref, BF, C, Kc, want_fwd=True, want_bwd=True)
    ref_f = K.pack_conv_weights(geom, w_khwc, dgrad=False)
    ref_b = K.pack_conv_weights(geom, w_chwk, dgrad=True)
    torch.cuda.synchronize()
    assert torch.equal(w_f.view(-1).view(torch.int16), ref_f.view(-1).view(torch.int16)[: w_f.numel()])
    assert torch.equal(w_b.view(-1).view(torch.int16), ref_b.view(-1).view(torch.int16)[: w_b.numel()])
    assert float((scale - scale_ref).abs().max()) < 1e-6
    assert torch.equal(u_f.view(torch.int16), w_khwc.view(torch.int16)) and torch.equal(u_b.view(torch.int16), w_chwk.view(torch.int16))


def _sparse_pm1(shape, per_row, g):
    """[rows, ...] tensor with `per_row` entries of +-1 / +-2 per leading index, zeros elsewhere."""
    rows = shape[0]
    flat = int(torch.tensor(shape[1:]).prod())
    w = torch.zeros((rows, flat))
    for r in range(rows):
        idx = torch.randperm(flat, generator=g)[:per_row]
        w[r, idx] = torch.randint(1, 3, (per_row,), generator=g).float() * (torch.randint(0, 2, (per_row,), generator=g).float() * 2 - 1)
    return w.view(shape)


@pytest.mark.parametrize("shape", SHAPES)
def test_packed_kernels_are_exact_on_integer_data(shape, dev):
    """a loose 1e-2 band cannot see one wrong tap among 2048 channels.  With small-integer operands every partial
    sum is an integer below 2^24, so fp32 accumulation is EXACT in any order, and with sparse +-1 / +-2 weights every result is an
    integer of magnitude < 256, which bf16 stores exactly: the packed forward / data gradient must equal torch's fp32 convolution
    BIT FOR BIT (residual, shift, ReLU, add and mask included), and a single mis-indexed tap, channel, chunk or pixel changes an
    output integer.  The dense weight gradients (fp32 slabs) are exact the same way."""
    N, H, W, Cin, Cout, R, stride, pad = shape
    g = torch.Generator().manual_seed(1234 + H * 5 + Cin + Cout + R)
    x = torch.randint(-4, 5, (N, Cin, H, W), generator=g).float()
    w = _sparse_pm1((Cout, Cin, R, R), 12, g)                                # forward sums: <= 12 * 2 * 4 = 96
    geom = K.make_geom(N, H, W, Cin, Cout, R, R, stride, pad)
    P, Q = geom.P, geom.Q
    w_khwc, w_chwk = K.weight_prep(w.to(dev), None, BF, Cin, Cout, want_fwd=True, want_bwd=True)
    xd = _nhwc(x, dev)
    if K.packed_supported(geom, BF, dgrad=False):
        shift = torch.randint(-8, 9, (Cout,), generator=g).float()
        res = torch.randint(-16, 17, (N, Cout, P, Q), generator=g).float()
        ref = F.conv2d(x, w, stride=stride, padding=pad)
        ref_full = torch.relu(ref + shift.view(1, -1, 1, 1) + res)
        assert float(ref_full.abs().max()) <= 256 and float(ref.abs().max()) <= 256
        wp = K.pack_conv_weights(geom, w_khwc, dgrad=False)
        y_plain = K.conv_fwd_packed(geom, xd, wp)
        y_full, bits = K.conv_fwd_packed(geom, xd, wp, shift.to(dev), _nhwc(res, dev), K.CS_ACT_RELU, want_bits=True)
        torch.cuda.synchronize()
        assert torch.equal(_from_nhwc(y_plain), ref), f"max diff {float((_from_nhwc(y_plain) - ref).abs().max())}"
        assert torch.equal(_from_nhwc(y_full), ref_full)
        assert torch.equal(_unpack_bits(bits, Cout), ref_full > 0)
    if stride == 1 and K.packed_supported(geom, BF, dgrad=True):
        # the data gradient contracts over (k, taps): sparse per INPUT channel
        wt = _sparse_pm1((Cin, Cout, R, R), 12, g).permute(1, 0, 2, 3).contiguous()
        _, wt_chwk = K.weight_prep(wt.to(dev), None, BF, Cin, Cout, want_fwd=False, want_bwd=True)
        dy = torch.randint(-4, 5, (N, Cout, P, Q), generator=g).float()
        add = torch.randint(-16, 17, (N, Cin, H, W), generator=g).float()
        mask = torch.rand((N, Cin, H, W), generator=g) > 0.4
        ref_dx = torch.nn.grad.conv2d_input((N, Cin, H, W), wt, dy, stride=stride, padding=pad)
        ref_dx2 = (ref_dx + add) * mask
        assert float(ref_dx2.abs().max()) <= 256 and float(ref_dx.abs().max()) <= 256
        dyd = _nhwc(dy, dev)
        wpd = K.pack_conv_weights(geom, wt_chwk, dgrad=True)
        dx = K.conv_dgrad_packed(geom, dyd, wpd)
        dx2, pc = K.conv_dgrad_packed(geom, dyd, wpd, add=_nhwc(add, dev), mask_bits=_pack_bits(mask, dev), want_colsum=True)
        cs = pc.vector()
        torch.cuda.synchronize()
        assert torch.equal(_from_nhwc(dx), ref_dx), f"max diff {float((_from_nhwc(dx) - ref_dx).abs().max())}"
        assert torch.equal(_from_nhwc(dx2), ref_dx2)
        assert torch.equal(cs.cpu()[:Cin], ref_dx2.sum(dim=(0, 2, 3)))       # integer column sums below 2^24: exact too
    # weight gradients (first- and second-generation kernels; fp32 output, dense integer operands)
    dyw = torch.randint(-3, 4, (N, Cout, P, Q), generator=g).float()
    ref_dw = torch.nn.grad.conv2d_weight(x, (Cout, Cin, R, R), dyw, stride=stride, padding=pad)
    assert float(ref_dw.abs().max()) < 2 ** 24
    slabs = K.wgrad_batched(geom, [xd, xd], [_nhwc(dyw, dev)] * 2)
    torch.cuda.synchronize()
    for i in range(2):
        got = slabs[i].double().sum(0).float().cpu().permute(0, 3, 1, 2)     # [K][R][S][C] -> [K][C][R][S]
        assert torch.equal(got, ref_dw), f"wgrad item {i}: max diff {float((got - ref_dw).abs().max())}"


def test_wide_kernel_forced_on_every_shape(dev):
    """conv2_wide_kernel<TM, NBW> is picked by rule (csrc/conv_v2.hip: pick_wide_tm) for a few geometries only; CELLSEG_WIDE = 4 / 6 / 8
    (A/B flavour of the library) forces that pixel-tile height wherever the window fits, = 1 switches the kernel off (the halo kernels
    serve everything, as before round 4).  Every 3x3 shape of this file must pass the parity and the exact-integer tests on every form
    (the knob is read once per process: child interpreters)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("1", "4", "6", "8"):
        # (CELLSEG_WIDE1: the same switch for the wide kernel's two-plane 1x1 form, served where the contraction has >= 8 chunks)
        env = dict(os.environ, CELLSEG_WIDE=mode, CELLSEG_WIDE1=mode, CELLSEG_LIB_FLAVOUR="ab", CELLSEG_TEST_ONLY_3X3="1")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_conv_packed_gpu.py"), "-m", "gpu", "-x", "-q",
                            "-k", "packed_fwd_and_dgrad or exact_on_integer"], capture_output=True, text=True, timeout=1200, env=env, cwd=root)
        assert r.returncode == 0, (mode, (r.stdout + r.stderr)[-3000:])


def _resnet_geometries():
    """(H, Cin, Cout, R) of every stride-1 packed convolution of the ResNet-50 / ResNet-18 trunks and the decoder."""
    out = []
    for H, w in ((75, 64), (38, 128), (19, 256), (10, 512)):
        out += [(H, w, w, 3), (H, w, 4 * w, 1), (H, 4 * w, w, 1)]
        if w > 64:
            out += [(2 * H - (1 if H in (38, 10) else 0), 2 * w, w, 1)]      # conv1 of a down-sampling block (input of the previous stage)
    out += [(19, 2048, 1024, 3), (38, 1024, 512, 3)]
    return out


@pytest.mark.parametrize("N", [1, 3, 16, 40])
def test_packed_kernels_agree_with_the_first_generation_over_the_resnet_family(N, dev):
    """Routing depends on the number of tiles (halo / wide 3x3 / ring / wide 1x1, pixel-tile heights), i.e. on the bag size: every
    stride-1 packed geometry of the ResNet family at four bag sizes, forward and data gradient (with the add operand -- dense, and compact
    where the layer is conv1 of a down-sampling block --, mask bits and column sums), against the first-generation kernels on the same
    device tensors.  The folded column sums must equal the sums of the stored gradient: a launch that writes other partial rows than
    cs_conv2d_packed_partial_rows reported (the round-4 fault of the wide 1x1 form with a compact operand) cannot pass."""
    g = torch.Generator().manual_seed(5 + N)
    for H, Cin, Cout, R in _resnet_geometries():
        if N * H * H * max(Cin, Cout) > 64 * 75 * 75 * 256:
            continue
        geom = K.make_geom(N, H, H, Cin, Cout, R, R, 1, R // 2)
        if not K.packed_supported(geom, BF, dgrad=False):
            continue
        x = torch.randn((N, H, H, Cin), generator=g).to(BF).to(dev)
        w = torch.randn((Cout, Cin, R, R), generator=g) / (Cin * R * R) ** 0.5
        w_khwc, w_chwk = K.weight_prep(w.to(dev), None, BF, Cin, Cout, want_fwd=True, want_bwd=True)
        y = K.conv_fwd_packed(geom, x, K.pack_conv_weights(geom, w_khwc, dgrad=False))
        y0 = K.conv_fwd(geom, x, w_khwc)
        torch.cuda.synchronize()
        assert _relerr(y.float().cpu(), y0.float().cpu()) < TOL, (N, H, Cin, Cout, R, "fwd")
        if not K.packed_supported(geom, BF, dgrad=True):
            continue
        dy = torch.randn((N, H, H, Cout), generator=g).to(BF).to(dev)
        mask = (torch.rand((N, H, H, Cin), generator=g) > 0.4)
        bits = _pack_bits(mask.permute(0, 3, 1, 2), dev)
        wpd = K.pack_conv_weights(geom, w_chwk, dgrad=True)
        adds = [torch.randn((N, H, H, Cin), generator=g).to(BF).to(dev)]
        if R == 1 and H >= 2:
            adds.append(K.CompactGrad(torch.randn((N, (H + 1) // 2, (H + 1) // 2, Cin), generator=g).to(BF).to(dev), 2))
        for add in adds:
            dense = add
            if isinstance(add, K.CompactGrad):
                dense = torch.zeros((N, H, H, Cin), dtype=BF, device=dev)
                dense[:, ::2, ::2, :] = add.t
            dx, pc = K.conv_dgrad_packed(geom, dy, wpd, add=add, mask_bits=bits, want_colsum=True)
            dx0 = K.conv_dgrad(geom, dy, w_chwk, add=dense, mask_bits=bits)
            torch.cuda.synchronize()
            assert _relerr(dx.float().cpu(), dx0.float().cpu()) < TOL, (N, H, Cin, Cout, R, "dgrad", type(add).__name__)
            assert _relerr(pc.vector().cpu(), dx.float().sum(dim=(0, 1, 2)).cpu()) < 2e-3, (N, H, Cin, Cout, R, "colsum", pc.rows)
