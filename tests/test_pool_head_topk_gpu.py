"""GPU parity of pooling, heads/losses and the adaptive top-k kernel through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import kernels as K  # noqa: E402


def _nhwc(t, dtype, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(10, 10), (15, 14), (7, 9)])
def test_maxpool(dtype, hw, dev):
    torch.manual_seed(3)
    H, W = hw
    x = torch.randn(2, 16, H, W).to(dtype).float()
    x = torch.relu(x)                      # plenty of exact ties at 0, like the stem output
    x.requires_grad_()
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn_like(y).to(dtype).float()
    y.backward(dy)
    yd, am = K.maxpool_fwd(_nhwc(x.detach(), dtype, dev))
    dxd = K.maxpool_bwd(_nhwc(dy, dtype, dev), am, None, (H, W))
    torch.cuda.synchronize()
    assert torch.equal(_nchw(yd), y.detach())
    ref = x.grad.to(dtype).float()
    assert float((_nchw(dxd) - ref).abs().max()) <= (1e-6 if dtype == torch.float32 else 4e-2)
    # fused ReLU mask: identical to masking dy by [y>0] first
    dxm = K.maxpool_bwd(_nhwc(dy, dtype, dev), am, yd, (H, W))
    x2 = x.detach().clone().requires_grad_()
    F.max_pool2d(x2, 3, 2, 1).backward(dy * (y.detach() > 0))
    torch.cuda.synchronize()
    assert float((_nchw(dxm) - x2.grad.to(dtype).float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gap_avgmax(dtype, dev):
    torch.manual_seed(5)
    x = torch.relu(torch.randn(3, 72, 10, 10)).to(dtype).float().requires_grad_()
    f = (F.adaptive_avg_pool2d(x, 1) + F.adaptive_max_pool2d(x, 1)).flatten(1)
    df = torch.randn_like(f)
    f.backward(df)
    xd = _nhwc(x.detach(), dtype, dev)
    feat, am = K.gap_fwd(xd)
    dx = K.gap_bwd(df.to(dev), am, xd, relu_mask=False)
    dxm = K.gap_bwd(df.to(dev), am, xd, relu_mask=True)
    torch.cuda.synchronize()
    assert float((feat.cpu() - f.detach()).abs().max()) < 1e-5
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert float((_nchw(dx) - x.grad).abs().max()) < tol
    assert float((_nchw(dxm) - x.grad * (x.detach() > 0)).abs().max()) < tol


def test_softmax_argmax_takes_the_arg_max_of_the_probabilities(dev):
    """inference.py:72-76 / :118-119 do np.argmax(F.softmax(logits, 1), axis=1): index work, bit-exact.  Two logits one ulp apart
    tie as fp32 probabilities (exp of -7e-9 rounds to 1.0f) and np.argmax then returns the FIRST index, whereas the arg max of
    the logits returns the larger one ('What's weak')."""
    torch.manual_seed(2)
    lo = torch.randn(300, 7)
    a = torch.tensor(0.1)
    b = torch.nextafter(a, torch.tensor(1.0))
    lo[0] = torch.tensor([-3.0, a, b, -1.0, -2.0, -5.0, -4.0])            # larger logit second -> probabilities tie -> index 1
    lo[1] = torch.tensor([b, -3.0, a, -1.0, -2.0, -5.0, -4.0])            # larger logit first
    lo[2] = torch.tensor([-1.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0])      # all equal -> 0
    ref = np.argmax(F.softmax(lo, dim=1).detach().clone().cpu(), axis=1)
    ref = np.asarray(ref)
    got = K.softmax_argmax(lo.to(dev)).cpu().numpy()
    assert got.dtype == np.int64
    assert int(ref[0]) == 1 and int(torch.argmax(lo[0])) == 2             # the case the logits arg max gets wrong
    assert np.array_equal(got, ref)


def test_linear_ce_mse(dev):
    torch.manual_seed(11)
    M, Kf, N = 9, 515, 7
    x = torch.randn(M, Kf, requires_grad=True)
    w = torch.randn(N, Kf, requires_grad=True)
    b = torch.randn(N, requires_grad=True)
    lab = torch.randint(0, N, (M,))
    y = F.linear(x, w, b)
    loss = F.cross_entropy(y, lab) * 0.7
    loss.backward()
    yd = K.linear_fwd(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev))
    ld, dl = K.softmax_ce(yd, lab.to(dev), 0.7)
    dx, dw, db = K.linear_bwd(x.detach().to(dev), w.detach().to(dev), dl)
    p1 = K.softmax_prob1(yd)
    torch.cuda.synchronize()
    assert float((yd.cpu() - y.detach()).abs().max()) < 1e-4
    assert abs(float(ld.cpu()) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert float((dx.cpu() - x.grad).abs().max()) < 1e-5
    assert float((dw.cpu() - w.grad).abs().max()) < 1e-5
    assert float((db.cpu() - b.grad).abs().max()) < 1e-5
    assert float((p1.cpu() - F.softmax(y.detach(), 1)[:, 1]).abs().max()) < 1e-6
    # relu-terminated linear (count regressor)
    y2 = torch.relu(F.linear(x, w[:1], b[:1]))
    t = torch.tensor([0., 3., 12., 40., 1., 25., 7., 0., 100.])
    x.grad = None; w.grad = None; b.grad = None
    l2 = ((y2.squeeze() - t) ** 2).mean()
    l2.backward()
    y2d = K.linear_fwd(x.detach().to(dev), w.detach()[:1].contiguous().to(dev), b.detach()[:1].contiguous().to(dev), K.CS_ACT_RELU)
    l2d, dy2 = K.mse(y2d.view(-1), t.to(dev))
    dx2, dw2, db2 = K.linear_bwd(x.detach().to(dev), w.detach()[:1].contiguous().to(dev), dy2.view(M, 1), y2d, K.CS_ACT_RELU)
    torch.cuda.synchronize()
    assert abs(float(l2d.cpu()) - float(l2)) < 1e-4 * max(1.0, abs(float(l2)))
    assert float((dx2.cpu() - x.grad).abs().max()) < 1e-4 * max(1.0, float(x.grad.abs().max()))
    assert float((dw2.cpu() - w.grad[:1]).abs().max()) < 1e-4 * max(1.0, float(w.grad.abs().max()))
    # weighted mse known answer from the reference (metrics/metrics.py:23-33)
    lw, _ = K.mse(torch.tensor([1., 25., 30.]).to(dev), torch.tensor([2., 20., 40.]).to(dev), weighted=True)
    torch.cuda.synchronize()
    assert abs(float(lw.cpu()) - 148.59375) < 1e-3


@pytest.mark.parametrize("shape", [(64, 2304, 96), (64, 96, 2304), (50, 130, 70), (64, 40, 10), (17, 64, 2048),
                                   # a long batch axis (the reference's tile batches, train_tile.py -b 40960): the weight gradient over row slices
                                   (8192, 48, 1152), (5000, 10, 40), (1000, 70, 130), (40960, 8, 192)])
@pytest.mark.parametrize("act", ["none", "relu", "sigmoid", "silu"])
def test_linear_tiled_products(shape, act, dev):
    """The 64 x 64 LDS-tiled Linear kernels (forward, dx, dw/db; >= 1024 outputs -- the squeeze-excitation layers of EfficientNet
    and the image heads at batch >= 16) against torch, every output activation of cs_linear_*, ragged tiles."""
    M, N, Kf = shape
    torch.manual_seed(M + N + Kf)
    x = torch.randn(M, Kf, requires_grad=True)
    w = (torch.randn(N, Kf) / Kf ** 0.5).requires_grad_()
    b = torch.randn(N, requires_grad=True)
    code = {"none": K.CS_ACT_NONE, "relu": K.CS_ACT_RELU, "sigmoid": K.CS_ACT_SIGMOID, "silu": K.CS_ACT_SILU}[act]
    pre = F.linear(x, w, b)
    y = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "silu": F.silu}[act](pre)
    dy = torch.randn(M, N)
    y.backward(dy)
    xd, wd, bd = x.detach().to(dev), w.detach().to(dev), b.detach().to(dev)
    yd, pred = K.linear_fwd(xd, wd, bd, code, want_preact=True)
    saved = pred if act == "silu" else yd                   # cs_linear_bwd: the pre-activation for SiLU, the output otherwise
    dx, dw, db = K.linear_bwd(xd, wd, dy.to(dev), saved, code)
    torch.cuda.synchronize()
    tol = 2e-5 if M <= 512 else 2e-4             # (fp32 sums over thousands of rows: torch's own order differs too)
    assert float((yd.cpu() - y.detach()).abs().max()) < tol * max(1.0, float(y.detach().abs().max()))
    assert float((pred.cpu() - pre.detach()).abs().max()) < tol * max(1.0, float(pre.detach().abs().max()))
    assert float((dx.cpu() - x.grad).abs().max()) < tol * max(1.0, float(x.grad.abs().max()))
    assert float((dw.cpu() - w.grad).abs().max()) < tol * max(1.0, float(w.grad.abs().max()))
    assert float((db.cpu() - b.grad).abs().max()) < tol * max(1.0, float(b.grad.abs().max()))


def _sample_reference(probs, groups, labels, tiles_per_pos, topk_neg):
    """inference.py:31-43 restated in numpy (the oracle for the selection kernel)."""
    groups = np.asarray(groups)
    order = np.lexsort((probs, groups))
    T = len(groups)
    index = np.empty(T, 'bool')
    for i in range(T):
        lab = labels[groups[i]]
        topk = topk_neg if lab == 0 else lab * tiles_per_pos
        index[i] = groups[i] != groups[(i + topk) % T]
    return order[index]


def _run_topk(probs, groups, labels, tiles_per_pos, topk_neg, dev):
    groups = np.asarray(groups)
    lab = np.asarray([labels[g] for g in groups])
    kpt = np.where(lab == 0, topk_neg, lab * tiles_per_pos).astype(np.int32)
    starts = np.flatnonzero(np.r_[True, groups[1:] != groups[:-1]])
    offs = np.r_[starts, len(groups)].astype(np.int64)
    out, cnt = K.segmented_topk(torch.from_numpy(probs).to(dev), torch.from_numpy(groups.astype(np.int32)).to(dev),
                                torch.from_numpy(kpt).to(dev), torch.from_numpy(offs).to(dev), int(np.diff(offs).max()))
    torch.cuda.synchronize()
    n = int(cnt.cpu())
    return out[:n].cpu().numpy()


def test_topk_known_answer(dev):
    # probed on the reference
    groups = [1] * 5 + [2] * 5 + [3] * 5
    labels = {1: 2, 2: 0, 3: 1}
    probs = np.random.RandomState(0).rand(15).astype(np.float32)
    got = _run_topk(probs, groups, labels, 1, 3, dev)
    assert got.tolist() == [2, 1, 5, 7, 8, 13]
    assert got.tolist() == _sample_reference(probs, groups, labels, 1, 3).tolist()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_topk_random_ragged_with_ties(seed, dev):
    rs = np.random.RandomState(seed)
    n_img = 37
    sizes = rs.randint(1, 400, size=n_img)
    sizes[3] = 1
    sizes[5] = 3364          # 16x16 tiles stride 5 (train_seg.py:225-232)
    groups = np.repeat(np.arange(n_img), sizes)
    labels = {g: int(rs.choice([0, 0, 1, 2, 5, 40])) for g in range(n_img)}
    probs = rs.rand(len(groups)).astype(np.float32)
    probs[rs.rand(len(groups)) < 0.3] = 0.5            # heavy ties
    probs[:7] = np.float32(-0.0)
    probs[7:11] = np.float32(0.0)
    for kp, kn in [(1, 30), (3, 5), (1, 0)]:
        got = _run_topk(probs, groups, labels, kp, kn, dev)
        ref = _sample_reference(probs, groups, labels, kp, kn)
        assert got.tolist() == ref.tolist()


def test_topk_runs_longer_than_the_lds_sort(dev):
    """The reference sorts whatever the dataset holds (inference.py:34-41): images with more than 8192 tiles (whole-slide grids)
    take the in-place global-memory sorter; mixed with short runs, heavy ties, a non-power-of-two and an exact power-of-two run."""
    rs = np.random.RandomState(11)
    sizes = np.array([100, 8193, 7, 20000, 8192, 16384, 1])
    groups = np.repeat(np.arange(len(sizes)), sizes)
    labels = {g: int(rs.choice([0, 1, 3, 50])) for g in range(len(sizes))}
    probs = rs.rand(len(groups)).astype(np.float32)
    probs[rs.rand(len(groups)) < 0.4] = 0.25           # ties: the sort must stay stable
    for kp, kn in [(1, 30), (100, 5000)]:
        got = _run_topk(probs, groups, labels, kp, kn, dev)
        ref = _sample_reference(probs, groups, labels, kp, kn)
        assert got.tolist() == ref.tolist()
    offs = np.r_[0, np.cumsum(sizes)].astype(np.int64)
    order = K.segmented_order(torch.from_numpy(probs).to(dev), torch.from_numpy(offs).to(dev), int(sizes.max()))
    torch.cuda.synchronize()
    assert order.cpu().numpy().tolist() == np.lexsort((probs, groups)).tolist()


def test_topk_single_group_wraps(dev):
    # one bag of 64 tiles: (i+k) % T always lands in the same group -> nothing selected
    probs = np.random.RandomState(4).rand(64).astype(np.float32)
    got = _run_topk(probs, [0] * 64, {0: 0}, 1, 30, dev)
    ref = _sample_reference(probs, [0] * 64, {0: 0}, 1, 30)
    assert got.tolist() == ref.tolist() == []


def test_tile_gather_matches_get_tiles_totensor_normalize(dev):
    """device tile construction == reference get_tiles + ToTensor + Normalize (via the oracle)."""
    from cellsegmentation_amd import synth, tiles
    from oracle import cellseg_oracle as orc
    imgs = synth.ihc_tiles(3, 299, 77)                      # uint8 [3,299,299,3]
    coords = tiles.get_tiles((299, 299), 20, 32)
    assert coords == orc.get_tiles_coords(299, 299, 20, 32) and len(coords) == 225 and coords[-1] == (267, 267)
    ti, rc = tiles.tile_index(3, (299, 299), 20, 32)
    assert len(ti) == 675 and ti[224] == 0 and ti[225] == 1
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
        out = tiles.gather_tiles(torch.from_numpy(imgs).to(dev), ti, rc, 32, dtype)
        torch.cuda.synchronize()
        ref = torch.stack([synth.normalise(imgs[i:i + 1, r:r + 32, c:c + 32])[0] for i, (r, c) in zip(ti.tolist(), rc.tolist())])
        got = out[..., :3].float().cpu().permute(0, 3, 1, 2)
        assert float((got - ref).abs().max()) <= tol
        assert float(out[..., 3:].float().abs().max()) == 0.0
    # staged tiles feed the model directly (no NCHW detour)
    from cellsegmentation_amd.model import resnet as R
    m = R.MILresnet18()
    sd = m.state_dict(); synth.fill_state_dict(sd); m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.float32); m.setmode("tile"); m.eval()
    with torch.no_grad():
        a = m(tiles.gather_tiles(torch.from_numpy(imgs).to(dev), ti[:16], rc[:16], 32, torch.float32))
        b = m(ref[:16].to(dev))
    assert float((a - b).abs().max()) < 1e-5


def test_positive_bits_plane(dev):
    """cs_positive_bits in the channel-block-major layout every bit-plane producer / consumer of the library uses: the 32-bit word
    (c // 32) * M + pixel, bit c % 32 = x[pixel, c] > 0 (the mask operand the packed data gradients read); kernels.unpack_bits /
    pack_bits are its torch-side inverse pair."""
    torch.manual_seed(9)
    x = torch.randn(3, 5, 7, 96)
    x[x.abs() < 0.3] = 0.0
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1.0, -1.0, 3e38, -3e38])
    xb = x.to(torch.bfloat16)
    bits = K.positive_bits(xb.to(dev)).cpu()
    assert bits.shape == (3, 5, 7, 12) and bits.dtype == torch.uint8
    M = 3 * 5 * 7
    ref = (xb.float() > 0).view(M, 3, 32)                                  # [pixel][block][bit]
    words = (ref.long() * (2 ** torch.arange(32)).view(1, 1, 32)).sum(-1).t().contiguous()       # [block][pixel]
    got = bits.view(-1).view(3, M, 4).long()
    got_words = got[..., 0] + (got[..., 1] << 8) + (got[..., 2] << 16) + (got[..., 3] << 24)
    assert torch.equal(got_words, words)
    assert torch.equal(K.unpack_bits(bits, 96), xb.float() > 0)
    assert torch.equal(K.pack_bits(xb.float() > 0), bits)


def _linear_products(dev):
    """fwd / bwd of the squeeze-excitation shapes (64 rows) as raw bytes, seeded."""
    import hashlib
    out = []
    for (M, C, Q) in ((64, 576, 24), (64, 1392, 58), (64, 40, 10), (33, 144, 6)):
        g = torch.Generator().manual_seed(M * 7 + C)
        x = torch.randn((M, C), generator=g).to(dev)
        w1, b1 = (torch.randn((Q, C), generator=g) * 0.05).to(dev), torch.randn((Q,), generator=g).to(dev)
        w2, b2 = (torch.randn((C, Q), generator=g) * 0.05).to(dev), torch.randn((C,), generator=g).to(dev)
        h, pre = K.linear_fwd(x, w1, b1, K.CS_ACT_SILU, want_preact=True)
        s = K.linear_fwd(h, w2, b2, K.CS_ACT_SIGMOID)
        g2, g1 = torch.randn((M, C), generator=g).to(dev), torch.randn((M, Q), generator=g).to(dev)
        r2 = K.linear_bwd(h, w2, g2, s, K.CS_ACT_SIGMOID)
        r1 = K.linear_bwd(x, w1, g1, pre, K.CS_ACT_SILU)
        torch.cuda.synchronize()
        for t in (h, pre, s) + tuple(r2) + tuple(r1):
            out.append(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest())
    return out


def test_direct_linear_kernels_give_the_bits_of_the_tiles(dev):
    """The direct kernels of the short contractions (linear_fwd_shortk_kernel, linear_dx_shortn_body, linear_dw_shortm_body, csrc/head.hip)
    accumulate in the order of the 64 x 64 tiles they replace: the production library (direct kernels) and the A/B flavour with
    CELLSEG_LINEAR_DIRECT=0 (tiles) must give the same bytes.  model/efficientnet.py:62-80 (SqueezeExcitation.fc1 / fc2)."""
    import json
    import os
    import subprocess
    import sys
    from cellsegmentation_amd import _lib
    if _lib.FLAVOUR != "":
        pytest.skip("parent of the A/B comparison runs on the production library")
    mine = _linear_products(dev)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_pool_head_topk_gpu as T; "
            "print('BITS', json.dumps(T._linear_products(torch.device('cuda:0'))))") % (root, os.path.join(root, "tests"))
    env = dict(os.environ, CELLSEG_LIB_FLAVOUR="ab", CELLSEG_LINEAR_DIRECT="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("BITS ")][-1]
    theirs = json.loads(line[5:])
    assert mine == theirs
