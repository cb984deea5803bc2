"""GPU: the train-mode BatchNorm reductions (cs_bn_stats, cs_bn_bwd_reduce + cs_bn_bwd_apply) on every decomposition the launch rule
produces -- one chunk / balanced 16-, 32-, 64-group chunks with a ragged last one, the atomics path and the partial-rows path, row
counts that are no multiple of a row block -- against fp64 torch, and twice for bit-identical results.  The model-level parity tests
(C1, C5, EfficientNet vs the oracle) only reach the shapes of those networks.  Reference: nn.BatchNorm2d / BatchNorm1d in train mode,
model/resnet.py:21,52-56,134-148, model/efficientnet.py:97-103."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import kernels as K  # noqa: E402

SHAPES = [(6400, 1392), (6400, 2304), (9001, 288), (5000, 816), (23104, 576), (37, 8), (1000, 40), (300, 520), (8, 512), (70001, 24),
          (6400, 232), (2000, 136), (123457, 64)]


def _inputs(M, C, dtype, dev):
    g = torch.Generator().manual_seed(7 * M + C)
    z = (torch.randn((M, C), generator=g) * 1.3 + 0.2).to(dtype).to(dev)
    dy = torch.randn((M, C), generator=g).to(dtype).to(dev)
    gamma = (torch.rand((C,), generator=g) + 0.5).to(dev)
    beta = (torch.randn((C,), generator=g) * 0.1).to(dev)
    return z, dy, gamma, beta


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bn_stats_sums(shape, dtype):
    dev = torch.device("cuda:0")
    M, C = shape
    z, _, _, _ = _inputs(M, C, dtype, dev)
    got = K.stats_values(K.bn_stats(z)).cpu()
    again = K.stats_values(K.bn_stats(z)).cpu()
    zd = z.double().cpu()
    want = torch.stack([zd.sum(0), (zd * zd).sum(0)])
    # per-thread fp32 partial sums over <= a row block's share of rows, everything above that exact
    tol = 2e-6 * torch.stack([zd.abs().sum(0), (zd * zd).sum(0)]) + 1e-9
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
    assert torch.equal(got, again)                      # order-independent accumulation: bit for bit


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("act", [K.CS_ACT_NONE, K.CS_ACT_SILU])
def test_bn_backward_against_fp64_autograd(shape, act):
    dev = torch.device("cuda:0")
    M, C = shape
    if M < 2:
        pytest.skip("batch statistics need two rows")
    z, dy, gamma, beta = _inputs(M, C, torch.bfloat16, dev)
    eps = 1e-3
    stats = K.bn_stats(z)
    mean, rstd = K.bn_finalize(stats, M, eps, 0.1)
    dz, dgamma, dbeta = K.bn_bwd(dy, z, mean, rstd, gamma, True, beta=beta, act=act)
    dz2, dgamma2, dbeta2 = K.bn_bwd(dy, z, mean, rstd, gamma, True, beta=beta, act=act)
    torch.cuda.synchronize()
    assert torch.equal(dz.view(torch.int16), dz2.view(torch.int16)) and torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    zr = z.double().cpu().requires_grad_(True)
    gr = gamma.double().cpu().requires_grad_(True)
    br = beta.double().cpu().requires_grad_(True)
    mu = zr.mean(0)
    var = zr.var(0, unbiased=False)
    u = (zr - mu) / torch.sqrt(var + eps) * gr + br
    y = u * torch.sigmoid(u) if act == K.CS_ACT_SILU else u
    y.backward(dy.double().cpu())
    scale = float(zr.grad.abs().max()) + 1e-12
    assert float((dz.double().cpu() - zr.grad).abs().max()) < 1.2e-2 * scale          # bf16 output, fp32 arithmetic
    for got, want in ((dgamma, gr.grad), (dbeta, br.grad)):
        err = (got.double().cpu() - want).abs()
        assert bool((err <= 2e-3 * want.abs() + 2e-4 * float(want.abs().max()) + 1e-6).all()), float(err.max())
