"""Tensor-level wrappers over the C ABI (no autograd here).

Every function checks that its operands live on the GPU and are contiguous, then enqueues the HIP
kernel on torch's current stream.  Activations are NHWC tensors of dtype float32 (parity mode) or
bfloat16 (throughput mode) whose channel count is already padded (see ``pad_channels``).
"""
import ctypes

import torch

from . import _lib
from ._lib import (CS_ACT_NONE, CS_ACT_RELU, CS_ACT_SIGMOID, CS_ACT_SILU, CS_BF16, CS_BN_BWD_FROZEN, CS_BN_BWD_OWN_RELU, CS_F32,  # noqa: F401
                   CsConvGeom)


def _code(dtype):
    if dtype == torch.float32:
        return CS_F32
    if dtype == torch.bfloat16:
        return CS_BF16
    raise TypeError(f"unsupported activation dtype {dtype}; use torch.float32 or torch.bfloat16")


def chunk(dtype):
    """Elements per 16-byte chunk."""
    return 4 if dtype == torch.float32 else 8


def pad_channels(c, dtype=None):
    """Stored channel count: multiple of 8 (a multiple of both chunk sizes)."""
    return (c + 7) // 8 * 8


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("cellsegmentation_amd kernels need GPU tensors: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("cellsegmentation_amd kernels need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    # the raw handle of torch's current stream; ~10x cheaper than building a torch.cuda.Stream object per launch (the
    # BN-train image path issues ~1000 launches per step and is host-bound)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def make_geom(N, H, W, C, K, R, S, stride, pad, groups=1):
    P = (H + 2 * pad - R) // stride + 1
    Q = (W + 2 * pad - S) // stride + 1
    return CsConvGeom(N, H, W, C, K, R, S, stride, pad, P, Q, groups)


# ---------------------------------------------------------------- layout
def to_nhwc(x, dtype, Cp):
    """x[N,C,H,W] fp32 -> [N,H,W,Cp] dtype (zero padded channels)."""
    N, C, H, W = x.shape
    if x.dtype != torch.float32:
        raise TypeError("to_nhwc expects an fp32 NCHW tensor")
    y = torch.empty((N, H, W, Cp), dtype=dtype, device=x.device)
    _lib.check(_lib.load().cs_nchw_to_nhwc(_p(x), _p(y), _code(dtype), N, C, H, W, Cp, _stream()), "nchw_to_nhwc")
    return y


def to_nchw(y, C):
    """y[N,H,W,Cp] -> [N,C,H,W] fp32 (first C channels)."""
    N, H, W, Cp = y.shape
    x = torch.empty((N, C, H, W), dtype=torch.float32, device=y.device)
    _lib.check(_lib.load().cs_nhwc_to_nchw(_p(y), _code(y.dtype), _p(x), N, C, H, W, Cp, _stream()), "nhwc_to_nchw")
    return x


# ---------------------------------------------------------------- parameters
def bn_fold(gamma, beta, mean, var, eps, conv_bias=None):
    C = mean.numel()
    out = torch.empty((3, C), dtype=torch.float32, device=mean.device)
    _lib.check(_lib.load().cs_bn_fold(_p(gamma), _p(beta), _p(mean), _p(var), eps, _p(conv_bias), _p(out[0]), _p(out[1]), _p(out[2]),
                                      C, _stream()), "bn_fold")
    return out[0], out[1], out[2]


def stage_conv_bn(w, gamma, beta, mean, var, eps, conv_bias, dtype, Cp, Kp, want_bwd=False):
    """Fused BN fold + weight staging. Returns w_khwc, w_chwk, scale, shift, rstd (the last three [Kp])."""
    K_, Cin, R, S = w.shape
    w_khwc = torch.empty((Kp, R, S, Cp), dtype=dtype, device=w.device)
    w_chwk = torch.empty((Cp, R, S, Kp), dtype=dtype, device=w.device) if want_bwd else None
    vec = torch.empty((3, Kp), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().cs_stage_conv_bn(_p(w), _p(gamma), _p(beta), _p(mean), _p(var), eps, _p(conv_bias), _code(dtype), K_, Cin, R, S,
                                            Cp, Kp, _p(w_khwc), _p(w_chwk), _p(vec[0]), _p(vec[1]), _p(vec[2]), _stream()), "stage_conv_bn")
    return w_khwc, w_chwk, vec[0], vec[1], vec[2]


class StagePack:
    """Every Conv2d(+eval-mode BatchNorm2d) of a network staged by ONE launch (cs_stage_conv_bn_multi).

    layers: [(conv, bn, Cp, Kp, want_bwd, fwd_packed, bwd_packed)]; bn = None for a layer whose BatchNorm runs on batch statistics
    (scale 1, shift = the convolution's bias or 0).  The staging buffers and the device descriptor table are
    allocated once and rewritten by every launch(); valid() tells whether the parameter tensors are still the ones the table
    points at.  *_packed: the operand is written in the MFMA-fragment order of the packed-operand kernels (conv_v2.hip)."""

    def __init__(self, layers, dtype):
        lib = _lib.load()
        dev = layers[0][0].weight.device
        self.dtype, self.layers, self.staged = dtype, layers, []
        arr = (_lib.CsStageDesc * len(layers))()
        block = 0
        for i, (conv, bn, Cp, Kp, want_bwd, pkf, pkb) in enumerate(layers):
            K_, Cin, R, S = conv.weight.shape
            w_khwc = torch.empty((Kp, R, S, Cp), dtype=dtype, device=dev)
            w_chwk = torch.empty((Cp, R, S, Kp), dtype=dtype, device=dev) if want_bwd else None
            vec = torch.empty((3, Kp), dtype=torch.float32, device=dev)
            d = arr[i]
            d.w = conv.weight.data_ptr()
            if bn is not None:
                d.gamma, d.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                d.mean, d.var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            else:                                   # a layer whose BatchNorm runs on batch statistics: nothing of it is folded
                d.gamma = d.beta = d.mean = d.var = None
            d.conv_bias = conv.bias.data_ptr() if conv.bias is not None else None
            d.w_khwc, d.w_chwk = w_khwc.data_ptr(), (w_chwk.data_ptr() if want_bwd else None)
            d.scale, d.shift, d.rstd = vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr()
            d.eps = float(bn.eps) if bn is not None else 0.0
            d.K, d.Cin, d.R, d.S, d.Cp, d.Kp, d.block0 = K_, Cin, R, S, Cp, Kp, block
            d.fwd_packed, d.bwd_packed = int(bool(pkf)), int(bool(pkb and want_bwd))
            if (pkf and (Kp % 32 or Cp % 64)) or (pkb and want_bwd and (Cp % 32 or Kp % 64)):
                raise ValueError("StagePack: packed layouts need 32-row tiles and 64-channel chunks")
            nb = lib.cs_stage_conv_bn_blocks(K_, Cin, R, S, Cp, Kp, 1, 1 if want_bwd else 0)
            if nb < 1:
                raise ValueError("StagePack: bad layer extents")
            block += nb
            self.staged.append((w_khwc, w_chwk, vec[0], vec[1], vec[2]))
        self.total_blocks = block
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.key = self._key()

    def _key(self):
        out = []
        for conv, bn, *_ in self.layers:
            for t in (conv.weight, conv.bias) + ((bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else ()):
                out.append(t.data_ptr() if t is not None else 0)
        return tuple(out)

    def valid(self):
        return self._key() == self.key

    def launch(self):
        _lib.check(_lib.load().cs_stage_conv_bn_multi(self.table.data_ptr(), len(self.layers), self.total_blocks, _code(self.dtype), _stream()),
                   "stage_conv_bn_multi")


def weight_prep(w, scale, dtype, Cp, Kp, want_fwd=True, want_bwd=False):
    K, Cin, R, S = w.shape
    w_khwc = torch.empty((Kp, R, S, Cp), dtype=dtype, device=w.device) if want_fwd else None
    w_chwk = torch.empty((Cp, R, S, Kp), dtype=dtype, device=w.device) if want_bwd else None
    _lib.check(_lib.load().cs_weight_prep(_p(w), _p(scale), _code(dtype), K, Cin, R, S, Cp, Kp, _p(w_khwc), _p(w_chwk),
                                          _stream()), "weight_prep")
    return w_khwc, w_chwk


# ---------------------------------------------------------------- convolution family
class LaunchTimer:
    """Optional per-launch HIP-event timing of the conv family (used by bench.py for the roofline
    object).  Events are recorded on torch's current stream = the stream the kernels run on."""

    def __init__(self):
        self.records = []      # (kind, geom tuple, dtype, start, end)
        self.enabled = True    # the caller may switch the bracketing off for some steps

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None

    def results(self):
        """[(kind, geom dict, dtype, milliseconds)] -- call after torch.cuda.synchronize()."""
        return [(k, g, d, s.elapsed_time(e)) for k, g, d, s, e in self.records]


_timer = None


def _timed(kind, geom, dtype, fn, extra_tensors=0, batch=1):
    """extra_tensors: how many destination-shaped tensors the epilogue also reads (residual / add / mask)."""
    if _timer is None or not _timer.enabled:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    g = {f: getattr(geom, f) for f, _ in geom._fields_}
    g["extra"] = extra_tensors
    g["batch"] = batch
    g["variant"] = (_lib.load().cs_last_conv_variant() or b"").decode()      # what the library actually launched
    _timer.records.append((kind, g, dtype, s, e))
    return rc


def _stats_ws(M, n_out, device):
    nbytes = _lib.load().cs_conv2d_stats_workspace(int(M), int(n_out))
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=device)


def set_igemm_path(path):
    """0 = LDS-DMA staging (default), 1 = register staging, 3 = LDS-DMA + experimental streaming kernel. Returns the previous value.
    A/B flavour of the library only (CELLSEG_LIB_FLAVOUR=ab, `make AB=1`): the production library has no such switch."""
    if _lib.FLAVOUR not in ("ab", "dbg"):
        raise RuntimeError("set_igemm_path: the production library has no A/B switches; run with CELLSEG_LIB_FLAVOUR=ab")
    return _lib.load().cs_set_igemm_path(int(path))


def igemm_tile(M, n_out):
    """(BM, BN) the fwd/dgrad dispatcher picks."""
    v = _lib.load().cs_igemm_tile(int(M), int(n_out))
    return v // 1000, v % 1000


def weight_prep_grouped(w, scale, dtype, want_fwd=True, want_bwd=False):
    """w[K][Cg][R][S] (groups = K // Cg... i.e. Conv2d(groups=G).weight) -> slab-dense operands [K][R][S][64]."""
    K_, Cg, R, S = w.shape
    w_khwc = torch.empty((K_, R, S, 64), dtype=dtype, device=w.device) if want_fwd else None
    w_chwk = torch.empty((K_, R, S, 64), dtype=dtype, device=w.device) if want_bwd else None
    _lib.check(_lib.load().cs_weight_prep_grouped(_p(w), _p(scale), _code(dtype), K_, Cg, R, S, _p(w_khwc), _p(w_chwk), _stream()),
               "weight_prep_grouped")
    return w_khwc, w_chwk


def wgrad_finalize_grouped(dw_slab, w, scale, rstd, mean, gsum, dw, dgamma=None, dbeta=None, dot=None):
    K_, Cg, R, S = dw.shape
    if dgamma is not None and dot is None:
        dot = torch.zeros((K_,), dtype=torch.float32, device=dw.device)
    _lib.check(_lib.load().cs_wgrad_finalize_grouped(_p(dw_slab), dw_slab.shape[0], _p(w), _p(scale), _p(rstd), _p(mean), _p(gsum), K_, Cg, R, S, _p(dw),
                                                     _p(dgamma), _p(dbeta), _p(dot), _stream()), "wgrad_finalize_grouped")


def _grouped_geom(geom, grouped):
    """The geometry the C side sees: `groups > 1` selects the slab-dense grouped kernels (an explicit field of CsConvGeom; the
    library keeps no per-thread state between calls)."""
    if not grouped or geom.groups > 1:
        return geom
    g = CsConvGeom.from_buffer_copy(geom)
    g.groups = max(2, geom.C // 64 if geom.C >= 128 else 2)      # any value > 1: the staged weights carry the block structure
    return g


def conv_fwd(geom, x, w_khwc, scale=None, shift=None, residual=None, act=CS_ACT_NONE, stats=None, out=None, grouped=False,
             want_bits=False):
    """want_bits: also return a uint8 [N,P,Q,K/8] tensor with one bit per output element (> 0), the 1-byte-per-16 form of the
    ReLU mask a later data gradient needs -> (y, bits)."""
    y = out if out is not None else torch.empty((geom.N, geom.P, geom.Q, geom.K), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    geom = _grouped_geom(geom, grouped)
    if want_bits:
        if stats is not None:
            raise ValueError("conv_fwd: want_bits and stats are separate entry points")
        bits = torch.empty((geom.N, geom.P, geom.Q, geom.K // 8), dtype=torch.uint8, device=x.device)
        _lib.check(_timed("fwd", geom, x.dtype, lambda: lib.cs_conv2d_fwd_bits(
            ctypes.byref(geom), _code(x.dtype), _p(x), _p(w_khwc), _p(scale), _p(shift), _p(residual), act, _p(y), _p(bits), _stream()),
            extra_tensors=int(residual is not None) + 1.0 / (8 * y.element_size())), "conv2d_fwd_bits")
        return y, bits
    ws = _stats_ws(geom.N * geom.P * geom.Q, geom.K, x.device) if stats is not None else None
    _lib.check(_timed("fwd", geom, x.dtype, lambda: lib.cs_conv2d_fwd(
        ctypes.byref(geom), _code(x.dtype), _p(x), _p(w_khwc), _p(scale), _p(shift), _p(residual), act, _p(y), _p(stats), _p(ws),
        _stream()), extra_tensors=int(residual is not None)), "conv2d_fwd")
    return y


# ---------------------------------------------------------------- packed-operand bf16 convolutions (csrc/conv_v2.hip)
def packed_supported(geom, dtype, dgrad=False):
    """Does the halo-tile / packed-weight kernel serve this geometry (bf16 only)?"""
    return dtype == torch.bfloat16 and bool(_lib.load().cs_conv2d_packed_supported(ctypes.byref(geom), 1 if dgrad else 0))


def pack_conv_weights(geom, w_staged, dgrad=False, out=None):
    """staged w_khwc (forward) / w_chwk (data gradient) -> MFMA-fragment order (a flat bf16 tensor)."""
    lib = _lib.load()
    n = lib.cs_conv2d_packed_weight_bytes(ctypes.byref(geom), 1 if dgrad else 0) // 2
    if out is None:
        out = torch.empty((n,), dtype=torch.bfloat16, device=w_staged.device)
    _lib.check(lib.cs_pack_conv_weights(ctypes.byref(geom), 1 if dgrad else 0, _p(w_staged), _p(out), _stream()), "pack_conv_weights")
    return out


def conv_fwd_packed(geom, x, w_packed, shift=None, residual=None, act=CS_ACT_NONE, want_bits=False, out=None):
    y = out if out is not None else torch.empty((geom.N, geom.P, geom.Q, geom.K), dtype=x.dtype, device=x.device)
    bits = torch.empty((geom.N, geom.P, geom.Q, geom.K // 8), dtype=torch.uint8, device=x.device) if want_bits else None
    lib = _lib.load()
    _lib.check(_timed("fwd", geom, x.dtype, lambda: lib.cs_conv2d_fwd_packed(
        ctypes.byref(geom), _p(x), _p(w_packed), _p(shift), _p(residual), act, _p(y), _p(bits), _stream()),
        extra_tensors=int(residual is not None) + (1.0 / 16 if want_bits else 0.0)), "conv2d_fwd_packed")
    return (y, bits) if want_bits else y


class CompactGrad:
    """The data gradient of a stride-2 1x1 convolution in compact form: `t` [N][P][Q][C] holds the values of the destination pixels
    (2y, 2x); every other pixel of that gradient is zero and was never written.  Only a packed 1x1 data gradient can consume it
    (conv_dgrad_packed(add=CompactGrad))."""

    def __init__(self, t, stride):
        self.t, self.stride = t, stride


def conv_dgrad_packed(geom, dy, w_packed, add=None, mask_bits=None, want_colsum=False):
    """-> dx, or (dx, PartialColsum) with want_colsum.  A stride-2 1x1 geometry returns a CompactGrad (no add / mask / column sums);
    `add` may be such a CompactGrad of the same destination."""
    lib = _lib.load()
    if geom.stride != 1:
        if add is not None or mask_bits is not None or want_colsum:
            raise ValueError("conv_dgrad_packed: a strided 1x1 data gradient is compact -- no add, mask or column sums")
        dxc = torch.empty((geom.N, geom.P, geom.Q, geom.C), dtype=dy.dtype, device=dy.device)
        _lib.check(_timed("dgrad", geom, dy.dtype, lambda: lib.cs_conv2d_dgrad_packed(
            ctypes.byref(geom), _p(dy), _p(w_packed), None, 1, None, _p(dxc), None, _stream()), extra_tensors=-0.75), "conv2d_dgrad_packed")
        return CompactGrad(dxc, geom.stride)
    add_stride = 1
    if isinstance(add, CompactGrad):
        add_stride, add = add.stride, add.t
        if tuple(add.shape) != (geom.N, (geom.H + 1) // 2, (geom.W + 1) // 2, geom.C):
            raise ValueError("conv_dgrad_packed: compact add operand of another destination")
    dx = torch.empty((geom.N, geom.H, geom.W, geom.C), dtype=dy.dtype, device=dy.device)
    rows = lib.cs_conv2d_packed_partial_rows(ctypes.byref(geom), 1) if want_colsum else 0
    ws = torch.empty((rows, 2 * geom.C), dtype=torch.float32, device=dy.device) if want_colsum else None
    _lib.check(_timed("dgrad", geom, dy.dtype, lambda: lib.cs_conv2d_dgrad_packed(
        ctypes.byref(geom), _p(dy), _p(w_packed), _p(add), add_stride, _p(mask_bits), _p(dx), _p(ws), _stream()),
        extra_tensors=(int(add is not None) if add_stride == 1 else 0.25) + (1.0 / 16 if mask_bits is not None else 0.0)), "conv2d_dgrad_packed")
    if want_colsum:
        return dx, PartialColsum(ws, rows, geom.C)
    return dx


# ---------------------------------------------------------------- stem on a pixel-paired image (cs_stem_*)
def is_stem_geom(geom):
    """Conv2d(3 -> K, 7x7, stride 2, padding 3) on an NHWC8 image: the layer the paired path replaces."""
    return geom.R == 7 and geom.S == 7 and geom.stride == 2 and geom.pad == 3 and geom.C == 8


def stem_pair_input(x):
    """x[N,H,W,8] (channels 3.. zero) -> [N,H,ceil(W/2),8]: two neighbouring pixels x 4 channels per 16-byte chunk."""
    N, H, W, C = x.shape
    if C != 8:
        raise ValueError("stem_pair_input expects an NHWC image with 8 stored channels")
    out = torch.empty((N, H, (W + 1) // 2, 8), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().cs_stem_pair_input(_p(x), _code(x.dtype), N, H, W, _p(out), _stream()), "stem_pair_input")
    return out


def stem_pair_from_nchw(x_nchw, dtype):
    """fp32 NCHW image [N,3,H,W] -> the paired stem operand [N,H,ceil(W/2),8] of `dtype` (bf16), without the NHWC8 intermediate."""
    N, C, H, W = x_nchw.shape
    if C != 3 or x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous():
        raise ValueError("stem_pair_from_nchw expects a contiguous fp32 [N,3,H,W] image")
    out = torch.empty((N, H, (W + 1) // 2, 8), dtype=dtype, device=x_nchw.device)
    _lib.check(_lib.load().cs_stem_pair_from_nchw(_p(x_nchw), _code(dtype), N, H, W, _p(out), _stream()), "stem_pair_from_nchw")
    return out


def stem_pair_weights(w_khwc):
    """staged [K,7,7,8] -> paired [K,7,4,8]."""
    K_ = w_khwc.shape[0]
    out = torch.empty((K_, 7, 4, 8), dtype=w_khwc.dtype, device=w_khwc.device)
    _lib.check(_lib.load().cs_stem_pair_weights(_p(w_khwc), _code(w_khwc.dtype), K_, _p(out), _stream()), "stem_pair_weights")
    return out


def stem_fwd(geom, x_pair, w_pair, scale=None, shift=None, act=CS_ACT_NONE, stats=None):
    """cs_conv2d_fwd of the stem geometry `geom` (the 7x7 one: it names the launch for the timers) on paired operands."""
    y = torch.empty((geom.N, geom.P, geom.Q, geom.K), dtype=x_pair.dtype, device=x_pair.device)
    lib = _lib.load()
    ws = _stats_ws(geom.N * geom.P * geom.Q, geom.K, x_pair.device) if stats is not None else None
    _lib.check(_timed("fwd", geom, x_pair.dtype, lambda: lib.cs_stem_fwd(
        geom.N, geom.H, geom.W, geom.K, _code(x_pair.dtype), _p(x_pair), _p(w_pair), _p(scale), _p(shift), act, _p(y), _p(stats), _p(ws),
        _stream())), "stem_fwd")
    return y


def stem_pack_weights(w_pair):
    """paired stem weights [K][7][4][8] bf16 -> the packed operand of stem_fwd_packed: one zero filter row appended ([K][256]) and
    re-ordered like a 1x1 convolution's forward weights."""
    K_ = w_pair.shape[0]
    w256 = torch.zeros((K_, 1, 1, 256), dtype=w_pair.dtype, device=w_pair.device)
    w256.view(K_, 256)[:, :224] = w_pair.reshape(K_, 224)
    return pack_conv_weights(make_geom(1, 2, 2, 256, K_, 1, 1, 1, 0), w256, dgrad=False)


def stem_fwd_packed_supported(geom, dtype):
    return dtype == torch.bfloat16 and is_stem_geom(geom) and geom.K == 64 and geom.P >= 2 and geom.Q >= 2


def stem_fwd_packed(geom, x_pair, w_packed, shift=None, act=CS_ACT_NONE, want_bits=False):
    """The stem forward on the ring kernel (bf16): y, or (y, bits)."""
    y = torch.empty((geom.N, geom.P, geom.Q, geom.K), dtype=x_pair.dtype, device=x_pair.device)
    bits = torch.empty((geom.N, geom.P, geom.Q, geom.K // 8), dtype=torch.uint8, device=x_pair.device) if want_bits else None
    lib = _lib.load()
    _lib.check(_timed("fwd", geom, x_pair.dtype, lambda: lib.cs_stem_fwd_packed(
        geom.N, geom.H, geom.W, geom.K, _p(x_pair), _p(w_packed), _p(shift), act, _p(y), _p(bits), _stream()),
        extra_tensors=1.0 / 16 if want_bits else 0), "stem_fwd_packed")
    return (y, bits) if want_bits else y


def stem_wgrad(geom, x_pair, dy, use_tr_read=True):
    """Raw weight gradient of the stem in the ORDINARY slab layout [1, K, 7, 7, 8] (one slab: the paired split-K slabs are summed
    while they are un-paired), ready for wgrad_finalize / wgrad_finalize_batched."""
    lib = _lib.load()
    nsplit = lib.cs_stem_wgrad_splits(geom.N, geom.H, geom.W, geom.K)
    pair = torch.empty((nsplit, geom.K, 7, 4, 8), dtype=torch.float32, device=dy.device)
    _lib.check(_timed("wgrad", geom, dy.dtype, lambda: lib.cs_stem_wgrad(
        geom.N, geom.H, geom.W, geom.K, _code(dy.dtype), _p(x_pair), _p(dy), _p(pair), 1 if use_tr_read else 0, _stream())), "stem_wgrad")
    raw = torch.empty((1, geom.K, 7, 7, 8), dtype=torch.float32, device=dy.device)
    _lib.check(lib.cs_stem_unpair_slabs(_p(pair), nsplit, geom.K, _p(raw), _stream()), "stem_unpair_slabs")
    return raw


class PartialColsum:
    """Column sums of a dgrad output still in per-workgroup partial rows (cs_conv2d_dgrad with colsum == NULL): the batched
    weight-gradient finalize folds them itself; vector() folds them now for any other consumer."""

    def __init__(self, buf, rows, n_out):
        self.buf, self.rows, self.n_out = buf, rows, n_out
        self._vec = None
        self._vec_stream = None      # the stream the fold ran on (weight gradients may run on a side stream)

    def _set(self, vec):
        self._vec, self._vec_stream = vec, torch.cuda.current_stream()

    def _sync(self):
        cur = torch.cuda.current_stream()
        if self._vec_stream is not None and cur != self._vec_stream:
            cur.wait_stream(self._vec_stream)
        return self._vec

    def vector(self):
        if self._vec is None:
            vec = torch.zeros((self.n_out,), dtype=torch.float32, device=self.buf.device)
            _lib.check(_lib.load().cs_fold_partial_rows(_p(self.buf), self.rows, self.n_out, _p(vec), _stream()), "fold_partial_rows")
            self._set(vec)
        return self._sync()


def colsum_vector(g):
    """A [C] fp32 tensor from either form of column sums."""
    return g.vector() if isinstance(g, PartialColsum) else g


def conv_dgrad(geom, dy, w_chwk, add=None, mask=None, colsum=None, grouped=False, defer_colsum=False, mask_bits=None):
    """colsum: zeroed fp32 [C] to accumulate the column sums of dx into; or defer_colsum=True (stride 1, ungrouped) to get
    (dx, PartialColsum) with the fold left to the consumer.  mask_bits: uint8 [N,H,W,C/8] from conv_fwd(want_bits=True), used
    instead of `mask`."""
    dx = torch.empty((geom.N, geom.H, geom.W, geom.C), dtype=dy.dtype, device=dy.device)
    lib = _lib.load()
    geom = _grouped_geom(geom, grouped)
    # (stride 2: deferrable when the library merges the parity classes into one launch -- it then reports the row count, else 0)
    defer = defer_colsum and not grouped and (geom.stride == 1 or lib.cs_conv2d_dgrad_partial_rows(ctypes.byref(geom)) > 0)
    ws = _stats_ws(geom.N * geom.H * geom.W, geom.C, dy.device) if (colsum is not None or defer) else None
    if mask_bits is not None:
        _lib.check(_timed("dgrad", geom, dy.dtype, lambda: lib.cs_conv2d_dgrad_bits(
            ctypes.byref(geom), _code(dy.dtype), _p(dy), _p(w_chwk), _p(add), _p(mask_bits), _p(dx), None if defer else _p(colsum), _p(ws),
            _stream()), extra_tensors=int(add is not None) + 1.0 / (8 * dx.element_size())), "conv2d_dgrad_bits")
    else:
        _lib.check(_timed("dgrad", geom, dy.dtype, lambda: lib.cs_conv2d_dgrad(
            ctypes.byref(geom), _code(dy.dtype), _p(dy), _p(w_chwk), _p(add), _p(mask), _p(dx), None if defer else _p(colsum), _p(ws), _stream()),
            extra_tensors=int(add is not None) + int(mask is not None)), "conv2d_dgrad")
    if defer_colsum:
        if defer:
            return dx, PartialColsum(ws, lib.cs_conv2d_dgrad_partial_rows(ctypes.byref(geom)), geom.C)
        return dx, colsum
    return dx


def wgrad_splits(geom, grouped=False):
    return _lib.load().cs_conv2d_wgrad_splits(ctypes.byref(geom), 1 if grouped else 0)


def new_wgrad_buffer(geom, device, grouped=False):
    """[nsplit, K, R, S, Cp|64] fp32 split-K partial slabs (no zero-fill needed)."""
    return torch.empty((wgrad_splits(geom, grouped), geom.K, geom.R, geom.S, 64 if grouped else geom.C), dtype=torch.float32, device=device)


def conv_wgrad(geom, x, dy, dw_raw, use_tr_read=True, grouped=False):
    """dw_raw = new_wgrad_buffer(...): every slab is fully written by the kernel."""
    if dw_raw.dim() != 5 or dw_raw.shape[0] != wgrad_splits(geom, grouped):
        raise ValueError("conv_wgrad: dw_raw must come from new_wgrad_buffer(geom, ...)")
    lib = _lib.load()
    geom = _grouped_geom(geom, grouped)
    _lib.check(_timed("wgrad", geom, x.dtype, lambda: lib.cs_conv2d_wgrad(
        ctypes.byref(geom), _code(x.dtype), _p(x), _p(dy), _p(dw_raw), 1 if use_tr_read else 0, _stream())), "conv2d_wgrad")
    return dw_raw


def wgrad2_serves(geom, dtype):
    """True when the batched weight-gradient entry runs this geometry on the second-generation kernel (csrc/wgrad_v2.hip)."""
    return bool(_lib.load().cs_conv2d_wgrad2_supported(ctypes.byref(geom), _code(dtype)))


def wgrad_batched(geom, xs, dys, use_tr_read=True):
    """Batched wgrad over len(xs) <= 8 layers of identical geometry.  Returns the split-K slab buffer [n, nsplit, K, R, S, Cp]."""
    n = len(xs)
    lib = _lib.load()
    nsplit = lib.cs_conv2d_wgrad_batched_splits(ctypes.byref(geom), _code(xs[0].dtype), n)
    slabs = torch.empty((n, nsplit, geom.K, geom.R, geom.S, geom.C), dtype=torch.float32, device=xs[0].device)
    for t in list(xs) + list(dys):
        _p(t)                                  # device / contiguity checks
    arr = ctypes.c_void_p * n
    xt, dt_, wt = arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dys]), arr(*[slabs[i].data_ptr() for i in range(n)])
    _lib.check(_timed("wgrad", geom, xs[0].dtype, lambda: lib.cs_conv2d_wgrad_batched(
        ctypes.byref(geom), _code(xs[0].dtype), xt, dt_, wt, n, 1 if use_tr_read else 0, _stream()), batch=n), "conv2d_wgrad_batched")
    return slabs


def wgrad_finalize_batched(slabs, ws, scales, rstds, means, gsums, dws, dgammas, dbetas, Cin):
    """Batched finalize for the eval-BN trunk: lists of n <= 8 tensors each, ONE launch.  gsums: [K] vectors or PartialColsum objects
    (deferred column sums of a data-gradient launch: the kernel folds the partial rows of its channel itself)."""
    n, nsplit, Kp, R, S, Cp = slabs.shape
    K_ = dws[0].shape[0]
    want_bn = dgammas is not None

    def ptrs(lst):
        return [None] * n if lst is None else [t.data_ptr() for t in lst]

    grows, gstride, gptr = [0] * n, 0, [None] * n
    if gsums is not None:
        for i, g in enumerate(gsums):
            if isinstance(g, PartialColsum) and g._vec is None:
                grows[i], gptr[i] = g.rows, g.buf.data_ptr()
                if gstride not in (0, 2 * g.n_out):
                    raise ValueError("wgrad_finalize_batched: mixed partial-row strides")
                gstride = 2 * g.n_out           # partial rows are [rows][2][n_out] in every producer
            else:
                gptr[i] = colsum_vector(g).data_ptr()
    flat = [slabs[i].data_ptr() for i in range(n)] + ptrs(ws) + ptrs(scales) + ptrs(rstds) + ptrs(means) + gptr + ptrs(dws) + \
        ptrs(dgammas) + ptrs(dbetas)
    table = (ctypes.c_void_p * len(flat))(*flat)
    rows_arr = (ctypes.c_int * n)(*grows)
    _lib.check(_lib.load().cs_wgrad_finalize_batched(table, rows_arr, gstride, n, nsplit, Kp, K_, Cin, R, S, Cp, 1 if want_bn else 0,
                                                     _stream()), "wgrad_finalize_batched")


def wgrad_finalize(dw_raw, w, scale, rstd, mean, gsum, Cin, dw, dbias=None, dgamma=None, dbeta=None, accumulate=False, dot=None):
    """dot: zeroed fp32 [K] scratch (allocated here when omitted)."""
    nsplit, Kp, R, S, Cp = dw_raw.shape
    K = dw.shape[0]
    if dgamma is not None and dot is None:
        dot = torch.zeros((K,), dtype=torch.float32, device=dw.device)
    _lib.check(_lib.load().cs_wgrad_finalize(_p(dw_raw), nsplit, Kp, _p(w), _p(scale), _p(rstd), _p(mean), _p(gsum), K, Cin, R, S, Cp,
                                             _p(dw), _p(dbias), _p(dgamma), _p(dbeta), _p(dot), 1 if accumulate else 0, _stream()),
               "wgrad_finalize")


def colsum(g, out=None):
    """Per-channel sums of an NHWC tensor.  Without `out`: per-workgroup partial rows + one fold (no atomics: 512 adds per
    address serialise at the memory side and cost 3x the streaming time); with `out`: accumulated into it atomically."""
    C = g.shape[-1]
    M = g.numel() // C
    if out is None:
        return colsum_partial(g).vector()
    _lib.check(_lib.load().cs_colsum(_p(g), _code(g.dtype), M, C, _p(out), _stream()), "colsum")
    return out


def positive_bits(x):
    """Bit plane of a bf16 NHWC tensor: bit = x[..., c] > 0, in the channel-block-major layout of the convolution epilogues (the 32-bit
    word (c // 32) * M + pixel holds channels 32 (c // 32) .. + 31 of that pixel; unpack_bits / pack_bits translate) -- the `mask_bits`
    operand of the packed data gradients for a post-ReLU tensor that no convolution epilogue produced (train-mode BN + ReLU outputs,
    their concatenation).  Returned as uint8 [..., C/8] (a shape carrier: the bytes are NOT in NHWC order)."""
    C = x.shape[-1]
    if x.dtype != torch.bfloat16 or C % 32 or not x.is_contiguous():
        raise ValueError("positive_bits: contiguous bf16 tensor with a channel count that is a multiple of 32")
    bits = torch.empty(x.shape[:-1] + (C // 8,), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().cs_positive_bits(_p(x), _code(x.dtype), x.numel() // C, C, _p(bits), _stream()), "positive_bits")
    return bits


def unpack_bits(bits, C):
    """bool [..., C] from a bit plane of the library (uint8, M * C / 8 bytes, channel-block-major: see positive_bits)."""
    M = bits.numel() * 8 // C
    words = bits.contiguous().view(-1).view(C // 32, M, 4)                       # [block][pixel][byte of the little-endian word]
    sh = torch.arange(8, dtype=torch.uint8, device=bits.device)
    b = ((words.unsqueeze(-1) >> sh) & 1).bool()                                  # [block][pixel][byte][bit]
    return b.permute(1, 0, 2, 3).reshape(tuple(bits.shape[:-1]) + (C,))


def pack_bits(mask):
    """The inverse: a bool / 0-1 tensor [..., C] (C % 32 == 0) -> uint8 [..., C/8] in the library's bit-plane layout."""
    C = mask.shape[-1]
    M = mask.numel() // C
    b = mask.reshape(M, C // 32, 4, 8).to(torch.uint8)
    w = (b << torch.arange(8, dtype=torch.uint8, device=mask.device)).sum(-1).to(torch.uint8)       # [pixel][block][byte]
    return w.permute(1, 0, 2).contiguous().view(tuple(mask.shape[:-1]) + (C // 8,))


def colsum_partial(g):
    """Column sums as a PartialColsum (per-workgroup partial rows, folded later together with its group's other sums)."""
    C = g.shape[-1]
    M = g.numel() // C
    lib = _lib.load()
    rows = lib.cs_colsum_partial_rows(M)
    buf = torch.empty((rows, 2 * C), dtype=torch.float32, device=g.device)
    _lib.check(lib.cs_colsum_partial(_p(g), _code(g.dtype), M, C, _p(buf), _stream()), "colsum_partial")
    return PartialColsum(buf, rows, C)


# ---------------------------------------------------------------- BatchNorm (train mode)
# Zero-initialised fp64 accumulators ([2, C] per BatchNorm reduction) come out of one pool per device that the engine clears ONCE
# per training pass (begin_pass) instead of one fill launch per accumulator (EfficientNet-B3: 104 fills of 4.8 us per step).
# Every user consumes its accumulator inside the call that took it (stream order), so clearing the whole pool at the start of the
# next pass is safe; without begin_pass (direct calls from tests / tools) the pool is used once and then falls back to torch.zeros.
_STATS_POOL_ELEMS = 1 << 21           # 8-byte words (16 MB): an exact accumulator takes 6 C + 1 of them (cs_bn_accum_words)
_stats_pools = {}


def accum_words(C):
    """8-byte words of the exact per-channel accumulator of a C-channel BatchNorm reduction (cs_bn_accum_words: three fp64 limbs per sum
    and channel, each added to exactly, + one flag word; order-independent adds, so batch statistics repeat bit for bit)."""
    return 6 * C + 1


def begin_pass(device):
    pool = _stats_pools.get(device)
    if pool is None:
        _stats_pools[device] = [torch.zeros((_STATS_POOL_ELEMS,), dtype=torch.float64, device=device), 0]
    else:
        pool[0].zero_()
        pool[1] = 0


def new_stats(C, device):
    """A zeroed exact accumulator for C channels (opaque words, dtype float64 for history; stats_values() reads the totals)."""
    n = accum_words(C)
    pool = _stats_pools.get(device)
    if pool is not None and pool[1] + n <= _STATS_POOL_ELEMS:
        v = pool[0][pool[1]:pool[1] + n]
        pool[1] += n
        return v
    return torch.zeros((n,), dtype=torch.float64, device=device)


def stats_channels(stats):
    n = stats.numel()
    if n < 7 or (n - 1) % 6:
        raise ValueError("not a BatchNorm accumulator (kernels.new_stats)")
    return (n - 1) // 6


def stats_values(stats):
    """fp64 [2, C]: the two per-channel totals an exact accumulator holds (cs_bn_accum_read; tests and tools)."""
    C = stats_channels(stats)
    out = torch.empty((2, C), dtype=torch.float64, device=stats.device)
    _lib.check(_lib.load().cs_bn_accum_read(_p(stats), C, _p(out), _stream()), "bn_accum_read")
    return out


def bn_stats(z, stats=None):
    C = z.shape[-1]
    M = z.numel() // C
    if stats is None:
        stats = new_stats(C, z.device)
    _lib.check(_lib.load().cs_bn_stats(_p(z), _code(z.dtype), M, C, _p(stats), _p(_bn_ws(M, C, z.device)), _stream()), "bn_stats")
    return stats


def _bn_ws(M, C, device):
    """Partial-sum workspace of the BN reductions, or None where the library wants none (cs_bn_partial_workspace returns 0: few
    workgroups, fp64 atomics -- nearly every layer; the decision is the library's alone)."""
    nbytes = _lib.load().cs_bn_partial_workspace(M, C)
    if nbytes == 0:
        return None
    return torch.empty((nbytes // 8,), dtype=torch.float64, device=device)


def bn_finalize(stats, M, eps, momentum, running_mean=None, running_var=None):
    C = stats_channels(stats)
    out = torch.empty((2, C), dtype=torch.float32, device=stats.device)
    _lib.check(_lib.load().cs_bn_finalize(_p(stats), M, eps, momentum, _p(running_mean), _p(running_var), _p(out[0]), _p(out[1]), C,
                                          _stream()), "bn_finalize")
    return out[0], out[1]


def bn_apply(z, mean, rstd, gamma, beta, residual=None, act=CS_ACT_NONE, out=None):
    C = z.shape[-1]
    M = z.numel() // C
    y = out if out is not None else torch.empty_like(z)
    _lib.check(_lib.load().cs_bn_apply(_p(z), _code(z.dtype), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual), act, _p(y), M, C,
                                       _stream()), "bn_apply")
    return y


def bn_apply_stats(z, stats, eps, momentum, running_mean, running_var, gamma, beta, residual=None, act=CS_ACT_NONE):
    """bn_finalize + bn_apply as one launch: returns (y, mean, rstd); the running statistics (nullable) are updated in place."""
    C = z.shape[-1]
    M = z.numel() // C
    y = torch.empty_like(z)
    out = torch.empty((2, C), dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().cs_bn_apply_stats(_p(z), _code(z.dtype), _p(stats), eps, momentum, _p(running_mean), _p(running_var), _p(gamma),
                                             _p(beta), _p(residual), act, _p(y), _p(out[0]), _p(out[1]), M, C, _stream()), "bn_apply_stats")
    return y, out[0], out[1]


def bn_bwd(dy, z, mean, rstd, gamma, want_param_grads=True, beta=None, act=CS_ACT_NONE):
    """returns dz, dgamma, dbeta.  act=CS_ACT_SILU differentiates through the SiLU that follows the BN."""
    C = z.shape[-1]
    M = z.numel() // C
    sums = new_stats(C, z.device)
    lib = _lib.load()
    _lib.check(lib.cs_bn_bwd_reduce(_p(dy), _p(z), _code(z.dtype), _p(mean), _p(rstd), _p(gamma), _p(beta), act, M, C, _p(sums),
                                    _p(_bn_ws(M, C, z.device)), _stream()), "bn_bwd_reduce")
    dz = torch.empty_like(z)
    dg = torch.empty((2, C), dtype=torch.float32, device=z.device) if want_param_grads else None
    _lib.check(lib.cs_bn_bwd_apply(_p(dy), _p(z), _code(z.dtype), _p(mean), _p(rstd), _p(gamma), _p(beta), act, _p(sums), M, C, _p(dz),
                                   _p(dg[0]) if dg is not None else None, _p(dg[1]) if dg is not None else None, _stream()),
               "bn_bwd_apply")
    return dz, (dg[0] if dg is not None else None), (dg[1] if dg is not None else None)


# ---------------------------------------------------------------- decoder data movement
def bilinear_fwd(x, out_hw):
    N, H, W, C = x.shape
    P, Q = out_hw
    y = torch.empty((N, P, Q, C), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().cs_bilinear_ac_fwd(_p(x), _code(x.dtype), _p(y), N, H, W, C, P, Q, _stream()), "bilinear_fwd")
    return y


def bilinear_bwd(dy, in_hw, mask=None):
    N, P, Q, C = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().cs_bilinear_ac_bwd(_p(dy), _p(mask), _code(dy.dtype), _p(dx), N, H, W, C, P, Q, _stream()), "bilinear_bwd")
    return dx


def concat(a, b):
    Ca, Cb = a.shape[-1], b.shape[-1]
    M = a.numel() // Ca
    out = torch.empty(tuple(a.shape[:-1]) + (Ca + Cb,), dtype=a.dtype, device=a.device)
    _lib.check(_lib.load().cs_concat_channels(_p(a), _p(b), _code(a.dtype), _p(out), M, Ca, Cb, _stream()), "concat")
    return out


def split(whole, Ca, want_a=True, want_b=True):
    C = whole.shape[-1]
    Cb = C - Ca
    M = whole.numel() // C
    a = torch.empty(tuple(whole.shape[:-1]) + (Ca,), dtype=whole.dtype, device=whole.device) if want_a else None
    b = torch.empty(tuple(whole.shape[:-1]) + (Cb,), dtype=whole.dtype, device=whole.device) if want_b else None
    _lib.check(_lib.load().cs_split_channels(_p(whole), _code(whole.dtype), _p(a), _p(b), M, Ca, Cb, _stream()), "split")
    return a, b


# ---------------------------------------------------------------- MBConv pieces
def dwconv_fwd(geom, x, w_hwc, scale=None, shift=None, act=CS_ACT_NONE):
    y = torch.empty((geom.N, geom.P, geom.Q, geom.C), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().cs_dwconv_fwd(ctypes.byref(geom), _code(x.dtype), _p(x), _p(w_hwc), _p(scale), _p(shift), act, _p(y), _stream()),
               "dwconv_fwd")
    return y


def dwconv_fwd_stats(geom, x, w_hwc):
    """Raw depthwise output z and its train-mode BN statistics (fp64 [2, C]) from one pass."""
    lib = _lib.load()
    y = torch.empty((geom.N, geom.P, geom.Q, geom.C), dtype=x.dtype, device=x.device)
    ws = torch.empty((lib.cs_dwconv_fwd_stats_workspace(ctypes.byref(geom)) // 8,), dtype=torch.float64, device=x.device)
    rows = ctypes.c_int(0)
    _lib.check(lib.cs_dwconv_fwd_stats(ctypes.byref(geom), _code(x.dtype), _p(x), _p(w_hwc), _p(y), _p(ws), ctypes.byref(rows), _stream()),
               "dwconv_fwd_stats")
    stats = new_stats(geom.C, x.device)
    _lib.check(lib.cs_bn_partial_fold(_p(ws), rows.value, geom.C, _p(stats), _stream()), "bn_partial_fold")
    return y, stats


def dwconv_dgrad(geom, dy, w_hwc):
    dx = torch.empty((geom.N, geom.H, geom.W, geom.C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().cs_dwconv_dgrad(ctypes.byref(geom), _code(dy.dtype), _p(dy), _p(w_hwc), _p(dx), _stream()), "dwconv_dgrad")
    return dx


def dwconv_wgrad(geom, x, dy, param_layout=False):
    """fp32 [R, S, C]; param_layout: [C, 1, R, S], the layout of nn.Conv2d(groups=C).weight (no permute + copy afterwards)."""
    lib = _lib.load()
    ws = torch.empty((max(lib.cs_dwconv_wgrad_workspace(ctypes.byref(geom)) // 4, 1),), dtype=torch.float32, device=x.device)
    if param_layout:
        dw = torch.empty((geom.C, 1, geom.R, geom.S), dtype=torch.float32, device=x.device)
        _lib.check(lib.cs_dwconv_wgrad_oihw(ctypes.byref(geom), _code(x.dtype), _p(x), _p(dy), _p(dw), _p(ws), _stream()), "dwconv_wgrad_oihw")
        return dw
    dw = torch.empty((geom.R, geom.S, geom.C), dtype=torch.float32, device=x.device)
    _lib.check(lib.cs_dwconv_wgrad(ctypes.byref(geom), _code(x.dtype), _p(x), _p(dy), _p(dw), _p(ws), _stream()), "dwconv_wgrad")
    return dw


class DwStagePack:
    """The depthwise filters of a plan in the layout the depthwise kernels read ([R, S, C] fp32), restaged from the parameters by ONE launch
    per forward (cs_dw_weights_hwc_multi).  `weights`: the nn.Conv2d(groups=C).weight parameters, [C, 1, R, S] fp32 contiguous."""

    def __init__(self, weights):
        dev = weights[0].device
        sizes = [w.numel() for w in weights]
        self.flat = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
        self.key = tuple(w.data_ptr() for w in weights)
        self.hwc, rows, first = [], [], 0
        for w, n in zip(weights, sizes):
            C, RS = w.shape[0], w.shape[2] * w.shape[3]
            dst = self.flat[first:first + n].view(w.shape[2], w.shape[3], C)
            self.hwc.append(dst)
            rows.append([w.data_ptr(), dst.data_ptr(), C, RS, first])
            first += n
        self.total = first
        self.desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.n = len(rows)

    def run(self):
        _lib.check(_lib.load().cs_dw_weights_hwc_multi(_p(self.desc), self.n, self.total, _stream()), "dw_weights_hwc_multi")


def se_scale(x, s):
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().cs_se_scale(_p(x), _code(x.dtype), _p(s), _p(y), N, H * W, C, _stream()), "se_scale")
    return y


def sample_sum(a, b=None, scale=1.0):
    """fp32 [N, C]: scale * sum over the pixels of a (* b) per sample and channel (cs_sample_sum: row-strided, fixed-order fold)."""
    N, H, W, C = a.shape
    lib = _lib.load()
    out = torch.empty((N, C), dtype=torch.float32, device=a.device)
    ws = torch.empty((lib.cs_sample_sum_workspace(N, H * W, C) // 4,), dtype=torch.float32, device=a.device)
    _lib.check(lib.cs_sample_sum(_p(a), _p(b), _code(a.dtype), float(scale), _p(out), _p(ws), N, H * W, C, _stream()), "sample_sum")
    return out


def se_scale_bwd_ds(dy, x):
    return sample_sum(dy, x)


def se_scale_bwd_dx(dy, s, davg):
    N, H, W, C = dy.shape
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().cs_se_scale_bwd(_p(dy), None, _code(dy.dtype), _p(s), _p(davg), None, _p(dx), N, H * W, C, 1, _stream()),
               "se_scale_bwd(dx)")
    return dx


def rowscale_add(a, row_scale, b):
    N = a.shape[0]
    y = torch.empty_like(a)
    _lib.check(_lib.load().cs_rowscale_add(_p(a), _code(a.dtype), _p(row_scale), _p(b), _p(y), N, a.numel() // N, _stream()), "rowscale_add")
    return y


# ---------------------------------------------------------------- pooling
def maxpool_fwd(x, want_argmax=True):
    N, H, W, C = x.shape
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, P, Q, C), dtype=x.dtype, device=x.device)
    am = torch.empty((N, P, Q, C), dtype=torch.uint8, device=x.device) if want_argmax else None
    _lib.check(_lib.load().cs_maxpool3x3s2_fwd(_p(x), _code(x.dtype), _p(y), _p(am), N, H, W, C, P, Q, _stream()), "maxpool_fwd")
    return y, am


def maxpool_bwd(dy, argmax, y_mask, in_hw):
    N, P, Q, C = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().cs_maxpool3x3s2_bwd(_p(dy), _p(argmax), _p(y_mask), _code(dy.dtype), _p(dx), N, H, W, C, P, Q,
                                               _stream()), "maxpool_bwd")
    return dx


def gap_fwd(x, with_max=True):
    N, H, W, C = x.shape
    if not with_max:
        return sample_sum(x, None, 1.0 / (H * W)), None          # the squeeze-excitation average pool
    feat = torch.empty((N, C), dtype=torch.float32, device=x.device)
    am = torch.empty((N, C), dtype=torch.int32, device=x.device) if with_max else None
    _lib.check(_lib.load().cs_gap_avgmax_fwd(_p(x), _code(x.dtype), _p(feat), _p(am), N, H * W, C, 1 if with_max else 0, _stream()),
               "gap_fwd")
    return feat, am


def gap_bwd(dfeat, argmax, x, relu_mask, with_max=True):
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.load().cs_gap_avgmax_bwd(_p(dfeat), _p(argmax), _p(x), _code(x.dtype), _p(dx), N, H * W, C,
                                             1 if relu_mask else 0, 1 if with_max else 0, _stream()), "gap_bwd")
    return dx


# ---------------------------------------------------------------- heads / losses
_LOSS_WORDS = 16            # cs_loss_words(): floats behind a CE / MSE value (the value + the exact accumulator of its partial sums)
def linear_fwd(x, w, b, act=CS_ACT_NONE, want_preact=False):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if want_preact else None
    _lib.check(_lib.load().cs_linear_fwd(_p(x), _p(w), _p(b), _p(y), _p(pre), M, N, K, act, _stream()), "linear_fwd")
    return (y, pre) if want_preact else y


def linear_bwd(x, w, dy, y=None, act=CS_ACT_NONE, need_dx=True, need_dw=True, need_db=True):
    M, K = x.shape
    N = w.shape[0]
    dx = torch.empty((M, K), dtype=torch.float32, device=x.device) if need_dx else None
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device) if (need_dw or need_db) else None
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if need_db else None
    lib = _lib.load()
    nws = lib.cs_linear_bwd_workspace(M, N, K) if dw is not None else 0          # (> 512 rows: row slices of the weight gradient)
    ws = torch.empty((nws // 4,), dtype=torch.float32, device=x.device) if nws else None
    _lib.check(lib.cs_linear_bwd(_p(x), _p(w), _p(dy), _p(y), act, _p(dx), _p(dw), _p(db), M, N, K, 0, _p(ws), _stream()),
               "linear_bwd")
    return dx, dw, db


def softmax_ce(logits, labels, gamma=1.0, want_grad=True):
    M, C = logits.shape
    loss = torch.empty((_LOSS_WORDS,), dtype=torch.float32, device=logits.device)      # [0] = the value, the rest: its exact accumulator
    dl = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.load().cs_softmax_ce(_p(logits), _p(labels), float(gamma), _p(loss), _p(dl), M, C, _stream()), "softmax_ce")
    return loss[:1], dl


def softmax_prob1(logits):
    M, C = logits.shape
    p1 = torch.empty((M,), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().cs_softmax_prob1(_p(logits), _p(p1), M, C, _stream()), "softmax_prob1")
    return p1


def softmax_argmax(logits):
    """np.argmax(F.softmax(logits, 1), axis=1) of inference.py:72-76 on the device: int64 [M], first index on probability ties."""
    M, C = logits.shape
    idx = torch.empty((M,), dtype=torch.int64, device=logits.device)
    _lib.check(_lib.load().cs_softmax_argmax(_p(logits), _p(idx), M, C, _stream()), "softmax_argmax")
    return idx


def mse(x, t, weighted=False, mean=True, want_grad=True):
    M = x.numel()
    loss = torch.empty((_LOSS_WORDS,), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if want_grad else None
    _lib.check(_lib.load().cs_mse(_p(x), _p(t), 1 if weighted else 0, 1 if mean else 0, _p(loss), _p(dx), M, _stream()), "mse")
    return loss[:1], dx


def dice_fwd(p, t, eps=1e-6, mean=True):
    """p, t: [N, HW] fp32 contiguous. returns loss[1], sums: the per-sample (sum p t, sum p^2, sum t^2) as an exact accumulator of
    accum_words(3 N) opaque words (cleared by the call, read by dice_bwd)"""
    N, HW = p.shape
    sums = torch.empty((accum_words(3 * N),), dtype=torch.float64, device=p.device)
    loss = torch.empty((1,), dtype=torch.float32, device=p.device)
    _lib.check(_lib.load().cs_dice_fwd(_p(p), _p(t), N, HW, eps, 1 if mean else 0, _p(sums), _p(loss), _stream()), "dice_fwd")
    return loss, sums


def dice_sums_values(sums, N):
    """fp64 [N, 3]: the per-sample (sum p t, sum p^2, sum t^2) that dice_fwd left in its exact accumulator."""
    return stats_values(sums)[0].view(N, 3)


def dice_bwd(p, t, sums, eps=1e-6, mean=True):
    N, HW = p.shape
    dp = torch.empty_like(p)
    _lib.check(_lib.load().cs_dice_bwd(_p(p), _p(t), _p(sums), N, HW, eps, 1 if mean else 0, _p(dp), _stream()), "dice_bwd")
    return dp


def softmax_channel_fwd(logits, ch=1):
    """logits [N,C,H,W] fp32 -> softmax over C, channel ch: [N,H,W]"""
    N, C, H, W = logits.shape
    pc = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().cs_softmax_channel_fwd(_p(logits), _p(pc), N, C, H * W, ch, _stream()), "softmax_channel_fwd")
    return pc


def softmax_channel_bwd(logits, dpc, ch=1):
    N, C, H, W = logits.shape
    dl = torch.empty_like(logits)
    _lib.check(_lib.load().cs_softmax_channel_bwd(_p(logits), _p(dpc), _p(dl), N, C, H * W, ch, _stream()), "softmax_channel_bwd")
    return dl


def segmented_topk(probs, groups, k_per_tile, seg_offsets, max_run):
    """Device-side order[index] of inference.py:31-43. Returns (out_idx[T] int64, count[1] int64)."""
    T = probs.numel()
    lib = _lib.load()
    ws_bytes = lib.cs_segmented_topk_workspace(T)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=probs.device)
    out = torch.empty((T,), dtype=torch.int64, device=probs.device)
    cnt = torch.zeros((1,), dtype=torch.int64, device=probs.device)
    _lib.check(lib.cs_segmented_topk(_p(probs), _p(groups), _p(k_per_tile), _p(seg_offsets), seg_offsets.numel() - 1,
                                     int(max_run), T, _p(out), _p(cnt), _p(ws), ws_bytes, _stream()), "segmented_topk")
    return out, cnt


# ---------------------------------------------------------------- either side of the top-k
def segmented_order(probs, seg_offsets, max_run):
    """np.lexsort((probs, groups)) for non-decreasing groups: int64 [T] on the device."""
    T = probs.numel()
    order = torch.empty((T,), dtype=torch.int64, device=probs.device)
    _lib.check(_lib.load().cs_segmented_order(_p(probs), _p(seg_offsets), seg_offsets.numel() - 1, int(max_run), T, _p(order), _stream()),
               "segmented_order")
    return order


def threshold_select(probs, order, threshold):
    """order[probs[order] > threshold] -> (out_idx[T] int64, count[1] int64)."""
    T = probs.numel()
    lib = _lib.load()
    ws_bytes = lib.cs_segmented_topk_workspace(T)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=probs.device)
    out = torch.empty((T,), dtype=torch.int64, device=probs.device)
    cnt = torch.zeros((1,), dtype=torch.int64, device=probs.device)
    _lib.check(lib.cs_threshold_select(_p(probs), _p(order), T, float(threshold), _p(out), _p(cnt), _p(ws), ws_bytes, _stream()), "threshold_select")
    return out, cnt


def evaluate_tile_counts(probs, order, groups, pos_from, threshold):
    """int64 [4] on the device: #(pred != real), #(pred & !real), #(!pred & real), #real."""
    counts = torch.zeros((4,), dtype=torch.int64, device=probs.device)
    _lib.check(_lib.load().cs_evaluate_tile_counts(_p(probs), _p(order), _p(groups), _p(pos_from), float(threshold), probs.numel(), _p(counts),
                                                   _stream()), "evaluate_tile_counts")
    return counts


def paint_tile_masks(selected, n_selected, groups, tile_xy, tile_size, n_images, H, W):
    """uint8 [n_images, H, W] masks with a tile_size^2 block of ones per selected tile."""
    masks = torch.zeros((n_images, H, W), dtype=torch.uint8, device=groups.device)
    _lib.check(_lib.load().cs_paint_tile_masks(_p(selected) if n_selected else None, int(n_selected), _p(groups), _p(tile_xy), int(tile_size), H, W,
                                               _p(masks), _stream()), "paint_tile_masks")
    return masks


def prune_excess(labels, flag, n_excess):
    """Positions that survive deleting the first n_excess entries with labels == flag -> (kept[n] int64, count[1] int64)."""
    n = labels.numel()
    kept = torch.empty((n,), dtype=torch.int64, device=labels.device)
    cnt = torch.zeros((1,), dtype=torch.int64, device=labels.device)
    _lib.check(_lib.load().cs_prune_excess(_p(labels), n, int(flag), int(n_excess), _p(kept), _p(cnt), _stream()), "prune_excess")
    return kept, cnt
