"""Plan executor: the explicit forward/backward kernel schedule of a CNN trunk on MI355X.

A *plan* is a list of fused units over numbered tensor slots (NHWC, bf16 or fp32):

* ``ConvUnit``     Conv2d (+bias) [+ BatchNorm2d] [+ residual add] [+ ReLU]
                   - BN in eval mode (``freeze_bn`` / ``model.eval()``): BN is folded into the
                     staged weights + a per-channel shift, everything fused in the conv epilogue;
                     its backward derives dgamma/dbeta from the raw weight gradient (no conv
                     output is kept).
                   - BN in train mode: conv (+fused per-channel fp64 statistics) -> finalize ->
                     normalise+residual+ReLU kernel; backward = 2-pass BN backward + dgrad/wgrad.
* ``PoolUnit``     MaxPool2d(3,2,1)
* ``UpsampleUnit`` bilinear, align_corners=True
* ``ConcatUnit``   channel concat

Gradient convention: a gradient w.r.t. a post-ReLU tensor is always stored already multiplied
by [tensor > 0]; the kernel producing the LAST contribution to a slot applies the mask (dgrad and
bilinear-backward epilogues), so ReLU backward never costs a separate pass, and residual
gradients are passed as the ``add`` operand of the dgrad epilogue instead of a separate add.

The whole plan is ONE torch.autograd.Function: torch only sees (inputs, parameters) -> outputs.
"""
import os
from types import SimpleNamespace

import torch

from . import _capture
from . import kernels as K

ACT_NONE, ACT_RELU, ACT_SILU = K.CS_ACT_NONE, K.CS_ACT_RELU, K.CS_ACT_SILU


class ConvUnit:
    kind = "conv"

    def __init__(self, name, conv, bn, act, src, dst, res=None):
        self.name, self.conv, self.bn, self.act = name, conv, bn, act
        self.src, self.dst, self.res = src, dst, res
        self.grouped = conv.groups != 1
        if self.grouped:
            cg = conv.in_channels // conv.groups
            if conv.in_channels != conv.out_channels or conv.in_channels % 64 != 0 or 64 % cg != 0 or conv.bias is not None:
                raise NotImplementedError(f"{name}: grouped convolution needs C == K, C % 64 == 0, (C/groups) | 64, no bias "
                                          "(the ResNeXt 32x4d / 32x8d 3x3 layers)")
        self._cache = None

    def params(self):
        out = [("weight", self.conv.weight)]
        if self.conv.bias is not None:
            out.append(("bias", self.conv.bias))
        if self.bn is not None:
            out += [("gamma", self.bn.weight), ("beta", self.bn.bias)]
        return out

    def inputs(self):
        return [self.src] + ([self.res] if self.res is not None else [])


class DwConvUnit:
    """Depthwise k x k Conv2d(groups=C, bias=False) + BatchNorm2d + activation (MBConv, efficientnet.py:101-103)."""
    kind = "dw"

    def __init__(self, name, conv, bn, act, src, dst):
        self.name, self.conv, self.bn, self.act, self.src, self.dst = name, conv, bn, act, src, dst
        if conv.groups != conv.in_channels or conv.in_channels != conv.out_channels or conv.bias is not None:
            raise ValueError(f"{name}: not a bias-free depthwise convolution")

    def params(self):
        return [("weight", self.conv.weight), ("gamma", self.bn.weight), ("beta", self.bn.bias)]

    def inputs(self):
        return [self.src]


class SEUnit:
    """torchvision SqueezeExcitation: x * sigmoid(fc2(silu(fc1(avgpool(x)))))  (efficientnet.py:105-107)."""
    kind = "se"

    def __init__(self, name, fc1, fc2, src, dst):
        self.name, self.fc1, self.fc2, self.src, self.dst = name, fc1, fc2, src, dst

    def params(self):
        return [("w1", self.fc1.weight), ("b1", self.fc1.bias), ("w2", self.fc2.weight), ("b2", self.fc2.bias)]

    def inputs(self):
        return [self.src]


class RowScaleAddUnit:
    """StochasticDepth(p, "row") on `a`, then + `b` (efficientnet.py:116-121); only planned when training with p > 0."""
    kind = "sd"

    def __init__(self, a, b, dst, p):
        self.a, self.b, self.dst, self.p = a, b, dst, p

    def params(self):
        return []

    def inputs(self):
        return [self.a, self.b]


class PoolUnit:
    kind = "pool"

    def __init__(self, src, dst):
        self.src, self.dst = src, dst

    def params(self):
        return []

    def inputs(self):
        return [self.src]


class UpsampleUnit:
    kind = "up"

    def __init__(self, src, dst, size_like=None, size_fn=None):
        """output size = spatial size of slot ``size_like`` or ``size_fn(input_hw_of_plan)``"""
        self.src, self.dst, self.size_like, self.size_fn = src, dst, size_like, size_fn

    def params(self):
        return []

    def inputs(self):
        return [self.src]


class ConcatUnit:
    kind = "cat"

    def __init__(self, a, b, dst):
        self.a, self.b, self.dst = a, b, dst

    def params(self):
        return []

    def inputs(self):
        return [self.a, self.b]


class Plan:
    def __init__(self, units, inputs, outputs, relu_inputs=()):
        self.units, self.inputs, self.outputs = units, list(inputs), list(outputs)
        self.relu_slots = set(relu_inputs)
        for u in units:
            if u.kind == "conv" and u.act == ACT_RELU:
                self.relu_slots.add(u.dst)
            elif u.kind == "pool" and u.src in self.relu_slots:
                self.relu_slots.add(u.dst)       # max of non-negative values; see PoolUnit backward
            elif u.kind == "cat" and u.a in self.relu_slots and u.b in self.relu_slots:
                self.relu_slots.add(u.dst)
        self.consumers = {}
        for u in units:
            for s in u.inputs():
                self.consumers[s] = self.consumers.get(s, 0) + 1
        self.param_list = []          # [(unit_index, role, tensor)]
        for ui, u in enumerate(units):
            for role, t in u.params():
                self.param_list.append((ui, role, t))

    def param_tensors(self):
        return [t for _, _, t in self.param_list]


def _pad_vec(v, n):
    if v is None or v.numel() == n:
        return v
    out = torch.zeros((n,), dtype=v.dtype, device=v.device)
    out[: v.numel()] = v
    return out


def _bn_uses_batch_stats(bn, bn_train):
    return bn is not None and bn_train and bn.training


def _packed_flags(plan, u, geom, dtype, need_bwd, batch_stats, bits, bn_train=False):
    """(forward operand packed?, data-gradient operand packed?) for the packed-operand kernels of csrc/conv_v2.hip.
    The data gradient only qualifies when its ReLU mask (if it applies one) exists as a bit tensor."""
    if not PACKED or u.grouped or u.act not in (ACT_NONE, ACT_RELU):
        return False, False
    if batch_stats:
        # Batch-statistics BN: the packed kernels have no statistics epilogue (16 more live registers spill under their cap), so
        # z is written first and one cs_bn_stats pass reads it back.  Only where that pass is noise next to the MFMA work: dense
        # stride-1 3x3 convolutions of >= 128 channels (the segmentation decoder, resnet.py:195-200 -- 86 % of segment-mode FLOPs).
        # Their ReLU masks, where no bit plane exists, are made on demand in backward (kernels.positive_bits).
        # (Round 4 tried the forward-only 1x1 convolutions of the frozen encoder too -- 21 us launches at 0.08 of HBM on the
        # first-generation kernel at B = 8: with the ring kernel + the statistics pass C5 went 1454 -> 1407 img/s, 512 x 512 580 -> 562.)
        if not (PACKED_TRAIN_BN and geom.R == 3 and geom.S == 3 and geom.stride == 1 and min(geom.C, geom.K) >= 128):
            return False, False
        pkf = K.packed_supported(geom, dtype, dgrad=False)
        pkb = bool(need_bwd) and K.packed_supported(geom, dtype, dgrad=True) and geom.C % 32 == 0
        return pkf, pkb
    pkf = K.packed_supported(geom, dtype, dgrad=False)
    pkb = bool(need_bwd) and K.packed_supported(geom, dtype, dgrad=True) and (u.src not in plan.relu_slots or u.src in bits)
    if pkb and geom.stride != 1:
        # a strided 1x1 data gradient is served in COMPACT form (kernels.CompactGrad): only where the other consumer of the block
        # input is the stride-1 1x1 convolution whose packed data gradient runs AFTER this one in backward order and adds it
        pkb = _compact_partner(plan, u, geom, dtype, bits, bn_train) is not None
    return pkf, pkb


def _compact_partner(plan, u, geom, dtype, bits, bn_train=False):
    """The unit whose data gradient will take u's compact gradient as its strided add operand, or None.  The partner must itself
    be staged bwd_packed in this pass: same eligibility as _packed_flags applies to it, including ITS BatchNorm mode (a model
    with mixed frozen / train-mode BN layers falls back to the dense strided gradient instead of raising in backward)."""
    others = [(i, v) for i, v in enumerate(plan.units) if v is not u and v.kind == "conv" and v.src == u.src]
    if len(others) != 1 or plan.consumers.get(u.src, 0) != 2:
        return None
    iv, v = others[0]
    iu = plan.units.index(u)
    c = v.conv
    if iv > iu or v.grouped or v.act not in (ACT_NONE, ACT_RELU) or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0):
        return None
    if not PACKED or _bn_uses_batch_stats(v.bn, bn_train):
        return None
    if u.src in plan.relu_slots and u.src not in bits:
        return None
    gv = K.make_geom(geom.N, geom.H, geom.W, geom.C, K.pad_channels(c.out_channels), 1, 1, 1, 0)
    if not K.packed_supported(gv, dtype, dgrad=True):
        return None
    return v


_STAGE_EPOCH = [0]


def invalidate_staged():
    """Drop every cached staged-weight set (of every plan).  Needed whenever parameters or BN buffers change behind the
    host's back: a HIP-graph replay of a training step updates them without running any Python (graphed.GraphedStep)."""
    _STAGE_EPOCH[0] += 1


def _stage_weights(u, dtype, Cp, Kp, need_bwd, folded, training, geom=None, pkf=False, pkb=False):
    """BN folding + weight staging.  Cached only for frozen parameters and for no-grad passes: a forward that will
    be followed by an optimizer step (`training` and the parameter requires grad) always stages afresh and leaves the
    cache invalid, because tensor version counters cannot be trusted to see the update -- torch's fused optimizers
    (`Adam(fused=True)`) write the parameters without bumping `_version`."""
    conv, bn = u.conv, u.bn
    # (a batch-statistics pass folds nothing of the BatchNorm into the operands: they depend on the convolution's own tensors only, so a
    # frozen encoder under train-mode BN -- the segmentation stage -- stages once, not once per step)
    key_t = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if (bn is not None and folded) else [])
    key = (_STAGE_EPOCH[0], dtype, Cp, Kp, need_bwd, folded, pkf, pkb) + tuple((t._version, t.data_ptr()) if t is not None else None for t in key_t)
    if training and any(t is not None and t.requires_grad for t in key_t):
        key = None
    elif u._cache is not None and u._cache[0] == key:
        return u._cache[1]
    w = conv.weight.detach()
    bias = conv.bias.detach() if conv.bias is not None else None
    scale = shift = rstd = None
    if folded and bn is not None and not u.grouped:
        w_khwc, w_chwk, scale, shift, rstd = K.stage_conv_bn(w, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                             bn.eps, bias, dtype, Cp, Kp, want_bwd=need_bwd)
        staged = _packed(SimpleNamespace(w_khwc=w_khwc, w_chwk=w_chwk, scale=scale, shift=shift, rstd=rstd), geom, pkf, pkb)
        u._cache = (key, staged)
        return staged
    if folded and bn is not None:
        scale, shift, rstd = K.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bias)
    else:
        shift = bias
    if u.grouped:
        w_khwc, w_chwk = K.weight_prep_grouped(w, scale, dtype, want_fwd=True, want_bwd=need_bwd)
    else:
        w_khwc, w_chwk = K.weight_prep(w, scale, dtype, Cp, Kp, want_fwd=True, want_bwd=need_bwd)
    staged = _packed(SimpleNamespace(w_khwc=w_khwc, w_chwk=w_chwk, scale=scale, shift=_pad_vec(shift, Kp), rstd=rstd), geom, pkf, pkb)
    u._cache = (key, staged)
    return staged


def _packed(st, geom, pkf, pkb):
    """Layer-by-layer staging path: re-order the plain operands for the packed-operand kernels (the one-launch StagePack
    writes that order directly)."""
    st.fwd_packed, st.bwd_packed = bool(pkf), bool(pkb and st.w_chwk is not None)
    if st.fwd_packed:
        st.w_khwc = K.pack_conv_weights(geom, st.w_khwc, dgrad=False)
    if st.bwd_packed:
        st.w_chwk = K.pack_conv_weights(geom, st.w_chwk, dgrad=True)
    return st


def _stem_takes_nchw(plan, dtype):
    """True when the plan's input feeds exactly one unit and that unit is the pixel-paired bf16 stem: the fp32 NCHW image can go
    straight to the paired operand (cs_stem_pair_from_nchw) and the NHWC8 tensor is never made."""
    if not (STEM_PAIRED and dtype == torch.bfloat16 and len(plan.inputs) == 1 and plan.units):
        return False
    u = plan.units[0]
    if u.kind != "conv" or u.src != plan.inputs[0] or plan.consumers.get(u.src, 0) != 1 or u.grouped or u.res is not None:
        return False
    c = u.conv
    return c.in_channels == 3 and c.kernel_size == (7, 7) and c.stride == (2, 2) and c.padding == (3, 3)


def forward(plan, feeds, dtype, bn_train, save, requires, image_hw=None, input_nchw=None):
    """feeds: {slot: NHWC tensor}.  Returns state with .t (slot tensors) and .aux (per unit).
    input_nchw: the network input as the contiguous fp32 [N,3,H,W] image instead (feeds then holds an UNINITIALISED placeholder of
    the NHWC8 shape for that slot): converted here -- directly into the paired stem operand where the stem takes it."""
    t = dict(feeds)
    nchw_for_stem = None
    if input_nchw is not None:
        if _stem_takes_nchw(plan, dtype):
            nchw_for_stem = input_nchw
        else:
            t[plan.inputs[0]] = K.to_nhwc(input_nchw, dtype, K.pad_channels(3))
    aux = [None] * len(plan.units)
    remaining = dict(plan.consumers)
    in_hw = image_hw
    if in_hw is None:
        in_hw = tuple(t[plan.inputs[0]].shape[1:3])

    def release(s):
        if save or s in plan.outputs:
            return
        remaining[s] -= 1
        if remaining[s] == 0:
            t.pop(s, None)

    # Training passes re-stage every folded Conv+BN after each optimizer update: one launch for the whole plan (the first such
    # pass stages layer by layer and records the layout; see kernels.StagePack)
    # (one pack per (dtype, input shape): which layers take packed operands depends on the geometry, and a ragged last batch must not
    # evict the pack of the full batches -- a captured step keeps replaying into it)
    packs = plan.__dict__.setdefault("_stage_packs", {})
    pack_key = (dtype, tuple(t[plan.inputs[0]].shape))
    pack = packs.get(pack_key) if save else None
    prestaged, record = {}, None
    capturing = _capture.capturing()
    if save:
        if pack is not None and pack.valid():
            pack.launch()
            prestaged = pack.by_unit
            if capturing:
                _capture.keep(pack)
        else:
            packs.pop(pack_key, None)
            while len(packs) >= 4:
                packs.pop(next(iter(packs)))
            record = []

    # train-mode BN bookkeeping for the whole plan in two launches instead of two per layer: one zero-filled fp64 arena for the
    # per-channel statistics, one foreach-add for the num_batches_tracked counters (this path is host-bound: ~1000 launches/step)
    stats_arena, stats_pos, bumped = None, [0], []
    if bn_train and save:
        K.begin_pass(t[plan.inputs[0]].device)
    bits = {}            # slot -> uint8 "output > 0" bit tensor (folded Conv+BN+ReLU outputs of a pass that will run backward)
    if bn_train:
        need = sum(K.accum_words(K.pad_channels(u_.conv.out_channels)) for u_ in plan.units
                   if u_.kind == "conv" and _bn_uses_batch_stats(u_.bn, bn_train))
        if need:
            stats_arena = torch.zeros((need,), dtype=torch.float64, device=t[plan.inputs[0]].device)

    def take_stats(C):
        n = K.accum_words(C)
        if stats_arena is None or stats_pos[0] + n > stats_arena.numel():
            return K.new_stats(C, t[plan.inputs[0]].device)
        v = stats_arena[stats_pos[0]:stats_pos[0] + n]
        stats_pos[0] += n
        return v

    # depthwise filters: the kernels read [R, S, C]; all layers of the plan restaged from the parameters by one launch per pass
    # (a permute + strided copy per layer: 26 launches of 4.4 us on EfficientNet-B3)
    dw_hwc = {}
    dw_units = [ui_ for ui_, u_ in enumerate(plan.units) if u_.kind == "dw"]
    if dw_units:
        ws_ = [plan.units[ui_].conv.weight.detach() for ui_ in dw_units]
        if all(w_.is_contiguous() and w_.dtype == torch.float32 and w_.dim() == 4 and w_.shape[1] == 1 for w_ in ws_):
            dpack = plan.__dict__.get("_dw_stage_pack")
            if dpack is None or dpack.key != tuple(w_.data_ptr() for w_ in ws_):
                dpack = plan.__dict__["_dw_stage_pack"] = K.DwStagePack(ws_)
            dpack.run()
            if capturing:
                _capture.keep(dpack)
            dw_hwc = dict(zip(dw_units, dpack.hwc))

    for ui, u in enumerate(plan.units):
        if u.kind == "conv":
            x = t[u.src]
            N, H, W, Cp = x.shape
            conv = u.conv
            Kc = conv.out_channels
            Kp = K.pad_channels(Kc)
            R, S = conv.kernel_size
            geom = K.make_geom(N, H, W, Cp, Kp, R, S, conv.stride[0], conv.padding[0])
            res = t[u.res] if u.res is not None else None
            need_bwd = save and requires.get(u.src, False)
            batch_stats = _bn_uses_batch_stats(u.bn, bn_train)
            pre = prestaged.get(ui)
            pkf, pkb = _packed_flags(plan, u, geom, dtype, need_bwd, batch_stats, bits, bn_train)
            # A pass that will be followed by an optimizer step, or that updates BN running statistics in place, must not leave
            # a version-keyed cache behind: fused optimizers and our own kernels write those tensors without bumping `_version`
            # (a later no-grad pass would be served the old weights; the one-launch pack path below never touches `_cache`)
            if save and any(p_ is not None and p_.requires_grad
                            for p_ in (conv.weight, conv.bias) + ((u.bn.weight, u.bn.bias) if (u.bn is not None and not batch_stats) else ())):
                u._cache = None
            if pre is not None and pre[0] == (Cp, Kp, need_bwd, pkf, pkb, batch_stats):
                st = pre[1]
            else:
                if pre is not None:
                    packs.pop(pack_key, None)       # the layout changed (other requires_grad pattern): rebuild next time
                st = _stage_weights(u, dtype, Cp, Kp, need_bwd, folded=not batch_stats, training=save, geom=geom, pkf=pkf, pkb=pkb)
                if capturing:
                    _capture.keep(st)               # (a cached staged set may be replaced by a later eager pass: the graph keeps this one)
                # (a batch-statistics layer joins the one-launch staging with bn = None: its operands depend on the convolution only)
                if (record is not None and not u.grouped and (batch_stats or u.bn is not None)
                        and any(p_ is not None and p_.requires_grad
                                for p_ in (conv.weight, conv.bias) + (() if batch_stats else (u.bn.weight, u.bn.bias)))):
                    record.append((ui, conv, None if batch_stats else u.bn, Cp, Kp, need_bwd, pkf, pkb))
            # the stem runs on a pixel-paired image (kernels.stem_*): 28 instead of 49 K chunks
            stem = STEM_PAIRED and K.is_stem_geom(geom) and not u.grouped and res is None and conv.in_channels <= 3
            if ui == 0 and nchw_for_stem is not None:
                if not stem:
                    raise RuntimeError("engine: the stem was expected to take the NCHW image (placeholder input would be read)")
                xp = K.stem_pair_from_nchw(nchw_for_stem, dtype)
            else:
                xp = K.stem_pair_input(x) if stem else None
            wp = K.stem_pair_weights(st.w_khwc) if stem else None
            if not batch_stats:
                if stem and PACKED and K.stem_fwd_packed_supported(geom, dtype):
                    y = K.stem_fwd_packed(geom, xp, K.stem_pack_weights(wp), st.shift, u.act)       # ring kernel, gathered rows
                elif stem:
                    y = K.stem_fwd(geom, xp, wp, None, st.shift, u.act)
                elif getattr(st, "fwd_packed", False):
                    if save and RELU_BITS and u.act == ACT_RELU:
                        y, bits[u.dst] = K.conv_fwd_packed(geom, x, st.w_khwc, st.shift, res, u.act, want_bits=True)
                    else:
                        y = K.conv_fwd_packed(geom, x, st.w_khwc, st.shift, res, u.act)
                elif save and RELU_BITS and u.act == ACT_RELU and Kp % 32 == 0:
                    # one bit per output next to the ReLU output: the data gradient that later masks with this tensor reads
                    # 1 byte per 16 (the bf16 masks are ~1/8 of a training step's HBM traffic)
                    y, bits[u.dst] = K.conv_fwd(geom, x, st.w_khwc, None, st.shift, res, u.act, grouped=u.grouped, want_bits=True)
                else:
                    y = K.conv_fwd(geom, x, st.w_khwc, None, st.shift, res, u.act, grouped=u.grouped)
                aux[ui] = SimpleNamespace(geom=geom, st=st, train=False, xp=xp)
            else:
                bn = u.bn
                stats = take_stats(Kp)
                if stem:
                    z = K.stem_fwd(geom, xp, wp, None, st.shift, ACT_NONE, stats=stats)
                elif getattr(st, "fwd_packed", False):
                    # halo kernel + one statistics pass over the stored z (the statistics of exactly the values bn_apply normalises)
                    z = K.conv_fwd_packed(geom, x, st.w_khwc, st.shift, None, ACT_NONE)
                    K.bn_stats(z, stats)
                else:
                    z = K.conv_fwd(geom, x, st.w_khwc, None, st.shift, None, ACT_NONE, stats=stats, grouped=u.grouped)
                M = N * geom.P * geom.Q
                momentum = bn.momentum if bn.momentum is not None else 0.1
                # (finalize + apply in one launch: mean / rstd are derived from the sums inside the apply pass)
                y, mean, rstd = K.bn_apply_stats(z, stats, bn.eps, momentum, bn.running_mean if bn.track_running_stats else None,
                                                 bn.running_var if bn.track_running_stats else None, bn.weight.detach(), bn.bias.detach(),
                                                 res, u.act)
                if bn.track_running_stats and bn.num_batches_tracked is not None:
                    bumped.append(bn.num_batches_tracked)
                aux[ui] = SimpleNamespace(geom=geom, st=st, train=True, z=z if save else None, mean=mean, rstd=rstd, xp=xp)
            t[u.dst] = y
        elif u.kind == "dw":
            x = t[u.src]
            N, H, W, C = x.shape
            conv, bn = u.conv, u.bn
            R = conv.kernel_size[0]
            geom = K.make_geom(N, H, W, C, C, R, R, conv.stride[0], conv.padding[0])
            w_hwc = dw_hwc[ui] if ui in dw_hwc else conv.weight.detach()[:, 0].permute(1, 2, 0).contiguous()
            if _bn_uses_batch_stats(bn, bn_train):
                z, zstats = K.dwconv_fwd_stats(geom, x, w_hwc)
                M = N * geom.P * geom.Q
                momentum = bn.momentum if bn.momentum is not None else 0.1
                t[u.dst], mean, rstd = K.bn_apply_stats(z, zstats, bn.eps, momentum, bn.running_mean if bn.track_running_stats else None,
                                                        bn.running_var if bn.track_running_stats else None, bn.weight.detach(),
                                                        bn.bias.detach(), None, u.act)
                if bn.track_running_stats and bn.num_batches_tracked is not None:
                    bumped.append(bn.num_batches_tracked)
                aux[ui] = SimpleNamespace(geom=geom, train=True, z=z if save else None, mean=mean, rstd=rstd, w_hwc=w_hwc)
            else:
                scale, shift, _ = K.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
                t[u.dst] = K.dwconv_fwd(geom, x, w_hwc, scale, shift, u.act)
                aux[ui] = SimpleNamespace(geom=geom, train=False)
        elif u.kind == "se":
            x = t[u.src]
            C = x.shape[-1]
            w1 = u.fc1.weight.detach().view(u.fc1.out_channels, C)
            w2 = u.fc2.weight.detach().view(C, u.fc2.in_channels)
            avg, _ = K.gap_fwd(x, with_max=False)
            h1, u1 = K.linear_fwd(avg, w1, u.fc1.bias.detach(), K.CS_ACT_SILU, want_preact=True)
            sc = K.linear_fwd(h1, w2, u.fc2.bias.detach(), K.CS_ACT_SIGMOID)
            t[u.dst] = K.se_scale(x, sc)
            aux[ui] = SimpleNamespace(avg=avg, h1=h1, u1=u1, s=sc, w1=w1, w2=w2)
        elif u.kind == "sd":
            a = t[u.a]
            keep = 1.0 - u.p
            noise = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device).bernoulli_(keep)
            if keep > 0:
                noise.div_(keep)
            t[u.dst] = K.rowscale_add(a, noise, t[u.b])
            aux[ui] = SimpleNamespace(noise=noise)
        elif u.kind == "pool":
            y, am = K.maxpool_fwd(t[u.src], want_argmax=save)
            aux[ui] = SimpleNamespace(argmax=am, in_hw=tuple(t[u.src].shape[1:3]))
            t[u.dst] = y
        elif u.kind == "up":
            x = t[u.src]
            size = tuple(t[u.size_like].shape[1:3]) if u.size_like is not None else tuple(u.size_fn(in_hw))
            t[u.dst] = K.bilinear_fwd(x, size)
            aux[ui] = SimpleNamespace(in_hw=tuple(x.shape[1:3]))
        elif u.kind == "cat":
            a, b = t[u.a], t[u.b]
            t[u.dst] = K.concat(a, b)
            aux[ui] = SimpleNamespace(ca=a.shape[-1])
        for s in u.inputs():
            release(s)
    if bumped:
        torch._foreach_add_(bumped, 1)
    if record:
        pk = K.StagePack([r_[1:] for r_ in record], dtype)
        pk.by_unit = {}
        for (ui_, _, bn_, cp_, kp_, nb_, pkf_, pkb_), (w_khwc, w_chwk, scale, shift, rstd) in zip(record, pk.staged):
            if bn_ is None:                          # batch statistics: the consumers expect no folded scale (and no rstd)
                scale = rstd = None
            pk.by_unit[ui_] = ((cp_, kp_, nb_, pkf_, pkb_, bn_ is None),
                               SimpleNamespace(w_khwc=w_khwc, w_chwk=w_chwk, scale=scale, shift=shift, rstd=rstd,
                                               fwd_packed=bool(pkf_), bwd_packed=bool(pkb_ and nb_)))
        packs[pack_key] = pk
    return SimpleNamespace(t=t, aux=aux, in_hw=in_hw, bits=bits)


def compute_requires(plan, param_needs, input_needs):
    """requires[slot]: does anything upstream of (and including the producer of) slot need a gradient?"""
    requires = {s: bool(input_needs.get(s, False)) for s in plan.inputs}
    pi = 0
    unit_trainable = []
    for u in plan.units:
        n = len(u.params())
        unit_trainable.append(any(param_needs[pi:pi + n]))
        pi += n
    for ui, u in enumerate(plan.units):
        requires[u.dst] = unit_trainable[ui] or any(requires.get(s, False) for s in u.inputs())
    return requires, unit_trainable


def _geom_key(geom):
    return (geom.N, geom.H, geom.W, geom.C, geom.K, geom.R, geom.S, geom.stride, geom.pad)


def _defers(u, a, need, ui):
    """Ungrouped convolutions under a BatchNorm take the batched weight-gradient path (one split-K launch + ONE fused finalize per group of
    identical geometry; the fused finalize turns the [rs][c] slabs into torch's [c][rs] order through LDS): eval-BN layers without a
    bias whose weight AND BN gradients are wanted (the finalize also writes dgamma / dbeta), and batch-statistics layers whose weight
    gradient is wanted (their BN gradients come from the BN backward pass; the finalize only folds the slabs) -- round 5: also the
    biased 3x3 convolutions of the decoder (resnet.py:195-200), whose bias gradient is the column sum of dz, taken on its own."""
    if not (u.kind == "conv" and need(ui, "weight") and not u.grouped and u.bn is not None):
        return False
    if a.train:
        return True
    return u.conv.bias is None and not need(ui, "bias") and bool(need(ui, "gamma") or need(ui, "beta"))


def _group_key(geom, a):
    return (_geom_key(geom), bool(a.train))


PACKED_TRAIN_BN = os.environ.get("CELLSEG_PACKED_TRAIN_BN", "1") != "0"   # ... also for heavy 3x3 convolutions under batch-statistics BN
PACKED = os.environ.get("CELLSEG_PACKED", "1") != "0"                # packed-operand conv kernels (0: first-generation igemm everywhere)
RELU_BITS = os.environ.get("CELLSEG_RELU_BITS", "1") != "0"          # bit masks next to ReLU outputs (0: bf16 tensors as masks)
STEM_PAIRED = os.environ.get("CELLSEG_STEM_PAIRED", "1") != "0"      # pixel-paired stem (0: the generic 7x7 path, for A/B runs)
_grad_sink = None
# batched weight gradients on a second HIP stream, concurrently with the dgrad chain: measured 2 % SLOWER on the ResNet-50 tile step
# (6395 vs 6522 tiles/s: the groups complete late and then compete with the HBM-bound dgrads of the next stage), so opt-in only
WGRAD_SIDE_STREAM = os.environ.get("CELLSEG_WGRAD_STREAM", "0") == "1"
# ... or only the groups of the deep stages (destination pixels <= this many: their weight gradients AND the data gradients they would run
# beside are launches of <= 256 workgroups that leave most of the chip idle); 0 = off.  Round 5 A/B inside the captured step.
WGRAD_SIDE_MAX_M = int(os.environ.get("CELLSEG_WGRAD_STREAM_MAX_M", "0") or 0)
_side_streams = {}


def _side_stream(dev):
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def set_grad_sink(sink):
    """Data-parallel hook (parallel.GradReducer.attach): `sink.view_for(param)` may hand out the tensor a parameter
    gradient is to be written into (a slice of a flat all-reduce bucket) and `sink.deliver(param, grad)` is called, in
    the order gradients are finished inside backward, as soon as the kernels producing `grad` are enqueued -- so the
    bucket's collective can start while the rest of backward still runs.  Returns the previous sink."""
    global _grad_sink
    prev, _grad_sink = _grad_sink, sink
    return prev


def backward(plan, state, grad_feeds, param_needs, requires, use_tr_read=True):
    """grad_feeds: {output slot: grad (already ReLU-masked where the slot is post-ReLU)}.
    Returns ({input slot: grad}, [param grads in plan.param_list order])."""
    t, aux = state.t, state.aux
    grads = dict(grad_feeds)
    gsum_cache = {}
    sink = _grad_sink
    # One zero-filled fp32 arena for every raw weight-gradient buffer and column-sum vector of this backward pass
    # (a single fill kernel instead of ~2 per convolution).
    need_w_units, arena_elems = set(), 0
    pi_ = 0
    for ui_, u_ in enumerate(plan.units):
        n_ = len(u_.params())
        if u_.kind == "conv" and any(param_needs[pi_:pi_ + n_]):
            need_w_units.add(ui_)
        if u_.kind == "conv" and aux[ui_] is not None:
            arena_elems += aux[ui_].geom.C + aux[ui_].geom.K + 16
        pi_ += n_
    dev_ = next(iter(grad_feeds.values())).device if grad_feeds else None
    arena = torch.zeros((arena_elems,), dtype=torch.float32, device=dev_) if (arena_elems and dev_ is not None) else None
    arena_pos = [0]
    main = torch.cuda.current_stream() if (dev_ is not None and dev_.type == "cuda") else None
    side = _side_stream(dev_) if (main is not None and (WGRAD_SIDE_STREAM or WGRAD_SIDE_MAX_M > 0)) else None
    if side is not None and arena is not None:
        arena.record_stream(side)

    def take(shape):
        n = 1
        for d in shape:
            n *= d
        n8 = (n + 7) // 8 * 8
        if arena is None or arena_pos[0] + n8 > arena.numel():
            return torch.zeros(shape, dtype=torch.float32, device=dev_)
        v = arena[arena_pos[0]:arena_pos[0] + n].view(shape)
        arena_pos[0] += n8
        return v
    left = dict(plan.consumers)
    pgrads = [None] * len(plan.param_list)
    pindex = {}
    for i, (ui, role, _) in enumerate(plan.param_list):
        pindex[(ui, role)] = i

    def need(ui, role):
        i = pindex.get((ui, role))
        return i is not None and param_needs[i]

    def emit(ui, role, g):
        """Record a finished parameter gradient (and hand it to the data-parallel sink straight away)."""
        i = pindex[(ui, role)]
        pgrads[i] = g if sink is None else sink.deliver(plan.param_list[i][2], g)

    def grad_buffer(param, shape=None):
        """Destination for a parameter gradient: the sink's bucket slice when there is one."""
        v = sink.view_for(param) if sink is not None else None
        if v is not None:
            return v
        return torch.empty_like(param) if shape is None else torch.empty(shape, dtype=torch.float32, device=param.device)

    # how many layers of each geometry will ask for a batched weight gradient: a group is launched as soon as it is complete
    # (or holds 8 layers), so its gradients are final -- and its activations released -- long before backward ends
    group_total = {}
    for ui_, u_ in enumerate(plan.units):
        if ui_ in need_w_units and aux[ui_] is not None and _defers(u_, aux[ui_], need, ui_):
            k_ = _group_key(aux[ui_].geom, aux[ui_])
            group_total[k_] = group_total.get(k_, 0) + 1
    group_seen = {}

    def flush(items):
        # Weight gradients are off the critical path (the dgrad chain): optionally on a side stream (WGRAD_SIDE_STREAM)
        if side is None:
            return flush_on_current(items)
        if not WGRAD_SIDE_STREAM:
            g_ = items[0].a.geom
            if g_.N * g_.P * g_.Q > WGRAD_SIDE_MAX_M:
                return flush_on_current(items)
        for it in items:
            for t_ in (it.x, it.dz, it.a.xp, it.a.st.scale, it.a.st.rstd, getattr(it.gsum, "buf", it.gsum)):
                if t_ is not None:
                    t_.record_stream(side)           # allocated on the main stream, read by side-stream kernels
        side.wait_stream(main)
        with torch.cuda.stream(side):
            flush_on_current(items)

    def flush_on_current(items):
        geom = items[0].a.geom
        n = len(items)
        convs = [it.u.conv for it in items]
        Cin = convs[0].in_channels
        train = bool(items[0].a.train)                   # (a group is all eval-BN or all batch-statistics: _group_key)
        dws = [grad_buffer(c.weight) for c in convs]
        dgs = None if train else [grad_buffer(it.u.bn.weight) for it in items]
        dbs = None if train else [grad_buffer(it.u.bn.bias) for it in items]
        # (single layers take the same path: its finalize folds deferred column sums, the stand-alone one does not)
        if n == 1 and items[0].a.xp is not None:
            slabs = K.stem_wgrad(geom, items[0].a.xp, items[0].dz, use_tr_read=use_tr_read).unsqueeze(0)     # [1, 1, K, 7, 7, 8]
        else:
            slabs = K.wgrad_batched(geom, [it.x for it in items], [it.dz for it in items], use_tr_read=use_tr_read)
        if train:
            K.wgrad_finalize_batched(slabs, None, None, None, None, None, dws, None, None, Cin)
            for it, dw in zip(items, dws):
                emit(it.ui, "weight", dw)
            return
        K.wgrad_finalize_batched(slabs, [c.weight.detach() for c in convs], [it.a.st.scale for it in items],
                                 [it.a.st.rstd for it in items], [it.u.bn.running_mean for it in items], [it.gsum for it in items],
                                 dws, dgs, dbs, Cin)
        for it, dw, dg, db in zip(items, dws, dgs, dbs):
            emit(it.ui, "weight", dw)
            if need(it.ui, "gamma"):
                emit(it.ui, "gamma", dg)
            if need(it.ui, "beta"):
                emit(it.ui, "beta", db)

    def contribute(slot, g, masked):
        """Non-fused contribution (alias when first)."""
        left[slot] -= 1
        if slot in grads:
            raise NotImplementedError("engine: unfused gradient accumulation is not expected for these networks")
        if left[slot] == 0 and slot in plan.relu_slots and not masked:
            raise NotImplementedError("engine: last contribution to a post-ReLU slot must come from a masking kernel")
        grads[slot] = g

    deferred = {}
    for ui in reversed(range(len(plan.units))):
        u = plan.units[ui]
        g = grads.pop(u.dst, None)
        a = aux[ui]
        if g is None:
            continue
        if u.kind == "conv":
            conv = u.conv
            Kc, Cin = conv.out_channels, conv.in_channels
            geom = a.geom
            x = t[u.src]
            if u.res is not None and requires.get(u.res, False):
                contribute(u.res, g, masked=False)
            want_w = need(ui, "weight")
            want_b = need(ui, "bias")
            want_bn = need(ui, "gamma") or need(ui, "beta")
            dz = g
            if u.act == ACT_SILU and not a.train:
                raise NotImplementedError(f"{u.name}: backward through a folded (eval-mode) BN + SiLU is not supported; "
                                          "EfficientNet trains with batch statistics (efficientnet.py:308-312)")
            if a.train:
                bn = u.bn
                dz, dgamma, dbeta = K.bn_bwd(g, a.z, a.mean, a.rstd, bn.weight.detach(), want_param_grads=want_bn,
                                             beta=bn.bias.detach(), act=ACT_SILU if u.act == ACT_SILU else ACT_NONE)
                if need(ui, "gamma"):
                    emit(ui, "gamma", dgamma[:Kc])
                if need(ui, "beta"):
                    emit(ui, "beta", dbeta[:Kc])
            if _defers(u, a, need, ui):
                # weight gradients of identical-geometry layers are launched together (one batched split-K launch per shape
                # group: proportionally fewer partial slabs to write and fold)
                gsum = None
                if not a.train:
                    gsum = gsum_cache.pop(u.dst, None)
                    if gsum is None:
                        gsum = K.colsum_partial(dz)
                    if u.res is not None and grads.get(u.res) is g:
                        gsum_cache[u.res] = gsum
                elif want_b:
                    # the bias of a convolution in front of a batch-statistics BN: d bias = column sums of dz (analytically zero --
                    # the BN backward removes the batch mean --, the reference computes the same rounding noise: train_seg.py decoder)
                    dbias = grad_buffer(conv.bias)
                    dbias.copy_(K.colsum(dz)[:Kc])
                    emit(ui, "bias", dbias)
                key = _group_key(geom, a)
                items = deferred.setdefault(key, [])
                items.append(SimpleNamespace(ui=ui, u=u, a=a, x=x, dz=dz, gsum=gsum))
                group_seen[key] = group_seen.get(key, 0) + 1
                if len(items) == 8 or group_seen[key] == group_total.get(key, 0):      # kernel-argument tables hold at most 8 layers
                    flush(deferred.pop(key))
            elif want_w or want_b or (want_bn and not a.train):
                gsum = None
                if want_b or (want_bn and not a.train):
                    gsum = K.colsum_vector(gsum_cache.pop(u.dst, None)) if not a.train else None
                    if gsum is None:
                        gsum = K.colsum(dz)
                    if not a.train and u.res is not None and grads.get(u.res) is g:
                        gsum_cache[u.res] = gsum      # the residual branch receives the very same gradient tensor
                if a.xp is not None:
                    raw = K.stem_wgrad(geom, a.xp, dz, use_tr_read=use_tr_read)
                elif not u.grouped and K.wgrad2_serves(geom, x.dtype):
                    raw = K.wgrad_batched(geom, [x], [dz], use_tr_read=use_tr_read)[0]      # wgrad_v2.hip (the batched entry routes to it)
                else:
                    raw = K.new_wgrad_buffer(geom, x.device, u.grouped)
                    K.conv_wgrad(geom, x, dz, raw, use_tr_read=use_tr_read, grouped=u.grouped)
                dw = grad_buffer(conv.weight)
                dbias = grad_buffer(conv.bias) if want_b else None
                dgamma = dbeta = None
                if want_bn and not a.train:
                    dgamma, dbeta = grad_buffer(u.bn.weight), grad_buffer(u.bn.bias)
                dot = take((Kc,)) if dgamma is not None else None
                if u.grouped:
                    K.wgrad_finalize_grouped(raw, conv.weight.detach() if dgamma is not None else None, None if a.train else a.st.scale,
                                             None if a.train else a.st.rstd, u.bn.running_mean if dgamma is not None else None, gsum, dw,
                                             dgamma=dgamma, dbeta=dbeta, dot=dot)
                else:
                    K.wgrad_finalize(raw, conv.weight.detach() if dgamma is not None else None, None if a.train else a.st.scale,
                                     None if a.train else a.st.rstd, u.bn.running_mean if dgamma is not None else None, gsum, Cin, dw,
                                     dbias=dbias, dgamma=dgamma, dbeta=dbeta, dot=dot)
                if want_w:
                    emit(ui, "weight", dw)
                if want_b:
                    emit(ui, "bias", dbias)
                if dgamma is not None:
                    if need(ui, "gamma"):
                        emit(ui, "gamma", dgamma)
                    if need(ui, "beta"):
                        emit(ui, "beta", dbeta)
            gsum_cache.pop(u.dst, None)
            if requires.get(u.src, False):
                left[u.src] -= 1
                final = left[u.src] == 0
                pending = grads.pop(u.src, None)
                mask = x if (u.src in plan.relu_slots and final) else None
                mbits = state.bits.get(u.src) if mask is not None else None
                if mbits is not None:
                    mask = None
                if getattr(a.st, "bwd_packed", False) and geom.stride != 1:
                    if final or pending is not None:
                        raise RuntimeError(f"{u.name}: compact strided data gradient out of order (it must be the first contribution)")
                    dx = K.conv_dgrad_packed(geom, dz, a.st.w_chwk)          # kernels.CompactGrad: consumed as a strided add operand
                elif getattr(a.st, "bwd_packed", False):
                    if mask is not None and a.train:
                        mbits, mask = K.positive_bits(mask), None          # (a train-mode BN + ReLU output / a concatenation of such)
                    if mask is not None:
                        raise RuntimeError(f"{u.name}: packed data-gradient operand staged but the ReLU mask is not a bit tensor")
                    if final:
                        dx, gsum_cache[u.src] = K.conv_dgrad_packed(geom, dz, a.st.w_chwk, add=pending, mask_bits=mbits, want_colsum=True)
                    else:
                        dx = K.conv_dgrad_packed(geom, dz, a.st.w_chwk, add=pending, mask_bits=mbits)
                elif final:
                    # column sums of the finished gradient feed its producer's BN/bias gradients: left as per-workgroup partial
                    # rows where the launch allows it (the batched finalize folds them; one small launch less per layer)
                    if isinstance(pending, K.CompactGrad):
                        raise RuntimeError(f"{u.name}: a compact gradient reached a data gradient that cannot add it")
                    cs = take((geom.C,)) if (geom.stride != 1 or u.grouped) else None
                    dx, gsum_cache[u.src] = K.conv_dgrad(geom, dz, a.st.w_chwk, add=pending, mask=mask, colsum=cs, grouped=u.grouped,
                                                         defer_colsum=True, mask_bits=mbits)
                else:
                    if isinstance(pending, K.CompactGrad):
                        raise RuntimeError(f"{u.name}: a compact gradient reached a data gradient that cannot add it")
                    dx = K.conv_dgrad(geom, dz, a.st.w_chwk, add=pending, mask=mask, grouped=u.grouped, mask_bits=mbits)
                grads[u.src] = dx
        elif u.kind == "dw":
            if not a.train:
                raise NotImplementedError(f"{u.name}: backward through an eval-mode depthwise block is not supported")
            bn, x = u.bn, t[u.src]
            want_bn = need(ui, "gamma") or need(ui, "beta")
            dz, dgamma, dbeta = K.bn_bwd(g, a.z, a.mean, a.rstd, bn.weight.detach(), want_param_grads=want_bn, beta=bn.bias.detach(),
                                         act=ACT_SILU if u.act == ACT_SILU else ACT_NONE)
            if need(ui, "gamma"):
                emit(ui, "gamma", dgamma)
            if need(ui, "beta"):
                emit(ui, "beta", dbeta)
            if need(ui, "weight"):
                emit(ui, "weight", K.dwconv_wgrad(a.geom, x, dz, param_layout=True))
            if requires.get(u.src, False):
                left[u.src] -= 1
                dx = K.dwconv_dgrad(a.geom, dz, a.w_hwc)
                pending = grads.pop(u.src, None)
                grads[u.src] = dx if pending is None else K.rowscale_add(dx, None, pending)
        elif u.kind == "se":
            x = t[u.src]
            ds = K.se_scale_bwd_ds(g, x)
            want2 = need(ui, "w2") or need(ui, "b2")
            want1 = need(ui, "w1") or need(ui, "b1")
            dh1, dw2, db2 = K.linear_bwd(a.h1, a.w2, ds, a.s, K.CS_ACT_SIGMOID, True, want2, want2)
            davg, dw1, db1 = K.linear_bwd(a.avg, a.w1, dh1, a.u1, K.CS_ACT_SILU, True, want1, want1)
            if need(ui, "w2"):
                emit(ui, "w2", dw2.view_as(u.fc2.weight))
            if need(ui, "b2"):
                emit(ui, "b2", db2)
            if need(ui, "w1"):
                emit(ui, "w1", dw1.view_as(u.fc1.weight))
            if need(ui, "b1"):
                emit(ui, "b1", db1)
            if requires.get(u.src, False):
                left[u.src] -= 1
                if u.src in grads:
                    raise NotImplementedError("engine: SE input with several consumers")
                grads[u.src] = K.se_scale_bwd_dx(g, a.s, davg)
        elif u.kind == "sd":
            if requires.get(u.b, False):
                contribute(u.b, g, masked=True)
            if requires.get(u.a, False):
                contribute(u.a, K.rowscale_add(g, a.noise, None), masked=True)
        elif u.kind == "pool":
            if requires.get(u.src, False):
                left[u.src] -= 1
                if u.src in grads:
                    raise NotImplementedError("engine: pool input with several consumers")
                # dy is masked by [pool_out>0]; the argmax element equals pool_out, so dx is masked too
                grads[u.src] = K.maxpool_bwd(g, a.argmax, None, a.in_hw)
        elif u.kind == "up":
            if requires.get(u.src, False):
                left[u.src] -= 1
                if u.src in grads:
                    raise NotImplementedError("engine: upsample input with several consumers")
                mask = t[u.src] if u.src in plan.relu_slots else None
                grads[u.src] = K.bilinear_bwd(g, a.in_hw, mask=mask)
        elif u.kind == "cat":
            na, nb = requires.get(u.a, False), requires.get(u.b, False)
            if na or nb:
                ga, gb = K.split(g, a.ca, want_a=na, want_b=nb)
                if na:
                    contribute(u.a, ga, masked=True)
                if nb:
                    contribute(u.b, gb, masked=True)
    for items in list(deferred.values()):       # groups whose count fell short of the forecast (defensive; not expected)
        flush(items)
    if side is not None:
        main.wait_stream(side)                   # every parameter gradient is final before autograd hands it on
        for g_ in pgrads:
            if g_ is not None:
                g_.record_stream(main)
    return {s: grads.get(s) for s in plan.inputs}, pgrads


class _PlanFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, cfg, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        feeds = {s: x for s, x in zip(plan.inputs, inputs)}
        input_needs = {s: ctx.needs_input_grad[3 + i] for i, s in enumerate(plan.inputs)}
        param_needs = [ctx.needs_input_grad[3 + n_inputs + i] for i in range(len(params))]
        save = any(param_needs) or any(input_needs.values())
        requires, _ = compute_requires(plan, param_needs, input_needs)
        state = forward(plan, feeds, cfg.dtype, cfg.bn_train, save, requires, cfg.image_hw, getattr(cfg, "input_nchw", None))
        ctx.plan, ctx.cfg, ctx.n_inputs = plan, cfg, n_inputs
        ctx.param_needs, ctx.requires = param_needs, requires
        ctx.state = state if save else None
        outs = tuple(state.t[s] for s in plan.outputs)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        grad_feeds = {}
        for s, g in zip(plan.outputs, gouts):
            if g is not None:
                grad_feeds[s] = g.contiguous()
        in_grads, pgrads = backward(plan, ctx.state, grad_feeds, ctx.param_needs, ctx.requires, ctx.cfg.use_tr_read)
        ctx.state = None
        return (None, None, None) + tuple(in_grads[s] for s in plan.inputs) + tuple(pgrads)


def run_plan(plan, inputs, dtype, bn_train, use_tr_read=True, image_hw=None, input_nchw=None):
    """Differentiable execution of a plan. inputs: NHWC tensors for plan.inputs; returns NHWC outputs.
    image_hw: spatial size of the network input (for UpsampleUnit.size_fn).
    input_nchw: a contiguous fp32 [N,3,H,W] image that does NOT require a gradient, given INSTEAD of the single NHWC input (inputs is
    then ignored): the engine converts it itself, straight into the paired stem operand where the first unit is that stem."""
    cfg = SimpleNamespace(dtype=dtype, bn_train=bn_train, use_tr_read=use_tr_read, image_hw=image_hw, input_nchw=None)
    if input_nchw is not None:
        if len(plan.inputs) != 1 or input_nchw.requires_grad or input_nchw.dtype != torch.float32 or input_nchw.dim() != 4 or input_nchw.shape[1] != 3:
            raise ValueError("run_plan: input_nchw is a single fp32 [N,3,H,W] image without gradient")
        cfg.input_nchw = input_nchw.contiguous()
        N, _, H, W = input_nchw.shape
        inputs = [torch.empty((N, H, W, K.pad_channels(3)), dtype=dtype, device=input_nchw.device)]      # shape carrier only (never read, never filled)
    return _PlanFunction.apply(plan, cfg, len(inputs), *inputs, *plan.param_tensors())
