"""MIL ResNet-18/34/50 on the HIP engine.

Host-side mirror of the reference's ``MILResNet`` (model/resnet.py:81-333): same attribute names,
``setmode`` / ``forward(x, freeze_bn)`` / ``set_*_grads`` API, prefixes, error messages and
state_dict keys (torchvision ResNet names + fc_tile / fc_image_* / upconvK / seg_out_conv), so
reference checkpoints load and the reference drivers run unchanged.  The modules below only HOLD
parameters; all arithmetic runs through ``engine`` plans (hand-written HIP kernels).  There is no
CPU path: a non-GPU input raises.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine as E
from .. import functional as HF
from .. import kernels as K

__all__ = ["MILResNet", "MILresnet18", "MILresnet34", "MILresnet50", "MILresnext50_32x4d", "MILresnext101_32x8d"]

# name -> (block kind, blocks per stage)            model/resnet.py:336-361
_ARCH = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnext50_32x4d": ("bottleneck", (3, 4, 6, 3)),          # model/resnext.py:418-442
    "resnext101_32x8d": ("bottleneck", (3, 4, 23, 3)),
}
# per block kind: expansion and the (kernel, padding, carries-stride) of its convs, in order
_BLOCK = {
    "basic": (1, ((3, 1, True), (3, 1, False))),
    "bottleneck": (4, ((1, 0, False), (3, 1, True), (1, 0, False))),
}


def default_compute_dtype():
    name = os.environ.get("CELLSEG_COMPUTE_DTYPE", "bf16").lower()
    return torch.float32 if name in ("fp32", "f32", "float32") else torch.bfloat16


class _Block(nn.Module):
    """Parameter holder for one residual block (conv{i}/bn{i}[/downsample]); never called."""

    def __init__(self, kind, inplanes, planes, stride, groups=1, base_width=64):
        super().__init__()
        self.kind = kind
        exp, convs = _BLOCK[kind]
        width = planes if kind == "basic" else int(planes * (base_width / 64.0)) * groups
        chans = [inplanes] + ([planes, planes] if kind == "basic" else [width, width, planes * exp])
        for i, (k, pad, strided) in enumerate(convs, start=1):
            g = groups if (kind == "bottleneck" and i == 2) else 1
            setattr(self, f"conv{i}", nn.Conv2d(chans[i - 1], chans[i], k, stride if strided else 1, pad, groups=g, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(chans[i]))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != planes * exp:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * exp, 1, stride, bias=False), nn.BatchNorm2d(planes * exp))
        self.stride = stride
        self.n_convs = len(convs)

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("blocks are executed by the HIP engine, not called directly")


def _image_head(features, n_out, final_relu):
    layers = [nn.Flatten(), nn.BatchNorm1d(features), nn.Dropout(p=0.25), nn.ReLU(inplace=True), nn.Linear(features, 64),
              nn.BatchNorm1d(64), nn.Dropout(), nn.Linear(64, n_out)]
    if final_relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


def _upsample_conv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class MILResNet(nn.Module):
    def __init__(self, encoder, num_classes=1000, groups=1, width_per_group=64, block_table=None, decoder_expansion=None,
                 fan_out_init=False):
        super().__init__()
        kind, depths = (block_table or _ARCH)[encoder]
        exp = _BLOCK[kind][0]
        self.encoder_name = encoder
        self.mode = None
        self.encoder_prefix = ("conv1", "bn1", "relu", "layer1", "layer2", "layer3", "layer4")
        self.image_module_prefix = ("fc_image_cls", "fc_image_reg")
        self.tile_module_prefix = ("fc_tile",)
        self.seg_module_prefix = ("upconv", "seg_out_conv")
        self.compute_dtype = default_compute_dtype()
        self.use_tr_read = True
        self._kind, self._exp = kind, exp

        # encoder
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, depth in enumerate(depths, start=1):
            planes = 64 * 2 ** (li - 1)
            blocks = []
            for b in range(depth):
                blocks.append(_Block(kind, inplanes, planes, 2 if (b == 0 and li > 1) else 1, groups, width_per_group))
                inplanes = planes * exp
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        feat = 512 * exp
        self.feature_dim = feat

        # heads (model/resnet.py:121-152)
        self.avgpool_tile, self.maxpool_tile = nn.AdaptiveAvgPool2d((1, 1)), nn.AdaptiveMaxPool2d((1, 1))
        self.fc_tile = nn.Sequential(nn.Flatten(), nn.Linear(feat, num_classes))
        self.avgpool_image, self.maxpool_image = nn.AdaptiveAvgPool2d((1, 1)), nn.AdaptiveMaxPool2d((1, 1))
        self.fc_image_cls = _image_head(feat, 7, False)
        self.fc_image_reg = _image_head(feat, 1, True)

        # decoder (model/resnet.py:154-164)
        e = exp if decoder_expansion is None else decoder_expansion
        mid = 64 if e == 1 else 32 * e
        dec = [(512 * e, 256 * e), (512 * e, 256 * e), (256 * e, 128 * e), (256 * e, 128 * e), (128 * e, 64 * e), (128 * e, 64 * e),
               (64 * e, mid), (mid, 64)]
        for i, (ci, co) in enumerate(dec, start=1):
            setattr(self, f"upconv{i}", _upsample_conv(ci, co))
        self.seg_out_conv = nn.Conv2d(64, 2, 1)

        for m in self.modules():                      # model/resnet.py:170-177 ; model/resnext.py:223-230 (fan_out, bias untouched)
            if isinstance(m, nn.Conv2d):
                if fan_out_init:
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                    continue
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._plans = {}

    # ------------------------------------------------------------------ requires_grad groups
    def _set(self, names, flag):
        for n in names:
            getattr(self, n).requires_grad_(flag)

    def set_encoder_grads(self, requires_grad):
        self._set(("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"), requires_grad)

    def set_tile_module_grads(self, requires_grad):
        self._set(("avgpool_tile", "maxpool_tile", "fc_tile"), requires_grad)

    def set_image_module_grads(self, requires_grad):
        self._set(("avgpool_image", "maxpool_image", "fc_image_cls", "fc_image_reg"), requires_grad)

    def set_seg_module_grads(self, requires_grad):
        # the reference toggles upconv1-4 and seg_out_conv only (resnet.py:226-232): upconv5-8 stay as they are
        self._set(("upconv1", "upconv2", "upconv3", "upconv4", "seg_out_conv"), requires_grad)

    def setmode(self, mode):
        table = {"tile": (False, True, False, False), "image": (True, False, True, False), "segment": (False, False, False, True)}
        if mode not in table:
            raise Exception("Invalid mode: {}.".format(mode))
        enc, tile, image, seg = table[mode]
        self.set_encoder_grads(enc)
        self.set_tile_module_grads(tile)
        self.set_image_module_grads(image)
        self.set_seg_module_grads(seg)
        self.mode = mode

    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (throughput) or torch.float32 (parity mode, exact-f32 MFMA)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    # ------------------------------------------------------------------ plans
    def _encoder_plan(self, with_skips):
        key = ("enc", with_skips)
        if key in self._plans:
            return self._plans[key]
        units, slot = [], 0
        nxt = [1]

        def new():
            nxt[0] += 1
            return nxt[0] - 1

        s = new()
        units.append(E.ConvUnit("conv1", self.conv1, self.bn1, E.ACT_RELU, 0, s))
        p = new()
        units.append(E.PoolUnit(s, p))
        cur, skips = p, []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                x_in = cur
                name = f"layer{li}.{bi}"
                for ci in range(1, blk.n_convs):
                    d = new()
                    units.append(E.ConvUnit(f"{name}.conv{ci}", getattr(blk, f"conv{ci}"), getattr(blk, f"bn{ci}"), E.ACT_RELU, cur, d))
                    cur = d
                res = x_in
                if blk.downsample is not None:
                    res = new()
                    units.append(E.ConvUnit(f"{name}.downsample", blk.downsample[0], blk.downsample[1], E.ACT_NONE, x_in, res))
                out = new()
                n = blk.n_convs
                units.append(E.ConvUnit(f"{name}.conv{n}", getattr(blk, f"conv{n}"), getattr(blk, f"bn{n}"), E.ACT_RELU, cur, out, res=res))
                cur = out
            skips.append(cur)
        outputs = [skips[3], skips[2], skips[1], skips[0]] if with_skips else [skips[3]]
        plan = E.Plan(units, [0], outputs)
        self._plans[key] = plan
        return plan

    def _decoder_plan(self):
        if "dec" in self._plans:
            return self._plans["dec"]
        X4, X3, X2, X1 = 0, 1, 2, 3
        nxt = [4]

        def new():
            nxt[0] += 1
            return nxt[0] - 1

        units = []

        def upconv(i, src):
            d = new()
            seq = getattr(self, f"upconv{i}")
            units.append(E.ConvUnit(f"upconv{i}", seq[0], seq[1], E.ACT_RELU, src, d))
            return d

        def up(src, like=None, fn=None):
            d = new()
            units.append(E.UpsampleUnit(src, d, size_like=like, size_fn=fn))
            return d

        def cat(a, b):
            d = new()
            units.append(E.ConcatUnit(a, b, d))
            return d

        o = upconv(1, up(X4, like=X3))
        o = upconv(2, cat(o, X3))
        o = upconv(3, up(o, like=X2))
        o = upconv(4, cat(o, X2))
        o = upconv(5, up(o, like=X1))
        o = upconv(6, cat(o, X1))
        o = upconv(7, up(o, fn=lambda hw: ((hw[0] + 6 - 7) // 2 + 1, (hw[1] + 6 - 7) // 2 + 1)))   # conv1-output size
        o = upconv(8, o)
        o = up(o, fn=lambda hw: hw)
        d = new()
        units.append(E.ConvUnit("seg_out_conv", self.seg_out_conv, None, E.ACT_NONE, o, d))
        plan = E.Plan(units, [X4, X3, X2, X1], [d], relu_inputs=(X4, X3, X2, X1))
        self._plans["dec"] = plan
        return plan

    # ------------------------------------------------------------------ forward
    def _trunk(self, x, bn_train, with_skips):
        if not x.is_cuda:
            raise RuntimeError("cellsegmentation_amd models run on the GPU only (HIP kernels, no CPU fallback); "
                               "move the model and its input to a cuda device")
        if x.dim() == 4 and x.shape[-1] == 8 and x.shape[1] != 3 and x.dtype == self.compute_dtype:
            xh = x                      # already staged NHWC tiles (cellsegmentation_amd.tiles.gather_tiles)
        elif x.dim() == 4 and x.shape[1] == 3 and not x.requires_grad:
            # the image as it comes from the loader: the engine converts it (straight into the paired stem operand for bf16)
            return E.run_plan(self._encoder_plan(with_skips), [], self.compute_dtype, bn_train, self.use_tr_read, input_nchw=x.float())
        else:
            xh = HF.to_nhwc(x.float(), self.compute_dtype)
        return E.run_plan(self._encoder_plan(with_skips), [xh], self.compute_dtype, bn_train, self.use_tr_read)

    def _image_branch(self, seq, feat):
        final_relu = len(seq) == 9
        h = HF.batch_norm_rows(feat, seq[1], K.CS_ACT_RELU)          # BN -> Dropout -> ReLU == BN+ReLU -> Dropout
        h = F.dropout(h, seq[2].p, self.training)
        h = HF.linear(h, seq[4].weight, seq[4].bias)
        h = HF.batch_norm_rows(h, seq[5])
        h = F.dropout(h, seq[6].p, self.training)
        return HF.linear(h, seq[7].weight, seq[7].bias, K.CS_ACT_RELU if final_relu else K.CS_ACT_NONE)

    def forward(self, x, freeze_bn=False):
        if self.mode == "tile" and freeze_bn:
            (x4,) = self._trunk(x, False, False)
            # the reference's eval()/train() flip (resnet.py:256-258) leaves EVERY module in train mode.  nn.Module.train() walks the
            # ~200 modules with a __setattr__ each (0.8 ms of host time per step, profiles/round4_notes.md): only when one is not
            if not self._all_training():
                self.train()
        elif self.mode == "segment":
            x4, x3, x2, x1 = self._trunk(x, self.training, True)
        else:
            (x4,) = self._trunk(x, self.training, False)

        if self.mode == "tile":
            feat = HF.gap_avgmax(x4, self.feature_dim)
            lin = self.fc_tile[1]
            return HF.linear(feat, lin.weight, lin.bias)
        elif self.mode == "image":
            feat = HF.gap_avgmax(x4, self.feature_dim)
            return self._image_branch(self.fc_image_cls, feat), self._image_branch(self.fc_image_reg, feat)
        elif self.mode == "segment":
            cfg_hw = tuple(x.shape[-2:])
            (o,) = E.run_plan(self._decoder_plan(), [x4, x3, x2, x1], self.compute_dtype, self.training, self.use_tr_read, image_hw=cfg_hw)
            return HF.to_nchw(o, 2)
        else:
            raise Exception("Something wrong in setmode.")


def _all_training(self):
    mods = self.__dict__.get("_module_list")
    if mods is None or len(mods) != self.__dict__.get("_module_count", -1):
        mods = list(self.modules())
        self.__dict__["_module_list"] = mods
        self.__dict__["_module_count"] = len(mods)
    for m in mods:
        if not m.training:
            return False
    return True


MILResNet._all_training = _all_training


def _make(name, pretrained, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True needs a download; load ImageNet weights with load_state_dict(strict=False) instead")
    model = MILResNet(name, **kwargs)
    # 2-way tile classifier, as the reference factories do after loading weights (resnet.py:342,351,360)
    model.fc_tile[1] = nn.Linear(model.fc_tile[1].in_features, 2)
    return model


def MILresnet18(pretrained=False, **kwargs):
    return _make("resnet18", pretrained, **kwargs)


def MILresnet34(pretrained=False, **kwargs):
    return _make("resnet34", pretrained, **kwargs)


def MILresnet50(pretrained=False, **kwargs):
    return _make("resnet50", pretrained, **kwargs)


def MILresnext50_32x4d(pretrained=False, progress=True, **kwargs):
    """model/resnext.py:418-429.  The decoder of the reference's ResNeXt is built for expansion 1 (resnext.py:209-217), so
    segment mode cannot run there either; tile / image modes only."""
    return _make("resnext50_32x4d", pretrained, groups=32, width_per_group=4, decoder_expansion=1, fan_out_init=True, **kwargs)


def MILresnext101_32x8d(pretrained=False, progress=True, **kwargs):
    """model/resnext.py:431-442"""
    return _make("resnext101_32x8d", pretrained, groups=32, width_per_group=8, decoder_expansion=1, fan_out_init=True, **kwargs)
