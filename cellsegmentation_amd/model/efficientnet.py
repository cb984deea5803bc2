"""MIL EfficientNet-B0/B2 (+B3) on the HIP engine.

Host mirror of the reference's ``MILEfficientNet`` (model/efficientnet.py:125-389) and of the three
torchvision==0.11.2 building blocks it imports (ConvNormActivation, SqueezeExcitation,
StochasticDepth), restated from their published semantics because torchvision is neither vendored in
the reference nor installed here.  Module/parameter names follow torchvision's EfficientNet
(``features.{i}.{j}.block.{k}...``, ``fc1``/``fc2`` in SE) so its checkpoints load.

Reference quirks kept: ``freeze_bn`` is a no-op for this family (the eval()/train() flip is commented
out, efficientnet.py:308-312), there is no segmentation decoder (``init_seg_modules()`` commented,
:259) so segment mode raises, and ``fc_tile`` keeps ``num_classes`` outputs (the factories never
resize it, :405-414).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine as E
from .. import functional as HF
from .. import kernels as K
from .resnet import default_compute_dtype

__all__ = ["MILEfficientNet", "MILefficientnetB0", "MILefficientnetB2", "MILefficientnetB3", "mbconv_table"]


def _make_divisible(v, divisor=8, min_value=None):
    """Channel rounding rule of the EfficientNet/MobileNet family (efficientnet.py:32-45)."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# (expand_ratio, kernel, stride, in, out, layers) of the B0 baseline (efficientnet.py:392-403)
_BASE = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
         (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
_SCALING = {"efficientnet_b0": (1.0, 1.0, 0.2), "efficientnet_b2": (1.1, 1.2, 0.3), "efficientnet_b3": (1.2, 1.4, 0.3)}


def mbconv_table(width_mult, depth_mult):
    """[(expand_ratio, kernel, stride, in_ch, out_ch, num_layers)] after width/depth scaling."""
    return [(e, k, s, _make_divisible(ci * width_mult), _make_divisible(co * width_mult), int(math.ceil(n * depth_mult)))
            for e, k, s, ci, co, n in _BASE]


class ConvNormActivation(nn.Sequential):
    """Conv2d(bias=False, padding=(k-1)//2) + BatchNorm2d [+ SiLU]; holder only."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, activation=True):
        layers = [nn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout)]
        if activation:
            layers.append(nn.SiLU(inplace=True))
        super().__init__(*layers)
        self.out_channels = cout


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze_channels):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze_channels, 1)
        self.fc2 = nn.Conv2d(squeeze_channels, channels, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()


class StochasticDepth(nn.Module):
    def __init__(self, p, mode):
        super().__init__()
        self.p, self.mode = p, mode


class MBConv(nn.Module):
    def __init__(self, expand_ratio, kernel, stride, cin, cout, sd_prob):
        super().__init__()
        if not (1 <= stride <= 2):
            raise ValueError('illegal stride value')
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        expanded = _make_divisible(cin * expand_ratio)
        if expanded != cin:
            layers.append(ConvNormActivation(cin, expanded, 1))
        layers.append(ConvNormActivation(expanded, expanded, kernel, stride, groups=expanded))
        layers.append(SqueezeExcitation(expanded, max(1, cin // 4)))
        layers.append(ConvNormActivation(expanded, cout, 1, activation=False))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")
        self.out_channels = cout


class MILEfficientNet(nn.Module):
    def __init__(self, arch, inverted_residual_setting, dropout, stochastic_depth_prob=0.2, num_classes=1000):
        super().__init__()
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        self.encoder_name = arch
        self.mode = None
        self.encoder_prefix = ("features",)
        self.image_module_prefix = ("fc_image_cls", "fc_image_reg")
        self.tile_module_prefix = ("fc_tile",)
        self.seg_module_prefix = ("upconv", "seg_out_conv")
        self.compute_dtype = default_compute_dtype()
        self.use_tr_read = True

        layers = [ConvNormActivation(3, inverted_residual_setting[0][3], 3, 2)]
        total = sum(c[5] for c in inverted_residual_setting)
        block_id = 0
        for e, k, s, cin, cout, n in inverted_residual_setting:
            stage = []
            for j in range(n):
                sd = stochastic_depth_prob * float(block_id) / total
                stage.append(MBConv(e, k, s if j == 0 else 1, cin if j == 0 else cout, cout, sd))
                block_id += 1
            layers.append(nn.Sequential(*stage))
        last_in = inverted_residual_setting[-1][4]
        self.lastconv_output_channels = 4 * last_in
        layers.append(ConvNormActivation(last_in, self.lastconv_output_channels, 1))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(p=dropout, inplace=True), nn.Linear(self.lastconv_output_channels, num_classes))
        feat = self.lastconv_output_channels
        self.feature_dim = feat
        self.avgpool_tile, self.maxpool_tile = nn.AdaptiveAvgPool2d((1, 1)), nn.AdaptiveMaxPool2d((1, 1))
        self.fc_tile = nn.Sequential(nn.Flatten(), nn.Linear(feat, num_classes))
        self.avgpool_image, self.maxpool_image = nn.AdaptiveAvgPool2d((1, 1)), nn.AdaptiveMaxPool2d((1, 1))
        self.fc_image_cls = nn.Sequential(nn.Flatten(), nn.Dropout(p=0.3), nn.Linear(feat, 7))
        self.fc_image_reg = nn.Sequential(nn.Flatten(), nn.Dropout(p=0.3), nn.Linear(feat, 1), nn.ReLU(inplace=True))
        for m in self.modules():                       # efficientnet.py:261-268
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._plans = {}

    # ------------------------------------------------------------------ requires_grad groups (efficientnet.py:270-293)
    def set_encoder_grads(self, requires_grad):
        self.features.requires_grad_(requires_grad)

    def set_tile_module_grads(self, requires_grad):
        for n in ("avgpool_tile", "maxpool_tile", "fc_tile"):
            getattr(self, n).requires_grad_(requires_grad)

    def set_image_module_grads(self, requires_grad):
        for n in ("avgpool_image", "maxpool_image", "fc_image_cls", "fc_image_reg"):
            getattr(self, n).requires_grad_(requires_grad)

    def set_seg_module_grads(self, requires_grad):
        raise AttributeError("MILEfficientNet has no segmentation decoder (the reference comments init_seg_modules() out)")

    def setmode(self, mode):
        table = {"tile": (False, True, False), "image": (True, False, True), "segment": (False, False, False)}
        if mode not in table:
            raise Exception("Invalid mode: {}.".format(mode))
        enc, tile, image = table[mode]
        self.set_encoder_grads(enc)
        self.set_tile_module_grads(tile)
        self.set_image_module_grads(image)
        self.mode = mode

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    # ------------------------------------------------------------------ plan
    def _plan(self, stochastic):
        key = ("enc", stochastic)
        if key in self._plans:
            return self._plans[key]
        units = []
        nxt = [1]

        def new():
            nxt[0] += 1
            return nxt[0] - 1

        def cna(name, seq, src, res=None):
            d = new()
            act = E.ACT_SILU if len(seq) == 3 else E.ACT_NONE
            if seq[0].groups == 1:
                units.append(E.ConvUnit(name, seq[0], seq[1], act, src, d, res=res))
            else:
                units.append(E.DwConvUnit(name, seq[0], seq[1], act, src, d))
            return d

        cur = cna("features.0", self.features[0], 0)
        n_stages = len(self.features) - 2
        for si in range(1, n_stages + 1):
            for bi, mb in enumerate(self.features[si]):
                x_in, h = cur, cur
                name = f"features.{si}.{bi}.block"
                last = len(mb.block) - 1
                for li, layer in enumerate(mb.block):
                    if isinstance(layer, SqueezeExcitation):
                        d = new()
                        units.append(E.SEUnit(f"{name}.{li}", layer.fc1, layer.fc2, h, d))
                        h = d
                    elif li == last:
                        use_sd = stochastic and mb.use_res_connect and mb.stochastic_depth.p > 0
                        if mb.use_res_connect and not use_sd:
                            h = cna(f"{name}.{li}", layer, h, res=x_in)
                        else:
                            h = cna(f"{name}.{li}", layer, h)
                            if use_sd:
                                d = new()
                                units.append(E.RowScaleAddUnit(h, x_in, d, mb.stochastic_depth.p))
                                h = d
                    else:
                        h = cna(f"{name}.{li}", layer, h)
                cur = h
        cur = cna(f"features.{n_stages + 1}", self.features[n_stages + 1], cur)
        plan = E.Plan(units, [0], [cur])
        self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ forward
    def forward(self, x, freeze_bn=False):
        if self.mode == "segment":
            raise Exception("MILEfficientNet has no segmentation path (reference: efficientnet.py:259,314,335-360)")
        if self.mode not in ("tile", "image"):
            raise Exception("Something wrong in setmode.")
        if not x.is_cuda:
            raise RuntimeError("cellsegmentation_amd models run on the GPU only (HIP kernels, no CPU fallback); "
                               "move the model and its input to a cuda device")
        xh = HF.to_nhwc(x.float(), self.compute_dtype)
        (x4,) = E.run_plan(self._plan(self.training), [xh], self.compute_dtype, self.training, self.use_tr_read)
        feat = HF.gap_avgmax(x4, self.feature_dim, relu_input=False)
        if self.mode == "tile":
            lin = self.fc_tile[1]
            return HF.linear(feat, lin.weight, lin.bias)
        cls, reg = self.fc_image_cls, self.fc_image_reg
        out_cls = HF.linear(F.dropout(feat, cls[1].p, self.training), cls[2].weight, cls[2].bias)
        out_reg = HF.linear(F.dropout(feat, reg[1].p, self.training), reg[2].weight, reg[2].bias, K.CS_ACT_RELU)
        return out_cls, out_reg


def _make(arch, pretrained, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained=True needs a download; load torchvision weights with load_state_dict(strict=False) instead")
    w, d, dropout = _SCALING[arch]
    return MILEfficientNet(arch, mbconv_table(w, d), dropout, **kwargs)


def MILefficientnetB0(pretrained=False, progress=True, **kwargs):
    return _make("efficientnet_b0", pretrained, **kwargs)


def MILefficientnetB2(pretrained=False, progress=True, **kwargs):
    return _make("efficientnet_b2", pretrained, **kwargs)


def MILefficientnetB3(pretrained=False, progress=True, **kwargs):
    """Not constructible in the reference (only its checkpoint URL is listed, efficientnet.py:21); same
    scaling rule with width 1.2 / depth 1.4 -- BASELINE.json config 4."""
    return _make("efficientnet_b3", pretrained, **kwargs)
