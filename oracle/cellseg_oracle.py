"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain, functional restatement (torch CPU fp32, NCHW) of the reference's hot path, driven by a
state_dict that uses the reference's own key names.  It exists to CHECK the HIP path; nothing in
``cellsegmentation_amd/`` may import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it.

Pinned: ``tests/golden/make_golden.py`` imports the real reference modules from /root/reference
(model/resnet.py, model/resnext.py, train/losses.py, metrics/metrics.py, inference.py) in the build
container, asserts this restatement reproduces them (<=1e-6) on identical weights/inputs, and
commits the resulting vectors under tests/golden/.  The EfficientNet branch is NOT pinned by the
reference (its torchvision==0.11.2 dependency is absent from /root/reference and from this image):
"parity unpinned" for that branch -- it restates torchvision's published ConvNormActivation /
SqueezeExcitation / StochasticDepth semantics.

Each function cites the reference lines it follows.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# architecture tables (model/resnet.py:336-361, model/resnext.py:418-442)
# ---------------------------------------------------------------------------------------------
RESNETS = {
    "resnet18": dict(block="basic", depths=(2, 2, 2, 2), groups=1, width_per_group=64),
    "resnet34": dict(block="basic", depths=(3, 4, 6, 3), groups=1, width_per_group=64),
    "resnet50": dict(block="bottleneck", depths=(3, 4, 6, 3), groups=1, width_per_group=64),
    "resnext50_32x4d": dict(block="bottleneck", depths=(3, 4, 6, 3), groups=32, width_per_group=4),
    "resnext101_32x8d": dict(block="bottleneck", depths=(3, 4, 23, 3), groups=32, width_per_group=8),
}


def expansion(arch):
    return 4 if RESNETS[arch]["block"] == "bottleneck" else 1


def _bn(sd, pfx, x, train, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d / BatchNorm1d forward; train=True uses batch statistics and updates the
    running buffers in place exactly like the module does."""
    if train and (pfx + ".num_batches_tracked") in sd:
        sd[pfx + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"], sd[pfx + ".weight"], sd[pfx + ".bias"],
                        train, momentum, eps)


def _conv(sd, name, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride, padding, 1, groups)


def _basic_block(sd, p, x, stride, bn_train):
    """model/resnet.py:28-43"""
    out = torch.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1), bn_train))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1), bn_train)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), bn_train)
    return torch.relu(out + res)


def _bottleneck(sd, p, x, stride, groups, bn_train):
    """model/resnet.py:60-78 (v1.5: stride on the 3x3), model/resnext.py:93-113 (grouped 3x3)"""
    out = torch.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x), bn_train))
    out = torch.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride, 1, groups), bn_train))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out), bn_train)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), bn_train)
    return torch.relu(out + res)


def encoder(sd, x, arch, bn_train):
    """resnet_forward, model/resnet.py:234-248: returns (x4, x3, x2, x1)."""
    cfg = RESNETS[arch]
    x = torch.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 2, 3), bn_train))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, depth in enumerate(cfg["depths"], start=1):
        for b in range(depth):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            if cfg["block"] == "basic":
                x = _basic_block(sd, p, x, stride, bn_train)
            else:
                x = _bottleneck(sd, p, x, stride, cfg["groups"], bn_train)
        feats.append(x)
    x1, x2, x3, x4 = feats
    return x4, x3, x2, x1


def _pooled(x4):
    """AdaptiveAvgPool2d(1)+AdaptiveMaxPool2d(1), model/resnet.py:266,274"""
    return (F.adaptive_avg_pool2d(x4, 1) + F.adaptive_max_pool2d(x4, 1)).flatten(1)


def _image_head(sd, name, feat, train, final_relu):
    """fc_image_cls / fc_image_reg, model/resnet.py:132-152. Dropout is the identity here
    (parity runs use p=0; the RNG stream of nn.Dropout cannot be reproduced)."""
    h = _bn(sd, name + ".1", feat, train)
    h = torch.relu(h)
    h = F.linear(h, sd[name + ".4.weight"], sd[name + ".4.bias"])
    h = _bn(sd, name + ".5", h, train)
    h = F.linear(h, sd[name + ".7.weight"], sd[name + ".7.bias"])
    return torch.relu(h) if final_relu else h


def _upconv(sd, name, x, train):
    """upsample_conv = Conv3x3(bias)+BN+ReLU, model/resnet.py:195-200"""
    return torch.relu(_bn(sd, name + ".1", _conv(sd, name + ".0", x, 1, 1), train))


def _up(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def seg_decoder(sd, x4, x3, x2, x1, in_hw, train):
    """model/resnet.py:280-303; the five target sizes are those of x3, x2, x1, conv1-out, input
    (19/38/75/150/299 for the reference's hard-wired 299 input)."""
    h, w = in_hw
    c1 = ((h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1)
    o = _upconv(sd, "upconv1", _up(x4, x3.shape[-2:]), train)
    o = _upconv(sd, "upconv2", torch.cat([o, x3], 1), train)
    o = _upconv(sd, "upconv3", _up(o, x2.shape[-2:]), train)
    o = _upconv(sd, "upconv4", torch.cat([o, x2], 1), train)
    o = _upconv(sd, "upconv5", _up(o, x1.shape[-2:]), train)
    o = _upconv(sd, "upconv6", torch.cat([o, x1], 1), train)
    o = _upconv(sd, "upconv7", _up(o, c1), train)
    o = _upconv(sd, "upconv8", o, train)
    return _conv(sd, "seg_out_conv", _up(o, (h, w)))


def forward(sd, x, arch, mode, freeze_bn=False, training=True):
    """MILResNet.forward / MILResNeXt.forward (model/resnet.py:250-306).
    training=True & mode=='tile' & freeze_bn: every BN of the trunk runs on running statistics
    (the self.eval()/self.train() flip of :254-258); the tile head has no BN/dropout."""
    if mode == "tile":
        bn_train = training and not freeze_bn
        x4, _, _, _ = encoder(sd, x, arch, bn_train)
        feat = _pooled(x4)
        return F.linear(feat, sd["fc_tile.1.weight"], sd["fc_tile.1.bias"])
    if mode == "image":
        x4, _, _, _ = encoder(sd, x, arch, training)
        feat = _pooled(x4)
        return (_image_head(sd, "fc_image_cls", feat, training, False), _image_head(sd, "fc_image_reg", feat, training, True))
    if mode == "segment":
        x4, x3, x2, x1 = encoder(sd, x, arch, training)
        return seg_decoder(sd, x4, x3, x2, x1, x.shape[-2:], training)
    raise Exception("Something wrong in setmode.")


# ---------------------------------------------------------------------------------------------
# state_dict construction with the reference's key names and shapes (model/resnet.py:108-168)
# ---------------------------------------------------------------------------------------------
def _bn_entries(sd, pfx, c):
    sd[pfx + ".weight"] = torch.ones(c)
    sd[pfx + ".bias"] = torch.zeros(c)
    sd[pfx + ".running_mean"] = torch.zeros(c)
    sd[pfx + ".running_var"] = torch.ones(c)
    sd[pfx + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def empty_state_dict(arch):
    cfg = RESNETS[arch]
    exp = expansion(arch)
    sd = OrderedDict()
    sd["conv1.weight"] = torch.zeros(64, 3, 7, 7)
    _bn_entries(sd, "bn1", 64)
    inplanes = 64
    for li, depth in enumerate(cfg["depths"], start=1):
        planes = 64 * 2 ** (li - 1)
        for b in range(depth):
            p = f"layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            if cfg["block"] == "basic":
                sd[p + ".conv1.weight"] = torch.zeros(planes, inplanes, 3, 3)
                _bn_entries(sd, p + ".bn1", planes)
                sd[p + ".conv2.weight"] = torch.zeros(planes, planes, 3, 3)
                _bn_entries(sd, p + ".bn2", planes)
            else:
                width = int(planes * (cfg["width_per_group"] / 64.0)) * cfg["groups"]
                sd[p + ".conv1.weight"] = torch.zeros(width, inplanes, 1, 1)
                _bn_entries(sd, p + ".bn1", width)
                sd[p + ".conv2.weight"] = torch.zeros(width, width // cfg["groups"], 3, 3)
                _bn_entries(sd, p + ".bn2", width)
                sd[p + ".conv3.weight"] = torch.zeros(planes * exp, width, 1, 1)
                _bn_entries(sd, p + ".bn3", planes * exp)
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                sd[p + ".downsample.0.weight"] = torch.zeros(planes * exp, inplanes, 1, 1)
                _bn_entries(sd, p + ".downsample.1", planes * exp)
            inplanes = planes * exp
    feat = 512 * exp
    sd["fc_tile.1.weight"] = torch.zeros(2, feat)
    sd["fc_tile.1.bias"] = torch.zeros(2)
    for name, nout in (("fc_image_cls", 7), ("fc_image_reg", 1)):
        _bn_entries(sd, name + ".1", feat)
        sd[name + ".4.weight"] = torch.zeros(64, feat)
        sd[name + ".4.bias"] = torch.zeros(64)
        _bn_entries(sd, name + ".5", 64)
        sd[name + ".7.weight"] = torch.zeros(nout, 64)
        sd[name + ".7.bias"] = torch.zeros(nout)
    # decoder: resnet.py:156-164 uses `expansion`; resnext.py:209-217 hard-codes expansion 1
    e = exp if arch.startswith("resnet") else 1
    chans = [(512 * e, 256 * e), (512 * e, 256 * e), (256 * e, 128 * e), (256 * e, 128 * e), (128 * e, 64 * e),
             (128 * e, 64 * e), (64 * e, 64 if e == 1 else 32 * e), (64 if e == 1 else 32 * e, 64)]
    for i, (ci, co) in enumerate(chans, start=1):
        sd[f"upconv{i}.0.weight"] = torch.zeros(co, ci, 3, 3)
        sd[f"upconv{i}.0.bias"] = torch.zeros(co)
        _bn_entries(sd, f"upconv{i}.1", co)
    sd["seg_out_conv.weight"] = torch.zeros(2, 64, 1, 1)
    sd["seg_out_conv.bias"] = torch.zeros(2)
    return sd


# ---------------------------------------------------------------------------------------------
# losses / metrics / selection (train/losses.py, metrics/metrics.py, inference.py)
# ---------------------------------------------------------------------------------------------
def weighted_mse(inputs, targets, reduction="mean"):
    """metrics/metrics.py:23-33: weights start as a clone of targets; entries >= 20 become ln(t)."""
    w = torch.where(targets >= 20, torch.log(targets.clamp(min=1e-30)), targets)
    tmp = w * (inputs - targets) ** 2
    return tmp.mean() if reduction == "mean" else tmp.sum()


def dice_coef(inputs, targets, eps=1e-6):
    """metrics/metrics.py:36-53"""
    if inputs.ndim == 2 and targets.ndim == 2:
        a = (inputs * targets).sum(); b = (inputs * inputs).sum(); c = (targets * targets).sum()
    else:
        i = inputs.reshape(inputs.shape[0], -1)
        t = targets.reshape(targets.shape[0], -1).float()
        a = (i * t).sum(1); b = (i * i).sum(1); c = (t * t).sum(1)
    return (2 * a + eps) / (b + c + eps)


def dice_loss(inputs, targets, eps=1e-6, reduction="mean"):
    """train/losses.py:52-62"""
    d = 1 - dice_coef(inputs, targets, eps)
    return d.mean() if reduction == "mean" else d.sum()


def tile_step_loss(sd, x, labels, arch, gamma=1.0):
    """train_tile loop body, train/train.py:32-34"""
    return F.cross_entropy(forward(sd, x, arch, "tile", freeze_bn=True, training=True), labels) * gamma


def image_step_loss(sd, x, cls, count, arch, alpha=1.0, beta=1.0):
    """train_image loop body, train/train.py:77-83 (nn.MSELoss on out_reg.squeeze())"""
    out_cls, out_reg = forward(sd, x, arch, "image", training=True)
    l_cls = F.cross_entropy(out_cls, cls)
    l_reg = F.mse_loss(out_reg.squeeze(), count.float())
    return l_cls, l_reg, alpha * l_cls + beta * l_reg


def seg_step_loss(sd, x, mask01, arch):
    """train_seg loop body, train/train.py:184-195: Dice on softmax channel 1 (implicit dim = 1)."""
    out = forward(sd, x, arch, "segment", training=True)
    return dice_loss(F.softmax(out, dim=1)[:, 1], mask01)


def tile_probs(sd, x, arch):
    """inference_tiles, inference.py:9-28: eval forward, softmax, prob of class 1."""
    with torch.no_grad():
        return F.softmax(forward(sd, x, arch, "tile", training=False), dim=1)[:, 1].numpy()


def sample_indices(probs, groups, labels, tiles_per_pos, topk_neg):
    """The list inference.sample hands to make_train_data (inference.py:34-42)."""
    groups = np.asarray(groups)
    order = np.lexsort((probs, groups))
    T = len(groups)
    keep = np.empty(T, dtype=bool)
    for i in range(T):
        lab = labels[groups[i]]
        k = topk_neg if lab == 0 else lab * tiles_per_pos
        keep[i] = groups[i] != groups[(i + k) % T]
    return order[keep]


def categorize(x):
    """dataset/dataset.py:745-761 count -> 7-way class label."""
    for lab, hi in enumerate((0, 5, 10, 20, 50, 200)):
        if x <= hi:
            return lab
    return 6


def _axis_origins(length, interval, size):
    o = list(range(0, length - size + 1, interval))
    if o[-1] + size != length:          # the sliding window adds one border-aligned tile (dataset.py:731-732,734-740)
        o.append(length - size)
    return o


def get_tiles_coords(h, w, interval, size):
    """dataset/dataset.py:718-742: upper-left (row, col) of every tile, row-major, with a final
    border-aligned row/column when the stride does not land on the edge."""
    return [(x, y) for x in _axis_origins(h, interval, size) for y in _axis_origins(w, interval, size)]


# ---------------------------------------------------------------------------------------------
# EfficientNet branch -- PARITY UNPINNED by the reference (torchvision==0.11.2 is neither vendored in
# /root/reference nor installed): restates model/efficientnet.py:81-122,179-214,295-333 on top of the
# published semantics of torchvision's ConvNormActivation (conv no-bias, pad (k-1)//2, BN, SiLU),
# SqueezeExcitation (avgpool -> 1x1+bias -> SiLU -> 1x1+bias -> sigmoid -> scale) and
# StochasticDepth("row") (identity in eval or p=0).
# ---------------------------------------------------------------------------------------------
_EFF_BASE = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
             (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
EFF_SCALING = {"efficientnet_b0": (1.0, 1.0), "efficientnet_b2": (1.1, 1.2), "efficientnet_b3": (1.2, 1.4)}


def _divisible(v, d=8):
    nv = max(d, int(v + d / 2) // d * d)
    return nv + d if nv < 0.9 * v else nv


def eff_table(arch):
    w, dm = EFF_SCALING[arch]
    return [(e, k, s, _divisible(ci * w), _divisible(co * w), int(math.ceil(n * dm))) for e, k, s, ci, co, n in _EFF_BASE]


def _cna(sd, pfx, x, stride, groups, train, act=True):
    w = sd[pfx + ".0.weight"]
    y = _bn(sd, pfx + ".1", F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, groups), train)
    return F.silu(y) if act else y


def eff_encoder(sd, x, arch, train, sd_noise=None):
    """sd_noise: optional iterator over the residual blocks in order -- per block the StochasticDepth("row") factor of torchvision 0.11.2
    (ops/stochastic_depth.py: bernoulli(1 - p) / (1 - p), one value per sample) or None for the identity (eval / p = 0)."""
    x = _cna(sd, "features.0", x, 2, 1, train)
    table = eff_table(arch)
    for si, (e, k, s, cin, cout, n) in enumerate(table, start=1):
        for b in range(n):
            ci, st = (cin, s) if b == 0 else (cout, 1)
            p = f"features.{si}.{b}.block"
            h, li = x, 0
            expanded = _divisible(ci * e)
            if expanded != ci:
                h = _cna(sd, f"{p}.{li}", h, 1, 1, train); li += 1
            h = _cna(sd, f"{p}.{li}", h, st, expanded, train); li += 1
            sc = F.adaptive_avg_pool2d(h, 1)
            sc = F.silu(F.conv2d(sc, sd[f"{p}.{li}.fc1.weight"], sd[f"{p}.{li}.fc1.bias"]))
            sc = torch.sigmoid(F.conv2d(sc, sd[f"{p}.{li}.fc2.weight"], sd[f"{p}.{li}.fc2.bias"]))
            h = sc * h; li += 1
            h = _cna(sd, f"{p}.{li}", h, 1, 1, train, act=False)
            if st == 1 and ci == cout:
                noise = next(sd_noise) if sd_noise is not None else None
                x = (h if noise is None else h * noise.view(-1, 1, 1, 1)) + x
            else:
                x = h
    return _cna(sd, f"features.{len(table) + 1}", x, 1, 1, train)


def eff_forward(sd, x, arch, mode, training=True, sd_noise=None):
    """MILEfficientNet.forward (efficientnet.py:305-333); freeze_bn is a no-op there. Dropout = identity."""
    feat = _pooled(eff_encoder(sd, x, arch, training, sd_noise))
    if mode == "tile":
        return F.linear(feat, sd["fc_tile.1.weight"], sd["fc_tile.1.bias"])
    if mode == "image":
        return (F.linear(feat, sd["fc_image_cls.2.weight"], sd["fc_image_cls.2.bias"]),
                torch.relu(F.linear(feat, sd["fc_image_reg.2.weight"], sd["fc_image_reg.2.bias"])))
    raise Exception("Something wrong in setmode.")


# ---------------------------------------------------------------- either side of the top-k
def make_train_data(tile_idx, tiles_grid, labels, idxs, pos_neg_ratio, perm):
    """dataset/dataset.py:166-201 with `np.random.shuffle(self.train_data)` replaced by the explicit permutation `perm`
    (position p of the shuffled array holds entry perm[p]).  Returns (rows [(tileIDX, x, y, label)], pos, neg)."""
    train = [(int(tile_idx[i]), int(tiles_grid[i][0]), int(tiles_grid[i][1]), 0 if labels[tile_idx[i]] == 0 else 1) for i in idxs]
    pos = sum(r[3] for r in train)
    neg = len(train) - pos
    train = [train[j] for j in perm]
    if pos_neg_ratio is not None:
        if pos > int(neg * pos_neg_ratio):
            flag, n = 1, pos - int(neg * pos_neg_ratio)
            pos = int(neg * pos_neg_ratio)
        elif neg > int(pos / pos_neg_ratio):
            flag, n = 0, neg - int(pos / pos_neg_ratio)
            neg = int(pos / pos_neg_ratio)
        else:
            return np.asarray(train, dtype=np.int64).reshape(-1, 4), pos, neg
        excess = []
        for i, r in enumerate(train):
            if r[3] == flag:
                excess.append(i)
            if len(excess) == n:
                break
        drop = set(excess)
        train = [r for i, r in enumerate(train) if i not in drop]
    return np.asarray(train, dtype=np.int64).reshape(-1, 4), pos, neg


def evaluate_tile(tile_idx, labels, probs, tiles_per_pos, threshold):
    """evaluate.py:8-27 + metrics/metrics.py:7-16 (calc_err)."""
    groups = np.array(tile_idx)
    probs = np.asarray(probs)
    order = np.lexsort((probs, groups))
    groups = groups[order]
    p = probs[order]
    pred = np.array([x > threshold for x in p])
    real = np.zeros(len(p))
    for i in range(1, len(p) + 1):
        if i == len(p) or groups[i] != groups[i - 1]:
            n = labels[groups[i - 1]] * tiles_per_pos
            real[i - n: i] = [1] * n
    neq = np.not_equal(pred, real)
    with np.errstate(divide="ignore", invalid="ignore"):
        err = float(neq.sum()) / pred.shape[0]
        fpr = float(np.logical_and(pred == 1, neq).sum()) / (real == 0).sum()
        fnr = float(np.logical_and(pred == 0, neq).sum()) / (real == 1).sum()
    return err, fpr, fnr


def rank_tiles(tile_idx, tiles_grid, probs, threshold):
    """The nested `rank` of train_seg.py:234-249."""
    groups = np.array(tile_idx)
    tiles = np.array(tiles_grid)
    probs = np.asarray(probs)
    order = np.lexsort((probs, groups))
    groups, p, tiles = groups[order], probs[order], tiles[order]
    index = [x > threshold for x in p]
    return tiles[index], p[index], groups[index]


def generate_masks(n_images, image_size, tile_size, tiles, groups):
    """utils/image_processing.py:90-98 (square painting; no pre-processing, nothing saved).  Tiles lie inside the image
    (dataset.get_tiles aligns the last row / column to the border); the reference's slice assignment fails otherwise."""
    masks = np.zeros((n_images, *image_size)).astype(np.uint8)
    for i in range(len(groups)):
        x, y = int(tiles[i][0]), int(tiles[i][1])
        masks[groups[i]][x: x + tile_size, y: y + tile_size] = np.ones((tile_size, tile_size)).astype(np.uint8)
    return masks
