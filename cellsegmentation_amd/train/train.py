"""Epoch loops with the reference's signatures (train/train.py:12-207).

The loop bodies are the timed "step" of the hot path: zero_grad -> HIP forward -> HIP loss ->
HIP backward -> optimizer.step.  Two deliberate host-side differences from the reference, neither
changing results: the per-step ``loss.item()`` sync is replaced by an on-device accumulator read once
per epoch, and ``print`` of ce/dice per step in train_seg is dropped (the CE there is never used).
"""
import os
import weakref

import torch
from torch.optim.lr_scheduler import CyclicLR, OneCycleLR

from .. import functional as HF

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(x, **kw):
        return x


def _per_step(scheduler):
    return isinstance(scheduler, (CyclicLR, OneCycleLR))


# ---- optional: the loop bodies replay zero_grad -> forward -> loss -> backward as ONE HIP graph (graphed.GraphedGrad); the optimizer
# and the scheduler the driver passed in stay eager.  Off by default (the reference's semantics, launch by launch); CELLSEG_GRAPH_STEPS=1
# or train.use_graphed_steps(True) switch it on.  A step is captured once a batch SHAPE has been seen `_WARM` times eagerly (lazily built
# tables, allocator); other shapes -- the ragged last batch -- keep running eagerly.  The eager launches of the ResNet-18 image counter
# at batch 8 are host-bound (~500 launches, 4.5 ms of enqueueing for 1.9 ms of GPU work).
_GRAPH_STEPS = [os.environ.get("CELLSEG_GRAPH_STEPS", "0") == "1"]
_WARM = int(os.environ.get("CELLSEG_GRAPH_WARM", "2") or 2)
_RUNNERS = weakref.WeakKeyDictionary()       # model -> {key: _Runner}


def use_graphed_steps(on=True):
    """Replay the forward + backward of the train_* loops as HIP graphs (see above).  Returns the previous setting."""
    prev, _GRAPH_STEPS[0] = _GRAPH_STEPS[0], bool(on)
    return prev


class _Runner:
    """One loop's step: eager, or (graphed steps on) a GraphedGrad per input signature.  body(*tensors) -> tuple of loss tensors, the
    first one is back-propagated."""

    def __init__(self, model, optimizer, body):
        self.model, self.optimizer, self.body = model, optimizer, body
        self.seen, self.graphs, self.broken = {}, {}, False
        self.params = [p for g in optimizer.param_groups for p in g["params"]]

    def __call__(self, *tensors):
        if _GRAPH_STEPS[0] and all(t.is_cuda for t in tensors):
            key = tuple((tuple(t.shape), t.dtype) for t in tensors)
            g = self.graphs.get(key)
            if g is None and self.seen.get(key, 0) >= _WARM and len(self.graphs) < 2 and not self.broken:
                from ..graphed import GraphedGrad
                try:
                    g = self.graphs[key] = GraphedGrad(self.params, self.body, tensors)
                except Exception as e:  # noqa: BLE001 -- e.g. a criterion that synchronises (.item()): this loop stays eager
                    import warnings
                    self.broken = True
                    for p in self.params:
                        p.grad = None
                    warnings.warn(f"cellsegmentation_amd.train: the step could not be captured into a HIP graph ({type(e).__name__}: {e}); "
                                  "this loop keeps running eagerly")
            if g is not None:
                return g(*tensors)
            self.seen[key] = self.seen.get(key, 0) + 1
            if self.graphs:
                # an eager step between replays: the graphs' static gradient buffers must not be accumulated into
                for p in self.params:
                    p.grad = None
        self.optimizer.zero_grad()
        outs = tuple(self.body(*tensors))
        outs[0].backward()
        # detached: a loss tensor the loop keeps until its next iteration would keep this step's autograd graph alive -- and with it the
        # AccumulateGrad nodes of the parameters, which remember the stream they were created on (the default stream here).  A capture
        # starting while they live re-uses them, autograd then synchronises the capture stream with the legacy default stream, and
        # hipStreamEndCapture crashes (seen: segmentation fault in capture_end).
        return tuple(o.detach() for o in outs)


def _runner(model, optimizer, name, key, body):
    """The loop's _Runner.  With graphed steps on it is kept on the model across epochs (a capture per epoch would cost more than it
    saves), keyed by everything the captured body closes over: the loop, the optimizer, the criteria / weights (`key`), the model's mode
    and which parameters train."""
    if not _GRAPH_STEPS[0]:
        return _Runner(model, optimizer, body)
    full = (name, id(optimizer), key, getattr(model, "mode", None), tuple(p.requires_grad for p in model.parameters()))
    cache = _RUNNERS.setdefault(model, {})            # (not on the model: train_ensemble.py:202 deep-copies models)
    r = cache.get(full)
    if r is None:
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        r = cache[full] = _Runner(model, optimizer, body)
    return r


def _ce(criterion, output, label, gamma=1.0):
    """nn.CrossEntropyLoss instances are routed to the HIP softmax-CE kernel; anything else is called as is."""
    if isinstance(criterion, torch.nn.CrossEntropyLoss) and criterion.weight is None and criterion.reduction == "mean" \
            and getattr(criterion, "label_smoothing", 0.0) == 0.0:
        return HF.cross_entropy(output, label, gamma)
    return criterion(output, label) * gamma


def _mse(criterion, output, target):
    if isinstance(criterion, torch.nn.MSELoss) and criterion.reduction == "mean":
        return HF.mse_loss(output, target)
    return criterion(output, target)


def train_tile(loader, epoch, total_epochs, model, device, criterion, optimizer, scheduler, gamma):
    """Tile training for one epoch (train/train.py:12-48)."""
    model.train()
    tile_num = 0
    train_loss = torch.zeros((), device=device)
    train_bar = tqdm(loader, desc="tile training")
    run = _runner(model, optimizer, "tile", (id(criterion), float(gamma)), lambda x, y: (_ce(criterion, model(x, freeze_bn=True), y, gamma),))
    for i, (data, label) in enumerate(train_bar):
        loss, = run(data.to(device), label.to(device))           # zero_grad -> forward -> loss -> backward (train/train.py:32-35)
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        tile_num += data.size(0)
        train_loss += loss.detach() * data.size(0)
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    return float(train_loss.item()) / tile_num


def train_image(loader, epoch, total_epochs, model, device, crit_cls, crit_reg, optimizer, scheduler, alpha, beta):
    """Image-level classification + count regression for one epoch (train/train.py:51-105)."""
    model.train()
    acc = torch.zeros((3,), device=device)

    def body(x, yc, yn):
        output = model(x)
        l_cls = _ce(crit_cls, output[0], yc)
        l_reg = _mse(crit_reg, output[1].squeeze(), yn)
        return alpha * l_cls + beta * l_reg, l_cls, l_reg
    run = _runner(model, optimizer, "image", (id(crit_cls), id(crit_reg), float(alpha), float(beta)), body)
    for i, (data, label_cls, label_num) in enumerate(tqdm(loader, desc="image training")):
        loss, l_cls, l_reg = run(data.to(device), label_cls.to(device), label_num.to(device, dtype=torch.float32))
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        acc += torch.stack([l_cls.detach(), l_reg.detach(), loss.detach()]) * data.size(0)
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    n = len(loader.dataset)
    c, r, t = (acc / n).tolist()
    return c, r, t


def train_image_cls(loader, epoch, total_epochs, model, device, crit_cls, optimizer, scheduler):
    """train/train.py:109-137"""
    model.train()
    acc = torch.zeros((), device=device)
    run = _runner(model, optimizer, "image_cls", (id(crit_cls),), lambda x, yc: (_ce(crit_cls, model(x)[0], yc),))
    for i, (data, label_cls, label_num) in enumerate(tqdm(loader, desc="image training")):
        l_cls, = run(data.to(device), label_cls.to(device))
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        acc += l_cls.detach() * data.size(0)
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    return float(acc.item()) / len(loader.dataset)


def train_image_reg(loader, epoch, total_epochs, model, device, crit_reg, optimizer, scheduler):
    """train/train.py:140-169"""
    model.train()
    acc = torch.zeros((), device=device)
    run = _runner(model, optimizer, "image_reg", (id(crit_reg),), lambda x, yn: (_mse(crit_reg, model(x)[1].squeeze(), yn),))
    for i, (data, label_cls, label_num) in enumerate(tqdm(loader, desc="image training")):
        l_reg, = run(data.to(device), label_num.to(device, dtype=torch.float32))
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        acc += l_reg.detach() * data.size(0)
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    return float(acc.item()) / len(loader.dataset)


def train_seg(loader, epoch, total_epochs, model, device, optimizer, scheduler):
    """Segmentation training for one epoch (train/train.py:172-207): loss = Dice(softmax(out)[:,1], mask/255)."""
    model.train()
    acc = torch.zeros((), device=device)
    run = _runner(model, optimizer, "seg", (), lambda x, m: (HF.dice_loss(HF.softmax_channel(model(x).to(dtype=torch.float32), 1), m),))
    for i, (image, mask, label) in enumerate(tqdm(loader, desc="segmentation training")):
        mask = (mask / 255).to(device, dtype=torch.float32)
        loss, = run(image.to(device), mask)
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        acc += loss.detach() * image.size(0)
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    return float(acc.item()) / len(loader.dataset)


def train_alternative(loader, epoch, total_epochs, model, device, crit_cls, crit_reg, optimizer, scheduler, threshold, alpha, beta,
                      gamma, delta):
    """Alternating tile / image training for one epoch (train/train.py:210-300; no driver of the reference calls it).  Per batch
    `(data, labels)` with data = (images, tiles) and labels = (image class, image count, tile label): a TILE step (setmode("tile"),
    plain forward -- the reference passes no freeze_bn here --, loss gamma * crit_cls) and an IMAGE step (setmode("image"),
    alpha * crit_cls + beta * crit_reg) on ONE optimizer, zero_grad in between.  setmode("tile") freezes the encoder, as in the
    reference (resnet.py:308-333), so the tile step updates the tile head only and the image step the encoder + the image heads: in
    this loop every trained parameter ends an iteration at the same step count (cellsegmentation_amd.optim.Adam keeps a step count
    per parameter, as torch.optim.Adam does, for schedules where parameters do skip steps).  `threshold` and `delta`
    are unused by the reference too (its segmentation part is commented out).  Returns
    (tile_loss, image_cls_loss, image_reg_loss, image_seg_loss = 0.0, image_loss)."""
    tile_num = 0
    acc = torch.zeros((4,), device=device)          # tile, image cls, image reg, image total (weighted by batch sizes)
    for i, (data, labels) in enumerate(tqdm(loader, desc="alternative training")):
        # pt.1: tile training
        model.setmode("tile")
        model.train()
        optimizer.zero_grad()
        output = model(data[1].to(device))
        tile_loss_i = _ce(crit_cls, output, labels[2].to(device), gamma)
        tile_loss_i.backward()
        optimizer.step()
        tile_num += data[1].size(0)
        # pt.2: image training
        model.setmode("image")
        model.train()
        optimizer.zero_grad()
        output = model(data[0].to(device))
        l_cls = _ce(crit_cls, output[0], labels[0].to(device))
        l_reg = _mse(crit_reg, output[1].squeeze(), labels[1].to(device, dtype=torch.float32))
        image_loss_i = alpha * l_cls + beta * l_reg
        image_loss_i.backward()
        optimizer.step()
        if _per_step(scheduler):
            scheduler.step()
        nt, ni = data[1].size(0), data[0].size(0)
        acc += torch.stack([tile_loss_i.detach() * nt, l_cls.detach() * ni, l_reg.detach() * ni, image_loss_i.detach() * ni])
    if not (scheduler is None or _per_step(scheduler)):
        scheduler.step()
    t, c, r, tot = acc.tolist()
    n = len(loader.dataset)
    return t / max(1, tile_num), c / n, r / n, 0.0, tot / n
