"""``Adam``: torch.optim.Adam's update as ONE HIP launch (csrc/optim.hip, cs_adam_step).

The reference's drivers construct ``optim.Adam(filter(requires_grad, model.parameters()), lr, weight_decay)`` (train_tile.py:282,
train_image.py:476, train_seg.py:309).  This class keeps that constructor, ``step`` / ``zero_grad`` / ``state_dict`` /
``load_state_dict`` (the state is torch's: ``step``, ``exp_avg``, ``exp_avg_sq`` per parameter, so a checkpoint written with either
class loads into the other) and ``param_groups`` (schedulers change ``lr`` there); only the arithmetic runs in our kernel: one
launch per <= 320 tensors instead of torch's five multi-tensor launches (2.9 TB/s on the ResNet-50 tile step).

fp32 CUDA parameters only, no amsgrad / maximize.  Parameters without a gradient are skipped, like in torch.

``capturable=True`` (torch.optim.Adam's flag, same state layout: ``step`` is an fp32 scalar ON THE DEVICE): the update reads and
advances the step counts in device memory (cs_adam_step_dev: a one-thread-per-tensor launch forms the bias corrections in double,
the update kernel reads them), so ``step()`` may be captured into a HIP graph (graphed.GraphedStep) and every replay is the next
Adam step.  The learning rate is read from a device double as well: ``sync_hyper()`` (cheap; pass it as a ``pre_replay`` hook of
GraphedStep when a scheduler changes ``lr``) copies ``param_groups[i]["lr"]`` there; betas / eps / weight_decay are launch
arguments, fixed at capture.  Parameters at different step counts need no separate launches in this mode.
"""
import ctypes

import torch

from . import _capture, _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, capturable=False):
        if amsgrad:
            raise ValueError("cellsegmentation_amd.optim.Adam: amsgrad is not implemented (the reference never enables it)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        # (`capturable` in the defaults, as in torch: Optimizer.load_state_dict then keeps `step` on the parameter's device as fp32)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, capturable=bool(capturable)))
        self._plans = {}            # (group index, tensor addresses, step classes) -> plan; a few entries (alternating parameter sets)
        self._hsteps = {}           # id(param) -> [address of its `step` tensor, host copy of the value]
        self._lr_dev = {}           # group index -> [device double, the host value it holds]  (capturable groups)

    def _init_state(self, p, capturable=False):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if capturable else torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # new moments / step tensors: no cached device table or host step count may survive them (state reset from outside,
            # e.g. `opt.state.clear()`, would otherwise leave the tables pointing at freed memory -- ADVICE r4)
            self._hsteps.pop(id(p), None)
            self._plans.clear()
        return st

    def sync_hyper(self):
        """Copy every capturable group's current `lr` into the device double its (possibly captured) launches read.  One tiny fill
        per CHANGED value: call before replaying a captured step whenever a scheduler moved `lr` (GraphedStep(pre_replay=...))."""
        for gi, group in enumerate(self.param_groups):
            if not group.get("capturable", False):
                continue
            lr = float(group["lr"])
            slot = self._lr_dev.get(gi)
            if slot is None:
                p0 = next((p for p in group["params"]), None)
                if p0 is None:
                    continue
                slot = self._lr_dev[gi] = [torch.empty((), dtype=torch.float64, device=p0.device), None]
            if slot[1] != lr:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("cellsegmentation_amd.optim.Adam: lr changed since the last sync_hyper(); call opt.sync_hyper() "
                                       "before capturing (a fill captured into the graph would reset lr at every replay)")
                slot[0].fill_(lr)
                slot[1] = lr

    def _host_step(self, p):
        """Host copy of the parameter's step count: this class is the only writer of `step` (one foreach add per launch), so the
        tensor is read back only when it was replaced (load_state_dict) or never seen."""
        st = self.state[p]["step"]
        h = self._hsteps.get(id(p))
        if h is None or h[0] != st.data_ptr():
            h = self._hsteps[id(p)] = [st.data_ptr(), float(st)]
        return h

    def load_state_dict(self, state_dict):
        """torch's loader takes every hyper-parameter -- `capturable` included -- from the LOADED groups.  Where the step counts live is
        a property of this optimizer object, not of the checkpoint: each group keeps the flag it was constructed with, and the loaded
        counts move to the device (fp32 scalars) or to the host accordingly, so a checkpoint written by either form resumes in both."""
        want = [bool(g.get("capturable", False)) for g in self.param_groups]
        super().load_state_dict(state_dict)
        for g, capt in zip(self.param_groups, want):
            g["capturable"] = capt
            for p in g["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    step = torch.as_tensor(st["step"], dtype=torch.float32)
                    st["step"] = step.to(p.device).reshape(()).clone() if capt else step.cpu().reshape(()).clone()
        self._plans.clear()
        self._hsteps.clear()
        for slot in self._lr_dev.values():
            slot[1] = None

    _MAX_PLANS = 8

    def _plan(self, gi, plist, classes):
        """Device tables for one set of parameters.  `classes[i]` numbers the distinct step counts among `plist` in order of first
        appearance: the bias corrections are per-launch arguments, so parameters at different step counts (one optimizer over
        parameter sets that skip steps; the reference's train_alternative, train/train.py:240-268, alternates two sets but ends
        every iteration with equal counts) go to different launches -- one per (step count, <= cs_adam_max_tensors tensors) -- as torch.optim.Adam's
        per-parameter step allows.  Plans are cached per (parameter set, class pattern), so alternating sets do not re-plan."""
        lib = _lib.load()
        # (the moments are this class's own tensors, replaced in load_state_dict only -- which drops the plans; a parameter's storage
        # can be replaced from outside, e.g. module.to(device) after the optimizer was built, so its address is part of the key)
        key = (gi, tuple([p.data_ptr() for p in plist]), classes)
        cached = self._plans.get(key)
        if cached is not None:
            return cached
        dev = plist[0].device
        if classes is None:
            classes = (0,) * len(plist)         # capturable: the step counts are read on the device, one launch serves any mixture
        chunk, cap = lib.cs_adam_chunk_elems(), lib.cs_adam_max_tensors()
        order = sorted(range(len(plist)), key=lambda i: classes[i])            # stable: members of a class stay in parameter order
        rows = [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in plist]
        table = torch.tensor([list(rows[i]) for i in order], dtype=torch.int64).to(dev)          # CsAdamTensor[] (p, m, v, n)
        launches, t0 = [], 0
        while t0 < len(order):
            t1 = t0
            while t1 < len(order) and t1 - t0 < cap and classes[order[t1]] == classes[order[t0]]:
                t1 += 1
            members = order[t0:t1]
            ch = [(t0 + j, c) for j, i in enumerate(members) for c in range((plist[i].numel() + chunk - 1) // chunk)]
            # (first table row, members as indices into plist, chunk table, number of chunks)
            launches.append((t0, members, torch.tensor(ch, dtype=torch.int32).to(dev), len(ch)))
            t0 = t1
        extra = None
        if key[2] is None:
            # capturable: one `float*` per table row (the parameter's device step count) + 2 floats of coefficient scratch per row
            for p in plist:
                st = self.state[p]["step"]
                if not st.is_cuda or st.dtype != torch.float32 or st.numel() != 1:
                    raise RuntimeError("cellsegmentation_amd.optim.Adam(capturable=True): `step` must be an fp32 scalar on the GPU "
                                       "(state loaded from a non-capturable optimizer? load it through load_state_dict)")
            steps = torch.tensor([self.state[plist[i]]["step"].data_ptr() for i in order], dtype=torch.int64).to(dev)
            extra = (steps, torch.empty((len(order), 2), dtype=torch.float32, device=dev))
        while len(self._plans) >= self._MAX_PLANS * max(1, len(self.param_groups)):
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = (table, launches, extra)
        return self._plans[key]

    @torch.no_grad()
    def step(self, closure=None):
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing and not all(g.get("capturable", False) for g in self.param_groups):
            # the bias corrections are kernel ARGUMENTS computed from a host-side step count: a captured launch would replay the
            # step number of the capture forever
            raise RuntimeError("cellsegmentation_amd.optim.Adam.step() cannot be captured into a HIP graph (the step count lives on "
                               "the host); construct it with capturable=True")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            state = self.state
            capt = bool(group.get("capturable", False))
            for p in plist:
                if p not in state or len(state[p]) == 0:      # first step of this parameter: validate once, then trust the plan key
                    if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.is_sparse:
                        raise RuntimeError("cellsegmentation_amd.optim.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only")
                    if capturing:
                        raise RuntimeError("cellsegmentation_amd.optim.Adam: run one eager step before capturing (the optimizer state is "
                                           "created on the first step; GraphedStep's warm-up does)")
                    self._init_state(p, capt)
                g = p.grad
                if g.dtype != torch.float32 or g.is_sparse:
                    raise RuntimeError("cellsegmentation_amd.optim.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only")
            beta1, beta2 = group["betas"]
            stream = torch.cuda.current_stream(plist[0].device).cuda_stream
            if capt:
                self.sync_hyper()
                plan = self._plan(gi, plist, None)
                table, launches, (steps, coef) = plan
                lr_dev = self._lr_dev[gi][0]
                if capturing:
                    _capture.keep((plan, lr_dev))            # the plan cache is LRU: a captured launch keeps its tables alive itself
                for t0, members, chunks, n_chunks in launches:
                    n = len(members)
                    gts = [plist[i].grad if plist[i].grad.is_contiguous() else plist[i].grad.contiguous() for i in members]
                    grads = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gts])
                    _lib.check(lib.cs_adam_step_dev(table.data_ptr(), grads, t0, n, chunks.data_ptr(), n_chunks, steps.data_ptr(), coef.data_ptr(),
                                                    lr_dev.data_ptr(), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                                    float(group["weight_decay"]), stream), "adam_step_dev")
                continue
            hs = [self._host_step(p) for p in plist]
            seen = {}
            classes = tuple(seen.setdefault(h[1], len(seen)) for h in hs)
            table, launches, _ = self._plan(gi, plist, classes)
            for t0, members, chunks, n_chunks in launches:
                n = len(members)
                t = hs[members[0]][1] + 1.0
                gts = [plist[i].grad if plist[i].grad.is_contiguous() else plist[i].grad.contiguous() for i in members]   # (alive until queued)
                grads = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gts])
                _lib.check(lib.cs_adam_step(table.data_ptr(), grads, t0, n, chunks.data_ptr(), n_chunks, float(group["lr"]), float(beta1),
                                            float(beta2), float(group["eps"]), float(group["weight_decay"]), t, stream), "adam_step")
                for i in members:
                    hs[i][1] = t
                torch._foreach_add_([self.state[plist[i]]["step"] for i in members], 1.0)
        return loss
