"""``Adam``: torch.optim.Adam's update as ONE HIP launch (csrc/optim.hip, cs_adam_step).

The reference's drivers construct ``optim.Adam(filter(requires_grad, model.parameters()), lr, weight_decay)`` (train_tile.py:282,
train_image.py:476, train_seg.py:309).  This class keeps that constructor, ``step`` / ``zero_grad`` / ``state_dict`` /
``load_state_dict`` (the state is torch's: ``step``, ``exp_avg``, ``exp_avg_sq`` per parameter, so a checkpoint written with either
class loads into the other) and ``param_groups`` (schedulers change ``lr`` there); only the arithmetic runs in our kernel: one
launch per <= 320 tensors instead of torch's five multi-tensor launches (2.9 TB/s on the ResNet-50 tile step).

fp32 CUDA parameters only, no amsgrad / maximize / capturable (a step captured into a HIP graph keeps the step count on the host:
use torch.optim.Adam(capturable=True) there, as tools/bench_configs.py's graphed configs do).  Parameters without a gradient are
skipped, like in torch.
"""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise ValueError("cellsegmentation_amd.optim.Adam: amsgrad is not implemented (the reference never enables it)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._plans = {}            # (group index, tensor addresses, step classes) -> plan; a few entries (alternating parameter sets)
        self._hsteps = {}           # id(param) -> [address of its `step` tensor, host copy of the value]

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _host_step(self, p):
        """Host copy of the parameter's step count: this class is the only writer of `step` (one foreach add per launch), so the
        tensor is read back only when it was replaced (load_state_dict) or never seen."""
        st = self.state[p]["step"]
        h = self._hsteps.get(id(p))
        if h is None or h[0] != st.data_ptr():
            h = self._hsteps[id(p)] = [st.data_ptr(), float(st)]
        return h

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()
        self._hsteps.clear()

    _MAX_PLANS = 8

    def _plan(self, gi, plist, classes):
        """Device tables for one set of parameters.  `classes[i]` numbers the distinct step counts among `plist` in order of first
        appearance: the bias corrections are per-launch arguments, so parameters at different step counts (the reference's
        train_alternative, train/train.py:240-268: one optimizer, tile steps and image steps in turn, the encoder ahead of both
        heads) go to different launches -- one per (step count, <= cs_adam_max_tensors tensors) -- as torch.optim.Adam's
        per-parameter step allows.  Plans are cached per (parameter set, class pattern), so alternating sets do not re-plan."""
        lib = _lib.load()
        # (the moments are this class's own tensors, replaced in load_state_dict only -- which drops the plans; a parameter's storage
        # can be replaced from outside, e.g. module.to(device) after the optimizer was built, so its address is part of the key)
        key = (gi, tuple([p.data_ptr() for p in plist]), classes)
        cached = self._plans.get(key)
        if cached is not None:
            return cached
        dev = plist[0].device
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
        while len(self._plans) >= self._MAX_PLANS:
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = (table, launches)
        return self._plans[key]

    @torch.no_grad()
    def step(self, closure=None):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # the bias corrections are kernel ARGUMENTS computed from a host-side step count: a captured launch would replay the
            # step number of the capture forever
            raise RuntimeError("cellsegmentation_amd.optim.Adam.step() cannot be captured into a HIP graph (the step count lives on "
                               "the host); use torch.optim.Adam(capturable=True) inside graphs")
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
            for p in plist:
                if p not in state or len(state[p]) == 0:      # first step of this parameter: validate once, then trust the plan key
                    if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.is_sparse:
                        raise RuntimeError("cellsegmentation_amd.optim.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only")
                    self._init_state(p)
                g = p.grad
                if g.dtype != torch.float32 or g.is_sparse:
                    raise RuntimeError("cellsegmentation_amd.optim.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only")
            beta1, beta2 = group["betas"]
            hs = [self._host_step(p) for p in plist]
            seen = {}
            classes = tuple(seen.setdefault(h[1], len(seen)) for h in hs)
            table, launches = self._plan(gi, plist, classes)
            stream = torch.cuda.current_stream(plist[0].device).cuda_stream
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
