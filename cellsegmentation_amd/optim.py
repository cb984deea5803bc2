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
        self._plans = {}            # group index -> (key, tensor table, [(t0, n, chunk table, n_chunks)])

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _plan(self, gi, plist):
        lib = _lib.load()
        key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel(),
                     self.state[p]["step"].data_ptr()) for p in plist)
        cached = self._plans.get(gi)
        if cached is not None and cached[0] == key:
            return cached
        dev = plist[0].device
        chunk, cap = lib.cs_adam_chunk_elems(), lib.cs_adam_max_tensors()
        table = torch.tensor([[k[0], k[1], k[2], k[3]] for k in key], dtype=torch.int64).to(dev)          # CsAdamTensor[] (p, m, v, n)
        launches = []
        for t0 in range(0, len(plist), cap):
            sub = plist[t0:t0 + cap]
            ch = [(t0 + i, c) for i, p in enumerate(sub) for c in range((p.numel() + chunk - 1) // chunk)]
            by_step = {float(self.state[p]["step"]) for p in sub}
            if len(by_step) != 1:
                raise RuntimeError("cellsegmentation_amd.optim.Adam: parameters of one launch are at different step counts "
                                   "(a parameter skipped earlier steps); use torch.optim.Adam for such schedules")
            # [t0, n, chunk table, n_chunks, step count so far (host copy: the per-parameter `step` tensors are advanced with one
            #  foreach add per step, not read back), the step tensors]
            launches.append([t0, len(sub), torch.tensor(ch, dtype=torch.int32).to(dev), len(ch), by_step.pop(), [self.state[p]["step"] for p in sub]])
        self._plans[gi] = (key, table, launches)
        return self._plans[gi]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("cellsegmentation_amd.optim.Adam: contiguous fp32 CUDA parameters with dense fp32 gradients only")
                self._init_state(p)
            beta1, beta2 = group["betas"]
            _, table, launches = self._plan(gi, plist)
            stream = torch.cuda.current_stream(plist[0].device).cuda_stream
            for ln in launches:
                t0, n, chunks, n_chunks = ln[0], ln[1], ln[2], ln[3]
                sub = plist[t0:t0 + n]
                t = ln[4] + 1.0
                gts = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in sub]       # (kept alive until the launch is queued)
                grads = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gts])
                _lib.check(lib.cs_adam_step(table.data_ptr(), grads, t0, n, chunks.data_ptr(), n_chunks, float(group["lr"]), float(beta1),
                                            float(beta2), float(group["eps"]), float(group["weight_decay"]), t, stream), "adam_step")
                ln[4] = t
                torch._foreach_add_(ln[5], 1.0)
        return loss
