"""One training step as a HIP graph.

The BN-train paths (image-wise counter, EfficientNet) issue ~1000 small launches per step from Python and are host-bound:
4 ms of GPU work in a 7-10 ms step for the ResNet-18 counter at batch 8.  Every kernel of this library is enqueued on torch's
current stream with caller-owned buffers and no host synchronisation, so a whole step -- forward, loss, the HIP backward,
``optimizer.step()`` -- captures into one graph (`torch.cuda.CUDAGraph`, i.e. hipGraph on ROCm) and replays in a single launch:
3.4 ms per step for that configuration.

Constraints (torch's graph-capture rules): fixed input shapes (use the eager step for a ragged last batch), no `.item()` /
`.cpu()` inside the step, an optimizer that supports capture (`cellsegmentation_amd.optim.Adam(..., capturable=True)` -- the
one-launch HIP Adam with its step counts on the device --, `torch.optim.Adam(..., capturable=True)`, SGD).  Parameters, BN
running statistics and optimizer state are updated in place by the replay exactly as by the eager step.  Host code of the step
does not run at a replay: a learning-rate scheduler is stepped by the caller after the call, and what it changed reaches the
captured launches through a `pre_replay` hook (`optimizer.sync_hyper` of the HIP Adam).

Round 5: the ResNet-50 tile step of train/train.py:29-42 (`--scratch`) is captured too (bench.py's headline; bit-for-bit equality
with eager steps in tests/test_graphed_gpu.py): 6.0 ms of host work per step leave the critical path.
"""
import time

import torch

from . import _capture, engine


def _capture_mode():
    """`global` (torch's default) forbids potentially unsafe HIP calls from ANY thread while a capture is open.  With a NCCL / RCCL process
    group alive its watchdog thread polls the events of earlier collectives (hipEventQuery) every ~100 ms: inside a global-mode capture
    that call fails with hipErrorStreamCaptureUnsupported, the watchdog throws and the PROCESS aborts (seen in bench.py's one-rank RCCL
    side run, one run in three).  With a process group alive: let the queue drain, give the watchdog a poll interval to retire what
    completed, and capture in `thread_local` mode (only the capturing thread is policed)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            torch.cuda.synchronize()
            time.sleep(0.3)
            return "thread_local"
    except Exception:  # noqa: BLE001
        pass
    return "global"


class GraphedStep:
    """``step = GraphedStep(step_fn, (x, y, ...))`` then ``loss = step(x, y, ...)`` per batch.

    step_fn(*tensors) runs ONE full training step on its arguments and returns a tensor (e.g. the loss) or a tuple of tensors.
    Construction runs `warmup` REAL steps on the example batch on a side stream (allocator warm-up, lazily built staging tables:
    they do update the model), then captures one more into the graph without executing it.  Each call copies the new batch into
    the captured input buffers and replays; the returned tensors are the captured outputs (overwritten by the next call)."""

    def __init__(self, step_fn, example_inputs, warmup=3, pre_replay=()):
        """pre_replay: callables run (eagerly, on the host) before the capture and before every replay -- e.g. `optimizer.sync_hyper`."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        self.pre_replay = tuple(pre_replay)
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for hook in self.pre_replay:
            hook()
        self.graph = torch.cuda.CUDAGraph()
        _capture.take()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.outputs = step_fn(*self.static_inputs)
        self._refs = _capture.take()        # cached library objects the captured launches point into (see _capture.py)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise ValueError("GraphedStep: expected %d inputs" % len(self.static_inputs))
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("GraphedStep: input shape/dtype differs from the captured one; run the eager step for this batch")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        for hook in self.pre_replay:
            hook()
        self.graph.replay()
        # the replay updated parameters / BN statistics without running host code: cached staged weights are stale now
        engine.invalidate_staged()
        return self.outputs


class GraphedGrad:
    """zero_grad -> forward -> loss -> backward of a training step as one HIP graph; ``optimizer.step()`` stays outside.

    For loops whose optimizer cannot be captured -- the reference's drivers build a plain ``optim.Adam`` whose step counts live on the
    host (train_tile.py:282) -- the ~95 % of a step's launches that are NOT the optimizer still replay in one launch
    (``train.use_graphed_steps``).  ``fn(*tensors)`` returns a tuple of tensors, the first of which is the loss to back-propagate.
    During the capture every ``.grad`` of ``params`` starts as None, so autograd adopts the gradient buffers the HIP backward
    wrote -- after the capture ``p.grad`` IS that static buffer, every replay overwrites it, and the caller must NOT call
    ``optimizer.zero_grad()`` between replays (the parameters that receive no gradient keep ``.grad is None``, as in the eager loop).
    The constructor runs no step; the first call replays.  No autograd graph of an EARLIER eager step may be alive at construction (a
    loss tensor with a grad_fn kept in a variable is enough): its AccumulateGrad nodes belong to the default stream, which cannot join a
    capture."""

    def __init__(self, params, fn, example_inputs):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedGrad needs a GPU")
        self.params = [p for p in params if p.requires_grad]
        self.static_inputs = [t.clone() for t in example_inputs]
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        _capture.take()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            for p in self.params:
                p.grad = None
            outs = tuple(fn(*self.static_inputs))
            outs[0].backward()
            self.outputs = tuple(o.detach() for o in outs)       # (no autograd graph is kept: see train._Runner)
        del outs
        self._refs = _capture.take()
        self.grads = [p.grad for p in self.params]

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("GraphedGrad: input shape/dtype differs from the captured one")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        for p, g in zip(self.params, self.grads):       # (a zero_grad() between replays is undone, not obeyed: the buffers are the graph's)
            if p.grad is not g:
                p.grad = g
        self.graph.replay()
        engine.invalidate_staged()
        return self.outputs
