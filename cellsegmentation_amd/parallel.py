"""Data-parallel gradient exchange: one process per GPU, RCCL (`backend="nccl"`) over xGMI.

The tile minibatch shards by bag across ranks: forward/backward are rank-local (no SyncBN
in the reference), and the only collective is a SUM all-reduce of the trainable gradients per step,
divided by world size -- what the reference's DistributedDataParallel stub (train_tile.py:228-235)
would do implicitly.

Gradients live in a few flat fp32 buckets (fewer, larger collectives suit point-to-point xGMI links):

* ``attach()`` registers the reducer as the engine's gradient sink.  The HIP backward then writes each
  weight / BN gradient straight into its bucket slice (no pack copy) and reports it the moment its kernels
  are enqueued; a bucket whose last gradient has arrived is all-reduced on a side stream while the rest of
  backward keeps the compute queue busy.  Buckets are laid out in the order gradients were seen to finish
  during the first step (deepest layers first), so the big layer3/layer4 buckets -- ~90 % of the bytes of a
  ResNet-50 -- are on the wire long before the stem's gradients exist.
* ``reduce()`` (after ``backward()``, before ``optimizer.step()``) sends whatever has not gone yet
  (gradients produced outside the engine, e.g. the classifier head), waits for the side stream and makes
  every ``p.grad`` hold the average (RCCL: ``ReduceOp.AVG``, the division happens inside the collective; gloo: SUM and one scaling
  pass).  A trainable parameter that received no gradient keeps ``grad is None`` as in a one-process run.

One ``backward()`` per ``reduce()`` (the reference's loops: zero_grad -> backward -> step), with
``zero_grad(set_to_none=True)`` (torch's default) so autograd adopts the bucket slices as ``.grad``
instead of accumulating into a stale tensor; anything else falls back to the post-backward path for that
step.  Works unchanged with the ``gloo`` backend (CPU tensors in the unit tests, CUDA tensors in the
one-GPU two-process rehearsal).
"""
import torch
import torch.distributed as dist

from . import engine as _engine

_ALIGN = 64      # floats: every bucket slice starts on a 256-byte boundary


class _Bucket:
    __slots__ = ("flat", "items", "views", "ptrs", "dirty", "pending", "launched", "streams")

    def __init__(self, n, device):
        self.flat = torch.zeros((n,), dtype=torch.float32, device=device)
        self.items = []          # (param, offset, numel)
        self.views = []          # the slice of `flat` shaped like each parameter (made once: slicing costs microseconds per call)
        self.ptrs = []           # data_ptr() of each slice
        self.dirty = []          # the slice has held a gradient since it was last zero (`flat` starts as zeros)
        self.pending = 0
        self.launched = False
        self.streams = set()     # streams whose kernels wrote gradients into this bucket since the last reduce()


class GradReducer:
    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, force_collectives=False):
        """`force_collectives`: issue the broadcasts / all-reduces even in a world of ONE rank (a one-rank RCCL communicator is a
        real communicator: the bucket hand-off, the side-stream ordering and the RCCL launch cost can be exercised and timed on a
        single GPU -- tests/test_rccl_world1_gpu.py, bench.py --force-reduce).  Requires an initialised process group."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force_collectives and not dist.is_initialized():
            raise RuntimeError("GradReducer(force_collectives=True) needs torch.distributed to be initialised")
        self._active = self.world > 1 or bool(force_collectives)       # False: every method is a pass-through
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_bytes)
        self._stream = None
        self._attached = False
        self._order = []           # parameters in the order the engine finished them (recorded on the first attached step)
        self._order_ids = set()
        self._overlap_ok = True    # this step: gradients may be reduced from inside backward
        self._slot = {}            # id(param) -> (bucket, offset, numel)
        self.launches_in_backward = 0     # buckets sent before reduce() in the last step (diagnostics / tests)
        self.unadopted_after_early_launch = 0      # gradients of the last step that had to be all-reduced on their own (see reduce())
        self.n_rest_buckets = 0           # buckets holding parameters the engine never reports (they can only go in reduce())
        self._timed = []
        self._exposed = []
        self._build(list(reversed(self.params)))       # reverse order ~ the order gradients become ready

    # ------------------------------------------------------------------ layout
    def _build(self, ordered, taper=False, n_engine=None):
        """Cut `ordered` (parameters in the order their gradients become ready) into buckets of at most `bucket_bytes`.
        taper: the LAST buckets are small (1/16, 1/8, 1/4, 1/2 of `bucket_bytes` from the end) -- whatever is still on the wire when
        backward ends is exposed, and the stem / first-stage gradients that finish last are a few hundred KiB, so the tail that
        cannot overlap shrinks from a full bucket to ~2 MB at the price of three more (small) collectives.  n_engine: only the
        first n_engine parameters are reported by the engine during backward (the taper is theirs); the rest -- gradients that plain
        autograd produces, parameters that get none -- can only go in reduce() and sit in buckets of their own behind them."""
        self.buckets, self._slot = [], {}
        sizes = [(p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for p in ordered]
        n_engine = len(ordered) if n_engine is None else n_engine
        cuts = {n_engine} if 0 < n_engine < len(ordered) else set()          # indices i: a bucket ends before parameter i
        if taper and n_engine > 1:
            caps = [self.bucket_bytes // 16, self.bucket_bytes // 8, self.bucket_bytes // 4, self.bucket_bytes // 2]
            i, acc, ci = n_engine - 1, 0, 0
            while i > 0 and ci < len(caps):
                acc += sizes[i] * 4
                if acc >= caps[ci]:
                    cuts.add(i)
                    acc, ci = 0, ci + 1
                i -= 1
        cur, cur_n = [], 0

        def close():
            b = _Bucket(cur_n, cur[0][0].device)
            for p, o, n in cur:
                b.items.append((p, o, n))
                view = b.flat[o:o + n].view_as(p)
                b.views.append(view)
                b.ptrs.append(view.data_ptr())
                b.dirty.append(False)
                self._slot[id(p)] = (b, view, view.data_ptr(), len(b.views) - 1)
            self.buckets.append(b)

        for i, p in enumerate(ordered):
            n = p.numel()
            if cur and ((cur_n + n) * 4 > self.bucket_bytes or i in cuts):
                close()
                cur, cur_n = [], 0
            cur.append((p, cur_n, n))
            cur_n += sizes[i]
        if cur:
            close()
        self.n_rest_buckets = sum(1 for b in self.buckets if any(id(p) not in self._order_ids for p, _, _ in b.items)) if self._order_ids else 0
        self._begin_step()

    def _begin_step(self):
        for b in self.buckets:
            b.pending = len(b.items)
            b.launched = False
            b.streams = set()
        self._overlap_ok = True

    def broadcast_parameters(self, module, src=0):
        """DDP-constructor semantics: rank `src`'s parameters and buffers everywhere."""
        if not self._active:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    # ------------------------------------------------------------------ engine-facing (gradient sink)
    def attach(self):
        """Reduce from inside the HIP backward (see the module docstring).  Returns self."""
        _engine.set_grad_sink(self)
        self._attached = True
        return self

    def detach(self):
        if self._attached:
            _engine.set_grad_sink(None)
            self._attached = False

    def view_for(self, param):
        """Bucket slice the engine should write this parameter's gradient into (None: not one of ours)."""
        s = self._slot.get(id(param))
        if s is None or not self._active or param.grad is not None:      # an existing .grad may BE this slice: never write under it
            return None
        b, view, _, i = s
        if b.launched:
            raise RuntimeError("GradReducer: a second backward() before reduce(); call reduce() once per backward")
        b.dirty[i] = True
        return view

    def deliver(self, param, grad):
        """The kernels producing `grad` are enqueued on the current stream.  Returns the tensor autograd should see."""
        s = self._slot.get(id(param))
        if s is None or not self._active:
            return grad
        if not self._layout_final and id(param) not in self._order_ids:
            self._order_ids.add(id(param))
            self._order.append(param)
        b, view, vptr, i = s
        if b.launched:
            raise RuntimeError("GradReducer: a second backward() before reduce(); call reduce() once per backward")
        b.dirty[i] = True
        if param.grad is not None:
            # autograd will ACCUMULATE `grad` into .grad (which may alias our slice) after we return: leave the slice alone,
            # send nothing of this bucket early; reduce() picks the total up from .grad
            self._overlap_ok = False
            return grad
        if grad is not view and grad.data_ptr() != vptr:
            view.copy_(grad)
        if (_engine.WGRAD_SIDE_STREAM or _engine.WGRAD_SIDE_MAX_M > 0) and view.is_cuda:
            b.streams.add(torch.cuda.current_stream())      # (opt-in engine mode: weight gradients finished on a side stream)
        b.pending -= 1
        if b.pending == 0 and self._overlap_ok and self._layout_final:
            self._launch(b)
        # a FRESH alias: autograd's AccumulateGrad adopts an incoming gradient as .grad only when nobody else holds a reference to
        # the tensor object; the cached view would be cloned instead (161 copies per ResNet-50 step, and as many back in reduce())
        return view.detach()

    # ------------------------------------------------------------------ collectives
    def _all_reduce_mean(self, t):
        """SUM over the ranks / world, in place.  RCCL divides inside the collective (ReduceOp.AVG: no second pass over the bucket);
        gloo has no AVG."""
        if self._avg is None:
            self._avg = dist.get_backend(self.group) == "nccl"
        if self._avg:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                t.mul_(1.0 / self.world)

    def _launch(self, b):
        if b.flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            for st in b.streams:
                self._stream.wait_stream(st)
            with torch.cuda.stream(self._stream):
                if self.time_collectives:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._stream)
                self._all_reduce_mean(b.flat)
                if self.time_collectives:
                    e1.record(self._stream)
                    self._timed.append((b.flat.numel() * 4, e0, e1))
        else:
            self._all_reduce_mean(b.flat)
        b.launched = True

    _avg = None
    _layout_final = False
    time_collectives = False          # True: bracket every bucket's all-reduce (+ scale) with events on the side stream

    def collective_times(self):
        """[(bucket bytes, ms)] of the collectives issued since the last call (synchronises; diagnostics only)."""
        torch.cuda.synchronize()
        out = [(n, e0.elapsed_time(e1)) for n, e0, e1 in self._timed]
        self._timed = []
        return out

    def exposed_ms(self):
        """Milliseconds the compute stream waited for the side stream in the reduce() calls since the last call (with
        time_collectives on): the exchange time that backward did not cover.  Synchronises; diagnostics only."""
        torch.cuda.synchronize()
        out = sum(a.elapsed_time(b) for a, b in self._exposed)
        self._exposed = []
        return out

    @torch.no_grad()
    def reduce(self):
        """All-reduce(SUM)/world of every .grad, in place (finishes what backward has not already sent)."""
        if not self._active:
            return
        sent_early = sum(1 for b in self.buckets if b.launched)
        # Gradients delivered from inside backward are expected to have been ADOPTED by autograd as an alias of their bucket slice.
        # AccumulateGrad does not always adopt: it clones when somebody else still references the tensor, and it SUMS when the
        # parameter also received a gradient from plain autograd in the same backward (or under create_graph).  Then `.grad`
        # holds the rank-local total and the slice only what the engine wrote.  For a bucket that is still here the pack below
        # takes the total from `.grad`; for a bucket that already left, the parameter's `.grad` is all-reduced on its own (same
        # program on every rank: the same parameters take this path everywhere, in the same order).
        # A parameter WITHOUT a gradient (trainable but not part of this mode's forward: the decoder in tile mode) keeps
        # `grad is None`, exactly as in a one-process run -- the optimizer skips it; its slice travels as zeros (written once:
        # `flat` starts as zeros and is only re-zeroed after the slice has held a gradient).  Pure data parallelism runs the same
        # program on every rank, so such a parameter has no gradient anywhere.
        stragglers, back = [], []
        for b in self.buckets:
            early = b.launched
            dirty = b.dirty
            for i, (item, vptr) in enumerate(zip(b.items, b.ptrs)):
                g = item[0].grad
                if g is None:
                    if dirty[i] and not early:
                        b.views[i].zero_()
                        dirty[i] = False
                elif g.data_ptr() != vptr:
                    if early:
                        stragglers.append(g)
                    else:
                        # pack what the engine did not write in place: gradients that came through plain autograd, gradients autograd
                        # did not adopt, or every gradient when this step could not overlap (accumulation into an existing .grad)
                        b.views[i].copy_(g)
                        dirty[i] = True
                        back.append((g, b.views[i]))
            if not early:
                self._launch(b)
        if stragglers:
            if stragglers[0].is_cuda:
                self._stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._stream):
                    for g in stragglers:
                        self._all_reduce_mean(g)
            else:
                for g in stragglers:
                    self._all_reduce_mean(g)
        self.unadopted_after_early_launch = len(stragglers)
        if self._stream is not None:
            cur = torch.cuda.current_stream()
            if self.time_collectives:
                # how long the compute stream sits in this wait = the part of the exchange backward did not hide
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(cur)
                cur.wait_stream(self._stream)
                eb.record(cur)
                self._exposed.append((ea, eb))
            else:
                cur.wait_stream(self._stream)
        for g, view in back:
            g.copy_(view)                  # (.grad that IS the slice: nothing to copy)
        if self._attached and not self._layout_final and self._order:
            # first attached step done: lay the buckets out in the order the engine finishes gradients (everything the engine
            # never reported goes last, it is only available after backward anyway).  Old slices that .grad may still alias
            # stay alive through those tensors; the next zero_grad() drops them.
            rest = [p for p in reversed(self.params) if id(p) not in self._order_ids]
            self._layout_final = True
            self._build(self._order + rest, taper=True, n_engine=len(self._order))
        else:
            self._begin_step()
        self.launches_in_backward = sent_early


def shard_bags(n_bags, rank, world):
    """Round-robin whole bags (images with all their tiles) to ranks: a bag is never split, so the
    segmented top-k stays rank-local."""
    return list(range(rank, n_bags, world))
