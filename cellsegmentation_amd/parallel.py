"""Data-parallel gradient exchange: one process per GPU, RCCL (`backend="nccl"`) over xGMI.

The tile minibatch shards by bag across ranks: forward/backward are rank-local (no SyncBN
in the reference), and the only collective is one SUM all-reduce of the trainable gradients per
step, divided by world size -- what the reference's DistributedDataParallel stub
(train_tile.py:228-235) would do implicitly.  Gradients are packed into a few large flat fp32
buckets (fewer, larger collectives suit point-to-point xGMI links), reduced on a dedicated stream so
packing bucket k+1 overlaps the collective of bucket k, then scaled and unpacked.
Works unchanged with the ``gloo`` backend on CPU tensors (unit tests).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []          # [(flat buffer, [(param, offset, numel)])]
        cur, cur_n = [], 0
        # reverse order = the order gradients become ready in backward
        for p in reversed(self.params):
            n = p.numel()
            if cur and (cur_n + n) * 4 > bucket_bytes:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, n))
            cur_n += n
        if cur:
            self._close(cur, cur_n)
        self._stream = None

    def _close(self, items, n):
        dev = items[0][0].device
        self.buckets.append((torch.zeros((n,), dtype=torch.float32, device=dev), items))

    def broadcast_parameters(self, module, src=0):
        """DDP-constructor semantics: rank `src`'s parameters and buffers everywhere."""
        if self.world == 1:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    @torch.no_grad()
    def reduce(self):
        """All-reduce(SUM)/world of every .grad, in place."""
        if self.world == 1:
            return
        cuda = self.buckets and self.buckets[0][0].is_cuda
        if cuda and self._stream is None:
            self._stream = torch.cuda.Stream()
        works = []
        for flat, items in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p, _, _ in items]
            views = [flat[o:o + n].view_as(g) for (_, o, n), g in zip(items, grads)]
            torch._foreach_copy_(views, grads)
            if cuda:
                self._stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._stream):
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            else:
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if cuda:
            torch.cuda.current_stream().wait_stream(self._stream)
        inv = 1.0 / self.world
        for flat, items in self.buckets:
            flat.mul_(inv)
            for p, o, n in items:
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))


def shard_bags(n_bags, rank, world):
    """Round-robin whole bags (images with all their tiles) to ranks: a bag is never split, so the
    segmented top-k stays rank-local."""
    return list(range(rank, n_bags, world))
