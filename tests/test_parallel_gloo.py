"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (what runs over RCCL on the GPUs)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cellsegmentation_amd.parallel import GradReducer, shard_bags


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                     # different initial weights on purpose
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    model[2].bias.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    red = GradReducer(params, bucket_bytes=64)        # tiny buckets: several collectives
    red.broadcast_parameters(model)
    w0 = model[0].weight.detach().clone()
    # rank-local gradients: grad = (rank+1) * ones, one param left without grad
    for i, p in enumerate(params):
        p.grad = None if i == 1 else torch.full_like(p, float(rank + 1))
    red.reduce()
    ok = True
    for i, p in enumerate(params):
        if i == 1:
            ok = ok and p.grad is None                # no gradient on any rank: stays None, the optimizer skips it as in a one-process run
        else:
            ok = ok and torch.allclose(p.grad, torch.full_like(p, (1.0 + 2.0) / 2))
    # a slice that held a gradient in an earlier step travels as zeros again once its parameter has none (and .grad stays None)
    for i, p in enumerate(params):
        p.grad = None if i == 0 else torch.full_like(p, float(rank + 1))
    red.reduce()
    b, view, _, _ = red._slot[id(params[0])]
    ok = ok and params[0].grad is None and not bool(view.abs().max() > 0) and torch.allclose(params[1].grad, torch.full_like(params[1], 1.5))
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    ok = ok and torch.equal(gathered[0], gathered[1])  # broadcast made the replicas identical
    q.put((rank, bool(ok), len(red.buckets)))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(nb > 1 for _, _, nb in res)


def _emulated_backward(red, params, rank, step):
    """What engine.backward + autograd's AccumulateGrad do with a gradient sink: ask for the destination, write the
    rank-local gradient there, report it, then adopt (or accumulate) the returned tensor as .grad.  Deepest first."""
    for i, p in reversed(list(enumerate(params))):
        g = torch.full_like(p, float((rank + 1) * (i + 1) + step))
        v = red.view_for(p)
        if v is not None:
            v.copy_(g)
            g = v
        out = red.deliver(p, g)
        if p.grad is None:
            p.grad = out
        else:
            p.grad += out


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 300, 5, 129, 64, 2)]
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    red = GradReducer(params + [frozen], bucket_bytes=1024).attach()
    early, ok = [], True

    def mean_local(i, step):
        return sum(float((r + 1) * (i + 1) + step) for r in range(world)) / world

    for step in range(3):                                  # zero_grad(set_to_none=True) loops
        for p in params:
            p.grad = None
        _emulated_backward(red, params, rank, step)
        red.reduce()
        early.append(red.launches_in_backward)
        for i, p in enumerate(params):
            ok = ok and torch.allclose(p.grad, torch.full_like(p, mean_local(i, step)))
    # no zero_grad at all: gradients accumulate into the (bucket-aliasing) .grad, nothing may be sent early
    before = [p.grad.clone() for p in params]
    _emulated_backward(red, params, rank, 7)
    red.reduce()
    acc_early = red.launches_in_backward
    for i, p in enumerate(params):
        ok = ok and torch.allclose(p.grad, before[i] + mean_local(i, 7))
    # and the normal loop still works afterwards
    for p in params:
        p.grad = None
    _emulated_backward(red, params, rank, 9)
    red.reduce()
    for i, p in enumerate(params):
        ok = ok and torch.allclose(p.grad, torch.full_like(p, mean_local(i, 9)))
    # a second backward without reduce() is refused instead of silently double-counting
    for p in params:
        p.grad = None
    _emulated_backward(red, params, rank, 1)
    refused = False
    try:
        for p in params:
            p.grad = None
        _emulated_backward(red, params, rank, 1)
    except RuntimeError:
        refused = True
    red.detach()
    q.put((rank, bool(ok), early, acc_early, refused, len(red.buckets)))
    dist.destroy_process_group()


def test_grad_reducer_overlaps_from_inside_backward_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ok, early, acc_early, refused, nb in res:
        assert ok
        assert nb > 1
        assert early[0] == 0                 # first step only records the order gradients finish in
        assert early[1] == nb and early[2] == nb      # afterwards every bucket leaves during backward
        assert acc_early == 0
        assert refused


def _unadopted_worker(rank, world, port, q):
    """autograd's AccumulateGrad does NOT always adopt the tensor deliver() returns as `.grad` -- a parameter that
    also receives a gradient from plain autograd in the same backward gets the SUM in a fresh tensor, an extra reference makes it a
    clone.  Emulated here for two parameters: one whose bucket leaves from inside backward, one whose bucket is still there in
    reduce().  `.grad` must come out as the average of the rank-local TOTALS either way."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 300, 5, 129, 64, 2)]
    red = GradReducer(params, bucket_bytes=1024).attach()
    summed, cloned = {1, 4}, {3}             # indices: engine gradient + an autograd contribution / adopted as a clone

    def local(i, r, step):
        return float((r + 1) * (i + 1) + step)

    def extra(i, r):
        return float(10 * (r + 1) + i)

    ok, stragglers = True, []
    for step in range(3):
        for p in params:
            p.grad = None
        for i, p in reversed(list(enumerate(params))):
            v = red.view_for(p)
            v.copy_(torch.full_like(p, local(i, rank, step)))
            out = red.deliver(p, v)
            if i in summed:
                p.grad = out + extra(i, rank)          # a new tensor holding the rank-local total
            elif i in cloned:
                p.grad = out.clone()
            else:
                p.grad = out
        red.reduce()
        stragglers.append(red.unadopted_after_early_launch)
        for i, p in enumerate(params):
            want = sum(local(i, r, step) + (extra(i, r) if i in summed else 0.0) for r in range(world)) / world
            ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
    red.detach()
    q.put((rank, bool(ok), stragglers, len(red.buckets)))
    dist.destroy_process_group()


def test_grad_reducer_survives_gradients_autograd_did_not_adopt_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unadopted_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ok, stragglers, nb in res:
        assert ok
        assert nb > 1
        assert stragglers[0] == 0            # first step: nothing leaves early, everything is packed from .grad in reduce()
        assert stragglers[1] == stragglers[2] == 3      # later steps: every bucket leaves during backward -> three own all-reduces


def test_shard_bags_never_splits_a_bag():
    shards = [shard_bags(10, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10))
    assert shards[0] == [0, 4, 8] and shards[3] == [3, 7]


def test_bucket_layout_tapers_the_engine_part_and_keeps_the_rest_apart():
    """GradReducer._build: the gradients the engine reports during backward (in the order it finishes them) get a tapered tail --
    buckets of at most 1/2, 1/4, 1/8, 1/16 of `bucket_bytes` at the very end, where nothing is left to hide an all-reduce behind --
    and parameters the engine never reports sit in buckets of their own behind them.  No process group involved."""
    params = [torch.nn.Parameter(torch.zeros(n)) for n in [4096] * 40 + [1000, 300, 64, 64]]      # 40 x 16 KiB, then small ones
    rest = [torch.nn.Parameter(torch.zeros(n)) for n in (512, 8)]
    red = GradReducer(params + rest, bucket_bytes=128 << 10)
    for p in params:                                  # what the first attached step records
        red._order_ids.add(id(p))
        red._order.append(p)
    red._build(red._order + rest, taper=True, n_engine=len(params))
    sizes = [b.flat.numel() * 4 for b in red.buckets]
    members = [[id(p) for p, _, _ in b.items] for b in red.buckets]
    assert [i for m in members for i in m] == [id(p) for p in params + rest]          # order kept, everything placed once
    rest_ids = {id(p) for p in rest}
    n_rest = sum(1 for m in members if set(m) & rest_ids)
    assert n_rest == red.n_rest_buckets == 1 and set(members[-1]) == rest_ids           # the rest: its own bucket(s), last
    eng = sizes[:len(sizes) - n_rest]
    assert max(eng) <= 128 << 10
    # tapered tail: a bucket closes once it holds at least its cap (whole parameters: 16 KiB here), caps 1/16, 1/8, 1/4, 1/2 from the end
    assert eng[-1] <= (128 << 10) // 4 and eng[-2] <= (128 << 10) // 4 and eng[-3] <= (128 << 10) // 2 and eng[-4] <= 128 << 10
    assert eng[-2] < eng[-3] < eng[-4] <= eng[0] == 128 << 10                               # ... behind full-size buckets
    # slices start on 256-byte boundaries and views alias the flat buffers
    for b in red.buckets:
        for (p, o, n), v in zip(b.items, b.views):
            assert o % 64 == 0 and v.data_ptr() == b.flat.data_ptr() + 4 * o and v.shape == p.shape
