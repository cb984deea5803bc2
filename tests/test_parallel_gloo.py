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
        want = (1.0 + 2.0) / 2 if i != 1 else (0.0 + 0.0) / 2
        ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
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


def test_shard_bags_never_splits_a_bag():
    shards = [shard_bags(10, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10))
    assert shards[0] == [0, 4, 8] and shards[3] == [3, 7]
