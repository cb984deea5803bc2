"""GPU, two processes on the one card, gloo carrying CUDA tensors: the data-parallel path exactly as bench.py --gpus N
runs it (engine backward writing into the reducer's buckets, buckets leaving from inside backward), with only the
transport swapped -- RCCL refuses two ranks on one device."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:       # a silent child would leave the parent waiting on the queue
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cellsegmentation_amd import synth
    from cellsegmentation_amd.model import resnet as R
    from cellsegmentation_amd.parallel import GradReducer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    m = R.MILresnet18()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.float32)
    m.setmode("tile")
    m.set_encoder_grads(True)                  # --scratch: the whole trunk trains (batched weight-gradient path)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                        # the expectation below re-runs the same forward
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    xs = [synth.normalise(synth.ihc_tiles(8, 32, 500 + r)).to(dev) for r in range(world)]
    ys = [torch.tensor([(i + r) % 2 for i in range(8)], device=dev) for r in range(world)]

    def local_grads(r):
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(m(xs[r], freeze_bn=True), ys[r]).backward()
        return [p.grad.clone() if p.grad is not None else None for p in params]      # tile mode leaves the image heads unused

    expect = [None if gs[0] is None else sum(gs) / world for gs in zip(*[local_grads(r) for r in range(world)])]      # no reducer involved

    red = GradReducer(params, bucket_bytes=4 << 20).attach()
    red.broadcast_parameters(m)
    early, worst = [], 0.0
    for step in range(3):
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(m(xs[rank], freeze_bn=True), ys[rank]).backward()
        red.reduce()
        early.append(red.launches_in_backward)
        for p, e in zip(params, expect):
            if e is None:
                worst = max(worst, 0.0 if p.grad is None else 1.0)       # no gradient on any rank: stays None, as in a one-process run
            else:
                worst = max(worst, float((p.grad - e).abs().max() / (e.abs().max() + 1e-12)))
    red.detach()
    torch.cuda.synchronize()
    q.put((rank, worst, early, len(red.buckets) - red.n_rest_buckets + 1))      # (+1: the historical "all but the head's bucket")
    dist.destroy_process_group()


def test_engine_backward_feeds_the_reducer_two_ranks_one_gpu(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for r in res:
        assert len(r) == 4, r[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, worst, early, nb in res:
        assert worst < 1e-6, worst
        assert nb >= 3
        assert early[0] == 0                      # the first step records the order
        assert early[1] >= nb - 1 and early[2] >= nb - 1     # then all but (at most) the head's bucket leave inside backward


def _ddp_worker(rank, world, port, q):
    try:
        _ddp_worker_body(rank, world, port, q)
    except BaseException:
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


def _ddp_worker_body(rank, world, port, q):
    """The reference's own `--distributed` stub (train_tile.py:227-235): `nn.parallel.DistributedDataParallel(model,
    device_ids=[local_rank], output_device=local_rank)` around the model -- here around the HIP model, whose whole trunk is ONE
    autograd.Function: DDP's per-parameter hooks must still fire for every trainable parameter and leave the rank average in
    `.grad`.  gloo instead of nccl only because RCCL refuses two ranks on one card."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cellsegmentation_amd import synth
    from cellsegmentation_amd.model import resnet as R
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    m = R.MILresnet18()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.float32)
    m.setmode("tile")
    m.set_encoder_grads(True)
    # the reference never toggles upconv5-8 (resnet.py:226-232): they stay trainable but unused in tile mode, which DDP (without
    # find_unused_parameters, as in the reference's call) refuses from the second iteration on -- freeze what the mode does not use
    for name, p in m.named_parameters():
        if name.startswith(("upconv", "seg_out")):
            p.requires_grad_(False)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    xs = [synth.normalise(synth.ihc_tiles(8, 32, 700 + r)).to(dev) for r in range(world)]
    ys = [torch.tensor([(i + r) % 2 for i in range(8)], device=dev) for r in range(world)]

    def local_grads(r):
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(m(xs[r], freeze_bn=True), ys[r]).backward()
        return [p.grad.clone() for p in params]

    expect = [sum(gs) / world for gs in zip(*[local_grads(r) for r in range(world)])]      # single-process gradients, no DDP
    for p in params:
        p.grad = None
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0)
    worst = 0.0
    for step in range(3):
        ddp.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(ddp(xs[rank], freeze_bn=True), ys[rank]).backward()
        for p, e in zip(params, expect):
            assert p.grad is not None
            worst = max(worst, float((p.grad - e).abs().max() / (e.abs().max() + 1e-12)))
    torch.cuda.synchronize()
    q.put((rank, worst, len(params)))
    dist.destroy_process_group()


def test_hip_model_inside_torch_ddp_two_ranks_one_gpu(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for r in res:
        assert len(r) == 3, r[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, worst, n in res:
        assert n > 60                      # the whole ResNet-18 trunk + the tile head
        assert worst < 1e-6, worst
