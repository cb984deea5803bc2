"""GPU: the data-parallel gradient exchange on a REAL RCCL communicator (backend "nccl"), one rank.

A world of one is all a 1-GPU box offers, but it is a genuine RCCL communicator: `dist.all_reduce` launches RCCL's kernel on
the reducer's side stream, so the bucket hand-off from inside the HIP backward, the stream ordering against the collective and
the guard against a second backward are exercised on the transport bench.py --gpus N uses (the reference's counterpart is the
DDP/NCCL stub of train_tile.py:227-238).  `GradReducer(force_collectives=True)` switches off the world == 1 short cuts."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    try:
        q.put(_body(port))
    except BaseException:       # a silent child would leave the parent waiting on the queue
        import traceback
        q.put(traceback.format_exc())
        raise


def _body(port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from cellsegmentation_amd import synth
    from cellsegmentation_amd.model import resnet as R
    from cellsegmentation_amd.parallel import GradReducer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"

    m = R.MILresnet18()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.float32)
    m.setmode("tile")
    m.set_encoder_grads(True)                  # --scratch: the whole trunk trains
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    x = synth.normalise(synth.ihc_tiles(8, 32, 500)).to(dev)
    y = torch.tensor([i % 2 for i in range(8)], device=dev)

    def backward():
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(m(x, freeze_bn=True), y).backward()

    backward()                                 # no reducer involved
    expect = [p.grad.clone() if p.grad is not None else None for p in params]

    red = GradReducer(params, bucket_bytes=4 << 20, force_collectives=True).attach()
    red.time_collectives = True
    red.broadcast_parameters(m)                # RCCL broadcast, one rank
    early, equal, side_stream = [], True, True
    for step in range(3):
        backward()
        side_stream &= red._stream is None or red._stream != torch.cuda.current_stream()
        red.reduce()
        early.append(red.launches_in_backward)
        for p, e in zip(params, expect):
            if e is None:
                equal &= p.grad is None or not bool(p.grad.abs().max() > 0)
            else:
                equal &= bool(torch.equal(p.grad, e))       # SUM over one rank, x 1/1: bit for bit
    times = red.collective_times()
    # a second backward() without reduce() must raise (the buckets of the first are already on the wire)
    backward()
    raised = False
    try:
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(m(x, freeze_bn=True), y).backward()
    except RuntimeError as e:
        raised = "second backward" in str(e)
    red.detach()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    return {"early": early, "equal": equal, "buckets": len(red.buckets), "rest_buckets": red.n_rest_buckets, "raised": raised, "side_stream": side_stream,
            "n_timed": len(times), "ms": [round(t, 4) for _, t in times]}


def test_grad_reducer_on_a_one_rank_rccl_communicator(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert isinstance(res, dict), res
    assert p.exitcode == 0
    nb = res["buckets"]
    assert nb >= 3
    assert res["equal"], "gradients after the RCCL all-reduce differ from the single-process gradients"
    assert res["early"][0] == 0                                       # the first step records the order
    engine_buckets = nb - res["rest_buckets"]                         # (parameters the engine never reports can only go in reduce())
    assert res["early"][1] >= engine_buckets and res["early"][2] >= engine_buckets    # then the buckets leave from inside backward
    assert res["side_stream"]
    assert res["raised"], "a second backward() without reduce() did not raise"
    assert res["n_timed"] >= 2 * nb              # (the first step runs on the provisional layout, with fewer buckets)
    print("RCCL world-1 all-reduce per bucket (ms):", res["ms"])
