"""GPU: the headline step -- ResNet-50 tile classifier, `--scratch` semantics, loop body train/train.py:29-42 -- captured into one
HIP graph (graphed.GraphedStep + the capturable one-launch Adam) against the same steps enqueued eagerly: parameters, loss and
optimizer state BIT FOR BIT.  bench.py replays this graph for its headline number and interleaves eager, event-timed steps for the
roofline; the equality below is what makes the two kinds of step interchangeable."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import functional as HF  # noqa: E402
from cellsegmentation_amd import synth  # noqa: E402
from cellsegmentation_amd.graphed import GraphedStep  # noqa: E402
from cellsegmentation_amd.model import resnet as R  # noqa: E402
from cellsegmentation_amd.optim import Adam  # noqa: E402


def _build(dev):
    m = R.MILresnet50()
    sd = m.state_dict()
    synth.fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.bfloat16)
    m.setmode("tile")
    m.set_encoder_grads(True)
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    return m, Adam(params, lr=5e-4, weight_decay=1e-4, capturable=True)


def _make_step(m, opt):
    def step(xb, yb):
        opt.zero_grad(set_to_none=True)
        loss = HF.cross_entropy(m(xb, freeze_bn=True), yb, 1.0)
        loss.backward()
        opt.step()
        return loss.detach()
    return step


@pytest.mark.parametrize("bag,size", [(16, 299), (64, 299)])
def test_graphed_resnet50_tile_step_equals_eager_steps_bit_for_bit(bag, size, dev):
    batches = [(synth.normalise(synth.ihc_tiles(bag, size, 100 + i)).to(dev), torch.tensor([(j * 7 + i) % 2 for j in range(bag)], device=dev))
               for i in range(3)]
    m1, o1 = _build(dev)
    eager = _make_step(m1, o1)
    losses1 = []
    for _ in range(2):                                   # GraphedStep's warm-up runs on its example batch
        eager(*batches[0])
    for k in range(4):
        if k == 2:
            o1.param_groups[0]["lr"] = 1e-3              # what a scheduler does between steps
        losses1.append(eager(*batches[k % 3]).clone())
    m2, o2 = _build(dev)
    graphed = GraphedStep(_make_step(m2, o2), batches[0], warmup=2, pre_replay=(o2.sync_hyper,))
    losses2 = []
    for k in range(4):
        if k == 2:
            o2.param_groups[0]["lr"] = 1e-3
        losses2.append(graphed(*batches[k % 3]).clone())
    # ... and eager steps on the graphed model continue the same trajectory (bench.py interleaves them)
    l1, l2 = eager(*batches[1]), _make_step(m2, o2)(*batches[1])
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(losses1, losses2)), (losses1, losses2)
    assert torch.equal(l1, l2)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    stepped = 0
    for p, q in zip(o1.param_groups[0]["params"], o2.param_groups[0]["params"]):
        if not o1.state.get(p):                      # upconv5-8 stay trainable in every mode (resnet.py:226-232) but get no gradient here
            assert not o2.state.get(q)
            continue
        stepped += 1
        assert float(o1.state[p]["step"]) == float(o2.state[q]["step"]) == 7.0
        assert torch.equal(o1.state[p]["exp_avg_sq"], o2.state[q]["exp_avg_sq"])
    assert stepped > 150
