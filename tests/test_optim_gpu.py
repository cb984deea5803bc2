"""cellsegmentation_amd.optim.Adam (one HIP launch, csrc/optim.hip) against torch.optim.Adam on the same parameters and gradients:
the optimizer the reference's drivers construct (train_tile.py:282: lr 5e-4, weight_decay 1e-4)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cellsegmentation_amd import optim as O  # noqa: E402


def _params(dev, shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(dev).requires_grad_() for s in shapes]


@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_adam_matches_torch_over_several_steps(wd, dev):
    shapes = [(64, 3, 7, 7), (64,), (256, 64, 1, 1), (512, 512, 3, 3), (2, 2048), (2,), (17,), (1000003,)] + [(33, 5)] * 400   # > 320 tensors
    ours, ref = _params(dev, shapes, 1), _params(dev, shapes, 1)
    a = O.Adam(ours, lr=5e-4, weight_decay=wd)
    b = torch.optim.Adam(ref, lr=5e-4, weight_decay=wd)
    g = torch.Generator().manual_seed(7)
    for step in range(5):
        for p, q in zip(ours, ref):
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad, q.grad = gr, gr.clone()
        if step == 3:
            a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 1e-3          # what a scheduler does
        a.step()
        b.step()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 5.0
        assert float((sa["state"][k]["exp_avg_sq"] - sb["state"][k]["exp_avg_sq"]).abs().max()) <= 1e-6 * float(sb["state"][k]["exp_avg_sq"].abs().max())
    # state interchange: torch's state into ours, one more identical step
    a2 = O.Adam(ours, lr=5e-4, weight_decay=wd)
    import copy
    a2.load_state_dict(copy.deepcopy(b.state_dict()))       # (load_state_dict may alias same-device tensors: the two optimizers must not share moments)
    for p, q in zip(ours, ref):
        gr = torch.ones_like(p)
        p.grad, q.grad = gr, gr.clone()
    a2.step()
    b.step()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))


def test_adam_skips_parameters_without_gradients_and_unaligned_views(dev):
    flat = torch.randn(4099, device=dev)
    p1 = flat[3:1030].detach().clone().requires_grad_()                      # fresh storage, aligned
    base = torch.randn(5000, device=dev)
    p2 = base[1:4098].detach().requires_grad_()                              # 4-byte-aligned view: the scalar path of the kernel
    p3 = torch.randn(10, device=dev, requires_grad=True)                     # never gets a gradient
    q1, q2 = p1.detach().clone().requires_grad_(), p2.detach().clone().requires_grad_()
    a = O.Adam([p1, p2, p3], lr=1e-2)
    b = torch.optim.Adam([q1, q2], lr=1e-2)
    for _ in range(3):
        for p, q in ((p1, q1), (p2, q2)):
            p.grad = torch.sin(p.detach() * 3)
            q.grad = p.grad.clone()
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert float((p1 - q1).abs().max()) < 2e-6 and float((p2 - q2).abs().max()) < 2e-6
    assert len(a.state[p3]) == 0
    with pytest.raises(ValueError):
        O.Adam([p1], amsgrad=True)


def test_adam_alternating_parameter_sets_at_different_step_counts(dev):
    """The reference's train_alternative (train/train.py:240-268): ONE optimizer, tile steps (encoder + tile head have gradients)
    and image steps (encoder + image heads) in turn, zero_grad(set_to_none=True) in between -- after the first tile step the encoder
    is one step ahead of the image heads, then two, ...  torch.optim.Adam keeps a step count per parameter; so does this class:
    one launch per distinct step count, plans cached per parameter set."""
    shapes = {"enc": [(64, 3, 7, 7), (64,), (128, 64, 3, 3)], "tile": [(2, 128), (2,)], "img": [(7, 128), (7,), (1, 128)]}
    g = torch.Generator().manual_seed(3)
    mk = lambda: {k: [torch.randn(s, generator=torch.Generator().manual_seed(11 + i)).to(dev).requires_grad_() for i, s in enumerate(v)]
                  for k, v in shapes.items()}
    ours, ref = mk(), mk()
    flat = lambda d: d["enc"] + d["tile"] + d["img"]
    a = O.Adam(flat(ours), lr=5e-4, weight_decay=1e-4)
    b = torch.optim.Adam(flat(ref), lr=5e-4, weight_decay=1e-4)
    plans = []
    for step in range(7):
        active = ("enc", "tile") if step % 2 == 0 else ("enc", "img")
        for grp in shapes:
            for p, q in zip(ours[grp], ref[grp]):
                if grp in active:
                    gr = torch.randn(p.shape, generator=g).to(dev)
                    p.grad, q.grad = gr, gr.clone()
                else:
                    p.grad = q.grad = None
        a.step()
        b.step()
        plans.append(len(a._plans))
    torch.cuda.synchronize()
    for p, q in zip(flat(ours), flat(ref)):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
        assert float(a.state[p]["step"]) == float(b.state[q]["step"])
    assert float(a.state[ours["enc"][0]]["step"]) == 7.0 and float(a.state[ours["tile"][0]]["step"]) == 4.0 and float(a.state[ours["img"][0]]["step"]) == 3.0
    assert plans[-1] == plans[3] <= 3, plans          # steady state: no new plan per step (all-equal tile set, tile set one behind, image set)


def test_adam_step_refuses_stream_capture(dev):
    p = torch.randn(1000, device=dev, requires_grad=True)
    a = O.Adam([p], lr=1e-3)
    p.grad = torch.ones_like(p)
    a.step()                                  # warm: the plan exists
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.graph(graph, stream=s):
                a.step()
    torch.cuda.synchronize()


def test_adam_capturable_matches_torch_eagerly(dev):
    """capturable=True: step counts on the device (cs_adam_step_dev), run eagerly -- against torch.optim.Adam over alternating
    parameter sets (per-parameter step counts in ONE launch), a learning-rate change, and state interchange both ways."""
    import copy
    shapes = [(64, 3, 7, 7), (64,), (256, 64, 1, 1), (2, 2048), (17,), (100003,)] + [(33, 5)] * 330          # > 320 tensors: two launches
    ours, ref = _params(dev, shapes, 1), _params(dev, shapes, 1)
    a = O.Adam(ours, lr=5e-4, weight_decay=1e-4, capturable=True)
    b = torch.optim.Adam(ref, lr=5e-4, weight_decay=1e-4)
    g = torch.Generator().manual_seed(7)
    for step in range(6):
        for i, (p, q) in enumerate(zip(ours, ref)):
            if i % 3 == 1 and step % 2 == 1:                  # a third of the tensors skips every other step
                p.grad = q.grad = None
                continue
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad, q.grad = gr, gr.clone()
        if step == 3:
            a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 1e-3
        a.step()
        b.step()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
        st = a.state[p]["step"]
        assert st.is_cuda and st.dtype == torch.float32 and float(st) == float(b.state[q]["step"])
    # ours -> torch (capturable) and torch -> ours, one more identical step each way
    b2 = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4, capturable=True)
    b2.load_state_dict(copy.deepcopy(a.state_dict()))
    a2 = O.Adam(ours, lr=1e-3, weight_decay=1e-4, capturable=True)
    a2.load_state_dict(copy.deepcopy(b.state_dict()))           # host step counts -> device fp32 scalars (Optimizer.load_state_dict)
    for p, q in zip(ours, ref):
        gr = torch.ones_like(p)
        p.grad, q.grad = gr, gr.clone()
    a2.step()
    b2.step()
    torch.cuda.synchronize()
    for p, q in zip(ours, ref):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
        assert float(a2.state[p]["step"]) == float(b2.state[q]["step"])


def test_adam_capturable_replays_as_successive_steps(dev):
    """step() of the capturable form captured into a HIP graph: every replay is the NEXT Adam step (bias corrections from the
    device step counts), a learning-rate change reaches the captured launches through sync_hyper(), and the result equals the same
    optimizer stepped eagerly BIT FOR BIT."""
    shapes = [(128, 64, 3, 3), (128,), (2, 2048), (4099,)]
    ours, eag, ref = _params(dev, shapes, 5), _params(dev, shapes, 5), _params(dev, shapes, 5)
    a = O.Adam(ours, lr=5e-4, weight_decay=1e-4, capturable=True)
    e = O.Adam(eag, lr=5e-4, weight_decay=1e-4, capturable=True)
    b = torch.optim.Adam(ref, lr=5e-4, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(9)
    static = [torch.zeros_like(p) for p in ours]
    for p, s in zip(ours, static):
        p.grad = s

    def feed():
        for s, q, r in zip(static, eag, ref):
            gr = torch.randn(s.shape, generator=gen).to(dev)
            s.copy_(gr)
            q.grad, r.grad = gr.clone(), gr.clone()

    feed()
    a.step(); e.step(); b.step()              # eager first step: state and device tables exist
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a.step()                           # captured, not executed
    torch.cuda.current_stream().wait_stream(side)
    for k in range(5):
        if k == 3:
            for o in (a, e, b):
                o.param_groups[0]["lr"] = 2e-3
            a.sync_hyper()
        feed()
        graph.replay()
        e.step(); b.step()
    torch.cuda.synchronize()
    for p, q, r in zip(ours, eag, ref):
        assert torch.equal(p, q)                                        # graph replay == eager, same kernels
        assert float((p - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max()))
        assert float(a.state[p]["step"]) == 6.0 == float(b.state[r]["step"])
    # a changed lr without sync_hyper() is refused inside a capture (the fill would be baked into the graph)
    a.param_groups[0]["lr"] = 3e-3
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="sync_hyper"):
            with torch.cuda.graph(g2, stream=side):
                a.step()
    torch.cuda.synchronize()
