"""Keep-alive list for HIP-graph captures.

A captured graph holds raw addresses.  Tensors allocated DURING a capture live in the graph's private pool, but objects the library
caches across steps -- the one-launch staging buffers of a plan (kernels.StagePack), the staged operands of frozen layers
(engine ConvUnit._cache), the optimizer's device tables -- are ordinary allocations that a later EAGER pass may replace (another
batch shape, another requires_grad pattern), which would free memory a graph still reads and writes.  Whoever uses such an object
while the current stream is capturing calls ``keep(obj)``; the capturing wrapper (graphed.GraphedStep / GraphedGrad) takes the list
over with ``take()`` and holds it for the graph's lifetime."""
import torch

_refs = []


def capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def keep(obj):
    _refs.append(obj)


def take():
    global _refs
    out, _refs = _refs, []
    return out
