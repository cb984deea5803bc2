import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cellsegmentation_amd import functional as HF, synth
from cellsegmentation_amd.model import resnet as R
from cellsegmentation_amd.graphed import GraphedGrad
variant = sys.argv[1]
dev = torch.device("cuda:0")
m = R.MILresnet18(); sd = m.state_dict(); synth.fill_state_dict(sd); m.load_state_dict(sd)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
m = m.to(dev).set_compute_dtype(torch.float32 if "f32" in variant else torch.bfloat16)
m.setmode("tile" if "tile" in variant else "image"); m.train()
if "tile" in variant: m.set_encoder_grads(True)
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-3)
x = synth.normalise(synth.ihc_tiles(4, 64, 1)).to(dev)
yc = torch.tensor([0, 1, 2, 3], device=dev) % (2 if "tile" in variant else 7); yn = torch.tensor([0., 3., 9., 20.], device=dev)
def body(x, yc, yn):
    if "tile" in variant:
        return (HF.cross_entropy(m(x, freeze_bn=True), yc),)
    o = m(x)
    if "ceonly" in variant: return (HF.cross_entropy(o[0], yc),)
    return (HF.cross_entropy(o[0], yc) + 0.5 * HF.mse_loss(o[1].squeeze(), yn),)
def eager():
    opt.zero_grad(); body(x, yc, yn)[0].backward(); opt.step()
if "side" in variant:
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eager()
    torch.cuda.current_stream().wait_stream(s)
else:
    for _ in range(3): eager()
torch.cuda.synchronize()
print(variant, "capturing", flush=True)
g = GraphedGrad(params, body, (x, yc, yn))
print(variant, "captured", flush=True)
for _ in range(3):
    out = g(x, yc, yn); opt.step()
torch.cuda.synchronize()
print(variant, "OK", float(out[0]), flush=True)
