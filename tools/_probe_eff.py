import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cellsegmentation_amd import synth, functional as HF
from cellsegmentation_amd.model import efficientnet as EN
from cellsegmentation_amd.optim import Adam
dev = torch.device("cuda:0")
for name, ctor, B in (("b0", EN.MILefficientnetB0, 40960),):
    m = ctor(num_classes=2); sd = m.state_dict(); synth.fill_state_dict(sd); m.load_state_dict(sd)
    m = m.to(dev).set_compute_dtype(torch.bfloat16); m.setmode("tile"); m.set_encoder_grads(True); m.train()
    x = synth.normalise(synth.ihc_tiles(64, 32, 7)).repeat(B // 64, 1, 1, 1).contiguous().to(dev)
    y = torch.tensor([(i * 7 + 1) % 2 for i in range(B)], device=dev)
    opt = Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-4)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = HF.cross_entropy(m(x, freeze_bn=True), y, 1.0)
        loss.backward(); opt.step(); return loss.detach()
    for _ in range(2): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    with torch.no_grad():
        m.eval(); p = m(x[:128].contiguous()); m.train()
    print(name, "B=%d 32x32 train step:" % B, round(B / dt), "tiles/s", round(dt * 1e3, 1), "ms loss", float(l), "finite", bool(torch.isfinite(p).all()),
          "mem GB", round(torch.cuda.max_memory_allocated() / 1e9, 1), flush=True)
    del m, opt, x, y; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
