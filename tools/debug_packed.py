#!/usr/bin/env python3
"""Structured debugging of the packed 3x3 kernel: delta weights (tap t, channel identity) -> the output must be a shifted copy of the input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K

dev = torch.device("cuda:0")
BF = torch.bfloat16
N, H, W, C, Kc = 2, 19, 19, 64, 128
g = K.make_geom(N, H, W, C, Kc, 3, 3, 1, 1)
x = torch.arange(N * H * W * C, dtype=torch.float32).reshape(N, H, W, C) % 251
x = x.to(BF).to(dev)
for tap in (4, 0, 8, 5):
    w = torch.zeros((Kc, C, 3, 3))
    for k in range(Kc):
        w[k, k % C, tap // 3, tap % 3] = 1.0
    wk, wc = K.weight_prep(w.to(dev), None, BF, C, Kc, True, True)
    wp = K.pack_conv_weights(g, wk, False)
    y = K.conv_fwd_packed(g, x, wp)
    y0 = K.conv_fwd(g, x, wk)
    torch.cuda.synchronize()
    d = (y.float() - y0.float()).abs()
    bad = d > 0
    print(f"tap {tap}: mismatches {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        per_pix = bad.reshape(-1, Kc).any(1).cpu()
        per_ch = bad.reshape(-1, Kc).any(0).cpu()
        print("  bad pixels:", int(per_pix.sum()), "of", per_pix.numel(), " first bad pixel idx:", per_pix.nonzero()[:10].flatten().tolist())
        print("  bad channels:", per_ch.nonzero().flatten().tolist()[:40])
        idx = bad.reshape(-1, Kc).nonzero()[:8]
        yy, y00 = y.reshape(-1, Kc).float().cpu(), y0.reshape(-1, Kc).float().cpu()
        for m, c in idx.tolist():
            print(f"   m={m} ch={c}: got {yy[m, c]} want {y00[m, c]}")
        good_pix = (~per_pix).nonzero().flatten().tolist()[:20]
        print("  good pixels:", good_pix)

print("---- shift-only probe")
w = torch.zeros((Kc, C, 3, 3))
wk, wc = K.weight_prep(w.to(dev), None, BF, C, Kc, True, True)
wp = K.pack_conv_weights(g, wk, False)
sh = torch.arange(Kc, dtype=torch.float32, device=dev)
y = torch.full((N, H, W, Kc), -7.0, dtype=BF, device=dev)
K.conv_fwd_packed(g, x, wp, shift=sh, out=y)
torch.cuda.synchronize()
yy = y.float().cpu().reshape(-1, Kc)
print("row0:", yy[0, :40].tolist())
print("row5:", yy[5, :40].tolist())
print("untouched (-7) count:", int((yy == -7).sum()), "of", yy.numel())
print("packed weights nonzero check (delta tap4):")
w = torch.zeros((Kc, C, 3, 3))
for k in range(Kc):
    w[k, k % C, 1, 1] = 1.0
wk, wc = K.weight_prep(w.to(dev), None, BF, C, Kc, True, True)
wp = K.pack_conv_weights(g, wk, False)
torch.cuda.synchronize()
wpc = wp.float().cpu().reshape(Kc // 32, 1, 9, 4, 64, 8)
print("nonzeros per tap:", [int((wpc[:, :, t] != 0).sum()) for t in range(9)])
nz = (wpc[0, 0, 4] != 0).nonzero()[:10].tolist()
print("first nonzeros in tile0 tap4 (k16, lane, e):", nz)
