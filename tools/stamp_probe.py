#!/usr/bin/env python3
"""Diagnostic build only (make -C cellsegmentation_amd/csrc DEBUG=1): per-phase s_memtime shares of the packed 3x3 kernel.
   python tools/stamp_probe.py [shape ...]     (shapes of tools/conv_microbench.py)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cellsegmentation_amd import kernels as K, _lib
from conv_microbench import SHAPES

lib = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["l1_3x3", "l2_3x3", "l3_3x3", "l4_3x3"]:
    N, H, W, C, Kc, R, s, p = SHAPES[name]
    g = K.make_geom(N, H, W, C, Kc, R, R, s, p)
    x = torch.randn((N, H, W, C), device=dev).to(torch.bfloat16)
    w = torch.randn((Kc, C, R, R), device=dev) / (C * R * R) ** 0.5
    wk, wc = K.weight_prep(w, None, torch.bfloat16, C, Kc, True, True)
    wp = K.pack_conv_weights(g, wk, False)
    shift = torch.zeros((Kc,), device=dev)
    nblk = 8 * ((N * g.P * g.Q + 127) // 128 + 8) * max(1, Kc // 64) + 4096
    buf = torch.zeros((nblk, 4, 6), dtype=torch.int64, device=dev)
    resid = torch.randn((N, g.P, g.Q, Kc), device=dev).to(torch.bfloat16) if os.environ.get("RESID") else None
    ACT = (301 if os.environ.get("EPI") == "2" else 201) if os.environ.get("EPI") else K.CS_ACT_RELU
    for _ in range(3):
        K.conv_fwd_packed(g, x, wp, shift, resid, K.CS_ACT_RELU, want_bits=True)
    torch.cuda.synchronize()
    lib.cs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    K.conv_fwd_packed(g, x, wp, shift, resid, ACT, want_bits=True)
    torch.cuda.synchronize()
    lib.cs_debug_set_stamp_buffer(None)
    b = buf.cpu().reshape(-1, 6)
    b = b[b[:, 0] != 0].double()
    if float(b[:, 0].max()) == 3.0:           # wide kernel: phase lengths in shader cycles + the 100 MHz counter over the wave's life
        clk = b[:, 4] / b[:, 5] * 0.1
        ntm = int(os.environ.get("CELLSEG_WIDE", "0")) or 0
        print(f"{name}: waves {len(b)}; in-kernel clock med {clk.median():.3f} GHz (p10 {clk.quantile(0.1):.3f}, p90 {clk.quantile(0.9):.3f}); wave life med "
              f"{b[:, 4].median():.0f} cycles = {(b[:, 5].median() / 100):.2f} us; MFMA cycles of the loop = {(C // 64) * R * R * 4 * 32} x TM")
        for n_, col in zip(["index math + first loads + barrier", "main loop", "epilogue + drain"], (1, 2, 3)):
            v = b[:, col]
            print(f"    {n_:36s} med {v.median():8.0f}   p10 {v.quantile(0.1):8.0f}   p90 {v.quantile(0.9):8.0f}")
        continue
    if float(b[:, 0].max()) == 2.0:           # ring kernel: per-phase sums over the workgroup's steps
        packed = b[:, 5].long()
        nt, ns = (packed & 0xffffffff).double(), (packed >> 32).double()
        print(f"{name}: waves {len(b)}, tiles per workgroup med {nt.median():.0f} (min {nt.min():.0f}, max {nt.max():.0f}), steps med {ns.median():.0f}; per STEP (cycles):")
        names_ = ["weights issue + wait for own pieces", "barrier", "DMA issue + non-final MFMA steps", "final step: MFMA + epilogue (per TILE)"]
        if os.environ.get("EPI"):
            names_ = ["non-final MFMA steps", "barrier", "DMA issue", "final step: MFMA + epilogue (per TILE)"]
        dens = (ns, ns, ns, nt)
        if os.environ.get("EPI") == "2":
            names_ = ["epilogue per TILE: residual read + arithmetic", "exchange (LDS write, read back)", "mask / bits / stores", "arm next operands"]
            dens = (nt, nt, nt, nt)
        for n_, col, den in zip(names_, (1, 2, 3, 4), dens):
            v = b[:, col] / den
            print(f"    {n_:44s} med {v.median():8.0f}   p10 {v.quantile(0.1):8.0f}   p90 {v.quantile(0.9):8.0f}")
        tot = (b[:, 1] + b[:, 2] + b[:, 3] + b[:, 4])
        print(f"    total per workgroup med {tot.median():.0f} cycles")
        continue
    d = [b[:, i + 1] - b[:, i] for i in range(5)]
    ncc = C // 64
    ideal = ncc * R * R * 16 * 32
    names = ["index math", "first loads + barrier", "main loop", "epilogue issue", "store drain"]
    print(f"{name}: waves {len(b)}, main loop alone = {ideal} MFMA cycles; total med {(b[:, 5] - b[:, 0]).median():.0f}")
    for n_, v in zip(names, d):
        print(f"    {n_:24s} med {v.median():8.0f}   p10 {v.quantile(0.1):8.0f}   p90 {v.quantile(0.9):8.0f}")
