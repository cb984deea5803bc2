#!/usr/bin/env python3
"""Time individual conv-family launches (HIP events) on chosen ResNet-50 shapes; also the target
program for `rocprofv3 --pmc ...` counter passes.   python tools/conv_microbench.py [shape ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellsegmentation_amd import kernels as K  # noqa: E402

#  name: (N, H, W, Cin, Cout, R, stride, pad)
SHAPES = {
    "l1_3x3": (64, 75, 75, 64, 64, 3, 1, 1),
    "l2_3x3": (64, 38, 38, 128, 128, 3, 1, 1),
    "l3_3x3": (64, 19, 19, 256, 256, 3, 1, 1),
    "l4_3x3": (64, 10, 10, 512, 512, 3, 1, 1),
    "l1_1x1_64_64": (64, 75, 75, 64, 64, 1, 1, 0),
    "l1_1x1_64_256": (64, 75, 75, 64, 256, 1, 1, 0),
    "l2_1x1_256_128": (64, 75, 75, 256, 128, 1, 1, 0),
    "l2_1x1_128_512": (64, 38, 38, 128, 512, 1, 1, 0),
    "l2_1x1_512_128": (64, 38, 38, 512, 128, 1, 1, 0),
    "l3_1x1_256_1024": (64, 19, 19, 256, 1024, 1, 1, 0),
    "l4_1x1_512_2048": (64, 10, 10, 512, 2048, 1, 1, 0),
    "l4_1x1_2048_512": (64, 10, 10, 2048, 512, 1, 1, 0),
    "l1_1x1_256_64": (64, 75, 75, 256, 64, 1, 1, 0),
    "l3_1x1_1024_256": (64, 19, 19, 1024, 256, 1, 1, 0),
    "l3_1x1_512_256": (64, 38, 38, 512, 256, 1, 1, 0),
    "l4_1x1_1024_512": (64, 19, 19, 1024, 512, 1, 1, 0),
    "l3_1x1_s2_512_1024": (64, 38, 38, 512, 1024, 1, 2, 0),
    "l4_1x1_s2_1024_2048": (64, 19, 19, 1024, 2048, 1, 2, 0),
    "l2_3x3_s2": (64, 75, 75, 128, 128, 3, 2, 1),
    "l3_3x3_s2": (64, 38, 38, 256, 256, 3, 2, 1),
    "l4_3x3_s2": (64, 19, 19, 512, 512, 3, 2, 1),
    "dec_3x3_2048_1024": (8, 19, 19, 2048, 1024, 3, 1, 1),
    "dec_3x3_1024_512": (8, 38, 38, 1024, 512, 3, 1, 1),
    "dec_3x3_512_256": (8, 75, 75, 512, 256, 3, 1, 1),
    "dec_3x3_256_128": (8, 150, 150, 256, 128, 3, 1, 1),
    # EfficientNet-B3 1x1 products (expand / project): HBM-bound, odd channel counts
    "eff_exp_24_144_150": (64, 150, 150, 24, 144, 1, 1, 0),
    "eff_prj_144_32_75": (64, 75, 75, 144, 32, 1, 1, 0),
    "eff_exp_32_192_75": (64, 75, 75, 32, 192, 1, 1, 0),
    "eff_prj_288_96_19": (64, 19, 19, 288, 96, 1, 1, 0),
    "eff_exp_96_576_19": (64, 19, 19, 96, 576, 1, 1, 0),
    "eff_exp_232_1392_10": (64, 10, 10, 232, 1392, 1, 1, 0),
}


def main():
    names = sys.argv[1:] or list(SHAPES)
    iters = int(os.environ.get("ITERS", "20"))
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
    if os.environ.get("IGEMM_PATH", "0") != "0":       # A/B flavour only (CELLSEG_LIB_FLAVOUR=ab)
        K.set_igemm_path(int(os.environ["IGEMM_PATH"]))
    for name in names:
        N, H, W, C, Kc, R, s, p = SHAPES[name]
        g = K.make_geom(N, H, W, C, Kc, R, R, s, p)
        x = torch.randn((N, H, W, C), device=dev).to(dt)
        w = torch.randn((Kc, C, R, R), device=dev) * (1.0 / (C * R * R) ** 0.5)
        wk, wc = K.weight_prep(w, None, dt, C, Kc, True, True)
        dy = torch.randn((N, g.P, g.Q, Kc), device=dev).to(dt)
        raw = K.new_wgrad_buffer(g, dev)
        shift = torch.zeros((Kc,), device=dev)
        flops = 2.0 * N * g.P * g.Q * Kc * C * R * R
        res = {}
        cases = [("fwd", lambda: K.conv_fwd(g, x, wk, None, shift, None, K.CS_ACT_RELU)),
                 ("dgrad", lambda: K.conv_dgrad(g, dy, wc, None, x)),
                 ("wgrad", lambda: K.conv_wgrad(g, x, dy, raw))]
        nb = int(os.environ.get("BATCH", "1"))
        cases.append(("wgrad_b", lambda: K.wgrad_batched(g, [x] * nb, [dy] * nb)))
        if os.environ.get("ONLY"):
            cases = [c for c in cases if c[0] in os.environ["ONLY"].split(",")]
        if dt == torch.bfloat16 and K.packed_supported(g, dt, False):
            wpk = K.pack_conv_weights(g, wk, False)
            resid = torch.randn((N, g.P, g.Q, Kc), device=dev).to(dt) if os.environ.get("RESID") else None
            cases.append(("fwd_pk", lambda: K.conv_fwd_packed(g, x, wpk, shift, resid, K.CS_ACT_RELU, want_bits=True)))
            if resid is not None and cases and cases[0][0] == "fwd":
                cases[0] = ("fwd", lambda: K.conv_fwd(g, x, wk, None, shift, resid, K.CS_ACT_RELU, want_bits=True))
        if dt == torch.bfloat16 and K.packed_supported(g, dt, True):
            wpd = K.pack_conv_weights(g, wc, True)
            mb = torch.randint(0, 255, (N, H, W, C // 8), dtype=torch.uint8, device=dev)
            if s == 1:
                cases.append(("dgrad_pk", lambda: K.conv_dgrad_packed(g, dy, wpd, None, mb, want_colsum=True)))
            else:
                cases.append(("dgrad_pk", lambda: K.conv_dgrad_packed(g, dy, wpd)))          # compact strided gradient
        for kind, fn in cases:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(iters):
                fn()
            s1.record()
            torch.cuda.synchronize()
            ms = s0.elapsed_time(s1) / iters
            res[kind] = (ms, flops / ms / 1e9)
            if os.environ.get("VARIANT"):
                print(f"    {kind}: {(K._lib.load().cs_last_conv_variant() or b'').decode()}", flush=True)
        io_mb = (x.numel() + dy.numel()) * x.element_size() / 1e6
        print(f"{name:20s} " + "  ".join(f"{k}: {v[0] * 1e3:7.1f} us {v[1]:7.1f} TF" for k, v in res.items()) + f"   in+out {io_mb:6.1f} MB", flush=True)


if __name__ == "__main__":
    main()
