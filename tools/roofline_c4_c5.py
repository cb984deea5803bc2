#!/usr/bin/env python3
"""Per-kernel roofline tables of C4 (EfficientNet-B3 tile step) and C5 (ResNet-50 segmentation step) from the committed rocprofv3
kernel statistics and PMC traffic of this round -> profiles/round<N>_roofline_c4_c5.md.

    python tools/roofline_c4_c5.py

Algorithmic bytes / FLOPs are derived here from the layer tables (B3 = width 1.2, depth 1.4; decoder convs of
model/resnet.py:154-164), time and launches from `*_kernel_stats.csv`, `traffic` from the FETCH_SIZE / WRITE_SIZE passes
(tools/collect_traffic.py: read side doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams)."""
import csv
import json
import math
import os
import re
import sys

ROUND = int(__import__("os").environ.get("ROUND", "5"))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_profile import demangle  # noqa: E402

PEAK_HBM, PEAK_MFMA = 8000.0, 2500.0      # GB/s, TFLOP/s (MI355X_MICROARCH.md chip-level parameters)


def stats(path, steps):
    out = {}
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\s+", "", demangle(r["Name"]))
        e = out.setdefault(k, [0.0, 0.0])
        e[0] += int(r["Calls"]) / steps
        e[1] += float(r["TotalDurationNs"]) / 1e6 / steps
    return out


def traffic(path):
    t = json.load(open(path))["kernels"]
    return {re.sub(r"\s+", "", k): v for k, v in t.items()}


def b3_tensors(N=64):
    base = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4),
            (6, 3, 1, 192, 320, 1)]

    def adj(c, w=1.2):
        v = c * w
        n = max(8, int(v + 4) // 8 * 8)
        return n + 8 if n < 0.9 * v else n
    H, cin = 150, adj(32)
    bn = N * H * H * cin * 2
    dw_in = dw_out = 0
    pw = 0                                   # 1x1 convolutions: input + output bytes of one forward pass (expand: e != 1, project, head)
    for (e, k, s, ci, co, l) in base:
        co, l = adj(co), int(math.ceil(l * 1.4))
        for i in range(l):
            st = s if i == 0 else 1
            ce = cin * e
            Ho = (H - 1) // 2 + 1 if st == 2 else H
            if e != 1:
                bn += N * H * H * ce * 2
            bn += N * Ho * Ho * ce * 2 + N * Ho * Ho * co * 2
            dw_in += N * H * H * ce * 2
            dw_out += N * Ho * Ho * ce * 2
            if e != 1:
                pw += N * H * H * (cin + ce) * 2
            pw += N * Ho * Ho * (ce + co) * 2
            H, cin = Ho, co
    bn += N * H * H * cin * 4 * 2
    pw += N * H * H * (cin + 4 * cin) * 2
    return bn, dw_in, dw_out, pw


def find_ms(tab, pat):
    return sum(v[1] for k, v in tab.items() if re.search(pat, k)) or 1e-9


def find(tab, pat):
    return [(k, v) for k, v in tab.items() if re.search(pat, k)]


def line(name, calls, ms, alg_gb=None, alg_tf=None, traf=None):
    cells = [f"`{name}`", f"{calls:.0f}", f"{ms:.3f}"]
    if alg_tf is not None:
        a = alg_tf / (ms * 1e-3) / 1e12
        cells += ["mfma", f"{alg_tf / 1e9:.0f} GFLOP", f"{a:.0f} TFLOP/s", f"{a / PEAK_MFMA:.3f}"]
    elif alg_gb is not None:
        a = alg_gb / (ms * 1e-3) / 1e9
        cells += ["hbm", f"{alg_gb / 1e9:.2f} GB", f"{a:.0f} GB/s", f"{a / PEAK_HBM:.3f}"]
    else:
        cells += ["-", "-", "-", "-"]
    cells.append(f"{traf / 1e9:.2f} GB" if traf else "null")
    return "| " + " | ".join(cells) + " |"


def summed(tab, traf, pat):
    rows = find(tab, pat)
    calls = sum(v[0] for _, v in rows)
    ms = sum(v[1] for _, v in rows)
    tb = 0.0
    for k, v in rows:
        if k in traf:
            tb += traf[k]["hbm_bytes_per_launch"] * v[0]
    return calls, ms, tb


def main():
    P = os.path.join(ROOT, "profiles")
    out = [f"# Round {ROUND} -- per-kernel rooflines of C4 and C5 (`ROUND={ROUND} python tools/roofline_c4_c5.py`)", "",
           "peak: HBM 8000 GB/s (spec; ~6300 achievable), dense bf16 MFMA 2500 TFLOP/s.  `achieved` = algorithmic bytes (or FLOPs) per step / "
           "kernel time per step from the rocprofv3 table; `traffic` = HBM bytes per step from the FETCH_SIZE / WRITE_SIZE passes (read side x2).", ""]
    # ---------------- C4
    tab, tr = stats(os.path.join(P, f"round{ROUND}_bench_kernel_stats.csv"), 25), None
    tab = stats(os.path.join(ROOT, "gpurun_out", "round", "prof_c4", "p_kernel_stats.csv"), 12) if os.path.exists(
        os.path.join(ROOT, "gpurun_out", "round", "prof_c4", "p_kernel_stats.csv")) else {}
    tr = traffic(os.path.join(P, f"round{ROUND}_traffic_c4.json"))
    bn, dw_in, dw_out, pw = b3_tensors()
    total = sum(v[1] for v in tab.values())
    out += [f"## C4: EfficientNet-B3 tile bag 64 bf16, BN train (GPU time {total:.2f} ms/step)", "",
            f"BatchNorm tensors of the 78 BN layers: {bn / 1e9:.2f} GB; depthwise inputs {dw_in / 1e9:.2f} GB, outputs {dw_out / 1e9:.2f} GB per pass.", "",
            "| kernel(s) | launches/step | ms/step | bound | algorithmic | achieved | frac | traffic |", "|---|---:|---:|---|---|---|---:|---|"]
    spec = [("bn_bwd_reduce_kernel<bf16>", r"^bn_bwd_reduce", 2 * bn), ("bn_bwd_apply_kernel<bf16>", r"^bn_bwd_apply", 3 * bn),
            ("bn_apply_kernel<bf16> (finalize folded in: cs_bn_apply_stats)", r"^bn_apply", 2 * bn),
            ("dw_wgrad_strip_kernel<R,ST,TS> (depthwise weight gradient)", r"^dw_wgrad_strip_kernel|^dw_wgrad_kernel", dw_in + dw_out),
            ("dw_conv_strip_kernel<..,STATS> (+ the one layer on dw_fwd_stats_kernel): depthwise forward + BN statistics",
             r"^dw_conv_strip_kernel<.*true,false>$|^dw_fwd_stats|^dw_tile_kernel<.*true,false>$", dw_in + dw_out),
            ("dw_conv_strip_kernel<..,FLIP> + dw_dgrad_s2_kernel: depthwise data gradient",
             r"^dw_conv_strip_kernel<.*false,true>$|^dw_dgrad|^dw_tile_kernel<.*false,true>$", dw_in + dw_out),
            ("sample_rowsum_kernel (SE avg pool + ds)", r"^sample_rowsum_kernel", 3 * dw_out),
            ("se_scale_kernel + se_dx_kernel", r"^se_scale_kernel|^se_dx_kernel", 4 * dw_out)]
    for name, pat, by in spec:
        c, ms, tb = summed(tab, tr, pat)
        if ms:
            out.append(line(name, c, ms, alg_gb=by, traf=tb))
    c, ms, tb = summed(tab, tr, r"^igemm_dma_kernel|^igemm_kernel")
    # 1x1 convolutions of B3 (1.666 GMAC forward per tile; 24-384 channels on one side: 20-190 FLOP/B): HBM-bound by construction --
    # algorithmic bytes = input + output of the forward, dy + dx of the data gradient, x + dy of the weight gradient
    out.append(line("igemm_dma_kernel<..> (1x1 expand / project / head, fwd + dgrad; the stem's 3x3 included in the time)", c, ms, alg_gb=2 * pw, traf=tb))
    c, ms, tb = summed(tab, tr, r"^wgrad_dma_kernel|^wgrad_spec_kernel|^wgrad_kernel")
    out.append(line("wgrad_dma / wgrad_spec (1x1 weight gradients)", c, ms, alg_gb=pw, traf=tb))
    out += ["", f"(1x1 products: {2 * 2 * 1.666e9 * 64 / 1e9:.0f} GFLOP fwd + dgrad per step = "
            f"{2 * 2 * 1.666e9 * 64 / (find_ms(tab, r'^igemm_dma_kernel|^igemm_kernel') * 1e-3) / 1e12:.0f} TFLOP/s: not a matrix-core problem)", ""]
    # ---------------- C5
    for tag, csvname, trname, n, hw, title in (("c5", "prof_c5", f"round{ROUND}_traffic_c5.json", 8, 299, "C5: ResNet-50 segmentation B=8 299x299 bf16 (decoder training)"),
                                              ("c5x", "prof_c5x", f"round{ROUND}_traffic_c5x.json", 4, 512, "C5: the same at 512x512 B=4")):
        pth = os.path.join(ROOT, "gpurun_out", "round", csvname, "p_kernel_stats.csv")
        if not os.path.exists(pth):
            continue
        tab = stats(pth, 9)
        tr = traffic(os.path.join(P, trname))
        total = sum(v[1] for v in tab.values())
        s3, s2, s1, s0 = (hw + 31) // 32 * 2 - 1 if hw == 299 else 32, 0, 0, 0
        sizes = [19, 38, 75, 150] if hw == 299 else [32, 64, 128, 256]
        convs = [(2048, 1024, sizes[0]), (2048, 1024, sizes[0]), (1024, 512, sizes[1]), (1024, 512, sizes[1]), (512, 256, sizes[2]), (512, 256, sizes[2]),
                 (256, 128, sizes[3]), (128, 64, sizes[3])]
        fl = [2.0 * n * s * s * c * k * 9 for c, k, s in convs]
        tot = sum(fl)
        # halo kernel: every forward but the last layer's (64 output channels: first generation) and every data gradient, the two widest
        # layers on 8 x 16-pixel tiles; wgrad2: image widths <= 158 (all eight layers at 299, upconv1-6 at 512)
        halo_fl = 2 * tot - fl[7]
        w2_fl = tot if hw == 299 else sum(fl[:6])
        out += [f"## {title} (GPU time {total:.2f} ms/step)", "",
                f"decoder 3x3 convolutions: {tot / 1e9:.0f} GFLOP forward per step ({sum(fl[6:]) / 1e9:.0f} of them in the two widest layers).", "",
                "| kernel(s) | launches/step | ms/step | bound | algorithmic | achieved | frac | traffic |", "|---|---:|---:|---|---|---|---:|---|"]
        # the second-generation stride-1 3x3 kernels: halo AND (round 4 on) the one-wave-per-SIMD wide kernel in its 9-tap form, which serves
        # the layers whose tiles are all resident at once -- round 4's table priced the halo launches alone against all the FLOPs
        S1_3X3 = r"^conv2_halo_kernel|^conv2_wide_kernel<\d+,\d+,(true|false),9>"
        c, ms, tb = summed(tab, tr, S1_3X3)
        # (the frozen encoder's stride-1 3x3 forwards run on the same kernels: 16 launches, ~0.16 of the forward FLOPs of a 299 x 299 image)
        enc3 = 2.0 * n * 9 * sum(cc * cc * ss * ss * rep for cc, ss, rep in ((128, (hw + 7) // 8, 3), (256, (hw + 15) // 16, 5), (512, (hw + 31) // 32, 2)))
        out.append(line("conv2_halo_kernel<9,3,..> + conv2_wide_kernel<..,9> (decoder forward + data gradient, the encoder's stride-1 3x3 forwards)", c, ms, alg_tf=halo_fl + enc3, traf=tb))
        ch_, msh_, _ = summed(tab, tr, r"^conv2_halo_kernel")
        cw_, msw_, _ = summed(tab, tr, r"^conv2_wide_kernel<\d+,\d+,(true|false),9>")
        out.append(f"| (of which halo: {ch_:.0f} launches, {msh_:.3f} ms; wide: {cw_:.0f} launches, {msw_:.3f} ms) | | | | | | | |")
        c, ms, tb = summed(tab, tr, r"^wgrad2_kernel")
        out.append(line("wgrad2_kernel<..> (decoder weight gradients)", c, ms, alg_tf=w2_fl, traf=tb))
        c1, ms1, tb1 = summed(tab, tr, r"^igemm_dma_kernel<bf16,\d+,\d+,1,")
        out.append(line("igemm_dma_kernel<..,1,..> (first generation: upconv8 forward, the encoder's three strided and three 64-channel 3x3)", c1, ms1, traf=tb1))
        c2, ms2, tb2 = summed(tab, tr, r"^wgrad_dma_kernel|^wgrad_spec_kernel")
        out.append(line("wgrad_dma / wgrad_spec (first generation weight gradients)", c2, ms2, traf=tb2))
        ch, msh, _ = summed(tab, tr, S1_3X3)
        cw, msw, _ = summed(tab, tr, r"^wgrad2_kernel")
        fam_ms = msh + msw + ms1 + ms2
        out += ["", f"3x3 family (all four rows: {3 * sum(fl) / 1e9:.0f} GFLOP decoder fwd + dgrad + wgrad, {enc3 / 1e9:.0f} GFLOP encoder stride-1 forwards; the encoder's three strided 3x3 forwards only in the "
                f"time): {(3 * sum(fl) + enc3) / (fam_ms * 1e-3) / 1e12:.0f} TFLOP/s = {(3 * sum(fl) + enc3) / (fam_ms * 1e-3) / 1e12 / PEAK_MFMA:.3f} of the dense bf16 peak.", ""]
        c, ms, tb = summed(tab, tr, r"^bn_apply|^bn_stats|^bn_bwd")
        out.append(f"BN passes (bn_stats / bn_apply / bn_bwd_*): {c:.0f} launches, {ms:.3f} ms/step, counter traffic {tb / 1e9:.2f} GB/step "
                   f"({tb / (ms * 1e-3) / 1e9:.0f} GB/s).")
        c, ms, tb = summed(tab, tr, r"^adam_multi")
        out.append(f"adam_multi_kernel: {ms:.3f} ms/step, counter traffic {tb / 1e9:.2f} GB/step ({tb / (ms * 1e-3) / 1e9:.0f} GB/s).")
        out.append("")
    open(os.path.join(P, f"round{ROUND}_roofline_c4_c5.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
