#!/usr/bin/env python3
"""HBM traffic per launch from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py ->
profiles/<round>_traffic.json, read back by bench.py for `roofline.traffic`.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/pmc_fetch -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-timing
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/pmc_write -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-timing
    python tools/collect_traffic.py out/pmc_fetch out/pmc_write profiles/round1_traffic.json

Units and corrections as MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read -- every load of these kernels is 16 B/lane (LDS-DMA pieces
and epilogue operands) -- so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle  # noqa: E402


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(os.path.join(path, "p_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            e = agg[demangle(r["Kernel_Name"]).replace(' ', '')]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}, {k: v[1] for k, v in agg.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    every = len(sys.argv) > 4 and sys.argv[4] == "all"          # other configs (EfficientNet, segmentation): every kernel of the step
    what = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 3 --warmup 1"
    f, fn = per_kernel(fetch_dir, "FETCH_SIZE")
    w, _ = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in f:
        if not every and not k.startswith(("igemm", "wgrad", "conv2_")):
            continue
        rd = f[k] * 1024 * 2.0          # gfx950: FETCH_SIZE = 1/2 of a wide coalesced read
        wr = w.get(k, 0.0) * 1024
        res[k] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                  "launches_sampled": fn[k], "fetch_size_kib_raw": round(f[k], 1), "write_size_kib_raw": round(w.get(k, 0.0), 1)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over " + what,
               "correction": "read side x2 (gfx950 FETCH_SIZE half-count for 16 B/lane streams); KiB -> bytes", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])


if __name__ == "__main__":
    main()
