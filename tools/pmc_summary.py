#!/usr/bin/env python3
"""Average rocprofv3 PMC counter values per kernel: python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [name-filter]"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        e = agg[demangle(r["Kernel_Name"])][r["Counter_Name"]]
        e[0] += float(r["Counter_Value"])
        e[1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(agg):
    if flt and flt not in k:
        continue
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:34s} {v / n:16.1f}   (n={n})")
