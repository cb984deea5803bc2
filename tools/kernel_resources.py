#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table for one .hip file (compile-only, no GPU needed).

usage: python tools/kernel_resources.py cellsegmentation_amd/csrc/conv_igemm.hip [name-filter]
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import demangle  # noqa: E402  (knows the bf16 vendor type c++filt does not)

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    body = m.group(1)
    if body.startswith("Function Name:"):
        name = body.split(":", 1)[1].strip()
        dem = demangle(name)
        cur = {"name": dem}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
