#!/usr/bin/env python3
"""rocprofv3 `*_kernel_stats.csv` -> markdown table with demangled, shortened kernel names and ms/step.
usage: summarize_profile.py <kernel_stats.csv> <steps_in_run> [title]"""
import csv
import re
import subprocess
import sys


def _pretty_local(n):
    """llvm-cxxfilt does not know the DF16b (bf16) vendor type: decode our own kernel templates by hand."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if not m:
        return None
    ln = int(m.group(1))
    start = m.end()
    name, rest = n[start:start + ln], n[start + ln:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16b", i):
            args.append("bf16"); i += 5
        elif rest[i] == "f":
            args.append("f32"); i += 1
        elif rest[i] == "L":
            j = rest.index("E", i)
            lit = rest[i + 2:j]
            args.append(("true" if lit == "1" else "false") if rest[i + 1] == "b" else lit)
            i = j + 1
        else:
            break
    return f"{name}<{','.join(args)}>"


def demangle(n):
    loc = _pretty_local(n)
    if loc:
        return loc
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        out = n
    out = out.replace("(anonymous namespace)::", "").replace("__bf16", "bf16")
    out = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", out)          # drop the parameter list
    out = re.sub(r"^void ", "", out)
    return out[:110]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"`rocprofv3 --kernel-trace --stats`, {steps} steps in the run (warm-up included); GPU time {tot / 1e6 / steps:.3f} ms/step "
          f"(the run's set-up -- parameter upload: `__amd_rocclr_copyBuffer`, fills -- is averaged over the steps too; no copy runs inside a step)\n")
    print("| kernel | calls/step | avg us | ms/step | % |")
    print("|---|---:|---:|---:|---:|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
        t = float(r["TotalDurationNs"])
        print(f"| `{demangle(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t / 1e6 / steps:.3f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main()
