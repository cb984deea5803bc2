#!/usr/bin/env python3
"""Audit the device assembly of the kernels whose inline asm OWNS a block of VGPRs by number (conv_v2.hip, wgrad_v2.hip).

Those main loops name v[LIMIT:255] literally and hipcc is capped below LIMIT with `amdgpu_num_vgpr`; correctness depends on the
compiler (a) never spilling (scratch, .private_segment_fixed_size, .vgpr_spill_count) and (b) never emitting an instruction of
its own (outside ;;#ASMSTART / ;;#ASMEND) that names an owned register.  `make` writes the .s of those sources into
csrc/build/ (-save-temps=obj: the very object that is linked); __graft_entry__.build() and tests/test_asm_audit.py run
`audit_build()` over them and fail on any finding.

CLI:  audit_asm.py                      audit csrc/build/*.s with RULES
      audit_asm.py file.s PATTERN LIMIT one ad-hoc kernel-name pattern"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "cellsegmentation_amd", "csrc", "build")

# source stem -> [(kernel-name substring, first owned VGPR)]; the limits are 2 x the amdgpu_num_vgpr of the kernel's declaration
RULES = {
    "conv_v2": [("conv2_halo_kernel", 100), ("conv2_ring_kernel", 96), ("conv2_wide_kernel", 100)],
    "wgrad_v2": [("wgrad2_kernel", 82)],
}


AGPR_OWNED = 128      # first accumulation register an owned-register kernel may name (conv2_wide_kernel: a[128:255])


def device_asm(stem):
    return os.path.join(BUILD, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")


def audit_text(text, rules):
    """-> {kernel: {"mfma", "scratch", "bad": [(line, text)], "private", "spill", "limit"}} for kernels matching a rule."""
    lines = text.split("\n")
    name, limit, inasm = None, None, False
    stats = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = None
            for pat, lim in rules:
                if pat in m.group(1):
                    name, limit = m.group(1), lim
                    stats[name] = dict(bad=[], scratch=0, mfma=0, private=None, spill=None, limit=lim)
            continue
        if name is None:
            continue
        if ".end_amdhsa_kernel" in l:
            name = None
            continue
        if "#ASMSTART" in l:
            inasm = True
            continue
        if "#ASMEND" in l:
            inasm = False
            continue
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        st = stats[name]
        op = t.split()[0]
        if "v_mfma" in op:
            st["mfma"] += 1
        if op.startswith("scratch_"):
            st["scratch"] += 1
        if not inasm:
            for mm in re.finditer(r"\bv\[?(\d+)(?::(\d+))?\]?", t):
                if int(mm.group(2) or mm.group(1)) >= limit:
                    st["bad"].append((i + 1, t))
                    break
            # accumulation registers: the wide kernel owns a[128:255]; hipcc may use LOW AGPRs as spill space for its own values
            for mm in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", t):
                if int(mm.group(2) or mm.group(1)) >= AGPR_OWNED:
                    st["bad"].append((i + 1, t))
                    break
    # kernel descriptors / metadata: private segment and spill counts
    cur = None
    for l in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            cur = m.group(1) if m.group(1) in stats else None
        elif cur and ".amdhsa_private_segment_fixed_size" in l:
            stats[cur]["private"] = int(l.split()[-1])
        m = re.match(r"\s*\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1) if m.group(1) in stats else cur
        m = re.match(r"\s*\.vgpr_spill_count:\s+(\d+)", l)
        if m and cur:
            stats[cur]["spill"] = int(m.group(1))
    return stats


def findings(stats):
    out = []
    for k, st in stats.items():
        if st["scratch"]:
            out.append(f"{k}: {st['scratch']} scratch instructions")
        if st["private"]:
            out.append(f"{k}: private segment {st['private']} bytes")
        if st["spill"]:
            out.append(f"{k}: {st['spill']} spilled VGPRs")
        for ln, t in st["bad"][:5]:
            out.append(f"{k}: compiler instruction names an owned register (>= v{st['limit']}), line {ln}: {t}")
        if st["mfma"] == 0:
            out.append(f"{k}: no MFMA found (rule pattern matched the wrong symbol?)")
    return out


def audit_build(verbose=False):
    """Audit every rule's .s under csrc/build; -> list of finding strings (empty = clean).  A missing .s is a finding."""
    out = []
    for stem, rules in RULES.items():
        path = device_asm(stem)
        if not os.path.exists(path):
            out.append(f"{path}: missing (run `make -C cellsegmentation_amd/csrc`)")
            continue
        deps = [os.path.join(ROOT, "cellsegmentation_amd", "csrc", stem + ".hip"), os.path.join(ROOT, "cellsegmentation_amd", "csrc", "cs_common.h"),
                os.path.join(ROOT, "include", "cellseg_hip.h")]
        stale = [os.path.basename(d) for d in deps if os.path.getmtime(path) < os.path.getmtime(d)]
        if stale:
            out.append(f"{path}: older than {', '.join(stale)} (stale assembly; the Makefile regenerates every owned-register stem together)")
            continue
        stats = audit_text(open(path).read(), rules)
        for pat, _ in rules:
            if not any(pat in k for k in stats):
                out.append(f"{path}: no kernel matches '{pat}'")
        if verbose:
            for k, st in stats.items():
                print(f"{k}: mfma={st['mfma']} scratch={st['scratch']} private={st['private']} spill={st['spill']} "
                      f"compiler_instrs_on_owned_regs={len(st['bad'])} (owned from v{st['limit']})")
        out += findings(stats)
    return out


def main():
    if len(sys.argv) >= 3:
        limit = int(sys.argv[3]) if len(sys.argv) > 3 else 208
        stats = audit_text(open(sys.argv[1]).read(), [(sys.argv[2], limit)])
        for k, st in stats.items():
            print(f"{k}: mfma={st['mfma']} scratch_ops={st['scratch']} compiler_instrs_on_owned_regs={len(st['bad'])}")
        f = [x for x in findings(stats) if "no MFMA" not in x]
    else:
        f = audit_build(verbose=True)
    for x in f:
        print("FINDING:", x)
    sys.exit(1 if f else 0)


if __name__ == "__main__":
    main()
