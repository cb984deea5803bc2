#!/usr/bin/env python3
"""Audit a hipcc -save-temps .s: for every kernel whose name contains PATTERN report scratch use and the compiler-emitted
instructions (outside ;;#ASMSTART/;;#ASMEND) that name a VGPR >= LIMIT (registers the inline asm of conv_v2.hip owns).
usage: audit_asm.py file.s PATTERN [LIMIT=208]"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 208
    text = open(path).read().split("\n")
    name, inasm, rc = None, False, 0
    stats = {}
    for i, l in enumerate(text):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1) if pat in m.group(1) else None
            if name:
                stats[name] = dict(bad=[], scratch=0, mfma=0, ops={})
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
            st["ops"][op] = st["ops"].get(op, 0) + 1
            for m in re.finditer(r"\bv\[?(\d+)(?::(\d+))?\]?", t):
                if int(m.group(2) or m.group(1)) >= limit:
                    st["bad"].append((i + 1, t))
                    break
    for k, st in stats.items():
        print(f"{k}: mfma={st['mfma']} scratch_ops={st['scratch']} compiler_instrs_on_owned_regs={len(st['bad'])}")
        for b in st["bad"][:5]:
            print("    ", b)
        if st["bad"] or st["scratch"]:
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
