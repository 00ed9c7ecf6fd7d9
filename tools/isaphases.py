#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop split at its s_barrier instructions (phases between workgroup barriers), priced with
the per-SIMD issue costs of tools/issuebench at two waves per SIMD (see tools/isamix.py):
    python tools/isaphases.py file.s <mangled-name-substring>"""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isamix import COST, classify  # noqa: E402


def main():
    txt = open(sys.argv[1]).read().splitlines()
    sub = sys.argv[2]
    start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*%s\S*:" % re.escape(sub), l))
    end = next(i for i in range(start, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in txt[i])
    body = txt[start:end]
    phases, cur = [], collections.Counter()
    for l in body:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op == "s_barrier":
            phases.append(cur)
            cur = collections.Counter()
        else:
            cur[classify(op)] += 1
            if op.startswith("scratch_") or (op.startswith("buffer_") and "off" in l and "s[0:3]" in l):
                cur["scratch"] += 1
    phases.append(cur)
    print("%3s %6s %6s %6s %6s %6s %6s %6s %8s" % ("#", "mfma", "valu", "pk", "trans", "lds", "vmem", "salu", "floor"))
    for i, c in enumerate(phases):
        fl = sum(COST[k] * c[k] for k in COST)
        print("%3d %6d %6d %6d %6d %6d %6d %6d %8.0f" % (i, c["mfma"], c["valu"], c["pk"], c["trans"], c["lds"], c["vmem"], c["salu"], fl))


if __name__ == "__main__":
    main()
