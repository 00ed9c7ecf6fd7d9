#!/usr/bin/env python3
"""Instruction mix of one kernel's main loop from hipcc's assembly (-S --cuda-device-only):
    python tools/isamix.py file.s <mangled-name-substring>
Finds the biggest loop (label .. backward branch) of the first kernel whose name contains the substring
and prices it with the per-SIMD issue costs measured by tools/issuebench (2 waves per SIMD):
plain VALU 2.6 cycles, packed / cvt_pk 4.45, transcendental 8.3, MFMA 16x16x32 16.1 (MFMA and VALU
do not overlap on one SIMD)."""
import collections
import re
import sys

COST = {"valu": 2.62, "pk": 4.45, "trans": 8.33, "mfma": 16.13}


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_") or op.startswith("v_cvt_pk"):
        return "pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    txt = open(sys.argv[1]).read().splitlines()
    sub = sys.argv[2]
    start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*%s\S*:" % re.escape(sub), l))
    end = next(i for i in range(start, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in txt[i])
    body = txt[start:end]
    # basic blocks of the kernel; a block belongs to a loop when hipcc's comment on its label says so
    # ("=>This Inner Loop Header" / "in Loop: Header=BBx_y"); take the loop with the most instructions
    blocks, cur = [], None
    for i, l in enumerate(body):
        m = re.match(r"^\.LBB(\d+_\d+):(.*)", l)
        if m:
            hdr = re.search(r"Header=BB(\d+_\d+)", m.group(2))
            loop = m.group(1) if "Loop Header" in m.group(2) else (hdr.group(1) if hdr else None)
            cur = {"loop": loop, "lines": []}
            blocks.append(cur)
        elif re.match(r"^; %bb\.\d+:(.*)", l):
            hdr = re.search(r"Header=BB(\d+_\d+)", l)
            cur = {"loop": hdr.group(1) if hdr else None, "lines": []}
            blocks.append(cur)
        elif cur is not None:
            cur["lines"].append(l)
    size = collections.Counter()
    for b in blocks:
        if b["loop"]:
            size[b["loop"]] += sum(1 for l in b["lines"] if re.match(r"^\s+[a-z]", l))
    top = size.most_common(1)[0][0] if size else None
    loop_lines = [l for b in blocks if b["loop"] == top for l in b["lines"]] if top else body
    cnt, ops = collections.Counter(), collections.Counter()
    for l in loop_lines:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m:
            cnt[classify(m.group(1))] += 1
            ops[m.group(1)] += 1
    tot = sum(COST[k] * cnt[k] for k in COST)
    print("kernel lines %d..%d, main loop %d instructions" % (start, end, sum(cnt.values())))
    print("  ".join("%s %d" % kv for kv in sorted(cnt.items())))
    print("issue floor at 2 waves/SIMD: %.0f cycles per trip (valu %.0f + pk %.0f + trans %.0f + mfma %.0f)" % (
        tot, COST["valu"] * cnt["valu"], COST["pk"] * cnt["pk"], COST["trans"] * cnt["trans"], COST["mfma"] * cnt["mfma"]))
    print("top ops:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
