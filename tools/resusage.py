#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (one line per kernel).

    hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> res.txt
    python tools/resusage.py res.txt [substring filter ...]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    filters = sys.argv[2:]
    recs, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            recs.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in recs), capture_output=True,
                           text=True).stdout.splitlines()
    for r, d in zip(recs, names):
        d = d.replace("(anonymous namespace)::", "").replace("void ", "")
        d = re.sub(r"\(.*\)$", "", d)
        if d.startswith("_Z"):      # c++filt gave up: kernel name + its integer / bool template arguments
            m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", d)
            if m:
                k = int(m.group(1))
                rest = d[m.end():]
                args = re.findall(r"L([ib])(\d+)E", rest[k:].split("EEv")[0] + "E")
                d = "%s<%s>" % (rest[:k], ", ".join(a[1] for a in args))
        if filters and not all(f in d for f in filters):
            continue
        print("%-64s V%-3d A%-3d S%-3d scratch %-5d occ %d  lds %d" % (
            d[:64], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
            r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
