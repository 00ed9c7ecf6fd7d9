"""Condense rocprofv3 csv output (kernel stats + PMC passes) into the markdown tables kept under profiles/, and write
<root>/traffic.json: HBM bytes per launch of every trunk kernel, FETCH_SIZE x 2 (gfx950 correction for wide coalesced
reads, MI355X_MICROARCH.md) + WRITE_SIZE, both in the KB rocprofv3 reports.  Copied to profiles/traffic.json, that file
is where bench.py takes `roofline.traffic` from."""
import collections
import csv
import glob
import json
import os
import re
import sys


def kernel_key(name):
    """'k_trunkD<128, 4, 1, 0>(...)' or the mangled 'k_trunk2ILi64ELi1EE...' -> 'trunkD_128' / 'trunk2_64' / ..."""
    m = re.search(r"k_(trunk[0-9A-Za-z_]*?)(?:<|I)(.*)", name)
    if not m:
        return None
    base, rest = m.group(1), m.group(2)
    nums = [int(v) for v in re.findall(r"(?:Li)?(-?\d+)", rest)][:4]
    if base in ("trunk2", "trunk2s", "trunkD"):
        return "%s_%d" % (base, nums[0]) if nums else base
    if base == "trunk16":            # <ACT, NPASS, MODE>: MODE 1 = full reverse sweep, 2 = frozen trunk
        return "trunk16_%s" % {1: "bwd", 2: "head", 0: "fwd"}.get(nums[2] if len(nums) > 2 else 1, "bwd")
    return base

root = sys.argv[1]
print("## kernel trace\n\n| kernel | calls | avg us | % |\n|---|---|---|---|")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("| `%s` | %s | %.1f | %s |" % (r["Name"][20:92] if r["Name"].startswith("_ZN12") else r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("\n## PMC, mean per launch\n\n| kernel | counter | value |\n|---|---|---|")
for d in ("pmc_a", "pmc_b", "pmc_f", "pmc_w"):
    for f in glob.glob(root + "/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float)
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][20:92], r["Counter_Name"])
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
        for k in sorted(acc):
            if "trunk" in k[0]:
                print("| `%s` | %s | %.4g |" % (k[0], k[1], acc[k] / cnt[k]))

fs, ws = {}, {}
for d, dst in (("pmc_f", fs), ("pmc_w", ws)):
    for f in glob.glob(root + "/%s/**/*counter_collection.csv" % d, recursive=True):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Kernel_Name"])
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[k] += float(r["Counter_Value"])
                cnt[k] += 1
        for k in acc:
            dst[k] = acc[k] / cnt[k]
traffic = {k: (2.0 * fs[k] + ws.get(k, 0.0)) * 1e3 for k in fs}
json.dump({"formula": "rocprofv3 PMC FETCH_SIZE*2 + WRITE_SIZE (KB), mean per launch", "source": os.path.basename(root.rstrip("/")),
           "bytes_per_launch": traffic, "fetch_kb": fs, "write_kb": ws}, open(os.path.join(root, "traffic.json"), "w"), indent=1)
