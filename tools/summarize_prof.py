"""Condense rocprofv3 csv output (kernel stats + PMC passes) into the markdown tables kept under profiles/."""
import collections
import csv
import glob
import sys

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
