"""Summarise a rocprofv3 --pmc counter_collection.csv per (kernel, grid): mean of every counter.
  python scripts/pmc_summary.py <counter_collection.csv> [name-substring ...]"""
import collections
import csv
import sys

src, pats = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pats and not any(p in name for p in pats):
            continue
        key = (name.split("(")[0][:70], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, ctr in sorted(agg.items()):
    print(key[0], "blocks", key[1])
    for c, v in sorted(ctr.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
