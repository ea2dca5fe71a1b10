#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel (short name), mean of each counter over dispatches."""
import csv, sys, collections, re, glob, os
d = sys.argv[1]
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void cheetah::", "").replace("cheetah::", "")
    return n[:70]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    if k.startswith("at::") or k.startswith("__amd") or "synth" in k or "reduce_kernel" in k:
        continue
    print(k)
    for c, v in sorted(rows[k].items()):
        print("    %-24s n=%-3d mean=%.6g  last=%.6g" % (c, len(v), sum(v) / len(v), v[-1]))
