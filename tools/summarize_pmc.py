#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel (short name), mean of each counter over dispatches.

    tools/summarize_pmc.py <dir with pmc_*/...> [log2_units_per_ladder_dispatch traffic.json]

With the two extra arguments it also writes the per-unit HBM traffic of the ladder kernels (FETCH_SIZE and
WRITE_SIZE are in KB; FETCH_SIZE is left as reported: the ladder's table reads are 8 bytes per lane, a width
MI355X_MICROARCH.md calls uncalibrated, so the x2 of 16-byte streaming reads is NOT applied)."""
import csv, sys, collections, re, glob, os, json
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
if len(sys.argv) > 3:
    units = 1 << int(sys.argv[2])
    out = {"units_per_dispatch": units, "note": "FETCH_SIZE / WRITE_SIZE (KB) x 1024 / units; FETCH_SIZE as reported (8-byte-per-lane reads, uncalibrated width)"}
    for key, pat in (("k_multiply_ct", "k_multiply_ct<MJac, 0, 0, false>"), ("k_multiply_ct_direct", "k_multiply_ct<MJac, 0, 0, true>"),
                     ("k_mul_by_generator_wide", "k_mul_by_generator_wide<false>")):
        for k in rows:
            if k.startswith(pat) and "FETCH_SIZE" in rows[k] and "WRITE_SIZE" in rows[k]:
                f = sum(rows[k]["FETCH_SIZE"]) / len(rows[k]["FETCH_SIZE"])
                w = sum(rows[k]["WRITE_SIZE"]) / len(rows[k]["WRITE_SIZE"])
                e = {"fetch_bytes_per_unit": f * 1024 / units, "write_bytes_per_unit": w * 1024 / units}
                if "SQ_INSTS_VALU" in rows[k] and "SQ_WAVES" in rows[k]:
                    e["valu_wave_insts_per_wave"] = sum(rows[k]["SQ_INSTS_VALU"]) / sum(rows[k]["SQ_WAVES"])
                out[key] = e
    json.dump(out, open(sys.argv[3], "w"), indent=1)
