#!/bin/bash
# usage: tools/pmc_one.sh <outname> <script.py> [args]   -- one SQ counter pass + kernel stats for one tool run (GPU box)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- python3 $ROOT/"$@" > $OUT/run_sq.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o grbm -- python3 $ROOT/"$@" > $OUT/run_grbm.txt 2>&1 || exit 1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "normalize" not in k and "multiply_ct" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-22s n=%d last=%.4g" % (c, len(v), v[-1]))
PY
