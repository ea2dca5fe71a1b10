#!/bin/bash
# usage: tools/lab/pmc_vt.sh <lib.so> <waves>  -- SQ counters of the parked and the pooled NAF kernel (GPU box)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_vt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -o sq -- python3 $ROOT/tools/lab/vt_lab.py 20 $ROOT/$1 $2 > $OUT/run1.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -o sq -- python3 $ROOT/tools/lab/vt_lab.py 20 $ROOT/$1 $2 > $OUT/run2.txt 2>&1 || exit 1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "multiply_vartime" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-22s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
