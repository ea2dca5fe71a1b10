#!/bin/bash
# usage: tools/pmc_mem.sh <outname> <script.py> [args]  -- memory-pipeline counters (TA / TCP / SQ VMEM) for one tool run (GPU box)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/mem -o mem -- python3 $ROOT/"$@" > $OUT/run_mem.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/sqmem -o sqmem -- python3 $ROOT/"$@" > $OUT/run_sqmem.txt 2>&1 || exit 1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "normalize" not in k and "multiply_ct" not in k and "k_add" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-36s n=%d last=%.5g" % (c, len(v), v[-1]))
PY
