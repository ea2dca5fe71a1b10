#!/bin/bash
# Run on the GPU box: kernel-trace stats of bench.py and of the per-kernel driver, then PMC passes
# (one rocprofv3 run per pass).  usage: tools/run_profiles.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the headline command itself (one timed pass; the CPU leg is not GPU work)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.json || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/tools/prof_driver.py 19 24 || exit 1
pass() { # name counters...
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/tools/prof_driver.py 19 22 || exit 1
}
pass pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU && \
pass pmc_sq2 SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM && \
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT && \
pass pmc_fetch FETCH_SIZE && \
pass pmc_write WRITE_SIZE || exit 1
cd $ROOT
python3 tools/summarize_pmc.py $OUT 19 $OUT/traffic.json > $OUT/pmc_summary.txt
echo profiles done
