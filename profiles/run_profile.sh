#!/bin/bash
# Profiling recipe used for the summaries in this directory (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, one block of counters per run.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 1 --cpu-sample 0 ${BENCH_EXTRA:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o pmc -- python3 $ARGS > $OUT/pmc_grbm.log 2>&1
python3 $R/profiles/summarize.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt | cut -c1-220
