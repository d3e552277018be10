#!/bin/bash
# Profiling recipe used for the summaries in this directory (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats of the default bench command (its secondary measurements launch the translation
# RANSAC, LM, triangulation and matcher kernels, so those are covered too).  Passes 2-6: PMC counters, one block
# of counters per run, on a short run of the same workload (counters are per dispatch).  Afterwards copy
# gpurun_out/prof_<tag>/summary.txt, counters.json and traffic.json to profiles/<tag>_summary.txt, profiles/<tag>_counters.json
# and profiles/<tag>_traffic.json: bench.py reads the per-launch instruction counts of its roofline block from the newest
# counters file and the memory traffic from the newest traffic file.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --cpu-sample 0 --cpu-all-cores 0 ${BENCH_EXTRA:-}"
# counters are device-wide: the PMC passes run one step at a time so that a dispatch is not charged with its neighbour's traffic
SHORT="$R/bench.py --cpu-sample 0 --steps 3 --warmup 1 --no-overlap --cpu-all-cores 0 ${BENCH_EXTRA:-}"
rc=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1 || rc=1
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $SHORT > $OUT/pmc_fetch.log 2>&1 || rc=2; }
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $SHORT > $OUT/pmc_write.log 2>&1 || rc=3; }
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_sq -o pmc -- python3 $SHORT > $OUT/pmc_sq.log 2>&1 || rc=4; }
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $OUT/pmc_mfma -o pmc -- python3 $SHORT > $OUT/pmc_mfma.log 2>&1 || rc=6; }
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o pmc -- python3 $SHORT > $OUT/pmc_grbm.log 2>&1 || rc=5; }
# memory traffic of the scoring kernels, one launch at a time against known byte counts (profiles/tools/traffic_probe.py)
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -o p -- python3 $R/profiles/tools/traffic_probe.py run > $OUT/probe_fetch.log 2>&1 || rc=7; }
[ $rc -eq 0 ] && { timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/probe_write -o p -- python3 $R/profiles/tools/traffic_probe.py run > $OUT/probe_write.log 2>&1 || rc=8; }
[ $rc -eq 0 ] && { timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/probe_trace -o trace -- python3 $R/profiles/tools/traffic_probe.py run > $OUT/probe_trace.log 2>&1 || rc=9; }
[ $rc -eq 0 ] && python3 $R/profiles/tools/traffic_probe.py collect $OUT/probe_fetch $OUT/probe_write $OUT/traffic.json > $OUT/traffic.txt
python3 $R/profiles/summarize.py $OUT $OUT/counters.json > $OUT/summary.txt
find $OUT -name "*.db" -delete
cut -c1-220 $OUT/summary.txt
echo "profile passes status: $rc"
exit $rc
