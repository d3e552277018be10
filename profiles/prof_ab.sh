#!/bin/bash
# kernel trace + two PMC passes of profiles/kernel_ab.py (run on the GPU box through gpurun):  bash profiles/prof_ab.sh <tag> [env assignments]
TAG=${1:-ab}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/profiles/kernel_ab.py 10 > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o pmc -- python3 $R/profiles/kernel_ab.py 3 > $OUT/pmc_sq.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- python3 $R/profiles/kernel_ab.py 3 > $OUT/pmc_mfma.log 2>&1 || exit 3
python3 $R/profiles/summarize.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
cut -c1-200 $OUT/summary.txt
