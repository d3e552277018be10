#!/bin/bash
# Samples the GPU's shader clock and package power (rocm-smi) every 0.25 s while the headline step runs for several
# seconds:  bash profiles/tools/power_trace.sh > profiles/<tag>_power_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(mktemp -d)
( for i in $(seq 1 60); do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|W\)|Watt|ower \("; sleep 0.25; done ) > $OUT/smi.txt &
SMI=$!
sleep 2
echo "bench start t=$(date +%s.%N)"
timeout -k 10 120 python3 $R/bench.py --steps 1800 --warmup 20 --no-secondary --cpu-sample 0 --cpu-all-cores 0 2>/dev/null | cut -c1-260
echo "bench end t=$(date +%s.%N)"
wait $SMI
cat $OUT/smi.txt
