#!/bin/bash
# Builds the diagnostic artefacts of profiles/tools/clock_diag.py (run here, on the CPU; hipcc cross-compiles):
#   profiles/exp/libsfm_hip_stamps.so = the library with -DSFM_CLOCK_STAMPS (clock stamps around the tile loops)
#   profiles/exp/clock_probe          = profiles/microbench/clock_probe.hip
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/profiles/exp
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -shared -fPIC -std=c++17"
/opt/rocm/bin/hipcc $FLAGS -DSFM_CLOCK_STAMPS -o $R/profiles/exp/libsfm_hip_stamps.so $R/structure-from-motion_amd/csrc/*.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/profiles/exp/clock_probe $R/profiles/microbench/clock_probe.hip
ls -la $R/profiles/exp
