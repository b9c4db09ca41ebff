#!/bin/bash
# 64^4 three-level setup: phase times (second process on the box) and kernel statistics of setup + 1 solve
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s64
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1   # warm the box
DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 64 3 > gpurun_out/s64/a.log 2>&1
DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 64 3 > gpurun_out/s64/b.log 2>&1
grep -E "ddamg setup|lattice" gpurun_out/s64/b.log
rocprofv3 --kernel-trace --stats -d gpurun_out/s64 -o s64 -- python3 tools/solve_profile.py 0 1 64 3 > gpurun_out/s64/run.log 2>&1
python3 tools/rocpd_export.py stats gpurun_out/s64/s64_results.db gpurun_out/s64/s64_stats.csv
rm -f gpurun_out/s64/s64_results.db
head -30 gpurun_out/s64/s64_stats.csv | cut -c1-70,150-420
