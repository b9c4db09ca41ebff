#!/bin/bash
# lockstep coarsest-level solves: parity tests, 32^4 setup phases, kernel statistics of the ls_* kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lsq
python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -2
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 32 2 2>&1 | grep -E "bootstrap|lattice" | cut -c1-170
rocprofv3 --kernel-trace --stats -d gpurun_out/lsq -o s -- python3 tools/solve_profile.py 0 1 32 2 > gpurun_out/lsq/run.log 2>&1
python3 tools/rocpd_export.py stats gpurun_out/lsq/s_results.db gpurun_out/lsq/stats.csv; rm -f gpurun_out/lsq/s_results.db
grep -E "ls_|coarse_batch" gpurun_out/lsq/stats.csv | sed 's/"[^"]*"/K/'; grep -E "ls_|coarse_batch" gpurun_out/lsq/stats.csv | cut -c1-60
