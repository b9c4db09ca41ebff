#!/bin/bash
# quick check of the Galerkin phase: multigrid tests, then 32^4 setup timing twice and kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gprof
python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -2 || exit 1
python3 tools/solve_profile.py 2 1 32 2 > /dev/null 2>&1   # warm the box
for i in 1 2; do
  DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 32 2 > gpurun_out/gq.log 2>&1
  echo "$(grep -E 'Galerkin' gpurun_out/gq.log | tr -s ' ') $(grep lattice gpurun_out/gq.log | cut -c1-120)"
done
rocprofv3 --kernel-trace --stats -d gpurun_out/gprof -o s32 -- python3 tools/solve_profile.py 1 1 32 2 > gpurun_out/gprof/run.log 2>&1
python3 tools/rocpd_export.py stats gpurun_out/gprof/s32_results.db gpurun_out/gprof/s32_stats.csv
grep -E "restrict_mfma|aggregate_dirac|store_column" gpurun_out/gprof/s32_stats.csv | cut -c1-60,160-400
