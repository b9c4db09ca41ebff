#!/bin/bash
# coarse-level kernels of the solve path: parity tests, 32^4 and 64^4 solve times, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ck
python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_vs_oracle.py -x -q -m gpu 2>&1 | tail -2
python3 tools/solve_profile.py 3 1 32 2 2>/dev/null | cut -c1-170
for cfg in "32 2 5" "64 3 2"; do set -- $cfg
  rocprofv3 --kernel-trace --stats -d gpurun_out/ck -o s$1 -- python3 tools/solve_profile.py $3 1 $1 $2 > gpurun_out/ck/run$1.log 2>&1
  python3 tools/rocpd_export.py stats gpurun_out/ck/s$1_results.db gpurun_out/ck/stats$1.csv; rm -f gpurun_out/ck/s$1_results.db
  tail -1 gpurun_out/ck/run$1.log | cut -c1-170
  grep -E "coarse_site_kernel|coarse_block_minres|coarse_apply_once|restrict_kernel|interpolate_kernel|ls_hop|ls_self" gpurun_out/ck/stats$1.csv | sed 's/(float\*.*)"/"/' | cut -c1-140
done
