#!/bin/bash
# kernel statistics of a 48^4 three-level setup with the coarse level's Galerkin restriction on the matrix cores / vector units
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/crp
for v in mfma valu; do
  unset DDAMG_COARSE_RESTRICT_VALU; [ $v = valu ] && export DDAMG_COARSE_RESTRICT_VALU=1
  rocprofv3 --kernel-trace --stats -d gpurun_out/crp -o $v -- python3 tools/solve_profile.py 0 1 48 3 > gpurun_out/crp/$v.log 2>&1
  python3 tools/rocpd_export.py stats gpurun_out/crp/${v}_results.db gpurun_out/crp/${v}_stats.csv
  rm -f gpurun_out/crp/${v}_results.db
  echo $v; grep -E "coarse_batch" gpurun_out/crp/${v}_stats.csv | cut -c1-60,200-400
done
