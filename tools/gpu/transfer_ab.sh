#!/bin/bash
# restriction / interpolation kernels: ddalphaamg_amd/libddamg_hip_base.so (another build) against libddamg_hip.so on ONE box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tab
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
for rep in 1 2; do for lib in base new; do
  unset DDAMG_HIP_LIBRARY; [ $lib = base ] && export DDAMG_HIP_LIBRARY=$GRAFT_REPO_ROOT/ddalphaamg_amd/libddamg_hip_base.so
  rocprofv3 --kernel-trace --stats -d gpurun_out/tab -o s -- python3 tools/solve_profile.py 5 1 32 2 > gpurun_out/tab/run.log 2>&1
  python3 tools/rocpd_export.py stats gpurun_out/tab/s_results.db gpurun_out/tab/stats.csv; rm -f gpurun_out/tab/s_results.db
  echo "$lib: $(grep -E 'restrict_kernel<float, 1>|interpolate_kernel<float>' gpurun_out/tab/stats.csv | sed 's/(float.*)"/"/' | tr '\n' ' ') $(grep lattice gpurun_out/tab/run.log | cut -c37-70)"
done; done
