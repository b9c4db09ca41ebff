#!/bin/bash
# level-1 Schwarz block solver: libddamg_hip_base.so (another build) against libddamg_hip.so on ONE box: 64^4 solve times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -k "three_level" 2>&1 | tail -1
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
for rep in 1 2; do for lib in base new; do
  unset DDAMG_HIP_LIBRARY; [ $lib = base ] && export DDAMG_HIP_LIBRARY=$GRAFT_REPO_ROOT/ddalphaamg_amd/libddamg_hip_base.so
  echo "$lib: $(python3 tools/solve_profile.py 4 1 64 3 2>&1 | tail -1 | cut -c1-110)"
done; done
