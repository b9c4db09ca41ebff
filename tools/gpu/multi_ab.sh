#!/bin/bash
# three-level setup with the intermediate level for all test vectors at once (coarse_multi.hip) against one vector at a time
# (DDAMG_BOOTSTRAP_NO_LOCKSTEP): phase times at the given extents (default 48 64), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/multi_ab
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1   # warm the box
for ext in ${@:-48 64}; do
  DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 $ext 3 > gpurun_out/multi_ab/new$ext.log 2>&1
  DDAMG_SETUP_TIMING=1 DDAMG_BOOTSTRAP_NO_LOCKSTEP=1 python3 tools/solve_profile.py 1 1 $ext 3 > gpurun_out/multi_ab/old$ext.log 2>&1
  echo "== $ext^4 batched"; grep -E "ddamg setup|lattice" gpurun_out/multi_ab/new$ext.log
  echo "== $ext^4 one vector at a time"; grep -E "ddamg setup|lattice" gpurun_out/multi_ab/old$ext.log
done
