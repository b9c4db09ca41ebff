#!/bin/bash
# aggregate Gram-Schmidt: columns per pass (DDAMG_GS_COLUMNS = 1, 2, default 4): tests, then 32^4 setup phase times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -2 || exit 1
python3 tools/solve_profile.py 2 1 32 2 > /dev/null 2>&1   # warm the box
for c in 1 2 4 1 2 4; do
  export DDAMG_GS_COLUMNS=$c; [ $c = 4 ] && unset DDAMG_GS_COLUMNS
  DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 32 2 > gpurun_out/gs_$c.log 2>&1
  echo "columns $c: $(grep -E 'aggregate Gram' gpurun_out/gs_$c.log | tr -s ' ') $(grep lattice gpurun_out/gs_$c.log | cut -c1-150)"
done
