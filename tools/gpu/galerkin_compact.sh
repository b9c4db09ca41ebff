#!/bin/bash
# A/B of the Galerkin construction's field kernels: DDAMG_GALERKIN_FULL_FIELDS=1 (five full fields per column, gather kernel),
# DDAMG_AGGREGATE_DIRAC_GATHER=1 (face-compacted fields, gather kernel), default (face-compacted, LDS-tiled kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -3 || exit 1
python3 tools/solve_profile.py 2 1 32 2 > /dev/null 2>&1   # warm the box
for e in 32 64; do
  lv=2; [ $e = 64 ] && lv=3
  for mode in tile gather full tile gather full; do
    unset DDAMG_GALERKIN_FULL_FIELDS DDAMG_AGGREGATE_DIRAC_GATHER
    [ $mode = full ] && export DDAMG_GALERKIN_FULL_FIELDS=1
    [ $mode = gather ] && export DDAMG_AGGREGATE_DIRAC_GATHER=1
    DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 $e $lv > gpurun_out/gc_${e}_${mode}.log 2>&1
    echo "extent $e $mode: $(grep -E 'Galerkin' gpurun_out/gc_${e}_${mode}.log | tr -s ' ' | tr '\n' ';') $(grep lattice gpurun_out/gc_${e}_${mode}.log | cut -c1-200)"
  done
done
