# level-1 Schwarz block solver: parity tests of the three-level hierarchies, then the 64^4 and 48^4 three-level solves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_schwarz_methods.py tests/test_gpu_reference_volumes.py -x -q -m gpu 2>&1 | tail -3 &&
for cfg in "64 3" "48 3"; do set -- $cfg
  echo "$1: $(python3 tools/solve_profile.py 3 1 $1 $2 2>&1 | tail -1 | cut -c1-120)"
done
