cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4 &&
for u in 0 1; do
  if [ $u = 1 ]; then export DDAMG_BOOTSTRAP_UNBATCHED=1; else unset DDAMG_BOOTSTRAP_UNBATCHED; fi
  for cfg in "32 2" "48 3"; do set -- $cfg
    echo "unbatched $u: $(DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 $1 $2 2>&1 | grep -v "^W2026" | grep "bootstrap V\|lattice" | tr '\n' ' ')"
  done
done
