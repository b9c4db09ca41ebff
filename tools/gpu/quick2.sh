# multigrid parity tests + 32^4 solve time, fused against unfused coarse Schur complement
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_vs_oracle.py tests/test_gpu_solver_variants.py -x -q -m gpu 2>&1 | tail -3 &&
for rep in 1 2; do
echo "fused: $(python3 tools/solve_profile.py 10 1 32 2 2>&1 | tail -1)"
echo "unfused: $(DDAMG_COARSE_SCHUR_UNFUSED=1 python3 tools/solve_profile.py 10 1 32 2 2>&1 | tail -1)"
done
