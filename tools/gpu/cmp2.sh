cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_dirac.py tests/test_gpu_multigrid.py tests/test_gpu_schwarz_methods.py tests/test_gpu_vs_oracle.py tests/test_gpu_self_exchange.py -x -q -m gpu 2>&1 | tail -6
for c in 1 0; do
  echo "== compression $c: $(DDAMG_LINK_COMPRESSION=$c SAP_BENCH_ITERS=4 python3 tools/sap_bench.py 2>&1 | grep block_iter)"
  DDAMG_LINK_COMPRESSION=$c python3 tools/solve_profile.py 10 1 32 2 2>&1 | tail -1
done
