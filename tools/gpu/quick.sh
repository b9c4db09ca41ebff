# quick check after a kernel change: operator / smoother / solver parity tests, then the headline number and the 32^4 solve
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_dirac.py tests/test_gpu_multigrid.py tests/test_gpu_schwarz_methods.py tests/test_gpu_vs_oracle.py tests/test_gpu_self_exchange.py -x -q -m gpu 2>&1 | tail -4 &&
python3 bench.py --steps 1000 --warmup 200 --no-solve --no-strong --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("apply us", round(d["ms_per_step"]*1000,2), "frac", d["roofline"]["frac"])' &&
echo "$(SAP_BENCH_ITERS=4 python3 tools/sap_bench.py 2>&1 | grep block_iter)" &&
python3 tools/solve_profile.py 10 1 32 2 2>&1 | tail -1
