# Schwarz kernel: smoother parity tests, then time per smoother call at 32^4 for block_iter 0 and 4 (fixed part / MinRes steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -k "smoother" 2>&1 | tail -3 &&
SAP_BENCH_ITERS=0,4 python3 tools/sap_bench.py 2>&1 | grep block_iter
