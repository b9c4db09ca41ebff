set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
timeout -k 10 1100 python3 -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_configs.py -x -q -m gpu --durations=8 > gpurun_out/t1/tests.log 2>&1 || { tail -60 gpurun_out/t1/tests.log; exit 1; }
tail -15 gpurun_out/t1/tests.log
