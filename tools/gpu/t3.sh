set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t3
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --durations=5 > gpurun_out/t3/tests.log 2>&1 || { tail -60 gpurun_out/t3/tests.log; exit 1; }
tail -12 gpurun_out/t3/tests.log
