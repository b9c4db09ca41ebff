cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/full/smoke.log
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/full/tests.log 2>&1; echo "tests rc=$?"
tail -14 gpurun_out/full/tests.log
