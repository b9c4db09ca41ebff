# iteration counts of the full-size tests (printed), for pinning the assertions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_configs.py tests/test_gpu_reference_volumes.py -q -m gpu -s 2>&1 | grep -E "level|method|passed|failed|Error|assert|iterations" > gpurun_out/bands.log; tail -30 gpurun_out/bands.log
