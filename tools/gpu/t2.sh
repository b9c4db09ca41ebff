set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t2
python3 tools/solve_profile.py 10 1 32 2 > gpurun_out/t2/s32.log 2>&1; tail -1 gpurun_out/t2/s32.log
DDAMG_READBACK_DMA=1 python3 tools/solve_profile.py 10 1 32 2 > gpurun_out/t2/s32_dma.log 2>&1; tail -1 gpurun_out/t2/s32_dma.log
timeout -k 10 1000 python3 -m pytest tests/test_multi_process.py tests/test_gpu_multigrid.py tests/test_gpu_self_exchange.py tests/test_gpu_library_interface.py -x -q -m gpu --durations=5 > gpurun_out/t2/tests.log 2>&1 || { tail -60 gpurun_out/t2/tests.log; exit 1; }
tail -12 gpurun_out/t2/tests.log
