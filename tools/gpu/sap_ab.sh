# A/B of the Schwarz kernel on one box: ddalphaamg_amd/libddamg_hip_base.so (baseline build) against libddamg_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_schwarz_methods.py -x -q -m gpu 2>&1 | tail -3 &&
for rep in 1 2; do
echo "base:"; DDAMG_HIP_LIBRARY=$GRAFT_REPO_ROOT/ddalphaamg_amd/libddamg_hip_base.so SAP_BENCH_ITERS=0,4 python3 tools/sap_bench.py 2>&1 | grep block_iter
echo "new:"; SAP_BENCH_ITERS=0,4 python3 tools/sap_bench.py 2>&1 | grep block_iter
done
