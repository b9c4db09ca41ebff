# A/B of the Schwarz kernel on one box: ddalphaamg_amd/libddamg_hip_base.so (another build) against libddamg_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "base:"; DDAMG_HIP_LIBRARY=$GRAFT_REPO_ROOT/ddalphaamg_amd/libddamg_hip_base.so SAP_BENCH_ITERS=${SAP_AB_ITERS:-0,4} python3 tools/sap_bench.py 2>&1 | grep block_iter
echo "new:"; SAP_BENCH_ITERS=${SAP_AB_ITERS:-0,4} python3 tools/sap_bench.py 2>&1 | grep block_iter
done
