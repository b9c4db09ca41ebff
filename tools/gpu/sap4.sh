set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sap4
timeout -k 10 900 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_schwarz_methods.py tests/test_gpu_vs_oracle.py -x -q -m gpu > gpurun_out/sap4/tests.log 2>&1 || { tail -40 gpurun_out/sap4/tests.log; exit 1; }
tail -3 gpurun_out/sap4/tests.log
for nb in 1; do
  DDAMG_SAP_BLOCKS_PER_WG=$nb SAP_BENCH_ITERS=0,4 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/sap4/kt$nb -o sap -- python3 tools/sap_bench.py > gpurun_out/sap4/sap_bench_$nb.log 2>&1
  grep block_iter gpurun_out/sap4/sap_bench_$nb.log
  KSEQ_PERIOD=5 python3 tools/kernel_seq.py gpurun_out/sap4/kt$nb/sap_results.db sap_ 100 | tail -6
done
