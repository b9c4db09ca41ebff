set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sap3
for nb in 1 2; do
  DDAMG_SAP_BLOCKS_PER_WG=$nb SAP_BENCH_ITERS=0,4 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/sap3/kt$nb -o sap -- python3 tools/sap_bench.py > gpurun_out/sap3/sap_bench_$nb.log 2>&1
  grep block_iter gpurun_out/sap3/sap_bench_$nb.log
  KSEQ_PERIOD=5 python3 tools/kernel_seq.py gpurun_out/sap3/kt$nb/sap_results.db sap_ 100 | tail -6
done
( time timeout -k 10 900 python3 bench.py > gpurun_out/sap3/bench_n1.log 2> gpurun_out/sap3/bench_n1.err ) 2>&1 | grep real
tail -c 3000 gpurun_out/sap3/bench_n1.log; tail -5 gpurun_out/sap3/bench_n1.err
( time timeout -k 10 900 python3 bench.py --gpus 2 --transport host --strong-lattice 16 16 16 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sap3/bench_n2.log 2> gpurun_out/sap3/bench_n2.err ) 2>&1 | grep real
echo rc=$?; tail -c 2500 gpurun_out/sap3/bench_n2.log; tail -5 gpurun_out/sap3/bench_n2.err
