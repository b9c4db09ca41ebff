set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/base1
SAP_BENCH_ITERS=0,4 rocprofv3 --kernel-trace -d gpurun_out/base1/kt -o sap -- python3 tools/sap_bench.py > gpurun_out/base1/sap_bench.log 2>&1
DB=$(ls gpurun_out/base1/kt/*/*.db gpurun_out/base1/kt/*.db 2>/dev/null | head -1)
echo DB=$DB
python3 tools/kernel_seq.py $DB sap_site_kernel 80 > gpurun_out/base1/seq.txt 2>&1
cat gpurun_out/base1/sap_bench.log gpurun_out/base1/seq.txt
