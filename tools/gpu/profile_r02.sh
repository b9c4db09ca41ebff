# collects the round-2 profile artefacts into gpurun_out/prof_r02/ (copied into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O
# 1. the default bench run under the kernel trace
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py > $O/bench_line.json 2> $O/bench.err
python3 tools/rocpd_export.py stats $O/bench/bench_results.db $O/r02_bench_kernel_stats.csv
tail -c 1500 $O/bench_line.json; echo
# 2. HBM-side traffic of the fine operator: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_dirac_$c -o p -- python3 bench.py --steps 25 --warmup 5 --no-solve --no-strong --no-cpu-baseline > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_dirac_FETCH_SIZE/p_results.db $O/pmc_dirac_WRITE_SIZE/p_results.db > $O/r02_pmc_bench.json
# 3. the Schwarz kernel: traffic and issue counters
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$(echo $c | tr ' ' '_' | cut -c1-40)
  SAP_BENCH_ITERS=4 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_sap_$d -o p -- python3 tools/sap_bench.py > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_sap_*/p_results.db > $O/r02_pmc_sap.json
SAP_BENCH_ITERS=4 rocprofv3 --kernel-trace -d $O/sap_kt -o sap -- python3 tools/sap_bench.py > $O/sap_bench.log 2>> $O/bench.err
KSEQ_PERIOD=5 python3 tools/kernel_seq.py $O/sap_kt/sap_results.db sap_ 100 > $O/r02_sap_launch_sequence.txt
cat $O/r02_sap_launch_sequence.txt | tail -6
# 4. solves: 32^4 two-level, 48^4 and 64^4 three-level
for cfg in "32 2 10" "48 3 5" "64 3 3"; do set -- $cfg
  rocprofv3 --kernel-trace --stats -d $O/s$1 -o s -- python3 tools/solve_profile.py $3 1 $1 $2 > $O/solve$1.log 2>> $O/bench.err
  python3 tools/rocpd_export.py stats $O/s$1/s_results.db $O/r02_solve$1_kernel_stats.csv
  tail -1 $O/solve$1.log
done
# 5. matrix-core utilisation of the batched coarse-operator kernels of the setup
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o p -- python3 tools/solve_profile.py 1 1 48 3 > /dev/null 2>> $O/bench.err
python3 tools/rocpd_export.py pmc $O/pmc_mfma/p_results.db > $O/r02_pmc_mfma.json
rm -rf $O/*/ ; ls -la $O
