# collects the round-3 profile artefacts into gpurun_out/prof_r03/ (then: python3 tools/commit_profiles.py gpurun_out/prof_r03 r03)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O; R=r03
# 1. the default bench run under the kernel trace (headline, 32^4 solve leg, 64^4 strong-scaling leg, rehearsal of the 8-GPU point);
#    a warm-up process first: the first process on a fresh box pays one-time allocation costs
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py > $O/bench_line.json 2> $O/bench.err
python3 tools/rocpd_export.py stats $O/bench/bench_results.db $O/${R}_bench_kernel_stats.csv
tail -c 600 $O/bench_line.json; echo
# 2. HBM-side traffic of the fine operator: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_dirac_$c -o p -- python3 bench.py --steps 25 --warmup 5 --no-solve --no-strong --no-cpu-baseline > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_dirac_FETCH_SIZE/p_results.db $O/pmc_dirac_WRITE_SIZE/p_results.db > $O/${R}_pmc_bench.json
# 3. the Schwarz kernel: traffic and issue counters, launch sequence
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$(echo $c | tr ' ' '_' | cut -c1-40)
  SAP_BENCH_ITERS=4 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_sap_$d -o p -- python3 tools/sap_bench.py > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_sap_*/p_results.db > $O/${R}_pmc_sap.json
SAP_BENCH_ITERS=4 rocprofv3 --kernel-trace -d $O/sap_kt -o sap -- python3 tools/sap_bench.py > $O/sap_bench.log 2>> $O/bench.err
KSEQ_PERIOD=5 python3 tools/kernel_seq.py $O/sap_kt/sap_results.db sap_ 100 > $O/${R}_sap_launch_sequence.txt
tail -4 $O/${R}_sap_launch_sequence.txt
# 4. solves: 32^4 two-level, 48^4 and 64^4 three-level; timeline of the 32^4 solve
for cfg in "32 2 10" "48 3 5" "64 3 3"; do set -- $cfg
  rocprofv3 --kernel-trace --stats -d $O/s$1 -o s -- python3 tools/solve_profile.py $3 1 $1 $2 > $O/solve$1.log 2>> $O/bench.err
  python3 tools/rocpd_export.py stats $O/s$1/s_results.db $O/${R}_solve$1_kernel_stats.csv
  tail -1 $O/solve$1.log
done
python3 tools/kernel_timeline.py $O/s32/s_results.db 1500 | grep "^#" > $O/${R}_solve32_timeline_summary.txt
# 5. matrix-core utilisation: the lockstep coarsest-level solves of the bootstrap (32^4 two-level setup) and the Galerkin kernels
#    (48^4 three-level setup)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma32 -o p -- python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>> $O/bench.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma48 -o p -- python3 tools/solve_profile.py 1 1 48 3 > /dev/null 2>> $O/bench.err
python3 tools/rocpd_export.py pmc $O/pmc_mfma32/p_results.db > $O/${R}_pmc_mfma_lockstep32.json
python3 tools/rocpd_export.py pmc $O/pmc_mfma48/p_results.db > $O/${R}_pmc_mfma.json
# 6. a mass change between two solves through the library interface's device path: no operator upload, no Galerkin kernels
rocprofv3 --kernel-trace --stats -d $O/mass -o m -- python3 tools/mass_shift_trace.py > $O/mass.log 2>> $O/bench.err
python3 tools/rocpd_export.py stats $O/mass/m_results.db $O/${R}_mass_shift_kernel_stats.csv
rm -rf $O/*/
ls $O
