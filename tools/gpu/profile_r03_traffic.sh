# the parts of profile_r03.sh that depend on fine_op.hip, after a change there: HBM-side traffic of the fine operator, the
# mass-shift trace (then: python3 tools/commit_profiles.py gpurun_out/prof_r03 r03)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O; R=r03
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_dirac_$c -o p -- python3 bench.py --steps 25 --warmup 5 --no-solve --no-strong --no-cpu-baseline > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_dirac_FETCH_SIZE/p_results.db $O/pmc_dirac_WRITE_SIZE/p_results.db > $O/${R}_pmc_bench.json
rocprofv3 --kernel-trace --stats -d $O/mass -o m -- python3 tools/mass_shift_trace.py > $O/mass.log 2>> $O/bench.err
python3 tools/rocpd_export.py stats $O/mass/m_results.db $O/${R}_mass_shift_kernel_stats.csv
rm -rf $O/*/
ls $O | head -30
