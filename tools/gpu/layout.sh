# operator layout / mass shift kernels: parity tests, then their durations (kernel statistics of the mass-shift trace, 16^4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/layout; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_dirac.py tests/test_gpu_multigrid.py tests/test_gpu_setup_mass.py tests/test_gpu_library_interface.py tests/test_gpu_vs_oracle.py -x -q -m gpu 2>&1 | tail -3 &&
rocprofv3 --kernel-trace --stats -d $O/mass -o m -- python3 tools/mass_shift_trace.py > $O/mass.log 2>> $O/err.log
python3 tools/rocpd_export.py stats $O/mass/m_results.db $O/mass_shift_kernel_stats.csv; rm -rf $O/mass
grep -E "clover_shift|operator_layout|invert_self|shift_self" $O/mass_shift_kernel_stats.csv | cut -d'(' -f1,3- | cut -c1-60,100-
python3 tools/set_gauge_timing.py 2>&1 | tail -4
