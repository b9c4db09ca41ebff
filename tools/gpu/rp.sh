cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_vs_oracle.py -x -q -m gpu 2>&1 | tail -3 &&
O=gpurun_out/rp; mkdir -p $O &&
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 tools/solve_profile.py 5 1 32 2 > $O/solve.log 2>$O/err.log &&
python3 tools/rocpd_export.py stats $O/t/t_results.db $O/stats.csv &&
grep "restrict_kernel\|interpolate_kernel<" $O/stats.csv | awk -F'",' '{print substr($1,1,50), $2}'; tail -1 $O/solve.log; rm -rf $O/t
