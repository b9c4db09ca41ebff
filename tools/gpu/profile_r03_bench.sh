# step 1 of profile_r03.sh alone: the default bench run under the kernel trace (a warm-up process first: the first process on a
# fresh box pays one-time allocation costs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O; R=r03
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py > $O/bench_line.json 2> $O/bench.err
python3 tools/rocpd_export.py stats $O/bench/bench_results.db $O/${R}_bench_kernel_stats.csv
rm -rf $O/*/
tail -c 400 $O/bench_line.json
