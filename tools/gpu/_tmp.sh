cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/seq; mkdir -p $O; cd $R
python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/rehearse_profile.py 8 0 2 > $O/s.log 2>$O/e.log
tail -1 $O/s.log | cut -c1-150
python3 tools/kernel_timeline.py $O/t/t_results.db 3000 > $O/timeline_plain.txt; grep "^#" $O/timeline_plain.txt | head -30
rm -rf $O/t
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/solve_profile.py 2 1 32 2 > $O/s.log 2>$O/e.log
tail -1 $O/s.log | cut -c1-150
python3 tools/kernel_timeline.py $O/t/t_results.db 3000 > $O/timeline_32.txt; grep "^#" $O/timeline_32.txt | head -24
rm -rf $O/t
