# kernel timeline of the last solve of a 32^4 two-level run (idle gaps = host latency)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/timeline; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/solve_profile.py 3 1 32 2 > $O/solve.log 2>$O/err.log
python3 tools/kernel_timeline.py $O/t/t_results.db ${1:-1500} > $O/timeline.txt
tail -30 $O/timeline.txt; tail -1 $O/solve.log; rm -rf $O/t
