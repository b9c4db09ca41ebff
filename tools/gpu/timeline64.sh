cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/timeline; mkdir -p $O
L=${1:-64}; N=${2:-8000}
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/solve_profile.py 2 1 $L 3 > $O/solve$L.log 2>$O/err.log
python3 tools/kernel_timeline.py $O/t/t_results.db $N > $O/timeline$L.txt
grep "^#" $O/timeline$L.txt | head -40; tail -1 $O/solve$L.log; rm -rf $O/t
