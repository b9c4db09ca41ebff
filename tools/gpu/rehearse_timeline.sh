# kernel timeline of the last solve of the rehearsed 8-GPU problem: busy / idle, per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rtl; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/rehearse_profile.py 8 1 3 > $O/solve.log 2>$O/err.log
python3 tools/kernel_timeline.py $O/t/t_results.db 3400 | grep "^#" > $O/summary.txt
rm -rf $O/t; head -30 $O/summary.txt; tail -1 $O/solve.log | cut -c1-120
