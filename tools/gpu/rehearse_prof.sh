# where the time of the rehearsed 8-GPU per-GPU solve goes: kernel totals of the last solve, self-exchange against plain; then the
# latency levers of the decomposed intermediate level
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rehearse; mkdir -p $O
for sx in 1 0; do
  rocprofv3 --kernel-trace -d $O/t$sx -o t -- python3 tools/rehearse_profile.py 8 $sx 2 > $O/solve$sx.log 2>$O/err$sx.log
  python3 tools/kernel_timeline.py $O/t$sx/t_results.db ${1:-6000} | grep "^#" > $O/agg$sx.txt
  rm -rf $O/t$sx
  tail -1 $O/solve$sx.log; head -40 $O/agg$sx.txt
done
echo "single allreduce:"; DDAMG_SINGLE_ALLREDUCE_ARNOLDI=1 python3 tools/rehearse_profile.py 8 1 3 | tail -1
echo "pipelined:"; DDAMG_PIPELINED_ARNOLDI=1 python3 tools/rehearse_profile.py 8 1 3 | tail -1
echo "default:"; python3 tools/rehearse_profile.py 8 1 3 | tail -1
