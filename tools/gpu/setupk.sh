# kernel statistics of the setup at 32^4 (two levels): the Galerkin kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/setupk; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 tools/solve_profile.py 1 1 ${1:-32} ${2:-2} > $O/solve.log 2>$O/err.log
python3 tools/rocpd_export.py stats $O/t/t_results.db $O/stats.csv
grep "restrict_mfma\|aggregate_dirac\|gs_aggregates\|coarse_batch" $O/stats.csv | cut -c1-60,200-400; tail -1 $O/solve.log; rm -rf $O/t
timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -2
