# fine operator with the halo exchange of a process grid, the process being its own neighbour (RCCL): 1, 2 and 3 split directions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "1,1,1,1" "-1,1,1,1" "-1,-1,1,1" "-1,-1,-1,1"; do
  echo "grid $g: $(python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=$g 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us")')"
done
O=gpurun_out/selfx; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 bench.py --steps 100 --warmup 20 --no-solve --no-strong --no-cpu-baseline --self-exchange=-1,-1,-1,1 > /dev/null 2>$O/err.log
python3 tools/rocpd_export.py stats $O/t/t_results.db $O/stats.csv; head -12 $O/stats.csv | cut -c1-150
python3 tools/kernel_timeline.py $O/t/t_results.db 60 | head -64
rm -rf $O/t
