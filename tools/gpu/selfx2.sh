cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1; do for g in "-1,1,1,1" "-1,-1,-1,1"; do
  if [ $d = 1 ]; then export DDAMG_HALO_DEFER=1; else unset DDAMG_HALO_DEFER; fi
  echo "defer $d grid $g: $(python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=$g 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us", d["config"].get("halo_check_vs_host_transport"))')"
done; done
export DDAMG_HALO_DEFER=1
O=gpurun_out/selfx; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t -- python3 bench.py --steps 100 --warmup 20 --no-solve --no-strong --no-cpu-baseline --self-exchange=-1,-1,-1,1 > /dev/null 2>$O/err.log
python3 tools/kernel_timeline.py $O/t/t_results.db 40 | head -24
rm -rf $O/t
