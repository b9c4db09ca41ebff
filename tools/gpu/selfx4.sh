cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 0 1; do for g in "-1,1,1,1" "-1,-1,1,1" "-1,-1,-1,1"; do
  if [ $p = 1 ]; then export DDAMG_NO_ROUND_SPLIT=1; else unset DDAMG_NO_ROUND_SPLIT; fi
  echo "no round split $p grid $g: $(python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=$g 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us")')"
done; done
