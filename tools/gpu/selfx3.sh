cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 0 4 8 16 24; do for g in "-1,1,1,1" "-1,-1,-1,1"; do
  echo "comm cus $c grid $g: $(DDAMG_COMM_CUS=$c python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=$g 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us")')"
done; done
