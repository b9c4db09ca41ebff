# reserved CUs (DDAMG_COMM_CUS): the fine operator alone through three self-exchanged directions, and the rehearsed 8-GPU solve
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/rehearse_profile.py 8 1 1 > /dev/null 2>&1
for c in 24 0 24 0; do
  a=$(DDAMG_COMM_CUS=$c python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange -1,-1,-1,1 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,1))')
  s=$(DDAMG_COMM_CUS=$c python3 tools/rehearse_profile.py 8 1 3 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["solve_s"]*1000,1), round(d["setup_s"],2))')
  echo "DDAMG_COMM_CUS=$c: apply us $a; solve ms, setup s: $s"
done
