cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in 0 2000; do for i in 1 2; do
echo "spin $sp: $(DDAMG_BENCH_SPIN_UP=$sp python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-solve --no-strong --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us wall;", round(d["roofline"]["us_per_launch"],2), "us events; frac", round(d["roofline"]["frac"],4))')"
done; done
