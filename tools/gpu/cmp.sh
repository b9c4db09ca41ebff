cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_dirac.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -6
for c in 1 0; do
  echo "== compression $c"; DDAMG_LINK_COMPRESSION=$c python3 bench.py --no-solve --no-strong --no-cpu-baseline --steps 1000 --warmup 200 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['roofline']['us_per_launch'], d['roofline']['frac'], d['value'])"
done
