#!/bin/bash
# what the FIRST process on a freshly started box sees (the driver's bench run is one): bench.py without a warm-up in front
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rehearsal']
print('first process: 32^4 setup', d['solve']['setup_seconds'], 'solve', d['solve']['seconds_per_solve'], '| 64^4 setup', d['strong_scaling']['setup_seconds'], 'solve', d['strong_scaling']['seconds_per_solve'], '| rehearsal setup', r['setup_seconds'], 'solve', r['seconds_per_solve_per_gpu'])"
for e in "32 2" "64 3"; do python3 tools/solve_profile.py 1 1 $e 2>/dev/null | cut -c1-150; done
