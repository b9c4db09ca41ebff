#!/bin/bash
# kernel statistics of a 32^4 setup (Galerkin kernels): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gprof
rocprofv3 --kernel-trace --stats -d gpurun_out/gprof -o s32 -- python3 tools/solve_profile.py 1 1 32 2 > gpurun_out/gprof/run.log 2>&1
f=$(ls gpurun_out/gprof/*kernel_stats.csv gpurun_out/gprof/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    v=list(r.values()); print(v[0][:80].ljust(80), v[1:5])
PY
