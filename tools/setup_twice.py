#!/usr/bin/env python3
"""tools/setup_twice.py -- the same setup twice in one context (64^4 three levels by default), with DDAMG_SETUP_TIMING=1 the
phase times of both on stderr (the library prints accumulated times: subtract).  python3 tools/setup_twice.py [extent [levels]]"""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench, synth  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api  # noqa: E402
ext = int(sys.argv[1]) if len(sys.argv) > 1 else 64
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = bench.amg_params(api, [ext] * 4, levels, 0)
if ext >= 64:
    p.restart, p.max_restart = 10, 100
ctx = dd.Context(p)
ctx.set_gauge(synth.synth_gauge([ext] * 4, bench.GAUGE_EPS, bench.GAUGE_SEED), anti_pbc=True)
ts = []
for i in range(3):
    t0 = time.time(); ctx.setup(p.setup_iter[0]); ctx.sync(); ts.append(time.time() - t0)
    print("setup", i, ts[-1], file=sys.stderr, flush=True)
print(json.dumps({"lattice": ext, "levels": levels, "setup_seconds": ts}))
ctx.close()
