#!/usr/bin/env python3
"""tools/realloc_probe.py -- does a 64^4 setup cost more in a process that has already built and released another hierarchy?
(bench.py reports 13.6 s for the 64^4 setup of its strong-scaling leg, tools/solve_profile.py 10.2 s in a fresh process.)"""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench, synth  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api  # noqa: E402


def run(ext, levels):
    p = bench.amg_params(api, [ext] * 4, levels, 0)
    if ext >= 64:
        p.restart, p.max_restart = 10, 100
    ctx = dd.Context(p)
    ctx.set_gauge(synth.synth_gauge([ext] * 4, bench.GAUGE_EPS, bench.GAUGE_SEED), anti_pbc=True)
    t0 = time.time(); ctx.setup(p.setup_iter[0]); ctx.sync(); t = time.time() - t0
    ctx.close()
    return t


order = [int(x) for x in sys.argv[1:]] or [32, 64]
for ext in order:
    print(ext, "setup", round(run(ext, 2 if ext == 32 else 3), 3), flush=True)
