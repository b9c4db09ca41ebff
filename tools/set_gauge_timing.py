#!/usr/bin/env python3
"""tools/set_gauge_timing.py -- where ddamg_hip_set_gauge spends its time at 32^4: clover term vs layout upload."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from bench import synth_gauge_random as synth_gauge
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
L = [32] * 4; V = 32 ** 4
p = api.default_params(); p.num_levels = 1
for mu in range(4):
    p.local_lattice[0][mu] = 32; p.block_lattice[0][mu] = 4
p.m0, p.csw = -0.1, 1.0
ctx = dd.Context(p)
U = synth_gauge(V, 1)
t0 = time.time(); plaq = ctx.set_gauge(U, True); t1 = time.time()
D, cl = ctx.get_operator()
t2 = time.time(); ctx.set_operator(D, cl); t3 = time.time()
print(f"set_gauge {t1 - t0:.2f} s (plaquette {plaq:.6f}); set_operator (layout conversion + upload only) {t3 - t2:.2f} s")
ctx.close()
