#!/usr/bin/env python3
"""tools/sap_bench.py -- time the SAP smoother kernel alone (32^4, 4^4 blocks) for several block_iter values."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
from conftest import splitmix_uniform
sys.path.insert(0, os.path.join(REPO, "tools"))
from solve_bench import near_unit_gauge
L = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else [32, 32, 32, 32])]
V = int(np.prod(L))
U = near_unit_gauge(V, 0.35, 1)
eta_h = splitmix_uniform(V * 24, 3).reshape(V, 12, 2)
for bi in ([int(x) for x in os.environ['SAP_BENCH_ITERS'].split(',')] if 'SAP_BENCH_ITERS' in os.environ else (0, 1, 2, 4, 8)):
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = L[mu] // 4
    p.num_vect[0] = 4; p.block_iter[0] = bi; p.mixed_precision = 1; p.method = 2; p.m0 = -0.3; p.csw = 1.0
    ctx = dd.Context(p)
    ctx.set_gauge(U, True)
    eta = ctx.vector(0, 32).upload(eta_h); phi = ctx.vector(0, 32)
    ctx.smoother(phi, eta, 2, True); ctx.sync()
    ctx.timer_begin()
    n = 10
    for _ in range(n):
        ctx.smoother(phi, eta, 2, False)   # 4 colour launches (FULLRES x2, NBOUNDARY x2) + copies
    ms = ctx.timer_end()
    print(f"block_iter {bi}: {ms / n * 1e3:.1f} us per smoother call (2 cycles = 4 colour launches)", flush=True)
    ctx.close()
