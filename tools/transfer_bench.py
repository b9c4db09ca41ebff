#!/usr/bin/env python3
"""tools/transfer_bench.py -- time restriction and interpolation alone (32^4, 4^4 aggregates, Nvec 24)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
L = [32] * 4; V = 32 ** 4
p = api.default_params(); p.num_levels = 2
for mu in range(4):
    p.local_lattice[0][mu] = 32; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 8
p.num_vect[0] = 24; p.mixed_precision = 1; p.method = 2; p.m0 = 0.3; p.csw = 0.0
p.test_vector_rng = 1
ctx = dd.Context(p)
D = np.zeros((V, 36, 2)); cl = np.zeros((V, 42, 2)); cl[:, :12, 0] = 4.3
ctx.set_operator(D, cl)
ctx.setup(0)
f = ctx.vector(0, 32); c = ctx.vector(1, 32)
for name, fn in (("restrict", lambda: ctx.restrict(c, f)), ("interpolate", lambda: ctx.interpolate(f, c, add=True))):
    for _ in range(3):
        fn()
    ctx.sync(); ctx.timer_begin()
    for _ in range(20):
        fn()
    ms = ctx.timer_end()
    print(f"{name}: {ms / 20 * 1e3:.1f} us  ({(96 * 24 + 96) * V / (ms / 20 * 1e-3) / 1e12:.2f} TB/s algorithmic)")
if os.environ.get("TRANSFER_BENCH_INTERLEAVE"):
    # the same launches with other memory touched in between (what a solve does between two restrictions): N GB of fp64 vectors
    others = [ctx.vector(0, 64) for _ in range(int(os.environ["TRANSFER_BENCH_INTERLEAVE"]) * 5)]
    for name, fn in (("restrict", lambda: ctx.restrict(c, f)), ("interpolate", lambda: ctx.interpolate(f, c, add=True))):
        tot = 0.0
        for _ in range(10):
            for a in others:
                if os.environ.get("TRANSFER_BENCH_READ_ONLY"):
                    ctx.vec_dot(a, a)
                else:
                    ctx.vec_axpy(a, a, a, 0.5)
            ctx.sync(); ctx.timer_begin(); fn(); tot += ctx.timer_end()
        print(f"{name} after {len(others) * 0.2:.0f} GB of other traffic: {tot / 10 * 1e3:.1f} us")
ctx.close()
