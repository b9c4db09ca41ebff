#!/usr/bin/env python3
"""tools/solve_profile.py -- set up once, then run N solves (for rocprofv3 --kernel-trace --stats: the solve kernels
dominate the trace when N is large enough).  python3 tools/solve_profile.py [N [mixed_precision [extent [levels]]]]"""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from bench import near_unit_gauge  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ext = int(sys.argv[3]) if len(sys.argv) > 3 else 32
levels = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L = [ext] * 4; V = ext ** 4
p = api.default_params(); p.num_levels = levels
for mu in range(4):
    p.local_lattice[0][mu] = ext; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = ext // 4
    if levels == 3:
        p.block_lattice[1][mu] = 2; p.local_lattice[2][mu] = ext // 8
p.num_vect[0] = 24; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 4
p.num_vect[1] = 28; p.post_smooth_iter[1] = 2; p.block_iter[1] = 4; p.setup_iter[1] = 2
p.restart, p.max_restart, p.tol = 50, 20, 1e-10
p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
p.mixed_precision, p.method, p.odd_even = mp, 2, 1
p.m0, p.csw = -0.3, 1.0
p.test_vector_rng, p.rng_seed = 1, 20260101
ctx = dd.Context(p)
ctx.set_gauge(near_unit_gauge(V, 0.35, 20260101), anti_pbc=True)
ctx.setup(4)
b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
ctx.solve(b, 1e-10)
t0 = time.time()
for _ in range(N):
    x, it, cit, rr = ctx.solve(b, 1e-10)
dt = (time.time() - t0) / N
print(json.dumps({"solve_s": dt, "iters": it, "coarse_iters": cit, "relres": rr, "mixed_precision": mp}))
ctx.close()
