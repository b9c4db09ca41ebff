#!/usr/bin/env python3
"""tools/four_level_check.py -- a four-level hierarchy (16^4 -> 8^4 -> 4^4 -> 2^4, 2^4 blocks, K-cycle on both intermediate levels)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
from bench import near_unit_gauge
L = [16] * 4; V = 16 ** 4
p = api.default_params(); p.num_levels = 4
for mu in range(4):
    for d in range(4):
        p.local_lattice[d][mu] = 16 >> d
        p.block_lattice[d][mu] = 2
for d in range(3):
    p.num_vect[d] = 20 + 4 * d; p.post_smooth_iter[d] = 2; p.block_iter[d] = 4; p.setup_iter[d] = [3, 2, 2][d]
p.restart, p.max_restart, p.tol = 50, 20, 1e-10
p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
p.mixed_precision, p.method = 1, 2
p.m0, p.csw = -0.3, 1.0
p.test_vector_rng, p.rng_seed = 1, 3
ctx = dd.Context(p)
ctx.set_gauge(near_unit_gauge(V, 0.35, 5), anti_pbc=True)
ci = ctx.setup(3)
b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
x, it, cit, rr = ctx.solve(b, 1e-10)
print("four-level:", it, "iterations,", cit, "coarsest iterations, relres", rr, "setup coarse its", ci)
ctx.close()
