#!/usr/bin/env python3
"""tools/three_level_check.py -- the reference's sample.ini case (8^4, 3 levels, K-cycle) on the GPU, printed next to
the reference's own numbers from tests/golden/ref_8x8_3lvl.npz."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
g8 = np.load(os.path.join(REPO, "tests/golden/ref_8x8_dirac.npz")); g3 = np.load(os.path.join(REPO, "tests/golden/ref_8x8_3lvl.npz"))
p = api.default_params(); p.num_levels = 3
for mu in range(4):
    p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 2; p.local_lattice[1][mu] = 4; p.block_lattice[1][mu] = 2; p.local_lattice[2][mu] = 2
p.num_vect[0] = p.num_vect[1] = 28
p.setup_iter[0] = 4; p.setup_iter[1] = 3
p.restart, p.max_restart, p.tol = 50, 20, 1e-10
p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
p.mixed_precision, p.method = 1, 2
p.m0, p.csw = -0.5, 1.0
ctx = dd.Context(p); ctx.set_gauge(g8["gauge"], True)
t0 = time.time(); ctx.setup(4); ctx.sync(); t1 = time.time()
b = np.zeros((4096, 12, 2)); b[..., 0] = 1
ctx.solve(b, 1e-10)
t2 = time.time(); x, it, cit, rr = ctx.solve(b, 1e-10); t3 = time.time()
print(f"GPU : setup {t1-t0:.2f}s solve {t3-t2:.4f}s  iters {it} coarse {cit} ({cit/it:.2f}/it) relres {rr:.3e}")
print("GPU  history:", " ".join(f"{h:.2e}" for h in ctx.residual_history()))
print(f"REF : iters {g3['ones_solve_iters'][0]} coarse {g3['ones_solve_iters'][1]} relres {g3['ones_solve_norm_res'][0]:.3e}  (reference CPU: solve 1.66 s on 1 thread, setup 31.6 s; BASELINE.md)")
print("REF  history:", " ".join(f"{h:.2e}" for h in g3["ref_log_ones_history"]))
