#!/usr/bin/env python3
"""tools/rehearse_profile.py N SELFX [SOLVES] -- the per-GPU problem of the N-GPU decomposition of the 64^4 strong-scaling case
(bench.py `rehearsal`) on one GPU: SELFX = 1 through the RCCL self-exchange with the coarsest level gathered, 0 as a plain
periodic lattice.  For rocprofv3 --kernel-trace (tools/kernel_timeline.py) and for timing variants (environment knobs)."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench, synth  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api, dist as ddist  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
selfx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = [64, 64, 64, 64]
grid = ddist.process_grid_for(N)
L = [G[mu] // grid[mu] for mu in range(4)]
V = int(np.prod(L))
p = bench.amg_params(api, L, 3, 0)
p.restart, p.max_restart = 10, 100
if selfx:
    p.gather_coarsest = 1
    for mu in range(4):
        p.process_grid[mu] = -1 if grid[mu] > 1 else 1
ctx = dd.Context(p)
if selfx:
    ctx.comm_init_rccl(api.rccl_unique_id())
ctx.set_gauge(synth.synth_gauge(L, bench.GAUGE_EPS, bench.GAUGE_SEED), anti_pbc=True)
t0 = time.time(); ctx.setup(p.setup_iter[0]); ctx.sync(); t_setup = time.time() - t0
b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
bv = ctx.vector(0, 64).upload(b); xv = ctx.vector(0, 64); del b
ctx.solve_vec(xv, bv, 1e-10)
t0 = time.time()
for _ in range(solves):
    it, cit, rr = ctx.solve_vec(xv, bv, 1e-10)
dt = (time.time() - t0) / solves
print(json.dumps({"local_lattice": L, "self_exchange": selfx, "solve_s": dt, "setup_s": t_setup, "iters": it, "coarse_iters": cit, "relres": rr}))
ctx.close()
