#!/usr/bin/env python3
"""tools/solve_profile.py -- set up once, then run N solves on device-resident vectors (for rocprofv3 --kernel-trace --stats:
the solve kernels dominate the trace when N is large enough).  python3 tools/solve_profile.py [N [mixed_precision [extent [levels]]]]
Same hierarchy, gauge generator and right-hand side as bench.py's solve legs (restart 10 at 64^4, as there)."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
if os.environ.get("DDAMG_IMPORT_TORCH"):   # the configuration bench.py runs in: torch's HIP runtime loaded first
    import torch
    torch.cuda.init(); torch.cuda.synchronize()
import bench, synth  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ext = int(sys.argv[3]) if len(sys.argv) > 3 else 32
levels = int(sys.argv[4]) if len(sys.argv) > 4 else 2
V = ext ** 4
p = bench.amg_params(api, [ext] * 4, levels, 0)
p.mixed_precision = mp
if ext >= 64:
    p.restart, p.max_restart = 10, 100
ctx = dd.Context(p)
ctx.set_gauge(synth.synth_gauge([ext] * 4, bench.GAUGE_EPS, bench.GAUGE_SEED), anti_pbc=True)
t0 = time.time(); ctx.setup(p.setup_iter[0]); ctx.sync(); t_setup = time.time() - t0
b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
bv = ctx.vector(0, 64).upload(b); xv = ctx.vector(0, 64); del b
it, cit, rr = ctx.solve_vec(xv, bv, 1e-10)
t0 = time.time()
for _ in range(N):
    it, cit, rr = ctx.solve_vec(xv, bv, 1e-10)
dt = (time.time() - t0) / max(N, 1)   # N = 0: the setup and the one warm-up solve only
if os.environ.get("DDAMG_REPEAT_SETUP"):   # the same setup once more in this context (bench.py's setup_seconds_repeated)
    t0 = time.time(); ctx.setup(p.setup_iter[0]); ctx.sync(); print("repeated setup", time.time() - t0, file=sys.stderr)
print(json.dumps({"lattice": ext, "levels": levels, "solve_s": dt, "setup_s": t_setup, "iters": it, "coarse_iters": cit, "relres": rr, "mixed_precision": mp}))
ctx.close()
