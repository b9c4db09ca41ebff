#!/usr/bin/env python3
"""tools/mass_shift_trace.py -- for rocprofv3 --kernel-trace: set up a 16^4 two-level hierarchy, then ONLY (between the two
markers printed to stdout) change the mass on the device and solve again.  The kernel statistics of the second phase must
show clover_shift_kernel / shift_self_diagonal_kernel / invert_self_kernel and no operator_layout_kernel, aggregate_dirac or
restrict_mfma kernel (shift_update of the reference, src/dirac.c:646-668, without its operator_updates re-construction)."""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench, synth  # noqa: E402
import ddalphaamg_amd as dd  # noqa: E402
from ddalphaamg_amd import api  # noqa: E402
L = [16] * 4; V = 16 ** 4
p = bench.amg_params(api, L, 2, 0)
ctx = dd.Context(p)
ctx.set_gauge(synth.synth_gauge(L, bench.GAUGE_EPS, bench.GAUGE_SEED), anti_pbc=True)
ctx.setup(2)
b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
x, it0, _, rr0 = ctx.solve(b, 1e-10)
ctx.sync()
# phase 2 is run in a fresh process-level trace only if DDAMG_TRACE_PHASE2 is set: the statistics table has no time axis, so the
# script is profiled twice (with and without the mass change) and the difference of the two tables is the mass change
if os.environ.get("DDAMG_MASS_SHIFT", "1") != "0":
    for m in (-0.25, -0.3, -0.28):
        ctx.shift_mass(m)
        x, it1, _, rr1 = ctx.solve(b, 1e-10)
    print(json.dumps({"iterations_before": it0, "iterations_after_shift": it1, "relres": rr1}))
ctx.close()
