#!/usr/bin/env python3
"""tools/pmc_calibrate.py -- known-byte streaming kernels (device vector copy, 16 B/lane, same load path as the
stencil) for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the copy moves exactly `bytes` each way."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
p = api.default_params(); p.num_levels = 1
for mu in range(4):
    p.local_lattice[0][mu] = 32; p.block_lattice[0][mu] = 4
ctx = dd.Context(p)
for prec in (32, 64):
    x = ctx.vector(0, prec); y = ctx.vector(0, prec)
    for _ in range(5):
        ctx.vec_copy(y, x)
    ctx.sync()
    print(f"copy fp{prec}: {32**4 * 24 * prec // 8} bytes read and written per launch")
ctx.close()
