import sys, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R,'tests'))
from conftest import load_golden
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
g = load_golden("ref_8x8_dirac.npz")
for B in ([2,2,2,2],[4,4,4,4]):
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = B[mu]
    p.m0, p.csw = float(g["meta_f64"][0]), float(g["meta_f64"][1])
    ctx = dd.Context(p)
    ctx.set_gauge(g["gauge"], anti_pbc=True)
    for prec in (64, 32):
        x = ctx.vector(0, prec).upload(g["dirac_in"]); y = ctx.vector(0, prec)
        ctx.dirac_apply(y, x)
        out = y.download()
        bad = np.isnan(out)
        print(B, prec, "nan sites", int(bad.any(axis=(1,2)).sum()), "nan comps per dof", bad.any(axis=2).sum(axis=0)[:12])
        xin = x.download(); print("   input nan:", int(np.isnan(xin).sum()))
    ctx.close()
