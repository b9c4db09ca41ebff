#!/usr/bin/env python3
"""tools/sap_chain.py -- where the cycles of one MinRes step of the fine-level Schwarz kernel (sap_pair_kernel) go.

Needs the diagnostic build (make -C ddalphaamg_amd/csrc diag -> libddamg_hip_diag.so, -DDDAMG_SAP_CHAIN_DIAG): every wavefront
stamps the shader clock (s_memtime) at the segment boundaries of the last MinRes step of its block visit.  This script runs the
smoother at 32^4 (4^4 blocks, block_iter 4, 2 cycles with an iterate = the V-cycle's call), reads the stamps of the last colour
launch back and prints the per-segment table (mean over the workgroups of the middle half of the launch, cycles of the shader
clock) as markdown: profiles/r04_sap_chain.md is this output.

  DDAMG_HIP_LIBRARY=ddalphaamg_amd/libddamg_hip_diag.so python3 tools/sap_chain.py [block_iter]
"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import ddalphaamg_amd as dd
from ddalphaamg_amd import api
from conftest import splitmix_uniform
from solve_bench import near_unit_gauge

bi = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = [32, 32, 32, 32]; V = int(np.prod(L))
lib = api.load_library()
if not hasattr(lib, "ddamg_hip_diag_sap_chain"):
    sys.exit("this library has no chain stamps: build `make -C ddalphaamg_amd/csrc diag` and set DDAMG_HIP_LIBRARY to libddamg_hip_diag.so")
p = api.default_params(); p.num_levels = 2
for mu in range(4):
    p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = L[mu] // 4
p.num_vect[0] = 4; p.block_iter[0] = bi; p.mixed_precision = 1; p.method = 2; p.m0 = -0.3; p.csw = 1.0
ctx = dd.Context(p)
ctx.set_gauge(near_unit_gauge(V, 0.35, 1), True)
eta = ctx.vector(0, 32).upload(splitmix_uniform(V * 24, 3).reshape(V, 12, 2)); phi = ctx.vector(0, 32)
ctx.smoother(phi, eta, 2, True); ctx.sync()
for _ in range(5):
    ctx.smoother(phi, eta, 2, False)
ctx.sync()
ctx.timer_begin()
for _ in range(10):
    ctx.smoother(phi, eta, 2, False)
ms = ctx.timer_end()
nwg = 4096
buf = (ctypes.c_ulonglong * (nwg * 4 * 16))()
lib.ddamg_hip_diag_sap_chain.restype = ctypes.c_int
n = lib.ddamg_hip_diag_sap_chain(buf, nwg)
if n <= 0:
    sys.exit("no stamps")
t = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4, 16)[:n].astype(np.int64)
# middle half of the launch: the first round of workgroups starts on cold caches, the last one runs on a half-empty chip
t = t[n // 4: 3 * n // 4]
ev = t[:, :2, :].reshape(-1, 16); od = t[:, 2:, :].reshape(-1, 16)    # wavefronts 0,1 even sites, 2,3 odd sites (one block per workgroup)


def seg(a, i, j):
    d = a[:, j] - a[:, i]
    return float(np.mean(d)), float(np.percentile(d, 10)), float(np.percentile(d, 90))


rows_even = [("send 1: plain projections written", 0, 1), ("wait at A1", 1, 2), ("send 2: link^H products written (the odd sites read the plain projections and multiply meanwhile)", 2, 11),
             ("wait at A2", 11, 3), ("D_ee product (the odd sites read the finished products, multiply with D_oo^-1, send 1 meanwhile)", 3, 4), ("wait at B1", 4, 5),
             ("collect 1: plain projections read, link products, reconstruction (the odd sites' send 2 meanwhile)", 5, 14), ("wait at B2", 14, 6),
             ("collect 2: finished products read, reconstruction; partial sums <Dr,r>, <Dr,Dr> of the lane", 6, 7), ("wavefront sums (DPP) + scratch write", 7, 8), ("wait at B4", 8, 9),
             ("alpha, iterate and residual update", 9, 10)]
rows_odd = [("wait at A1 (send 1 of the even sites)", 0, 2), ("collect 1: plain projections read, link products, reconstruction", 2, 11), ("wait at A2", 11, 3),
            ("collect 2: finished products read, reconstruction; D_oo^-1 product", 3, 12), ("send 1: plain projections written", 12, 13), ("wait at B1", 13, 5),
            ("send 2: link^H products written", 5, 14), ("wait at B2", 14, 6), ("B2 -> B4: parked while the even sites reduce", 6, 9)]
step_even = seg(ev, 0, 10)[0]
print(f"## The MinRes step of a block visit, segment by segment: half hops in three phases\n")
print(f"`tools/sap_chain.py` on the diagnostic build (`-DDDAMG_SAP_CHAIN_DIAG`, `make -C ddalphaamg_amd/csrc diag`): 32^4, 4^4 blocks, block_iter {bi}, "
      f"smoother call with an iterate (2 cycles = 4 colour launches): **{ms / 10 * 1e3:.0f} us per call** with the stamps compiled in.  "
      f"Stamps: `s_memtime` after `s_waitcnt lgkmcnt(0)`, scheduling barriers around them; last MinRes step of every block visit of the last "
      f"colour launch, workgroups {n // 4}..{3 * n // 4} of {n}; cycles of the shader clock, mean (10th .. 90th percentile).\n")
print("| even-site wavefronts (2 of the 4 of a block) | cycles | share of the step |\n|---|---|---|")
for name, i, j in rows_even:
    m, lo, hi = seg(ev, i, j)
    print(f"| {name} | {m:.0f} ({lo:.0f} .. {hi:.0f}) | {m / step_even:.3f} |")
print(f"| **whole step** | **{step_even:.0f}** | 1 |\n")
print("| odd-site wavefronts | cycles | share of the step |\n|---|---|---|")
for name, i, j in rows_odd:
    m, lo, hi = seg(od, i, j)
    print(f"| {name} | {m:.0f} ({lo:.0f} .. {hi:.0f}) | {m / step_even:.3f} |")
# the critical chain: even send 1 -> max(even send 2, odd collect 1) -> odd collect 2 + D_oo^-1 + send 1 -> max(odd send 2, even collect 1) -> even collect 2 + sums -> update
chain = [("even: send 1", seg(ev, 0, 1)[0]), ("max(even: send 2, odd: collect 1)", max(seg(ev, 2, 11)[0], seg(od, 2, 11)[0])), ("odd: collect 2 + D_oo^-1 + send 1", seg(od, 3, 13)[0]),
         ("max(odd: send 2, even: collect 1)", max(seg(od, 5, 14)[0], seg(ev, 5, 14)[0])), ("even: collect 2 + partial sums", seg(ev, 6, 7)[0]), ("even: wavefront sums + scratch", seg(ev, 7, 8)[0]),
         ("even: alpha + update", seg(ev, 9, 10)[0])]
work = sum(c for _, c in chain)
print(f"\nCritical chain (work segments only): " + ", ".join(f"{k} {c:.0f}" for k, c in chain) + f" = {work:.0f} cycles; the five barriers and their skew: "
      f"{step_even - work:.0f} cycles ({(step_even - work) / step_even:.2f} of the step).")
ctx.close()
