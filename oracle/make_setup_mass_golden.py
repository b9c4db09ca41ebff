#!/usr/bin/env python3
"""oracle/make_setup_mass_golden.py -- TEST INFRASTRUCTURE ONLY.

Runs the REAL reference library (oracle/_ref/libddamg_ref_scalar.so) through its own library interface with the setup mass
set apart from the solver mass (dd_alpha_amg_par::setup_m0, src/dd_alpha_amg.c:106,146; src/init.c:326-357), using the host
program tests/mpi/setup_mass_driver.c linked against it (oracle/_ref/setup_mass_driver_ref, `make -C oracle ref_facade`),
and commits what it prints as tests/golden/ref_setup_mass.json.  The GPU test runs the same program linked against
libddamg_hip.so.  Input: the reference's 4^4 sample configuration as stored in tests/golden/ref_4x4.npz.
"""
import json, os, re, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)

INI = """configuration: (links are handed over through dd_alpha_amg_set_conf)
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: 2
number of openmp threads: 1
d0 global lattice: 4 4 4 4
d0 local lattice: 4 4 4 4
d0 block lattice: 2 2 2 2
d0 post smooth iter: 2
d0 block iter: 4
d0 test vectors: 20
d0 setup iter: {setup_iter}
d1 global lattice: 2 2 2 2
d1 local lattice: 2 2 2 2
m0: {m0}
setup m0: {setup_m0}
csw: 1.0
tolerance for relative residual: 1E-10
iterations between restarts: 50
maximum of restarts: 20
coarse grid tolerance: 5E-2
coarse grid iterations: 100
coarse grid restarts: 5
print mode: 1
method: 2
mixed precision: 1
odd even preconditioning: 1
randomize test vectors: 0
"""


def write_inputs(tmp, m0, setup_m0, setup_iter):
    g = np.load(os.path.join(REPO, "tests", "golden", "ref_4x4.npz"))
    U = g["gauge"].copy()
    U[-64:, 0] *= -1.0          # anti-periodic in time: the caller's links carry the sign (src/dd_alpha_amg.c:188-252)
    gauge = os.path.join(tmp, "gauge.bin"); U.astype("<f8").tofile(gauge)
    ini = os.path.join(tmp, "case.ini"); open(ini, "w").write(INI.format(m0=m0, setup_m0=setup_m0, setup_iter=setup_iter))
    return gauge, ini


def parse(out):
    # the residual curves are printed by the library in front of the RESULT line of their solve
    solves, cur = {}, []
    for l in out.splitlines():
        if "approx. rel. res. after" in l:
            cur.append(float(l.split(":")[1].split("|")[0]))
        m = re.match(r"RESULT (solve|second_solve|scaled_solve|after_scaled_solve) iterations (-?\d+) coarse_iterations (\d+) relres ([0-9.eE+-]+)", l)
        if m:
            solves[m.group(1)] = dict(iterations=int(m.group(2)), coarse_iterations=int(m.group(3)), relres=float(m.group(4)), residual_history=cur)
            cur = []
    res = {"residual_history": solves.get("solve", {}).get("residual_history", [])}
    for k in ("scaled_solve", "after_scaled_solve"):
        if k in solves:
            res[k] = solves[k]
    m = re.search(r"RESULT scaled_solution_checksum ([0-9.eE+-]+)", out)
    if m:
        res["scaled_solution_checksum"] = float(m.group(1))
    m = re.search(r"RESULT setup_coarse_iterations (\d+)", out); res["setup_coarse_iterations"] = int(m.group(1)) if m else None
    m = re.search(r"RESULT solve iterations (-?\d+) coarse_iterations (\d+) relres ([0-9.eE+-]+)", out)
    if m:
        res.update(iterations=int(m.group(1)), coarse_iterations=int(m.group(2)), relres=float(m.group(3)))
    m = re.search(r"RESULT plaquette ([0-9.]+)", out); res["plaquette"] = float(m.group(1)) if m else None
    return res


def run(exe, mode, m0, setup_m0, setup_iter, scale=None):
    with tempfile.TemporaryDirectory() as tmp:
        gauge, ini = write_inputs(tmp, m0, setup_m0, setup_iter)
        cmd = [exe, mode, repr(m0), repr(setup_m0), gauge, str(setup_iter)] + ([ini] if mode == "init" else [])
        if scale is not None:
            cmd += ["-", repr(scale[0]), repr(scale[1])]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, timeout=600)
        if r.returncode:
            raise RuntimeError(f"{cmd}: exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
        return parse(r.stdout)


# init path only: the reference's struct path aborts in validate_parameters (uninitialised g.ncycle[], src/init.c:1034,1084;
# see tests/mpi/setup_mass_driver.c), so it cannot produce a fixture
CASES = [("init", -0.5, -0.35, 3), ("init", -0.5, -0.5, 3)]
# ... and a solve with the clover term scaled by parity around it (scale_even, scale_odd of dd_alpha_amg_wilson_solve:
# scale_clover + operator_updates, src/dirac.c:624-644, src/dd_alpha_amg.c:354-373), then an unscaled solve again
SCALED_CASE = ("init", -0.5, -0.5, 3, (1.1, 0.9))

if __name__ == "__main__":
    exe = os.path.join(HERE, "_ref", "setup_mass_driver_ref")
    out = {"program": "tests/mpi/setup_mass_driver.c linked against oracle/_ref/libddamg_ref_scalar.so", "lattice": [4, 4, 4, 4],
           "configuration": "conf/4x4x4x4b6.0000id3n1 (tests/golden/ref_4x4.npz)", "cases": []}
    for mode, m0, sm0, it in CASES:
        res = run(exe, mode, m0, sm0, it)
        res.update(mode=mode, m0=m0, setup_m0=sm0, setup_iter=it)
        print(json.dumps(res)); out["cases"].append(res)
    mode, m0, sm0, it, scale = SCALED_CASE
    res = run(exe, mode, m0, sm0, it, scale)
    res.update(mode=mode, m0=m0, setup_m0=sm0, setup_iter=it, scale_even=scale[0], scale_odd=scale[1])
    print(json.dumps(res)); out["scaled_case"] = res
    json.dump(out, open(os.path.join(REPO, "tests", "golden", "ref_setup_mass.json"), "w"), indent=1)
