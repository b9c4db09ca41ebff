#!/usr/bin/env python3
"""oracle/run_reference_cases.py -- TEST INFRASTRUCTURE ONLY.

Runs the REAL reference's own main program (oracle/_ref/dd_alpha_amg_scalar, or a variant build named in the case) on its
sample configurations with the parameter variants that the dump harness does not cover, and commits what its log says --
iteration counts, residual history, coarse-grid iterations -- to tests/golden/ref_runs.json.  Build container only.

usage: python oracle/run_reference_cases.py [case ...]
"""
import json, os, re, subprocess, sys, tempfile, shutil
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(HERE)
REF = os.environ.get("DDAMG_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden", "ref_runs.json")

INI = """configuration: {conf}
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: {levels}
number of openmp threads: 1
d0 global lattice: {L}
d0 local lattice: {L}
d0 block lattice: {B}
d0 post smooth iter: 2
d0 block iter: 4
d0 test vectors: {nvec}
d0 setup iter: {setup}
{extra}
m0: -0.5
csw: 1.0
tolerance for relative residual: 1E-10
iterations between restarts: 50
maximum of restarts: 20
coarse grid tolerance: 5E-2
coarse grid iterations: 100
coarse grid restarts: 5
print mode: 1
method: {method}
odd even preconditioning: {oe}
mixed precision: {mp}
randomize test vectors: 0
"""
C4 = dict(conf="conf/4x4x4x4b6.0000id3n1", L="4 4 4 4", B="2 2 2 2", levels=2, nvec=20, setup=4, extra="", oe=1, exe="dd_alpha_amg_scalar")
C8 = dict(conf="conf/8x8x8x8b6.0000id3n1", L="8 8 8 8", B="2 2 2 2", levels=2, nvec=20, setup=3, extra="d1 global lattice: 4 4 4 4\nd1 local lattice: 4 4 4 4", oe=1, exe="dd_alpha_amg_scalar")
CASES = {
    # method 5: FGMRES preconditioned by BiCGstab on the odd-even Schur complement, no multigrid (src/init.c:976-979)
    "4x4_m5_mp1": dict(C4, method=5, mp=1),
    "4x4_m5_mp0": dict(C4, method=5, mp=0),
    "8x8_m5_mp1": dict(C8, method=5, mp=1),
    # no odd-even preconditioning: MinRes on the whole Schwarz block, GMRES on the whole coarsest operator
    "4x4_oe0": dict(C4, method=2, mp=1, oe=0),
    "8x8_oe0": dict(C8, method=2, mp=1, oe=0),
    "4x4_oe0_m4": dict(C4, method=4, mp=1, oe=0),
    # the PIPELINED_ARNOLDI build (with SINGLE_ALLREDUCE_ARNOLDI): another recurrence on the coarsest level
    "4x4_pipelined": dict(C4, method=2, mp=1, exe="dd_alpha_amg_scalar_pipelined"),
    "8x8_pipelined": dict(C8, method=2, mp=1, exe="dd_alpha_amg_scalar_pipelined"),
}


def run(name):
    c = dict(CASES[name]); exe = c.pop("exe")
    tmp = tempfile.mkdtemp(prefix="ddamg_case_")
    try:
        c["conf"] = os.path.join(REF, c["conf"])
        open(os.path.join(tmp, "c.ini"), "w").write(INI.format(**c))
        r = subprocess.run([os.path.join(HERE, "_ref", exe), os.path.join(tmp, "c.ini")], capture_output=True, text=True, cwd=tmp)
        out = r.stdout
        hist = [float(l.split(":")[1].split("|")[0]) for l in out.splitlines() if "approx. rel. res. after" in l]
        it = re.search(r"FGMRES iterations:\s*(\d+)\s+coarse average:\s*([0-9.]+)", out)
        rr = re.search(r"exact relative residual: \|\|r\|\|/\|\|b\|\| = ([0-9.eE+-]+)", out)
        bi = [int(x) for x in re.findall(r"biCGstab relres: [0-9.eE+-]+,\s+iterations: (\d+)", out)]
        if not (it and rr):
            sys.stderr.write(out[-3000:] + r.stderr[-2000:]); raise SystemExit(f"{name}: reference run failed")
        res = {"iterations": int(it.group(1)), "coarse_average": float(it.group(2)), "true_relres": float(rr.group(1)),
               "residual_history": hist, "binary": exe, "parameters": {k: v for k, v in c.items() if k != "conf"}}
        if bi:
            res["bicgstab_iterations"] = bi
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    allres = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in (sys.argv[1:] or list(CASES)):
        allres[name] = run(name)
        print(name, {k: allres[name][k] for k in ("iterations", "coarse_average", "true_relres")}, allres[name].get("bicgstab_iterations", "")[:12] if "bicgstab_iterations" in allres[name] else "")
    json.dump(allres, open(OUT, "w"), indent=1)
