#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY.

Generates the committed golden fixtures tests/golden/*.npz by running the REAL reference
(oracle/_ref/ref_dump, built by `make -C oracle ref` from /root/reference) on the reference's own
sample gauge configurations.  Runs only in the build container (needs /root/reference); the
fixtures themselves are pure data (inputs + expected outputs) and travel with the repo.

usage: python oracle/make_golden.py [case ...]      cases: 4x4 8x8_dirac
"""
import os, subprocess, sys, tempfile, shutil
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("DDAMG_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")

INI_COMMON = """configuration: {conf}
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
m0: {m0}
csw: 1.0
tolerance for relative residual: 1E-10
iterations between restarts: 50
maximum of restarts: 20
coarse grid tolerance: 5E-2
coarse grid iterations: 100
coarse grid restarts: 5
print mode: 1
method: {method}
mixed precision: {mp}
randomize test vectors: 0
"""

CASE_3LVL_EXTRA = "d1 global lattice: 4 4 4 4\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 3"

THREE_LEVEL_KEEP = ["meta_int", "meta_f64", "meta3_int", "coarse_D", "coarse_clover", "l1_interp_vectors", "l2_coarse_D", "l2_coarse_clover",
                    "l1_apply_in", "l1_apply_out", "l2_apply_in", "l2_apply_out", "l1_restrict_in", "l1_restrict_out", "l1_interpolate_in",
                    "l1_interpolate_out", "l1_smoother_eta", "l1_smoother_nores_out_c1", "l1_smoother_nores_out_c2", "l1_smoother_nores_out_c3",
                    "l1_smoother_phi0", "l1_smoother_res_out_c2", "l1_vcycle_eta", "l1_vcycle_out", "ones_solve_iters", "ones_solve_norm_res"]

CASES = {
    # BASELINE.md explicit 2-level 4^4 case (11 iterations, 2.44e-11)
    "4x4": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=4,
                extra="", method=2, mp=1),
    # 8^4 fine operator only (BASELINE config 2): 2-level hierarchy is built but only stage 1 is kept
    "8x8_dirac": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=2, L="8 8 8 8", B="4 4 4 4", nvec=4, setup=0,
                      extra="d1 global lattice: 2 2 2 2", method=2, mp=1, keep=["meta_int", "meta_f64", "dirac_in",
                      "dirac_out_f64", "dirac_out_f32_as_f64", "clover_sample", "gauge", "conf_dims", "conf_plaq"]),
    # the reference's sample.ini: 8^4, 3 levels (8^4 -> 4^4 -> 2^4), Nvec 28/28, K-cycle; only iteration counts and the
    # residual history of the rhs=ones solve are kept (the gauge field is in ref_8x8_dirac.npz)
    "8x8_3lvl": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=28, setup=4,
                     extra="d1 global lattice: 4 4 4 4\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 3",
                     method=2, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # the same hierarchy with mixed precision 2 (fgmres_MP outside the K-cycle)
    "8x8_3lvl_mp2": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=28, setup=4,
                         extra="d1 global lattice: 4 4 4 4\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 3",
                         method=2, mp=2, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # mixed precision 2 (fgmres_MP): AMG-preconditioned on 4^4 and pure GMRES(50) on 8^4 (383 iterations)
    "4x4_mp2": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=4, extra="", method=2, mp=2,
                    keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # the ragged lattice below with mixed precision 2 (fgmres_MP)
    "ragged_mp2": dict(conf="", synthetic=4711, levels=2, L="8 4 4 8", B="4 2 2 2", nvec=12, setup=2, m0=0.3,
                       extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=2, mp=2,
                       keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # four different extents everywhere: lattice 8x4x4x8 (T,Z,Y,X), Schwarz blocks 4x2x2x2, aggregates 4x2x2x4
    # (coarse lattice 2x2x2x2), seeded random SU(3) links, m0 = 0.3
    "ragged": dict(conf="", synthetic=4711, levels=2, L="8 4 4 8", B="4 2 2 2", nvec=12, setup=2, m0=0.3,
                   extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=2, mp=1),
    # 4^4 Schwarz blocks (256 sites: the block shape of the production configurations and of the optimised kernels) on the
    # reference's 8^4 configuration: smoother dumps and the rhs=ones solve after the reference's own setup
    "8x8_b4": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=2, L="8 8 8 8", B="4 4 4 4", nvec=20, setup=3,
                   extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=2, mp=1,
                   keep=["meta_int", "meta_f64", "smoother_eta", "smoother_phi0", "smoother_nores_out_c1", "smoother_nores_out_c2",
                         "smoother_nores_out_c3", "smoother_res_out_c2", "ones_solve_iters", "ones_solve_norm_res"]),
    # production block shapes at a larger volume: 16^4, 4^4 blocks and aggregates on the fine level (-> 4^4), 2^4 on the
    # coarse level (-> 2^4), K-cycle; seeded random links (the test regenerates them: conftest.random_su3(V*4, 1616))
    "16x16_3lvl": dict(conf="", synthetic=1616, levels=3, L="16 16 16 16", B="4 4 4 4", nvec=24, setup=3, m0=0.3,
                       extra="d1 global lattice: 4 4 4 4\nd1 local lattice: 4 4 4 4\nd1 block lattice: 2 2 2 2\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 2",
                       method=2, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # the same hierarchy on smooth links (bench.near_unit_gauge(V, 0.35, 1617)), m0 = -0.3: a hard system
    "16x16_3lvl_hard": dict(conf="", synthetic=1617, near_unit=True, levels=3, L="16 16 16 16", B="4 4 4 4", nvec=24, setup=3, m0=-0.3,
                            extra="d1 global lattice: 4 4 4 4\nd1 local lattice: 4 4 4 4\nd1 block lattice: 2 2 2 2\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 2",
                            method=2, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # the other Schwarz schedules behind the same smoother seam: additive (method 1) and sixteen colours
    # (method 3), and the GMRES smoother on the global odd-even Schur complement (method 4); 4^4 with 2^4 blocks has one block per colour, the ragged lattice two, the 8^4 three-level runs use the
    # schedules on the coarse level as well (4^4 coarse lattice, 2^4 blocks)
    "4x4_m1": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=4, extra="", method=1, mp=1,
                   keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "ragged_m1": dict(conf="", synthetic=4711, levels=2, L="8 4 4 8", B="4 2 2 2", nvec=12, setup=2, m0=0.3,
                      extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=1, mp=1, keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "8x8_3lvl_m1": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=28, setup=4,
                        extra=CASE_3LVL_EXTRA, method=1, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    "4x4_m3": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=4, extra="", method=3, mp=1,
                   keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "ragged_m3": dict(conf="", synthetic=4711, levels=2, L="8 4 4 8", B="4 2 2 2", nvec=12, setup=2, m0=0.3,
                      extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=3, mp=1, keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "8x8_3lvl_m3": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=28, setup=4,
                        extra=CASE_3LVL_EXTRA, method=3, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    "4x4_m4": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=4, extra="", method=4, mp=1,
                   keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "ragged_m4": dict(conf="", synthetic=4711, levels=2, L="8 4 4 8", B="4 2 2 2", nvec=12, setup=2, m0=0.3,
                      extra="d1 global lattice: 2 2 2 2\nd1 local lattice: 2 2 2 2", method=4, mp=1, keep=["meta_int", "meta_f64", "smoother_nores_out_c1", "smoother_nores_out_c2", "smoother_nores_out_c3", "smoother_res_out_c2", "solve_iters", "solve_norm_res", "ones_solve_iters", "ones_solve_norm_res"]),
    "8x8_3lvl_m4": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=28, setup=4,
                        extra=CASE_3LVL_EXTRA, method=4, mp=1, keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
    # three levels, every level's data small enough to commit: the reference's 8^4 configuration, 8^4 -> 4^4 -> 2^4, 8 / 10 test
    # vectors (16 / 20 dof per coarse site).  Dumps of the level-0 AND level-1 interpolation vectors, both coarse operators, and the
    # intermediate level's hot-path functions (operator, restriction / interpolation, Schwarz smoother, V-cycle)
    "8x8_3lvl_small": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=3, L="8 8 8 8", B="2 2 2 2", nvec=8, setup=3,
                           extra="d1 global lattice: 4 4 4 4\nd1 local lattice: 4 4 4 4\nd1 block lattice: 2 2 2 2\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 10\nd1 setup iter: 2\nd2 global lattice: 2 2 2 2\nd2 local lattice: 2 2 2 2",
                           method=2, mp=1, keep=THREE_LEVEL_KEEP + ["interp_vectors"]),
    # the production dof counts (24 -> 48 dof, 28 -> 56 dof per site) on the smallest lattices that carry them: 16 x 8^3 with 4^4
    # aggregates -> 4 x 2^3 (blocks and aggregates of 2 x 1^3 sites) -> 2^4; seeded random links (the test regenerates them:
    # conftest.random_su3(V*4, 1618)); the fine level's vectors are not kept -- the level-1 operator is
    "16x8_3lvl_prod": dict(conf="", synthetic=1618, levels=3, L="16 8 8 8", B="4 4 4 4", nvec=24, setup=2, m0=0.3,
                           extra="d1 global lattice: 4 2 2 2\nd1 local lattice: 4 2 2 2\nd1 block lattice: 2 1 1 1\nd1 post smooth iter: 2\nd1 block iter: 4\nd1 test vectors: 28\nd1 setup iter: 2\nd2 global lattice: 2 2 2 2\nd2 local lattice: 2 2 2 2",
                           method=2, mp=1, keep=THREE_LEVEL_KEEP),
    # pure CGN (method -1): conjugate gradients on the normal equations, needs D^dagger = g5 D g5
    "4x4_cgn": dict(conf="conf/4x4x4x4b6.0000id3n1", levels=2, L="4 4 4 4", B="2 2 2 2", nvec=20, setup=0, extra="", method=-1, mp=1,
                    keep=["meta_int", "meta_f64", "cgn_x"]),
    "8x8_gmres_mp2": dict(conf="conf/8x8x8x8b6.0000id3n1", levels=1, L="8 8 8 8", B="2 2 2 2", nvec=4, setup=0, extra="", method=0, mp=2,
                          keep=["meta_int", "meta_f64", "ones_solve_iters", "ones_solve_norm_res"]),
}

def run_case(name):
    cfg = dict(CASES[name])
    keep = cfg.pop("keep", None)
    cfg.setdefault("m0", -0.5)
    synthetic = cfg.pop("synthetic", None)
    tmp = tempfile.mkdtemp(prefix="ddamg_gold_")
    if synthetic is None:
        cfg["conf"] = os.path.join(REF, cfg["conf"])
    else:
        # a gauge file in the reference's format (src/io.c:489-520) with seeded random SU(3) links: exercises
        # lattices and blocks with four different extents, which the reference's sample configurations do not
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from conftest import random_su3
        Ls = [int(x) for x in cfg["L"].split()]
        if cfg.pop("near_unit", False):
            sys.path.insert(0, REPO)
            from bench import near_unit_gauge      # smooth links exp(i 0.35 H): a system on which multigrid has work to do
            U = near_unit_gauge(int(np.prod(Ls)), 0.35, synthetic)
        else:
            U = random_su3(int(np.prod(Ls)) * 4, synthetic)
        cfg["conf"] = os.path.join(tmp, "synthetic.conf")
        with open(cfg["conf"], "wb") as f:
            f.write(np.asarray(Ls, dtype="<i4").tobytes()); f.write(np.asarray([0.0], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(U, dtype="<f8").tobytes())
    try:
        ini = os.path.join(tmp, "case.ini")
        open(ini, "w").write(INI_COMMON.format(**cfg))
        log = subprocess.run([os.path.join(HERE, "_ref", "ref_dump"), ini, tmp], capture_output=True, text=True)
        if log.returncode != 0 or not os.path.exists(os.path.join(tmp, "manifest.txt")):   # error0 aborts with exit code 0
            sys.stderr.write(log.stdout[-3000:] + log.stderr[-3000:])
            raise SystemExit("ref_dump failed")
        arrays = {}
        for line in open(os.path.join(tmp, "manifest.txt")):
            nm, dt, shape = line.split()
            shape = tuple(int(x) for x in shape.split(","))
            a = np.fromfile(os.path.join(tmp, nm + ".bin"), dtype=np.dtype("<" + dt)).reshape(shape)
            arrays[nm] = a
        if keep is not None and "gauge" not in keep:
            arrays.pop("gauge", None); arrays.pop("conf_dims", None)
        if keep is not None:
            if "clover" in arrays and "clover_sample" in keep:  # a few sites of the reference clover term, to pin set_gauge at 8^4
                arrays["clover_sample"] = arrays["clover"][::97].copy()
                arrays["D_sample"] = arrays["D"][::97].copy()
                keep = keep + ["D_sample"]
            keep_now = keep
        # the gauge configuration itself (input data of the reference's own sample runs):
        # 4 x int32 (T,Z,Y,X) + double plaquette header, then [t][z][y][x][mu][3x3] complex doubles
        # (reference src/io.c:489-520)
        raw = open(cfg["conf"], "rb").read()
        dims = np.frombuffer(raw[:16], dtype="<i4")
        arrays["conf_dims"] = dims.copy()
        arrays["conf_plaq"] = np.frombuffer(raw[16:24], dtype="<f8").copy()
        arrays["gauge"] = np.frombuffer(raw[24:], dtype="<f8").reshape(int(np.prod(dims)), 4, 9, 2).copy()
        # the text log holds the reference's residual history / iteration count
        lines = log.stdout.splitlines()
        hist = [float(l.split(":")[1].split("|")[0]) for l in lines if "approx. rel. res. after" in l]
        arrays["ref_log_residual_history"] = np.array(hist)
        if "BEGIN_CGN_SOLVE" in lines:
            seg = lines[lines.index("BEGIN_CGN_SOLVE"):lines.index("END_CGN_SOLVE")]
            arrays["ref_log_cgn_iterations"] = np.array([int(l.split(":")[1].split("|")[0]) for l in seg if "CGN iterations:" in l])
            arrays["ref_log_cgn_switch"] = np.array([[float(l.split("iter")[1].split("true")[0]), float(l.split("res:")[1].split("|")[0])] for l in seg if "switching to CGNR" in l])
        if "BEGIN_ONES_SOLVE" in lines:
            seg = lines[lines.index("BEGIN_ONES_SOLVE"):lines.index("END_ONES_SOLVE")]
            arrays["ref_log_ones_history"] = np.array([float(l.split(":")[1].split("|")[0]) for l in seg if "approx. rel. res. after" in l])
        if keep is not None:
            arrays = {k: v for k, v in arrays.items() if k in keep_now or k.startswith("ref_log")}
        # arrays whose values are exactly representable in fp32 (dumps of float data) are stored as fp32
        for k, v in list(arrays.items()):
            if v.dtype == np.float64 and v.size > 4096 and np.array_equal(v, v.astype(np.float32).astype(np.float64)):
                arrays[k] = v.astype(np.float32)
        os.makedirs(GOLD, exist_ok=True)
        out = os.path.join(GOLD, f"ref_{name}.npz")
        np.savez_compressed(out, **arrays)
        print(f"{out}: {os.path.getsize(out)/1e3:.0f} kB, arrays: {sorted(arrays)}")
    finally:
        shutil.rmtree(tmp)

if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        run_case(c)
