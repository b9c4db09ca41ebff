#!/usr/bin/env python3
"""tools/solve_bench.py -- set up a 2- or 3-level hierarchy on a synthetic lattice and time the FGMRES+AMG solve.

  python tools/solve_bench.py --lattice 16 16 16 16 --block 4 4 4 4 --nvec 24 --setup-iter 4
"""
import argparse, os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))


from bench import near_unit_gauge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lattice", type=int, nargs=4, default=[16, 16, 16, 16])
    ap.add_argument("--block", type=int, nargs=4, default=[4, 4, 4, 4])
    ap.add_argument("--agg", type=int, nargs=4, default=[4, 4, 4, 4])
    ap.add_argument("--nvec", type=int, default=24)
    ap.add_argument("--setup-iter", type=int, default=4)
    ap.add_argument("--m0", type=float, default=-0.3)
    ap.add_argument("--csw", type=float, default=1.0)
    ap.add_argument("--eps", type=float, default=0.35)
    ap.add_argument("--solves", type=int, default=2)
    ap.add_argument("--levels", type=int, default=2)
    ap.add_argument("--agg1", type=int, nargs=4, default=[2, 2, 2, 2], help="level-1 aggregates (= blocks) for --levels 3")
    ap.add_argument("--nvec1", type=int, default=28)
    ap.add_argument("--mixed-precision", type=int, default=1)
    ap.add_argument("--method", type=int, default=2, help="smoother: 1 additive, 2 red-black, 3 sixteen-colour SAP, 4 GMRES")
    ap.add_argument("--gauge", default="near_unit", choices=["near_unit", "random"])
    ap.add_argument("--self-exchange", default=None, help="e.g. -1,-1,-1,1: the process is its own neighbour in these directions (RCCL)")
    ap.add_argument("--rng", type=int, default=1, help="0: libc rand() in the reference's order, 1: device generator")
    ap.add_argument("--gather", type=int, default=0, help="gather the coarsest level (ddamg_hip_params::gather_coarsest)")
    args = ap.parse_args()
    import ddalphaamg_amd as dd
    from ddalphaamg_amd import api
    L = args.lattice; V = int(np.prod(L))
    p = api.default_params(); p.num_levels = args.levels
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = args.block[mu]; p.local_lattice[1][mu] = L[mu] // args.agg[mu]
        if args.levels == 3:
            p.block_lattice[1][mu] = args.agg1[mu]; p.local_lattice[2][mu] = p.local_lattice[1][mu] // args.agg1[mu]
    p.num_vect[0] = args.nvec; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = args.setup_iter
    if args.levels == 3:
        p.num_vect[1] = args.nvec1; p.post_smooth_iter[1] = 2; p.block_iter[1] = 4; p.setup_iter[1] = 2
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = args.mixed_precision, args.method, 1
    p.m0, p.csw = args.m0, args.csw
    p.test_vector_rng, p.rng_seed = args.rng, 20260101
    p.gather_coarsest = args.gather
    if args.self_exchange:
        for mu, v in enumerate(int(x) for x in args.self_exchange.split(",")):
            p.process_grid[mu] = v
    ctx = dd.Context(p)
    if args.self_exchange:
        ctx.comm_init_rccl(api.rccl_unique_id())
    t0 = time.time()
    if args.gauge == "near_unit":
        U = near_unit_gauge(V, args.eps, 20260101)
    else:
        sys.path.insert(0, REPO)
        from bench import synth_gauge_random as synth_gauge
        U = synth_gauge(V, 20260101)
    t1 = time.time()
    plaq = ctx.set_gauge(U, anti_pbc=True); t2 = time.time()
    print(f"gauge gen {t1-t0:.1f}s  set_gauge {t2-t1:.1f}s  plaquette {plaq:.6f}", flush=True)
    t0 = time.time(); ci = ctx.setup(args.setup_iter); ctx.sync(); t1 = time.time()
    print(f"setup: {t1-t0:.2f}s (coarse its {ci})", flush=True)
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    for s in range(args.solves):
        bv = ctx.vector(0, 64).upload(b); xv = ctx.vector(0, 64)
        t0 = time.time(); it, cit, rr = ctx.solve_vec(xv, bv, 1e-10); t1 = time.time()
        bv.free(); xv.free()
        print(json.dumps({"solve_s": t1 - t0, "iters": it, "coarse_iters": cit, "coarse_avg": cit / max(it, 1), "relres": rr}), flush=True)
    print("history", " ".join(f"{h:.3e}" for h in ctx.residual_history()))
    ctx.close()


if __name__ == "__main__":
    main()
