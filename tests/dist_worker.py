"""Worker of the multi-process tests (started through torch.distributed.run, gloo backend).

mode plan  (CPU only): every process builds its halo plan through the C-ABI (host-only entry
    point), the face-site coordinates travel as the halo messages would, and every receiver
    checks that slot i of a message really is its neighbour x +- mu on the global lattice.
mode dirac (GPU): the decomposed Wilson-Clover operator (pack -> exchange -> interior -> boundary)
    against the reference's own output for the undivided 8^4 lattice (tests/golden).
"""
import argparse, os, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ddalphaamg_amd import api, dist as ddist  # noqa: E402


def lex_coords(idx, L):
    c = np.zeros((len(idx), 4), dtype=np.int64)
    r = np.asarray(idx, dtype=np.int64).copy()
    for mu in (3, 2, 1, 0):
        c[:, mu] = r % L[mu]
        r //= L[mu]
    return c


def run_plan(rank, world, P, L):
    C = ddist.coords_of(rank, P)
    G = [L[mu] * P[mu] for mu in range(4)]
    origin = np.array([C[mu] * L[mu] for mu in range(4)])
    reqs, expect = [], []
    for mu in range(4):
        for face in (mu, 4 + mu):
            nb, sites = api.halo_plan(L, P, C, face)
            if P[mu] == 1:
                assert len(sites) == 0 and nb == rank
                continue
            assert len(sites) == int(np.prod(L)) // L[mu]
            gc = lex_coords(sites, L) + origin      # global coordinates of what I send
            assert np.all(lex_coords(sites, L)[:, mu] == (L[mu] - 1 if face < 4 else 0))
            reqs.append(dist.isend(torch.from_numpy(gc.copy()), dst=nb, tag=face))
            # what arrives with this tag comes from the opposite neighbour and describes ITS face
            # sites; they must be my opposite-face sites shifted by -+1 in mu
            opp_nb, opp_sites = api.halo_plan(L, P, C, (face + 4) % 8)
            buf = torch.empty((len(sites), 4), dtype=torch.int64)
            reqs.append(dist.irecv(buf, src=opp_nb, tag=face))
            mine = lex_coords(opp_sites, L) + origin
            shift = np.zeros(4, dtype=np.int64); shift[mu] = -1 if face < 4 else +1
            expect.append((buf, (mine + shift) % np.array(G)))
    for r in reqs:
        r.wait()
    for buf, want in expect:
        assert np.array_equal(buf.numpy(), want), "halo slot order does not match the neighbour's"
    return 0.0


def run_dirac(rank, world, P, prec, transport):
    import ddalphaamg_amd as dd
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "ref_8x8_dirac.npz"))
    G = [int(x) for x in g["meta_int"][:4]]
    L = [G[mu] // P[mu] for mu in range(4)]
    C = ddist.coords_of(rank, P)
    # the undivided operator, from the same entry point the single-GPU parity tests pin to the reference
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = G[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw = g["meta_f64"][0], g["meta_f64"][1]
    whole = dd.Context(p)
    whole.set_gauge(g["gauge"], anti_pbc=True)
    D, cl = whole.get_operator()
    whole.close()
    q = api.default_params(); q.num_levels = 1
    for mu in range(4):
        q.local_lattice[0][mu] = L[mu]; q.block_lattice[0][mu] = 4 if L[mu] % 4 == 0 else 2
        q.process_grid[mu] = P[mu]; q.process_coords[mu] = C[mu]
    q.m0, q.csw = p.m0, p.csw
    ctx = dd.Context(q)
    ctx.set_operator(ddist.local_part(D, G, P, C), ddist.local_part(cl, G, P, C))
    if transport == "host":
        ddist.attach_host(ctx)
    else:
        ddist.attach_rccl(ctx, rank)
    x = ctx.vector(0, prec).upload(ddist.local_part(g["dirac_in"], G, P, C))
    y = ctx.vector(0, prec)
    err = 0.0
    for _ in range(3):   # repeated applies reuse the send/recv arenas
        ctx.dirac_apply(y, x)
        want = ddist.local_part(g["dirac_out_f64"], G, P, C).reshape(-1, 12, 2)
        got = y.download()
        err = max(err, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
    dist.barrier()
    ctx.close()
    return err


def run_gauge(rank, world, P):
    """gauge -> operator on the process grid (links of the neighbouring processes fetched, corners included)
    against the parts of the operator built on the undivided lattice"""
    import ddalphaamg_amd as dd
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "ref_8x8_dirac.npz"))
    G = [8, 8, 8, 8]
    L = [G[mu] // P[mu] for mu in range(4)]
    C = ddist.coords_of(rank, P)
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = G[mu]; p.block_lattice[0][mu] = 2
    p.m0, p.csw = -0.5, 1.0
    whole = dd.Context(p)
    plaq = whole.set_gauge(g["gauge"], anti_pbc=True)
    D, cl = whole.get_operator()
    whole.close()
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.process_grid[mu] = P[mu]; p.process_coords[mu] = C[mu]
    ctx = dd.Context(p)
    ddist.attach_host(ctx)
    plaq_d = ctx.set_gauge(ddist.local_part(g["gauge"].reshape(4096, -1), G, P, C), anti_pbc=True)
    Dl, cll = ctx.get_operator()
    assert abs(plaq_d - plaq) < 1e-12 and abs(plaq - float(g["meta_f64"][2])) < 1e-12, (plaq_d, plaq)
    assert np.array_equal(Dl.reshape(-1), ddist.local_part(D, G, P, C).reshape(-1))
    err = float(np.linalg.norm(cll.reshape(-1) - ddist.local_part(cl, G, P, C).reshape(-1)) / np.linalg.norm(cll))
    dist.barrier()
    ctx.close()
    return err


def run_gmres(rank, world, P, mp):
    """pure GMRES (method 0) on the decomposed 8^4 sample configuration: global reductions + halo exchange;
    against the same solve on the undivided lattice (whose parity with the reference is tests/test_gpu_multigrid.py)"""
    import ddalphaamg_amd as dd
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "ref_8x8_dirac.npz"))
    G = [8, 8, 8, 8]
    L = [G[mu] // P[mu] for mu in range(4)]
    C = ddist.coords_of(rank, P)

    def params(lat, grid, coords):
        p = api.default_params(); p.num_levels = 1
        for mu in range(4):
            p.local_lattice[0][mu] = lat[mu]; p.block_lattice[0][mu] = 2
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        p.method, p.mixed_precision, p.restart, p.max_restart, p.tol = 0, mp, 50, 20, 1e-10
        p.m0, p.csw = -0.5, 1.0
        return p

    whole = dd.Context(params(G, [1] * 4, [0] * 4))
    whole.set_gauge(g["gauge"], anti_pbc=True)
    D, cl = whole.get_operator()
    b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
    x1, it1, _, rr1 = whole.solve(b, 1e-10)
    whole.close()
    ctx = dd.Context(params(L, P, C))
    ctx.set_operator(ddist.local_part(D, G, P, C), ddist.local_part(cl, G, P, C))
    ddist.attach_host(ctx)
    xl, it, _, rr = ctx.solve(ddist.local_part(b, G, P, C), 1e-10)
    want = ddist.local_part(x1, G, P, C).reshape(-1, 12, 2)
    err = float(np.linalg.norm(xl - want) / np.linalg.norm(want))
    assert abs(it - it1) <= 2, (it, it1)
    assert rr < 1.5e-10, rr
    dist.barrier()
    ctx.close()
    if rank == 0:
        print(f"gmres mp{mp}: {it} iterations on the process grid, {it1} undivided; relres {rr:.3e} / {rr1:.3e}", flush=True)
    return err


def run_amg(rank, world, P, mp, levels=2, G=None, method=2, gather=0):
    """two-level FGMRES+AMG on the decomposed 8^4 sample configuration (2^4 blocks and aggregates, Nvec 20):
    (1) the hierarchy of the undivided run is handed over (interpolation vectors) and the Galerkin operator,
    smoother, coarse operator and solve of the decomposed run are compared with it; (2) the decomposed run
    does its own setup (each process draws its own random test vectors, srand(1000*rank) as the reference)."""
    import ddalphaamg_amd as dd
    here = os.path.dirname(os.path.abspath(__file__))
    if G is None or list(G) == [8, 8, 8, 8]:
        g = np.load(os.path.join(here, "golden", "ref_8x8_dirac.npz"))
        G = [8, 8, 8, 8]; gauge = g["gauge"]; m0 = -0.5
    else:   # synthetic random links (same seed on every process); m0 keeps the hot configuration solvable
        sys.path.insert(0, here)
        from conftest import random_su3
        gauge = random_su3(int(np.prod(G)) * 4, 99).reshape(-1, 4, 9, 2); m0 = 0.25
    Gc = [x // 2 for x in G]
    L = [G[mu] // P[mu] for mu in range(4)]
    C = ddist.coords_of(rank, P)

    def params(lat, grid, coords):
        p = api.default_params(); p.num_levels = levels
        for mu in range(4):
            p.local_lattice[0][mu] = lat[mu]; p.block_lattice[0][mu] = 2; p.local_lattice[1][mu] = lat[mu] // 2
            if levels == 3:
                p.block_lattice[1][mu] = 2; p.local_lattice[2][mu] = lat[mu] // 4
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        p.num_vect[0] = 20; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 3
        p.num_vect[1] = 24; p.post_smooth_iter[1] = 2; p.block_iter[1] = 4; p.setup_iter[1] = 2
        p.restart, p.max_restart, p.tol = 30, 20, 1e-10
        p.coarse_iter, p.coarse_restart, p.coarse_tol = 30, 10, 5e-2
        p.mixed_precision, p.method, p.odd_even = mp, method, 1
        p.m0, p.csw = m0, 1.0
        p.gather_coarsest = gather if int(np.prod(grid)) > 1 else 0   # the coarsest level whole on every process (one all-gather per V-cycle)
        return p

    def rel(a, b):
        return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))

    V = int(np.prod(G)); Vc = V // 16
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    rng = np.random.default_rng(5)
    eta = rng.standard_normal((V, 12, 2)); phi0 = rng.standard_normal((V, 12, 2))
    vc = rng.standard_normal((Vc, 40, 2))
    whole = dd.Context(params(G, [1] * 4, [0] * 4))
    whole.set_gauge(gauge, anti_pbc=True)
    D, cl = whole.get_operator()
    whole.setup(3)
    Pint = whole.get_interpolation()
    Dc, clc = whole.get_coarse_operator()
    x1, it1, cit1, rr1 = whole.solve(b, 1e-10)
    prec = whole.vprec()
    e = whole.vector(0, prec).upload(eta); ph = whole.vector(0, prec).upload(phi0)
    whole.smoother(ph, e, 2, initial_guess_zero=False)
    smooth1 = ph.download()
    ci = whole.vector(1, prec).upload(vc); co = whole.vector(1, prec)
    whole.coarse_apply(co, ci)
    capp1 = co.download()
    # mass shift on the device (shift_update, ddamg_hip_shift_mass): undivided here, decomposed below
    m1 = m0 + 0.04
    whole.shift_mass(m1)
    xs1, its1, cits1, rrs1 = whole.solve(b, 1e-10)
    whole.close()

    lp = lambda a, lat=G: ddist.local_part(a, lat, P, C)
    # (1) same hierarchy
    ctx = dd.Context(params(L, P, C))
    ctx.set_operator(lp(D), lp(cl))
    ddist.attach_host(ctx)
    Ploc = np.stack([lp(Pint[k]) for k in range(20)])
    ctx.set_test_vectors(Ploc, orthonormalised=True)
    Dcl, clcl = ctx.get_coarse_operator()
    errs = {"galerkin_D": rel(Dcl.reshape(-1), lp(Dc, Gc).reshape(-1)), "galerkin_self": rel(clcl.reshape(-1), lp(clc, Gc).reshape(-1))}
    e = ctx.vector(0, prec).upload(lp(eta)); ph = ctx.vector(0, prec).upload(lp(phi0))
    ctx.smoother(ph, e, 2, initial_guess_zero=False)
    errs["smoother"] = rel(ph.download().reshape(-1), lp(smooth1).reshape(-1))
    ci = ctx.vector(1, prec).upload(lp(vc, Gc)); co = ctx.vector(1, prec)
    ctx.coarse_apply(co, ci)
    errs["coarse_apply"] = rel(co.download().reshape(-1), lp(capp1, Gc).reshape(-1))
    xl, it, cit, rr = ctx.solve(lp(b), 1e-10)
    errs["solution"] = rel(xl.reshape(-1), lp(x1).reshape(-1))
    ctx.shift_mass(m1)     # every process shifts its part of every level (the gathered coarsest level included)
    xls, its, cits, rrs = ctx.solve(lp(b), 1e-10)
    errs["solution_after_mass_shift"] = rel(xls.reshape(-1), lp(xs1).reshape(-1))
    assert abs(its - its1) <= (1 if levels == 2 else 2) and rrs < 1.5e-10, (its, its1, rrs)
    ctx.close()
    # (2) own setup on the process grid
    ctx = dd.Context(params(L, P, C))
    ctx.set_operator(lp(D), lp(cl))
    ddist.attach_host(ctx)
    ctx.setup(3)
    xl2, it2, cit2, rr2 = ctx.solve(lp(b), 1e-10)
    errs["solution_own_setup"] = rel(xl2.reshape(-1), lp(x1).reshape(-1))
    dist.barrier()
    ctx.close()
    if rank == 0:
        print(f"amg mp{mp} method {method} levels {levels}: undivided {it1} its ({cit1} coarse) relres {rr1:.2e} | same hierarchy {it} ({cit}) {rr:.2e} | own setup {it2} ({cit2}) {rr2:.2e}", flush=True)
        print("errs", {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)
    tol32 = {"galerkin_D": 2e-5, "galerkin_self": 2e-5, "smoother": 5e-5, "coarse_apply": 2e-5, "solution": 1e-7, "solution_own_setup": 1e-7, "solution_after_mass_shift": 1e-7}
    for k, v in errs.items():
        assert v < tol32[k], (k, v)
    # every rank must have taken the same stopping decisions (the Krylov control flow runs on every host from the same
    # reduced numbers: a rank that read a stale buffer would diverge here before it hangs in a collective)
    decisions = [None] * world
    dist.all_gather_object(decisions, (it, cit, it2, cit2, round(rr, 18), round(rr2, 18)))
    assert all(d == decisions[0] for d in decisions), decisions
    assert abs(it - it1) <= (1 if levels == 2 else 2) and abs(it2 - it1) <= 2, (it1, it, it2)
    if gather:   # the coarsest system is the undivided one: the same solver trajectory up to the rounding of the Galerkin operator
        assert it == it1 and abs(cit - cit1) <= max(2, cit1 // 50), (it1, cit1, it, cit)
    assert rr < 1.5e-10 and rr2 < 1.5e-10
    return max(errs["solution"], errs["solution_own_setup"])


def run_sample_np2(rank, world, P):
    """the reference's own sample.ini hierarchy on its 8^4 configuration, on the process grid 2x1x1x1, against the
    reference run on 2 MPI ranks (tests/golden/ref_8x8_3lvl_np2.json): same setup (srand(1000*rank) test vectors)"""
    import json
    import ddalphaamg_amd as dd
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "ref_8x8_dirac.npz"))
    ref = json.load(open(os.path.join(here, "golden", f"ref_8x8_3lvl_np{world}.json")))
    assert P == ref["process_grid"]
    G = [8, 8, 8, 8]
    C = ddist.coords_of(rank, P)
    L = [G[mu] // P[mu] for mu in range(4)]

    def params(lat, grid, coords):
        p = api.default_params(); p.num_levels = 3
        for mu in range(4):
            p.local_lattice[0][mu] = lat[mu]; p.block_lattice[0][mu] = 2
            p.local_lattice[1][mu] = lat[mu] // 2; p.block_lattice[1][mu] = 2
            p.local_lattice[2][mu] = lat[mu] // 4
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        p.num_vect[0] = 28; p.num_vect[1] = 28
        p.post_smooth_iter[0] = p.post_smooth_iter[1] = 2; p.block_iter[0] = p.block_iter[1] = 4
        p.setup_iter[0] = 4; p.setup_iter[1] = 3
        p.restart, p.max_restart, p.tol = 50, 20, 1e-10
        p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
        p.kcycle, p.kcycle_restart, p.kcycle_max_restart, p.kcycle_tol = 1, 5, 2, 1e-1
        p.mixed_precision, p.method, p.odd_even = 1, 2, 1
        p.m0, p.csw = -0.5, 1.0
        return p

    whole = dd.Context(params(G, [1] * 4, [0] * 4))
    whole.set_gauge(g["gauge"], anti_pbc=True)
    D, cl = whole.get_operator()
    whole.close()
    ctx = dd.Context(params(L, P, C))
    ctx.set_operator(ddist.local_part(D, G, P, C), ddist.local_part(cl, G, P, C))
    ddist.attach_host(ctx)
    ctx.setup(4)
    b = np.zeros((int(np.prod(L)), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    hist = ctx.residual_history()
    dist.barrier()
    ctx.close()
    if rank == 0:
        print(f"sample.ini on {world} processes: {it} iterations, {cit} coarse, relres {rr:.6e}  (reference: {ref['iterations']}, "
              f"{ref['coarse_iterations']}, {ref['exact_relative_residual']:.6e})", flush=True)
        print("history", " ".join(f"{h:.6e}" for h in hist), flush=True)
    assert it == ref["iterations"] and abs(cit - ref["coarse_iterations"]) <= 3, (it, cit, ref["iterations"], ref["coarse_iterations"])
    assert rr < 1e-10
    n = min(len(hist), len(ref["history"]))
    dev = float(np.max(np.abs(np.asarray(hist[:n]) / np.asarray(ref["history"][:n]) - 1.0)))
    assert dev < 2e-3, dev     # the reference's (scalar build) residual history on the same process grid, digit for digit at the start
    return dev * 1e-6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="plan")
    ap.add_argument("--grid", default="2,1,1,1")
    ap.add_argument("--lattice", default="4,4,4,4")
    ap.add_argument("--prec", type=int, default=64)
    ap.add_argument("--transport", default="host")
    ap.add_argument("--tol", type=float, default=1e-13)
    ap.add_argument("--gather", type=int, default=0, help="amg modes: gather the coarsest level on every process")
    ap.add_argument("--method", type=int, default=2, help="Schwarz schedule of the amg modes: 1 additive, 2 red-black, 3 sixteen colours")
    a = ap.parse_args()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    P = [int(x) for x in a.grid.split(",")]
    assert int(np.prod(P)) == world
    if a.mode == "plan":
        err = run_plan(rank, world, P, [int(x) for x in a.lattice.split(",")])
    elif a.mode == "gauge":
        err = run_gauge(rank, world, P)
    elif a.mode == "amg":
        err = run_amg(rank, world, P, a.prec, method=a.method, gather=a.gather)
    elif a.mode == "sample_np2":
        err = run_sample_np2(rank, world, P)
    elif a.mode == "amg3":
        err = run_amg(rank, world, P, a.prec, levels=3, G=[int(x) for x in a.lattice.split(",")], method=a.method, gather=a.gather)
    elif a.mode == "gmres":
        err = run_gmres(rank, world, P, a.prec)
    else:
        err = run_dirac(rank, world, P, a.prec, a.transport)
    t = torch.tensor([err], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"DIST_WORKER_OK mode={a.mode} grid={a.grid} err={t.item():.3e}", flush=True)
    dist.destroy_process_group()
    if not t.item() <= a.tol:
        sys.exit(3)


if __name__ == "__main__":
    main()
