#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DDalphaAMG hot path.

  python bench.py --gpus N --steps K --warmup W

A "step" is one fine Wilson-Clover apply (d_plus_clover_float, reference src/dirac_generic.c:159-277)
over a 32^4 local lattice of synthetic random SU(3) gauge links (BASELINE.json: the >=40 %-of-HBM
target is quoted on exactly this; the 8^4 reference configuration lives in L2 and is a parity
case, not a bandwidth case).  Inputs are resident in HBM before the timed region.  Rank 0 prints
ONE JSON line: metric fine_wilson_clover_gflops (reference flop model: 1920 flop/site,
src/init_generic.c:59,61), plus `roofline` (algorithmic 816 B/site, HIP-event time per launch on
the library's stream) and `cpu_baseline` (the oracle port, or the real reference when it runs,
timed on the host cores of the same box), plus `solve` (configs[2]: two-level FGMRES+AMG at 32^4) and, with
--small-lattice, `small_lattice` (BASELINE configs[1], the reference's 8^4 configuration: cache-resident, not used for
the roofline).
"""
import argparse, json, os, sys, time
import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

FLOP_PER_SITE = 1920          # reference model, src/init_generic.c:59,61
BYTES_PER_SITE_F32 = 816      # 4 B x (24 in + 24 out + 72 links + 84 clover reals), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_gauge(V, seed):
    from conftest import random_su3
    out = np.empty((V, 4, 9, 2))
    chunk = 1 << 18
    flat = out.reshape(V * 4, 9, 2)
    for i, s in enumerate(range(0, V * 4, chunk)):
        n = min(chunk, V * 4 - s)
        flat[s:s + n] = random_su3(n, seed * 1000 + i)
    return out


def near_unit_gauge(V, eps, seed):
    """SU(3) links exp(i eps H) with Gaussian Hermitian traceless H (smooth, solvable at m0 ~ -0.1...-0.5)"""
    rng = np.random.default_rng(seed)
    n = V * 4
    a = rng.standard_normal((n, 3, 3)) + 1j * rng.standard_normal((n, 3, 3))
    h = (a + a.conj().transpose(0, 2, 1)) / 2
    h -= np.trace(h, axis1=1, axis2=2)[:, None, None] * np.eye(3) / 3
    w, v = np.linalg.eigh(h)
    u = (v * np.exp(1j * eps * w)[:, None, :]) @ v.conj().transpose(0, 2, 1)
    out = np.empty((n, 9, 2)); out[..., 0] = u.reshape(n, 9).real; out[..., 1] = u.reshape(n, 9).imag
    return out.reshape(V, 4, 9, 2)


def write_conf(path, L, U, plaq):
    """gauge file in the reference's format (src/io.c:489-520): 4 x int32 (T,Z,Y,X), double plaquette, links"""
    with open(path, "wb") as f:
        f.write(np.asarray(L, dtype="<i4").tobytes()); f.write(np.asarray([plaq], dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(U, dtype="<f8").tobytes())


def cpu_baseline_reference(threads):
    """the REAL reference (oracle/_ref/dd_alpha_amg_sse, SSE build) on a bounded sample: pure GMRES
    (method 0, mixed precision 2) on a 16^4 random-gauge lattice; the fp32 d_plus_clover time is read from
    the reference's own profiling counters (self coupling + neighbor coupling, src/init_generic.c:58-61)"""
    import subprocess, tempfile, re
    exe = os.path.join(REPO, "oracle", "_ref", "dd_alpha_amg_sse")
    if not os.path.exists(exe):
        return None
    Lr = [16, 16, 16, 16]; Vr = int(np.prod(Lr))
    tmp = tempfile.mkdtemp(prefix="ddamg_cpu_")
    try:
        U = synth_gauge(Vr, 777)
        write_conf(os.path.join(tmp, "conf"), Lr, U, 0.0)
        ini = f"""configuration: {tmp}/conf
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: 1
number of openmp threads: {threads}
d0 global lattice: 16 16 16 16
d0 local lattice: 16 16 16 16
d0 block lattice: 4 4 4 4
m0: -0.1
csw: 1.0
tolerance for relative residual: 1E-30
iterations between restarts: 50
maximum of restarts: 4
print mode: 1
method: 0
mixed precision: 2
randomize test vectors: 0
"""
        open(os.path.join(tmp, "b.ini"), "w").write(ini)
        out = subprocess.run([exe, os.path.join(tmp, "b.ini")], capture_output=True, text=True, timeout=240, cwd=tmp).stdout
        m1 = re.search(r"self coupling, float:\s*([0-9.e+-]+)\(\s*(\d+)\)", out)
        m2 = re.search(r"neighbor coupling, float:\s*([0-9.e+-]+)\(\s*(\d+)\)", out)
        if not (m1 and m2):
            return None
        t = float(m1.group(1)) + float(m2.group(1)); n = int(m2.group(2))
        return {"value": FLOP_PER_SITE * Vr * n / t / 1e9, "unit": "GFLOP/s", "cores": threads, "kind": "reference",
                "sample": f"{n} d_plus_clover_float applies inside the reference's pure-GMRES run (SSE build, {threads} OpenMP threads, "
                          f"16^4 random gauge), {t / n * 1e3:.2f} ms/apply from its own profiling counters"}
    except Exception:
        return None
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(L, D, cl, phi, budget_s=12.0):
    """CPU baseline on the host cores of this box: the real reference when oracle/_ref runs here, else the
    oracle port (bounded sample of the same workload)"""
    from oracle import orc
    ref = cpu_baseline_reference(orc.host_threads())
    if ref is not None:
        return ref
    t1, nt = orc.dirac_time_f32(L, D, cl, phi, 1)
    reps = int(max(2, min(200, budget_s / max(t1, 1e-4))))
    t, nt = orc.dirac_time_f32(L, D, cl, phi, reps)
    V = int(np.prod(L))
    return {"value": FLOP_PER_SITE * V / t / 1e9, "unit": "GFLOP/s", "cores": nt, "kind": "port",
            "sample": f"{reps} fp32 applies of the same {'x'.join(map(str, L))} operator, OpenMP over sites, "
                      f"{t*1e3:.2f} ms/apply"}


def pmc_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, calibrated as MI355X_MICROARCH.md prescribes);
    measured on this same workload (32^4 fp32), None for any other"""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "r01_traffic.json")))
        return d["dirac_apply_lds_kernel<float>"]["bytes_per_launch"] if precision == 32 else None
    except Exception:
        return None


def small_lattice_leg(device):
    """BASELINE configs[1]: the reference's own 8^4 sample configuration (gauge field from tests/golden), fine operator
    only, fp32, 1000 timed applies after 50 warm-ups.  Its 3 MB working set lives in the caches, so it is reported
    next to the headline and not used for the HBM roofline."""
    import ddalphaamg_amd as dd
    from ddalphaamg_amd import api
    g = np.load(os.path.join(REPO, "tests", "golden", "ref_8x8_dirac.npz"))
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4
    p.m0, p.csw, p.device = float(g["meta_f64"][0]), float(g["meta_f64"][1]), device
    ctx = dd.Context(p)
    ctx.set_gauge(g["gauge"], anti_pbc=True)
    x = ctx.vector(0, 32).upload(g["dirac_in"]); y = ctx.vector(0, 32)
    for _ in range(50):
        ctx.dirac_apply(y, x)
    ctx.sync(); ctx.timer_begin()
    for _ in range(1000):
        ctx.dirac_apply(y, x)
    us = ctx.timer_end()   # milliseconds for 1000 applies == microseconds per apply
    err = float(np.linalg.norm(y.download() - g["dirac_out_f32_as_f64"]) / np.linalg.norm(g["dirac_out_f32_as_f64"]))
    ctx.close()
    return {"workload": "8^4 conf/8x8x8x8b6.0000id3n1 (BASELINE configs[1]), fine Wilson-Clover apply, fp32, cache-resident",
            "us_per_apply": us, "gflops": FLOP_PER_SITE * 4096 / us / 1e3, "algorithmic_gbs": BYTES_PER_SITE_F32 * 4096 / us / 1e3,
            "rel_err_vs_reference_output": err}


def solve_leg(ctx_params, U, V, L, world=1, rank=0, transport="rccl", group=None):
    """secondary measurement: two-level FGMRES+AMG solve (BASELINE config 3) on the same gauge field; on N GPUs the
    global lattice is N times larger (process grid as in the headline measurement), operator as described there"""
    import ddalphaamg_amd as dd
    p = ctx_params
    if world == 1:
        ctx = dd.Context(p)
        ctx.set_gauge(U, anti_pbc=True)
    else:
        from ddalphaamg_amd import dist as ddist
        grid = ddist.process_grid_for(world); coords = ddist.coords_of(rank, grid)
        for mu in range(4):
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        ctx = dd.Context(p)
        ddist.attach_host(ctx, group)
        ctx.set_gauge(U, anti_pbc=True)     # global clover term: neighbours' links over the host transport
        if transport == "rccl":
            ddist.attach_rccl(ctx, rank)
    t0 = time.perf_counter(); ctx.setup(p.setup_iter[0]); ctx.sync(); t_setup = time.perf_counter() - t0
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    t0 = time.perf_counter(); x, it, cit, rr = ctx.solve(b, 1e-10, out=x); t_host = time.perf_counter() - t0   # host arrays of a caller that keeps its vectors
    bv = ctx.vector(0, 64).upload(b); xv = ctx.vector(0, 64)
    ctx.solve_vec(xv, bv, 1e-10)
    t0 = time.perf_counter(); it, cit, rr = ctx.solve_vec(xv, bv, 1e-10); t_solve = time.perf_counter() - t0
    ctx.close()
    return {"workload": f"{'x'.join(map(str, L))} per GPU x {world} GPU(s), near-unit gauge exp(0.35 i H), m0 -0.3, 2-level AMG (4^4 blocks/aggregates, Nvec 24, SAP 2x4, coarse tol 5e-2), "
                        "FGMRES(50) to 1e-10, rhs=ones",
            "seconds_per_solve": t_solve, "seconds_per_solve_host_vectors": t_host, "setup_seconds": t_setup, "iterations": it, "coarse_iterations": cit, "true_relres": rr}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32])
    ap.add_argument("--global-lattice", type=int, nargs=4, default=None,
                    help="strong scaling: fixed global lattice divided over the process grid (overrides --lattice, which is "
                         "the per-GPU lattice of the default weak-scaling run)")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--small-lattice", action="store_true",
                    help="also time BASELINE configs[1] (the reference's 8^4 configuration, cache-resident); off by default so that "
                         "a kernel profile of the default run holds only launches of the headline workload")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="halo transport for --gpus > 1; 'host' (gloo, staged through pinned memory) lets several "
                         "processes share one card for a rehearsal and is never the reported configuration")
    ap.add_argument("--self-exchange", default=None,
                    help="single GPU only, e.g. -1,-1,-1,1: run the fine operator through the multi-GPU machinery with the process "
                         "as its own neighbour in the directions marked -1 (RCCL transport): cost of pack + exchange + "
                         "interior/boundary split, not a reported configuration")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.transport == "rccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
            local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    import ddalphaamg_amd as dd
    from ddalphaamg_amd import api
    from conftest import splitmix_uniform

    strong = args.global_lattice is not None
    if strong:
        from ddalphaamg_amd import dist as _d
        _g = _d.process_grid_for(world)
        if any(args.global_lattice[mu] % (_g[mu] * 8) for mu in range(4)):
            raise SystemExit("--global-lattice: every extent must be a multiple of 8 x the process grid " + str(_g))
        args.lattice = [args.global_lattice[mu] // _g[mu] for mu in range(4)]
    L = list(args.lattice); V = int(np.prod(L))
    p = api.default_params()
    p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw, p.device = -0.1, 1.0, local_rank
    U = synth_gauge(V, 20260101 + rank)
    phi = splitmix_uniform(V * 24, 1234 + rank).reshape(V, 12, 2)
    halo_check = None
    if world == 1:
        if args.self_exchange:
            for mu, v in enumerate(int(x) for x in args.self_exchange.split(",")):
                p.process_grid[mu] = v
        ctx = dd.Context(p)
        if args.self_exchange:
            ctx.comm_init_rccl(api.rccl_unique_id())
        ctx.set_gauge(U, anti_pbc=True)
        grid = [1, 1, 1, 1]
    else:
        # domain decomposition: one process per GPU on a Cartesian grid, `--lattice` sites per GPU (weak
        # scaling), halo exchange over RCCL.  Every process draws the random links of its own part; the clover
        # term is built on the global field (neighbours' links fetched over the host transport).
        from ddalphaamg_amd import dist as ddist
        grid = ddist.process_grid_for(world)
        coords = ddist.coords_of(rank, grid)
        for mu in range(4):
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        ctx = dd.Context(p)
        gloo = dist.new_group(backend="gloo")
        ddist.attach_host(ctx, gloo)
        ctx.set_gauge(U, anti_pbc=True)
    x = ctx.vector(0, args.precision).upload(phi)
    y = ctx.vector(0, args.precision)
    if world > 1:
        # the same exchange through the host transport (gloo) first, as a cross-check of the RCCL path
        ctx.dirac_apply(y, x)
        y_host = y.download()
        if args.transport == "rccl":
            ddist.attach_rccl(ctx, rank)
            ctx.dirac_apply(y, x)
            halo_check = float(np.abs(y.download() - y_host).max())
            if not (halo_check == 0.0 and np.isfinite(y_host).all()):
                raise RuntimeError(f"rank {rank}: RCCL halo exchange differs from the host transport by {halo_check}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.dirac_apply(y, x)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(args.steps):
        ctx.dirac_apply(y, x)
    ev_ms = ctx.timer_end()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if args.transport == "rccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        bytes_site = BYTES_PER_SITE_F32 * (args.precision // 32)
        launch_s = ev_ms * 1e-3 / args.steps
        achieved = bytes_site * V / launch_s / 1e9
        out = {
            "metric": "fine_wilson_clover_gflops", "value": FLOP_PER_SITE * V * world * args.steps / dt / 1e9,
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": f"f{args.precision}", "data": "synthetic",
            "config": {"workload": f"fine Wilson-Clover apply (d_plus_clover), {'x'.join(map(str, L))} local lattice per GPU, "
                                   "random SU(3) gauge, csw=1.0, anti-periodic T",
                       "flop_per_site": FLOP_PER_SITE,
                       "parallelism": ("domain decomposition, process grid " + "x".join(map(str, grid)) + " (T,Z,Y,X), " + args.transport.upper() + " halo exchange "
                                       "overlapped with the interior tiles") if world > 1 else ("single" if not args.self_exchange else "single GPU, self-exchange " + args.self_exchange + " through RCCL"),
                       "halo_check_vs_host_transport": halo_check},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.precision),
                         "kernel": "dirac_apply_lds_kernel<float>" if args.precision == 32 else "dirac_apply_lds_kernel<double>", "us_per_launch": launch_s * 1e6,
                         "algorithmic_bytes_per_site": bytes_site},
        }
        if not args.no_cpu_baseline:
            D, cl = ctx.get_operator()
            if world == 1:   # "on rank 0 at N=1 only"
                out["cpu_baseline"] = cpu_baseline(L, D, cl, phi)
        if world == 1 and not args.self_exchange and args.small_lattice:
            try:
                out["small_lattice"] = small_lattice_leg(local_rank)
            except Exception as e:
                out["small_lattice"] = {"error": str(e)[:200]}
    ctx.close()
    if not args.no_solve and all(x % 8 == 0 for x in L):
        q = api.default_params(); q.num_levels = 2
        for mu in range(4):
            q.local_lattice[0][mu] = L[mu]; q.block_lattice[0][mu] = 4; q.local_lattice[1][mu] = L[mu] // 4
        q.num_vect[0] = 24; q.post_smooth_iter[0] = 2; q.block_iter[0] = 4; q.setup_iter[0] = 4
        q.restart, q.max_restart, q.tol = 50, 20, 1e-10
        q.coarse_iter, q.coarse_restart, q.coarse_tol = 100, 5, 5e-2
        q.mixed_precision, q.method, q.odd_even = 1, 2, 1
        q.m0, q.csw, q.device = -0.3, 1.0, local_rank
        q.test_vector_rng, q.rng_seed = 1, 20260101     # device generator for the random test vectors
        U = near_unit_gauge(V, 0.35, 20260101 + rank)   # smooth links: a system on which the multigrid has work to do
        # the headline number must survive the secondary leg: an exception is recorded, and on several GPUs a
        # watchdog prints the headline line and ends the process if the leg does not come back (a process that
        # failed alone would leave the others waiting in a collective)
        import threading

        def give_up():
            if rank == 0:
                out["solve"] = {"error": "distributed solve leg did not finish within 400 s"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(400.0, give_up) if world > 1 else None
        if dog:
            dog.daemon = True; dog.start()
        try:
            res = solve_leg(q, U, V, L, world, rank, args.transport, gloo if world > 1 else None)
        except Exception as e:
            res = {"error": str(e)[:200]}
        if dog:
            dog.cancel()
        if rank == 0:
            out["solve"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
