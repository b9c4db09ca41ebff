#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DDalphaAMG hot path.

  python bench.py --gpus N --steps K --warmup W

Headline (`metric`, `value`, `roofline`): a "step" is one fine Wilson-Clover apply (d_plus_clover_float, reference
src/dirac_generic.c:159-277) over a 32^4 local lattice per GPU of synthetic random SU(3) links (BASELINE.json quotes the
>= 40 %-of-HBM target on exactly this volume); inputs are resident in HBM before the timed region; on N GPUs every rank
holds 32^4 sites of a decomposed lattice (weak scaling, halo exchange over RCCL overlapped with the interior tiles).

Secondary objects of the same JSON line:
  solve           (N = 1) BASELINE configs[2]: two-level FGMRES+AMG on 32^4, next to the reference's own run of the same case
  strong_scaling  BASELINE configs[4] / SURVEY 8(d) config 5: ONE global 64^4 lattice, 3-level AMG, divided over process
                  grids 1 / 2 (T) / 4 (T,Z) / 8 (T,Z,Y): same global gauge field and right-hand side at every N; seconds per
                  solve, iterations, and the speed-up against the committed N = 1 time
  cpu_baseline    (N = 1) the REAL reference (oracle/_ref, SSE build) on the host cores of this box: its fp32 d_plus_clover
                  on the headline volume, and a bounded sample of the solve

N > 1 without a launcher: `python bench.py --gpus N` starts the N ranks itself (one child process per GPU, before anything
touches the GPU in the parent); under torchrun (WORLD_SIZE set) it is one of the ranks and checks N against the world size.
"""
import argparse, hashlib, json, os, subprocess, sys, time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))

FLOP_PER_SITE = 1920          # reference model, src/init_generic.c:59,61
BYTES_PER_SITE_F32 = 816      # 4 B x (24 in + 24 out + 72 links + 84 clover reals), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
GAUGE_SEED, GAUGE_EPS = 20260101, 0.35   # near-unit links exp(i eps H): a system on which the multigrid has work to do


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32], help="local lattice per GPU of the headline measurement")
    ap.add_argument("--strong-lattice", type=int, nargs=4, default=[64, 64, 64, 64],
                    help="global lattice of the strong-scaling solve (BASELINE configs[4]); every extent a multiple of 8 x the process grid")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the 32^4 two-level solve (N = 1)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling solve")
    ap.add_argument("--small-lattice", action="store_true",
                    help="also time BASELINE configs[1] (the reference's 8^4 configuration, cache-resident)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="halo transport for N > 1; 'host' (gloo, staged through pinned memory) lets several processes share one "
                         "card for a rehearsal and is never the reported configuration")
    ap.add_argument("--self-exchange", default=None,
                    help="single GPU only, e.g. -1,-1,-1,1: run the fine operator through the multi-GPU machinery with the process "
                         "as its own neighbour in the directions marked -1 (RCCL transport); not a reported configuration")
    ap.add_argument("--rehearse", type=int, default=8, choices=[0, 2, 4, 8],
                    help="N = 1 only: also run the PER-GPU problem of the N-GPU strong-scaling decomposition on this one GPU, every split "
                         "direction through the RCCL self-exchange (0: skip); printed as `rehearsal` next to `strong_scaling`")
    ap.add_argument("--leg-timeout", type=float, default=900.0, help="watchdog for the solve legs on N > 1 (seconds)")
    return ap.parse_args()


# ---- N > 1 without a launcher: the parent only starts and reaps the ranks ------------------------------------------
def spawn_ranks(n):
    """one child process per GPU; the parent imports neither torch nor the library and never touches a GPU"""
    import tempfile
    # rendezvous through a file the parent owns (torch's file:// store): no port is probed and then bound by somebody else
    rdzv_dir = tempfile.mkdtemp(prefix="ddamg_bench_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   DDAMG_BENCH_RDZV="file://" + os.path.join(rdzv_dir, "store"))
        env.pop("MASTER_PORT", None)
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 3300
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0:
                print(f"bench.py: rank {procs.index(p)} ended with exit code {c}", file=sys.stderr, flush=True)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
        if (rc != 0 or time.time() > deadline) and live:
            # a rank failed (or the job overran): the others would wait in a collective for ever
            for p in live:
                p.terminate()
            t_end = time.time() + 10
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            if rc == 0:
                rc = 3
            break
    import shutil
    shutil.rmtree(rdzv_dir, ignore_errors=True)
    return rc


# ---- inputs ---------------------------------------------------------------------------------------------------------
def synth_gauge_random(V, seed):
    """random SU(3) links of the headline measurement (conftest.random_su3, numpy)"""
    import numpy as np
    from conftest import random_su3
    out = np.empty((V, 4, 9, 2))
    chunk = 1 << 18
    flat = out.reshape(V * 4, 9, 2)
    for i, s in enumerate(range(0, V * 4, chunk)):
        n = min(chunk, V * 4 - s)
        flat[s:s + n] = random_su3(n, seed * 1000 + i)
    return out


def near_unit_gauge(V, eps, seed):
    """SU(3) links exp(i eps H) with Gaussian Hermitian traceless H (numpy form, kept for the golden-fixture generators)"""
    import numpy as np
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
    import synth
    synth.write_conf(path, L, U, plaq)


# ---- CPU baseline: the real reference on this box's host cores -------------------------------------------------------
def _run_reference(ini_text, tmp, timeout):
    exe = os.path.join(REPO, "oracle", "_ref", "dd_alpha_amg_sse")
    open(os.path.join(tmp, "b.ini"), "w").write(ini_text)
    return subprocess.run([exe, os.path.join(tmp, "b.ini")], capture_output=True, text=True, timeout=timeout, cwd=tmp).stdout


def cpu_baseline_reference(threads, L=(32, 32, 32, 32)):
    """the REAL reference (oracle/_ref/dd_alpha_amg_sse) on the headline volume: pure GMRES (method 0, mixed precision 2),
    one restart cycle of 25 on the same near-unit gauge field as the solve leg; the fp32 d_plus_clover time is read from the
    reference's own profiling counters (self coupling + neighbor coupling, src/init_generic.c:58-61)"""
    import re, tempfile, shutil
    import numpy as np
    import synth
    if not os.path.exists(os.path.join(REPO, "oracle", "_ref", "dd_alpha_amg_sse")):
        return None
    Lr = list(L); Vr = int(np.prod(Lr)); Ls = " ".join(map(str, Lr))
    tmp = tempfile.mkdtemp(prefix="ddamg_cpu_")
    try:
        synth.write_conf(os.path.join(tmp, "conf"), Lr, synth.synth_gauge(Lr, GAUGE_EPS, GAUGE_SEED), 0.0)
        out = _run_reference(f"""configuration: {tmp}/conf
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: 1
number of openmp threads: {threads}
d0 global lattice: {Ls}
d0 local lattice: {Ls}
d0 block lattice: 4 4 4 4
m0: -0.1
csw: 1.0
tolerance for relative residual: 1E-30
iterations between restarts: 25
maximum of restarts: 1
print mode: 1
method: 0
mixed precision: 2
randomize test vectors: 0
""", tmp, 400)
        m1 = re.search(r"self coupling, float:\s*([0-9.e+-]+)\(\s*(\d+)\)", out)
        m2 = re.search(r"neighbor coupling, float:\s*([0-9.e+-]+)\(\s*(\d+)\)", out)
        if not (m1 and m2):
            return None
        t = float(m1.group(1)) + float(m2.group(1)); n = int(m2.group(2))
        return {"value": FLOP_PER_SITE * Vr * n / t / 1e9, "unit": "GFLOP/s", "cores": threads, "kind": "reference",
                "sample": f"{n} d_plus_clover_float applies inside the reference's pure-GMRES run (SSE build, {threads} OpenMP threads, "
                          f"{'x'.join(map(str, Lr))} lattice = the headline volume), {t / n * 1e3:.2f} ms/apply from its own profiling counters"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline_solve(threads):
    """the reference's FGMRES+AMG solve beside ours: a bounded live sample (16^4, same parameters and gauge generator as the
    solve leg, 1/16 of its volume) on this box's cores, and the committed full-size run (tests/golden/ref_32x32_2lvl.json:
    the same 32^4 gauge field as the `solve` leg, run in the build container by oracle/run_reference_big.py)"""
    import re, tempfile, shutil
    import synth
    out = {}
    try:
        g = json.load(open(os.path.join(REPO, "tests", "golden", "ref_32x32_2lvl.json")))
        out["reference_32"] = {"seconds": g["solve_seconds"], "iterations": g["iterations"], "setup_seconds": sum(g["setup_seconds"]),
                               "cores": g["threads"], "kind": "reference", "where": "build container, " + g["host"],
                               "true_relres": g["true_relres"]}
    except Exception:
        pass
    if not os.path.exists(os.path.join(REPO, "oracle", "_ref", "dd_alpha_amg_sse")):
        return out or None
    tmp = tempfile.mkdtemp(prefix="ddamg_cpu_")
    try:
        Lr = [16] * 4
        synth.write_conf(os.path.join(tmp, "conf"), Lr, synth.synth_gauge(Lr, GAUGE_EPS, GAUGE_SEED), 0.0)
        t0 = time.time()
        log = _run_reference(f"""configuration: {tmp}/conf
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: 2
number of openmp threads: {threads}
d0 global lattice: 16 16 16 16
d0 local lattice: 16 16 16 16
d0 block lattice: 4 4 4 4
d0 post smooth iter: 2
d0 block iter: 4
d0 test vectors: 24
d0 setup iter: 4
d1 global lattice: 4 4 4 4
d1 local lattice: 4 4 4 4
m0: -0.3
csw: 1.0
tolerance for relative residual: 1E-10
iterations between restarts: 50
maximum of restarts: 20
coarse grid tolerance: 5E-2
coarse grid iterations: 100
coarse grid restarts: 5
print mode: 1
method: 2
odd even preconditioning: 1
mixed precision: 1
randomize test vectors: 0
""", tmp, 600)
        wall = time.time() - t0
        it = re.search(r"FGMRES iterations:\s*(\d+)", log)
        ts = re.findall(r"elapsed wall clock time:\s*([0-9.]+)\s+seconds", log)
        st = re.findall(r"elapsed time: ([0-9.]+) seconds", log)
        if it and ts:
            out.update({"seconds": float(ts[-1]), "iterations": int(it.group(1)), "setup_seconds": sum(float(x) for x in st),
                        "cores": threads, "kind": "reference",
                        "sample": f"16^4 lattice (1/16 of the solve leg's volume), same generator, parameters and right-hand side; "
                                  f"SSE build, {threads} OpenMP threads, {wall:.1f} s wall for setup + solve"})
    except Exception as e:
        out["error"] = str(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out or None


def cpu_baseline(L, D, cl, phi, budget_s=12.0):
    """CPU baseline on the host cores of this box: the real reference when oracle/_ref runs here, else the oracle port"""
    import numpy as np
    from oracle import orc
    threads = orc.host_threads()
    ref = cpu_baseline_reference(threads, L) if all(x % 4 == 0 for x in L) else None
    if ref is None:
        t1, nt = orc.dirac_time_f32(L, D, cl, phi, 1)
        reps = int(max(2, min(200, budget_s / max(t1, 1e-4))))
        t, nt = orc.dirac_time_f32(L, D, cl, phi, reps)
        V = int(np.prod(L))
        ref = {"value": FLOP_PER_SITE * V / t / 1e9, "unit": "GFLOP/s", "cores": nt, "kind": "port",
               "sample": f"{reps} fp32 applies of the same {'x'.join(map(str, L))} operator, OpenMP over sites, {t*1e3:.2f} ms/apply"}
    return ref


# ---- roofline.traffic: a measured constant with its provenance --------------------------------------------------------
def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("fine_op.hip", "dirac_device.h", "common.h"):
        h.update(open(os.path.join(REPO, "ddalphaamg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def mfma_source_hash():
    h = hashlib.sha256()
    for f in ("coarse_lockstep.hip", "coarse_lockstep.h", "coarse_multi.hip", "coarse_multi.h", "mfma_tile.h", "coarse_batch.hip", "coarse_op.h", "transfer.hip"):
        h.update(open(os.path.join(REPO, "ddalphaamg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (2*FETCH_SIZE + WRITE_SIZE,
    calibrated as MI355X_MICROARCH.md prescribes).  The file records the hash of the kernel sources it was measured on; when
    the kernel has changed since, the number is stale and null is reported instead."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            e = d["dirac_apply_lds_kernel<float>"]
            if precision != 32:
                return None, None
            src = d.get("kernel_source_sha16")
            if src is not None and src != kernel_source_hash():
                return None, f"profiles/{name} was measured on kernel sources {src}, current {kernel_source_hash()}"
            return e["bytes_per_launch"], f"profiles/{name}, commit {d.get('commit', 'unrecorded')}, kernel sources {src or 'unrecorded'}"
        except Exception:
            continue
    return None, None


def coarse_operator_report():
    """what the north star asks to see next to the solve: the coarse operator against its roofline (one right-hand side: the
    solve path) and the matrix-core utilisation where the coarse operator is applied to many right-hand sides (the bootstrap's
    coarsest-level solves in lockstep, the Galerkin construction).  From the committed profiles, not measured in this run: the
    MFMA figures carry the hash of the kernel sources they were measured on and are reported as null once those changed."""
    import csv
    out = {}
    for R in ("r04", "r03", "r02"):
        try:
            rows = list(csv.DictReader(open(os.path.join(REPO, "profiles", R + "_solve32_kernel_stats.csv"))))
            r = [x for x in rows if "coarse_site_kernel<float, 6, 1>" in x["Name"]][0]     # <T, n/8, MODE_HOP>
            us = float(r["AverageUs"]); n = 48; sites = 8 ** 4 // 2
            byts = sites * 8 * n * n * 8        # a half hopping term reads 4 own and 4 neighbours' links per site, 8 B per complex
            out["solve_path"] = {"kernel": "coarse_site_kernel, hopping-term instantiation (one right-hand side: VALU tile GEMV, arithmetic intensity ~2 flop/B)",
                                 "us_per_half_hopping_term_8^4_n48": us, "GB/s": byts / us / 1e3, "frac_of_hbm_peak": byts / us / 1e3 / 8000.0,
                                 "note": "302 MB of couplings per launch (the whole 8^4 coarse operator: a half hopping term uses every link once)",
                                 "source": "profiles/" + R + "_solve32_kernel_stats.csv"}
            break
        except Exception:
            continue
    for name in ("r04_mfma_busy.json", "r03_mfma_busy.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
        except Exception:
            continue
        stale = d.get("kernel_source_sha16") != mfma_source_hash()
        def entry(key, what):
            e = d.get(key, {})
            return {"kernel": what, "mfma_busy": None if (stale or "mfma_busy" not in e) else e["mfma_busy"], "source": e.get("source"),
                    "stale": (f"measured on kernel sources {d.get('kernel_source_sha16')}, current {mfma_source_hash()}" if stale else None)}
        out["multi_rhs"] = {
            "formula": d.get("formula"),
            "level1_schwarz_block_solve": entry("level1_block_minres", "cm_block_minres_op_kernel: the Schwarz block solve of the intermediate level for all test vectors of a bootstrap "
                                                "iteration at once -- per coupling a complex 48 x 48 times 48 x 32 product on v_mfma_f32_16x16x4_f32, per-column MinRes "
                                                "coefficients (64^4 three-level setup, level 1 = 16^4)"),
            "level1_operator": entry("level1_apply", "cm_apply_op_kernel: the operator of the intermediate level for all columns of the K-cycles in lockstep (64^4 three-level setup)"),
            "level1_restrict": entry("level1_restrict", "cm_restrict_kernel: level 1 -> 2 for all columns"),
            "level1_interpolate": entry("level1_interpolate", "cm_interpolate_kernel: level 2 -> 1 for all columns"),
            "bootstrap_coarsest_solves_in_lockstep": entry("lockstep_hop", "ls_hop_op_kernel: hopping terms of the coarsest-level Schur complement for all Nvec test vectors of a "
                                                           "bootstrap iteration at once, complex n x n times n x 32 on v_mfma_f32_16x16x4_f32"),
            "galerkin_coarse_apply": entry("galerkin_coarse_apply", "coarse_batch_apply_kernel: all 2*Nvec columns of the coarse-level Galerkin construction"),
            "galerkin_restrict": entry("galerkin_restrict", "restrict_mfma_kernel<2>: the 2*Nvec columns of the fine-level Galerkin construction, five parts each (the four forward "
                                       "parts on the aggregate faces only), times 24 vectors per aggregate"),
            "galerkin_coarse_restrict": entry("galerkin_coarse_restrict", "coarse_batch_restrict_store_mfma_kernel: the coarse level's five batches times its 28 vectors per "
                                              "aggregate, conj(P) staged in LDS"),
        }
        break
    return out or None


def small_lattice_leg(device):
    """BASELINE configs[1]: the reference's own 8^4 sample configuration (gauge field from tests/golden), fine operator
    only, fp32, 1000 timed applies after 50 warm-ups.  Its 3 MB working set lives in the caches, so it is reported
    next to the headline and not used for the HBM roofline."""
    import numpy as np
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


# ---- solve legs ---------------------------------------------------------------------------------------------------------
def amg_params(api, Lloc, levels, device):
    q = api.default_params(); q.num_levels = levels
    for mu in range(4):
        q.local_lattice[0][mu] = Lloc[mu]; q.block_lattice[0][mu] = 4; q.local_lattice[1][mu] = Lloc[mu] // 4
        if levels == 3:
            q.block_lattice[1][mu] = 2; q.local_lattice[2][mu] = Lloc[mu] // 8
    q.num_vect[0] = 24; q.post_smooth_iter[0] = 2; q.block_iter[0] = 4; q.setup_iter[0] = 4
    q.num_vect[1] = 28; q.post_smooth_iter[1] = 2; q.block_iter[1] = 4; q.setup_iter[1] = 2
    q.restart, q.max_restart, q.tol = 50, 20, 1e-10
    q.coarse_iter, q.coarse_restart, q.coarse_tol = 100, 5, 5e-2
    q.mixed_precision, q.method, q.odd_even = 1, 2, 1
    q.m0, q.csw, q.device = -0.3, 1.0, device
    q.test_vector_rng, q.rng_seed = 1, 20260101     # device generator for the random test vectors
    return q


def run_solve(q, G, grid, coords, world, rank, transport, group):
    """set-up + solve of one decomposed (or undivided) lattice: global lattice G over `grid`, this process at `coords`"""
    import numpy as np
    import synth
    import ddalphaamg_amd as dd
    for mu in range(4):
        q.process_grid[mu] = grid[mu]; q.process_coords[mu] = coords[mu]
    V = int(np.prod([G[mu] // max(1, grid[mu]) for mu in range(4)]))
    t0 = time.perf_counter()
    U = synth.synth_gauge(G, GAUGE_EPS, GAUGE_SEED, [max(1, g) for g in grid], coords)
    t_gauge = time.perf_counter() - t0
    ctx = dd.Context(q)
    if world > 1:
        from ddalphaamg_amd import dist as ddist
        ddist.attach_host(ctx, group)
    elif any(g == -1 for g in grid):
        from ddalphaamg_amd import api as _api
        ctx.comm_init_rccl(_api.rccl_unique_id())     # one GPU, the process its own neighbour: everything through RCCL
    ctx.set_gauge(U, anti_pbc=True)     # global clover term: on a process grid the neighbours' links travel over the host transport
    del U
    if world > 1 and transport == "rccl":
        ddist.attach_rccl(ctx, rank)
    t0 = time.perf_counter(); ctx.setup(q.setup_iter[0]); ctx.sync(); t_setup = time.perf_counter() - t0
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    bv = ctx.vector(0, 64).upload(b); xv = ctx.vector(0, 64)
    ctx.solve_vec(xv, bv, 1e-10)                                    # warm-up
    on_grid = world > 1 or any(g == -1 for g in grid)
    if on_grid:
        ctx.comm_stats(reset=True)
    t0 = time.perf_counter(); it, cit, rr = ctx.solve_vec(xv, bv, 1e-10); t_solve = time.perf_counter() - t0
    res = {"seconds_per_solve": t_solve, "setup_seconds": t_setup, "iterations": it, "coarse_iterations": cit, "true_relres": rr,
           "gauge_generation_seconds": t_gauge}
    if on_grid:
        res["messages"] = describe_messages(ctx.comm_stats(), q, it, cit)
    if world == 1 and not any(g == -1 for g in grid):
        x, it2, _, _ = ctx.solve(b, 1e-10)
        t0 = time.perf_counter(); ctx.solve(b, 1e-10, out=x); res["seconds_per_solve_host_vectors"] = time.perf_counter() - t0
    if world == 1 and not any(g == -1 for g in grid):
        # The same setup once more in this context (what every later setup of a process costs, e.g. one per HMC trajectory):
        # `setup_seconds` above is the first setup of the context and, on a freshly started box, also pays for device memory no
        # process has allocated before (13-30 ms per GB, docs/design/09_rounds_2_3.md) -- the two are reported side by side.
        try:
            t0 = time.perf_counter(); ctx.setup(q.setup_iter[0]); ctx.sync(); res["setup_seconds_repeated"] = time.perf_counter() - t0
            it3, cit3, rr3 = ctx.solve_vec(xv, bv, 1e-10)
            res["repeated_setup_solve"] = {"iterations": it3, "coarse_iterations": cit3, "true_relres": rr3}
        except Exception as e:
            res["setup_seconds_repeated"] = None; res["repeated_setup_solve"] = {"error": str(e)[:200]}
    if world == 1:
        # seconds per GMRES iteration of the coarsest-level solve (for the rehearsal's correction: on N GPUs the gathered
        # coarsest level is the GLOBAL one, the rehearsal's is 1/N of it)
        try:
            lc = q.num_levels - 1
            n = ctx.ndof(lc); Vc = ctx.volume(lc)
            bc = np.zeros((Vc, n, 2)); bc[..., 0] = 1.0
            bcv = ctx.vector(lc, 32).upload(bc); xcv = ctx.vector(lc, 32)
            ctx.coarse_solve(xcv, bcv); ctx.sync()
            t0 = time.perf_counter(); cits = sum(ctx.coarse_solve(xcv, bcv) for _ in range(5)); ctx.sync()
            res["coarsest"] = {"sites": Vc, "dof_per_site": n, "seconds_per_iteration": (time.perf_counter() - t0) / max(1, cits)}
        except Exception as e:
            res["coarsest"] = {"error": str(e)[:200]}
    ctx.close()
    return res


def describe_messages(st, q, iterations, coarse_iterations):
    """what this rank sent during the timed solve (ddamg_hip_comm_stats), payload by payload, next to the model of DESIGN
    (multi-GPU, 'what travels between the GPUs'): per outer iteration the fp64 residual exchanges once, the fp32 fine level once per
    operator application and once per colour sweep of the Schwarz smoother (4 per smoother call), level 1 once per hopping term of
    the K-cycle's operator and smoother; two global sums per Arnoldi step of the outer solver and of the K-cycle; one all-gather per
    coarsest solve when that level is gathered"""
    names = {48: "level 0 fp32: half spinor, 6 complex per face site", 96: "level 0 fp64: half spinor, 6 complex per face site"}
    for l in range(1, q.num_levels):
        n = 2 * q.num_vect[l - 1]
        names[8 * n] = f"level {l} fp32: {n} complex per face site"
        names[8 * n * 64] = f"level {l} setup: {n} x 64 complex per face site (batched Galerkin construction)"
    out = {"transport": st.get("transport"), "halo_exchanges": []}
    for e in st.get("halo_exchanges", []):
        e = dict(e); e["what"] = names.get(e["bytes_per_face_site"], "?")
        e["exchanges_per_outer_iteration"] = e["exchanges"] / max(1, iterations)
        e["bytes_per_message"] = e["bytes_sent"] / max(1, e["messages"])
        out["halo_exchanges"].append(e)
    for k in ("allreduce", "allgather"):
        if k in st:
            out[k] = dict(st[k])
    if "allreduce" in out:
        out["allreduce"]["calls_per_outer_iteration"] = out["allreduce"]["calls"] / max(1, iterations)
    if "allgather" in out:
        out["allgather"]["calls_per_coarsest_solve_model"] = 1
        out["allgather"]["coarsest_iterations"] = coarse_iterations
    return out


def committed_n1_strong(G):
    """seconds per solve of the strong-scaling configuration on ONE GPU, measured by the build on its own MI355X box and
    committed with its provenance (profiles/r03_strong_scaling_n1.json, else r02): the denominator of `speedup_vs_n1` on N > 1"""
    for name in ("r04_strong_scaling_n1.json", "r03_strong_scaling_n1.json", "r02_strong_scaling_n1.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            if list(d["global_lattice"]) == list(G):
                return d
        except Exception:
            continue
    return None


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args.gpus))     # nothing below runs in the parent

    # stdout carries the one JSON line and nothing else: libraries that write to file descriptor 1 from C (RCCL prints a version
    # banner when its first communicator is created) are sent to stderr, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    gloo = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # self-spawned ranks meet through the parent's file store; under torchrun the launcher's env:// rendezvous is used
        rdzv = {}
        if os.environ.get("DDAMG_BENCH_RDZV"):
            rdzv = dict(init_method=os.environ["DDAMG_BENCH_RDZV"], rank=rank, world_size=world)
        if args.transport == "rccl":
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, {torch.cuda.device_count()} visible "
                                 "(use --transport host to rehearse on fewer)")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), **rdzv)
        else:
            dist.init_process_group(backend="gloo", **rdzv)
            local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    import ddalphaamg_amd as dd
    from ddalphaamg_amd import api
    from ddalphaamg_amd import dist as ddist
    from conftest import splitmix_uniform

    L = list(args.lattice); V = int(np.prod(L))
    p = api.default_params()
    p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw, p.device = -0.1, 1.0, local_rank
    U = synth_gauge_random(V, 20260101 + rank)
    phi = splitmix_uniform(V * 24, 1234 + rank).reshape(V, 12, 2)
    halo_check = None
    grid = ddist.process_grid_for(world); coords = ddist.coords_of(rank, grid)
    if world == 1:
        if args.self_exchange:
            for mu, v in enumerate(int(x) for x in args.self_exchange.split(",")):
                p.process_grid[mu] = v
        ctx = dd.Context(p)
        if args.self_exchange:
            ctx.comm_init_rccl(api.rccl_unique_id())
        ctx.set_gauge(U, anti_pbc=True)
    else:
        # domain decomposition: one process per GPU on a Cartesian grid, `--lattice` sites per GPU, halo exchange over
        # RCCL.  Every process draws the random links of its own part; the clover term is built on the global field
        # (neighbours' links fetched over the host transport).
        for mu in range(4):
            p.process_grid[mu] = grid[mu]; p.process_coords[mu] = coords[mu]
        ctx = dd.Context(p)
        gloo = dist.new_group(backend="gloo")
        ddist.attach_host(ctx, gloo)
        ctx.set_gauge(U, anti_pbc=True)
    del U
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

    # the card's clocks need a few hundred milliseconds of load to settle (a cold 25-launch run reads 4 % slower than the
    # same launches in the middle of a solve): a fixed, untimed spin-up in front of the W warm-up steps, so that a short
    # driver run (--steps 20 --warmup 5) measures the same steady state as the default 1000 / 200
    # (a fixed COUNT, the same on every rank: the applies of a process grid exchange halos with their neighbours)
    spin_up = int(os.environ.get("DDAMG_BENCH_SPIN_UP", "2000"))
    for _ in range(spin_up):
        ctx.dirac_apply(y, x)
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

    out = None
    if rank == 0:
        bytes_site = BYTES_PER_SITE_F32 * (args.precision // 32)
        launch_s = ev_ms * 1e-3 / args.steps
        achieved = bytes_site * V / launch_s / 1e9
        traffic, traffic_src = pmc_traffic(args.precision)
        out = {
            "metric": "fine_wilson_clover_gflops", "value": FLOP_PER_SITE * V * world * args.steps / dt / 1e9,
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": f"f{args.precision}", "data": "synthetic",
            "config": {"workload": f"fine Wilson-Clover apply (d_plus_clover), {'x'.join(map(str, L))} local lattice per GPU, "
                                   "random SU(3) gauge, csw=1.0, anti-periodic T",
                       "flop_per_site": FLOP_PER_SITE,
                       "parallelism": ("domain decomposition, process grid " + "x".join(map(str, grid)) + " (T,Z,Y,X), " + args.transport.upper() + " halo exchange "
                                       "overlapped with the interior tiles") if world > 1 else ("single" if not args.self_exchange else "single GPU, self-exchange " + args.self_exchange + " through RCCL"),
                       "ranks": world, "halo_check_vs_host_transport": halo_check, "untimed_spin_up_applies": spin_up},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "dirac_apply_lds_kernel<float>" if args.precision == 32 else "dirac_apply_lds_kernel<double>", "us_per_launch": launch_s * 1e6,
                         "algorithmic_bytes_per_site": bytes_site},
        }
        if not args.no_cpu_baseline and world == 1:   # "on rank 0 at N=1 only"
            D, cl = ctx.get_operator()
            out["cpu_baseline"] = cpu_baseline(L, D, cl, phi)
            del D, cl
        if world == 1 and not args.self_exchange and args.small_lattice:
            try:
                out["small_lattice"] = small_lattice_leg(local_rank)
            except Exception as e:
                out["small_lattice"] = {"error": str(e)[:200]}
    ctx.close()
    del phi

    # ---- the solve legs: the headline line must survive them.  On several GPUs a watchdog prints the headline with the
    # failure recorded and ends the process with a NON-ZERO code if a leg does not come back (a rank that failed alone would
    # leave the others waiting in a collective); an exception on one rank does the same.
    failed = False
    die_lock = __import__("threading").Lock()

    def emit_and_die(msg):
        with die_lock:      # first caller wins; the process ends inside
            print(f"bench.py: rank {rank}: {msg}", file=sys.stderr, flush=True)
            if rank == 0:
                out.setdefault("strong_scaling", {})["error"] = msg
                emit(json.dumps(out))
            os._exit(3)

    if world > 1:
        # a launcher that lost another rank sends SIGTERM (torchrun, or spawn_ranks above).  The main thread may sit in a
        # collective inside the library and never return to the interpreter, so the signal is picked up through the wake-up
        # descriptor by a thread of its own: rank 0 still prints the headline, every rank ends non-zero.
        import signal, socket, threading
        rs, ws = socket.socketpair(); ws.setblocking(False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(ws.fileno())

        def on_signal():
            while True:     # the descriptor sees every signal the interpreter handles (SIGCHLD of a helper process, ...)
                b = rs.recv(1)
                if b and b[0] == signal.SIGTERM:
                    break
            emit_and_die("terminated by the launcher: another rank failed or the job overran")
        threading.Thread(target=on_signal, daemon=True).start()

    if world == 1 and not args.no_solve and not args.self_exchange and all(v % 8 == 0 for v in L):
        try:
            res = run_solve(amg_params(api, L, 2, local_rank), L, [1, 1, 1, 1], [0, 0, 0, 0], 1, 0, args.transport, None)
            res["workload"] = (f"{'x'.join(map(str, L))}, near-unit gauge exp({GAUGE_EPS} i H) seed {GAUGE_SEED} (tools/synth_gauge.c), m0 -0.3, 2-level AMG "
                               "(4^4 blocks/aggregates, Nvec 24, SAP 2x4, coarse tol 5e-2), FGMRES(50) to 1e-10, rhs=ones (BASELINE configs[2])")
            if not args.no_cpu_baseline:
                cb = cpu_baseline_solve(out["cpu_baseline"]["cores"] if out.get("cpu_baseline") else (os.cpu_count() or 1))
                if cb:
                    if not isinstance(out.get("cpu_baseline"), dict):
                        out["cpu_baseline"] = {}
                    out["cpu_baseline"]["solve"] = cb
                    if "reference_32" in cb and L == [32, 32, 32, 32]:
                        res["iterations_reference"] = cb["reference_32"]["iterations"]
                        res["speedup_vs_reference_32"] = cb["reference_32"]["seconds"] / res["seconds_per_solve"]
            res["coarse_operator"] = coarse_operator_report()
            out["solve"] = res
        except Exception as e:
            out["solve"] = {"error": str(e)[:300]}

    if world == 1 and not args.no_solve and not args.no_strong and not args.self_exchange and L == [32, 32, 32, 32]:
        # BASELINE configs[3]: 48^4, three levels, fp32 smoother / fp64 outer solver, one GPU (timed next to configs[2] and [4])
        try:
            L48 = [48, 48, 48, 48]
            res = run_solve(amg_params(api, L48, 3, local_rank), L48, [1, 1, 1, 1], [0, 0, 0, 0], 1, 0, args.transport, None)
            res["workload"] = (f"48x48x48x48, near-unit gauge exp({GAUGE_EPS} i H) seed {GAUGE_SEED}, m0 -0.3, csw 1, 3-level AMG (4^4 then 2^4 aggregates, Nvec 24/28, "
                               "SAP 2x4 on both smoothing levels, K-cycle 5/2/0.1, coarsest odd-even GMRES to 5e-2), fp64 FGMRES(50) to 1e-10 with the fp32 V-cycle, "
                               "rhs=ones (BASELINE configs[3]); no same-volume reference run exists (about 87 GB there): 12 iterations at 32^4 and 64 x 32^3")
            out["three_level_48"] = res
        except Exception as e:
            out["three_level_48"] = {"error": str(e)[:300]}

    G = list(args.strong_lattice)
    if not args.no_strong and not args.self_exchange:
        if any(G[mu] % (grid[mu] * 8) for mu in range(4)):
            raise SystemExit("--strong-lattice: every extent must be a multiple of 8 x the process grid " + str(grid))
        import threading
        dog = None
        if world > 1:
            dog = threading.Timer(args.leg_timeout, emit_and_die, args=(f"strong-scaling leg did not finish within {args.leg_timeout:.0f} s",))
            dog.daemon = True; dog.start()
        try:
            Lloc = [G[mu] // grid[mu] for mu in range(4)]
            q = amg_params(api, Lloc, 3, local_rank)
            # the reference's default restart length (src/init.c:861-866: 10 x 100): at 64^4 on ONE GPU an fp64 flexible
            # Krylov space of 50 would need 2 x 51 vectors of 3.2 GB; the same algorithm is run at every N
            q.restart, q.max_restart = 10, 100
            # on a process grid the coarsest level (8^4 of the 64^4 lattice) is gathered on every process: one all-gather per
            # coarsest solve instead of two halo exchanges and two global sums per GMRES step on 512 sites per GPU -- what the
            # reference does with a coarse "local lattice" larger than global / process grid (src/init.c:56-72)
            q.gather_coarsest = 1 if world > 1 else 0
            if os.environ.get("DDAMG_BENCH_FAIL_RANK") == str(rank):
                raise RuntimeError("injected failure (DDAMG_BENCH_FAIL_RANK)")
            if os.environ.get("DDAMG_BENCH_HANG_RANK") == str(rank):
                time.sleep(1e6)
            res = run_solve(q, G, grid, coords, world, rank, args.transport, gloo)
            if world > 1:
                t = torch.tensor([res["seconds_per_solve"]], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=gloo)
                res["seconds_per_solve"] = float(t.item())
            res["workload"] = (f"ONE global {'x'.join(map(str, G))} lattice over the process grid {'x'.join(map(str, grid))} (T,Z,Y,X), local "
                               f"{'x'.join(map(str, Lloc))}; near-unit gauge exp({GAUGE_EPS} i H) seed {GAUGE_SEED}, m0 -0.3, csw 1, 3-level AMG (4^4 then 2^4 "
                               "aggregates, Nvec 24/28, SAP 2x4 on both smoothing levels, K-cycle 5/2/0.1, coarsest odd-even GMRES to 5e-2, the coarsest "
                               "level gathered on every process for N > 1), "
                               "fp64 FGMRES(10) to 1e-10 with the fp32 V-cycle, rhs=ones (BASELINE configs[4])")
            res["scaling"] = "strong"; res["n_gpus"] = world; res["transport"] = args.transport if world > 1 else None
            n1 = committed_n1_strong(G)
            if n1:
                res["n1_seconds_per_solve"] = n1["seconds_per_solve"]; res["n1_source"] = n1.get("source")
                res["speedup_vs_n1"] = n1["seconds_per_solve"] / res["seconds_per_solve"]
            if rank == 0:
                out["strong_scaling"] = res
                # the headline `value` is the fine operator at 32^4 sites per GPU (weak scaling: the one metric that exists at every N
                # and carries the roofline); the north star's other metric, seconds per 64^4 solve at N GPUs, is repeated at the top
                # level of the line so that a reader of the N > 1 records finds it without descending into `strong_scaling`
                out["north_star_strong_scaling"] = {"metric": "seconds per FGMRES+AMG solve of ONE global 64^4 lattice", "n_gpus": world,
                                                    "seconds_per_solve": res["seconds_per_solve"], "iterations": res["iterations"],
                                                    "speedup_vs_n1": res.get("speedup_vs_n1"), "n1_seconds_per_solve": res.get("n1_seconds_per_solve"),
                                                    "scaling": "strong", "target": ">= 3.5 x from 1 to 8 GPUs (BASELINE.json north_star)",
                                                    "note": "`value` / `scaling: weak` above is the fine Wilson-Clover operator at a fixed local volume; THIS object "
                                                            "is the strong-scaling solve"}
        except Exception as e:
            failed = True
            if world > 1:
                emit_and_die(f"rank {rank}: {str(e)[:300]}")
            out["strong_scaling"] = {"error": str(e)[:300]}
        if dog:
            dog.cancel()
    if world == 1 and args.rehearse > 1 and not args.no_strong and not args.self_exchange:
        # ---- rehearsal of the N-GPU point of the strong-scaling curve on ONE GPU (no multi-GPU node is available to the build):
        # the per-GPU problem of the decomposition -- local lattice G / grid_N, the same three-level method, the coarsest level
        # gathered -- with the process as its own neighbour in every split direction, so that every boundary coupling runs
        # pack -> ncclSend/ncclRecv -> halo kernels and every reduction ncclAllReduce, as on N GPUs.  What it cannot show: xGMI
        # instead of on-device copies, and the global coarsest level (gathered on N GPUs it is N times the rehearsal's; the
        # correction below uses the measured seconds per coarsest GMRES iteration at both sizes).
        try:
            N = args.rehearse
            grid_n = ddist.process_grid_for(N)
            Lloc = [G[mu] // grid_n[mu] for mu in range(4)]
            q = amg_params(api, Lloc, 3, local_rank)
            q.restart, q.max_restart = 10, 100
            q.gather_coarsest = 1
            sx = [-1 if grid_n[mu] > 1 else 1 for mu in range(4)]
            r8 = run_solve(q, Lloc, sx, [0, 0, 0, 0], 1, 0, "rccl", None)
            q1 = amg_params(api, Lloc, 3, local_rank)
            q1.restart, q1.max_restart = 10, 100
            r1 = run_solve(q1, Lloc, [1, 1, 1, 1], [0, 0, 0, 0], 1, 0, "rccl", None)     # the same lattice, plain periodic wrap
            reh = {"n_gpus_rehearsed": N, "process_grid": grid_n, "local_lattice": Lloc, "self_exchange": sx, "transport": "rccl (self-exchange on one device)",
                   "seconds_per_solve_per_gpu": r8["seconds_per_solve"], "setup_seconds": r8["setup_seconds"], "iterations": r8["iterations"],
                   "coarse_iterations": r8["coarse_iterations"], "true_relres": r8["true_relres"],
                   "same_lattice_without_the_machinery": {"seconds_per_solve": r1["seconds_per_solve"], "iterations": r1["iterations"],
                                                          "setup_seconds": r1["setup_seconds"]},
                   "cost_of_the_machinery": r8["seconds_per_solve"] / r1["seconds_per_solve"], "messages": r8.get("messages")}
            ss = out.get("strong_scaling", {})
            corr = 0.0
            c_n1, c_r = ss.get("coarsest", {}), r8.get("coarsest", {})
            if "seconds_per_iteration" in c_n1 and "seconds_per_iteration" in c_r:
                corr = r8["coarse_iterations"] * max(0.0, c_n1["seconds_per_iteration"] - c_r["seconds_per_iteration"])
                reh["coarsest_level"] = {"rehearsed_sites": c_r["sites"], "gathered_sites_on_n_gpus": c_n1["sites"],
                                         "seconds_per_iteration_rehearsed": c_r["seconds_per_iteration"],
                                         "seconds_per_iteration_gathered": c_n1["seconds_per_iteration"], "correction_seconds": corr}
            reh["predicted_seconds_per_solve_per_gpu"] = r8["seconds_per_solve"] + corr
            if "seconds_per_solve" in ss:
                reh["n1_seconds_per_solve"] = ss["seconds_per_solve"]; reh["n1_iterations"] = ss["iterations"]
                reh["predicted_speedup_vs_n1"] = ss["seconds_per_solve"] / reh["predicted_seconds_per_solve_per_gpu"]
                # per outer iteration, in case the smaller torus of the rehearsal needs another count than the 64^4 lattice
                reh["predicted_speedup_vs_n1_per_iteration"] = (ss["seconds_per_solve"] / ss["iterations"]) / (reh["predicted_seconds_per_solve_per_gpu"] / r8["iterations"])
            reh["not_covered"] = "xGMI latency and bandwidth (messages are on-device copies here); load imbalance between ranks"
            out["rehearsal"] = reh
        except Exception as e:
            out["rehearsal"] = {"error": str(e)[:300]}
    if rank == 0:
        emit(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
