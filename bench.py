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
timed on the host cores of the same box).
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


def cpu_baseline(L, D, cl, phi, budget_s=12.0):
    """oracle port timed on the host cores (bounded sample of the same workload)"""
    from oracle import orc
    t1, nt = orc.dirac_time_f32(L, D, cl, phi, 1)
    reps = int(max(2, min(200, budget_s / max(t1, 1e-4))))
    t, nt = orc.dirac_time_f32(L, D, cl, phi, reps)
    V = int(np.prod(L))
    return {"value": FLOP_PER_SITE * V / t / 1e9, "unit": "GFLOP/s", "cores": nt, "kind": "port",
            "sample": f"{reps} fp32 applies of the same {'x'.join(map(str, L))} operator, OpenMP over sites, "
                      f"{t*1e3:.2f} ms/apply"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32])
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import ddalphaamg_amd as dd
    from ddalphaamg_amd import api
    from conftest import splitmix_uniform

    L = list(args.lattice); V = int(np.prod(L))
    p = api.default_params()
    p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw, p.device = -0.1, 1.0, local_rank
    ctx = dd.Context(p)
    U = synth_gauge(V, 20260101 + rank)
    ctx.set_gauge(U, anti_pbc=True)
    phi = splitmix_uniform(V * 24, 1234 + rank).reshape(V, 12, 2)
    x = ctx.vector(0, args.precision).upload(phi)
    y = ctx.vector(0, args.precision)

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
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        bytes_site = BYTES_PER_SITE_F32 * (args.precision // 32)
        launch_s = ev_ms * 1e-3 / args.steps
        achieved = bytes_site * V / launch_s / 1e9
        out = {
            "metric": "fine_wilson_clover_gflops", "value": FLOP_PER_SITE * V * world * args.steps / dt / 1e9,
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": f"f{args.precision}", "data": "synthetic",
            "config": {"workload": f"fine Wilson-Clover apply (d_plus_clover), {'x'.join(map(str, L))} local lattice per GPU, "
                                   "random SU(3) gauge, csw=1.0, anti-periodic T",
                       "flop_per_site": FLOP_PER_SITE, "parallelism": "replicas" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "dirac_apply_kernel", "us_per_launch": launch_s * 1e6,
                         "algorithmic_bytes_per_site": bytes_site},
        }
        if not args.no_cpu_baseline:
            D, cl = ctx.get_operator()
            out["cpu_baseline"] = cpu_baseline(L, D, cl, phi)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
