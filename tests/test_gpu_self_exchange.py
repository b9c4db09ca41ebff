"""The multi-GPU code path on ONE GPU (-m gpu): a process-grid entry of -1 makes the process its own neighbour in that
direction, so every coupling across the lattice boundary goes through pack -> transport -> halo kernels instead of the
periodic wrap.  With the RCCL transport this exercises ncclCommInitRank, grouped ncclSend/ncclRecv, ncclAllReduce and
the stream/event ordering for real (the multi-process tests have to use the host transport, because RCCL refuses two
ranks on one device).  Results must equal the ordinary single-GPU ones."""
import os, subprocess, sys
import numpy as np
import pytest
from conftest import relerr
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def params(gold8, grid, levels=1, mp=1):
    p = api.default_params(); p.num_levels = levels
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 2
        p.local_lattice[1][mu] = 4; p.block_lattice[1][mu] = 2
        p.local_lattice[2][mu] = 2
        p.process_grid[mu] = grid[mu]
    p.num_vect[0] = p.num_vect[1] = 16
    p.setup_iter[0] = 2; p.setup_iter[1] = 2
    p.restart, p.max_restart, p.tol = 30, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 50, 10, 5e-2
    p.mixed_precision, p.method = mp, 2
    p.m0, p.csw = float(gold8["meta_f64"][0]), float(gold8["meta_f64"][1])
    p.test_vector_rng, p.rng_seed = 1, 11
    return p


@pytest.mark.parametrize("grid", [[-1, 1, 1, 1], [1, -1, 1, -1], [-1, -1, -1, -1]])
def test_dirac_and_gauge_through_rccl_self_exchange(gold8, grid):
    ctx = dd.Context(params(gold8, grid))
    ctx.comm_init_rccl(api.rccl_unique_id())
    plaq = ctx.set_gauge(gold8["gauge"], anti_pbc=True)          # halo of links (corners included) through RCCL
    assert abs(plaq - float(gold8["meta_f64"][2])) < 1e-12
    D, cl = ctx.get_operator()
    assert np.array_equal(D[::97], gold8["D_sample"]) and relerr(cl[::97], gold8["clover_sample"]) < 1e-14
    for prec, ref, tol in ((64, "dirac_out_f64", 1e-13), (32, "dirac_out_f32_as_f64", 2e-6)):
        x = ctx.vector(0, prec).upload(gold8["dirac_in"]); y = ctx.vector(0, prec)
        for _ in range(3):
            ctx.dirac_apply(y, x)
        assert relerr(y.download(), gold8[ref]) < tol
        x.free(); y.free()
    ctx.close()


@pytest.mark.parametrize("levels,mp,gather", [(2, 1, 0), (3, 1, 0), (3, 0, 0), (3, 2, 0), (2, 1, 1), (3, 1, 1)])
def test_amg_solve_through_rccl_self_exchange(gold8, levels, mp, gather):
    """smoother, Galerkin construction, coarse operator, K-cycle, coarsest solve and the reductions, all through RCCL;
    mixed precision 0 (all fp64), 1 (fp64 outer / fp32 V-cycle) and 2 (fgmres_MP); gather: the coarsest level collected
    with ncclAllGather and solved whole (ddamg_hip_params::gather_coarsest)"""
    b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
    res = []
    for grid in ([1, 1, 1, 1], [-1, -1, -1, -1]):
        pp = params(gold8, grid, levels, mp)
        pp.gather_coarsest = gather
        ctx = dd.Context(pp)
        if grid[0] == -1:
            ctx.comm_init_rccl(api.rccl_unique_id())
        ctx.set_gauge(gold8["gauge"], anti_pbc=True)
        ctx.setup(2)
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((x, it, cit, rr))
        ctx.close()
    (x0, it0, cit0, rr0), (x1, it1, cit1, rr1) = res
    assert rr1 < 1e-10 and abs(it1 - it0) <= 1
    assert relerr(x1, x0) < 1e-7


def test_rccl_self_exchange_inside_a_torch_process():
    """the same with torch imported first: libddamg_hip.so then resolves to the HIP runtime and the RCCL that torch
    bundles (the configuration bench.py runs in)"""
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(HERE)!r}); sys.path.insert(0, {HERE!r})
from conftest import load_golden, relerr
from ddalphaamg_amd import api
import ddalphaamg_amd as dd
g = load_golden("ref_8x8_dirac.npz")
p = api.default_params(); p.num_levels = 1
for mu in range(4):
    p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4; p.process_grid[mu] = -1
p.m0, p.csw = float(g["meta_f64"][0]), float(g["meta_f64"][1])
ctx = dd.Context(p)
ctx.comm_init_rccl(api.rccl_unique_id())
ctx.set_gauge(g["gauge"], anti_pbc=True)
x = ctx.vector(0, 64).upload(g["dirac_in"]); y = ctx.vector(0, 64)
ctx.dirac_apply(y, x)
err = relerr(y.download(), g["dirac_out_f64"])
maps = open("/proc/self/maps").read()
print("TORCH_RCCL_OK" if err < 1e-13 else "MISMATCH", err, "torch rccl" if "torch/lib/librccl" in maps else "system rccl")
ctx.close()
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TORCH_RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_256_site_blocks_through_rccl_self_exchange(gold8):
    """the production block shape (4^4 Schwarz blocks: resident-operator smoother kernel, arithmetic-neighbour stencil
    kernel with process-boundary tiles, matrix-core Galerkin) through the transport, against the reference's run with
    these blocks (tests/golden/ref_8x8_b4.npz: 15 iterations) -- same rand() stream, same hierarchy"""
    from conftest import load_golden
    gb = load_golden("ref_8x8_b4.npz")
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 2
        p.process_grid[mu] = -1
    p.num_vect[0] = 20; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 3
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = float(gb["meta_f64"][0]), float(gb["meta_f64"][1])
    ctx = dd.Context(p)
    ctx.comm_init_rccl(api.rccl_unique_id())
    ctx.set_gauge(gold8["gauge"], anti_pbc=True)
    eta = ctx.vector(0, 32).upload(gb["smoother_eta"]); phi = ctx.vector(0, 32)
    ctx.setup(3)
    # with the process as its own neighbour every block touches the process boundary on both sides, exactly as every
    # block touches the lattice boundary in the undivided 2-blocks-per-direction run: the same block lists
    ctx.smoother(phi, eta, 2, initial_guess_zero=True)
    assert relerr(phi.download(), gb["smoother_nores_out_c2"]) < 5e-5
    b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_hist = gb["ref_log_ones_history"]
    assert it == int(gb["ones_solve_iters"][0]) and rr < 1e-10
    assert np.all(np.abs(ctx.residual_history() / ref_hist - 1.0) < 5e-3)
    ctx.close()


@pytest.mark.parametrize("split", ["0123", "0", "23"])
@pytest.mark.parametrize("kernel", ["tile", "gather"])
def test_galerkin_operator_through_the_self_exchange_equals_the_undivided_one(gold8, split, kernel, monkeypatch):
    """the coarse operator built on a process grid (the process its own neighbour in the directions of `split`) against the one
    of the undivided lattice, element by element: the distributed instantiations of the Galerkin construction's stencil kernels
    (LDS-tiled by default, gather form with DDAMG_AGGREGATE_DIRAC_GATHER) read the columns of the aggregate-major interpolation
    operator and take the neighbours' boundary through the transport.  (Round 4: the tiled one once came out of the compiler with
    wrong Y and X parts -- every direction is checked by itself here.)"""
    from conftest import load_golden
    gb = load_golden("ref_8x8_b4.npz")
    if kernel == "gather":
        monkeypatch.setenv("DDAMG_AGGREGATE_DIRAC_GATHER", "1")

    def build(selfx):
        p = api.default_params(); p.num_levels = 2
        for mu in range(4):
            p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 2
            if selfx and str(mu) in split:
                p.process_grid[mu] = -1
        p.num_vect[0] = 20; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4
        p.mixed_precision, p.method, p.odd_even = 1, 2, 1
        p.m0, p.csw = float(gb["meta_f64"][0]), float(gb["meta_f64"][1])
        ctx = dd.Context(p)
        if selfx:
            ctx.comm_init_rccl(api.rccl_unique_id())
        ctx.set_gauge(gold8["gauge"], anti_pbc=True)
        ctx.setup(0)
        D, cl = ctx.get_coarse_operator()
        ctx.close()
        return np.asarray(D), np.asarray(cl)

    D0, c0 = build(False)
    D1, c1 = build(True)
    for mu in range(4):
        assert np.max(np.abs(D0[:, mu] - D1[:, mu])) < 1e-6, mu
    assert np.max(np.abs(c0 - c1)) < 5e-6

