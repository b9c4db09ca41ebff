"""GPU parity tests (-m gpu) of the other smoothers behind the smoother seam of the reference
(smoother_PRECISION, src/vcycle_generic.c:25-84): additive Schwarz (method 1, additive_schwarz_PRECISION
src/schwarz_generic.c:1077-1257), sixteen colours (method 3, sixteen_color_schwarz_PRECISION :1652-1804) and GMRES on the
global odd-even Schur complement (method 4, solve_oddeven_PRECISION src/oddeven_generic.c:740-777), on the fine level
and on an intermediate level, against dumps and runs of the reference with the same `method:` line
(oracle/make_golden.py: 4x4_m*, ragged_m*, 8x8_3lvl_m*)."""
import numpy as np
import pytest
from conftest import load_golden, relerr
from ddalphaamg_amd import api
import ddalphaamg_amd as dd
from test_gpu_multigrid import make_ctx, lattice, volume, setup_iterations, TOL_SWEEP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[("ref_4x4.npz", "4x4", 1), ("ref_4x4.npz", "4x4", 3), ("ref_4x4.npz", "4x4", 4),
                                        ("ref_ragged.npz", "ragged", 1), ("ref_ragged.npz", "ragged", 3), ("ref_ragged.npz", "ragged", 4)],
                ids=["4x4-additive", "4x4-sixteen", "4x4-gmres", "ragged-additive", "ragged-sixteen", "ragged-gmres"])
def case(request):
    base, name, method = request.param
    return load_golden(base), load_golden(f"ref_{name}_m{method}.npz"), method


@pytest.fixture(scope="module")
def ctx(case):
    g, gm, method = case
    c = make_ctx(g, method=method)
    yield c
    c.close()


@pytest.mark.parametrize("cycles", [1, 2, 3])
def test_smoother_from_zero(case, ctx, cycles):
    g, gm, method = case
    eta = ctx.vector(0, 32).upload(g["smoother_eta"]); phi = ctx.vector(0, 32)
    ctx.smoother(phi, eta, cycles, initial_guess_zero=True)
    assert relerr(phi.download(), gm[f"smoother_nores_out_c{cycles}"]) < TOL_SWEEP
    eta.free(); phi.free()


def test_smoother_with_initial_guess(case, ctx):
    g, gm, method = case
    eta = ctx.vector(0, 32).upload(g["smoother_eta"])
    phi = ctx.vector(0, 32).upload(g["smoother_phi0"])
    ctx.smoother(phi, eta, 2, initial_guess_zero=False)
    assert relerr(phi.download(), gm["smoother_res_out_c2"]) < TOL_SWEEP
    eta.free(); phi.free()


def test_schedules_differ(case):
    """the dumps of the three schedules are different functions of the same input (guards against a fixture mix-up)"""
    g, gm, method = case
    assert relerr(gm["smoother_nores_out_c2"], g["smoother_nores_out_c2"]) > 1e-3


def test_setup_and_solve_iteration_parity(case):
    """full setup with this smoother (same libc rand() stream as the reference) + solve of rhs = ones"""
    g, gm, method = case
    c = make_ctx(g, method=method)
    c.setup(setup_iterations(g))
    b = np.zeros((volume(g), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = c.solve(b, 1e-10)
    ref_hist = gm["ref_log_ones_history"]
    assert it == int(gm["ones_solve_iters"][0]) == len(ref_hist)
    assert rr < 1e-10
    hist = c.residual_history()
    # same curve; an entry below 1e-13 (the last one of the easy ragged system with the GMRES smoother: 5e-15) is fp32 noise
    assert len(hist) == len(ref_hist) and np.all(np.abs(hist / ref_hist - 1.0) < np.where(ref_hist > 1e-13, 5e-3, 0.5))
    assert abs(cit - int(gm["ones_solve_iters"][1])) <= max(8, int(gm["ones_solve_iters"][1]) // 20)
    c.close()


@pytest.mark.parametrize("mp", [1, 2, 0])
@pytest.mark.parametrize("method", [1, 3, 4])
def test_three_level_kcycle_solve(method, mp):
    """the reference's sample.ini hierarchy on conf/8x8x8x8b6.0000id3n1 (3 levels, K-cycle) with the additive / sixteen-colour
    schedule on both smoothing levels.  Mixed precision 2 and 0 (fp64 V-cycle: the double instantiations of the coarse-level
    kernels, incl. the fused block solver with its 98 KB of LDS) have no reference run here: they must agree with mixed precision 1
    (the additive smoother then also hands back D*phi, src/schwarz_generic.c:1180-1222)."""
    gold8 = load_golden("ref_8x8_dirac.npz")
    g3 = load_golden(f"ref_8x8_3lvl_m{method}.npz")
    p = api.default_params()
    p.num_levels = 3
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 2
        p.local_lattice[1][mu] = 4; p.block_lattice[1][mu] = 2
        p.local_lattice[2][mu] = 2
    p.num_vect[0] = 28; p.num_vect[1] = 28
    p.post_smooth_iter[0] = p.post_smooth_iter[1] = 2; p.block_iter[0] = p.block_iter[1] = 4
    p.setup_iter[0] = 4; p.setup_iter[1] = 3
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.kcycle, p.kcycle_restart, p.kcycle_max_restart, p.kcycle_tol = 1, 5, 2, 1e-1
    p.mixed_precision, p.method, p.odd_even = mp, method, 1
    p.m0, p.csw = float(g3["meta_f64"][0]), float(g3["meta_f64"][1])
    c = dd.Context(p)
    plaq = c.set_gauge(gold8["gauge"], anti_pbc=True)
    assert abs(plaq - float(g3["meta_f64"][2])) < 1e-9
    c.setup(4)
    b = np.zeros((8 ** 4, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = c.solve(b, 1e-10)
    ref_it = int(g3["ones_solve_iters"][0]); ref_hist = g3["ref_log_ones_history"]
    assert rr < 1e-10
    if mp == 1:
        assert it == ref_it
        hist = c.residual_history()
        assert len(hist) == len(ref_hist) and np.all(np.abs(hist / ref_hist - 1.0) < 5e-3)
        assert abs(cit - int(g3["ones_solve_iters"][1])) <= max(10, int(g3["ones_solve_iters"][1]) // 20)
    else:
        assert abs(it - ref_it) <= 1
    from oracle import orc
    D, cl, _ = orc.gauge_to_operator([8, 8, 8, 8], gold8["gauge"], 1, p.m0, p.csw)
    assert relerr(orc.dirac_apply([8, 8, 8, 8], D, cl, x, 64), b) < 1e-9
    c.close()


def test_fp64_vcycle_mode(case):
    """mixed_precision 0 (the whole V-cycle in fp64, double instantiations of the same kernels) with each smoother: the
    solve on the reference's hierarchy needs the reference's iteration count within one"""
    g, gm, method = case
    c = make_ctx(g, mixed_precision=0, method=method)
    c.setup(setup_iterations(g))
    b = np.zeros((volume(g), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = c.solve(b, 1e-10)
    assert abs(it - int(gm["ones_solve_iters"][0])) <= 1 and rr < 1e-10
    from oracle import orc
    assert relerr(orc.dirac_apply(lattice(g), g["D"], g["clover"], x, 64), b) < 1e-9
    c.close()
