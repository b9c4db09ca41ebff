"""GPU parity tests (-m gpu) of the two-level V-cycle hot path against golden vectors dumped from the
real reference (oracle/ref_dump_stages.h) on conf/4x4x4x4b6.0000id3n1 (tests/golden/ref_4x4.npz) and on a
lattice whose extents, Schwarz blocks and aggregates all differ by direction (8x4x4x8 / 4x2x2x2 / 4x2x2x4,
seeded random links, tests/golden/ref_ragged.npz):
SAP smoother, restriction / interpolation, Galerkin coarse operator, coarse operator apply, coarsest
odd-even solve, V-cycle and the full FGMRES+AMG solve."""
import numpy as np
import pytest
from conftest import relerr, splitmix_uniform
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

pytestmark = pytest.mark.gpu

# fp32 kernels: a handful of ulp per operation, accumulated over a Schwarz sweep / Krylov cycle
TOL_KERNEL = 5e-6
TOL_SWEEP = 5e-5


@pytest.fixture(scope="module", params=["ref_4x4.npz", "ref_ragged.npz"], ids=["4x4", "ragged"])
def gold4(request):
    """every test below runs on both golden sets (the name is historical)"""
    from conftest import load_golden
    return load_golden(request.param)


def lattice(g):
    return [int(x) for x in g["meta_int"][:4]]


def volume(g):
    return int(np.prod(lattice(g)))


def setup_iterations(g):
    return 4 if volume(g) == 256 else 2     # "d0 setup iter" of the two golden runs (oracle/make_golden.py)


def make_ctx(g, mixed_precision=1, method=2):
    L = lattice(g)
    p = api.default_params()
    p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]
        p.block_lattice[0][mu] = int(g["meta_int"][4 + mu])
        p.local_lattice[1][mu] = int(g["meta_int"][11 + mu]) or L[mu] // 2     # older fixtures do not carry the coarse lattice
    p.num_vect[0] = int(g["meta_int"][9])
    p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = setup_iterations(g)
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = mixed_precision, method, 1
    p.m0, p.csw = float(g["meta_f64"][0]), float(g["meta_f64"][1])
    ctx = dd.Context(p)
    ctx.set_operator(g["D"], g["clover"])
    return ctx


@pytest.fixture(scope="module")
def ref_ctx(gold4):
    """context carrying the REFERENCE hierarchy: its interpolation vectors and its coarse operator"""
    ctx = make_ctx(gold4)
    ctx.set_test_vectors(gold4["interp_vectors"], orthonormalised=True)
    yield ctx
    ctx.close()


def test_site_order_matches_reference_schwarz_layout(gold4, ref_ctx):
    # reference translation_table: lex -> Schwarz index ; ours: site -> lex
    lex_of_site = ref_ctx.site_order(0)
    assert np.array_equal(gold4["schwarz_order"][lex_of_site], np.arange(volume(gold4)))


@pytest.mark.parametrize("cycles", [1, 2, 3])
def test_smoother_from_zero(gold4, ref_ctx, cycles):
    eta = ref_ctx.vector(0, 32).upload(gold4["smoother_eta"]); phi = ref_ctx.vector(0, 32)
    ref_ctx.smoother(phi, eta, cycles, initial_guess_zero=True)
    assert relerr(phi.download(), gold4[f"smoother_nores_out_c{cycles}"]) < TOL_SWEEP
    eta.free(); phi.free()


def test_smoother_with_initial_guess(gold4, ref_ctx):
    eta = ref_ctx.vector(0, 32).upload(gold4["smoother_eta"])
    phi = ref_ctx.vector(0, 32).upload(gold4["smoother_phi0"])
    ref_ctx.smoother(phi, eta, 2, initial_guess_zero=False)
    assert relerr(phi.download(), gold4["smoother_res_out_c2"]) < TOL_SWEEP
    eta.free(); phi.free()


def test_smoother_reduces_residual(gold4, ref_ctx):
    """property (reference -DSCHWARZ_RES check): ||eta - D phi|| drops with every Schwarz cycle"""
    eta_h = gold4["smoother_eta"]
    eta = ref_ctx.vector(0, 32).upload(eta_h); phi = ref_ctx.vector(0, 32); Dphi = ref_ctx.vector(0, 32)
    last = np.linalg.norm(eta_h)
    for cycles in (1, 2, 4):
        ref_ctx.smoother(phi, eta, cycles, True)
        ref_ctx.dirac_apply(Dphi, phi)
        r = np.linalg.norm(eta.download() - Dphi.download())
        assert r < 0.95 * last
        last = r


def test_restrict_interpolate(gold4, ref_ctx):
    f = ref_ctx.vector(0, 32).upload(gold4["restrict_in"]); c = ref_ctx.vector(1, 32)
    ref_ctx.restrict(c, f)
    assert relerr(c.download(), gold4["restrict_out"]) < TOL_KERNEL
    c.upload(gold4["interpolate_in"])
    ref_ctx.interpolate(f, c, add=False)
    assert relerr(f.download(), gold4["interpolate_out"]) < TOL_KERNEL
    # interpolate (+=): phi += P phi_c
    f.upload(gold4["restrict_in"])
    ref_ctx.interpolate(f, c, add=True)
    assert relerr(f.download(), gold4["interpolate_out"].astype(np.float64) + gold4["restrict_in"]) < TOL_KERNEL
    # reference self-check "( P* P - 1 ) phi_c" (src/coarse_operator_generic.c:459-467)
    ref_ctx.interpolate(f, c, add=False)
    c2 = ref_ctx.vector(1, 32)
    ref_ctx.restrict(c2, f)
    assert relerr(c2.download(), gold4["interpolate_in"]) < 2e-6


def test_interpolation_vectors_roundtrip(gold4, ref_ctx):
    assert np.array_equal(ref_ctx.get_interpolation(), gold4["interp_vectors"].astype(np.float64))


def test_gram_schmidt_on_aggregates(gold4):
    """our aggregate-wise Gram-Schmidt of the reference's raw test vectors == its interpolation vectors"""
    ctx = make_ctx(gold4)
    ctx.set_test_vectors(gold4["test_vectors"], orthonormalised=False)
    assert relerr(ctx.get_interpolation(), gold4["interp_vectors"]) < 2e-5
    ctx.close()


def test_galerkin_coarse_operator(gold4, ref_ctx):
    """D_c = P^H D P built on the GPU from the reference's P vs the reference's own coarse operator"""
    D, cl = ref_ctx.get_coarse_operator()
    assert relerr(D, gold4["coarse_D"]) < 1e-5
    assert relerr(cl, gold4["coarse_clover"]) < 1e-5


@pytest.fixture(scope="module")
def refop_ctx(gold4):
    """reference P AND the reference's own coarse operator values"""
    ctx = make_ctx(gold4)
    ctx.set_test_vectors(gold4["interp_vectors"], orthonormalised=True)
    ctx.set_coarse_operator(gold4["coarse_D"], gold4["coarse_clover"])
    yield ctx
    ctx.close()


def test_coarse_operator_roundtrip(gold4, refop_ctx):
    D, cl = refop_ctx.get_coarse_operator()
    assert np.array_equal(D, gold4["coarse_D"].astype(np.float64))
    assert np.array_equal(cl, gold4["coarse_clover"].astype(np.float64))


def test_coarse_apply(gold4, refop_ctx):
    x = refop_ctx.vector(1, 32).upload(gold4["coarse_apply_in"]); y = refop_ctx.vector(1, 32)
    refop_ctx.coarse_apply(y, x)
    assert relerr(y.download(), gold4["coarse_apply_out"]) < TOL_KERNEL


def test_galerkin_identity(gold4, ref_ctx):
    """reference self-check "( P* D P - D_c ) phi_c" (src/coarse_operator_generic.c:482-500)"""
    c = ref_ctx.vector(1, 32).upload(gold4["coarse_apply_in"]); y = ref_ctx.vector(1, 32); y2 = ref_ctx.vector(1, 32)
    f = ref_ctx.vector(0, 32); Df = ref_ctx.vector(0, 32)
    ref_ctx.coarse_apply(y, c)
    ref_ctx.interpolate(f, c, add=False); ref_ctx.dirac_apply(Df, f); ref_ctx.restrict(y2, Df)
    assert relerr(y.download(), y2.download()) < 5e-6


def test_coarse_solve(gold4, refop_ctx):
    b = refop_ctx.vector(1, 32).upload(gold4["coarse_solve_in"]); x = refop_ctx.vector(1, 32)
    it = refop_ctx.coarse_solve(x, b)
    assert abs(it - int(gold4["coarse_solve_iters"][0])) <= 1
    assert relerr(x.download(), gold4["coarse_solve_out"]) < 2e-4
    # the solution satisfies the coarse system to the coarse tolerance
    y = refop_ctx.vector(1, 32)
    refop_ctx.coarse_apply(y, x)
    assert relerr(y.download(), gold4["coarse_solve_in"]) < 5e-2


def test_vcycle(gold4, refop_ctx):
    eta = refop_ctx.vector(0, 32).upload(gold4["vcycle_eta"]); phi = refop_ctx.vector(0, 32)
    refop_ctx.vcycle(phi, eta)
    assert relerr(phi.download(), gold4["vcycle_out"]) < 2e-4


def test_solve_on_reference_hierarchy(gold4, refop_ctx):
    x, it, cit, rr = refop_ctx.solve(gold4["solve_rhs"], 1e-10)
    assert it == int(gold4["solve_iters"][0])
    assert abs(cit - int(gold4["solve_iters"][1])) <= 3
    assert rr < 1e-10
    assert relerr(x, gold4["solve_x"]) < 1e-8


def test_full_setup_and_solve_iteration_parity(gold4):
    """our own setup (same libc rand() stream as the reference) + solve with rhs = ones:
    the reference needs 11 iterations (residual history fixture)"""
    ctx = make_ctx(gold4)
    ctx.setup(setup_iterations(gold4))
    b = np.zeros((volume(gold4), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_hist = gold4["ref_log_ones_history"]
    assert it == int(gold4["ones_solve_iters"][0]) == len(ref_hist)
    assert abs(cit - int(gold4["ones_solve_iters"][1])) <= 8
    assert rr < 1e-10
    hist = ctx.residual_history()
    # same convergence rate as the reference (history agrees within a factor 2 per step)
    assert len(hist) == len(ref_hist) and np.all(np.abs(hist / ref_hist - 1.0) < 5e-3)
    # true solution: D x = b
    from oracle import orc
    Dx = orc.dirac_apply(lattice(gold4), gold4["D"], gold4["clover"], x, 64)
    assert relerr(Dx, b) < 1e-9
    ctx.close()


def test_fp64_vcycle_mode(gold4):
    """mixed_precision 0: the whole V-cycle in fp64"""
    ctx = make_ctx(gold4, mixed_precision=0)
    ctx.set_test_vectors(gold4["interp_vectors"], orthonormalised=True)
    eta = ctx.vector(0, 64).upload(gold4["vcycle_eta"]); phi = ctx.vector(0, 64)
    ctx.vcycle(phi, eta)
    assert relerr(phi.download(), gold4["vcycle_out"]) < 2e-4
    x, it, cit, rr = ctx.solve(gold4["solve_rhs"], 1e-10)
    assert abs(it - int(gold4["solve_iters"][0])) <= 1 and rr < 1e-10
    ctx.close()


def test_pure_gmres_method0(gold4):
    L = lattice(gold4)
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 2
    p.method, p.mixed_precision, p.restart, p.max_restart = 0, 1, 50, 20
    ctx = dd.Context(p)
    ctx.set_operator(gold4["D"], gold4["clover"])
    x, it, cit, rr = ctx.solve(gold4["solve_rhs"], 1e-10)
    assert rr < 1.2e-10 and it > 10
    from oracle import orc
    assert relerr(orc.dirac_apply(L, gold4["D"], gold4["clover"], x, 64), gold4["solve_rhs"]) < 2e-10
    ctx.close()


@pytest.mark.parametrize("mp,fixture", [(1, "ref_8x8_3lvl.npz"), (2, "ref_8x8_3lvl_mp2.npz"), (0, "ref_8x8_3lvl.npz")])
def test_three_level_kcycle_solve(gold8, mp, fixture):
    """BASELINE config: the reference's sample.ini on conf/8x8x8x8b6.0000id3n1 -- 3 levels (8^4 -> 4^4 -> 2^4),
    Nvec 28/28, 2^4 blocks, K-cycle(5,2,0.1), setup 4 (+3 on level 1), rhs = ones, with mixed precision 1 and 2.
    Reference: 11 FGMRES iterations, true residual 1.3e-11 (tests/golden/ref_8x8_3lvl.npz, ref_8x8_3lvl_mp2.npz)."""
    from conftest import load_golden
    g3 = load_golden(fixture)
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
    p.mixed_precision, p.method, p.odd_even = mp, 2, 1
    p.m0, p.csw = float(g3["meta_f64"][0]), float(g3["meta_f64"][1])
    ctx = dd.Context(p)
    plaq = ctx.set_gauge(gold8["gauge"], anti_pbc=True)
    assert abs(plaq - float(g3["meta_f64"][2])) < 1e-9
    ctx.setup(4)
    b = np.zeros((8 ** 4, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_it = int(g3["ones_solve_iters"][0]); ref_hist = g3["ref_log_ones_history"]
    # measured: identical to the reference -- 11 iterations, 192 coarse iterations, same history to 3 digits
    if mp == 0:
        # fp64 V-cycle (double instantiations of every multigrid kernel): not the reference's arithmetic, same convergence
        assert abs(it - ref_it) <= 1 and rr < 1e-10
    else:
        assert it == ref_it and rr < 1e-10
        hist = ctx.residual_history()
        assert len(hist) == len(ref_hist) and np.all(np.abs(hist / ref_hist - 1.0) < 5e-3)
        assert abs(cit - int(g3["ones_solve_iters"][1])) <= 10
        assert abs(rr / float(g3["ones_solve_norm_res"][0]) - 1.0) < 0.05
    from oracle import orc
    D, cl, _ = orc.gauge_to_operator([8, 8, 8, 8], gold8["gauge"], 1, p.m0, p.csw)
    assert relerr(orc.dirac_apply([8, 8, 8, 8], D, cl, x, 64), b) < 1e-9
    ctx.close()


def test_mixed_precision_2_amg(gold4):
    """fgmres_MP (fp32 Krylov basis + V-cycle returning D*phi from the smoother residual, fp64 outer updates):
    reference runs: 4^4 11 iterations, 73 coarse iterations, 3.34e-11 (tests/golden/ref_4x4_mp2.npz); ragged lattice 6 / 12 / 5.2e-12
    (ref_ragged_mp2.npz)"""
    from conftest import load_golden
    gm = load_golden("ref_4x4_mp2.npz" if volume(gold4) == 256 else "ref_ragged_mp2.npz")
    ctx = make_ctx(gold4, mixed_precision=2)
    ctx.setup(setup_iterations(gold4))
    b = np.zeros((volume(gold4), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    assert it == int(gm["ones_solve_iters"][0]) and abs(cit - int(gm["ones_solve_iters"][1])) <= 8
    assert rr < 1e-10 and abs(rr / float(gm["ones_solve_norm_res"][0]) - 1.0) < 0.2
    hist = ctx.residual_history(); ref = gm["ref_log_ones_history"]
    assert len(hist) == len(ref) and np.all(np.abs(hist / ref - 1.0) < 0.05)   # fp32 Krylov basis: the first iterations agree to 1e-7, the last ones to a few percent
    from oracle import orc
    assert relerr(orc.dirac_apply(lattice(gold4), gold4["D"], gold4["clover"], x, 64), b) < 1e-9
    ctx.close()


def test_pure_gmres_mixed_precision_2(gold8):
    """method 0, mixed precision 2 on the 8^4 sample configuration: reference 356 iterations of GMRES(50)
    (tests/golden/ref_8x8_gmres_mp2.npz; printed every 10th iteration)"""
    from conftest import load_golden
    gm = load_golden("ref_8x8_gmres_mp2.npz")
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 2
    p.method, p.mixed_precision, p.restart, p.max_restart, p.tol = 0, 2, 50, 20, 1e-10
    p.m0, p.csw = -0.5, 1.0
    ctx = dd.Context(p)
    ctx.set_gauge(gold8["gauge"], anti_pbc=True)
    b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    assert abs(it - int(gm["ones_solve_iters"][0])) <= 10 and rr < 1.5e-10
    hist = ctx.residual_history(); ref = gm["ref_log_ones_history"]
    k = np.arange(10, 10 * (len(ref) + 1), 10) - 1
    k = k[k < len(hist)]
    assert np.all(np.abs(np.log10(hist[k] / ref[:len(k)])) < 0.3)
    ctx.close()


# ---- the block shape of the production configurations: 4^4 Schwarz blocks (256 sites, the resident-operator kernel) ------
@pytest.fixture(scope="module")
def gold_b4():
    from conftest import load_golden
    return load_golden("ref_8x8_b4.npz")


def make_ctx_b4(gold_b4, gold8):
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 2
    p.num_vect[0] = int(gold_b4["meta_int"][9]); p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 3
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = float(gold_b4["meta_f64"][0]), float(gold_b4["meta_f64"][1])
    ctx = dd.Context(p)
    ctx.set_gauge(gold8["gauge"], anti_pbc=True)
    return ctx


@pytest.mark.parametrize("variant", ["2", "1"])
def test_smoother_with_256_site_blocks_vs_reference(gold_b4, gold8, variant, monkeypatch):
    """red_black_schwarz on 4^4 blocks, both block-solver kernels (resident operator / site pairs), against the
    reference's dumps: from zero with 1-3 cycles (list-4/5 rule: with 2 blocks per direction every block touches both
    lattice boundaries) and with an initial guess"""
    import subprocess, sys, os, textwrap
    # the kernel variant is read once per process: run the comparison in a child with the variable set
    code = textwrap.dedent(f"""
        import sys, os, numpy as np
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
        import test_gpu_multigrid as t
        from conftest import load_golden, relerr
        g = load_golden("ref_8x8_b4.npz"); g8 = load_golden("ref_8x8_dirac.npz")
        ctx = t.make_ctx_b4(g, g8)
        ctx.setup(0)
        eta = ctx.vector(0, 32).upload(g["smoother_eta"]); phi = ctx.vector(0, 32)
        for c in (1, 2, 3):
            ctx.smoother(phi, eta, c, initial_guess_zero=True)
            assert relerr(phi.download(), g[f"smoother_nores_out_c{{c}}"]) < t.TOL_SWEEP, c
        phi.upload(g["smoother_phi0"])
        ctx.smoother(phi, eta, 2, initial_guess_zero=False)
        assert relerr(phi.download(), g["smoother_res_out_c2"]) < t.TOL_SWEEP
        print("B4_SMOOTHER_OK")
    """)
    env = dict(os.environ, DDAMG_SAP_VARIANT=variant)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "B4_SMOOTHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_setup_and_solve_with_256_site_blocks_vs_reference(gold_b4, gold8):
    """the reference's run with 4^4 blocks / aggregates on its 8^4 configuration (setup 3, Nvec 20, rhs = ones): same
    iteration count and convergence curve after our own setup on the same rand() stream"""
    ctx = make_ctx_b4(gold_b4, gold8)
    ctx.setup(3)
    b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_hist = gold_b4["ref_log_ones_history"]
    assert it == int(gold_b4["ones_solve_iters"][0]) == len(ref_hist)
    assert abs(cit - int(gold_b4["ones_solve_iters"][1])) <= 8 and rr < 1e-10
    hist = ctx.residual_history()
    assert np.all(np.abs(hist / ref_hist - 1.0) < 5e-3)
    ctx.close()


def test_slab_wise_galerkin_construction_is_the_same_operator(gold_b4, gold8, monkeypatch):
    """volumes whose Galerkin workspace does not hold all columns for the whole lattice (64^4) walk the lattice in slabs of
    whole aggregates with ALL columns (docs/design/05_host.md); forced here with slabs of 5 of the 16 aggregates (an uneven last slab):
    the coarse operator, and with it every number of the solve, must not change"""
    res = []
    for slabs in (None, "5"):
        if slabs:
            monkeypatch.setenv("DDAMG_GALERKIN_SLAB_AGGS", slabs)
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        Dc, clc = ctx.get_coarse_operator()
        b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((Dc, clc, x, it, cit, rr))
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][3:] == res[1][3:] and np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("slabs", [None, "5"], ids=["whole-lattice", "slabs"])
def test_face_compacted_galerkin_fields_give_the_same_operator(gold_b4, gold8, monkeypatch, slabs):
    """the four forward parts of D P are zero away from the aggregate faces; the Galerkin construction keeps and restricts them
    on the face sites only, and builds them on 256-site tiles through LDS (docs/design/05_host.md).  Against the five full fields per column
    (DDAMG_GALERKIN_FULL_FIELDS) and against the gather form of the field kernel (DDAMG_AGGREGATE_DIRAC_GATHER): the same
    coarse operator up to the rounding of another summation order, the same solve; against the restriction's results passing
    through coarse column vectors (DDAMG_GALERKIN_STORE_COLUMNS): identical"""
    if slabs:
        monkeypatch.setenv("DDAMG_GALERKIN_SLAB_AGGS", slabs)
    res = []
    for knob in (None, "DDAMG_GALERKIN_FULL_FIELDS", "DDAMG_AGGREGATE_DIRAC_GATHER", "DDAMG_GALERKIN_STORE_COLUMNS"):
        if knob:
            monkeypatch.setenv(knob, "1")
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        Dc, clc = ctx.get_coarse_operator()
        b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((Dc, clc, x, it, rr))
        ctx.close()
        if knob:
            monkeypatch.delenv(knob)
    for other in res[1:3]:
        assert relerr(res[0][0], other[0]) < 2e-5 and relerr(res[0][1], other[1]) < 2e-5
        assert res[0][3] == other[3] and other[4] < 1e-10 and relerr(res[0][2], other[2]) < 1e-8
    # another code path did run: the full fields are summed in another order (the gather form of the field kernel differs from
    # the tiled one only through the two-row links it does not use -- on full link storage the two agree bit for bit)
    assert not np.array_equal(res[0][0], res[1][0])
    # the restriction writes straight into the coarse matrices; through coarse column vectors and store launches: the same bits
    assert np.array_equal(res[0][0], res[3][0]) and np.array_equal(res[0][1], res[3][1]) and np.array_equal(res[0][2], res[3][2])


def test_bootstrap_with_one_restriction_and_one_interpolation_for_all_test_vectors(gold_b4, gold8, monkeypatch):
    """the setup's bootstrap V-cycles of the fine level share their two passes over the interpolation operator (restriction
    on the matrix cores, batched interpolation; docs/design/05_host.md).  Against the vector-by-vector form (DDAMG_BOOTSTRAP_UNBATCHED):
    the same test vectors up to the rounding of a different summation order, the same solve"""
    res = []
    for unbatched in (None, "1"):
        if unbatched:
            monkeypatch.setenv("DDAMG_BOOTSTRAP_UNBATCHED", unbatched)
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        tv = ctx.get_test_vectors()
        b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((tv, x, it, rr))
        ctx.close()
    assert relerr(res[0][0], res[1][0]) < 2e-4 and not np.array_equal(res[0][0], res[1][0])   # two code paths did run
    assert abs(res[0][2] - res[1][2]) <= 1 and res[0][3] < 1e-10 and res[1][3] < 1e-10
    assert relerr(res[0][1], res[1][1]) < 1e-8


def test_gram_schmidt_on_256_site_aggregates_one_wavefront_per_chirality(gold_b4, gold8, monkeypatch):
    """gram_schmidt_on_aggregates (src/linalg_generic.c:400-480) on 4^4 aggregates: one wavefront per aggregate and chirality,
    three columns per pass, sums by lane exchanges (default) against the workgroup form of rounds 1-3 (DDAMG_GS_WORKGROUP): the
    same orthonormal columns up to the rounding of another summation order, P^H P = 1 either way, the same solve"""
    res = []
    for knob in (None, "1"):
        if knob:
            monkeypatch.setenv("DDAMG_GS_WORKGROUP", knob)
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        P = ctx.get_interpolation()
        b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((P, x, it, rr))
        ctx.close()
    assert relerr(res[0][0], res[1][0]) < 2e-4 and not np.array_equal(res[0][0], res[1][0])   # two code paths did run
    assert abs(res[0][2] - res[1][2]) <= 1 and res[0][3] < 1e-10 and res[1][3] < 1e-10
    assert relerr(res[0][1], res[1][1]) < 1e-8
    # orthonormal on every aggregate and chirality: 2 aggregates per direction, 256 sites each, chirality = first / second 6 dof
    P = np.asarray(res[0][0]); Pc = (P[..., 0] + 1j * P[..., 1]).reshape(P.shape[0], 8, 8, 8, 8, 12)
    blk = Pc[:, :4, :4, :4, :4, :6].reshape(P.shape[0], -1)
    assert np.abs(blk.conj() @ blk.T - np.eye(P.shape[0])).max() < 5e-6


def test_test_vector_gram_schmidt_by_panels(gold_b4, gold8, monkeypatch):
    """gram_schmidt_PRECISION on the test vectors (src/linalg_generic.c:483-528) by panels of four: the projections of a panel on
    all earlier vectors share their two passes over those (blas.hip vec_panel_project).  Against the column-by-column form
    (DDAMG_TV_GS_COLUMNWISE): the same orthonormal vectors up to rounding -- the projections inside a panel are taken from the
    vector after the first pass -- and the same solve"""
    res = []
    for columnwise in (None, "1"):
        if columnwise:
            monkeypatch.setenv("DDAMG_TV_GS_COLUMNWISE", columnwise)
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        tv = ctx.get_test_vectors()
        b = np.zeros((4096, 12, 2)); b[..., 0] = 1.0
        x, it, cit, rr = ctx.solve(b, 1e-10)
        res.append((tv, x, it, rr))
        ctx.close()
    assert relerr(res[0][0], res[1][0]) < 2e-4 and not np.array_equal(res[0][0], res[1][0])   # two code paths did run
    assert abs(res[0][2] - res[1][2]) <= 1 and res[0][3] < 1e-10 and res[1][3] < 1e-10
    assert relerr(res[0][1], res[1][1]) < 1e-8


def test_bootstrap_in_groups_when_the_workspace_holds_fewer_vectors(gold_b4, gold8, monkeypatch):
    """a setup repeated in a context that already holds its solver workspace finds less free device memory (64^4: room for 17
    of the 24 iterates): interpolation + smoothing at the end of the batched bootstrap then go through the workspace in groups
    (DDAMG_BOOTSTRAP_GROUP forces groups at any volume).  The columns are independent: the same bits"""
    res = []
    for group in (None, "5"):
        if group:
            monkeypatch.setenv("DDAMG_BOOTSTRAP_GROUP", group)
        ctx = make_ctx_b4(gold_b4, gold8)
        ctx.setup(3)
        res.append(ctx.get_test_vectors())
        ctx.close()
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("coarse_restrict", ["mfma", "valu"])
@pytest.mark.parametrize("fixture", ["ref_16x16_3lvl.npz", "ref_16x16_3lvl_hard.npz"], ids=["random-links", "smooth-links"])
def test_three_level_production_block_shapes_16x16(fixture, coarse_restrict, monkeypatch):
    """16^4 with the block shapes of the production configurations -- 4^4 Schwarz blocks and aggregates on the fine level
    (-> 4^4), 2^4 on the coarse level (-> 2^4), K-cycle, Nvec 24/28, setup 3 (+2) -- against the reference's runs on
    seeded random links (m0 0.3, an easy system) and on smooth links exp(0.35 i H) (m0 -0.3, a hard one):
    resident-operator smoother, matrix-core Galerkin construction on both levels, arithmetic-neighbour stencil.
    Same rand() stream, same iteration count and residual history.  The restriction of the coarse level's Galerkin
    construction on the matrix cores (default) and in its vector-unit form (DDAMG_COARSE_RESTRICT_VALU)."""
    from conftest import load_golden, random_su3
    if coarse_restrict == "valu":
        monkeypatch.setenv("DDAMG_COARSE_RESTRICT_VALU", "1")
    g = load_golden(fixture)
    V = 16 ** 4
    p = api.default_params(); p.num_levels = 3
    for mu in range(4):
        p.local_lattice[0][mu] = 16; p.block_lattice[0][mu] = 4
        p.local_lattice[1][mu] = 4; p.block_lattice[1][mu] = 2
        p.local_lattice[2][mu] = 2
    p.num_vect[0] = 24; p.num_vect[1] = 28
    p.post_smooth_iter[0] = p.post_smooth_iter[1] = 2; p.block_iter[0] = p.block_iter[1] = 4
    p.setup_iter[0] = 3; p.setup_iter[1] = 2
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = float(g["meta_f64"][0]), float(g["meta_f64"][1])
    ctx = dd.Context(p)
    if "hard" in fixture:
        import os, sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from bench import near_unit_gauge
        U = near_unit_gauge(V, 0.35, 1617)
    else:
        U = random_su3(V * 4, 1616).reshape(V, 4, 9, 2)
    plaq = ctx.set_gauge(U, anti_pbc=True)
    assert abs(plaq - float(g["meta_f64"][2])) < 1e-10
    ctx.setup(3)
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_hist = g["ref_log_ones_history"]
    assert it == int(g["ones_solve_iters"][0]) == len(ref_hist) and rr < 1e-10
    assert abs(cit - int(g["ones_solve_iters"][1])) <= max(3, int(g["ones_solve_iters"][1]) // 20)
    assert np.all(np.abs(ctx.residual_history() / ref_hist - 1.0) < 5e-3)
    ctx.close()


def test_single_allreduce_arnoldi(gold4, monkeypatch):
    """the reference's SINGLE_ALLREDUCE_ARNOLDI build option (src/linsolve_generic.c:735-800) as a run-time switch: the
    norm of the new Krylov vector comes out of the Gram-Schmidt reduction, as ||w||^2 - sum |h_i|^2 -- a difference that
    loses digits once w is nearly in the span (measured: single entries of the residual curve move by 10 % to 40 % against
    the reference's default recurrence, depending on the last bits of the setup).  Same solution, iteration count within one."""
    monkeypatch.setenv("DDAMG_SINGLE_ALLREDUCE_ARNOLDI", "1")
    ctx = make_ctx(gold4)
    ctx.setup(setup_iterations(gold4))
    b = np.zeros((volume(gold4), 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    assert abs(it - int(gold4["ones_solve_iters"][0])) <= 1 and rr < 1e-10
    hist = ctx.residual_history(); ref_hist = gold4["ref_log_ones_history"]
    m = min(len(hist), len(ref_hist))
    assert np.all(np.abs(hist[:m] / ref_hist[:m] - 1.0) < 0.5)
    from oracle import orc
    assert relerr(orc.dirac_apply(lattice(gold4), gold4["D"], gold4["clover"], x, 64), b) < 1e-9
    ctx.close()


def test_setup_persistence_through_a_test_vector_file(gold4, tmp_path):
    """the reference keeps a setup by writing the test vectors (vector_io_single_file "test vectors", src/io.c:951-1124) and
    reading them back into a fresh method (src/setup_generic.c:131-160): the file written from one context rebuilds the same
    hierarchy in another -- same iteration count, same residual curve, same solution"""
    L = lattice(gold4)
    ctx = make_ctx(gold4)
    ctx.setup(setup_iterations(gold4))
    b = np.zeros((volume(gold4), 12, 2)); b[..., 0] = 1.0
    x1, it1, cit1, rr1 = ctx.solve(b, 1e-10)
    h1 = ctx.residual_history()
    path = tmp_path / "test_vectors"
    api.write_vectors(path, L, ctx.get_test_vectors(), dict(vector_type="test vectors", m0=float(gold4["meta_f64"][0]), csw=float(gold4["meta_f64"][1])))
    ctx.close()
    ctx = make_ctx(gold4)
    ctx.set_test_vectors(api.read_vectors(path, L, int(gold4["meta_int"][9])))
    x2, it2, cit2, rr2 = ctx.solve(b, 1e-10)
    assert (it2, cit2) == (it1, cit1)
    assert np.allclose(ctx.residual_history(), h1, rtol=1e-6)
    assert relerr(x2, x1) < 1e-9
    ctx.close()


def test_pure_cgn_method_minus_1():
    """method -1: conjugate gradients on the normal equations with D^H = g5 D g5 (cgn_double, src/linsolve_generic.c:503-640)
    on conf/4x4x4x4b6.0000id3n1, rhs = ones, against the reference's run (tests/golden/ref_4x4_cgn.npz): 195 iterations, the
    switch to the true residual after 191, and its solution."""
    from conftest import load_golden
    g = load_golden("ref_4x4.npz"); gc = load_golden("ref_4x4_cgn.npz")
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = 4; p.block_lattice[0][mu] = 2
    p.method, p.mixed_precision, p.restart, p.max_restart, p.tol = -1, 1, 50, 20, 1e-10
    ctx = dd.Context(p)
    ctx.set_operator(g["D"], g["clover"])
    b = np.zeros((256, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    ref_it = int(gc["ref_log_cgn_iterations"][0]); switch_it, switch_res = gc["ref_log_cgn_switch"][0]
    assert abs(it - ref_it) <= 2 and rr < 1e-10            # 200-step recurrences in fp64: a step more or less at the threshold
    hist = ctx.residual_history()
    first_true = int(np.argmax(np.diff(hist) > 0)) + 1 if np.any(np.diff(hist) > 0) else len(hist)
    assert abs(first_true - switch_it) <= 2                  # normal-equation phase ends where the reference's does
    assert relerr(x, gc["cgn_x"]) < 1e-8
    from oracle import orc
    assert relerr(orc.dirac_apply([4, 4, 4, 4], g["D"], g["clover"], x, 64), b) < 2e-10
    ctx.close()


def test_coarse_operator_single_read_form(gold4, monkeypatch):
    """CoarseOp::apply has two forms: small lattices read every link from both of its end points in one launch, large ones
    read it once (coarse_apply_once_kernel + finish).  The second form, forced here onto the golden lattices, must give the
    reference's coarse apply as well."""
    monkeypatch.setenv("DDAMG_COARSE_APPLY_ONCE_MIN_SITES", "0")
    import subprocess, sys, os, textwrap
    # the threshold is read once per process: run the comparison in a child
    name = "ref_4x4.npz" if volume(gold4) == 256 else "ref_ragged.npz"
    code = textwrap.dedent(f'''
        import sys, numpy as np
        sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")!r})
        from conftest import load_golden, relerr
        import test_gpu_multigrid as t
        g = load_golden({name!r})
        ctx = t.make_ctx(g)
        ctx.set_test_vectors(g["interp_vectors"], orthonormalised=True)
        ctx.set_coarse_operator(g["coarse_D"], g["coarse_clover"])
        vi = ctx.vector(1, 32).upload(g["coarse_apply_in"]); vo = ctx.vector(1, 32)
        ctx.coarse_apply(vo, vi)
        err = relerr(vo.download(), g["coarse_apply_out"])
        print("ERR", err)
        assert err < t.TOL_KERNEL, err
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ), timeout=300)
    assert r.returncode == 0 and "ERR" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_coarsest_solves_of_many_right_hand_sides_in_lockstep(gold_b4, gold8):
    """coarse_lockstep.h: the coarsest-level odd-even Schur GMRES for many right-hand sides at once -- independent recurrences
    advanced together, the coarse operator on the matrix cores -- against the one-at-a-time solver column by column: the same
    iteration count (one more or less where the stopping test falls on the rounding of the operator kernel), the same solution
    to the accuracy of the solve (both stop at the relative residual 5e-2; the Krylov spaces are the same up to rounding)"""
    ctx = make_ctx_b4(gold_b4, gold8)
    ctx.setup(2)
    lc = 1
    n = ctx.ndof(lc); Vc = ctx.volume(lc)
    ncols = 7
    bs, xs, xr, itr = [], [], [], []
    for c in range(ncols):
        bh = splitmix_uniform(Vc * n * 2, 100 + c).reshape(Vc, n, 2)
        if c == 3:
            bh[:] = 0.0                       # a zero right-hand side among the columns: x = 0, no iterations
        b = ctx.vector(lc, 32).upload(bh); x = ctx.vector(lc, 32); r = ctx.vector(lc, 32)
        bs.append(b); xs.append(x)
        itr.append(ctx.coarse_solve(r, b) if c != 3 else 0)
        xr.append(r.download() if c != 3 else np.zeros_like(bh))
    its = ctx.coarse_solve_many(xs, bs)
    for c in range(ncols):
        assert its[c] >= 0 and abs(its[c] - itr[c]) <= 1, (c, its, itr)
        got = xs[c].download()
        if c == 3:
            assert np.all(got == 0.0)
        elif its[c] == itr[c]:
            assert relerr(got, xr[c]) < 1e-4, (c, relerr(got, xr[c]))
        # the defining property either way: the residual of the full coarse system is below the tolerance of the solve
        if c != 3:
            Dx = ctx.vector(lc, 32); ctx.coarse_apply(Dx, xs[c])
            bh = bs[c].download()
            assert np.linalg.norm(Dx.download() - bh) / np.linalg.norm(bh) < 5.5e-2
            Dx.free()
    ctx.close()
