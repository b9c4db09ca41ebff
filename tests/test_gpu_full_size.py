"""GPU tests at BASELINE.json's full single-GPU size (32^4, configs[2]): the oracle cannot run a solve at this
volume in test time, so parity is checked through size-independent properties of the domain --
linearity and gamma5-hermiticity of the operator, the Galerkin identity P^H D P = D_c and P^H P = 1 on the
hierarchy built by the batched (matrix-core) setup, monotone smoother convergence, and the true residual and
iteration count of the FGMRES+AMG solve through both the host-vector and the device-vector entry points -- and the
iteration counts of the REFERENCE ITSELF on the same seeded fields (tests/golden/ref_32x32_2lvl*.json, ref_16x16_4lvl.json,
oracle/run_reference_big.py)."""
import os, sys
import numpy as np
import pytest
from conftest import relerr, splitmix_uniform
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
pytestmark = pytest.mark.gpu
CURVE_BAND = 0.10      # measured in round 4: 0.023 (32^4 two-level) and 0.047 (16^4 four-level) -- a factor 1.26 at every step
# the seeded field the reference itself was run on (oracle/run_reference_big.py)
GAUGE_EPS, GAUGE_SEED = 0.35, 20260101

L = [32, 32, 32, 32]
V = 32 ** 4

@pytest.fixture(scope="module")
def ctx32():
    import synth
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = 32; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 8
    p.num_vect[0] = 24; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 4
    p.restart, p.max_restart, p.tol = 20, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = -0.3, 1.0
    p.test_vector_rng, p.rng_seed = 1, 7
    ctx = dd.Context(p)
    ctx.set_gauge(synth.synth_gauge(L, GAUGE_EPS, GAUGE_SEED), anti_pbc=True)
    ctx.setup(4)
    yield ctx
    ctx.close()


def cx(a):
    return a[..., 0] + 1j * a[..., 1]


def test_operator_properties(ctx32):
    x = splitmix_uniform(V * 24, 21).reshape(V, 12, 2)
    y = splitmix_uniform(V * 24, 22).reshape(V, 12, 2)
    g5 = np.array([-1] * 6 + [1] * 6, dtype=np.float64)[None, :, None]

    def D(v, prec=64):
        a = ctx32.vector(0, prec).upload(v); b = ctx32.vector(0, prec)
        ctx32.dirac_apply(b, a); out = b.download(); a.free(); b.free(); return out
    Dx, Dy = D(x), D(y)
    assert relerr(D(2.0 * x - 3.0 * y), 2.0 * Dx - 3.0 * Dy) < 1e-13
    lhs = np.vdot(cx(y), cx(Dx)); rhs = np.vdot(cx(g5 * D(g5 * y)), cx(x))
    assert abs(lhs - rhs) / abs(lhs) < 1e-12
    assert relerr(D(x, 32), Dx) < 2e-6


def test_galerkin_identity_and_orthonormality(ctx32):
    """restrict(D interpolate(e)) == D_c e and restrict(interpolate(e)) == e for a random coarse vector"""
    n = 48; Vc = 8 ** 4
    e = splitmix_uniform(Vc * n * 2, 5).reshape(Vc, n, 2)
    ec = ctx32.vector(1, 32).upload(e)
    f = ctx32.vector(0, 32); Df = ctx32.vector(0, 32); r = ctx32.vector(1, 32); Dce = ctx32.vector(1, 32)
    ctx32.interpolate(f, ec, add=False)
    ctx32.restrict(r, f)
    assert relerr(r.download(), e) < 5e-6
    ctx32.dirac_apply(Df, f)
    ctx32.restrict(r, Df)
    ctx32.coarse_apply(Dce, ec)
    assert relerr(r.download(), Dce.download()) < 2e-5
    for v in (ec, f, Df, r, Dce):
        v.free()


def test_smoother_converges(ctx32):
    eta = splitmix_uniform(V * 24, 9).reshape(V, 12, 2)
    e = ctx32.vector(0, 32).upload(eta); phi = ctx32.vector(0, 32); Dphi = ctx32.vector(0, 32)
    res = []
    for cycles in (1, 2, 4):
        ctx32.smoother(phi, e, cycles, initial_guess_zero=True)
        ctx32.dirac_apply(Dphi, phi)
        res.append(np.linalg.norm(eta - Dphi.download()) / np.linalg.norm(eta))
    assert res[0] < 0.7 and res[1] < res[0] and res[2] < res[1]
    for v in (e, phi, Dphi):
        v.free()


def test_solve_host_and_device_vectors(ctx32):
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx32.solve(b, 1e-10)
    # the reference on this field with these parameters: 12 iterations (tests/golden/ref_32x32_2lvl.json); +-1 for the other
    # random test vectors (device generator here, rand() there)
    import json
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "ref_32x32_2lvl.json")))
    print("32^4 two-level:", it, cit, rr, "reference", ref["iterations"])
    assert rr < 1e-10 and abs(it - ref["iterations"]) <= 1, (it, ref["iterations"])
    hist = np.array(ctx32.residual_history()); href = np.array(ref["residual_history"])
    n = min(len(hist), len(href)) - 1
    dev = float(np.max(np.abs(np.log10(hist[:n] / href[:n]))))
    print("residual curve against the reference: max |log10 ratio|", dev)
    assert dev < CURVE_BAND      # the same convergence rate at every step
    # the returned relative residual is the true one: recompute it with the fp64 operator
    xv = ctx32.vector(0, 64).upload(x); Dx = ctx32.vector(0, 64)
    ctx32.dirac_apply(Dx, xv)
    assert abs(np.linalg.norm(b - Dx.download()) / np.linalg.norm(b) - rr) < 1e-12
    bv = ctx32.vector(0, 64).upload(b); yv = ctx32.vector(0, 64)
    it2, cit2, rr2 = ctx32.solve_vec(yv, bv, 1e-10)
    assert (it2, cit2) == (it, cit) and rr2 == rr
    assert np.array_equal(yv.download(), x)
    for v in (xv, Dx, bv, yv):
        v.free()


def test_four_level_hierarchy():
    """MAX_MG_LEVELS = 4 (src/dd_alpha_amg_parameters.h:23): 16^4 -> 8^4 -> 4^4 -> 2^4 with K-cycles on both
    intermediate levels converges like the shallower hierarchies"""
    import json, synth
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "ref_16x16_4lvl.json")))     # the reference itself on this field
    Vl = 16 ** 4
    p = api.default_params(); p.num_levels = 4
    for mu in range(4):
        for d in range(4):
            p.local_lattice[d][mu] = 16 >> d; p.block_lattice[d][mu] = 2
    for d in range(3):
        p.num_vect[d] = 20 + 4 * d; p.post_smooth_iter[d] = 2; p.block_iter[d] = 4; p.setup_iter[d] = [3, 2, 2][d]
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.m0, p.csw = 1, 2, -0.3, 1.0
    p.test_vector_rng, p.rng_seed = 1, 3
    ctx = dd.Context(p)
    ctx.set_gauge(synth.synth_gauge([16] * 4, GAUGE_EPS, GAUGE_SEED), anti_pbc=True)
    ctx.setup(3)
    b = np.zeros((Vl, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    print("16^4 four-level:", it, cit, rr, "reference", ref["iterations"])
    assert rr < 1e-10 and abs(it - ref["iterations"]) <= 1, (it, ref["iterations"])
    hist = np.array(ctx.residual_history()); href = np.array(ref["residual_history"])
    n = min(len(hist), len(href)) - 1
    dev = float(np.max(np.abs(np.log10(hist[:n] / href[:n]))))
    print("residual curve against the reference: max |log10 ratio|", dev)
    assert dev < CURVE_BAND
    ctx.close()


@pytest.mark.parametrize("method", [1, 3, 4])
def test_other_smoothers_at_full_size(method):
    """additive / sixteen-colour Schwarz and the GMRES smoother at 32^4: every cycle lowers the residual, and the solve with
    the smoother inside the V-cycle reaches the target with the true residual it reports (the device-side generator makes
    the setup cheap enough to run it per smoother)"""
    import synth
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = 32; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 8
    p.num_vect[0] = 24; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 2
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, method, 1
    p.m0, p.csw = -0.3, 1.0
    p.test_vector_rng, p.rng_seed = 1, 7
    ctx = dd.Context(p)
    ctx.set_gauge(synth.synth_gauge(L, GAUGE_EPS, GAUGE_SEED), anti_pbc=True)
    ctx.setup(2)
    eta = splitmix_uniform(V * 24, 9).reshape(V, 12, 2)
    e = ctx.vector(0, 32).upload(eta); phi = ctx.vector(0, 32); Dphi = ctx.vector(0, 32)
    res = []
    for cycles in (1, 2, 4):
        ctx.smoother(phi, e, cycles, initial_guess_zero=True)
        ctx.dirac_apply(Dphi, phi)
        res.append(np.linalg.norm(eta - Dphi.download()) / np.linalg.norm(eta))
    assert res[0] < 0.9 and res[1] < res[0] and res[2] < res[1], res
    for v in (e, phi, Dphi):
        v.free()
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    # the reference itself with this smoother on this field (oracle/run_reference_big.py 32x32_2lvl_m<method>, setup 2)
    import json
    ref = json.load(open(os.path.join(REPO, "tests", "golden", f"ref_32x32_2lvl_m{method}.json")))
    print("32^4 method", method, ":", it, cit, rr, "reference", ref["iterations"])
    assert ref["method"] == method and rr < 1e-10 and abs(it - ref["iterations"]) <= 1, (it, ref["iterations"], rr)
    xv = ctx.vector(0, 64).upload(x); Dx = ctx.vector(0, 64)
    ctx.dirac_apply(Dx, xv)
    assert abs(np.linalg.norm(b - Dx.download()) / np.linalg.norm(b) - rr) < 1e-12
    ctx.close()
