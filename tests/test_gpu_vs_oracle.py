"""GPU parity (-m gpu) against the pinned numpy/scipy restatement (oracle/mg_oracle.py) on lattices that are in no
golden set, seeded random links: 4x8x4x4 (T,Z,Y,X) with Schwarz blocks 2x4x2x2 = aggregates; 4x4x4x8 with blocks
2x2x2x4; and 4x4x8x8 with 2^4 blocks inside 2x2x4x4 aggregates (several blocks per aggregate).  The hierarchy comes from
the GPU setup (device generator); the oracle receives the same interpolation vectors and coarse operator."""
import numpy as np
import pytest
from conftest import relerr, random_su3, splitmix_uniform
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

pytestmark = pytest.mark.gpu

SHAPES = {"blocks=aggregates": ([4, 8, 4, 4], [2, 4, 2, 2], [2, 2, 2, 2]),
          "long-x": ([4, 4, 4, 8], [2, 2, 2, 4], [2, 2, 2, 2]),
          "blocks-in-aggregates": ([4, 4, 8, 8], [2, 2, 2, 2], [2, 2, 2, 2])}


@pytest.fixture(scope="module", params=list(SHAPES), ids=list(SHAPES))
def pair(request):
    global L, B, LC, V
    L, B, LC = SHAPES[request.param]
    V = int(np.prod(L))
    from oracle import mg_oracle as mo
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = B[mu]; p.local_lattice[1][mu] = LC[mu]
    p.num_vect[0] = 10; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = 2
    p.restart, p.max_restart, p.tol = 30, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = 0.3, 1.0
    p.test_vector_rng, p.rng_seed = 1, 99
    ctx = dd.Context(p)
    ctx.set_gauge(random_su3(V * 4, 31).reshape(V, 4, 9, 2), anti_pbc=True)
    ctx.setup(2)
    D, cl = ctx.get_operator()
    Dc, clc = ctx.get_coarse_operator()
    tl = mo.TwoLevel(L, LC, B, D, cl, ctx.get_interpolation(), Dc, clc)
    yield ctx, tl, mo
    ctx.close()


def vec(a, mo):
    return mo.cplx(np.asarray(a, dtype=np.float64)).ravel()


def test_galerkin_operator_vs_oracle(pair):
    ctx, tl, mo = pair
    G = (tl.P.conj().T @ tl.A @ tl.P).toarray()
    assert np.abs(G - tl.Mc.toarray()).max() / np.abs(G).max() < 2e-5


@pytest.mark.parametrize("cycles", [1, 3])
def test_smoother_vs_oracle(pair, cycles):
    ctx, tl, mo = pair
    eta = splitmix_uniform(V * 24, 3).reshape(V, 12, 2)
    e = ctx.vector(0, 32).upload(eta); phi = ctx.vector(0, 32)
    ctx.smoother(phi, e, cycles, initial_guess_zero=True)
    assert relerr(phi.download().reshape(-1, 2), mo.reim(tl.sap.smooth(vec(eta, mo), cycles))) < 5e-5
    phi0 = splitmix_uniform(V * 24, 4).reshape(V, 12, 2)
    phi.upload(phi0)
    ctx.smoother(phi, e, cycles, initial_guess_zero=False)
    assert relerr(phi.download().reshape(-1, 2), mo.reim(tl.sap.smooth(vec(eta, mo), cycles, phi0=vec(phi0, mo)))) < 5e-5
    e.free(); phi.free()


def test_transfer_and_coarse_apply_vs_oracle(pair):
    ctx, tl, mo = pair
    f = splitmix_uniform(V * 24, 5).reshape(V, 12, 2)
    c = splitmix_uniform(16 * 20 * 2, 6).reshape(16, 20, 2)
    fv = ctx.vector(0, 32).upload(f); cv = ctx.vector(1, 32); cw = ctx.vector(1, 32)
    ctx.restrict(cv, fv)
    assert relerr(cv.download().reshape(-1, 2), mo.reim(tl.restrict(vec(f, mo)))) < 5e-6
    cv.upload(c)
    ctx.interpolate(fv, cv, add=False)
    assert relerr(fv.download().reshape(-1, 2), mo.reim(tl.interpolate(vec(c, mo)))) < 5e-6
    ctx.coarse_apply(cw, cv)
    assert relerr(cw.download().reshape(-1, 2), mo.reim(tl.Mc @ vec(c, mo))) < 5e-6
    for v in (fv, cv, cw):
        v.free()


def test_vcycle_and_solve_vs_oracle(pair):
    ctx, tl, mo = pair
    eta = splitmix_uniform(V * 24, 8).reshape(V, 12, 2)
    e = ctx.vector(0, 32).upload(eta); phi = ctx.vector(0, 32)
    ctx.vcycle(phi, e)
    assert relerr(phi.download().reshape(-1, 2), mo.reim(tl.vcycle(vec(eta, mo)))) < 2e-4
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    xo, ito, hist = tl.solve(vec(b, mo), 1e-10, restart=30)
    assert it == ito and abs(cit - tl.coarse_its) <= 3
    assert relerr(x.reshape(-1, 2), mo.reim(xo)) < 1e-8
    ratio = ctx.residual_history() / np.array(hist)   # fp32 V-cycle against the fp64 restatement: the curves drift apart slowly
    assert np.all(np.abs(ratio[:3] - 1.0) < 0.05) and np.all(np.abs(ratio - 1.0) < 0.3)
    e.free(); phi.free()
