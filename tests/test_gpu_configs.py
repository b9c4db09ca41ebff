"""GPU tests of BASELINE.json configs[3] and configs[4] on ONE GPU: 48^4 and 64^4, three-level AMG (4^4 then 2^4 aggregates,
K-cycle), fp32 V-cycle inside fp64 FGMRES.  No CPU code can check a solve at this volume in test time, so parity rests on
size-independent properties: the true residual recomputed with the fp64 operator, the iteration band of the reference on this
kind of field (12 +- a few: tests/golden/ref_32x32_2lvl.json, ref_16x16_3lvl_hard.npz), the Galerkin identity P^H D P = D_c
and P^H P = 1 on BOTH coarse levels, and bit-identical repeated solves.  The 64^4 case is the N = 1 point of the
strong-scaling curve (bench.py `strong_scaling`): it runs the production Schwarz kernel (two-level face buffers, no 2 GiB
descriptor limit), which the test checks by timing nothing and asserting only what the library reports."""
import os, sys
import numpy as np
import pytest
from conftest import relerr, splitmix_uniform
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
pytestmark = pytest.mark.gpu


def hierarchy(ext, restart, max_restart):
    import bench
    p = bench.amg_params(api, [ext] * 4, 3, 0)
    p.restart, p.max_restart = restart, max_restart
    return p


@pytest.fixture(scope="module", params=[(48, 50, 20), (64, 10, 100)], ids=["48^4 configs[3]", "64^4 configs[4] N=1"])
def big(request):
    import synth
    ext, restart, max_restart = request.param
    p = hierarchy(ext, restart, max_restart)
    ctx = dd.Context(p)
    U = synth.synth_gauge([ext] * 4, 0.35, 20260101)
    ctx.set_gauge(U, anti_pbc=True)
    del U
    ctx.setup(p.setup_iter[0])
    yield ctx, ext
    ctx.close()


def device_norm_of_difference(ctx, a, b):
    z = ctx.vector(a.level, a.precision)
    ctx.vec_axpy(z, a, b, -1.0)
    _, n = ctx.vec_dot(z, z)
    z.free()
    return n


def test_solve_true_residual_iteration_band_and_repeatability(big):
    ctx, ext = big
    V = ext ** 4
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    bv = ctx.vector(0, 64).upload(b); del b
    x1 = ctx.vector(0, 64); x2 = ctx.vector(0, 64)
    it, cit, rr = ctx.solve_vec(x1, bv, 1e-10)
    assert rr < 1e-10, rr
    # the reference itself on this field and this hierarchy shape: 12 iterations at 32^4 and at 64 x 32^3, the largest volumes
    # it fits into the build container (tests/golden/ref_32x32_3lvl.json, ref_64x32_3lvl.json; 48^4 needs about 87 GB there)
    print(f"{ext}^4 three-level:", it, cit, rr)
    assert abs(it - 12) <= 1, it
    # the true residual, recomputed here with the fp64 operator on the device
    Dx = ctx.vector(0, 64)
    ctx.dirac_apply(Dx, x1)
    _, nb = ctx.vec_dot(bv, bv)
    res = device_norm_of_difference(ctx, bv, Dx) / nb
    assert res < 1e-10 and abs(res - rr) < 1e-3 * rr + 1e-14, (res, rr)
    # a second solve on the same hierarchy: bit-identical (deterministic reductions, no atomics)
    it2, cit2, rr2 = ctx.solve_vec(x2, bv, 1e-10)
    assert (it2, cit2, rr2) == (it, cit, rr)
    assert device_norm_of_difference(ctx, x1, x2) == 0.0
    for v in (bv, x1, x2, Dx):
        v.free()


@pytest.mark.parametrize("level", [0, 1])
def test_galerkin_identity_and_orthonormality_on_both_coarse_levels(big, level):
    """restrict(D_l interpolate(e)) == D_{l+1} e and restrict(interpolate(e)) == e for a random vector e of level l+1"""
    ctx, ext = big
    n = ctx.ndof(level + 1); Vc = ctx.volume(level + 1)
    e = splitmix_uniform(Vc * n * 2, 5 + level).reshape(Vc, n, 2)
    ec = ctx.vector(level + 1, 32).upload(e)
    f = ctx.vector(level, 32); Df = ctx.vector(level, 32); r = ctx.vector(level + 1, 32); Dce = ctx.vector(level + 1, 32)
    ctx.interpolate(f, ec, add=False)
    ctx.restrict(r, f)
    assert relerr(r.download(), e) < 5e-6
    if level == 0:
        ctx.dirac_apply(Df, f)
    else:
        ctx.coarse_apply(Df, f)
    ctx.restrict(r, Df)
    ctx.coarse_apply(Dce, ec)
    assert relerr(r.download(), Dce.download()) < 5e-5
    for v in (ec, f, Df, r, Dce):
        v.free()
