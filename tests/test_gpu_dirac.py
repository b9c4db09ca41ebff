"""GPU parity tests (-m gpu): fine Wilson-Clover operator through the C-ABI vs reference golden
vectors and vs the oracle."""
import numpy as np
import pytest
from conftest import relerr, splitmix_uniform, random_su3
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

pytestmark = pytest.mark.gpu

TOL32 = 2e-6   # fp32: reference's own fp32-vs-fp64 operator check prints ~1e-7
TOL64 = 1e-13


def make_ctx(L, B=None, m0=-0.5, csw=1.0):
    p = api.default_params()
    p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]
        p.block_lattice[0][mu] = (B or L)[mu]
    p.m0, p.csw = m0, csw
    return dd.Context(p)


@pytest.mark.parametrize("case,block", [("4", [2, 2, 2, 2]), ("4", [4, 4, 4, 4]), ("8", [4, 4, 4, 4]), ("8", [2, 2, 2, 2])])
def test_dirac_apply_golden(case, block, gold4, gold8):
    g = gold4 if case == "4" else gold8
    L = [int(x) for x in g["meta_int"][:4]]
    ctx = make_ctx(L, block, g["meta_f64"][0], g["meta_f64"][1])
    plaq = ctx.set_gauge(g["gauge"], anti_pbc=True)
    assert abs(plaq - g["meta_f64"][2]) < 1e-12
    D, cl = ctx.get_operator()
    if case == "4":
        assert np.array_equal(D, g["D"])
        assert relerr(cl, g["clover"]) < 1e-14
    else:
        assert np.array_equal(D[::97], g["D_sample"])
        assert relerr(cl[::97], g["clover_sample"]) < 1e-14
    for prec, ref, tol in ((64, "dirac_out_f64", TOL64), (32, "dirac_out_f32_as_f64", TOL32), (32, "dirac_out_f64", TOL32)):
        x = ctx.vector(0, prec).upload(g["dirac_in"])
        y = ctx.vector(0, prec)
        ctx.dirac_apply(y, x)
        assert relerr(y.download(), g[ref]) < tol, (prec, ref)
        x.free(); y.free()
    ctx.close()


def test_vector_roundtrip():
    ctx = make_ctx([4, 4, 8, 4], [2, 2, 4, 2])
    a = splitmix_uniform(4 * 4 * 8 * 4 * 24, 3).reshape(-1, 12, 2)
    v = ctx.vector(0, 64).upload(a)
    assert np.array_equal(v.download(), a)
    v32 = ctx.vector(0, 32).upload(a)
    assert np.array_equal(v32.download(), a.astype(np.float32).astype(np.float64))
    ctx.close()


@pytest.mark.parametrize("L,B", [([4, 6, 8, 4], [2, 2, 4, 2]), ([8, 4, 4, 16], [4, 4, 2, 4]),
                                 # volumes that do not fill whole 256-site tiles, odd numbers of blocks, a single partial tile
                                 ([4, 4, 4, 6], [2, 2, 2, 2]), ([6, 6, 6, 6], [2, 2, 2, 2]), ([2, 4, 6, 4], [2, 2, 2, 2]), ([2, 2, 2, 2], [2, 2, 2, 2])])
def test_dirac_apply_vs_oracle_ragged(L, B):
    """non-cubic lattices / blocks, random gauge, against the oracle"""
    from oracle import orc
    V = int(np.prod(L))
    U = random_su3(V * 4, 11).reshape(V, 4, 9, 2)
    ctx = make_ctx(L, B, m0=0.1, csw=1.3)
    plaq = ctx.set_gauge(U, anti_pbc=True)
    D, cl, plaq_o = orc.gauge_to_operator(L, U, 1, 0.1, 1.3)
    assert abs(plaq - plaq_o) < 1e-12
    Dg, clg = ctx.get_operator()
    assert np.array_equal(Dg, D) and relerr(clg, cl) < 1e-13
    phi = splitmix_uniform(V * 24, 5).reshape(V, 12, 2)
    for prec, tol in ((64, TOL64), (32, TOL32)):
        x = ctx.vector(0, prec).upload(phi); y = ctx.vector(0, prec)
        ctx.dirac_apply(y, x)
        assert relerr(y.download(), orc.dirac_apply(L, D, cl, phi, prec)) < tol
    ctx.close()


def test_csw_zero():
    from oracle import orc
    L = [4, 4, 4, 4]; V = 256
    U = random_su3(V * 4, 2).reshape(V, 4, 9, 2)
    ctx = make_ctx(L, [2, 2, 2, 2], m0=0.2, csw=0.0)
    ctx.set_gauge(U, anti_pbc=False)
    D, cl, _ = orc.gauge_to_operator(L, U, 0, 0.2, 0.0)
    phi = splitmix_uniform(V * 24, 9).reshape(V, 12, 2)
    x = ctx.vector(0, 64).upload(phi); y = ctx.vector(0, 64)
    ctx.dirac_apply(y, x)
    assert relerr(y.download(), orc.dirac_apply(L, D, cl, phi, 64)) < TOL64
    ctx.close()


def test_linearity_and_g5_hermiticity_32cube():
    """size-independent properties at a larger volume (16^3 x 32): linearity and
    gamma5-hermiticity <y, D x> = <g5 D g5 y, x>  (src/dirac_generic.c:281-305)"""
    L = [32, 16, 16, 16]; V = int(np.prod(L))
    U = random_su3(V * 4, 4).reshape(V, 4, 9, 2)
    ctx = make_ctx(L, [4, 4, 4, 4], m0=-0.1, csw=1.0)
    ctx.set_gauge(U, anti_pbc=True)
    x = splitmix_uniform(V * 24, 21).reshape(V, 12, 2)
    y = splitmix_uniform(V * 24, 22).reshape(V, 12, 2)
    g5 = np.array([-1] * 6 + [1] * 6, dtype=np.float64)[None, :, None]
    def D(v, prec=64):
        a = ctx.vector(0, prec).upload(v); b = ctx.vector(0, prec)
        ctx.dirac_apply(b, a); out = b.download(); a.free(); b.free(); return out
    cx = lambda a: a[..., 0] + 1j * a[..., 1]
    Dx, Dy = D(x), D(y)
    assert relerr(D(2.0 * x - 3.0 * y), 2.0 * Dx - 3.0 * Dy) < 1e-13
    lhs = np.vdot(cx(y), cx(Dx)); rhs = np.vdot(cx(g5 * D(g5 * y)), cx(x))
    assert abs(lhs - rhs) / abs(lhs) < 1e-12
    assert relerr(D(x, 32), Dx) < TOL32
    ctx.close()


@pytest.mark.parametrize("prec,tol", [(32, 1e-6), (64, 1e-14)])
def test_blas1(prec, tol):
    """vector_PRECISION_copy / saxpy / inner product / norm (src/linalg_generic.c:29-353)"""
    ctx = make_ctx([4, 4, 8, 8], [2, 2, 2, 2])
    n = 4 * 4 * 8 * 8 * 24
    a = splitmix_uniform(n, 41).reshape(-1, 12, 2); b = splitmix_uniform(n, 42).reshape(-1, 12, 2)
    if prec == 32:
        a = a.astype(np.float32).astype(np.float64); b = b.astype(np.float32).astype(np.float64)
    x = ctx.vector(0, prec).upload(a); y = ctx.vector(0, prec).upload(b); z = ctx.vector(0, prec)
    ctx.vec_copy(z, x)
    assert np.array_equal(z.download(), a)
    al = 0.3 - 1.7j
    ctx.vec_axpy(z, x, y, al)
    cx = lambda v: v[..., 0] + 1j * v[..., 1]
    ref = cx(a) + al * cx(b)
    got = cx(z.download())
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol
    d, nx = ctx.vec_dot(x, y)
    assert abs(d - np.vdot(cx(a), cx(b))) / abs(d) < 1e-12 and abs(nx - np.linalg.norm(cx(a))) / nx < 1e-13
    ctx.close()


def test_error_paths():
    ctx = make_ctx([4, 4, 4, 4], [2, 2, 2, 2])
    x = ctx.vector(0, 32); y = ctx.vector(0, 64)
    with pytest.raises(dd.DDAMGError):
        ctx.dirac_apply(y, x)       # no operator yet
    ctx.set_gauge(random_su3(256 * 4, 1).reshape(256, 4, 9, 2))
    with pytest.raises(dd.DDAMGError):
        ctx.dirac_apply(y, x)       # precision mismatch
    with pytest.raises(dd.DDAMGError):
        ctx.dirac_apply(x, x)       # in-place
    with pytest.raises(dd.DDAMGError):
        x.upload(np.zeros(5))
    ctx.close()


@pytest.mark.parametrize("field,value,message", [("method", 6, "src/init.c:982"), ("method", -2, "method must be"),
                                                 ("mixed_precision", 3, "mixed_precision")])
def test_unsupported_parameters_are_refused_at_creation(field, value, message):
    """the variants of the reference that are not implemented fail loudly instead of running something else"""
    from ddalphaamg_amd import api
    p = api.default_params(); p.num_levels = 2
    for mu in range(4):
        p.local_lattice[0][mu] = 4; p.block_lattice[0][mu] = 2; p.local_lattice[1][mu] = 2
    setattr(p, field, value)
    with pytest.raises(dd.DDAMGError, match=message):
        dd.Context(p)


def test_two_row_link_storage_and_its_fall_back(monkeypatch):
    """operators on 4^4-block lattices keep two rows per link and rebuild the third (one third less link traffic in the
    operator and in the Schwarz block solver) when every link is +-1/2 of an SU(3) matrix -- the anti-periodic time slice
    carries the minus sign -- and fall back to the full storage otherwise: same results either way, also for links that are
    not unitary"""
    from ddalphaamg_amd import api
    from oracle import orc
    L = [8, 4, 8, 4]; V = int(np.prod(L))
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw = -0.1, 1.0
    phi = splitmix_uniform(V * 24, 77).reshape(V, 12, 2)
    U = random_su3(V * 4, 3).reshape(V, 4, 9, 2)
    Ubad = U.copy(); Ubad[V // 3, 2] *= 1.05           # one link that is no multiple of a unitary matrix
    for prec, tol, close in ((32, 2e-6, 5e-7), (64, 1e-13, 1e-14)):
        for links in (U, Ubad):
            outs = []
            for comp in ("1", "0"):
                monkeypatch.setenv("DDAMG_LINK_COMPRESSION", comp)
                ctx = dd.Context(p)
                ctx.set_gauge(links, anti_pbc=True)
                x = ctx.vector(0, prec).upload(phi); y = ctx.vector(0, prec)
                ctx.dirac_apply(y, x)
                outs.append(y.download())
                if comp == "1":
                    D, cl = ctx.get_operator()
                    assert relerr(outs[0], orc.dirac_apply(L, D, cl, phi, prec)) < tol
                ctx.close()
            if links is Ubad:
                assert np.array_equal(outs[0], outs[1])       # compression refused: the same kernel ran twice
            else:
                assert relerr(outs[0], outs[1]) < close and not np.array_equal(outs[0], outs[1])


def test_two_row_links_are_not_taken_for_an_fp64_operator_that_is_unitary_to_1e_13_only(monkeypatch):
    """links whose third row deviates from 2 conj(row0 x row1) by 1e-13: the fp64 operator (outer solver, reported residual) must
    be the caller's matrix, not a reconstruction that differs from it by more than its own 1e-13 parity tolerance -> it keeps
    the full storage (bit-identical to the run with compression switched off); the fp32 operator still compresses (its own
    rounding is 6e-8)"""
    from ddalphaamg_amd import api
    L = [8, 4, 8, 4]; V = int(np.prod(L))
    p = api.default_params(); p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 4
    p.m0, p.csw = -0.1, 1.0
    phi = splitmix_uniform(V * 24, 78).reshape(V, 12, 2)
    U = random_su3(V * 4, 4).reshape(V, 4, 9, 2)
    U[:, :, 6:9, :] *= 1.0 + 1e-13 * splitmix_uniform(V * 4 * 6, 5).reshape(V, 4, 3, 2)     # third rows off by ~1e-13 relative
    outs = {}
    for comp in ("1", "0"):
        monkeypatch.setenv("DDAMG_LINK_COMPRESSION", comp)
        ctx = dd.Context(p)
        ctx.set_gauge(U, anti_pbc=True)
        for prec in (32, 64):
            x = ctx.vector(0, prec).upload(phi); y = ctx.vector(0, prec)
            ctx.dirac_apply(y, x)
            outs[(comp, prec)] = y.download()
        ctx.close()
    assert np.array_equal(outs[("1", 64)], outs[("0", 64)])                    # fp64: full storage either way
    assert not np.array_equal(outs[("1", 32)], outs[("0", 32)])                # fp32: two-row storage taken
    assert relerr(outs[("1", 32)], outs[("0", 32)]) < 5e-7
