"""CPU tests (-m "not gpu"): pin the numpy/scipy restatement of the multigrid path (oracle/mg_oracle.py) against the
dumps of the real reference (tests/golden/ref_4x4.npz and the non-cubic ref_ragged.npz) -- restriction,
interpolation, coarse operator, Galerkin identity, Schwarz smoother with and without initial guess, coarsest-level
odd-even solve, V-cycle, and the iteration count of the full FGMRES+AMG solve.  The reference's V-cycle runs in fp32,
the restatement in fp64: tolerances are fp32-sized."""
import numpy as np
import pytest
from conftest import relerr, load_golden
from oracle import mg_oracle as mo


@pytest.fixture(scope="module", params=["ref_4x4.npz", "ref_ragged.npz"], ids=["4x4", "ragged"])
def hier(request):
    g = load_golden(request.param)
    L = [int(x) for x in g["meta_int"][:4]]
    B = [int(x) for x in g["meta_int"][4:8]]
    Lc = [int(x) or L[mu] // 2 for mu, x in enumerate(g["meta_int"][11:15])]
    tl = mo.TwoLevel(L, Lc, B, g["D"], g["clover"], g["interp_vectors"], g["coarse_D"], g["coarse_clover"])
    return g, tl


def vec(a):
    return mo.cplx(np.asarray(a, dtype=np.float64)).ravel()


def test_fine_matrix_is_the_oracle_operator(hier):
    g, tl = hier
    assert relerr(mo.reim(tl.A @ vec(g["dirac_in"])), np.asarray(g["dirac_out_f64"]).reshape(-1, 2)) < 1e-13


def test_restrict_interpolate(hier):
    g, tl = hier
    assert relerr(mo.reim(tl.restrict(vec(g["restrict_in"]))), np.asarray(g["restrict_out"]).reshape(-1, 2)) < 2e-6
    assert relerr(mo.reim(tl.interpolate(vec(g["interpolate_in"]))), np.asarray(g["interpolate_out"]).reshape(-1, 2)) < 2e-6
    # P^H P = 1 (aggregate-wise orthonormal vectors)
    PhP = (tl.P.conj().T @ tl.P).toarray()
    assert np.abs(PhP - np.eye(PhP.shape[0])).max() < 5e-6


def test_coarse_operator(hier):
    g, tl = hier
    assert relerr(mo.reim(tl.Mc @ vec(g["coarse_apply_in"])), np.asarray(g["coarse_apply_out"]).reshape(-1, 2)) < 2e-6
    # Galerkin: the reference's coarse operator is P^H D P of its own interpolation (built in fp32)
    G = (tl.P.conj().T @ tl.A @ tl.P).toarray()
    assert np.abs(G - tl.Mc.toarray()).max() / np.abs(G).max() < 2e-5


@pytest.mark.parametrize("cycles", [1, 2, 3])
def test_schwarz_from_zero(hier, cycles):
    g, tl = hier
    out = tl.sap.smooth(vec(g["smoother_eta"]), cycles)
    assert relerr(mo.reim(out), np.asarray(g[f"smoother_nores_out_c{cycles}"]).reshape(-1, 2)) < 5e-5


def test_schwarz_with_initial_guess(hier):
    g, tl = hier
    out = tl.sap.smooth(vec(g["smoother_eta"]), 2, phi0=vec(g["smoother_phi0"]))
    assert relerr(mo.reim(out), np.asarray(g["smoother_res_out_c2"]).reshape(-1, 2)) < 5e-5


def test_coarse_solve(hier):
    g, tl = hier
    x, it = mo.coarse_solve(tl.Mc, tl.Lc, tl.n, vec(g["coarse_solve_in"]), 5e-2, 100, 5)
    assert abs(it - int(g["coarse_solve_iters"][0])) <= 1
    assert relerr(mo.reim(x), np.asarray(g["coarse_solve_out"]).reshape(-1, 2)) < 2e-4


def test_vcycle(hier):
    g, tl = hier
    out = tl.vcycle(vec(g["vcycle_eta"]))
    assert relerr(mo.reim(out), np.asarray(g["vcycle_out"]).reshape(-1, 2)) < 2e-4


def test_solve_iteration_count(hier):
    g, tl = hier
    x, it, hist = tl.solve(vec(g["solve_rhs"]), 1e-10)
    assert it == int(g["solve_iters"][0])
    assert abs(tl.coarse_its - int(g["solve_iters"][1])) <= 3
    assert relerr(mo.reim(x), np.asarray(g["solve_x"]).reshape(-1, 2)) < 1e-8


# ---- the other smoothers: additive (method 1), sixteen colours (method 3), GMRES on the odd-even system (method 4) ---------------------------
@pytest.fixture(scope="module", params=[("4x4", 1), ("4x4", 3), ("4x4", 4), ("ragged", 1), ("ragged", 3), ("ragged", 4)], ids=lambda p: f"{p[0]}-method{p[1]}")
def schedule(request):
    name, method = request.param
    g = load_golden(f"ref_{name}.npz"); gm = load_golden(f"ref_{name}_m{method}.npz")
    L = [int(x) for x in g["meta_int"][:4]]
    B = [int(x) for x in g["meta_int"][4:8]]
    A = mo.fine_matrix(L, g["D"], g["clover"])
    return g, gm, (mo.GmresSmoother(L, A, 4) if method == 4 else mo.Schwarz(L, B, A, 4, method))


@pytest.mark.parametrize("cycles", [1, 2, 3])
def test_schwarz_schedules_from_zero(schedule, cycles):
    g, gm, sap = schedule
    out = sap.smooth(vec(g["smoother_eta"]), cycles)
    assert relerr(mo.reim(out), np.asarray(gm[f"smoother_nores_out_c{cycles}"]).reshape(-1, 2)) < 5e-5


def test_schwarz_schedules_with_initial_guess(schedule):
    g, gm, sap = schedule
    out = sap.smooth(vec(g["smoother_eta"]), 2, phi0=vec(g["smoother_phi0"]))
    assert relerr(mo.reim(out), np.asarray(gm["smoother_res_out_c2"]).reshape(-1, 2)) < 5e-5


# ---- three levels: the intermediate level's hot-path functions and the SECOND coarse operator -------------------------------------------------
# dumps of the reference on 8^4 -> 4^4 -> 2^4 with 16 / 20 dof per coarse site (its own configuration) and on 16 x 8^3 -> 4 x 2^3 -> 2^4
# with the production 48 / 56 dof (oracle/ref_dump_stages.h dump_three_level)
@pytest.fixture(scope="module", params=["ref_8x8_3lvl_small.npz", "ref_16x8_3lvl_prod.npz"], ids=["8x8-16-20dof", "16x8-48-56dof"])
def hier3(request):
    g = load_golden(request.param)
    m3 = [int(x) for x in g["meta3_int"]]
    L1, B1, L2, n1, n2 = m3[0:4], m3[4:8], m3[8:12], 2 * m3[12], 2 * m3[13]
    A1 = mo.coarse_matrix(L1, g["coarse_D"], g["coarse_clover"], n1)
    A2 = mo.coarse_matrix(L2, g["l2_coarse_D"], g["l2_coarse_clover"], n2)
    P1 = mo.coarse_interpolation_matrix(L1, L2, g["l1_interp_vectors"], n1)
    sap = mo.Schwarz(L1, B1, A1, m3[15], 2, ndof=n1, odd_even=False)
    return dict(g=g, L1=L1, L2=L2, n1=n1, n2=n2, A1=A1, A2=A2, P1=P1, sap=sap, post=m3[14])


def test_coarse_operators_of_three_levels(hier3):
    h = hier3; g = h["g"]
    assert relerr(mo.reim(h["A1"] @ vec(g["l1_apply_in"])), np.asarray(g["l1_apply_out"]).reshape(-1, 2)) < 2e-6
    assert relerr(mo.reim(h["A2"] @ vec(g["l2_apply_in"])), np.asarray(g["l2_apply_out"]).reshape(-1, 2)) < 2e-6


def test_transfer_between_coarse_levels(hier3):
    h = hier3; g = h["g"]
    assert relerr(mo.reim(h["P1"].conj().T @ vec(g["l1_restrict_in"])), np.asarray(g["l1_restrict_out"]).reshape(-1, 2)) < 2e-6
    assert relerr(mo.reim(h["P1"] @ vec(g["l1_interpolate_in"])), np.asarray(g["l1_interpolate_out"]).reshape(-1, 2)) < 2e-6
    PhP = (h["P1"].conj().T @ h["P1"]).toarray()
    assert np.abs(PhP - np.eye(PhP.shape[0])).max() < 5e-6


def test_second_coarse_operator_element_by_element(hier3):
    """set_coarse_self_coupling / set_coarse_neighbor_coupling (src/coarse_operator_generic.c:103-205) on a coarse level: every
    entry of the level-2 self couplings and forward links from the level-1 operator and interpolation vectors"""
    h = hier3; g = h["g"]
    parts = mo.coarse_matrix(h["L1"], g["coarse_D"], g["coarse_clover"], h["n1"], parts=True)
    D2, cl2 = mo.galerkin_coarse_operator(h["L1"], h["L2"], parts, h["P1"], h["n2"])
    rd, rc = mo.cplx(g["l2_coarse_D"]), mo.cplx(g["l2_coarse_clover"])
    assert np.abs(D2 - rd).max() / np.abs(rd).max() < 5e-6
    assert np.abs(cl2 - rc).max() / np.abs(rc).max() < 5e-6


@pytest.mark.parametrize("cycles", [1, 2, 3])
def test_schwarz_on_the_intermediate_level_from_zero(hier3, cycles):
    h = hier3; g = h["g"]
    out = h["sap"].smooth(vec(g["l1_smoother_eta"]), cycles)
    assert relerr(mo.reim(out), np.asarray(g[f"l1_smoother_nores_out_c{cycles}"]).reshape(-1, 2)) < 1e-4


def test_schwarz_on_the_intermediate_level_with_initial_guess(hier3):
    h = hier3; g = h["g"]
    out = h["sap"].smooth(vec(g["l1_smoother_eta"]), 2, phi0=vec(g["l1_smoother_phi0"]))
    assert relerr(mo.reim(out), np.asarray(g["l1_smoother_res_out_c2"]).reshape(-1, 2)) < 1e-4


def test_vcycle_of_the_intermediate_level(hier3):
    h = hier3; g = h["g"]
    eta = vec(g["l1_vcycle_eta"])
    xc, _ = mo.coarse_solve(h["A2"], h["L2"], h["n2"], h["P1"].conj().T @ eta, 5e-2, 100, 5)
    out = h["sap"].smooth(eta, h["post"], phi0=h["P1"] @ xc)
    assert relerr(mo.reim(out), np.asarray(g["l1_vcycle_out"]).reshape(-1, 2)) < 2e-4
