"""GPU tests (-m gpu) of the two masses of the library interface.

1. Setup at a mass different from the solver mass (dd_alpha_amg_par::setup_m0, src/dd_alpha_amg.c:106,146; method_update shifts
   the operator to it for the iterative setup and back, src/init.c:326-357): the host program tests/mpi/setup_mass_driver.c,
   which knows only include/dd_alpha_amg.h, was run against the REFERENCE library (oracle/make_setup_mass_golden.py ->
   tests/golden/ref_setup_mass.json); here the same program runs against libddamg_hip.so, through dd_alpha_amg_init and through
   the parameter struct.  (The reference's own struct path aborts in validate_parameters on an uninitialised g.ncycle[], see the
   driver's header; it hard-wires the parameters the file of the init-path run spells out, so that run is the fixture for both.)
2. The mass shift itself (shift_update, src/dirac.c:646-668) as diagonal updates on the device on every level
   (ddamg_hip_shift_mass): the operator, the hierarchy and the solve afterwards against a context that got the shifted
   operator through a full upload + Galerkin rebuild."""
import json, os, subprocess, sys
import numpy as np
import pytest
from conftest import GOLDEN, REPO, relerr, splitmix_uniform
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

sys.path.insert(0, os.path.join(REPO, "oracle"))
pytestmark = pytest.mark.gpu
DRIVER = os.path.join(REPO, "tests", "mpi", "setup_mass_driver")


@pytest.fixture(scope="module")
def golden():
    return {(c["mode"], c["setup_m0"]): c for c in json.load(open(os.path.join(GOLDEN, "ref_setup_mass.json")))["cases"]}


@pytest.mark.parametrize("mode", ["init", "struct"])
@pytest.mark.parametrize("setup_m0", [-0.35, -0.5])
def test_setup_mass_apart_from_solver_mass_against_the_reference_library(golden, tmp_path, mode, setup_m0):
    import make_setup_mass_golden as mk      # checker side: writes the same inputs the reference run got, parses the same output
    if not os.path.exists(DRIVER):
        pytest.fail("tests/mpi/setup_mass_driver not built (make -C ddalphaamg_amd/csrc mpi)")
    ref = golden[("init", setup_m0)]
    gauge, ini = mk.write_inputs(str(tmp_path), ref["m0"], setup_m0, ref["setup_iter"])
    cmd = [DRIVER, mode, repr(ref["m0"]), repr(setup_m0), gauge, str(ref["setup_iter"])] + ([ini] if mode == "init" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = mk.parse(r.stdout)
    assert abs(got["plaquette"] - ref["plaquette"]) < 1e-11
    assert got["iterations"] == ref["iterations"], (got, ref)
    assert got["relres"] < 1e-10
    # same Krylov trajectory as the reference: the residual curve, digit by digit at the start, to a few per cent at the end
    # (fp32 V-cycle; the two setup masses differ by 3 % in the FIRST entry, 0.0821 against 0.0851, so this tells them apart)
    h, hr = np.array(got["residual_history"]), np.array(ref["residual_history"])
    assert len(h) == len(hr)
    assert abs(h[0] / hr[0] - 1.0) < 2e-5 and np.max(np.abs(h / hr - 1.0)) < 3e-2, (h, hr)
    assert abs(got["coarse_iterations"] - ref["coarse_iterations"]) <= 3
    assert abs(got["setup_coarse_iterations"] - ref["setup_coarse_iterations"]) <= 0.03 * ref["setup_coarse_iterations"]


def hierarchy_params(L, levels, m0):
    p = api.default_params(); p.num_levels = levels
    for mu in range(4):
        p.local_lattice[0][mu] = L[mu]; p.block_lattice[0][mu] = 2; p.local_lattice[1][mu] = L[mu] // 2
        if levels == 3:
            p.block_lattice[1][mu] = 2; p.local_lattice[2][mu] = L[mu] // 4
    p.num_vect[0] = 12; p.num_vect[1] = 14; p.setup_iter[0] = 2; p.setup_iter[1] = 2
    p.post_smooth_iter[0] = p.post_smooth_iter[1] = 2; p.block_iter[0] = p.block_iter[1] = 4
    p.restart, p.max_restart, p.tol = 30, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = 1, 2, 1
    p.m0, p.csw = m0, 1.0
    return p


@pytest.mark.parametrize("levels", [2, 3])
def test_mass_shift_on_the_device_equals_upload_and_rebuild(gold8, levels):
    L = [8, 8, 8, 8]; V = 4096
    U = gold8["gauge"]
    m_a, m_b = -0.5, -0.42
    b = splitmix_uniform(V * 24, 3).reshape(V, 12, 2)

    A = dd.Context(hierarchy_params(L, levels, m_a))
    A.set_gauge(U, anti_pbc=True)
    A.setup(2)
    D0, cl0 = A.get_operator()
    coarse0 = [A.get_coarse_operator(level=l) for l in range(1, levels)]
    A.shift_mass(m_b)                      # device: diagonals of every level
    DA, clA = A.get_operator()
    cl_expect = cl0.copy(); cl_expect[:, :12, 0] += m_b - m_a
    assert np.array_equal(DA, D0) and np.array_equal(clA, cl_expect)
    xa, ita, cita, rra = A.solve(b, 1e-10)
    hist_a = np.array(A.residual_history())
    phi = splitmix_uniform(V * 24, 4).reshape(V, 12, 2)
    ya = {}
    for prec in (32, 64):
        v = A.vector(0, prec).upload(phi); w = A.vector(0, prec); A.dirac_apply(w, v); ya[prec] = w.download()
    DcA, clcA = A.get_coarse_operator()

    B = dd.Context(hierarchy_params(L, levels, m_a))
    B.set_gauge(U, anti_pbc=True)
    B.setup(2)                             # same rand() stream as A (seeded at create): same test vectors
    B.set_operator(D0, cl_expect)          # the long way: upload of the shifted field, Galerkin construction on every level
    xb, itb, citb, rrb = B.solve(b, 1e-10)
    hist_b = np.array(B.residual_history())
    for prec in (32, 64):
        v = B.vector(0, prec).upload(phi); w = B.vector(0, prec); B.dirac_apply(w, v)
        assert np.array_equal(w.download(), ya[prec])          # fine operator: bit for bit, both precisions
    DcB, clcB = B.get_coarse_operator()
    assert np.array_equal(DcA, DcB)                            # links untouched
    assert relerr(clcA, clcB) < 2e-6                           # P^H (D + d) P = D_c + d up to the rounding of P^H P = 1 in fp32
    assert ita == itb and abs(cita - citb) <= 2 and max(rra, rrb) < 1e-10
    assert np.max(np.abs(hist_a / hist_b - 1.0)) < 2e-2
    assert relerr(xa, xb) < 1e-8
    # and back: the original operator again, bit for bit on the fine level
    A.shift_mass(m_a)
    _, cl_back = A.get_operator()
    assert np.max(np.abs(cl_back - cl0)) < 1e-15
    # ... and bit for bit on every coarse level, however often the shift is repeated (an HMC stream shifts to the setup mass and back at
    # every setup update): the shifts are one accumulated fp64 number on top of the diagonal the Galerkin construction left
    for _ in range(3):
        A.shift_mass(-0.37); A.shift_mass(m_a)
    for l in range(1, levels):
        Dl, cll = A.get_coarse_operator(level=l)
        assert np.array_equal(Dl, coarse0[l - 1][0]) and np.array_equal(cll, coarse0[l - 1][1]), l
    A.close(); B.close()


@pytest.mark.parametrize("csw", [0.0, 1.0])
def test_mass_shift_of_the_operator_alone_with_and_without_a_clover_term(gold8, csw):
    """csw == 0: the reference's clover field is the 12 diagonal entries 4 + m0 only (src/dirac.c:41-43, shift_update_PRECISION adds to
    them, src/dirac_generic.c:517-527); here the same kernel serves both cases.  Shifted context against one built at the new
    mass: the same stored operator, the same applied operator in both precisions, the same odd-even smoother (the 6x6 inverses)."""
    L = [8, 8, 8, 8]; V = 4096
    U = gold8["gauge"]
    phi = splitmix_uniform(V * 24, 6).reshape(V, 12, 2)
    outs = []
    for shifted in (True, False):
        p = api.default_params(); p.num_levels = 2
        for mu in range(4):
            p.local_lattice[0][mu] = 8; p.block_lattice[0][mu] = 4; p.local_lattice[1][mu] = 2
        p.num_vect[0] = 8; p.mixed_precision, p.method, p.odd_even = 1, 2, 1
        p.m0, p.csw = (-0.3 if shifted else 0.15), csw
        ctx = dd.Context(p)
        ctx.set_gauge(U, anti_pbc=True)
        if shifted:
            ctx.shift_mass(0.15)
        D, cl = ctx.get_operator()
        res = [D, cl]
        for prec in (32, 64):
            x = ctx.vector(0, prec).upload(phi); y = ctx.vector(0, prec)
            ctx.dirac_apply(y, x); res.append(y.download())
        e = ctx.vector(0, 32).upload(phi); s = ctx.vector(0, 32)
        ctx.smoother(s, e, 2, initial_guess_zero=True)
        res.append(s.download())
        outs.append(res)
        ctx.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.max(np.abs(a[1] - b[1])) < 1e-15
    assert relerr(a[2], b[2]) < 1e-6 and relerr(a[3], b[3]) < 1e-14 and relerr(a[4], b[4]) < 1e-5
    if csw == 0.0:
        assert np.all(a[1][:, 12:, :] == 0.0) and np.allclose(a[1][:, :12, 0], 4.15)


# ---- scale_clover on the device (src/dirac.c:624-644 + operator_updates, as dd_alpha_amg_wilson_solve applies them around a solve) ---------
def test_scaled_solve_against_the_reference_library(tmp_path):
    """The reference LIBRARY through its own interface with scale_even = 1.1, scale_odd = 0.9 around a solve (tests/golden/
    ref_setup_mass.json "scaled_case", oracle/make_setup_mass_golden.py): the same host program on libddamg_hip.so gives the same
    iteration count and residual curve for the scaled solve (first entry 0.0878 against 0.0851 unscaled), the same solution, and the
    unscaled solve after it is the one before it -- the operator is back bit for bit."""
    import make_setup_mass_golden as mk
    ref = json.load(open(os.path.join(GOLDEN, "ref_setup_mass.json")))["scaled_case"]
    gauge, ini = mk.write_inputs(str(tmp_path), ref["m0"], ref["setup_m0"], ref["setup_iter"])
    cmd = [DRIVER, "init", repr(ref["m0"]), repr(ref["setup_m0"]), gauge, str(ref["setup_iter"]), ini, "-", repr(ref["scale_even"]), repr(ref["scale_odd"])]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = mk.parse(r.stdout)
    for key in ("scaled_solve", "after_scaled_solve"):
        g, rf = got[key], ref[key]
        assert g["iterations"] == rf["iterations"] and g["relres"] < 1e-10, (key, g, rf)
        h, hr = np.array(g["residual_history"]), np.array(rf["residual_history"])
        assert len(h) == len(hr) and abs(h[0] / hr[0] - 1.0) < 2e-5 and np.max(np.abs(h / hr - 1.0)) < 3e-2, (key, h, hr)
        assert abs(g["coarse_iterations"] - rf["coarse_iterations"]) <= 3
    assert abs(got["scaled_solution_checksum"] / ref["scaled_solution_checksum"] - 1.0) < 1e-8
    # scaled and unscaled curves differ by 3 % in their first entry: the test tells them apart
    assert abs(got["scaled_solve"]["residual_history"][0] / got["residual_history"][0] - 1.0) > 0.02
    # the operator is back: the unscaled solve after the scaled one repeats the one before it digit for digit
    assert got["after_scaled_solve"]["residual_history"] == got["residual_history"]
    assert got["after_scaled_solve"]["relres"] == got["relres"]


@pytest.mark.parametrize("levels", [2, 3])
def test_scale_clover_on_the_device_equals_upload_and_rebuild(gold8, levels):
    """ddamg_hip_scale_clover (parity-masked kernels on both precisions + the coarse Galerkin rebuild) against a context that got the
    scaled field through a full upload: the same applied operator in both precisions bit for bit, the same coarse operator, the same
    solve; and scaling back by (1, 1) restores the original operator bit for bit."""
    L = [8, 8, 8, 8]; V = 4096
    U = gold8["gauge"]
    se, so = 1.1, 0.9
    b = splitmix_uniform(V * 24, 3).reshape(V, 12, 2)
    phi = splitmix_uniform(V * 24, 4).reshape(V, 12, 2)

    def apply_both(ctx):
        out = {}
        for prec in (32, 64):
            v = ctx.vector(0, prec).upload(phi); w = ctx.vector(0, prec); ctx.dirac_apply(w, v); out[prec] = w.download()
        return out

    A = dd.Context(hierarchy_params(L, levels, -0.5))
    A.set_gauge(U, anti_pbc=True)
    A.setup(2)
    D0, cl0 = A.get_operator()
    y0 = apply_both(A)
    A.scale_clover(se, so)
    ya = apply_both(A)
    xa, ita, cita, rra = A.solve(b, 1e-10)
    hist_a = np.array(A.residual_history())
    DcA, clcA = A.get_coarse_operator()

    # the scaled field the long way: host loop over the reference's storage, upload, Galerkin construction on every level
    c = np.indices(L).reshape(4, -1).sum(axis=0) % 2          # lexicographic sites, global parity (one process: no offset)
    cl_scaled = cl0 * np.where(c == 1, so, se)[:, None, None]
    B = dd.Context(hierarchy_params(L, levels, -0.5))
    B.set_gauge(U, anti_pbc=True)
    B.setup(2)
    B.set_operator(D0, cl_scaled)
    yb = apply_both(B)
    for prec in (32, 64):
        assert np.array_equal(ya[prec], yb[prec]) and not np.array_equal(ya[prec], y0[prec])
    xb, itb, citb, rrb = B.solve(b, 1e-10)
    hist_b = np.array(B.residual_history())
    DcB, clcB = B.get_coarse_operator()
    assert np.array_equal(DcA, DcB) and np.array_equal(clcA, clcB)
    assert ita == itb and cita == citb and max(rra, rrb) < 1e-10 and np.array_equal(hist_a, hist_b)
    assert relerr(xa, xb) < 1e-12
    # the true residual of the scaled system, by the oracle
    from oracle import orc
    Dh, clh, _ = orc.gauge_to_operator(L, U, 1, -0.5, 1.0)
    assert relerr(orc.dirac_apply(L, Dh, cl_scaled, xa, 64), b) < 1e-9
    # and back
    A.scale_clover(1.0, 1.0)
    yr = apply_both(A)
    for prec in (32, 64):
        assert np.array_equal(yr[prec], y0[prec])
    _, cl_back = A.get_operator()
    assert np.array_equal(cl_back, cl0)
    A.close(); B.close()
