"""GPU tests of the remaining solver variants behind the same seams, each pinned to a run of the reference's
own main program on its sample configurations (oracle/run_reference_cases.py -> tests/golden/ref_runs.json):
method 5 (FGMRES preconditioned by BiCGstab on the odd-even Schur complement, no multigrid), odd_even = 0 (MinRes on whole
Schwarz blocks, GMRES on the whole coarsest operator), and the pipelined Arnoldi recurrence of the coarsest level."""
import json, os
import numpy as np
import pytest
from conftest import GOLDEN, relerr
from ddalphaamg_amd import api
import ddalphaamg_amd as dd

pytestmark = pytest.mark.gpu
RUNS = json.load(open(os.path.join(GOLDEN, "ref_runs.json")))


def context(gold, ext, method, mp, odd_even=1, levels=2, block=2, nvec=20, setup=0):
    p = api.default_params(); p.num_levels = levels
    for mu in range(4):
        p.local_lattice[0][mu] = ext; p.block_lattice[0][mu] = block; p.local_lattice[1][mu] = ext // 2 if ext == 8 else 2
    p.num_vect[0] = nvec; p.post_smooth_iter[0] = 2; p.block_iter[0] = 4; p.setup_iter[0] = setup
    p.restart, p.max_restart, p.tol = 50, 20, 1e-10
    p.coarse_iter, p.coarse_restart, p.coarse_tol = 100, 5, 5e-2
    p.mixed_precision, p.method, p.odd_even = mp, method, odd_even
    p.m0, p.csw = -0.5, 1.0
    ctx = dd.Context(p)
    ctx.set_gauge(gold["gauge"], anti_pbc=True)
    return ctx


@pytest.mark.parametrize("case,ext,mp", [("4x4_m5_mp1", 4, 1), ("4x4_m5_mp0", 4, 0), ("8x8_m5_mp1", 8, 1)])
def test_method_5_fgmres_with_bicgstab(gold4, gold8, case, ext, mp):
    ref = RUNS[case]
    ctx = context(gold4 if ext == 4 else gold8, ext, 5, mp)
    ctx.setup()                                   # nothing to set up: the reference switches the interpolation off
    V = ext ** 4
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, inner, rr = ctx.solve(b, 1e-10)
    assert it == ref["iterations"] and rr < 1e-10
    # inner BiCGstab iterations: the same adaptive tolerance, fp32 rounding moves single calls by an iteration or two
    assert abs(inner - sum(ref["bicgstab_iterations"])) <= 2 + 0.1 * sum(ref["bicgstab_iterations"]), (inner, ref["bicgstab_iterations"])
    h = ctx.residual_history()
    assert len(h) == len(ref["residual_history"])
    # the residual after an outer step is set by where the inner iteration happened to stop: same order of magnitude
    assert np.all(np.abs(np.log10(h / np.array(ref["residual_history"]))) < 1.0)
    xv = ctx.vector(0, 64).upload(x); Dx = ctx.vector(0, 64)
    ctx.dirac_apply(Dx, xv)
    assert abs(np.linalg.norm(b - Dx.download()) / np.linalg.norm(b) - rr) < 1e-12
    ctx.close()


@pytest.mark.parametrize("case,ext,method,setup", [("4x4_oe0", 4, 2, 4), ("8x8_oe0", 8, 2, 3), ("4x4_oe0_m4", 4, 4, 4)])
def test_without_odd_even_preconditioning(gold4, gold8, case, ext, method, setup):
    """odd_even = 0: MinRes on the whole Schwarz block (local_minres on block_d_plus_clover), GMRES on the whole coarsest
    operator, GMRES smoother on the operator itself.  Our own setup on the reference's rand() stream, then rhs = ones: the
    reference's iteration count and residual curve."""
    ref = RUNS[case]
    ctx = context(gold4 if ext == 4 else gold8, ext, method, 1, odd_even=0, setup=setup)
    ctx.setup(setup)
    V = ext ** 4
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    x, it, cit, rr = ctx.solve(b, 1e-10)
    assert rr < 1e-10
    assert abs(it - ref["iterations"]) <= 1, (it, ref["iterations"])
    assert abs(cit / it - ref["coarse_average"]) <= 0.15 * ref["coarse_average"] + 1, (cit / it, ref["coarse_average"])
    h = ctx.residual_history(); rh = np.array(ref["residual_history"])
    n = min(len(h), len(rh))
    assert np.all(np.abs(np.log10(h[:n] / rh[:n])) < 0.3)
    xv = ctx.vector(0, 64).upload(x); Dx = ctx.vector(0, 64)
    ctx.dirac_apply(Dx, xv)
    assert abs(np.linalg.norm(b - Dx.download()) / np.linalg.norm(b) - rr) < 1e-12
    ctx.close()


def amg_ctx(gold, ext, setup):
    ctx = context(gold, ext, 2, 1, setup=setup)
    ctx.setup(setup)
    return ctx


def test_pipelined_arnoldi_on_the_coarsest_level(gold4, monkeypatch):
    """the reference's PIPELINED_ARNOLDI build option (src/linsolve_generic.c:668-733) as a run-time switch: on the coarsest
    level the Gram-Schmidt coefficients of step k come from one reduction that travels while the operator is applied for
    step k+1 (a one-step-delayed recurrence on unnormalised vectors).  Same Krylov space in exact arithmetic."""
    V = 256
    b = np.zeros((V, 12, 2)); b[..., 0] = 1.0
    ctx = amg_ctx(gold4, 4, 4)
    x0, it0, cit0, rr0 = ctx.solve(b, 1e-10)
    ctx.close()
    monkeypatch.setenv("DDAMG_PIPELINED_ARNOLDI", "1")
    ctx = amg_ctx(gold4, 4, 4)
    x1, it1, cit1, rr1 = ctx.solve(b, 1e-10)
    ctx.close()
    assert rr1 < 1e-10 and abs(it1 - it0) <= 1 and abs(cit1 - cit0) <= 0.15 * cit0, (it0, cit0, it1, cit1)
    assert relerr(x1, x0) < 1e-8
    # next to the reference's own pipelined build, whose outer FGMRES uses the single-reduction norm as well: the same
    # residuals while that norm is still accurate (it stalls near sqrt(eps) in the reference), the same coarse-grid work
    ref = RUNS["4x4_pipelined"]
    monkeypatch.setenv("DDAMG_SINGLE_ALLREDUCE_ARNOLDI", "1")
    ctx = amg_ctx(gold4, 4, 4)
    x2, it2, cit2, rr2 = ctx.solve(b, 1e-10)
    h = ctx.residual_history()
    ctx.close()
    assert rr2 < 1e-10
    assert np.all(np.abs(h[:6] / np.array(ref["residual_history"][:6]) - 1.0) < 0.05), (h[:8], ref["residual_history"][:8])
    assert abs(cit2 / it2 - ref["coarse_average"]) < 0.15 * ref["coarse_average"], (cit2 / it2, ref["coarse_average"])
