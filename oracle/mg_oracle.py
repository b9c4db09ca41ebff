"""oracle/mg_oracle.py -- TEST INFRASTRUCTURE ONLY.

Plain numpy/scipy restatement of the multigrid part of the hot path, written in
terms of explicit sparse matrices so that every step is the textbook definition of what the reference computes:

  fine_matrix            the Wilson-Clover operator as a matrix, column by column through the pinned C oracle
                         (oracle/ddamg_oracle.c <-> d_plus_clover_PRECISION, src/dirac_generic.c:159-277)
  interpolation_matrix   P from the interpolation vectors: block diagonal over aggregates and chirality
                         (interpolate_PRECISION / restrict_PRECISION, src/interpolation_generic.c:93-207)
  coarse_matrix          the coarse operator from the reference's storage: links [A B; C D] in four (n/2)^2
                         column-major blocks ordered A,C,B,D (src/coarse_operator_generic.h:124-143), backward
                         coupling G5 U^H G5 (:152-171), packed self coupling (src/coarse_operator_generic.c:109-111,288-315)
  fgmres                 restarted flexible GMRES, classical Gram-Schmidt, Givens QR
                         (fgmres_PRECISION / arnoldi_step / qr_update / compute_solution, src/linsolve_generic.c:219-413,810-982)
  coarse_solve           odd-even Schur complement solve on the even sites
                         (coarse_solve_odd_even_PRECISION, src/coarse_oddeven_generic.c:1139-1189)
  red_black_schwarz      multiplicative red-black Schwarz with odd-even preconditioned MinRes block solves
                         (red_black_schwarz_PRECISION src/schwarz_generic.c:1260-1431, block_solve_oddeven_PRECISION
                         src/oddeven_generic.c:1317-1360, local_minres_PRECISION src/linsolve_generic.c:985-1029),
                         including the reference's start-without-residual rule (src/schwarz_generic.c:1344)
  vcycle / solve         post-smoothing V-cycle and the outer FGMRES (src/vcycle_generic.c:91-141, src/top_level.c:64-104)

Everything is computed in double precision; the reference runs the V-cycle in single precision, so agreement with
its dumps is to fp32 accuracy (tolerances in tests/test_oracle_multigrid.py).  Only tests may import this module.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from . import orc


# ---- geometry ---------------------------------------------------------------------------------------
def coords(L):
    """[V][4] coordinates of the lexicographic sites (T,Z,Y,X; X fastest, src/data_layout.h:30-32)"""
    g = np.indices(L).reshape(4, -1).T
    return g


def lex(c, L):
    return ((c[..., 0] * L[1] + c[..., 1]) * L[2] + c[..., 2]) * L[3] + c[..., 3]


def cplx(a):
    a = np.asarray(a)
    return a[..., 0] + 1j * a[..., 1]


def reim(z):
    return np.stack([z.real, z.imag], axis=-1)


# ---- fine operator as a matrix ----------------------------------------------------------------------
def fine_matrix(L, D, clover):
    """12V x 12V sparse matrix of the fine operator, index = 12*site + dof; built column by column from the
    pinned oracle apply.  Columns with disjoint stencils are probed together: sources whose coordinates agree modulo
    m_mu >= 3 (m_mu the smallest such divisor of L_mu) are at least 3 apart and share no neighbour."""
    V = int(np.prod(L)); n = 12 * V
    c = coords(L)
    rows, cols, vals = [], [], []
    orc.set_threads(1 if V <= 4096 else orc.host_threads())   # thousands of tiny applies: threads only cost here
    # distance-2 colouring: colour = sum_mu (x_mu mod m_mu) * stride
    m = [min([d for d in range(3, x + 1) if x % d == 0] or [x]) for x in L]
    colour = np.zeros(V, dtype=np.int64); stride = 1
    for mu in range(4):
        colour += (c[:, mu] % m[mu]) * stride; stride *= m[mu]
    ncol = stride
    # neighbour table to attribute the entries of a probed column group to their source site
    owner = np.full(V, -1, dtype=np.int64)
    for col in range(ncol):
        src = np.nonzero(colour == col)[0]
        if len(src) == 0:
            continue
        owner[:] = -1
        owner[src] = src
        for mu in range(4):
            for sgn in (+1, -1):
                cc = c[src].copy(); cc[:, mu] = (cc[:, mu] + sgn) % L[mu]
                owner[lex(cc, L)] = src
        for d in range(12):
            e = np.zeros((V, 12, 2)); e[src, d, 0] = 1.0
            y = cplx(orc.dirac_apply(L, D, clover, e, 64))
            nz = np.nonzero(np.abs(y) > 0)
            rows.append(12 * nz[0] + nz[1]); cols.append(12 * owner[nz[0]] + d); vals.append(y[nz])
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    return A


# ---- interpolation ----------------------------------------------------------------------------------
def aggregate_of(L, Lc):
    """coarse lexicographic site of every fine site (aggregates = L/Lc boxes, src/coarsening_generic.c:114-165)"""
    agg = [L[mu] // Lc[mu] for mu in range(4)]
    c = coords(L) // np.array(agg)
    return lex(c, Lc)


def interpolation_matrix(L, Lc, interp_vectors):
    """P: (12 V) x (n Vc), n = 2 Nvec; coarse dof h*Nvec + j of aggregate a couples to the spin-(2h,2h+1) components of
    interpolation vector j on the sites of a (src/interpolation_generic.c:111-120)"""
    P_ = cplx(interp_vectors)               # [N][V][12]
    N, V = P_.shape[0], P_.shape[1]
    n = 2 * N
    a = aggregate_of(L, Lc)
    rows, cols, vals = [], [], []
    site = np.arange(V)
    for j in range(N):
        for d in range(12):
            h = d // 6
            rows.append(12 * site + d); cols.append(n * a + h * N + j); vals.append(P_[j, :, d])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(12 * V, n * int(np.prod(Lc))))


def coarse_interpolation_matrix(L1, L2, interp_vectors, n1):
    """the same between two coarse levels: (n1 V1) x (n2 V2), n2 = 2 Nvec; coarse dof h*Nvec + j of aggregate a couples to
    the dofs h*n1/2 .. (h+1)*n1/2 - 1 of interpolation vector j on the sites of a (src/interpolation_generic.c:111-120 is level
    independent: the first half of a site's dofs is chirality 0)"""
    P_ = cplx(interp_vectors)               # [N][V1][n1]
    N, V = P_.shape[0], P_.shape[1]
    n2 = 2 * N
    a = aggregate_of(L1, L2)
    rows, cols, vals = [], [], []
    site = np.arange(V)
    for j in range(N):
        for d in range(n1):
            h = d // (n1 // 2)
            rows.append(n1 * site + d); cols.append(n2 * a + h * N + j); vals.append(P_[j, :, d])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n1 * V, n2 * int(np.prod(L2))))


# ---- coarse operator --------------------------------------------------------------------------------
def coarse_matrix(Lc, coarse_D, coarse_clover, n, parts=False):
    """(n Vc) x (n Vc) sparse matrix of  eta(x) = M(x) phi(x) - sum_mu [ U_mu(x) phi(x+mu) + G5 U_mu(x-mu)^H G5 phi(x-mu) ];
    parts=True: the tuple (self couplings, [forward hops of mu], [backward hops of mu]) whose sum it is"""
    Vc = int(np.prod(Lc)); N = n // 2
    Dc = cplx(coarse_D).reshape(Vc, 4, 4, N, N)      # [site][mu][block A,C,B,D][col][row]  (column-major blocks)
    cl = cplx(coarse_clover).reshape(Vc, -1)
    g5 = np.concatenate([np.ones(N), -np.ones(N)])
    c = coords(Lc)
    blocks = {}
    pblocks = [dict() for _ in range(9)]      # 0 self, 1+mu forward, 5+mu backward
    which = [0]

    def add(i, j, M):
        blocks[(i, j)] = blocks.get((i, j), 0) + M
        d = pblocks[which[0]]
        d[(i, j)] = d.get((i, j), 0) + M
    tri = N * (N + 1) // 2
    for x in range(Vc):
        M = np.zeros((n, n), dtype=complex)
        for b in range(2):
            p = cl[x, b * tri:(b + 1) * tri]
            k = 0
            for j in range(N):
                for i in range(j + 1):
                    M[b * N + i, b * N + j] = p[k]
                    if i != j:
                        M[b * N + j, b * N + i] = np.conj(p[k])
                    k += 1
        B = cl[x, 2 * tri:2 * tri + N * N].reshape(N, N).T        # column-major -> B[i][j]
        M[:N, N:] = B
        M[N:, :N] = -B.conj().T
        which[0] = 0
        add(x, x, M)
        for mu in range(4):
            A_, C_, B_, D_ = (Dc[x, mu, q].T for q in range(4))   # column-major blocks -> [row][col]
            U = np.block([[A_, B_], [C_, D_]])
            cc = c[x].copy(); cc[mu] = (cc[mu] + 1) % Lc[mu]
            y = int(lex(cc, Lc))
            which[0] = 1 + mu
            add(x, y, -U)                                          # forward:  -U_mu(x) phi(x+mu)
            which[0] = 5 + mu
            add(y, x, -(g5[:, None] * U.conj().T * g5[None, :]))   # backward at y = x+mu:  -G5 U_mu(x)^H G5 phi(x)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")

    def assemble(bl):
        rows, cols, vals = [], [], []
        for (i, j), M in bl.items():
            rows.append((n * i + ii).ravel()); cols.append((n * j + jj).ravel()); vals.append(M.ravel())
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * Vc, n * Vc))
    if parts:
        return assemble(pblocks[0]), [assemble(pblocks[1 + mu]) for mu in range(4)], [assemble(pblocks[5 + mu]) for mu in range(4)]
    return assemble(blocks)


def galerkin_coarse_operator(L1, L2, parts1, P1, n2):
    """The next level's operator in the reference's storage, (D [V2][4][n2*n2], clover [V2][n2(n2+1)/2]) complex, from the
    parts of this level's operator and the interpolation matrix (coarse_operator_PRECISION_setup, set_coarse_self_coupling /
    set_coarse_neighbor_coupling, src/coarse_operator_generic.c:53-205): the self coupling of aggregate X collects P_X^H (self
    couplings + every hop that stays inside X) P_X, the forward link U_mu(X) = -P_X^H (forward hops in mu that leave X) P_{X+mu}."""
    selfp, fwd, bwd = parts1
    V2 = int(np.prod(L2)); N = n2 // 2; tri = N * (N + 1) // 2
    c2 = coords(L2)
    Ph = P1.conj().T.tocsr()
    G_self = (Ph @ selfp @ P1).tocsr()
    G_f = [(Ph @ fwd[mu] @ P1).tocsr() for mu in range(4)]
    G_b = [(Ph @ bwd[mu] @ P1).tocsr() for mu in range(4)]
    D = np.zeros((V2, 4, 4, N, N), dtype=complex)
    cl = np.zeros((V2, 2 * tri + N * N), dtype=complex)
    blk = lambda G, X, Y: G[n2 * X:n2 * (X + 1), n2 * Y:n2 * (Y + 1)].toarray()
    for X in range(V2):
        M = blk(G_self, X, X)
        for mu in range(4):
            M = M + blk(G_f[mu], X, X) + blk(G_b[mu], X, X)
        for b in range(2):
            k = 0
            for j in range(N):
                for i in range(j + 1):
                    cl[X, b * tri + k] = M[b * N + i, b * N + j]; k += 1
        cl[X, 2 * tri:] = M[:N, N:].T.ravel()
        for mu in range(4):
            cc = c2[X].copy(); cc[mu] = (cc[mu] + 1) % L2[mu]
            Y = int(lex(cc, L2))
            U = -blk(G_f[mu], X, Y)
            for q, bq in enumerate((U[:N, :N], U[N:, :N], U[:N, N:], U[N:, N:])):   # blocks A, C, B, D, column-major
                D[X, mu, q] = bq.T
    return D.reshape(V2, 4, n2 * n2), cl


# ---- Krylov -----------------------------------------------------------------------------------------
def fgmres(op, b, tol, restart, max_restart, prec=None, x0=None, breakdown_tol=None):
    """right-preconditioned restarted FGMRES; returns (x, iterations, history of gamma_{j+1}/||r0||)"""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    it = 0; hist = []; norm_r0 = 1.0; finish = False
    for ol in range(max_restart):
        r = b.copy() if (ol == 0 and x0 is None) else b - op(x)
        gamma0 = np.linalg.norm(r)
        if ol == 0:
            norm_r0 = gamma0
        if not gamma0 > 0:
            break
        Vb = [r / gamma0]; Zb = []
        H = np.zeros((restart + 2, restart + 1), dtype=complex)
        gam = np.zeros(restart + 2, dtype=complex); gam[0] = gamma0
        cs = np.zeros(restart + 1, dtype=complex); sn = np.zeros(restart + 1, dtype=complex)
        j = -1
        for il in range(restart):
            j = il; it += 1
            z = prec(Vb[j]) if prec is not None else Vb[j]
            Zb.append(z)
            w = op(z)
            # classical Gram-Schmidt: all inner products against the unmodified w (src/linsolve_generic.c:836-860)
            h = np.array([np.vdot(v, w) for v in Vb])
            for i, v in enumerate(Vb):
                w = w - h[i] * v
            H[:j + 1, j] = h
            H[j + 1, j] = np.linalg.norm(w)
            Vb.append(w / H[j + 1, j] if abs(H[j + 1, j]) > 1e-15 else w)
            if abs(H[j + 1, j]) > (tol / 10 if breakdown_tol is None else breakdown_tol):
                for i in range(j):
                    beta = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                    H[i, j] = np.conj(cs[i]) * H[i, j] + np.conj(sn[i]) * H[i + 1, j]
                    H[i + 1, j] = beta
                beta = np.sqrt(abs(H[j, j]) ** 2 + abs(H[j + 1, j]) ** 2)
                sn[j] = H[j + 1, j] / beta; cs[j] = H[j, j] / beta
                gam[j + 1] = -sn[j] * gam[j]; gam[j] = np.conj(cs[j]) * gam[j]
                H[j, j] = beta; H[j + 1, j] = 0
                hist.append(abs(gam[j + 1]) / norm_r0)
                if hist[-1] < tol or hist[-1] > 1e5:
                    finish = True
                    break
            else:
                finish = True
                break
        if j >= 0:
            y = np.zeros(j + 1, dtype=complex)
            for i in range(j, -1, -1):
                y[i] = (gam[i] - H[i, i + 1:j + 1] @ y[i + 1:]) / H[i, i]
            for i in range(j + 1):
                x = x + y[i] * Zb[i]
        if finish:
            break
    return x, it, hist


# ---- coarsest-level solve ---------------------------------------------------------------------------
def coarse_solve(Mc, Lc, n, b, tol, restart, max_restart):
    """even-odd reduced solve: S = D_ee - D_eo D_oo^-1 D_oe on the even sites, then the odd sites by back substitution"""
    par = coords(Lc).sum(axis=1) % 2
    dof_par = np.repeat(par, n)
    e = np.nonzero(dof_par == 0)[0]; o = np.nonzero(dof_par == 1)[0]
    Mc = Mc.tocsr()
    Dee, Deo, Doe, Doo = Mc[e][:, e], Mc[e][:, o], Mc[o][:, e], Mc[o][:, o]
    Doo_inv = spla.splu(Doo.tocsc())

    def S(v):
        return Dee @ v - Deo @ Doo_inv.solve(Doe @ v)
    xo = Doo_inv.solve(b[o])
    be = b[e] - Deo @ xo
    xe, it, _ = fgmres(S, be, tol, restart, max_restart)
    x = np.zeros_like(b)
    x[e] = xe
    x[o] = Doo_inv.solve(b[o] - Doe @ xe)
    return x, it


# ---- red-black Schwarz ------------------------------------------------------------------------------
class Schwarz:
    SIGMA = [0, 1, 3, 2, 6, 4, 5, 7, 15, 14, 12, 13, 9, 11, 10, 8]   # src/schwarz_generic.c:335

    def __init__(self, L, B, A, block_iter, method=2, ndof=12, odd_even=True):
        """A: matrix of the level's operator (ndof per site; odd_even=False: MinRes on the whole block, the smoother of an
        intermediate level, coarse_block_operator src/coarse_operator_generic.c:208-235); blocks of extent B; the 8 block lists of the reference (colour x {inner, -boundary only,
        both boundaries, +boundary only}, src/schwarz_generic.c:383-428).  method: 1 additive, 2 red-black, 3 sixteen
        colours (src/schwarz_generic.c:318-333; an odd number of blocks in a direction falls back to two colours)"""
        self.A = A.tocsr(); self.block_iter = block_iter; self.method = method; self.odd_even = odd_even
        self.sixteen = method == 3 and all((L[mu] // B[mu]) % 2 == 0 for mu in range(4))
        c = coords(L)
        nblk = [L[mu] // B[mu] for mu in range(4)]
        bc = c // np.array(B)
        self.block_of = lex(bc, nblk)
        self.blocks = []
        sitepar = c.sum(axis=1) % 2
        for b in range(int(np.prod(nblk))):
            sites = np.nonzero(self.block_of == b)[0]
            gb = coords(nblk)[b]
            minus = int(np.sum(gb == 0)); plus = int(np.sum(gb + 1 == np.array(nblk))); inner = int(np.sum((gb != 0) & (gb + 1 != np.array(nblk))))
            col = int(gb.sum() % 2)
            if inner == 4:
                lst = 4 * col
            elif minus == 0:
                lst = 1 if col == 0 else 7
            elif plus == 0:
                lst = 3 if col == 0 else 5
            else:
                lst = 2 + 4 * col
            ev = sites[sitepar[sites] == 0]; od = sites[sitepar[sites] == 1]
            dofs = lambda s: (ndof * s[:, None] + np.arange(ndof)[None, :]).ravel()
            I = dofs(sites); Ie = dofs(ev); Io = dofs(od)
            Abb_e_e = self.A[Ie][:, Ie]; Aeo = self.A[Ie][:, Io]; Aoe = self.A[Io][:, Ie]; Aoo = self.A[Io][:, Io]
            corner = 8 * (gb[0] % 2) + 4 * (gb[1] % 2) + 2 * (gb[2] % 2) + (gb[3] % 2)
            if method == 1:
                col = 0
            elif self.sixteen:
                col = self.SIGMA.index(int(corner))
            self.blocks.append(dict(colour=col, list=lst, I=I, Ie=Ie, Io=Io, Dee=Abb_e_e, Deo=Aeo, Doe=Aoe,
                                    Doo_inv=spla.splu(Aoo.tocsc()) if odd_even else None, rows=self.A[I],
                                    Dbb=None if odd_even else self.A[I][:, I]))

    def _outside(self, blk, v):
        """(couplings from other blocks into this block) * v"""
        full = blk["rows"] @ v
        inside = self.A[blk["I"]][:, blk["I"]] @ v[blk["I"]]
        return full - inside

    def _block_solve(self, blk, x, r, latest):
        if not self.odd_even:
            # local_minres_PRECISION (src/linsolve_generic.c:985-1029) on the block operator itself
            I, Dbb = blk["I"], blk["Dbb"]
            rb = r[I].copy(); d = np.zeros_like(rb)
            for _ in range(self.block_iter):
                Dr = Dbb @ rb
                dn = np.vdot(Dr, Dr).real
                alpha = np.vdot(Dr, rb) / dn if dn > 0 else 0.0
                d = d + alpha * rb
                rb = rb - alpha * Dr
            x[I] += d; latest[I] = d; r[I] = rb
            return
        Ie, Io = blk["Ie"], blk["Io"]
        Dee, Deo, Doe, Dinv = blk["Dee"], blk["Deo"], blk["Doe"], blk["Doo_inv"]
        ro = r[Io].copy()
        re = r[Ie] - Deo @ Dinv.solve(ro)
        de = np.zeros_like(re)
        for _ in range(self.block_iter):                      # local_minres on the Schur complement
            Dr = Dee @ re - Deo @ Dinv.solve(Doe @ re)
            dn = np.vdot(Dr, Dr).real
            alpha = np.vdot(Dr, re) / dn if dn > 0 else 0.0
            de = de + alpha * re
            re = re - alpha * Dr
        do = Dinv.solve(ro - Doe @ de)
        x[Ie] += de; x[Io] += do
        latest[Ie] = de; latest[Io] = do
        r[Ie] = re; r[Io] = 0

    def smooth(self, eta, cycles, phi0=None):
        if self.method != 2:
            return self._smooth_generic(eta, cycles, phi0)
        x = np.zeros_like(eta) if phi0 is None else phi0.copy()
        r = eta.copy()
        latest = np.zeros_like(eta)
        from_zero = phi0 is None
        for k in range(cycles):
            for colour in (0, 1):
                for blk in self.blocks:
                    if blk["colour"] != colour:
                        continue
                    if k == 0 and not from_zero:
                        r[blk["I"]] = eta[blk["I"]] - blk["rows"] @ x
                    elif k == 0 and from_zero:
                        # the residual update is switched on only after list 5 of the first sweep (src/schwarz_generic.c:1344)
                        if colour == 1 and blk["list"] not in (4, 5):
                            r[blk["I"]] -= self._outside(blk, latest)
                    else:
                        r[blk["I"]] -= self._outside(blk, latest)
                # all blocks of one colour are independent: solve them after the updates of this colour
                for blk in self.blocks:
                    if blk["colour"] == colour:
                        self._block_solve(blk, x, r, latest)
        return x


    def _smooth_generic(self, eta, cycles, phi0):
        """additive_schwarz_PRECISION (src/schwarz_generic.c:1077-1257), sixteen_color_schwarz_PRECISION (:1652-1804)
        and the two-colour schwarz_PRECISION (:1433-1650) it falls back to"""
        x = np.zeros_like(eta) if phi0 is None else phi0.copy()
        r = eta.copy()
        latest = np.zeros_like(eta) if (phi0 is None or self.method != 1) else phi0.copy()
        have_res = phi0 is not None
        ncol = 1 if self.method == 1 else 16 if self.sixteen else 2
        for k in range(cycles):
            new = np.zeros_like(eta) if self.method == 1 else latest    # additive: two generations of updates
            for colour in range(ncol):
                mine = [blk for blk in self.blocks if blk["colour"] == colour]
                for blk in mine:
                    if not have_res:
                        continue
                    full = k == 0 if (self.method == 1 or self.sixteen) else (k == 0 and phi0 is not None)
                    if full:
                        r[blk["I"]] = eta[blk["I"]] - blk["rows"] @ (latest if self.method == 1 else x)
                    else:
                        r[blk["I"]] -= self._outside(blk, latest)
                for blk in mine:
                    self._block_solve(blk, x, r, new)
                have_res = True
            latest = new
        return x


class GmresSmoother:
    """smoother_PRECISION with g.method == 4 and odd-even (src/vcycle_generic.c:48-71): solve_oddeven_PRECISION
    (src/oddeven_generic.c:740-777) -- GMRES(block_iter), `cycles` restarts, tolerance EPS_float, on the Schur complement
    D_ee - D_eo D_oo^-1 D_oe of the global odd-even splitting, from a zero start on the residual."""

    def __init__(self, L, A, block_iter):
        self.A = A.tocsr(); self.block_iter = block_iter
        par = coords(L).sum(axis=1) % 2
        dofs = lambda s: (12 * s[:, None] + np.arange(12)[None, :]).ravel()
        self.Ie = dofs(np.nonzero(par == 0)[0]); self.Io = dofs(np.nonzero(par == 1)[0])
        self.Dee = self.A[self.Ie][:, self.Ie]; self.Deo = self.A[self.Ie][:, self.Io]; self.Doe = self.A[self.Io][:, self.Ie]
        self.Doo_inv = spla.splu(self.A[self.Io][:, self.Io].tocsc())

    def smooth(self, eta, cycles, phi0=None):
        b = eta if phi0 is None else eta - self.A @ phi0
        bo = b[self.Io]
        be = b[self.Ie] - self.Deo @ self.Doo_inv.solve(bo)
        S = lambda v: self.Dee @ v - self.Deo @ self.Doo_inv.solve(self.Doe @ v)
        xe, _, _ = fgmres(S, be, 1e-6, self.block_iter, cycles)
        x = np.zeros_like(eta)
        x[self.Ie] = xe
        x[self.Io] = self.Doo_inv.solve(bo - self.Doe @ xe)
        return x if phi0 is None else phi0 + x


# ---- V-cycle and solve ------------------------------------------------------------------------------
class TwoLevel:
    def __init__(self, L, Lc, B, D, clover, interp_vectors, coarse_D, coarse_clover, post_smooth_iter=2, block_iter=4,
                 coarse_tol=5e-2, coarse_restart=5, coarse_iter=100, method=2):
        self.L, self.Lc = L, Lc
        self.A = fine_matrix(L, D, clover)
        self.P = interpolation_matrix(L, Lc, interp_vectors)
        self.n = 2 * np.asarray(interp_vectors).shape[0]
        self.Mc = coarse_matrix(Lc, coarse_D, coarse_clover, self.n)
        self.sap = Schwarz(L, B, self.A, block_iter, method) if method != 4 else GmresSmoother(L, self.A, block_iter)
        self.post = post_smooth_iter
        # "coarse grid iterations" is the restart length, "coarse grid restarts" the number of cycles (src/init.c:927-931)
        self.ctol, self.crestart, self.cmax = coarse_tol, coarse_iter, coarse_restart
        self.coarse_its = 0

    def restrict(self, phi):
        return self.P.conj().T @ phi

    def interpolate(self, phic):
        return self.P @ phic

    def vcycle(self, eta):
        bc = self.restrict(eta)
        xc, it = coarse_solve(self.Mc, self.Lc, self.n, bc, self.ctol, self.crestart, self.cmax)
        self.coarse_its += it
        phi = self.interpolate(xc)
        return self.sap.smooth(eta, self.post, phi0=phi)

    def solve(self, b, tol=1e-10, restart=50, max_restart=20):
        self.coarse_its = 0
        return fgmres(lambda v: self.A @ v, b, tol, restart, max_restart, prec=self.vcycle)
