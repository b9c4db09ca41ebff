"""ctypes mirror of include/ddamg_hip.h (same names, argument meaning and error behaviour)."""
import ctypes, os, re
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDAMG_HIP_LIBRARY: another build of the same library (A/B measurements of a kernel change on one GPU box); never a fallback
_LIBNAME = os.environ.get("DDAMG_HIP_LIBRARY") or os.path.join(_HERE, "libddamg_hip.so")
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "ddamg_hip.h")
MAX_LEVELS = 4


class DDAMGError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """struct ddamg_hip_params (include/ddamg_hip.h); lattices in T,Z,Y,X order."""
    _fields_ = [
        ("num_levels", ctypes.c_int),
        ("local_lattice", (ctypes.c_int * 4) * MAX_LEVELS),
        ("block_lattice", (ctypes.c_int * 4) * MAX_LEVELS),
        ("num_vect", ctypes.c_int * MAX_LEVELS),
        ("post_smooth_iter", ctypes.c_int * MAX_LEVELS),
        ("block_iter", ctypes.c_int * MAX_LEVELS),
        ("setup_iter", ctypes.c_int * MAX_LEVELS),
        ("restart", ctypes.c_int), ("max_restart", ctypes.c_int),
        ("tol", ctypes.c_double),
        ("coarse_iter", ctypes.c_int), ("coarse_restart", ctypes.c_int),
        ("coarse_tol", ctypes.c_double),
        ("kcycle", ctypes.c_int), ("kcycle_restart", ctypes.c_int), ("kcycle_max_restart", ctypes.c_int),
        ("kcycle_tol", ctypes.c_double),
        ("mixed_precision", ctypes.c_int),
        ("odd_even", ctypes.c_int),
        ("method", ctypes.c_int),
        ("m0", ctypes.c_double), ("csw", ctypes.c_double),
        ("device", ctypes.c_int),
        ("process_grid", ctypes.c_int * 4),
        ("process_coords", ctypes.c_int * 4),
        ("test_vector_rng", ctypes.c_int),
        ("rng_seed", ctypes.c_ulonglong),
        ("gather_coarsest", ctypes.c_int),
    ]


class HaloMsg(ctypes.Structure):
    """struct ddamg_hip_halo_msg"""
    _fields_ = [
        ("send_peer", ctypes.c_int), ("recv_peer", ctypes.c_int), ("tag", ctypes.c_int),
        ("send", ctypes.c_void_p), ("recv", ctypes.c_void_p), ("bytes", ctypes.c_ulonglong),
    ]


EXCHANGE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(HaloMsg))
ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int)


_lib = None


def library_path():
    return _LIBNAME


def declared_symbols():
    """every function include/ddamg_hip.h and include/ddamg_hip_io.h declare"""
    txt = open(_HEADER).read() + open(os.path.join(os.path.dirname(_HEADER), "ddamg_hip_io.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddamg_hip_[a-z0-9_]+)\s*\(", txt)))


def load_library():
    """Load libddamg_hip.so; raises DDAMGError (never falls back to anything) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBNAME):
        raise DDAMGError(f"{_LIBNAME} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                         "there is no CPU fallback for the HIP path")
    lib = ctypes.CDLL(_LIBNAME)
    vp = ctypes.c_void_p
    dp = ctypes.POINTER(ctypes.c_double)
    lib.ddamg_hip_last_error.restype = ctypes.c_char_p
    lib.ddamg_hip_comm_stats.restype = ctypes.c_char_p
    lib.ddamg_hip_comm_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ddamg_hip_default_params.argtypes = [ctypes.POINTER(Params)]
    lib.ddamg_hip_default_params.restype = None
    sigs = {
        "ddamg_hip_create": [ctypes.POINTER(Params), ctypes.POINTER(vp)],
        "ddamg_hip_destroy": [vp],
        "ddamg_hip_set_gauge": [vp, dp, ctypes.c_int, dp],
        "ddamg_hip_set_gauge2": [vp, dp, dp, ctypes.c_int, dp],
        "ddamg_hip_set_operator": [vp, dp, dp],
        "ddamg_hip_shift_mass": [vp, ctypes.c_double],
        "ddamg_hip_setup_at_mass": [vp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_scale_clover": [vp, ctypes.c_double, ctypes.c_double],
        "ddamg_hip_get_operator": [vp, dp, dp],
        "ddamg_hip_vec_create": [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)],
        "ddamg_hip_vec_destroy": [vp, vp],
        "ddamg_hip_vec_upload": [vp, vp, dp],
        "ddamg_hip_vec_download": [vp, vp, dp],
        "ddamg_hip_dirac_apply": [vp, vp, vp],
        "ddamg_hip_vec_copy": [vp, vp, vp],
        "ddamg_hip_vec_axpy": [vp, vp, vp, vp, ctypes.c_double, ctypes.c_double],
        "ddamg_hip_vec_dot": [vp, vp, vp, dp, dp, dp],
        "ddamg_hip_setup": [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_setup_update": [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_set_test_vectors": [vp, dp, ctypes.c_int],
        "ddamg_hip_get_interpolation": [vp, dp],
        "ddamg_hip_get_test_vectors": [vp, dp],
        "ddamg_hip_get_coarse_operator": [vp, dp, dp],
        "ddamg_hip_set_coarse_operator": [vp, dp, dp],
        "ddamg_hip_get_coarse_operator_level": [vp, ctypes.c_int, dp, dp],
        "ddamg_hip_set_coarse_operator_level": [vp, ctypes.c_int, dp, dp],
        "ddamg_hip_set_interpolation_level": [vp, ctypes.c_int, dp],
        "ddamg_hip_kcycle": [vp, vp, vp, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_coarse_apply_many": [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "ddamg_hip_smoother_many": [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int],
        "ddamg_hip_vcycle_many": [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "ddamg_hip_kcycle_many": [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_smoother": [vp, vp, vp, ctypes.c_int, ctypes.c_int],
        "ddamg_hip_restrict": [vp, vp, vp],
        "ddamg_hip_interpolate": [vp, vp, vp, ctypes.c_int],
        "ddamg_hip_coarse_apply": [vp, vp, vp],
        "ddamg_hip_coarse_solve": [vp, vp, vp, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_coarse_solve_many": [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_vcycle": [vp, vp, vp],
        "ddamg_hip_solve": [vp, dp, dp, ctypes.c_double, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), dp],
        "ddamg_hip_solve_vec": [vp, vp, vp, ctypes.c_double, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), dp],
        "ddamg_hip_preconditioner": [vp, dp, dp],
        "ddamg_hip_residual_history": [vp, dp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_get_site_order": [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
        "ddamg_hip_rccl_unique_id": [vp],
        "ddamg_hip_comm_init_rccl": [vp, vp],
        "ddamg_hip_comm_init_host": [vp, EXCHANGE_FN, ALLREDUCE_FN, vp],
        "ddamg_hip_halo_plan": [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3,
        "ddamg_hip_timer_begin": [vp],
        "ddamg_hip_timer_end": [vp, ctypes.POINTER(ctypes.c_float)],
        "ddamg_hip_sync": [vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


class VectorHeader(ctypes.Structure):
    """mirror of ddamg_hip_vector_header (include/ddamg_hip_io.h)"""
    _fields_ = [("vector_type", ctypes.c_char_p), ("m0", ctypes.c_double), ("csw", ctypes.c_double), ("clov_plaq", ctypes.c_double),
                ("hopp_plaq", ctypes.c_double), ("clov_conf_name", ctypes.c_char_p), ("hopp_conf_name", ctypes.c_char_p),
                ("has_eigenvalues", ctypes.c_int), ("eigenvalues", ctypes.POINTER(ctypes.c_double))]


def _io():
    lib = load_library()
    if not getattr(lib, "_io_ready", False):
        ip = ctypes.POINTER(ctypes.c_int); dp = ctypes.POINTER(ctypes.c_double); cs = ctypes.c_char_p
        lib.ddamg_hip_io_last_error.restype = ctypes.c_char_p
        lib.ddamg_hip_conf_info.argtypes = [cs, ctypes.c_int, ip, dp]
        lib.ddamg_hip_read_conf.argtypes = [cs, ip, ip, ip, ctypes.c_int, dp, dp]
        lib.ddamg_hip_write_conf.argtypes = [cs, ip, ip, ip, ctypes.c_int, dp, ctypes.c_double]
        lib.ddamg_hip_read_conf_multi.argtypes = [cs, ip, ip, ip, ctypes.c_int, dp, dp]
        lib.ddamg_hip_write_conf_multi.argtypes = [cs, ip, ip, ip, ctypes.c_int, dp, ctypes.c_double]
        lib.ddamg_hip_read_vectors.argtypes = [cs, ip, ip, ip, ctypes.c_int, ctypes.c_int, dp]
        lib.ddamg_hip_write_vectors.argtypes = [cs, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.POINTER(VectorHeader), dp]
        lib._io_ready = True
    return lib


def _io_check(rc):
    if rc != 0:
        raise DDAMGError(_io().ddamg_hip_io_last_error().decode())


def _i4(v):
    return (ctypes.c_int * 4)(*[int(x) for x in v])


def conf_info(path, big_endian=False):
    """(lattice [T,Z,Y,X], plaquette) stored in a configuration file of the reference's format (src/io.c:489-507)"""
    L = (ctypes.c_int * 4)(); plaq = ctypes.c_double(0)
    _io_check(_io().ddamg_hip_conf_info(os.fsencode(path), int(big_endian), L, ctypes.byref(plaq)))
    return list(L), plaq.value


def read_conf(path, global_lattice, process_grid=(1, 1, 1, 1), process_coords=(0, 0, 0, 0), big_endian=False):
    """this process's part of a configuration file: ([V_local][4][9][2] links, plaquette of the header)"""
    Vloc = int(np.prod([g // max(p, 1) for g, p in zip(global_lattice, process_grid)]))
    out = np.zeros((Vloc, 4, 9, 2)); plaq = ctypes.c_double(0)
    _io_check(_io().ddamg_hip_read_conf(os.fsencode(path), _i4(global_lattice), _i4(process_grid), _i4(process_coords), int(big_endian), _dp(out), ctypes.byref(plaq)))
    return out, plaq.value


def write_conf(path, global_lattice, gauge_local, plaq, process_grid=(1, 1, 1, 1), process_coords=(0, 0, 0, 0), big_endian=False):
    a = np.ascontiguousarray(gauge_local, dtype=np.float64)
    _io_check(_io().ddamg_hip_write_conf(os.fsencode(path), _i4(global_lattice), _i4(process_grid), _i4(process_coords), int(big_endian), _dp(a), float(plaq)))


def read_conf_multi(base, global_lattice, process_grid, process_coords, big_endian=False):
    """this process's file of a multi-file configuration (read_conf_multi, src/io.c:566-668): <base>.pt<T>pz<Z>py<Y>px<X>"""
    Vloc = int(np.prod([g // max(p, 1) for g, p in zip(global_lattice, process_grid)]))
    out = np.zeros((Vloc, 4, 9, 2)); plaq = ctypes.c_double(0)
    _io_check(_io().ddamg_hip_read_conf_multi(os.fsencode(base), _i4(global_lattice), _i4(process_grid), _i4(process_coords), int(big_endian), _dp(out), ctypes.byref(plaq)))
    return out, plaq.value


def write_conf_multi(base, global_lattice, gauge_local, plaq, process_grid, process_coords, big_endian=False):
    a = np.ascontiguousarray(gauge_local, dtype=np.float64)
    _io_check(_io().ddamg_hip_write_conf_multi(os.fsencode(base), _i4(global_lattice), _i4(process_grid), _i4(process_coords), int(big_endian), _dp(a), float(plaq)))


def read_vectors(path, global_lattice, n=1, process_grid=(1, 1, 1, 1), process_coords=(0, 0, 0, 0), big_endian=False):
    """n spinors ([n][V_local][12][2]) from a vector / test-vector file of the reference (src/io.c:704-846, 951-1124)"""
    Vloc = int(np.prod([g // max(p, 1) for g, p in zip(global_lattice, process_grid)]))
    out = np.zeros((n, Vloc, 12, 2))
    _io_check(_io().ddamg_hip_read_vectors(os.fsencode(path), _i4(global_lattice), _i4(process_grid), _i4(process_coords), int(n), int(big_endian), _dp(out)))
    return out


def write_vectors(path, global_lattice, vectors_local, header=None, process_grid=(1, 1, 1, 1), process_coords=(0, 0, 0, 0), big_endian=False):
    """header: None (bare data, one spinor) or a dict with the fields of write_header (src/io.c:671-702):
    vector_type, m0, csw, clov_plaq, hopp_plaq, clov_conf_name, hopp_conf_name, eigenvalues"""
    a = np.ascontiguousarray(vectors_local, dtype=np.float64)
    Vloc = int(np.prod([g // max(p, 1) for g, p in zip(global_lattice, process_grid)]))
    if a.size % (Vloc * 24):
        raise DDAMGError("write_vectors: expected [n][V_local][12] complex numbers")
    n = a.size // (Vloc * 24)
    hp = None
    if header is not None:
        ev = header.get("eigenvalues")
        evbuf = np.ascontiguousarray(ev, dtype=np.float64) if ev is not None else None
        h = VectorHeader(str(header.get("vector_type", "")).encode(), float(header.get("m0", 0)), float(header.get("csw", 0)),
                         float(header.get("clov_plaq", 0)), float(header.get("hopp_plaq", 0)), str(header.get("clov_conf_name", "")).encode(),
                         str(header.get("hopp_conf_name", "")).encode(), int(ev is not None), _dp(evbuf) if evbuf is not None else None)
        hp = ctypes.byref(h)
    _io_check(_io().ddamg_hip_write_vectors(os.fsencode(path), _i4(global_lattice), _i4(process_grid), _i4(process_coords), n, int(big_endian), hp, _dp(a)))


def _check(rc):
    if rc != 0:
        raise DDAMGError(load_library().ddamg_hip_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def default_params():
    p = Params()
    load_library().ddamg_hip_default_params(ctypes.byref(p))
    return p


class Vector:
    def __init__(self, ctx, level, precision):
        self.ctx, self.level, self.precision = ctx, level, precision
        self.ndof = ctx.ndof(level)
        self.V = ctx.volume(level)
        self._h = ctypes.c_void_p()
        _check(ctx._lib.ddamg_hip_vec_create(ctx._h, level, precision, ctypes.byref(self._h)))

    def upload(self, host_lex):
        a = np.ascontiguousarray(host_lex, dtype=np.float64)
        if a.size != self.V * self.ndof * 2:
            raise DDAMGError(f"vector upload: expected {self.V * self.ndof * 2} reals, got {a.size}")
        _check(self.ctx._lib.ddamg_hip_vec_upload(self.ctx._h, self._h, _dp(a)))
        return self

    def download(self):
        out = np.empty((self.V, self.ndof, 2), dtype=np.float64)
        _check(self.ctx._lib.ddamg_hip_vec_download(self.ctx._h, self._h, _dp(out)))
        return out

    def free(self):
        if self._h:
            self.ctx._lib.ddamg_hip_vec_destroy(self.ctx._h, self._h)
            self._h = ctypes.c_void_p()


class Context:
    """Mirror of ddamg_hip_ctx; one per process and GPU."""

    def __init__(self, params):
        self._lib = load_library()
        self.params = params
        self._h = ctypes.c_void_p()
        _check(self._lib.ddamg_hip_create(ctypes.byref(params), ctypes.byref(self._h)))

    def volume(self, level=0):
        return int(np.prod(list(self.params.local_lattice[level])))

    def ndof(self, level=0):
        return 12 if level == 0 else 2 * self.params.num_vect[level - 1]

    def set_gauge(self, gauge_lex, anti_pbc=True):
        a = np.ascontiguousarray(gauge_lex, dtype=np.float64)
        if a.size != self.volume(0) * 72:
            raise DDAMGError("set_gauge: gauge field must hold V*4*9 complex numbers")
        plaq = ctypes.c_double(0)
        _check(self._lib.ddamg_hip_set_gauge(self._h, _dp(a), int(bool(anti_pbc)), ctypes.byref(plaq)))
        return plaq.value

    def set_gauge2(self, hopp_gauge_lex, clover_gauge_lex, anti_pbc=False):
        """hopping term from the first field, clover term and plaquette from the second (open boundaries)"""
        a = np.ascontiguousarray(hopp_gauge_lex, dtype=np.float64); b = np.ascontiguousarray(clover_gauge_lex, dtype=np.float64)
        if a.size != self.volume(0) * 72 or b.size != a.size:
            raise DDAMGError("set_gauge2: both gauge fields must hold V*4*9 complex numbers")
        plaq = ctypes.c_double(0)
        _check(self._lib.ddamg_hip_set_gauge2(self._h, _dp(a), _dp(b), int(bool(anti_pbc)), ctypes.byref(plaq)))
        return plaq.value

    def set_operator(self, D_lex, clover_lex):
        D = np.ascontiguousarray(D_lex, dtype=np.float64)
        cl = np.ascontiguousarray(clover_lex, dtype=np.float64)
        V = self.volume(0)
        if D.size != V * 72 or cl.size != V * 84:
            raise DDAMGError("set_operator: D must be [V][36] complex and clover [V][42] complex")
        _check(self._lib.ddamg_hip_set_operator(self._h, _dp(D), _dp(cl)))

    def shift_mass(self, m0):
        """shift_update of the reference: change the mass of the operator that is set, on the device, on every level"""
        _check(self._lib.ddamg_hip_shift_mass(self._h, float(m0)))

    def scale_clover(self, scale_even, scale_odd):
        """clover term times scale_even / scale_odd by global parity, on the device (absolute: (1, 1) restores the operator)"""
        _check(self._lib.ddamg_hip_scale_clover(self._h, float(scale_even), float(scale_odd)))

    def get_operator(self):
        V = self.volume(0)
        D = np.empty((V, 36, 2)); cl = np.empty((V, 42, 2))
        _check(self._lib.ddamg_hip_get_operator(self._h, _dp(D), _dp(cl)))
        return D, cl

    def vector(self, level=0, precision=32):
        return Vector(self, level, precision)

    def dirac_apply(self, out, inp):
        _check(self._lib.ddamg_hip_dirac_apply(self._h, out._h, inp._h))

    # ---- BLAS-1 ----
    def vec_copy(self, dst, src):
        _check(self._lib.ddamg_hip_vec_copy(self._h, dst._h, src._h))

    def vec_axpy(self, z, x, y, alpha):
        alpha = complex(alpha)
        _check(self._lib.ddamg_hip_vec_axpy(self._h, z._h, x._h, y._h, alpha.real, alpha.imag))

    def vec_dot(self, x, y):
        """returns (<x,y>, ||x||)"""
        re = ctypes.c_double(0); im = ctypes.c_double(0); nx = ctypes.c_double(0)
        _check(self._lib.ddamg_hip_vec_dot(self._h, x._h, y._h, ctypes.byref(re), ctypes.byref(im), ctypes.byref(nx)))
        return complex(re.value, im.value), nx.value

    # ---- multigrid ----
    def vprec(self):
        return 64 if self.params.mixed_precision == 0 else 32

    def setup(self, setup_iterations=-1):
        ci = ctypes.c_int(0)
        _check(self._lib.ddamg_hip_setup(self._h, int(setup_iterations), ctypes.byref(ci)))
        return ci.value

    def setup_update(self, iterations):
        ci = ctypes.c_int(0)
        _check(self._lib.ddamg_hip_setup_update(self._h, int(iterations), ctypes.byref(ci)))
        return ci.value

    def set_test_vectors(self, tv_lex, orthonormalised=False):
        a = np.ascontiguousarray(tv_lex, dtype=np.float64)
        if a.size != self.params.num_vect[0] * self.volume(0) * 24:
            raise DDAMGError("set_test_vectors: expected [num_vect][V][12] complex numbers")
        _check(self._lib.ddamg_hip_set_test_vectors(self._h, _dp(a), int(bool(orthonormalised))))

    def get_interpolation(self):
        out = np.empty((self.params.num_vect[0], self.volume(0), 12, 2))
        _check(self._lib.ddamg_hip_get_interpolation(self._h, _dp(out)))
        return out

    def get_test_vectors(self):
        out = np.zeros((self.params.num_vect[0], self.volume(0), 12, 2))
        _check(self._lib.ddamg_hip_get_test_vectors(self._h, _dp(out)))
        return out

    def get_coarse_operator(self, level=1):
        n = self.ndof(level); Vc = self.volume(level)
        D = np.empty((Vc, 4, n * n, 2)); cl = np.empty((Vc, n * (n + 1) // 2, 2))
        _check(self._lib.ddamg_hip_get_coarse_operator_level(self._h, int(level), _dp(D), _dp(cl)))
        return D, cl

    def set_coarse_operator(self, D, cl, level=1):
        n = self.ndof(level); Vc = self.volume(level)
        D = np.ascontiguousarray(D, dtype=np.float64); cl = np.ascontiguousarray(cl, dtype=np.float64)
        if D.size != Vc * 4 * n * n * 2 or cl.size != Vc * n * (n + 1):
            raise DDAMGError("set_coarse_operator: wrong array sizes")
        _check(self._lib.ddamg_hip_set_coarse_operator_level(self._h, int(level), _dp(D), _dp(cl)))

    def set_interpolation(self, P_lex, level=0):
        """interpolation vectors of `level` as they are ([num_vect][V][ndof][2], lexicographic sites of that level); builds the
        operator of level + 1 and runs the initial setup of the levels below"""
        P = np.ascontiguousarray(P_lex, dtype=np.float64)
        if P.size != self.params.num_vect[level] * self.volume(level) * self.ndof(level) * 2:
            raise DDAMGError("set_interpolation: wrong array size")
        _check(self._lib.ddamg_hip_set_interpolation_level(self._h, int(level), _dp(P)))

    def kcycle(self, x, b):
        it = ctypes.c_int(0)
        _check(self._lib.ddamg_hip_kcycle(self._h, x._h, b._h, ctypes.byref(it)))
        return it.value

    def _many(self, fn, outs, ins, *extra):
        n = len(ins)
        VP = ctypes.c_void_p * n
        _check(fn(self._h, n, VP(*[v._h for v in outs]), VP(*[v._h for v in ins]), *extra))

    def coarse_apply_many(self, outs, ins):
        """the coarse operator of the vectors' level for up to 32 right-hand sides at once (matrix cores)"""
        self._many(self._lib.ddamg_hip_coarse_apply_many, outs, ins)

    def smoother_many(self, phis, etas, cycles, initial_guess_zero=True):
        self._many(self._lib.ddamg_hip_smoother_many, phis, etas, int(cycles), int(bool(initial_guess_zero)))

    def vcycle_many(self, phis, etas):
        self._many(self._lib.ddamg_hip_vcycle_many, phis, etas)

    def kcycle_many(self, xs, bs):
        its = (ctypes.c_int * len(bs))()
        self._many(self._lib.ddamg_hip_kcycle_many, xs, bs, its)
        return list(its)

    def smoother(self, phi, eta, cycles, initial_guess_zero=True):
        _check(self._lib.ddamg_hip_smoother(self._h, phi._h, eta._h, int(cycles), int(bool(initial_guess_zero))))

    def restrict(self, coarse, fine):
        _check(self._lib.ddamg_hip_restrict(self._h, coarse._h, fine._h))

    def interpolate(self, fine, coarse, add=False):
        _check(self._lib.ddamg_hip_interpolate(self._h, fine._h, coarse._h, int(bool(add))))

    def coarse_apply(self, out, inp):
        _check(self._lib.ddamg_hip_coarse_apply(self._h, out._h, inp._h))

    def coarse_solve(self, x, b):
        it = ctypes.c_int(0)
        _check(self._lib.ddamg_hip_coarse_solve(self._h, x._h, b._h, ctypes.byref(it)))
        return it.value

    def coarse_solve_many(self, xs, bs):
        """the coarsest-level solve for up to 32 right-hand sides in lockstep (matrix-core coarse operator); returns the list of
        iteration counts (-1: the column needs the one-at-a-time solver)"""
        n = len(bs)
        VP = ctypes.c_void_p * n
        its = (ctypes.c_int * n)()
        _check(self._lib.ddamg_hip_coarse_solve_many(self._h, n, VP(*[v._h for v in xs]), VP(*[v._h for v in bs]), its))
        return list(its)

    def vcycle(self, phi, eta):
        _check(self._lib.ddamg_hip_vcycle(self._h, phi._h, eta._h))

    def solve(self, b_lex, tol=0.0, out=None):
        """returns (x_lex, iterations, coarse_iterations, true relative residual); `out` reuses a solution array (a fresh
        one is first touched page by page while the device writes into it, which costs more than the solve at 32^4)"""
        b = np.ascontiguousarray(b_lex, dtype=np.float64)
        if b.size != self.volume(0) * 24:
            raise DDAMGError("solve: right-hand side must hold V*12 complex numbers")
        x = out if out is not None else np.empty((self.volume(0), 12, 2))
        if x.dtype != np.float64 or not x.flags.c_contiguous or x.size != b.size:
            raise DDAMGError("solve: `out` must be a C-contiguous float64 array of V*12 complex numbers")
        it = ctypes.c_int(0); ci = ctypes.c_int(0); rr = ctypes.c_double(0)
        _check(self._lib.ddamg_hip_solve(self._h, _dp(x), _dp(b), float(tol), ctypes.byref(it), ctypes.byref(ci), ctypes.byref(rr)))
        return x, it.value, ci.value, rr.value

    def solve_vec(self, x, b, tol=0.0):
        """device-resident solve: x, b fine-level fp64 Vectors; returns (iterations, coarse_iterations, relres)"""
        it = ctypes.c_int(0); ci = ctypes.c_int(0); rr = ctypes.c_double(0)
        _check(self._lib.ddamg_hip_solve_vec(self._h, x._h, b._h, float(tol), ctypes.byref(it), ctypes.byref(ci), ctypes.byref(rr)))
        return it.value, ci.value, rr.value

    def preconditioner(self, in_lex):
        a = np.ascontiguousarray(in_lex, dtype=np.float64)
        out = np.empty((self.volume(0), 12, 2))
        _check(self._lib.ddamg_hip_preconditioner(self._h, _dp(out), _dp(a)))
        return out

    def residual_history(self):
        n = ctypes.c_int(0)
        buf = np.empty(4096)
        _check(self._lib.ddamg_hip_residual_history(self._h, _dp(buf), 4096, ctypes.byref(n)))
        return buf[:n.value].copy()

    def site_order(self, level=0):
        out = np.empty(self.volume(level), dtype=np.int32)
        _check(self._lib.ddamg_hip_get_site_order(self._h, level, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out

    # ---- multi-GPU halo exchange ----
    def comm_stats(self, reset=False):
        """what this process sent since the last reset (dict): halo exchanges by payload, global sums, all-gathers"""
        import json
        return json.loads(self._lib.ddamg_hip_comm_stats(self._h, int(bool(reset))).decode() or "{}")

    def comm_init_rccl(self, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _check(self._lib.ddamg_hip_comm_init_rccl(self._h, ctypes.cast(buf, ctypes.c_void_p)))

    def comm_init_host(self, exchange, allreduce):
        """exchange(list of (send_peer, recv_peer, tag, send: np.uint8 array, recv: np.uint8 array));
        allreduce(np.float64 array) sums it over all processes in place"""
        def trampoline(_user, n, msgs):
            items = []
            for i in range(n):
                m = msgs[i]
                nb = int(m.bytes)
                snd = np.ctypeslib.as_array(ctypes.cast(m.send, ctypes.POINTER(ctypes.c_uint8)), shape=(nb,))
                rcv = np.ctypeslib.as_array(ctypes.cast(m.recv, ctypes.POINTER(ctypes.c_uint8)), shape=(nb,))
                items.append((m.send_peer, m.recv_peer, m.tag, snd, rcv))
            exchange(items)
        def reduce_trampoline(_user, buf, n):
            allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
        self._exchange_cb = EXCHANGE_FN(trampoline)   # keep the callback objects alive
        self._allreduce_cb = ALLREDUCE_FN(reduce_trampoline)
        _check(self._lib.ddamg_hip_comm_init_host(self._h, self._exchange_cb, self._allreduce_cb, None))

    def timer_begin(self):
        _check(self._lib.ddamg_hip_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float(0)
        _check(self._lib.ddamg_hip_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    def sync(self):
        _check(self._lib.ddamg_hip_sync(self._h))

    def close(self):
        if self._h:
            self._lib.ddamg_hip_destroy(self._h)
            self._h = ctypes.c_void_p()


def rccl_unique_id():
    buf = ctypes.create_string_buffer(128)
    _check(load_library().ddamg_hip_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
    return buf.raw


def halo_plan(local_lattice, process_grid, process_coords, face):
    """host-only: (neighbour rank, local lexicographic indices of the face sites in message order)"""
    lib = load_library()
    I4 = ctypes.c_int * 4
    nb = ctypes.c_int(0); cnt = ctypes.c_int(0)
    L, P, C = I4(*local_lattice), I4(*process_grid), I4(*process_coords)
    _check(lib.ddamg_hip_halo_plan(L, P, C, int(face), ctypes.byref(nb), ctypes.byref(cnt), None))
    sites = np.empty(cnt.value, dtype=np.int32)
    if cnt.value:
        _check(lib.ddamg_hip_halo_plan(L, P, C, int(face), ctypes.byref(nb), ctypes.byref(cnt),
                                       sites.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    return nb.value, sites
