"""oracle/orc.py -- TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/libddamg_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libddamg_oracle.so")
_lib = None

def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])

def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.orc_gauge_to_operator.restype = ctypes.c_double
        _lib.orc_gauge_to_operator.argtypes = [ip, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp]
        for f in (_lib.orc_dirac_apply_f64, _lib.orc_dirac_apply_f32):
            f.restype = None
            f.argtypes = [ip, dp, dp, dp, dp]
        _lib.orc_set_threads.restype = None
        _lib.orc_set_threads.argtypes = [ctypes.c_int]
        _lib.orc_set_threads(host_threads())   # never more than this box's CPU share
        _lib.orc_dirac_time_f32.restype = ctypes.c_double
        _lib.orc_dirac_time_f32.argtypes = [ip, dp, dp, dp, ctypes.c_int, ip]
    return _lib

def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

def _L(L):
    return (ctypes.c_int * 4)(*[int(x) for x in L])

def gauge_to_operator(L, gauge, anti_pbc, m0, csw):
    V = int(np.prod(L))
    gauge = np.ascontiguousarray(gauge, dtype=np.float64).reshape(V, 4, 9, 2)
    D = np.empty((V, 36, 2)); cl = np.empty((V, 42, 2))
    plaq = lib().orc_gauge_to_operator(_L(L), _dp(gauge), int(anti_pbc), float(m0), float(csw), _dp(D), _dp(cl))
    return D, cl, plaq

def dirac_apply(L, D, clover, phi, precision=64):
    V = int(np.prod(L))
    D = np.ascontiguousarray(D, dtype=np.float64); clover = np.ascontiguousarray(clover, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    eta = np.empty((V, 12, 2))
    f = lib().orc_dirac_apply_f64 if precision == 64 else lib().orc_dirac_apply_f32
    f(_L(L), _dp(D), _dp(clover), _dp(phi), _dp(eta))
    return eta

def set_threads(n):
    lib().orc_set_threads(int(n))


def host_threads():
    """threads to use on this box: its CPU share (16 per GPU on the pool), never the raw core count"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def dirac_time_f32(L, D, clover, phi, reps, threads=None):
    nt = ctypes.c_int(threads or host_threads())
    D = np.ascontiguousarray(D, dtype=np.float64); clover = np.ascontiguousarray(clover, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    t = lib().orc_dirac_time_f32(_L(L), _dp(D), _dp(clover), _dp(phi), int(reps), ctypes.byref(nt))
    return t, nt.value
