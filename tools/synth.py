"""tools/synth.py -- ctypes access to tools/libsynth_gauge.so (seeded synthetic gauge fields, see tools/synth_gauge.c).
Bench / test input generation only."""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsynth_gauge.so")
_lib = None


def build():
    src = os.path.join(_HERE, "synth_gauge.c")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", _LIB, src, "-lm"])


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        I4 = ctypes.POINTER(ctypes.c_int)
        _lib.synth_gauge.argtypes = [I4, I4, I4, ctypes.c_double, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_double)]
        _lib.synth_gauge.restype = None
    return _lib


def synth_gauge(global_lattice, eps, seed, grid=(1, 1, 1, 1), coords=(0, 0, 0, 0)):
    """[V_local][4][9][2] links of the process at `coords` of `grid`; the global field depends on (eps, seed) only.
    eps > 0: exp(i eps H) (near-unit, smooth); eps <= 0: Haar-like random SU(3)."""
    I4 = ctypes.c_int * 4
    L = [int(g) // int(p) for g, p in zip(global_lattice, grid)]
    V = int(np.prod(L))
    out = np.empty((V, 4, 9, 2))
    _load().synth_gauge(I4(*[int(v) for v in global_lattice]), I4(*[int(v) for v in grid]), I4(*[int(v) for v in coords]),
                        float(eps), int(seed), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def write_conf(path, L, U, plaq=0.0):
    """gauge file in the reference's format (src/io.c:489-520): 4 x int32 (T,Z,Y,X), double plaquette, links"""
    with open(path, "wb") as f:
        f.write(np.asarray(L, dtype="<i4").tobytes()); f.write(np.asarray([plaq], dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(U, dtype="<f8").tobytes())
