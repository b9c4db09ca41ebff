import os, sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def splitmix_uniform(n, seed):
    """deterministic uniform(-0.5,0.5) stream (same generator as oracle/ref_dump.c urand)"""
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + i * np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / 9007199254740992.0 - 0.5


def random_su3(n, seed):
    """n Haar-ish SU(3) matrices (QR of complex Gaussian, det-normalised), as [n][9][2] float64"""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, 3, 3)) + 1j * rng.standard_normal((n, 3, 3))
    q, r = np.linalg.qr(a)
    d = np.diagonal(r, axis1=1, axis2=2)
    q = q * (d / np.abs(d))[:, None, :]
    det = np.linalg.det(q)
    q = q / (det ** (1.0 / 3.0))[:, None, None]
    out = np.empty((n, 9, 2))
    out[..., 0] = q.reshape(n, 9).real
    out[..., 1] = q.reshape(n, 9).imag
    return out


@pytest.fixture(scope="session")
def gold4():
    return load_golden("ref_4x4.npz")


@pytest.fixture(scope="session")
def gold8():
    return load_golden("ref_8x8_dirac.npz")
