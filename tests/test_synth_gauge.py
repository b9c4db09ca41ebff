"""CPU tests of the seeded gauge-field generator behind bench.py and the full-size tests (tools/synth_gauge.c): links are
SU(3), the field is a function of (seed, global position) alone -- so every decomposition of one global lattice, and the
reference reading the field from a file, see the same links -- and the file written for the reference has its format."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import synth
from ddalphaamg_amd import api, dist as ddist


def mats(U):
    return (U[..., 0] + 1j * U[..., 1]).reshape(-1, 3, 3)


def test_links_are_su3_and_seeded():
    G = [4, 6, 4, 8]
    for eps in (0.35, 0.0):          # near-unit exp(i eps H), and Haar-like
        U = synth.synth_gauge(G, eps, 11)
        u = mats(U)
        assert np.abs(u @ u.conj().transpose(0, 2, 1) - np.eye(3)).max() < 1e-12
        assert np.abs(np.linalg.det(u) - 1).max() < 1e-12
        assert np.array_equal(U, synth.synth_gauge(G, eps, 11))
        assert not np.array_equal(U, synth.synth_gauge(G, eps, 12))
    near = mats(synth.synth_gauge(G, 0.35, 11)); haar = mats(synth.synth_gauge(G, 0.0, 11))
    assert np.abs(near - np.eye(3)).mean() < 0.5 * np.abs(haar - np.eye(3)).mean()


def test_every_decomposition_sees_the_same_global_field(tmp_path):
    G = [8, 4, 4, 8]
    U = synth.synth_gauge(G, 0.35, 5)
    for P in ([2, 1, 1, 1], [2, 2, 1, 2], [1, 1, 2, 4]):
        for r in range(int(np.prod(P))):
            C = ddist.coords_of(r, P)
            part = synth.synth_gauge(G, 0.35, 5, P, C)
            assert np.array_equal(part.reshape(len(part), -1), ddist.local_part(U, G, P, C))
    # the file handed to the reference: its configuration format (read back through the library's reader)
    path = tmp_path / "conf"
    synth.write_conf(path, G, U, 0.25)
    assert api.conf_info(path) == (G, 0.25)
    back, _ = api.read_conf(path, G)
    assert np.array_equal(back, U)
