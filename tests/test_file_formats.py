"""CPU tests (-m "not gpu"): the reference's on-disk formats (include/ddamg_hip_io.h, host code only) --
gauge configurations (read_conf, src/io.c:459-563) and spinor / test-vector files (vector_io :704-846,
vector_io_single_file :951-1124, write_header :671-702), read and written by every process of a grid for its own part.
The configuration format is pinned by the reference itself: it reads the files this code's Python twin writes
(oracle/make_golden.py, ragged and 16^4 cases), and where the reference tree is present the writer reproduces its
sample configuration byte for byte."""
import os
import numpy as np
import pytest
from conftest import load_golden
from ddalphaamg_amd import api, dist as ddist

GRIDS = [(1, 1, 1, 1), (2, 1, 1, 1), (1, 2, 1, 2), (2, 2, 2, 2)]


def grid_coords(P):
    return [ddist.coords_of(r, list(P)) for r in range(int(np.prod(P)))]


@pytest.fixture(scope="module")
def conf4():
    g = load_golden("ref_4x4.npz")
    return [int(x) for x in g["conf_dims"]], g["gauge"], float(g["conf_plaq"][0])


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("P", GRIDS)
def test_configuration_written_and_read_by_a_process_grid(tmp_path, conf4, P, big_endian):
    L, U, plaq = conf4
    path = tmp_path / "conf"
    for C in grid_coords(P):       # every process writes its rows into the one file
        api.write_conf(path, L, ddist.local_part(U, L, list(P), C), plaq, P, C, big_endian)
    assert os.path.getsize(path) == 16 + 8 + U.size * 8
    assert api.conf_info(path, big_endian) == (L, plaq)
    whole, pl = api.read_conf(path, L, big_endian=big_endian)
    assert pl == plaq and np.array_equal(whole, U)
    for C in grid_coords(P):
        part, _ = api.read_conf(path, L, P, C, big_endian)
        assert np.array_equal(part.reshape(len(part), -1), ddist.local_part(U, L, list(P), C))


def test_configuration_layout_is_the_reference_layout(tmp_path, conf4):
    """header int32 T,Z,Y,X + double plaquette, then [t][z][y][x][mu][3x3] complex doubles, little endian"""
    L, U, plaq = conf4
    path = tmp_path / "conf"
    api.write_conf(path, L, U, plaq)
    raw = open(path, "rb").read()
    assert np.array_equal(np.frombuffer(raw[:16], dtype="<i4"), L)
    assert np.frombuffer(raw[16:24], dtype="<f8")[0] == plaq
    assert np.array_equal(np.frombuffer(raw[24:], dtype="<f8"), U.ravel())
    ref = "/root/reference/conf/4x4x4x4b6.0000id3n1"
    if os.path.exists(ref):        # the reference's own sample configuration, where its tree is mounted
        assert open(ref, "rb").read() == raw


def test_configuration_errors(tmp_path, conf4):
    L, U, plaq = conf4
    path = tmp_path / "conf"
    api.write_conf(path, L, U, plaq)
    with pytest.raises(api.DDAMGError, match="expected 8x4x4x4"):
        api.read_conf(path, [8, 4, 4, 4])
    with pytest.raises(api.DDAMGError, match="cannot open"):
        api.read_conf(tmp_path / "missing", L)
    with pytest.raises(api.DDAMGError, match="does not divide"):
        api.read_conf(path, L, (3, 1, 1, 1), (0, 0, 0, 0))
    open(tmp_path / "short", "wb").write(open(path, "rb").read()[:1000])
    with pytest.raises(api.DDAMGError, match="ends early"):
        api.read_conf(tmp_path / "short", L)


HEADER = dict(vector_type="test vectors", m0=-0.5, csw=1.0, clov_plaq=1.6479691, hopp_plaq=1.6479691, clov_conf_name="conf/4x4x4x4b6.0000id3n1",
              hopp_conf_name="conf/4x4x4x4b6.0000id3n1")


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("P", GRIDS)
def test_test_vector_file_by_a_process_grid(tmp_path, P, big_endian):
    L = [4, 4, 4, 4]; n = 5
    tv = np.random.default_rng(3).standard_normal((n, 256, 12, 2))
    path = tmp_path / "tv"
    for C in grid_coords(P):
        api.write_vectors(path, L, np.stack([ddist.local_part(tv[k], L, list(P), C) for k in range(n)]), HEADER, P, C, big_endian)
    raw = open(path, "rb").read()
    end = raw.index(b"</header>\n") + len(b"</header>\n")
    text = raw[:end].decode().splitlines()
    # write_header, src/io.c:671-702
    assert text[0] == "<header>" and text[1] == "test vectors" and text[-1] == "</header>"
    assert "X local: %d" % (4 // P[3]) in text and "T: 4" in text and "number of vectors: 5" in text and "m0: -0.50000000000000" in text
    assert np.array_equal(np.frombuffer(raw[end:], dtype=">f8" if big_endian else "<f8").reshape(n, 256, 12, 2), tv)
    assert np.array_equal(api.read_vectors(path, L, n, big_endian=big_endian), tv)
    for C in grid_coords(P):
        part = api.read_vectors(path, L, n, P, C, big_endian)
        for k in range(n):
            assert np.array_equal(part[k].reshape(len(part[k]), -1), ddist.local_part(tv[k], L, list(P), C))


def test_single_spinor_with_and_without_header(tmp_path):
    """vector_io accepts both (src/io.c:735-743)"""
    L = [4, 2, 2, 6]
    v = np.random.default_rng(4).standard_normal((1, 96, 12, 2))
    api.write_vectors(tmp_path / "bare", L, v)
    assert os.path.getsize(tmp_path / "bare") == v.size * 8
    api.write_vectors(tmp_path / "hdr", L, v, dict(HEADER, vector_type="solution", eigenvalues=[0.25, -1.5]))
    assert b"eigenvalues: 0.2500000000000000 -1.5000000000000000 \n" in open(tmp_path / "hdr", "rb").read()
    for f in ("bare", "hdr"):
        assert np.array_equal(api.read_vectors(tmp_path / f, L), v)
        assert np.array_equal(api.read_vectors(tmp_path / f, L, 1, (1, 1, 1, 2), (0, 0, 0, 1)).reshape(48, 24), ddist.local_part(v[0], L, [1, 1, 1, 2], [0, 0, 0, 1]))
    with pytest.raises(api.DDAMGError, match="single vector"):
        api.read_vectors(tmp_path / "bare", L, 2)
    with pytest.raises(api.DDAMGError, match="fewer than 3 vectors"):
        api.read_vectors(tmp_path / "hdr", L, 3)


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("P", GRIDS[1:])
def test_multi_file_configuration(tmp_path, conf4, P, big_endian):
    """read_conf_multi (src/io.c:566-668): one file per process, named <base>.pt<T>pz<Z>py<Y>px<X>, each with the header of the
    GLOBAL lattice followed by the process's own links; byte layout checked against the single-file format"""
    L, U, plaq = conf4
    base = str(tmp_path / "conf")
    Vloc = int(np.prod(L)) // int(np.prod(P))
    for C in grid_coords(P):
        api.write_conf_multi(base, L, ddist.local_part(U, L, list(P), C), plaq, P, C, big_endian)
        name = base + ".pt%dpz%dpy%dpx%d" % tuple(C)
        assert os.path.getsize(name) == 16 + 8 + Vloc * 72 * 8
        assert api.conf_info(name, big_endian) == (L, plaq)      # header of the global lattice in every part
        raw = np.fromfile(name, dtype=">f8" if big_endian else "<f8", offset=24)
        assert np.array_equal(raw, ddist.local_part(U, L, list(P), C).ravel())
    for C in grid_coords(P):
        part, pl = api.read_conf_multi(base, L, P, C, big_endian)
        assert pl == plaq and np.array_equal(part.reshape(len(part), -1), ddist.local_part(U, L, list(P), C))
    with pytest.raises(api.DDAMGError):       # a part of another lattice
        api.read_conf_multi(base, [L[0] * 2] + L[1:], P, grid_coords(P)[0], big_endian)
    with pytest.raises(api.DDAMGError):       # a part of another decomposition (wrong size)
        api.read_conf_multi(base, L, (1, 1, 1, 1), (0, 0, 0, 0), big_endian)
