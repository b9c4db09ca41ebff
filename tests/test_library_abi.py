"""CPU tests: the C-ABI library builds/loads and exports every symbol include/*.h declares
(no compute calls without a GPU)."""
import ctypes, os
import pytest
import ddalphaamg_amd as dd


def test_library_exports_every_declared_symbol():
    if not os.path.exists(dd.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(dd.library_path())
    syms = dd.declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"symbols declared in include/ddamg_hip.h but not exported: {missing}"


def test_library_exports_reference_interface():
    """every function of the reference's dd_alpha_amg.h (src/dd_alpha_amg.h:42-83) is exported"""
    from ddalphaamg_amd import libiface
    lib = ctypes.CDLL(dd.library_path())
    missing = [s for s in libiface.SYMBOLS if not hasattr(lib, s)]
    assert not missing, missing
    import re
    hdr = open(os.path.join(os.path.dirname(dd.library_path()), "..", "include", "dd_alpha_amg.h")).read()
    declared = set(re.findall(r"\b(dd_alpha_amg_[a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    assert declared == set(libiface.SYMBOLS)
    # struct layout: by-value dd_alpha_amg_par must match the C definition
    # sizes as gcc lays out include/dd_alpha_amg.h on x86-64 (checked with a compiled sizeof probe)
    assert ctypes.sizeof(libiface.AmgParameters) == 328
    assert ctypes.sizeof(libiface.Par) == 888


def test_default_params_match_reference_defaults():
    from ddalphaamg_amd import api
    p = api.default_params()
    # reference src/init.c:833-868, 946-953
    assert (p.mixed_precision, p.method, p.odd_even) == (2, 2, 1)
    assert (p.restart, p.max_restart, p.tol) == (10, 100, 1e-10)
    assert (p.coarse_iter, p.coarse_restart, p.coarse_tol) == (25, 40, 5e-2)
    assert (p.kcycle, p.kcycle_restart, p.kcycle_max_restart, p.kcycle_tol) == (1, 5, 2, 1e-1)
    assert list(p.setup_iter)[:3] == [6, 3, 2] and p.num_vect[0] == 20


def test_create_without_gpu_fails_loudly():
    """no silent CPU fallback: without a HIP device the context cannot be created"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ddalphaamg_amd import api
    p = api.default_params()
    p.num_levels = 1
    for mu in range(4):
        p.local_lattice[0][mu] = 4
    with pytest.raises(dd.DDAMGError):
        dd.Context(p)


def test_params_struct_matches_header():
    """the ctypes mirror of ddamg_hip_params has one field per member of the C struct, in order, and the defaults for
    the members added after the reference's own parameters are neutral (single process, reference random numbers)"""
    import re
    from ddalphaamg_amd import api
    hdr = open(os.path.join(os.path.dirname(dd.library_path()), "..", "include", "ddamg_hip.h")).read()
    body = re.search(r"typedef struct ddamg_hip_params \{(.*?)\} ddamg_hip_params;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*(?:\[[^\]]*\])*\s*$", part.strip())[0])
    assert names == [f[0] for f in api.Params._fields_]
    p = api.default_params()
    assert list(p.process_grid) == [1, 1, 1, 1] and list(p.process_coords) == [0, 0, 0, 0]
    assert p.test_vector_rng == 0


def test_mpi_glue_library_exports_its_symbol():
    path = os.path.join(os.path.dirname(dd.library_path()), "libddamg_hip_mpi.so")
    if not os.path.exists(path):
        pytest.skip("libddamg_hip_mpi.so not built (no MPI in this image)")
    hdr = open(os.path.join(os.path.dirname(dd.library_path()), "..", "include", "ddamg_hip_mpi.h")).read()
    assert "ddamg_hip_comm_init_mpi" in hdr
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    assert " T ddamg_hip_comm_init_mpi" in out
