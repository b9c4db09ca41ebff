"""ctypes mirror of the reference's library interface include/dd_alpha_amg.h (served by libddamg_hip.so).

This is the binding a Python host code would write against the reference's libdd_alpha_amg: same struct
layout (dd_alpha_amg_par passed by value, callbacks returning offsets in doubles), same function names."""
import ctypes
from .api import load_library

MAX_MG_LEVELS = 4
STRINGLENGTH = 500


class AmgParameters(ctypes.Structure):
    """struct dd_alpha_amg_parameters (include/dd_alpha_amg_parameters.h); lattices in X,Y,Z,T order"""
    _fields_ = [
        ("number_of_levels", ctypes.c_int),
        ("global_lattice", (ctypes.c_int * 4) * MAX_MG_LEVELS),
        ("local_lattice", (ctypes.c_int * 4) * MAX_MG_LEVELS),
        ("block_lattice", (ctypes.c_int * 4) * MAX_MG_LEVELS),
        ("mg_basis_vectors", ctypes.c_int * MAX_MG_LEVELS),
        ("setup_iterations", ctypes.c_int * MAX_MG_LEVELS),
        ("discard_setup_after", ctypes.c_int),
        ("update_setup_iterations", ctypes.c_int * MAX_MG_LEVELS),
        ("update_setup_after", ctypes.c_int),
        ("post_smooth_iterations", ctypes.c_int * MAX_MG_LEVELS),
        ("post_smooth_block_iterations", ctypes.c_int * MAX_MG_LEVELS),
        ("coarse_grid_iterations", ctypes.c_int),
        ("coarse_grid_maximum_number_of_restarts", ctypes.c_int),
        ("coarse_grid_tolerance", ctypes.c_double),
        ("solver_mass", ctypes.c_double),
        ("setup_mass", ctypes.c_double),
        ("c_sw", ctypes.c_double),
    ]


CONF_INDEX_FCT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)
VECTOR_INDEX_FCT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)
GLOBAL_TIME_FCT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int)


class Par(ctypes.Structure):
    """dd_alpha_amg_par (include/dd_alpha_amg.h)"""
    _fields_ = [
        ("param_file_path", ctypes.c_char * STRINGLENGTH),
        ("conf_index_fct", CONF_INDEX_FCT),
        ("vector_index_fct", VECTOR_INDEX_FCT),
        ("global_time", GLOBAL_TIME_FCT),
        ("bc", ctypes.c_int),
        ("m0", ctypes.c_double),
        ("csw", ctypes.c_double),
        ("setup_m0", ctypes.c_double),
        ("amg_params", AmgParameters),
    ]


SYMBOLS = ["dd_alpha_amg_init", "dd_alpha_amg_init_external_threading", "dd_alpha_amg_get_gauge_pointer",
           "dd_alpha_amg_get_clover_pointer", "dd_alpha_amg_fields_updated", "dd_alpha_amg_set_conf",
           "dd_alpha_amg_update_parameters", "dd_alpha_amg_setup", "dd_alpha_amg_setup_external_threading",
           "dd_alpha_amg_setup_update", "dd_alpha_amg_setup_update_external_threading", "dd_alpha_amg_wilson_solve",
           "dd_alpha_amg_preconditioner", "dd_alpha_amg_preconditioner_external_threading", "dd_alpha_amg_free"]


def bind():
    lib = load_library()
    dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
    lib.dd_alpha_amg_init.argtypes = [Par]; lib.dd_alpha_amg_init.restype = None
    lib.dd_alpha_amg_init_external_threading.argtypes = [Par, ctypes.c_int, ctypes.c_int]
    lib.dd_alpha_amg_init_external_threading.restype = None
    lib.dd_alpha_amg_get_gauge_pointer.restype = dp
    lib.dd_alpha_amg_get_clover_pointer.restype = dp
    lib.dd_alpha_amg_fields_updated.restype = None
    lib.dd_alpha_amg_set_conf.argtypes = [dp]; lib.dd_alpha_amg_set_conf.restype = ctypes.c_double
    lib.dd_alpha_amg_update_parameters.argtypes = [ctypes.POINTER(AmgParameters)]; lib.dd_alpha_amg_update_parameters.restype = None
    lib.dd_alpha_amg_setup.argtypes = [ctypes.c_int, ip]; lib.dd_alpha_amg_setup.restype = None
    lib.dd_alpha_amg_setup_update.argtypes = [ctypes.c_int, ip]; lib.dd_alpha_amg_setup_update.restype = None
    lib.dd_alpha_amg_wilson_solve.argtypes = [dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ip]
    lib.dd_alpha_amg_wilson_solve.restype = ctypes.c_double
    lib.dd_alpha_amg_preconditioner.argtypes = [dp, dp, ctypes.c_double, ctypes.c_double, ip]
    lib.dd_alpha_amg_preconditioner.restype = None
    lib.dd_alpha_amg_free.restype = None
    return lib
