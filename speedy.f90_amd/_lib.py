"""ctypes loader for libspdy.so (the C-ABI declared in include/spdy.h).

There is no Python/NumPy fallback: if the HIP library is missing or no device is usable the
caller gets an exception, never a silently slower path.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPDY_LIB") or os.path.join(HERE, "libspdy.so")   # SPDY_LIB: an alternative build (experiments)

c_void_p, c_int, c_double, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p

# name -> argtypes (restype is int everywhere except spdy_last_error)
SIGNATURES = {
    "spdy_plan_create": [c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)],
    "spdy_plan_destroy": [c_void_p],
    "spdy_plan_set_stream": [c_void_p, c_void_p],
    "spdy_plan_synchronize": [c_void_p],
    "spdy_dev_alloc": [c_void_p, ctypes.c_size_t, ctypes.POINTER(c_void_p)],
    "spdy_dev_free": [c_void_p, c_void_p],
    "spdy_dev_upload": [c_void_p, c_void_p, c_void_p, ctypes.c_size_t],
    "spdy_dev_download": [c_void_p, c_void_p, c_void_p, ctypes.c_size_t],
    "spdy_plan_dims": [c_void_p, ctypes.POINTER(c_int)],
    "spdy_wave_placement": [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "spdy_plan_set_profiling": [c_void_p, c_int],
    "spdy_plan_set_fused": [c_void_p, c_int],
    "spdy_plan_set_option": [c_void_p, c_char_p, c_int],
    "spdy_plan_get_profile": [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int)],
    "spdy_get_table": [c_void_p, c_char_p, c_void_p, c_int],
    "spdy_spec_to_grid": [c_void_p, c_void_p, c_int, c_void_p],
    "spdy_grid_to_spec": [c_void_p, c_void_p, c_void_p],
    "spdy_spec_to_grid_batch": [c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "spdy_grid_to_spec_batch": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_spec_to_grid_dev": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "spdy_grid_to_spec_dev": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_legendre_inv": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_legendre_dir": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_fourier_inv": [c_void_p, c_int, c_void_p, c_int, c_void_p],
    "spdy_fourier_dir": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_laplacian": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_inverse_laplacian": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_trunct": [c_void_p, c_int, c_void_p],
    "spdy_grad": [c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "spdy_vds": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_uvspec": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_vdspec": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_laplacian_dev": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_inverse_laplacian_dev": [c_void_p, c_int, c_void_p, c_void_p],
    "spdy_trunct_dev": [c_void_p, c_int, c_void_p],
    "spdy_grad_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "spdy_vds_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_uvspec_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_vdspec_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_hdiff": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_hdiff_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_implicit_init": [c_void_p, c_double],
    "spdy_implicit_terms": [c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_implicit_terms_dev": [c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_device_table": [c_void_p, c_char_p, ctypes.POINTER(c_void_p)],
    "spdy_uvspec_to_grid_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_grad_to_grid_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_uvspec_to_grid": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_grad_to_grid": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_hdiff_multi_dev": [c_void_p, c_int, c_void_p],
    "spdy_direct_batch_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "spdy_inverse_batch_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "spdy_direct_batch_spectral_step_dev": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_double,
                                            c_double, c_double, c_void_p],
    "spdy_inverse_batch_segs_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_inverse_batch_grad_dev": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_int],
    "spdy_plan_set_sigma": [c_void_p, c_void_p],
    "spdy_geopotential": [c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_geopotential_dev": [c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_spectral_tendencies_dev": [c_void_p] * 9,
    "spdy_hdiff_step_dev": [c_void_p] * 7 + [c_double] + [c_void_p] * 4,
    "spdy_step_fields_dev": [c_void_p, c_int, c_void_p, c_int, c_double, c_double, c_double],
    "spdy_step_field": [c_void_p, c_int, c_int, c_double, c_double, c_double, c_void_p, c_void_p],
    "spdy_comm_unique_id": [c_void_p],
    "spdy_comm_create": [c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)],
    "spdy_comm_destroy": [c_void_p],
    "spdy_comm_level_range": [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "spdy_allgather_levels_dev": [c_void_p, c_int, c_int, c_void_p],
    "spdy_implicit_terms_sharded_dev": [c_void_p, c_void_p, c_void_p, c_void_p],
    "spdy_comm_group_create": [c_int, ctypes.POINTER(c_void_p)],
    "spdy_comm_group_destroy": [c_void_p],
    "spdy_comm_create_local": [c_void_p, c_void_p, c_int, ctypes.POINTER(c_void_p)],
    "spdy_sharded_step_workspace": [c_void_p],
    "spdy_sharded_step_dev": [c_void_p] * 9 + [c_double, c_int, c_int, c_double, c_double, c_double, c_void_p, c_void_p],
    "spdy_sharded_step_grid_dev": [c_void_p] * 6 + [c_int],
    "spdy_sharded_step_operands": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                   ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "spdy_sharded_step_spectral_dev": [c_void_p] * 9 + [c_double, c_int, c_double, c_double, c_double, c_void_p, c_void_p],
    "spdy_comm_set_option": [c_void_p, c_char_p, c_int],
    "spdy_sharded_state_gather_dev": [c_void_p] * 6,
    "spdy_sharded_gather_ranges_dev": [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int)],
    "spdy_comm_describe": [c_void_p, c_char_p, c_int],
    "spdy_sharded_step_stacks": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(c_void_p),
                                 ctypes.POINTER(ctypes.c_size_t)],
    "spdy_grid_tendencies_dev": [c_void_p] * 12,
    "spdy_tendency_combine_dev": [c_void_p, c_void_p, c_void_p],
    "spdy_spectral_step_dev": [c_void_p] * 12 + [c_double, c_int, c_double, c_double, c_double, c_void_p],
    "spdy_output_workspace": [c_void_p],
    "spdy_output_batch_dev": [c_void_p] * 13,
    "spdy_graph_begin": [c_void_p],
    "spdy_graph_end": [c_void_p, ctypes.POINTER(c_void_p)],
    "spdy_graph_launch": [c_void_p],
    "spdy_graph_destroy": [c_void_p],
    "spdy_graph_num_nodes": [c_void_p, ctypes.POINTER(c_int)],
}


class SpdyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("spdy error %d: %s" % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile libspdy.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=out)


_lib = None


def load():
    """Load libspdy.so and bind every symbol include/spdy.h declares.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the transform path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    lib.spdy_last_error.argtypes = []
    lib.spdy_last_error.restype = c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise SpdyError(rc, load().spdy_last_error().decode(errors="replace"))
    return rc
