"""ctypes loader for libvslam_hip.so (the C ABI declared in include/vslam_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvslam_hip.so")

VS_OK = 0
STATUS = {0: "VS_OK", -1: "VS_ERR_INVALID", -2: "VS_ERR_HIP", -3: "VS_ERR_NOMEM",
          -4: "VS_ERR_STATE", -5: "VS_ERR_NUMERIC", -6: "VS_ERR_UNSUPPORTED"}

# every symbol include/vslam_hip.h declares
SYMBOLS = [
    "vs_last_error_string", "vs_version", "vs_lm_default_options",
    "vs_ba_create", "vs_ba_destroy", "vs_ba_set_stream", "vs_ba_set_cameras", "vs_ba_set_points",
    "vs_ba_set_observations", "vs_ba_set_observations_sharded", "vs_ba_set_fixed",
    "vs_ba_set_huber_first", "vs_ba_evaluate", "vs_ba_linearize", "vs_ba_block_count",
    "vs_ba_block_index", "vs_ba_get_reduced_system", "vs_ba_get_landmark_blocks",
    "vs_ba_reduce_buffer", "vs_ba_lm_step", "vs_ba_solve", "vs_ba_set_allreduce",
    "vs_ba_get_cameras", "vs_ba_get_points", "vs_ba_profile_linearize", "vs_matcher_profile_run", "vs_debug_mfma_rate",
    "vs_matcher_create", "vs_matcher_destroy", "vs_matcher_set_stream", "vs_matcher_set_pair", "vs_matcher_set_pairs",
    "vs_matcher_run", "vs_matcher_run_sharded", "vs_matcher_set_comm", "vs_matcher_set_min_allreduce",
    "vs_matcher_get_matches", "vs_matcher_get_nn", "vs_matcher_get_rescans",
    "vs_match_two_way", "vs_match_stereo",
    "vs_reproject_points", "vs_search_projection", "vs_remove_outliers", "vs_compute_points3d",
    "vs_triangulate_tracks",
    "vs_match_two_way_noalloc",
    "vs_comm_unique_id", "vs_comm_create", "vs_comm_destroy", "vs_comm_allreduce_sum", "vs_ba_set_comm",
    "vs_ba_set_allgather", "vs_ba_exchange_reduced", "vs_ba_exchange_size", "vs_comm_allgather",
]


class VslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, code), msg))
        self.code = code


class LMOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("max_consecutive_invalid_steps", C.c_int32)]


class LMSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32),
                ("termination", C.c_int32), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("final_radius", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

_lib = None


def load():
    """Returns the loaded library; raises if it has not been built (run `make`)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch's ROCm wheel carries private copies of the HIP and HSA runtimes.  A process that
    # loads the system runtime first (through this library) and PyTorch afterwards ends up with two
    # HSA runtimes and PyTorch then sees no GPU; loaded in the other order this library binds to
    # the copy already in the process.  So when PyTorch is installed it is imported first.
    if os.environ.get("VSLAM_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = os.environ.get("VSLAM_HIP_LIB", LIB_PATH)     # developer knob: an alternative build
    if not os.path.exists(path):
        raise ImportError("libvslam_hip.so not found at %s: build it with `make` (hipcc, gfx950); "
                          "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.vs_last_error_string.restype = C.c_char_p
    lib.vs_version.restype = C.c_char_p
    lib.vs_lm_default_options.restype = None
    _lib = lib
    return lib


def check(code):
    if code != VS_OK:
        raise VslamError(code, load().vs_last_error_string().decode("utf-8", "replace"))


def ptr(a, ctype):
    """ctypes pointer to a numpy array's data, an int device address, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.cast(C.c_void_p(a), C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))
