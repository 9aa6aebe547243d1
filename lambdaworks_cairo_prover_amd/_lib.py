"""ctypes loader for libstark252_hip.so (the C ABI declared in include/stark252_hip.h).

The HIP library is the product path: there is NO CPU fallback.  Loading fails loudly when the shared object is
missing, and every device entry point returns SP_E_NO_DEVICE when no gfx950 GPU is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstark252_hip.so")

SP_OK = 0
SP_E_INVALID_ARG = -1
SP_E_NO_DEVICE = -2
SP_E_HIP = -3
SP_E_ALLOC = -4
SP_E_STATE = -5
SP_E_ZERO_INVERSE = -6
SP_E_UNSUPPORTED = -7
SP_E_PROGRAM = -8

SP_ABI_VERSION = 5   # include/stark252_hip.h

SP_FE_MONT_LIMBS = 0
SP_FE_CANON_BE = 1


class SpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"stark252_hip error {code}: {msg}")
        self.code = code


class ProofOptionsC(ctypes.Structure):
    _fields_ = [("blowup_factor", ctypes.c_uint8), ("fri_number_of_queries", ctypes.c_uint64),
                ("coset_offset", ctypes.c_uint64), ("grinding_factor", ctypes.c_uint8)]


class ConfigC(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("fe_encoding", ctypes.c_int)]


class CairoPublicInputsC(ctypes.Structure):
    _fields_ = [("pc_init", ctypes.c_uint8 * 32), ("ap_init", ctypes.c_uint8 * 32), ("fp_init", ctypes.c_uint8 * 32),
                ("pc_final", ctypes.c_uint8 * 32), ("ap_final", ctypes.c_uint8 * 32),
                ("range_check_min", ctypes.c_uint16), ("range_check_max", ctypes.c_uint16),
                ("n_segments", ctypes.c_uint32), ("segment_types", ctypes.c_void_p), ("segment_ranges", ctypes.c_void_p),
                ("n_public_memory", ctypes.c_uint64), ("public_memory", ctypes.c_void_p), ("num_steps", ctypes.c_uint64)]


# The entry points added most recently, probed at load time beside the version number (api.py binds symbols lazily: a stale build would
# otherwise fail with AttributeError in the middle of a run instead of with "rebuild the library" here).
NEWEST_SYMBOLS = ("sp_fe_mul", "sp_comm_measure", "sp_comm_time_ms", "sp_model_shard_interpolation", "sp_proof_file_verify", "sp_proof_options_checked",
                  "sp_host_cpu_budget")

_lib = None


def load():
    """Returns the loaded library; raises if libstark252_hip.so has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C lambdaworks_cairo_prover_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback for the HIP path")
    lib = ctypes.CDLL(LIB_PATH)
    lib.sp_version.restype = ctypes.c_char_p
    lib.sp_last_error.restype = ctypes.c_char_p
    lib.sp_air_desc_size.restype = ctypes.c_uint64
    if lib.sp_abi_version() != SP_ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.sp_abi_version()}, this binding was written for {SP_ABI_VERSION}: rebuild the library")
    missing = [name for name in NEWEST_SYMBOLS if not hasattr(lib, name)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks {', '.join(missing)}: a stale build under the current ABI number - rebuild the library")
    _lib = lib
    return lib


def check(code):
    if code != SP_OK:
        raise SpError(code, load().sp_last_error().decode(errors="replace"))
