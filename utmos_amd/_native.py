"""ctypes binding of libutmos_hip.so (C ABI: include/utmos_hip.h).  No torch, no cffi."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libutmos_hip.so")

UTM_OK = 0
AF_NONE, AF_F32, AF_F64 = 0, 1, 2
FLAG_PROFILE_EVENTS = 1
FLAG_AF_SEQUENTIAL = 2
FLAG_DECREMENTAL = 4
UNIQUE_ID_BYTES = 128


class NativeError(RuntimeError):
    """A libutmos_hip call failed (code, message from utm_last_error)."""

    def __init__(self, code, message):
        super().__init__(f"libutmos_hip error {code}: {message}")
        self.code = code


class Record(ctypes.Structure):
    _fields_ = [("score", ctypes.c_double), ("idx", ctypes.c_int64), ("new_count", ctypes.c_int64),
                ("pad", ctypes.c_int64 * 5)]


class Stats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int64), ("tot_captured", ctypes.c_int64),
                ("score_launches", ctypes.c_int64), ("score_ms", ctypes.c_double),
                ("loop_ms", ctypes.c_double), ("algo_bytes", ctypes.c_int64),
                ("af_mode", ctypes.c_int32), ("af_fixed_point", ctypes.c_int32),
                ("af_q", ctypes.c_int32), ("n_chunks", ctypes.c_int32),
                ("decr_iterations", ctypes.c_int64), ("brute_force_bytes", ctypes.c_int64),
                ("p2p_replica_bytes", ctypes.c_int64), ("decr_interleaved_bytes", ctypes.c_int64),
                ("exchange", ctypes.c_int32), ("rccl_ranks", ctypes.c_int32),
                ("af_chained_iterations", ctypes.c_int64), ("af_deferred_rows", ctypes.c_int64),
                ("persist_launches", ctypes.c_int64), ("persist_iterations", ctypes.c_int64),
                ("persist_unresolved", ctypes.c_int64),
                ("af_table_passes", ctypes.c_int64)]


EXCHANGE_NAMES = {0: "none", 1: "mailboxes", 2: "rccl", 3: "caller-driven", 4: "rccl-allreduce"}


_P = ctypes.c_void_p
_U64, _U32, _I32, _I64 = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int64

# name -> argtypes, exactly the prototypes of include/utmos_hip.h (tests/test_abi.py checks the list)
PROTOTYPES = {
    "utm_abi_version": [],
    "utm_env_overrides": [ctypes.c_char_p, _U64],
    "utm_device_count": [ctypes.POINTER(ctypes.c_int)],
    "utm_device_memory": [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)],
    "utm_ctx_create": [ctypes.c_int, _U32, _U32, _U32, _U32, ctypes.POINTER(_P)],
    "utm_ctx_destroy": [_P],
    "utm_add_chunk": [_P, _U64, ctypes.POINTER(_I32)],
    "utm_upload_columns": [_P, _I32, _U32, _U32, _P, _U64],
    "utm_upload_rows_packed": [_P, _I32, _U64, _U64, _P, _U64],
    "utm_download_columns": [_P, _I32, _U32, _U32, _P, _U64],
    "utm_var_count": [_P, _P],
    "utm_synth_fill": [_P, _I32, _U64, _U64],
    "utm_synth_host": [_U64, _U64, _U64, _U32, _U32, _U32, _P, _U64, _P],
    "utm_set_sample_state": [_P, _P],
    "utm_set_weights": [_P, _P],
    "utm_set_af": [_P, _I32, ctypes.c_int, _P],
    "utm_reset": [_P],
    "utm_step": [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(ctypes.c_double)],
    "utm_run": [_P, _I64, _P, _P, _P, ctypes.POINTER(_I64)],
    "utm_peek_scores": [_P, _P, _P],
    "utm_peek_estimates": [_P, _P, _P],
    "utm_get_covered": [_P, _I32, _P],
    "utm_get_stats": [_P, ctypes.POINTER(Stats)],
    "utm_set_profile": [_P, _I32],
    "utm_stream_calibration": [_P, _I32, ctypes.POINTER(ctypes.c_double)],
    "utm_set_af_exact_scores": [_P, _I32],
    "utm_set_decremental": [_P, _I32, ctypes.c_double],
    "utm_local_best": [_P, ctypes.POINTER(Record)],
    "utm_column_words": [_P, ctypes.POINTER(_U64)],
    "utm_get_column": [_P, _I64, _P],
    "utm_apply_records": [_P, ctypes.POINTER(Record), _I32, _P, ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                          ctypes.POINTER(ctypes.c_double)],
    "utm_p2p_blob_bytes": [_P, ctypes.POINTER(_U64)],
    "utm_p2p_export": [_P, _P],
    "utm_p2p_import": [_P, _I32, _I32, _P],
    "utm_p2p_selftest": [_P, ctypes.POINTER(_I32)],
    "utm_p2p_use_mailboxes": [_P, _I32],
    "utm_comm_get_unique_id": [_P],
    "utm_comm_init": [_P, _I32, _I32, _P],
    "utm_comm_allreduce_max": [_P, ctypes.POINTER(ctypes.c_double)],
    "utm_comm_column_by_allreduce": [_P, _I32],
}

_lib = None


def lib():
    """Load the shared library once.  Missing library = hard error (there is no CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C utmos_amd/csrc` (hipcc, --offload-arch=gfx950).  utmos_amd has no CPU fallback.")
        # dmabuf IPC is the only IPC mode this host driver supports (RCCL, hipIpc mappings between shards);
        # must be in the environment before the HIP runtime initialises
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        handle = ctypes.CDLL(LIB_PATH)
        handle.utm_last_error.restype = ctypes.c_char_p
        handle.utm_last_error.argtypes = []
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = handle
    return _lib


def check(code):
    if code != UTM_OK:
        raise NativeError(code, lib().utm_last_error().decode("utf-8", "replace"))


def env_overrides():
    """The UTM_* environment knobs that are set, as {name: value} (what a context would pick up at its next reset)."""
    buf = ctypes.create_string_buffer(2048)
    check(lib().utm_env_overrides(buf, 2048))
    return dict(kv.split("=", 1) for kv in buf.value.decode().split())


def device_count():
    n = ctypes.c_int(0)
    check(lib().utm_device_count(ctypes.byref(n)))
    return n.value


def device_memory(device=0):
    """(free, total) HBM bytes of a device."""
    free, total = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib().utm_device_memory(int(device), ctypes.byref(free), ctypes.byref(total)))
    return free.value, total.value
