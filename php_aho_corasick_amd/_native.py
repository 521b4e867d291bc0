"""ctypes binding of libacmi.so (the C ABI declared in include/acmi.h).

There is deliberately no fallback: if the shared library is missing or does not export a
symbol, importing this module raises.  Matching additionally needs a gfx950 GPU; without
one the library itself returns ACMI_E_NO_DEVICE and the callers raise ``AcmiError``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ACMI_LIBRARY selects another build of the same ABI (A/B kernel comparisons inside one process launch)
LIB_PATH = os.environ.get("ACMI_LIBRARY") or os.path.join(HERE, "libacmi.so")

OK, E_INVALID, E_NOT_INIT, E_FINALIZED, E_ARG, E_NOMEM, E_DEVICE, E_NO_DEVICE, E_TOO_LARGE, E_NO_REPLACEMENT, E_PARTIAL = \
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10
ID_DEFAULT, ID_NUMBER, ID_STRING = 0, 1, 2


class AcmiError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        msg = lib.acmi_strerror(status).decode() if "lib" in globals() else str(status)
        super().__init__(f"acmi error {status}: {msg}" + (f" ({detail})" if detail else ""))


class Pattern(C.Structure):
    _fields_ = [("value", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("id_type", C.c_uint32),
                ("id_number", C.c_int64), ("user", C.c_uint64)]


class Record(C.Structure):
    _fields_ = [("hay", C.c_uint32), ("pos", C.c_uint32), ("start", C.c_uint32), ("pattern", C.c_uint32)]


class Tables(C.Structure):
    _fields_ = [("n_states", C.c_uint32), ("n_columns", C.c_uint32), ("entry_bytes", C.c_uint32),
                ("first_hot_final", C.c_uint32), ("trap", C.c_uint32), ("first_cold_final", C.c_uint32),
                ("class_mode", C.c_uint32), ("swar_shift", C.c_uint32), ("swar_other", C.c_uint32),
                ("max_pattern_length", C.c_uint32),
                ("table", C.c_void_p), ("lds_image", C.c_void_p), ("class_of_byte", C.POINTER(C.c_uint8)),
                ("own_pattern", C.POINTER(C.c_uint32)), ("dict_link", C.POINTER(C.c_uint32)),
                ("out_count", C.POINTER(C.c_uint32)),
                ("anchor_q", C.c_uint32), ("anchor_mul", C.c_uint32), ("min_pattern_length", C.c_uint32),
                ("anchor_keys", C.c_uint32), ("anchor_table", C.POINTER(C.c_uint16)),
                ("anchor_mode", C.c_uint32), ("anchor_bucket_bits", C.c_uint32), ("anchor_bucket_shift", C.c_uint32),
                ("anchor_bitmap", C.POINTER(C.c_uint32)),
                ("anchor_choice", C.c_uint32), ("anchor_offsets", C.POINTER(C.c_uint8)),
                ("n_flat_outputs", C.c_uint32), ("compact", C.c_uint32), ("dense_nf_end", C.c_uint32),
                ("dense_f_end", C.c_uint32), ("n_dense_rows", C.c_uint32), ("n_cold_edges", C.c_uint32),
                ("cold_rec", C.POINTER(C.c_uint32)), ("cold_edges", C.POINTER(C.c_uint32)),
                ("chain_lo", C.c_uint32), ("chain_byte", C.POINTER(C.c_uint8)), ("chain_run", C.POINTER(C.c_uint8)),
                ("out_off", C.POINTER(C.c_uint32)), ("out_list", C.POINTER(C.c_uint32)), ("anchor_slots", C.c_uint32), ("anchor_dense", C.c_uint32),
                ("chain_image", C.POINTER(C.c_uint8)), ("chain_bytes", C.c_uint32), ("chain_dense", C.c_uint32), ("chain_index", C.c_uint32),
                ("chain_rec_off", C.c_uint32), ("chain_cnt_off", C.c_uint32), ("chain_cnt_n", C.c_uint32),
                ("chain_orig", C.POINTER(C.c_uint32)), ("chain_head", C.POINTER(C.c_uint32)),
                ("win_table", C.POINTER(C.c_uint16)), ("win_state", C.POINTER(C.c_uint32)), ("win_hash", C.POINTER(C.c_uint32)),
                ("win_hash_bits", C.c_uint32), ("win_list_bytes", C.c_uint32), ("shallow_hot", C.c_uint32), ("shallow_cold_end", C.c_uint32)]


Handle = C.c_void_p

# every symbol include/acmi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "acmi_create": (C.c_int, [C.c_int, C.POINTER(Handle)]),
    "acmi_add_patterns": (C.c_int, [Handle, C.POINTER(Pattern), C.c_size_t, C.POINTER(C.c_size_t)]),
    "acmi_finalize": (C.c_int, [Handle]),
    "acmi_is_valid": (C.c_int, [Handle]),
    "acmi_destroy": (C.c_int, [Handle]),
    "acmi_match": (C.c_int, [Handle, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(Record)),
                             C.POINTER(C.c_size_t)]),
    "acmi_stream_open": (C.c_int, [Handle, C.POINTER(C.c_void_p)]),
    "acmi_stream_feed": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(Record)),
                                   C.POINTER(C.c_size_t)]),
    "acmi_stream_position": (C.c_uint64, [C.c_void_p]),
    "acmi_stream_close": (None, [C.c_void_p]),
    "acmi_match_batch": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                   C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(Record)),
                                   C.POINTER(C.c_size_t)]),
    "acmi_match_batch_device": (C.c_int, [Handle, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int,
                                          C.POINTER(C.c_uint64)]),
    "acmi_match_batch_device_begin": (C.c_int, [Handle, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]),
    "acmi_match_batch_device_begin_uniform": (C.c_int, [Handle, C.c_void_p, C.c_uint64, C.c_size_t, C.c_int]),
    "acmi_match_batch_device_end": (C.c_int, [Handle, C.POINTER(C.c_uint64)]),
    "acmi_result_device": (C.c_int, [Handle, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_uint64)]),
    "acmi_result_to_host": (C.c_int, [Handle, C.c_void_p, C.c_void_p]),
    "acmi_result_to_device": (C.c_int, [Handle, C.c_void_p, C.c_void_p]),
    "acmi_set_replacement": (C.c_int, [Handle, C.c_uint32, C.c_char_p, C.c_uint32]),
    "acmi_replace_batch": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                     C.POINTER(C.POINTER(C.c_uint64))]),
    "acmi_replace_batch_device": (C.c_int, [Handle, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "acmi_comm_get_id": (C.c_int, [C.c_void_p]),
    "acmi_comm_create": (C.c_int, [Handle, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "acmi_comm_destroy": (C.c_int, [C.c_void_p]),
    "acmi_comm_allgather_result": (C.c_int, [C.c_void_p, Handle, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "acmi_comm_gather_result": (C.c_int, [C.c_void_p, Handle, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "acmi_comm_result_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "acmi_comm_create_local": (C.c_int, [C.POINTER(Handle), C.c_int, C.POINTER(C.c_void_p)]),
    "acmi_comm_get_stat": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "acmi_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "acmi_pattern_count": (C.c_size_t, [Handle]),
    "acmi_pattern_info": (C.c_int, [Handle, C.c_uint32, C.POINTER(Pattern)]),
    "acmi_set_option": (C.c_int, [Handle, C.c_char_p, C.c_int64]),
    "acmi_get_stat": (C.c_int, [Handle, C.c_char_p, C.POINTER(C.c_double)]),
    "acmi_set_stream": (C.c_int, [Handle, C.c_void_p]),
    "acmi_debug_tables": (C.c_int, [Handle, C.POINTER(Tables)]),
    "acmi_debug_stream_read": (C.c_int, [Handle, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "acmi_strerror": (C.c_char_p, [C.c_int]),
    "acmi_last_error": (C.c_char_p, [Handle]),
    "acmi_version": (C.c_char_p, []),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -m php_aho_corasick_amd.build` "
                      "(hipcc, gfx950). There is no pure-Python or CPU fallback.")



def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  torch wheels bundle their own libamdhip64.so (same soname as
    the system one); if libacmi.so pulled in /opt/rocm's copy first, torch would later load a second
    runtime and see no GPUs.  Loading torch's copy first makes libacmi.so's DT_NEEDED
    libamdhip64.so.7 resolve to it by soname.  Without torch the system runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)  # AttributeError here == the .so does not export a declared symbol
    _fn.restype, _fn.argtypes = _res, _args


def check(status: int, handle=None) -> int:
    if status < 0:
        detail = lib.acmi_last_error(handle).decode() if handle else ""
        raise AcmiError(status, detail)
    return status
