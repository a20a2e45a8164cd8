"""ctypes binding of libshine_hip.so (the C ABI declared in include/shine_hip.h).

The library is the product path.  If it is missing or fails to load this module raises — there is no
CPU or PyTorch fallback anywhere in shine_mapping_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libshine_hip.so")
# the product's objects + the training build of the lane-per-point reference step (kernel_variant 1): loaded by tests /
# tools only, never by the product path
CHECK_LIB_PATH = os.path.join(_HERE, "lib", "libshine_check.so")

MAX_LEVELS = 8
FEATURE_DIM = 8
HIDDEN_DIM = 32
MLP_PARAMS = 1377


class NextDraw(C.Structure):  # include/shine_hip.h shine_next_draw
    _fields_ = [("pool_size", C.c_int64), ("n", C.c_int64), ("seed", C.c_uint64), ("stream_state", C.c_void_p),
                ("idx_out", C.c_void_p), ("surf_bits", C.c_void_p), ("surf_parts", C.c_void_p), ("workspace", C.c_void_p)]


class RegRider(C.Structure):
    """struct shine_reg_rider (include/shine_hip.h)."""

    _fields_ = [("last", C.c_void_p * 8), ("imp", C.c_void_p * 8), ("stamp", C.c_void_p * 8), ("epoch", C.c_uint32),
                ("acc", C.c_void_p)]


class DrawRider(C.Structure):
    """struct shine_draw_rider (include/shine_hip.h)."""

    _fields_ = [("pool_size", C.c_int64), ("n", C.c_int64), ("seed", C.c_uint64), ("state", C.c_void_p), ("parity", C.c_int32),
                ("block_sum", C.c_void_p * 2), ("idx_out", C.c_void_p), ("surf_bits", C.c_void_p), ("surf_parts", C.c_void_p * 2),
                ("zero_ptr", C.c_void_p), ("zero_bytes", C.c_int64)]


class StepConfig(C.Structure):
    """struct shine_step_config (include/shine_hip.h)."""

    _fields_ = [
        ("n_levels", C.c_int32),
        ("max_level", C.c_int32),
        ("poly_int_on", C.c_int32),
        ("reduction_sum", C.c_int32),
        ("eikonal_on", C.c_int32),
        ("decoder_grad_on", C.c_int32),
        ("sorted_input", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("sigma", C.c_float),
        ("weight_e", C.c_float),
        ("inv_n", C.c_double),
        ("n_global", C.c_int64),
        ("sort_origin", C.c_int32 * 3),
        ("sort_bits", C.c_int32 * 3),
        ("loss_weight_on", C.c_int32),
        ("adam_state", C.c_void_p),   # iteration hooks (include/shine_hip.h): optional device pointers
        ("adam_beta1", C.c_float),
        ("adam_beta2", C.c_float),
        ("zero_f64", C.c_void_p),
        ("n_surf_parts", C.c_int32),
        ("next_draw", C.POINTER(NextDraw)),
        ("defer_reduce", C.c_int32),
        ("reg_rider", C.POINTER(RegRider)),
        ("draw_rider", C.POINTER(DrawRider)),
    ]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "shine_version": (C.c_int, []),
    "shine_draw_rider_prime": (C.c_int, [C.POINTER(DrawRider), C.c_uint64, _P]),
    "shine_error_string": (C.c_char_p, [C.c_int]),
    "shine_tables_create": (C.c_int, [C.c_int32, C.POINTER(_P)]),
    "shine_tables_destroy": (C.c_int, [_P]),
    "shine_tables_insert": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, _P]),
    "shine_tables_stats": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "shine_tables_retired_bytes": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "shine_tables_trim": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "shine_query_indices": (C.c_int, [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(_P), _P]),
    "shine_forward": (
        C.c_int,
        [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), _P, _P,
         C.POINTER(_P), _P, _P],
    ),
    "shine_tables_grow": (C.c_int, [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), _P]),
    "shine_tables_grow_fetch": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "shine_tables_insert_corners": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, _P]),
    "shine_tables_corner_count": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64)]),
    "shine_tables_rank_nodes": (C.c_int, [_P, C.POINTER(C.c_int64), _P]),
    "shine_query_points": (
        C.c_int,
        [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), C.c_int32,
         C.c_int32, _P, _P, _P],
    ),
    "shine_train_step": (
        C.c_int,
        [_P, C.POINTER(StepConfig), _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64),
         C.POINTER(_P), _P, _P, C.POINTER(_P), C.POINTER(_P), _P, C.POINTER(_P), _P, C.c_size_t, _P],
    ),
    "shine_interp_sdf_backward": (
        C.c_int,
        [_P, C.POINTER(StepConfig), _P, _P, _P, _P, _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P),
         C.POINTER(_P), C.POINTER(_P), _P, C.c_size_t, _P],
    ),
    "shine_regularize": (
        C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                  C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_float, _P, C.c_int32, C.c_int32, _P]),
    "shine_importance_accumulate": (C.c_int, [_P, _P, C.c_int64, _P]),
    "shine_append_rows": (C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.c_float, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P]),
    "shine_tables_grow_fetch_all": (C.c_int, [_P, _P, C.c_int64, _P]),
    "shine_importance_chunks": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, _P, C.POINTER(C.c_int64), C.c_int32, _P,
                                          C.POINTER(C.c_size_t), _P]),
    "shine_importance_sweep_sizes": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int64, C.c_size_t,
                                               C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "shine_importance_sweep": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, C.c_size_t,
                                         _P, C.c_size_t, _P]),
    "shine_adam_step": (
        C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64),
                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int64, C.c_int32,
                  C.POINTER(_P), _P]),
    "shine_adam_step_dev": (
        C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64),
                  _P, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, _P, C.c_int32, C.POINTER(_P), _P]),
    "shine_finish_iteration": (
        C.c_int, [C.POINTER(StepConfig), C.c_int64, _P, _P, _P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                  C.POINTER(C.c_int32), C.c_float, _P, C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                  C.POINTER(C.c_int64), _P, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, _P,
                  C.POINTER(NextDraw), C.c_int32, _P]),
    "shine_sample_sorted_finish": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, _P, _P, _P, C.c_size_t, _P,
                                             _P, _P, C.c_size_t, _P]),
    "shine_sample_sorted_dev": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, _P, _P, _P, C.c_size_t, _P, _P, _P,
                                          C.POINTER(C.c_size_t), _P]),
    "shine_sample_sorted_slice": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, _P,
                                            C.c_size_t, _P, _P, _P, C.POINTER(C.c_size_t), _P]),
    "shine_touched_index": (C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_P), _P, _P,
                                      C.POINTER(C.c_size_t), _P]),
    "shine_touched_pack": (C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     _P, _P]),
    "shine_rows_message_words": (C.c_int64, [C.c_int64, C.c_int64]),
    "shine_rows_pack": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.c_int32, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P,
                                  C.POINTER(C.c_size_t), _P]),
    "shine_rows_unpack_add": (C.c_int, [_P, C.c_int32, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P]),
    "shine_touched_unpack": (C.c_int, [C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.POINTER(_P), _P, _P]),
    "shine_train_step_workspace_bytes": (C.c_size_t, [C.POINTER(StepConfig), C.c_int64]),
    "shine_train_step_info": (C.c_int, [C.POINTER(StepConfig), C.c_int64, C.POINTER(C.c_int64)]),
    "shine_train_step_regime": (C.c_int, [C.POINTER(StepConfig), C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int32)]),
    "shine_mark_touched": (C.c_int, [_P, C.POINTER(StepConfig), _P, _P, _P, C.c_int64, C.POINTER(C.c_int64),
                                     C.POINTER(_P), _P]),
    "shine_morton_sort": (C.c_int, [C.POINTER(StepConfig), _P, C.c_int64, _P, _P, C.POINTER(C.c_size_t), _P]),
    "shine_selftest_mfma16": (C.c_int, [_P, _P, _P, _P]),
    "shine_selftest_permlane": (C.c_int, [_P, _P, _P, _P, _P]),
    "shine_debug_set_profile_buffer": (None, [_P]),
    "shine_sample_sorted": (C.c_int, [C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, C.c_size_t, _P, _P,
                                      _P, C.POINTER(C.c_size_t), _P]),
    "shine_tables_set_ranks": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, C.c_int64, _P]),
    "shine_plan_batch": (C.c_int, [_P, C.POINTER(StepConfig), _P, C.c_int64, _P, _P, _P, C.c_size_t, _P,
                                   C.POINTER(C.c_size_t), _P]),
    "shine_mlp_forward": (C.c_int, [_P, C.c_int64, C.POINTER(_P), _P, _P]),
    "shine_mlp_backward": (C.c_int, [_P, _P, C.c_int64, C.POINTER(_P), _P, C.POINTER(_P), _P]),
    "shine_mlp_backward_backward": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(_P), _P, C.POINTER(_P), _P]),
    "shine_interp_backward": (
        C.c_int, [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64), _P, _P,
                  C.POINTER(_P), _P]),
    "shine_interp_backward_backward": (
        C.c_int, [_P, C.POINTER(StepConfig), _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P,
                  C.POINTER(_P), _P]),
    # the iteration graph: set_step / set_finish take the arguments of shine_train_step / shine_finish_iteration minus the stream
    "shine_bce_loss": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_int32, _P, _P, _P]),
    "shine_iter_graph_create": (C.c_int, [C.c_int32, C.POINTER(_P)]),
    "shine_iter_graph_destroy": (C.c_int, [_P]),
    "shine_iter_graph_commit": (C.c_int, [_P]),
    "shine_iter_graph_launch": (C.c_int, [_P, C.c_int32, _P]),
    "shine_iter_graph_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None
_check = None


class ShineHipError(RuntimeError):
    pass


def exported_symbols():
    """Every symbol include/shine_hip.h declares (checked by tests/test_abi.py against the header text)."""
    return sorted(_SIGNATURES)


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ShineHipError(
            "libshine_hip.so not found at %s — build it with `python -m shine_mapping_amd.build` "
            "(there is no fallback path)" % LIB_PATH
        )
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(h, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


_SIGNATURES["shine_iter_graph_operand_image_floats"] = (C.c_int, [])
_SIGNATURES["shine_iter_graph_set_operand_image"] = (C.c_int, [_P, _P])
_SIGNATURES["shine_iter_graph_set_step"] = (C.c_int, [_P] + _SIGNATURES["shine_train_step"][1][:-1])
_SIGNATURES["shine_iter_graph_set_finish"] = (C.c_int, [_P] + _SIGNATURES["shine_finish_iteration"][1][:-1])


def check_lib():
    """The CHECK library (tests / tools only): same ABI, plus the kernel behind StepOptions.kernel_variant 1.  Table
    handles are plain process memory with one layout in both libraries, so a handle made by one works in the other."""
    global _check
    if _check is not None:
        return _check
    if not os.path.isfile(CHECK_LIB_PATH):
        raise ShineHipError("libshine_check.so not found at %s — `python -m shine_mapping_amd.build` builds it next to the "
                            "product library" % CHECK_LIB_PATH)
    h = C.CDLL(CHECK_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    _check = h
    return h


def check(code: int, what: str = "", library=None):
    """`library`: the handle whose call returned `code` (default: the product library) — the message lives in THAT library's
    thread-local buffer."""
    if code != 0:
        msg = (library if library is not None else lib()).shine_error_string(code)
        raise ShineHipError("%s failed (%d): %s" % (what or "libshine_hip call", code, (msg or b"").decode()))


def ptr_array(ptrs):
    """host array of device pointers (None -> NULL)."""
    arr = (_P * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p if p else None
    return arr


def i64_array(vals):
    arr = (C.c_int64 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


def device_constants(values, dtype, device):
    """A small device tensor holding `values`, WITHOUT a host-synchronous copy: torch.tensor(values, device=...) goes through
    pageable memory, and that copy waits for everything queued on the stream in front of it — a loop that runs the host ahead of
    the device (incremental mapping re-creates its optimiser every frame) would stall there once per frame.  Equal values are a
    fill launch; anything else goes through pinned memory as a non-blocking copy."""
    import torch

    vals = list(values)
    if not vals:
        return torch.empty(0, dtype=dtype, device=device)
    if all(v == vals[0] for v in vals):
        return torch.full((len(vals),), vals[0], dtype=dtype, device=device)
    if torch.device(device).type != "cuda":
        return torch.tensor(vals, dtype=dtype, device=device)
    return torch.tensor(vals, dtype=dtype).pin_memory().to(device, non_blocking=True)


def current_stream_handle():
    """Raw hipStream_t of torch's current stream on the current device.  The private fast path costs ~1 us,
    torch.cuda.current_stream().cuda_stream ~8 us — several of those per small-batch iteration add up."""
    import torch

    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:  # pragma: no cover
        return torch.cuda.current_stream().cuda_stream
