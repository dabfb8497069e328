"""ctypes binding of libwfhip.so (C ABI declared in include/wfhip.h).

PyTorch-ROCm is used for exactly three things here: device buffers
(``torch.empty(..., device="cuda")``), the current HIP stream handle, and H2D/D2H
copies.  There is NO CPU fallback: if the shared library is missing or no HIP device
is visible every numeric entry point raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import threading
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_uint64, c_void_p
from pathlib import Path

import numpy as np

import os

# WF_HIP_LIBRARY: load another build of the same C ABI (tools/sanitize.py points it at the
# host-sanitized build on the CPU box).  Not a fallback: a missing file still raises.
_LIB_PATH = Path(os.environ.get("WF_HIP_LIBRARY") or Path(__file__).resolve().parent / "csrc" / "libwfhip.so")

WF_ERR_VALUE, WF_ERR_KEY, WF_ERR_HIP, WF_ERR_DEVICE, WF_ERR_NOMEM = -1, -2, -3, -4, -5
WF_CPM_STATE_BYTES, WF_CPM_STREAM_STATE_BYTES = 16384, 20480      # include/wfhip.h (tests/test_cabi.py compares)
# wf_option (include/wfhip.h): per-context options set with wf_ctx_set_option
(WF_OPT_CPM_FORM, WF_OPT_CPM_CHUNK_CALLS, WF_OPT_DET_REPAIR, WF_OPT_DET_FINAL_VERIFY, WF_OPT_ITERATION_SERVER,
 WF_OPT_MCB_TAIL_PERMILLE, WF_OPT_PIPE_RESERVE_CUS, WF_OPT_CPM_SAMPLES_MIN_CALLS) = range(8)

# name -> (restype, argtypes); must list every function include/wfhip.h declares
# (tests/test_cabi.py parses the header and compares).
_P = c_void_p
SIGNATURES = {
    "wf_version": (c_char_p, []),
    "wf_last_error_string": (c_char_p, []),
    "wf_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "wf_ctx_destroy": (c_int, [_P]),
    "wf_ctx_retire": (c_int, [_P]),
    "wf_ctx_check": (c_int, [_P, _P]),
    "wf_ctx_set_option": (c_int, [_P, c_int, c_int64]),
    "wf_ctx_get_option": (c_int, [_P, c_int, POINTER(c_int64)]),
    "wf_ctx_forget_promises": (c_int, [_P]),
    "wf_lfsr_generate": (c_int, [_P, c_int, c_uint64, c_uint64, c_uint64, _P, c_int64, POINTER(c_uint64), _P]),
    "wf_fsm_encode": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int64, c_int64, c_int, _P, POINTER(c_int), _P]),
    "wf_symbol_map": (c_int, [_P, c_int, _P, c_int64, c_int, c_int, c_int, _P, _P]),
    "wf_fir_out_len": (c_int64, [c_int64, c_int, c_int]),
    "wf_upsample_fir_f64": (c_int, [_P, _P, c_int64, _P, c_int, _P, c_int, c_int, _P, _P]),
    "wf_phase_cexp_f64": (c_int, [_P, _P, c_int64, c_int, c_double, c_double, _P, _P, _P]),
    "wf_cpm_modulate_c128": (c_int, [_P, _P, c_int64, _P, c_int, _P, c_int, c_int, c_double, _P, _P]),
    "wf_phase_modulate_f64": (c_int, [_P, _P, c_int64, c_double, _P, _P]),
    "wf_time_axis_f64": (c_int, [_P, c_int64, c_double, _P, _P]),
    "wf_awgn_c128": (c_int, [_P, _P, c_int64, c_double, c_double, c_double, c_uint64, c_uint64, c_uint64, _P, _P]),
    "wf_box_muller32_c128": (c_int, [_P, _P, c_int64, c_double, _P, _P]),
    "wf_mf_bank_c128": (c_int, [_P, _P, c_int64, _P, c_int, c_int, c_int64, c_int, c_int64, _P, _P]),
    "wf_awgn_mf_bank_c128": (c_int, [_P, _P, c_int64, c_double, c_double, c_double, c_uint64, c_uint64, c_uint64, _P,
                                     c_int, c_int, c_int64, c_int, c_int64, _P, _P]),
    "wf_viterbi4_unmerged": (c_int, [_P, POINTER(c_int64), c_int, _P]),
    "wf_viterbi_repaired": (c_int, [_P, POINTER(c_int64), c_int, _P]),
    "wf_viterbi_cascaded": (c_int, [_P, POINTER(c_int64), c_int, _P]),
    "wf_viterbi4_detect_count": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, _P, _P]),
    "wf_viterbi4_detect": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "wf_viterbi4_detect_window": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P]),
    "wf_viterbi4_window_state_bytes": (c_int64, []),
    "wf_viterbi4_state_bytes": (c_int64, [c_int]),
    "wf_viterbi4_iteration": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P]),
    "wf_viterbi4_iteration_host": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P]),
    "wf_viterbi4_iteration_server_timing": (c_int, [_P, POINTER(ctypes.c_double)]),
    "wf_viterbi4_iteration_quiesce": (c_int, [_P]),
    "wf_link_join": (c_int, [_P, _P]),
    "wf_count_errors": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, _P]),
    "wf_link_workspace_bytes": (c_int64, [_P]),
    "wf_link_run": (c_int, [_P, _P, _P, c_int64, _P, POINTER(c_int64), _P]),
    "wf_link_stage_ms": (c_int, [_P, c_int, POINTER(ctypes.c_float)]),
    "wf_link_layout": (c_int, [_P, POINTER(c_int64)]),
    "wf_mod_tile_geometry": (c_int, [c_int, c_int, c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "wf_link_stream_workspace_bytes": (c_int64, [_P, c_int64]),
    "wf_link_stream_layout": (c_int, [_P, c_int64, c_int64, POINTER(c_int64)]),
    "wf_link_stream_chunk": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), _P]),
    "wf_link_stream_chunk_phase": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), c_int, _P]),
    "wf_link_stream_interior": (c_int, [_P, c_int64, c_int64]),
    "wf_link_stream_steady": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), _P]),
    "wf_link_stream_steady_phase": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), c_int, _P]),
    "wf_welch_scratch_doubles": (c_int64, [c_int64, c_int]),
    "wf_welch_psd_c128": (c_int, [_P, _P, c_int64, c_int, c_double, _P, c_double, _P, _P, _P]),
    "wf_phase_tree_f64": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_double, _P, _P]),
    "wf_eye_traces_c128": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_double, _P, _P, _P, _P]),
    "wf_cpm_mf_rows_c128": (c_int, [_P, _P, c_int64, _P, c_int, c_int, c_int, c_int64, c_int, c_int64, _P, _P]),
    "wf_cpm_awgn_mf_rows_c128": (c_int, [_P, _P, c_int64, c_double, c_double, c_double, c_uint64, c_uint64, c_uint64, _P, c_int, c_int,
                                         c_int, c_int64, c_int, c_int64, _P, _P]),
    "wf_viterbi4_state_read": (c_int, [_P, _P, c_int, POINTER(c_int64), _P, _P, _P, _P]),
    "wf_cpm_detector_form": (c_int, [_P, _P, c_int64, c_int, POINTER(c_int)]),
    "wf_cpm_viterbi_detect": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, _P, _P]),
    "wf_cpm_viterbi_detect_samples": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, c_int64, c_int, c_int64, c_int, _P, _P, _P]),
    "wf_cpm_count_errors": (c_int, [_P, _P, _P, c_int, c_int64, _P, _P]),
    "wf_cpm_link_workspace_bytes": (c_int64, [_P]),
    "wf_cpm_link_run": (c_int, [_P, _P, _P, c_int64, _P, POINTER(c_int64), _P]),
    "wf_cpm_link_layout": (c_int, [_P, POINTER(c_int64)]),
    "wf_cpm_link_form": (c_int, [_P, _P, POINTER(c_int)]),
    "wf_cpm_link_stream_workspace_bytes": (c_int64, [_P, c_int64]),
    "wf_cpm_link_stream_layout": (c_int, [_P, c_int64, c_int64, POINTER(c_int64)]),
    "wf_cpm_link_stream_chunk": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), _P]),
    "wf_cpm_link_stream_chunk_phase": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_int64, _P, POINTER(c_int64), c_int, _P]),
}


class CPMDetectorConfig(ctypes.Structure):
    """wf_cpm_detector_config of include/wfhip.h."""

    _fields_ = [("M", c_int), ("p", c_int), ("nh", c_int), ("K", c_int * 2), ("Lp", c_int), ("NC", c_int), ("D", c_int)]


class CPMLinkConfig(ctypes.Structure):
    """wf_cpm_link_config of include/wfhip.h."""

    _fields_ = [
        ("nsym", c_int64), ("sps", c_int), ("degree", c_int), ("mask", c_uint64), ("state", c_uint64), ("skip", c_uint64),
        ("mapper_kind", c_int), ("det", CPMDetectorConfig), ("d_h", c_void_p), ("d_pulse", c_void_p), ("ntaps", c_int),
        ("d_templates", c_void_p), ("d_rot_cs", c_void_p), ("sigma", c_double), ("seed", c_uint64), ("stream_id", c_uint64),
        ("warmup", c_int), ("skip_head", c_int), ("event_slot", c_int), ("fuse", c_int),
    ]


class LinkConfig(ctypes.Structure):
    """wf_link_config of include/wfhip.h."""

    _fields_ = [
        ("nsym", c_int64), ("sps", c_int), ("degree", c_int), ("mask", c_uint64), ("state", c_uint64),
        ("skip", c_uint64), ("differential", c_int), ("d_h", c_void_p), ("d_pulse", c_void_p),
        ("ntaps", c_int), ("d_mf_taps", c_void_p), ("mf_ntaps", c_int), ("mf_nfilt", c_int),
        ("timing_offset", c_int), ("sigma", c_double), ("seed", c_uint64), ("stream_id", c_uint64),
        ("warmup", c_int), ("fuse", c_int), ("event_slot", c_int), ("d_mf_factor", c_void_p),
    ]


_lib = None
_lock = threading.RLock()
_ctxs: dict[int, int] = {}


def lib() -> ctypes.CDLL:
    """Load libwfhip.so (no GPU needed for loading / symbol lookup)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not _LIB_PATH.exists():
                    raise RuntimeError(
                        f"{_LIB_PATH} is missing - build it with `python -m waveforms_amd.csrc.build` "
                        "(there is no CPU fallback)")
                # torch first: it ships its own libamdhip64.so.7; whichever copy of that
                # SONAME is loaded first serves the whole process, and the GPU boxes only
                # enumerate devices through torch's.
                import torch  # noqa: F401

                handle = ctypes.CDLL(str(_LIB_PATH))
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def torch():
    import torch as _t

    return _t


def require_device() -> int:
    t = torch()
    if not t.cuda.is_available():
        raise RuntimeError("waveforms_amd needs a HIP device (MI355X); none is visible and there is "
                           "no CPU fallback")
    return t.cuda.current_device()


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().wf_last_error_string().decode(errors="replace")
    if rc == WF_ERR_VALUE:
        raise ValueError(msg)
    if rc == WF_ERR_KEY:
        raise KeyError(msg)
    raise RuntimeError(f"libwfhip error {rc}: {msg}")


# Options (wf_option) every context this module creates starts with, and the per-device default contexts are switched
# to: the library itself has no process-wide switch (include/wfhip.h: wf_ctx_set_option), so a caller that wants one
# behaviour everywhere — the tests' "run this module in the lane form" fixture — says so here.
default_options: dict[int, int] = {}


def set_option(handle, key: int, value: int) -> None:
    check(lib().wf_ctx_set_option(handle, int(key), int(value)))


def get_option(handle, key: int) -> int:
    v = c_int64(0)
    check(lib().wf_ctx_get_option(handle, int(key), ctypes.byref(v)))
    return int(v.value)


def set_default_option(key: int, value: int) -> None:
    """Record ``key = value`` for every context created from now on and apply it to the live default contexts."""
    with _lock:
        if int(value) == 0:
            default_options.pop(int(key), None)
        else:
            default_options[int(key)] = int(value)
        for handle in _ctxs.values():
            set_option(handle, key, value)


def apply_option_args(pairs) -> None:
    """``["cpm_form=1", "cpm_chunk_calls=320"]`` (the --opt flags of bench.py and the tools) -> set_default_option."""
    for kv in pairs or []:
        key, _, val = str(kv).partition("=")
        set_default_option(globals()["WF_OPT_" + key.strip().upper()], int(val))


def ctx() -> int:
    """The wf_ctx* of the current torch device (created on first use)."""
    dev = require_device()
    if dev not in _ctxs:
        with _lock:
            if dev not in _ctxs:
                out = c_void_p()
                check(lib().wf_ctx_create(dev, ctypes.byref(out)))
                for k, v in default_options.items():
                    set_option(out.value, k, v)
                if not _ctxs:
                    import atexit

                    atexit.register(_destroy_default_contexts)
                _ctxs[dev] = out.value
    return _ctxs[dev]


def _destroy_default_contexts() -> None:
    """At interpreter exit (registered after torch was imported, so it runs before torch's own teardown): RETIRE the
    per-device default contexts — the persistent iteration server a context may still have running leaves the device
    (it would retire by itself within 10 ms of its last request) and the side stream of pipelined links drains, so the
    process never exits with one of its kernels on the device.  The contexts are NOT freed: module-level detectors and
    links (the reference's own examples/soqpsk_detection.py keeps `det` at __main__ level) cached the raw handle and
    their finalisers run after this hook; process teardown releases the memory."""
    for handle in list(_ctxs.values()):
        try:
            if _lib is not None:
                _lib.wf_ctx_retire(handle)
        except Exception:   # noqa: BLE001 - best effort at shutdown
            pass


def new_ctx() -> int:
    """A private wf_ctx* on the current device (own scratch: needed by anything that runs
    concurrently with other library calls on a different stream)."""
    out = c_void_p()
    check(lib().wf_ctx_create(require_device(), ctypes.byref(out)))
    for k, v in default_options.items():
        set_option(out.value, k, v)
    return out.value


def free_ctx(handle) -> None:
    """Destroy a context made by :func:`new_ctx` (synchronises the device).  Safe at interpreter
    shutdown: does nothing once the library handle is gone."""
    if handle and _lib is not None:
        try:
            _lib.wf_ctx_destroy(handle)
        except Exception:   # noqa: BLE001 - best effort in __del__ paths
            pass


def stream() -> int:
    return torch().cuda.current_stream().cuda_stream


def device_check() -> None:
    """Synchronise and raise if a kernel raised the device fault word."""
    check(lib().wf_ctx_check(ctx(), stream()))


# ------------------------------------------------------------------ buffers
_NP2T = {"uint8": "uint8", "int8": "int8", "float64": "float64", "int64": "int64", "int32": "int32"}


def empty(n, dtype: str):
    t = torch()
    require_device()
    return t.empty(n, dtype=getattr(t, _NP2T[dtype]), device="cuda")


def zeros(n, dtype: str):
    t = torch()
    require_device()
    return t.zeros(n, dtype=getattr(t, _NP2T[dtype]), device="cuda")


def to_device(a: np.ndarray):
    """Host ndarray -> device tensor.  complex128 becomes float64[..., 2]."""
    t = torch()
    require_device()
    a = np.ascontiguousarray(a)
    if a.dtype == np.complex128:
        a = a.view(np.float64).reshape(a.shape + (2,))
    if a.size == 0:
        return t.empty(a.shape, dtype=getattr(t, _NP2T[str(a.dtype)]), device="cuda")
    return t.from_numpy(a).to("cuda")


def to_host(x, complex_pairs: bool = False) -> np.ndarray:
    """Device tensor -> fresh host ndarray (float64[..., 2] -> complex128 if asked)."""
    a = x.cpu().numpy()
    if complex_pairs:
        a = np.ascontiguousarray(a).view(np.complex128).reshape(a.shape[:-1])
    return a


def ptr(x) -> int | None:
    return None if x is None else x.data_ptr()
