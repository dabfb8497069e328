"""Device-tensor level wrappers of the C ABI (one function per kernel entry point).

Inputs and outputs are torch tensors on the GPU (used purely as HBM buffers);
complex128 data is float64[..., 2].  Everything is asynchronous on the current torch
stream.  The reference-shaped host API (``waveforms_amd.cpm`` ...) is built on these.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _hip

_u64 = lambda v: ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)  # noqa: E731


def lfsr_bits(degree: int, mask: int, state: int, n: int, skip: int = 0):
    """K1 -> (bits u8[n] on device, register state after skip+n steps)."""
    out = _hip.empty(max(n, 1) + 16, "uint8")
    st = ctypes.c_uint64(0)
    _hip.check(_hip.lib().wf_lfsr_generate(_hip.ctx(), degree, _u64(mask), _u64(state), _u64(skip),
                                           _hip.ptr(out), n, ctypes.byref(st), _hip.stream()))
    return out[:n], st.value


def fsm_encode(next_tab: np.ndarray, out_tab: np.ndarray, bits, i0: int = 0, state0: int = 0,
               want_state: bool = True):
    """K2 -> (symbols i8 on device, final state or None)."""
    columns, states, ninp = next_tab.shape
    card = ninp.bit_length() - 1
    nbits = int(bits.numel())
    nxt = np.ascontiguousarray(next_tab, dtype=np.uint8)
    out = np.ascontiguousarray(out_tab, dtype=np.int8)
    sym = _hip.empty(nbits // card + 16, "int8")
    st = ctypes.c_int(state0)
    _hip.check(_hip.lib().wf_fsm_encode(
        _hip.ctx(), nxt.ctypes.data, out.ctypes.data, columns, states, card, _hip.ptr(bits), nbits, i0,
        state0, _hip.ptr(sym), ctypes.byref(st) if want_state else None, _hip.stream()))
    return sym[:nbits // card], (st.value if want_state else None)


def symbol_map(kind: int, bits, parity: int = 0, mem=(0, 0)):
    n = int(bits.numel())
    if kind == 1 and n % 2:
        raise ValueError("Odd length bit array passed into quaternary mapper.")
    nout = n // 2 if kind == 1 else n
    out = _hip.empty(nout + 16, "int8")
    _hip.check(_hip.lib().wf_symbol_map(_hip.ctx(), kind, _hip.ptr(bits), n, parity, int(mem[0]), int(mem[1]),
                                        _hip.ptr(out), _hip.stream()))
    return out[:nout]


def upsample_fir(symbols, h, pulse, sps: int):
    """K3 -> freq pulses f64[max((N+1)*sps, M)] on device."""
    nsym, ntaps = int(symbols.numel()), int(pulse.numel())
    n = _hip.lib().wf_fir_out_len(nsym, sps, ntaps)
    out = _hip.empty(n, "float64")
    _hip.check(_hip.lib().wf_upsample_fir_f64(_hip.ctx(), _hip.ptr(symbols), nsym, _hip.ptr(h), int(h.numel()),
                                              _hip.ptr(pulse), ntaps, sps, _hip.ptr(out), _hip.stream()))
    return out


def phase_cexp(freq, sps: int, phi0: float, revs_in: float = 0.0, revs_out=None):
    """K4 -> complex signal f64[n, 2] on device."""
    n = int(freq.numel())
    out = _hip.empty((n, 2), "float64")
    _hip.check(_hip.lib().wf_phase_cexp_f64(_hip.ctx(), _hip.ptr(freq), n, sps, float(phi0), float(revs_in),
                                            _hip.ptr(out), _hip.ptr(revs_out), _hip.stream()))
    return out


def cpm_modulate(symbols, h, pulse, sps: int, phi0: float = math.pi / 4, fused: bool = True):
    """K3 + K4.  Fused single-pass kernel when the configuration allows it, otherwise (or
    with fused=False) the FIR stage followed by the phase-scan stage."""
    nsym, ntaps = int(symbols.numel()), int(pulse.numel())
    if fused:
        n = _hip.lib().wf_fir_out_len(nsym, sps, ntaps)
        out = _hip.empty((n, 2), "float64")
        rc = _hip.lib().wf_cpm_modulate_c128(_hip.ctx(), _hip.ptr(symbols), nsym, _hip.ptr(h), int(h.numel()),
                                            _hip.ptr(pulse), ntaps, sps, float(phi0), _hip.ptr(out), _hip.stream())
        if rc == 0:
            return out
        if rc < 0:
            _hip.check(rc)
    return phase_cexp(upsample_fir(symbols, h, pulse, sps), sps, phi0)


def phase_modulate(phase, sens: float):
    n = int(phase.numel())
    out = _hip.empty((n, 2), "float64")
    _hip.check(_hip.lib().wf_phase_modulate_f64(_hip.ctx(), _hip.ptr(phase), n, float(sens), _hip.ptr(out),
                                                _hip.stream()))
    return out


def time_axis(n: int, step: float):
    out = _hip.empty(n, "float64")
    _hip.check(_hip.lib().wf_time_axis_f64(_hip.ctx(), n, float(step), _hip.ptr(out), _hip.stream()))
    return out


def awgn(signal, n: int, sigma: float, seed: int, stream_id: int = 0, first_index: int = 0,
         rot: complex = 1.0, out=None):
    """K5: out = signal*rot + noise (signal may be None -> pure noise)."""
    if out is None:
        out = _hip.empty((n, 2), "float64")
    rot = complex(rot)
    _hip.check(_hip.lib().wf_awgn_c128(_hip.ctx(), _hip.ptr(signal), n, rot.real, rot.imag, float(sigma),
                                       _u64(seed), _u64(stream_id), _u64(first_index), _hip.ptr(out),
                                       _hip.stream()))
    return out


def box_muller32(words, sigma: float):
    """Box-Muller of caller-supplied word pairs: ``words`` int32[n, 2] (bit patterns of the
    uint32 radius / angle words) -> complex samples f64[n, 2]."""
    n = int(words.shape[0])
    out = _hip.empty((n, 2), "float64")
    _hip.check(_hip.lib().wf_box_muller32_c128(_hip.ctx(), _hip.ptr(words), n, float(sigma), _hip.ptr(out), _hip.stream()))
    return out


def mf_bank(received, taps, first: int, step: int, ncols: int):
    """K6/K7 -> rows f64[ncols, nfilt, 2] on device; taps f64[nfilt, ntaps, 2]."""
    nsamp = int(received.shape[0])
    nfilt, ntaps = int(taps.shape[0]), int(taps.shape[1])
    out = _hip.empty((max(ncols, 0), nfilt, 2), "float64")
    _hip.check(_hip.lib().wf_mf_bank_c128(_hip.ctx(), _hip.ptr(received), nsamp, _hip.ptr(taps), nfilt, ntaps,
                                          first, step, ncols, _hip.ptr(out), _hip.stream()))
    return out


def awgn_mf_bank(signal, taps, first: int, step: int, ncols: int, sigma: float, seed: int, stream_id: int = 0,
                 first_index: int = 0, rot: complex = 1.0):
    """K5 + K6 fused: rows of MF(signal*rot + noise) without materialising the noisy signal."""
    nsamp = int(signal.shape[0])
    nfilt, ntaps = int(taps.shape[0]), int(taps.shape[1])
    out = _hip.empty((max(ncols, 0), nfilt, 2), "float64")
    rot = complex(rot)
    _hip.check(_hip.lib().wf_awgn_mf_bank_c128(_hip.ctx(), _hip.ptr(signal), nsamp, rot.real, rot.imag, float(sigma),
                                               _u64(seed), _u64(stream_id), _u64(first_index), _hip.ptr(taps), nfilt,
                                               ntaps, first, step, ncols, _hip.ptr(out), _hip.stream()))
    return out


def viterbi_detect_count(mf_rows, ref_bits, ref_syms, skip: int, ncompare: int, counts, differential: bool = True,
                         warmup: int = 0):
    """K8-K11 fused: decisions + error counts (added to ``counts``) in one launch."""
    ncalls = int(mf_rows.shape[0])
    bits = _hip.empty(ncalls + 16, "uint8")
    syms = _hip.empty(ncalls + 16, "int8")
    _hip.check(_hip.lib().wf_viterbi4_detect_count(_hip.ctx(), _hip.ptr(mf_rows), ncalls, int(bool(differential)), warmup,
                                                   _hip.ptr(bits), _hip.ptr(syms), _hip.ptr(ref_bits), _hip.ptr(ref_syms),
                                                   skip, ncompare, _hip.ptr(counts), _hip.stream()))
    return bits[:ncalls], syms[:ncalls]


def viterbi_detect(mf_rows, differential: bool = True, warmup: int = 0, state=None, ctx=None):
    """K8-K10 (length 2) -> (bits u8[ncalls], symbols i8[ncalls]) on device.  ``ctx``: a private
    wf_ctx (own scratch and own unmerged-chunk counter) instead of the device's shared one."""
    ncalls = int(mf_rows.shape[0])
    bits = _hip.empty(ncalls + 16, "uint8")
    syms = _hip.empty(ncalls + 16, "int8")
    _hip.check(_hip.lib().wf_viterbi4_detect(ctx if ctx is not None else _hip.ctx(), _hip.ptr(mf_rows), ncalls, int(bool(differential)),
                                             warmup, _hip.ptr(bits), _hip.ptr(syms), _hip.ptr(state),
                                             _hip.stream()))
    return bits[:ncalls], syms[:ncalls]


def viterbi_detect_window(mf_rows, length: int, differential: bool = True, warmup: int = 0, state=None, ctx=None):
    """K8-K10 for any window ``length`` (1 .. 64, odd ones as the reference pairs them) -> (bits u8[ncalls], symbols i8[ncalls]) on device;
    ``state``: float64[wf_viterbi4_window_state_bytes() / 8] carry (zeros = a fresh detector) or None."""
    ncalls = int(mf_rows.shape[0])
    bits = _hip.empty(ncalls + 16, "uint8")
    syms = _hip.empty(ncalls + 16, "int8")
    _hip.check(_hip.lib().wf_viterbi4_detect_window(ctx if ctx is not None else _hip.ctx(), _hip.ptr(mf_rows), ncalls, int(length),
                                                    int(bool(differential)), warmup, _hip.ptr(bits), _hip.ptr(syms),
                                                    _hip.ptr(state), _hip.stream()))
    return bits[:ncalls], syms[:ncalls]


def viterbi_unmerged(reset: bool = True, ctx=None) -> int:
    """Chunks of the chunk-parallel detectors left UNPROVEN since the last reset (``wf_viterbi4_unmerged``;
    synchronises).  0 = every batch call reproduced the sequential detector bit for bit — which, since chunks
    that miss their warm-up are repaired on the device (cascading into the following chunks where needed), is
    always the case unless the context's WF_OPT_DET_REPAIR option turned the repairs off."""
    n = ctypes.c_int64(0)
    _hip.check(_hip.lib().wf_viterbi4_unmerged(ctx if ctx is not None else _hip.ctx(), ctypes.byref(n), int(reset), _hip.stream()))
    return int(n.value)


def viterbi_repaired(reset: bool = True, ctx=None) -> int:
    """Chunk repairs the detectors ran on the device since the last reset, every round counted
    (``wf_viterbi_repaired``; synchronises): chunks that missed their warm-up, run again from the true state."""
    n = ctypes.c_int64(0)
    _hip.check(_hip.lib().wf_viterbi_repaired(ctx if ctx is not None else _hip.ctx(), ctypes.byref(n), int(reset), _hip.stream()))
    return int(n.value)


def viterbi_cascaded(reset: bool = True, ctx=None) -> int:
    """Of those repairs, the ones whose chunk ENDED in a different state than before and therefore handed on to
    the next chunk (``wf_viterbi_cascaded``; synchronises)."""
    n = ctypes.c_int64(0)
    _hip.check(_hip.lib().wf_viterbi_cascaded(ctx if ctx is not None else _hip.ctx(), ctypes.byref(n), int(reset), _hip.stream()))
    return int(n.value)


def count_errors(det_syms, ref_syms, det_bits, ref_bits, m: int, counts=None):
    """K11: counts[0] += symbol errors, counts[1] += bit errors over the first m elements."""
    if counts is None:
        counts = _hip.zeros(2, "int64")
    _hip.check(_hip.lib().wf_count_errors(_hip.ctx(), _hip.ptr(det_syms), _hip.ptr(ref_syms), _hip.ptr(det_bits),
                                          _hip.ptr(ref_bits), m, _hip.ptr(counts), _hip.stream()))
    return counts


def decimation(size: int, sps: int, length: int, timing_offset: int):
    """(first, ncols) of the samples n in range(size - length*sps) with
    (n + timing_offset) % sps == 0 (reference examples/soqpsk_detection.py:189-192)."""
    limit = size - length * sps
    first = (-timing_offset) % sps
    ncols = (limit - first + sps - 1) // sps if limit > first else 0
    return first, ncols



def cpm_mf_rows(received, templates, start0: int, sps: int, ncalls: int):
    """Generic CPM detector front end -> rows f64[ncalls, nfilt, 2]; templates f64[nh, nfilt, ntm, 2]."""
    nsamp = int(received.shape[0])
    nh, nfilt, ntm = (int(v) for v in templates.shape[:3])
    out = _hip.empty((max(ncalls, 0), nfilt, 2), "float64")
    _hip.check(_hip.lib().wf_cpm_mf_rows_c128(_hip.ctx(), _hip.ptr(received), nsamp, _hip.ptr(templates), nh, nfilt, ntm,
                                              start0, sps, ncalls, _hip.ptr(out), _hip.stream()))
    return out


def cpm_count_errors(decided_u, ref_alpha, M: int, m: int, counts=None):
    """counts[0] += symbol errors, counts[1] += bit errors of decided U against transmitted alpha."""
    if counts is None:
        counts = _hip.zeros(2, "int64")
    _hip.check(_hip.lib().wf_cpm_count_errors(_hip.ctx(), _hip.ptr(decided_u), _hip.ptr(ref_alpha), M, m, _hip.ptr(counts),
                                              _hip.stream()))
    return counts
