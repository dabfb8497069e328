"""Data products of the reference's plotting helpers (waveforms/viz) — SURVEY 8 row f4.

The reference's ``viz`` package draws figures with matplotlib (and is import-broken at the surveyed
commit: ``viz/__init__.py:1`` imports ``plot_constellation`` while ``viz/constellation.py:15``
defines ``constellation``).  Its numeric content — the Welch PSD behind ``Axes.psd``, the unwrapped
phase traces of the phase tree, the overlapping eye traces — is computed here on the GPU
(csrc/wf_viz.hip) and returned as arrays.  The ``plot_*`` names of the reference are kept as thin
wrappers that draw those arrays when matplotlib is installed.
"""
from __future__ import annotations

import numpy as np

from waveforms_amd import _hip

__all__ = ["power_spectral_density", "eye_diagram_data", "phase_tree_data", "cpm_phase_tree_signal",
           "plot_power_spectral_density", "plot_eye_diagram", "plot_phase_tree", "generate_cpm_phase_tree",
           "plot_constellation", "constellation", "constellation_data"]


def _dev_c128(signal):
    return _hip.to_device(np.ascontiguousarray(signal, dtype=np.complex128))


def power_spectral_density(signal, sps: int, bps: int = 1, nfft: int = 1024):
    """(freqs, pxx) exactly as ``axis.psd(signal * sqrt(bps), NFFT=nfft, Fs=sps / bps,
    scale_by_freq=False)`` computes them (waveforms/viz/psd.py:36-41): Hann window, no overlap,
    two-sided, centred frequency axis in units of the bit rate; 10 log10(pxx) is what is drawn."""
    x = _dev_c128(signal)
    n = int(x.shape[0])
    window = np.hanning(nfft)
    lib = _hip.lib()
    nscr = lib.wf_welch_scratch_doubles(n, nfft)
    if nscr < 0 or n < 1:
        raise ValueError("empty signal or invalid nfft")
    scratch, pxx = _hip.empty(nscr, "float64"), _hip.empty(nfft, "float64")
    _hip.check(lib.wf_welch_psd_c128(_hip.ctx(), _hip.ptr(x), n, nfft, float(np.sqrt(bps)), _hip.ptr(_hip.to_device(window)),
                                     float(np.abs(window).sum()), _hip.ptr(scratch), _hip.ptr(pxx), _hip.stream()))
    freqs = np.fft.fftshift(np.fft.fftfreq(nfft, d=1.0 / (sps / bps)))
    return freqs, _hip.to_host(pxx)


def eye_diagram_data(time, signal, sps: int = 8, modulo: int = 4, t_offset: float = 0):
    """(t, re, im), each [traces][sps * modulo + 1]: the curves waveforms/viz/eye.py:40-55 draws."""
    time = np.ascontiguousarray(time, dtype=np.float64)
    x = _dev_c128(signal)
    # both arrays are read up to n: the shorter one bounds it, as numpy slicing does in the reference
    # (waveforms/viz/eye.py:40-55); a signal shorter than `time` must never be read past its end on the device
    n = min(int(time.size), int(x.shape[0]))
    length = sps * modulo
    ntr = (n - 1) // length if n >= 1 else 0
    outs = [_hip.empty((max(ntr, 0), length + 1), "float64") for _ in range(3)]
    if ntr > 0:
        _hip.check(_hip.lib().wf_eye_traces_c128(_hip.ctx(), _hip.ptr(_hip.to_device(time)), _hip.ptr(x), n, sps, modulo,
                                                 float(t_offset), *(_hip.ptr(o) for o in outs), _hip.stream()))
    return tuple(_hip.to_host(o) for o in outs)


def phase_tree_data(signal, sps: int, off: float | None = None, modulo: int = 4):
    """(t, traces[chunks][modulo * sps]): np.unwrap(np.angle(chunk)) minus ``off`` (default: the
    chunk's first value), the curves of waveforms/viz/tree.py:64-70."""
    x = _dev_c128(signal)
    n = int(x.shape[0])
    length = sps * modulo
    out = _hip.empty((n // length, length), "float64")
    if n // length:
        _hip.check(_hip.lib().wf_phase_tree_f64(_hip.ctx(), _hip.ptr(x), n, sps, modulo, int(off is None),
                                                0.0 if off is None else float(off), _hip.ptr(out), _hip.stream()))
    return np.linspace(0, modulo, modulo * sps, endpoint=False), _hip.to_host(out)


def cpm_phase_tree_signal(pulse_filter, mod_index, encoder, sps: int):
    """The concatenated signal whose phase tree waveforms/viz/tree.py:99-146 draws: every input
    sequence of ``length = len(pulse) // sps`` symbols after an all-zero prefix, encoded from state
    0, duplicates removed, modulated (on the GPU), samples [length*sps - 1, 2*length*sps - 1) kept.
    Returns (signal, length)."""
    from waveforms_amd.cpm.modulate import cpm_modulate

    bps = encoder.input_cardinality
    length = int(np.asarray(pulse_filter).size // sps)
    packed = bps * length
    seqs = set()
    for i in range(2 ** packed):
        encoder.state = 0
        bits = np.array([0] * packed + [(i >> j) & 1 for j in range(packed)], dtype=np.uint8)
        seqs.add(tuple(int(v) for v in encoder.encode(bits)))
    seqs = sorted(seqs)
    out = np.zeros(len(seqs) * length * sps + 1, dtype=np.complex128)
    for i, symbols in enumerate(seqs):
        _t, sig = cpm_modulate(np.array(symbols, dtype=np.int8), mod_index=mod_index, pulse_filter=pulse_filter, sps=sps)
        out[i * length * sps:(i + 1) * length * sps] = sig[length * sps - 1:2 * length * sps - 1]
    return out, length


# ---------------------------------------------------------------- the reference's plot_* names
def _axes(n=1):
    try:
        import matplotlib.pyplot as plt
    except ImportError as exc:   # pragma: no cover
        raise RuntimeError("plotting needs matplotlib; the *_data functions return the arrays without it") from exc
    return plt.subplots(n)[-1]


def plot_power_spectral_density(signal, sps, bps=1, nfft=1024, axis=None):
    axis = axis or _axes()
    freqs, pxx = power_spectral_density(signal, sps, bps, nfft)
    axis.plot(freqs, 10 * np.log10(pxx))
    axis.set_ylabel("Amplitude [dBc]")
    axis.set_ylim([-80, 0])
    axis.set_xlim([-2, 2])
    axis.set_xlabel("Normalized Frequency [$T_b$ = 1]")
    axis.set_title("Power Spectral Density")
    axis.grid(which="both", linestyle=":")
    return axis.figure


def plot_eye_diagram(time, signal, sps=8, modulo=4, t_offset=0, color=None, axes=None):
    real_ax, imag_ax = axes if axes is not None else _axes(2)
    t, re, im = eye_diagram_data(time, signal, sps, modulo, t_offset)
    for ax, data, name in ((real_ax, re, "In-phase"), (imag_ax, im, "Quadrature")):
        ax.plot(t.T, data.T, linewidth=0.3, color=color, alpha=0.7)
        ax.set_title(f"Eye Diagram ({name})")
        ax.set_ylabel("Amplitude")
        ax.set_xlabel("Normalized Time [t/T]")
    return real_ax.figure


def plot_phase_tree(signal, sps, off=None, modulo=4, color=None, axis=None):
    axis = axis or _axes()
    t, traces = phase_tree_data(signal, sps, off, modulo)
    axis.plot(t, traces.T, color=color or "k", alpha=0.3, linewidth=0.5)
    axis.set_ylabel("Phase [radians]")
    axis.set_xlabel("Symbol Time [t/T]")
    axis.grid(which="both", linestyle=":")
    axis.set_title("Phase Tree")
    return axis.figure


def generate_cpm_phase_tree(pulse_filter, mod_index, encoder, sps, axis=None):
    signal, length = cpm_phase_tree_signal(pulse_filter, mod_index, encoder, sps)
    return plot_phase_tree(signal=signal, sps=sps, modulo=length, axis=axis)


def constellation_data(signal, n: int = 1024):
    """(re, im) of the first ``n`` samples: the polyline waveforms/viz/constellation.py:36-43 draws (the caller
    decimates to symbol rate — examples/soqpsk_example.py:168-171 — there is no arithmetic to accelerate)."""
    pts = np.asarray(signal)[:n]
    return np.ascontiguousarray(pts.real), np.ascontiguousarray(pts.imag)


def constellation(signal, n: int = 1024, color=None, axis=None):
    """The reference's viz/constellation.py:15-44, same arguments: the first ``n`` samples as a marked polyline."""
    axis = axis or _axes()
    re, im = constellation_data(signal, n)
    axis.plot(re, im, color=color, alpha=0.2, marker="o")
    axis.set_title("Constellation")
    axis.set_ylabel("Quadrature [V]")
    axis.set_xlabel("In-phase [V]")
    axis.grid(which="both", linestyle=":")
    return axis.figure


plot_constellation = constellation     # the name waveforms/viz/__init__.py:1 tries to import
