"""CPU statement of the arrays behind the reference's plots.

TEST INFRASTRUCTURE ONLY.  Parity status: PINNED — ``tests/golden/viz.npz`` holds what
matplotlib's own ``mlab.psd`` (the routine ``Axes.psd`` calls, waveforms/viz/psd.py:36-41) and
numpy's ``angle`` / ``unwrap`` (waveforms/viz/tree.py:64-70) return on a reference-modulated
signal; ``tests/test_viz.py`` checks these functions and the HIP kernels against it.
Citations are relative to ``/root/reference``."""
from __future__ import annotations

import numpy as np

__all__ = ["psd_welch", "eye_traces", "phase_tree_traces"]


def psd_welch(signal, sps: int, bps: int = 1, nfft: int = 1024):
    """waveforms/viz/psd.py:36-41 -> matplotlib.mlab.psd(x * sqrt(bps), NFFT, Fs = sps/bps,
    scale_by_freq=False): Hann window, no overlap / detrend, two-sided, fftshift order."""
    x = np.asarray(signal, dtype=np.complex128) * np.sqrt(bps)
    if x.size < nfft:
        x = np.concatenate([x, np.zeros(nfft - x.size, dtype=np.complex128)])
    nseg = x.size // nfft
    w = np.hanning(nfft)
    spec = np.fft.fft(x[:nseg * nfft].reshape(nseg, nfft) * w, axis=1)
    pxx = (np.abs(spec) ** 2).mean(axis=0) / np.abs(w).sum() ** 2
    return np.fft.fftshift(np.fft.fftfreq(nfft, 1.0 / (sps / bps))), np.fft.fftshift(pxx)


def eye_traces(time, signal, sps: int = 8, modulo: int = 4, t_offset: float = 0):
    """waveforms/viz/eye.py:40-55: trace i = samples [i L, i L + L], L = sps * modulo."""
    time, signal = np.asarray(time), np.asarray(signal)
    L = sps * modulo
    n = (time.size - 1) // L
    idx = np.arange(n)[:, None] * L + np.arange(L + 1)[None, :]
    return (time[idx] - time[idx[:, :1]]) + t_offset, signal.real[idx], signal.imag[idx]


def phase_tree_traces(signal, sps: int, off=None, modulo: int = 4):
    """waveforms/viz/tree.py:64-70."""
    phase = np.angle(np.asarray(signal))
    L = sps * modulo
    rows = []
    for chunk in range(phase.size // L):
        sl = np.unwrap(phase[chunk * L:(chunk + 1) * L])
        rows.append(sl - (off if off is not None else sl[0]))
    return np.linspace(0, modulo, L, endpoint=False), np.array(rows).reshape(-1, L)
