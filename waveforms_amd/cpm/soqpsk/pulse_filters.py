"""SOQPSK frequency pulses — API of reference waveforms/cpm/soqpsk/pulse_filters.py:5-116.

Tap design is host-side setup (<= a few hundred doubles, computed once); the taps are
uploaded to HBM by the modulator.
"""
import numpy as np
from numpy.typing import NDArray

SOQPSK_NUMER = 1
SOQPSK_DENOM = 4


def freq_pulse_soqpsk(
    t1: float = 1.5,
    t2: float = 0.5,
    rho: float = 0.7,
    b: float = 1.25,
    sps: int = 8,
) -> NDArray[np.float64]:
    """Windowed spectral-raised-cosine SOQPSK pulse, normalised to sum(g)/sps = 1/2.

    The pulse spans ``4 (t1 + t2)`` symbol times; the raised-cosine window tapers it
    over the last ``2 t2`` on each side (IRIG-106 SOQPSK-TG: defaults).
    """
    half_span = 2 * (t1 + t2)
    n_taps = int(half_span * sps * 2) + 1
    tau = np.linspace(-half_span, half_span, num=n_taps, dtype=np.float64, endpoint=True)
    body = np.cos(np.pi * rho * b * tau / 2) / (1 - np.power(rho * b * tau, 2)) * np.sinc(b * tau / 2)
    window = np.ones(tau.shape, dtype=np.float64)
    if t2 > 0:
        mag = np.abs(tau)
        taper = np.where((mag >= 2 * t1) & (mag <= half_span))
        window[taper] = (1 + np.cos(np.pi * (tau[taper] / 2 - t1) / t2)) / 2
        window[np.where(mag > half_span)] = 0
    gain = sps / (np.sum(body * window) * 2)
    return gain * body * window


def freq_pulse_soqpsk_a(sps: int = 8) -> NDArray[np.float64]:
    """SOQPSK-A."""
    return freq_pulse_soqpsk(t1=1.4, t2=0.6, rho=1.0, b=1.35, sps=sps)


def freq_pulse_soqpsk_b(sps: int = 8) -> NDArray[np.float64]:
    """SOQPSK-B."""
    return freq_pulse_soqpsk(t1=2.8, t2=1.2, rho=0.5, b=1.45, sps=sps)


def freq_pulse_soqpsk_mil(sps: int = 8) -> NDArray[np.float64]:
    """SOQPSK-MIL (MIL-STD 188-181): one-symbol rectangular pulse of height 1/2, with
    the reference's leading zero tap (``sps + 1`` taps)."""
    g = np.full(sps + 1, 0.5, dtype=np.float64)
    g[0] = 0.0
    return g


def freq_pulse_soqpsk_tg(sps: int = 8) -> NDArray[np.float64]:
    """SOQPSK-TG (IRIG-106)."""
    return freq_pulse_soqpsk(sps=sps)
