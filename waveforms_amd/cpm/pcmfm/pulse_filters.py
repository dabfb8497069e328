"""PCM/FM premodulation pulse — API of reference waveforms/cpm/pcmfm/pulse_filters.py:8-25.
Host-side tap design (scipy Bessel prototype)."""
import numpy as np
from numpy.typing import NDArray
from scipy.signal import besselap, impulse

from waveforms_amd.cpm.helpers import normalize_cpm_filter


def freq_pulse_pcmfm(sps: int = 8, order: int = 4) -> NDArray[np.float64]:
    """NRZ symbol convolved with the impulse response of an ``order``-pole Bessel
    low-pass (magnitude-normalised, sampled over 3/0.7 * 2 time units), normalised."""
    span = 3
    nrz = np.ones(sps) / (2 * sps)
    instants = np.linspace(0, span * 2 / 0.7, num=(span - 1) * sps + 1, endpoint=False)
    _, lowpass = impulse(besselap(order, norm="mag"), T=instants)
    return normalize_cpm_filter(sps, np.convolve(nrz, lowpass, mode="full"))
