"""Frequency pulse of the ARTM multi-h waveform (reference waveforms/cpm/multih/pulse_filters.py:7-23).

Host-side tap design: the taps are computed once and uploaded to HBM by the modulator.
"""
import numpy as np
from numpy.typing import NDArray

from waveforms_amd.cpm.helpers import normalize_cpm_filter

# h_i = MULTIH_IRIG_NUMER[i % 2] / MULTIH_IRIG_DENOM, i = symbol index
MULTIH_IRIG_DENOM = 16
MULTIH_IRIG_NUMER = np.array([4, 5])


def freq_pulse_multih_irig(sps: int = 8, length: float = 3) -> NDArray[np.float64]:
    """``length``-symbol raised cosine (3RC by default) on ``length * sps + 1`` points,
    scaled so that its running sum ends at sps / 2 (phase pulse q(LT) = 1/2)."""
    n_taps = length * sps + 1
    tau = np.linspace(0, length, num=n_taps)                   # symbol times across the pulse
    # operation order kept as in the reference so that the taps are bit-identical to its output
    lrc = (1 - np.cos(2 * np.pi * tau / length)) / (2 * length)
    return normalize_cpm_filter(sps, lrc)
