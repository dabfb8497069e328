"""ARTM multi-h CPM (IRIG-106 Tier II) pulse — API of reference
waveforms/cpm/multih/pulse_filters.py:7-23.  Host-side tap design."""
import numpy as np
from numpy.typing import NDArray

from waveforms_amd.cpm.helpers import normalize_cpm_filter

MULTIH_IRIG_NUMER = np.array([4, 5])
MULTIH_IRIG_DENOM = 16


def freq_pulse_multih_irig(sps: int = 8, length: float = 3) -> NDArray[np.float64]:
    """Raised-cosine frequency pulse spanning ``length`` symbols (3RC), normalised."""
    tau = np.linspace(0, length, num=length * sps + 1)
    raised_cosine = (1 - np.cos(2 * np.pi * tau / length)) / (2 * length)
    return normalize_cpm_filter(sps, raised_cosine)
