"""Pulse helpers — API of reference waveforms/cpm/helpers.py:5-19 (host-side tap design)."""
import numpy as np
from numpy.typing import NDArray


def normalize_cpm_filter(sps: int, g: NDArray[np.float64]) -> NDArray[np.float64]:
    """Scale ``g`` so that sum(g) / sps == 1/2 (the CPM phase-pulse end value)."""
    gain = sps / (np.sum(g) * 2)
    return gain * g
