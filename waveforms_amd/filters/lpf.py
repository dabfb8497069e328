"""Kaiser-window FIR low-pass designer — API of reference waveforms/filters/lpf.py:13-34
(host-side tap design through scipy, as in the reference)."""
from __future__ import annotations

from typing import TYPE_CHECKING

from scipy.signal import firwin, kaiserord

if TYPE_CHECKING:
    import numpy as np
    from numpy.typing import NDArray


def kaiser_fir_lpf(
    sps: int,
    f_cutoff: float,
    width: float | None = None,
    ripple_db: float = 80.0,
) -> NDArray[np.float64]:
    """Low-pass taps with cutoff ``f_cutoff`` (symbol-rate units, Nyquist = sps/2),
    transition ``width`` (default 1/sps, Nyquist-normalised) and ``ripple_db`` stop-band
    attenuation."""
    n_taps, beta = kaiserord(ripple_db, width or 1 / sps)
    return firwin(numtaps=n_taps, cutoff=f_cutoff / (sps / 2), window=("kaiser", beta))
