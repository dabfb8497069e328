"""Laurent / Perrins PAM decomposition pulses — API of reference
waveforms/cpm/pamapprox.py:12-102.  Setup-time host arithmetic on <= a few hundred taps;
the resulting pulses become matched-filter taps uploaded to HBM
(waveforms_amd.filters.matched)."""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray


def pam_unit_pulse(phase_pulse: NDArray[np.float64], mod_index: float) -> NDArray[np.float64]:
    """u(t) with a leading zero sample (length 2M + 1)."""
    m = phase_pulse.size
    pih = mod_index * np.pi
    u = np.zeros(2 * m + 1, dtype=np.float64)
    u[1:m + 1] = np.sin(2 * pih * phase_pulse) / np.sin(pih)
    u[m + 1:] = np.sin(pih - 2 * pih * phase_pulse) / np.sin(pih)
    return u


def pam_unit_pulse2(phase_pulse: NDArray[np.float64], mod_index: float) -> NDArray[np.float64]:
    """u(t) without the duplicated centre sample (length 2M - 1)."""
    m = phase_pulse.size
    pih = mod_index * np.pi
    u = np.zeros(2 * m - 1, dtype=np.float64)
    u[:m] = np.sin(2 * pih * phase_pulse) / np.sin(pih)
    u[m:] = np.sin(pih - 2 * pih * phase_pulse[1:]) / np.sin(pih)
    return u


def rho_pulses(
    pulse_filter: NDArray[np.float64],
    mod_index: float,
    sps: int,
    k_max: int = 2,
) -> list[NDArray[np.float64]]:
    """First ``k_max`` PAM pulses rho_k of a ternary CPM (Perrins, eq. 3.21).

    rho_k = (k+1) * prod over 2L shifted copies of u(t): rho_0 uses every whole-symbol
    shift 0..L-1 twice, rho_k (k >= 1) the shift pairs (x, x+1); the product is cut to
    its support (from the largest shift to L symbols before the end).
    """
    n_sym = int(pulse_filter.size / sps)
    u = pam_unit_pulse2(np.cumsum(pulse_filter) / sps, mod_index)
    total = n_sym * sps + u.size
    pulses: list[NDArray[np.float64]] = []
    for k in range(k_max):
        step = 1 if k > 0 else 0
        shifts = [x + step * c for x in range(n_sym) for c in (0, 1)]
        stack = np.zeros((len(shifts), total), dtype=np.float64)
        for row, shift in zip(stack, shifts):
            row[shift * sps:shift * sps + u.size] = u
        product = float(k + 1) * np.prod(stack, axis=0)
        pulses.append(product[shifts[-1] * sps:total - n_sym * sps])
    return pulses
