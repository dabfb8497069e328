"""Matched-filter banks for the SOQPSK detectors.

The reference builds these inline in its example (examples/soqpsk_detection.py:134-173):
three ``np.convolve(received, taps, "same")`` for the pulse-truncation (PT) bank and six
for the PAM bank.  Here the tap design stays on the host and the correlation runs on the
GPU (K6/K7, csrc/wf_mfbank.hip), evaluated only at the symbol-rate samples the detector
consumes.
"""
from __future__ import annotations

import numpy as np

PSEUDO_SYMBOLS = np.array(
    [[-1j, 1, 1j],
     [np.sqrt(2) / 2 * (1 - 1j), np.sqrt(2) / 2, np.sqrt(2) / 2 * (1 + 1j)]], dtype=np.complex128)


def pt_matched_filter_taps(pulse_filter, mod_index: float, sps: int, alphabet=(-2, 0, 2), truncation: int = 1):
    """exp(-2j pi h alpha q_t): phase pulse q = cumsum(g)/sps cut to ``truncation`` symbols
    around its centre (examples/soqpsk_detection.py:135-156).  One row per alpha."""
    n_sym = int(pulse_filter.size / sps)
    q = np.cumsum(pulse_filter) / sps
    lo = int((n_sym - truncation) * sps / 2)
    hi = int((n_sym + truncation) * sps / 2) + 1
    return np.array([np.exp(-2j * np.pi * mod_index * a * q[lo:hi]) for a in alphabet])


def pam_matched_filter_taps(pulse_filter, mod_index: float, sps: int, pseudo_symbols=PSEUDO_SYMBOLS):
    """PAM bank as ONE complex FIR per symbol hypothesis: sum_k conj(pseudo[k, s]) rho_k,
    rho_k zero-padded at the tail to a common length (examples/soqpsk_detection.py:158-173;
    convolution is linear, so weighting the taps equals weighting the outputs)."""
    from waveforms_amd.cpm.pamapprox import rho_pulses

    k_max, n_hyp = pseudo_symbols.shape
    rho = rho_pulses(pulse_filter, mod_index, sps, k_max=k_max)
    d_max = max(r.size for r in rho)
    taps = np.zeros((n_hyp, d_max), dtype=np.complex128)
    for s in range(n_hyp):
        for k in range(k_max):
            taps[s, :rho[k].size] += rho[k] * np.conj(pseudo_symbols[k, s])
    return taps


def pam_bank_factors(pulse_filter, mod_index: float, sps: int, pseudo_symbols=PSEUDO_SYMBOLS):
    """The PAM bank as the reference computes it (examples/soqpsk_detection.py:158-173): the real rho pulses, zero-padded
    at the tail to a common length, and the weights conj(pseudo[k, s]) their filter outputs are combined with.
    Returns (rho: float64[k_max, d_max], weights: complex128[n_hyp, k_max]); ``pam_matched_filter_taps`` is
    ``weights @ rho``."""
    from waveforms_amd.cpm.pamapprox import rho_pulses

    k_max, n_hyp = pseudo_symbols.shape
    rho = rho_pulses(pulse_filter, mod_index, sps, k_max=k_max)
    d_max = max(r.size for r in rho)
    basis = np.zeros((k_max, d_max), dtype=np.float64)
    for k in range(k_max):
        basis[k, :rho[k].size] = rho[k]
    return basis, np.conj(pseudo_symbols).T.copy()


def factor_long_bank(taps, rtol: float = 1e-13):
    """Any 3-filter bank whose complex taps span a TWO-dimensional real row space (the PAM bank: two real pulses, complex
    weights) as (basis: float64[2, n_taps], weights: complex128[3, 2]) with ``taps == weights @ basis`` to ``rtol``;
    None when the bank has no such form.  The basis is the orthonormal one of an SVD of [Re taps; Im taps]."""
    taps = np.asarray(taps, dtype=np.complex128)
    if taps.ndim != 2 or taps.shape[0] != 3:
        return None
    m = np.concatenate((taps.real, taps.imag))
    _u, sv, vt = np.linalg.svd(m, full_matrices=False)
    if sv.size > 2 and sv[2] > rtol * sv[0]:
        return None
    basis = np.ascontiguousarray(vt[:2])
    proj = m @ basis.T
    weights = proj[:3] + 1j * proj[3:]
    if np.abs(weights @ basis - taps).max() > 16 * rtol * np.abs(taps).max():
        return None
    return basis, weights


def pack_bank_factors(basis, weights):
    """(basis, weights) as the flat float64 buffer ``wf_link_config.d_mf_factor`` points at: b_0, b_1, then the 3 x 2 weights
    as (re, im) pairs in s-major order."""
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    weights = np.ascontiguousarray(weights, dtype=np.complex128)
    if basis.shape[0] != 2 or weights.shape != (3, 2):
        raise ValueError("a factored bank is two real filters and a 3 x 2 complex combination")
    return np.concatenate((basis.ravel(), weights.ravel().view(np.float64)))


class MatchedFilterBank:
    """``taps`` (n_filters x n_taps complex) resident in HBM; ``__call__`` filters a
    received burst and returns one row of ``n_filters`` outputs per kept sample."""

    def __init__(self, taps) -> None:
        from waveforms_amd import _hip

        self.taps = np.ascontiguousarray(taps, dtype=np.complex128)
        self.d_taps = _hip.to_device(self.taps)

    def apply_device(self, received, first: int, step: int, ncols: int):
        from waveforms_amd import device as dev

        return dev.mf_bank(received, self.d_taps, first, step, ncols)

    def __call__(self, received, first: int = 0, step: int = 1, ncols: int | None = None):
        """Host in / host out: complex128[ncols, n_filters]."""
        from waveforms_amd import _hip

        received = np.asarray(received, dtype=np.complex128)
        if ncols is None:
            ncols = (received.size - first + step - 1) // step
        rows = self.apply_device(_hip.to_device(received), first, step, ncols)
        return _hip.to_host(rows, complex_pairs=True)
