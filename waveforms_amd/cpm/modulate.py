"""CPM modulator — API of reference waveforms/cpm/modulate.py:12-101.

Every function takes and returns host ndarrays like the reference (fresh arrays the
caller may mutate in place); the arithmetic runs on the GPU:
  upsample + frequency-pulse FIR   -> K3  csrc/wf_fir.hip
  phase accumulate (mod sps) + exp -> K4  csrc/wf_phase.hip   (chained prefix scan)
``cpm_modulate_device`` is the same pipeline without the host round trips.
"""
from __future__ import annotations

from typing import TYPE_CHECKING

import numpy as np

if TYPE_CHECKING:
    from numpy.typing import NDArray


def phase_modulate(phase: NDArray[np.float64], sensitivity: float) -> NDArray[np.complex128]:
    """exp(1j * sensitivity * phase)."""
    from waveforms_amd import _hip, device as dev

    phase = np.asarray(phase, dtype=np.float64)
    out = dev.phase_modulate(_hip.to_device(phase.ravel()), sensitivity)
    return _hip.to_host(out, complex_pairs=True).reshape(phase.shape)


def frequency_modulate(
    freq_pulses: NDArray[np.float64],
    sps: int,
    initial_phase: float = 0,
) -> NDArray[np.complex128]:
    """Integrate frequency pulses with a running modulo-``sps`` accumulator
    (revolutions), scale by 2 pi / sps, add ``initial_phase`` and exponentiate."""
    from waveforms_amd import _hip, device as dev

    fp = np.asarray(freq_pulses, dtype=np.float64)
    out = dev.phase_cexp(_hip.to_device(fp.ravel()), int(sps), float(initial_phase))
    res = _hip.to_host(out, complex_pairs=True).reshape(fp.shape)
    _hip.device_check()
    return res


def _mod_index_vector(mod_index) -> np.ndarray:
    if isinstance(mod_index, (float, int)):
        mod_index = [float(mod_index)]
    return np.atleast_1d(np.asarray(mod_index, dtype=np.float64))


def cpm_modulate_device(symbols, mod_index, pulse_filter, sps: int = 8, fused: bool = True):
    """Device-resident modulator: ``symbols`` an int8 device tensor; returns the complex
    baseband signal as a float64[n, 2] device tensor (no time axis, no host copies).
    ``fused=False`` forces the two stage kernels (FIR, then phase scan)."""
    from waveforms_amd import _hip, device as dev

    h = _hip.to_device(_mod_index_vector(mod_index))
    g = _hip.to_device(np.asarray(pulse_filter, dtype=np.float64))
    return dev.cpm_modulate(symbols, h, g, int(sps), np.pi / 4, fused)


def cpm_modulate(
    symbols: NDArray[np.int8],
    mod_index: float | NDArray[np.float64],
    pulse_filter: NDArray[np.float64],
    sps: int = 8,
) -> tuple[NDArray[np.float64], NDArray[np.complex128]]:
    """Generic CPM modulation of already-mapped ``symbols``.

    Symbol k uses ``mod_index[k % len(mod_index)]``; impulses sit at samples
    sps, 2 sps, ..., N sps of an (N+1) sps long train, are shaped by ``pulse_filter``
    ("same" convolution) and frequency-modulated with a pi/4 start phase.

    Returns:
        (normalized_time, signal): float64 and complex128 host arrays.
    """
    from waveforms_amd import _hip, device as dev

    symbols = np.asarray(symbols)
    sps = int(sps)
    n_points = (symbols.size + 1) * sps
    d_time = dev.time_axis(n_points, (symbols.size + 1) / n_points)
    d_sig = cpm_modulate_device(_hip.to_device(symbols.astype(np.int8, copy=False)), mod_index,
                                pulse_filter, sps)
    time = _hip.to_host(d_time)
    signal = _hip.to_host(d_sig, complex_pairs=True)
    _hip.device_check()
    return time, signal
