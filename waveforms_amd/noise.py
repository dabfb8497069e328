"""Complex AWGN — API of reference waveforms/noise.py:5-32.

Two sources:
  * a numpy ``Generator`` (the reference's contract; ``rng=None`` -> the module-level
    ``DEFAULT_RNG`` = PCG64(seed=1)).  The draw is the caller's own numpy generator —
    a sequential PCG64/ziggurat stream that has no parallel form — so it is taken from
    that object exactly as the reference does.
  * a :class:`PhiloxStream` — the MI355X Monte-Carlo source (K5, csrc/wf_awgn.hip):
    counter-based Philox4x32-10 + Box-Muller evaluated on the GPU.
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray

DEFAULT_RNG = np.random.Generator(np.random.PCG64(seed=1))


class PhiloxStream:
    """Position in a counter-based Gaussian stream: (seed, stream id, next sample index)."""

    def __init__(self, seed: int = 1, stream: int = 0, offset: int = 0) -> None:
        self.seed, self.stream, self.offset = int(seed), int(stream), int(offset)

    def draw(self, sigma: float, size: int, signal=None, rot: complex = 1.0, out=None):
        """Device tensor float64[size, 2] = signal*rot + noise; advances the stream."""
        from waveforms_amd import device as dev

        res = dev.awgn(signal, int(size), float(sigma), self.seed, self.stream, self.offset, rot, out)
        self.offset += int(size)
        return res


def generate_complex_awgn(
    sigma: float,
    size: int,
    rng: np.random.Generator | PhiloxStream | None = None,
) -> NDArray[np.complex128]:
    """``size`` samples of circular complex Gaussian noise, std ``sigma`` per component."""
    if isinstance(rng, PhiloxStream):
        from waveforms_amd import _hip

        return _hip.to_host(rng.draw(sigma, size), complex_pairs=True)
    gen = rng or DEFAULT_RNG
    pairs = gen.normal(loc=0, scale=sigma, size=(size, 2))
    return pairs.view(np.complex128).flatten()
