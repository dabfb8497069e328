"""Maximal-length PRBS sources — API of reference waveforms/glfsr/pn.py:1-107."""
from __future__ import annotations

from .glfsr import GLFSR

# Maximal-length feedback taps per register degree (Xilinx XAPP052), index = degree;
# same values as the reference table (pn.py:6-72), kept as one row of text per decade.
_XAPP052 = (
    "0|1|2 1|3 2|4 3|5 3|6 5|7 6|8 6 5 4|9 5",
    "10 7|11 9|12 6 4 1|13 4 3 1|14 5 3 1|15 14|16 15 13 4|17 14|18 11|19 6 2 1",
    "20 17|21 19|22 21|23 18|24 23 22 17|25 22|26 6 2 1|27 5 2 1|28 25|29 27",
    "30 6 4 1|31 28|32 22 2 1|33 20|34 27 2 1|35 33|36 25|37 5 4 3 2 1|38 6 5 1|39 35",
    "40 38 21 19|41 38|42 41 20 19|43 42 38 37|44 43 18 17|45 44 42 41|46 45 26 25|47 42|48 47 21 20|49 40",
    "50 49 24 23|51 50 36 35|52 49|53 52 38 37|54 53 18 17|55 31|56 55 35 34|57 50|58 39|59 58 38 37",
    "60 59|61 60 46 45|62 61 6 5|63 62|64 63 61 60",
)
GALOIS_LFSR_POLYS: list[list[int]] = [
    [int(tap) for tap in entry.split()] for decade in _XAPP052 for entry in decade.split("|")
]


def generate_mask(degree: int) -> int:
    """Tap mask of the degree-``degree`` register: bit (tap-1) set for every tap.

    Raises:
        KeyError: no polynomial is tabulated for ``degree`` (valid: 2..64).
    """
    if degree <= 1 or degree >= len(GALOIS_LFSR_POLYS):
        raise KeyError(f"PRBS Polynomial Not Defined for {degree}.")
    mask = 0
    for tap in GALOIS_LFSR_POLYS[degree]:
        mask |= 1 << (tap - 1)
    return mask


class PNSequence(GLFSR):
    """PN<degree> source starting from the all-ones register."""

    def __init__(self, degree: int) -> None:
        self.degree = degree
        super().__init__(generate_mask(degree), (1 << degree) - 1)

    def generate_sequence(self) -> list[int]:
        """One full period (2**degree - 1 bits) as a list of ints, generated on the GPU."""
        return self.generate((1 << self.degree) - 1).tolist()

    def generate(self, n: int, *, device: bool = False):
        from .. import _hip, device as dev
        import numpy as np

        bits, self.state = dev.lfsr_bits(self.degree, self.mask, self.state, int(n))
        return bits if device else _hip.to_host(bits).astype(np.uint8, copy=False)
