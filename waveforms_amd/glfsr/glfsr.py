"""Galois LFSR — API of reference waveforms/glfsr/glfsr.py:1-19.

``next_bit`` advances the register by one step on the host (it is a stateful
one-bit-per-call accessor: three integer operations, no array work).  Bulk generation
(``generate``) runs the leap-ahead HIP kernel K1 (csrc/wf_lfsr.hip) and keeps the
register in sync, so single steps and bulk calls can be mixed freely.
"""
from __future__ import annotations

import numpy as np


class GLFSR:
    def __init__(self, mask: int, state: int) -> None:
        self.mask = mask
        self.state = state

    def next_bit(self) -> int:
        """Next bit of the sequence (right shift, conditional XOR with the tap mask)."""
        out = self.state & 1
        shifted = self.state >> 1
        self.state = shifted ^ self.mask if out else shifted
        return out

    def generate(self, n: int, *, device: bool = False):
        """The next ``n`` bits as uint8 (0/1), computed on the GPU.

        Returns a host ndarray, or the device tensor when ``device=True``.
        """
        from .. import _hip, device as dev

        degree = max(2, int(self.mask).bit_length(), int(self.state).bit_length())
        if degree > 64:
            raise KeyError(f"PRBS Polynomial Not Defined for {degree}.")
        bits, self.state = dev.lfsr_bits(degree, self.mask, self.state, int(n))
        return bits if device else _hip.to_host(bits).astype(np.uint8, copy=False)
