"""SOQPSK ternary precoder — API of reference waveforms/cpm/soqpsk/precoder.py:5-24.

alpha_k = (-1)^(k+1) (2 a_{k-1} - 1)(a_{k-2} - a_k); the element-wise map runs on the
GPU (wf_symbol_map kind 0), the two-bit memory and the parity are carried on the host.
"""
import numpy as np
from numpy.typing import NDArray


class SOQPSKPrecoder:
    def __init__(self) -> None:
        self.i = 0
        self.mem = 0, 0

    def __call__(self, bits: NDArray[np.uint8]) -> NDArray[np.int8]:
        from waveforms_amd import _hip, device as dev

        bits = np.asarray(bits)
        if bits.size == 0:
            return np.zeros(0, dtype=np.int8)
        d_bits = _hip.to_device(bits.astype(np.uint8, copy=False))
        out = _hip.to_host(dev.symbol_map(0, d_bits, self.i, self.mem))
        tail = np.concatenate((np.asarray(self.mem, dtype=np.int8), bits.astype(np.int8)))[-2:]
        self.i = (self.i + len(bits)) % 2
        self.mem = tail
        return out
