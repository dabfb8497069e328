"""ARTM multi-h quaternary mapper — API of reference waveforms/cpm/multih/precoder.py:5-23."""
import numpy as np
from numpy.typing import NDArray


class MultiHSymbolMapper:
    def __init__(self) -> None:
        self.i = 0

    def __call__(self, bits: NDArray[np.uint8]) -> NDArray[np.int8]:
        """Bit pairs (MSB first) -> {-3, -1, +1, +3}.

        Raises:
            ValueError: odd number of bits.
        """
        from waveforms_amd import _hip, device as dev

        bits = np.asarray(bits)
        if bits.size % 2:
            raise ValueError("Odd length bit array passed into quaternary mapper.")
        self.i = (self.i + len(bits)) % 2
        if bits.size == 0:
            return np.zeros(0, dtype=np.int8)
        return _hip.to_host(dev.symbol_map(1, _hip.to_device(bits.astype(np.uint8, copy=False)), self.i))
