"""PCM/FM antipodal mapper — API of reference waveforms/cpm/pcmfm/precoder.py:5-15."""
import numpy as np
from numpy.typing import NDArray


class PCMFMSymbolMapper:
    def __call__(self, bits: NDArray[np.uint8]) -> NDArray[np.int8]:
        """0/1 -> -1/+1 (GPU, wf_symbol_map kind 2)."""
        from waveforms_amd import _hip, device as dev

        bits = np.asarray(bits)
        if bits.size == 0:
            return np.zeros(0, dtype=np.int8)
        return _hip.to_host(dev.symbol_map(2, _hip.to_device(bits.astype(np.uint8, copy=False))))
