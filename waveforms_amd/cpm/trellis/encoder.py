"""Bits -> symbols FSM encoder — API of reference waveforms/cpm/trellis/encoder.py:7-59.

``encode`` runs on the GPU (K2, csrc/wf_encode.hip: prefix scan over state-transition
maps); ``.i`` and ``.state`` carry across calls exactly like the reference's.
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray

from waveforms_amd.cpm.trellis.model import Trellis, forward_map


class TrellisEncoder:
    def __init__(self, trellis: Trellis) -> None:
        self.trellis = trellis
        self.input_cardinality = trellis.input_cardinality
        self.i = 0
        self.state = 0
        self.forward_branch_mapping = [
            [forward_map(s, column) for s in range(trellis.states)] for column in trellis.branches
        ]
        self._tables = trellis.dense_tables()

    def encode(self, bits: NDArray[np.uint8]) -> NDArray[np.int8]:
        """Encode 0/1 ``bits`` (``input_cardinality`` per symbol, MSB first) into symbols.

        Raises:
            ValueError: ``bits.size`` is not a multiple of the input cardinality.
        """
        from waveforms_amd import _hip, device as dev

        bits = np.asarray(bits)
        if bits.size % self.input_cardinality:
            raise ValueError("Input length must be a multiple of FSM cardinality.")
        n_sym = bits.size // self.input_cardinality
        if n_sym == 0:
            return np.zeros(0, dtype=np.int8)
        d_bits = _hip.to_device(bits.astype(np.uint8, copy=False))
        d_sym, self.state = dev.fsm_encode(*self._tables, d_bits, self.i, self.state)
        self.i += n_sym
        return _hip.to_host(d_sym)

    def __call__(self, bits: NDArray[np.uint8]) -> NDArray[np.int8]:
        return self.encode(bits)
