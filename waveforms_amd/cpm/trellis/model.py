"""Trellis data model — API of reference waveforms/cpm/trellis/model.py:10-293.

The five trellises are generated from their defining rules instead of being listed
branch by branch; branch order inside a column (which fixes the detector's tie-break,
reference waveforms/viterbi/algorithm.py:77-83) is the reference's: ascending start
state, then ascending branch slot.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Branch:
    inp: int
    out: int
    start: int
    end: int


@dataclass
class Trellis:
    branches: list[list[Branch]]

    def _all(self):
        return (b for column in self.branches for b in column)

    @property
    def input_cardinality(self) -> int:
        """Bits consumed per symbol (log2 of the number of distinct input labels)."""
        n_labels = len({b.inp for b in self._all()})
        if n_labels & (n_labels - 1) or n_labels == 0:
            raise ValueError("Number of unique inputs should be multiple of 2.")
        return n_labels.bit_length() - 1

    @property
    def output_cardinality(self) -> int:
        """Number of distinct output symbols."""
        return len({b.out for b in self._all()})

    @property
    def columns(self) -> int:
        """Number of time-varying trellis sections."""
        return len(self.branches)

    @property
    def branches_per_column(self) -> int:
        return len(self.branches[0])

    @property
    def states(self) -> int:
        return 1 + max(b.start for b in self._all())

    def dense_tables(self) -> tuple[np.ndarray, np.ndarray]:
        """(next u8, out i8), each [column][state][input] — the layout wf_fsm_encode takes."""
        shape = (self.columns, self.states, 1 << self.input_cardinality)
        nxt, out = np.zeros(shape, dtype=np.uint8), np.zeros(shape, dtype=np.int8)
        for c, column in enumerate(self.branches):
            for b in column:
                nxt[c, b.start, b.inp], out[c, b.start, b.inp] = b.end, b.out
        return nxt, out


def filter_branches(state: int, branches: list[Branch], attr: str = "start") -> list[Branch]:
    """Branches whose ``attr`` ("start" or "end") equals ``state``, in list order."""
    return [b for b in branches if getattr(b, attr) == state]


def forward_branches(state: int, branches: list[Branch]) -> list[Branch]:
    return filter_branches(state, branches, "start")


def reverse_branches(state: int, branches: list[Branch]) -> list[Branch]:
    return filter_branches(state, branches, "end")


def forward_map(state: int, branches: list[Branch]) -> dict[int, Branch]:
    """input label -> branch, for the branches leaving ``state``."""
    return {b.inp: b for b in forward_branches(state, branches)}


def reverse_map(state: int, branches: list[Branch]) -> dict[int, Branch]:
    """output symbol -> branch, for the branches entering ``state``."""
    return {b.out: b for b in reverse_branches(state, branches)}


class FiniteStateMachine:
    """Lookup tables over a :class:`Trellis` (what the detector indexes at run time)."""

    def __init__(self, trellis: Trellis) -> None:
        self.trellis = trellis
        self.branches_per_column = trellis.branches_per_column
        self.columns = trellis.columns
        self.states = trellis.states
        per_state = range(self.states)
        cols = trellis.branches
        self.forward_branch_mapping = [[forward_map(s, col) for s in per_state] for col in cols]
        self.reverse_branch_mapping = [[reverse_map(s, col) for s in per_state] for col in cols]
        self.forward_transitions = [
            [{b.end: b for b in forward_branches(s, col)} for s in per_state] for col in cols]
        self.reverse_transitions = [
            [{b.start: b for b in reverse_branches(s, col)} for s in per_state] for col in cols]
        self.symbols = sorted({b.out for col in cols for b in col})
        self.symbol_idx_map = {symbol: k for k, symbol in enumerate(self.symbols)}


# ---------------------------------------------------------------------------------------
# SOQPSK.  State = the last two precoder bits; even (I) sections move bit 1 of the state,
# odd (Q) sections bit 0.  Ternary output per (start state, slot) for each section:
_SOQPSK_OUT = ((0, +2, 0, -2, -2, 0, +2, 0), (0, -2, +2, 0, 0, +2, -2, 0))


def _soqpsk_section(q: int, differential: bool, offset: tuple[int, int] = (0, 0)) -> list[Branch]:
    col = []
    for slot in range(8):
        start, bit = slot >> 1, slot & 1
        end = (start & 1) + 2 * bit if q == 0 else (start & 2) + bit
        # differential encoding relabels the inputs of the branches whose start state
        # has the bit being replaced set
        flip = ((start >> 1) if q == 0 else (start & 1)) if differential else 0
        col.append(Branch(inp=bit ^ flip, out=_SOQPSK_OUT[q][slot], start=start + offset[0],
                          end=end + offset[1]))
    return col


SOQPSKTrellis8x1 = Trellis(branches=[_soqpsk_section(0, False, (0, 4)) + _soqpsk_section(1, False, (4, 0))])
SOQPSKTrellis4x2 = Trellis(branches=[_soqpsk_section(0, False), _soqpsk_section(1, False)])
SOQPSKTrellis4x2DiffEncoded = Trellis(branches=[_soqpsk_section(0, True), _soqpsk_section(1, True)])


def _memoryless(n_symbols: int) -> Trellis:
    """out = 2*inp - (n_symbols - 1); the state just remembers the last input."""
    return Trellis(branches=[[Branch(inp=i, out=2 * i - (n_symbols - 1), start=s, end=i)
                              for s in range(n_symbols) for i in range(n_symbols)]])


SimpleTrellis2 = _memoryless(2)
SimpleTrellis4 = _memoryless(4)
