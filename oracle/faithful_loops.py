"""The reference's hot loops as the reference runs them: interpreted, one symbol (or one
sample) per trip.

TEST INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``).  ``numpy_ref.py`` /
``wf_oracle.c`` restate the same arithmetic in vectorised NumPy and C; this module keeps
the reference's *execution form* — Python loops over symbols and samples — so that
``bench.py``'s ``cpu_baseline`` can put the cost of the reference's own structure beside the
compiled port ("faithful-loop variant").  Parity status: PINNED — every
function here is checked against the reference-generated fixtures
(``tests/test_oracle_golden.py::test_faithful_loops_*``).

Citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import numpy as np

from .numpy_ref import TRELLISES, pt_taps

__all__ = ["lfsr_bits_loop", "fsm_encode_loop", "frequency_modulate_loop", "cpm_modulate_loop",
           "DetectorLoop", "detection_run_loop"]


def lfsr_bits_loop(mask: int, state: int, n: int):
    """waveforms/glfsr/glfsr.py:15-19 driven n times (pn.py:107): one interpreted step per bit."""
    out = np.empty(n, dtype=np.uint8)
    for k in range(n):
        low = state & 1
        state >>= 1
        if low:
            state ^= mask
        out[k] = low
    return out, state


def fsm_encode_loop(trellis: str, bits, i: int = 0, state: int = 0):
    """waveforms/cpm/trellis/encoder.py:33-46: per symbol, gather `cardinality` bits MSB first,
    look the branch up by (column, state, input), emit its output, move to its end state."""
    cols = TRELLISES[trellis]
    table = [{(st, inp): (out, end) for inp, out, st, end in col} for col in cols]
    card = max(1, (len({b[0] for b in cols[0]}) - 1).bit_length())
    if len(bits) % card:
        raise ValueError("Input length must be a multiple of FSM cardinality.")
    nsym = len(bits) // card
    out = np.empty(nsym, dtype=np.int8)
    for n in range(nsym):
        word = 0
        for b in bits[n * card:(n + 1) * card]:
            word = (word << 1) | int(b)
        sym, state = table[i % len(cols)][(state, word)]
        out[n] = sym
        i += 1
    return out, i, state


def frequency_modulate_loop(freq_pulses, sps: int, initial_phase: float = 0.0):
    """waveforms/cpm/modulate.py:48-54: revs = (revs + sample) % sps per sample, then exp(j.)."""
    phase = np.empty(len(freq_pulses), dtype=np.float64)
    scale = 2 * np.pi / sps
    revs = 0
    for k, f in enumerate(freq_pulses):
        revs = (revs + f) % sps
        phase[k] = revs * scale + initial_phase
    return np.exp(1j * phase)


def cpm_modulate_loop(symbols, mod_index, g, sps: int):
    """waveforms/cpm/modulate.py:75-101 with the sample loop above."""
    h = np.atleast_1d(np.asarray(mod_index, dtype=np.float64))
    n = len(symbols)
    stuffed = np.zeros((n + 1) * sps, dtype=np.float64)
    stuffed[sps:-1:sps] = np.asarray(symbols) * h[np.arange(n) % h.size]
    return frequency_modulate_loop(np.convolve(stuffed, g, mode="same"), sps, np.pi / 4)


class DetectorLoop:
    """waveforms/viterbi/algorithm.py:18-101, one interpreted call per symbol: history shift
    (:57-63), min-normalised entering metrics (:65-67), add-compare-select with strict '<' over
    the branches in list order (:69-87), traceback from the first arg-min through
    reverse_transitions (:90-98, cpm/trellis/model.py:171-174)."""

    ROT = (1j, -1, 1, -1j)   # state_exp_term, algorithm.py:30

    def __init__(self, length: int = 2, differential: bool = True):
        self.cols = TRELLISES["SOQPSKTrellis4x2DiffEncoded" if differential else "SOQPSKTrellis4x2"]
        self.length, self.i = length, 0
        nbr = len(self.cols[0])
        self.inc = np.zeros((nbr, length))
        self.metric = np.zeros((4, length))
        self.pred = np.zeros((4, length), dtype=np.uint8)
        self.row_of = {s: k for k, s in enumerate(sorted({b[1] for c in self.cols for b in c}))}
        self.back = [{(end, st): (inp, out) for inp, out, st, end in col} for col in self.cols]

    def iteration(self, mf3):
        L, ncol = self.length, len(self.cols)
        self.inc = np.roll(self.inc, -1, axis=1)
        self.inc[:, -1] = [(self.ROT[st] * mf3[self.row_of[out]]).real for _inp, out, st, _end in self.cols[self.i % ncol]]
        entering = self.metric[:, 0] - self.metric[:, 0].min()
        self.metric[:, :] = 0
        self.metric[:, -1] = entering
        self.pred[:, :] = 0
        for j in range(L):
            col = self.cols[(self.i + j - 1) % ncol]
            for end_state in range(4):
                best, who = np.inf, 0
                for (_inp, _out, st, end), d in zip(col, self.inc[:, j]):
                    if end != end_state:
                        continue
                    cand = self.metric[st, (j - 1) % L] + d
                    if cand < best:
                        best, who = cand, st
                self.metric[end_state, j] = best
                self.pred[end_state, j] = who
        bits, syms = np.zeros(L), np.zeros(L)
        state = int(np.argmin(self.metric[:, -1]))
        for j in reversed(range(L)):
            prev = int(self.pred[state, j])
            bits[j], syms[j] = self.back[(self.i + j - 1) % ncol][(state, prev)]
            state = prev
        self.i += 1
        return bits, syms


def detection_run_loop(bits, g, h, sps, sigma, rng, length=2, timing_offset=-1):
    """The per-waveform body of examples/soqpsk_detection.py:78-216 (PT detector) in the
    reference's execution form: interpreted encoder, per-sample modulator loop, three full-rate
    np.convolve filters (:141-156), the sample-rate decimation loop with one detector call per
    kept sample (:189-198), error count (:200-209)."""
    symbols, _, _ = fsm_encode_loop("SOQPSKTrellis4x2DiffEncoded", bits)
    sig = cpm_modulate_loop(symbols, h, g, sps)
    noise = rng.normal(0, sigma, size=(sig.size, 2)).view(np.complex128)[:, 0]   # waveforms/noise.py:24-32
    r = sig * np.exp(-1j * np.pi / 4) + noise
    mf = np.array([np.convolve(r, t, mode="same") for t in pt_taps(g, h, sps)])
    det = DetectorLoop(length, True)
    dbits, dsyms = [], []
    for n in range(r.size - length * sps):
        if (n + timing_offset) % sps:
            continue
        b, s = det.iteration(mf[:, n])
        dbits.append(b[0])
        dsyms.append(s[0])
    dsyms = np.asarray(dsyms[length:], dtype=np.int8)
    dbits = np.asarray(dbits[length:], dtype=np.uint8)
    m = min(dsyms.size, symbols.size)
    return dict(symbols=symbols, det_bits=dbits, det_syms=dsyms, compared=m,
                sym_errors=int(np.count_nonzero(dsyms[:m] - symbols[:m])),
                bit_errors=int(np.count_nonzero(dbits[:m] - np.asarray(bits[:m], dtype=np.uint8))))
