"""tests/golden/detector_state.npz: the public state arrays of the reference's SOQPSKTrellisDetector
(waveforms/viterbi/algorithm.py:25-42: bi_history float64[8, length], metrics float64[4, length], path uint8[4, length]) as
they stand before any call and after k calls of iteration() (:57-88 rewrites them on every call), for window lengths 1, 2, 5
and 16, both trellises.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_detector_state_golden.py

Stored: the inputs (seeded triplets) and the three arrays at the listed call counts.  No reference source text is stored.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent

import waveforms  # noqa: E402  (must resolve to /root/reference)

assert "/root/reference" in waveforms.__file__, waveforms.__file__

from waveforms.viterbi.algorithm import SOQPSKTrellisDetector  # noqa: E402

LENGTHS = (1, 2, 5, 16)
AT = (0, 1, 2, 3, 7, 40)


def main():
    rng = np.random.Generator(np.random.PCG64(2026))
    trip = rng.normal(size=(40, 3)) + 1j * rng.normal(size=(40, 3))
    d = {"triplets": trip, "lengths": np.array(LENGTHS), "at": np.array(AT)}
    for length in LENGTHS:
        for diff in (True, False):
            det = SOQPSKTrellisDetector(length=length, differantial_encoding=diff)
            for k in range(41):
                if k in AT:
                    tag = f"L{length}_diff{int(diff)}_k{k}"
                    d[tag + "_bi_history"] = np.array(det.bi_history, dtype=np.float64)
                    d[tag + "_metrics"] = np.array(det.metrics, dtype=np.float64)
                    d[tag + "_path"] = np.array(det.path, dtype=np.uint8)
                if k < 40:
                    det.iteration(trip[k])
    np.savez_compressed(OUT / "detector_state.npz", **d)
    print(len(d), "arrays")


if __name__ == "__main__":
    main()
