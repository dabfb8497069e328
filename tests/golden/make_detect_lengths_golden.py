"""tests/golden/detect_lengths.npz: what the reference's SOQPSKTrellisDetector.iteration returns for window lengths
outside the even 2 .. 16 range (waveforms/viterbi/algorithm.py:19-42 takes any `length`): odd lengths, where the
increments of a row and the stage that consumes them come from different trellis sections (:57-63 against :69-87),
length 1, where the stage updates its metrics in place, and long windows up to 64.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_detect_lengths_golden.py

Stored: the inputs (the `triplets` of detect.npz, regenerated from the same seed) and element [0] of both arrays every
call returned (what a caller keeps, examples/soqpsk_detection.py:196-198); for the short lengths the full arrays too.
No reference source text is stored.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent

import waveforms  # noqa: E402  (must resolve to /root/reference)

assert "/root/reference" in waveforms.__file__, waveforms.__file__

from waveforms.viterbi.algorithm import SOQPSKTrellisDetector  # noqa: E402

LENGTHS = (1, 3, 5, 7, 9, 17, 18, 24, 33, 64)
FULL = (1, 3, 5)


def main():
    rng = np.random.Generator(np.random.PCG64(99))
    trip = (rng.normal(size=(4000, 3)) + 1j * rng.normal(size=(4000, 3)))
    trip[100:110] = 0
    trip[200:210] = 1 + 1j
    d = {"triplets": trip, "lengths": np.array(LENGTHS)}
    for length in LENGTHS:
        n = 4000 if length <= 9 else 1500
        for diff in (True, False):
            det = SOQPSKTrellisDetector(length=length, differantial_encoding=diff)
            fb, fs = [], []
            for z in trip[:n]:
                b, s = det.iteration(z)
                fb.append(np.array(b, dtype=np.float64))
                fs.append(np.array(s, dtype=np.float64))
            fb, fs = np.array(fb), np.array(fs)
            d[f"L{length}_diff{int(diff)}_bits0"] = fb[:, 0].astype(np.uint8)
            d[f"L{length}_diff{int(diff)}_syms0"] = fs[:, 0].astype(np.int8)
            if length in FULL:
                d[f"L{length}_diff{int(diff)}_bits"] = fb
                d[f"L{length}_diff{int(diff)}_syms"] = fs
    np.savez_compressed(OUT / "detect_lengths.npz", **d)
    print({k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
