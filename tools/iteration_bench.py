"""Per-symbol drop-in path: microseconds per SOQPSKTrellisDetector.iteration() call (the loop of the
reference's examples/soqpsk_detection.py:189-198), next to the batch path on the same rows.

    python tools/iteration_bench.py [--calls 20000]

The reference's own iteration() costs ~57 us per call (one Xeon core)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=20000)
    a = ap.parse_args()
    import torch

    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    rng = np.random.default_rng(1)
    rows = rng.standard_normal((a.calls, 3)) + 1j * rng.standard_normal((a.calls, 3))
    rows[:, 1] += 1.5
    det = SOQPSKTrellisDetector(length=2)
    for z in rows[:200]:
        det.iteration(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = [det.iteration(z)[0][0] for z in rows]
    dt = time.perf_counter() - t0
    bdet = SOQPSKTrellisDetector(length=2)
    bdet.detect(rows[:1000])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    bdet2 = SOQPSKTrellisDetector(length=2)
    bb, _ = bdet2.detect(rows)
    db = time.perf_counter() - t1
    fresh = SOQPSKTrellisDetector(length=2)
    want = np.array([fresh.iteration(z)[0][0] for z in rows[:2000]])
    print(json.dumps({"calls": a.calls, "iteration_us_per_call": round(dt / a.calls * 1e6, 2),
                      "reference_us_per_call": 57.0, "batch_detect_us_per_call_incl_h2d_d2h": round(db / a.calls * 1e6, 4),
                      "iteration_equals_batch": bool(np.array_equal(want.astype(np.uint8), bb[:2000]))}))


if __name__ == "__main__":
    main()
