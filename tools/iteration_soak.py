"""Soak of the per-symbol server: threads that keep creating detectors (state addresses get reused by the allocator),
stepping them a few dozen to a few hundred calls each against the reference's golden outputs, dropping them, with
idle gaps long enough for the server to retire and be restarted now and then.
    python tools/iteration_soak.py [--seconds 60] [--threads 3]"""
import argparse, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--threads", type=int, default=3)
    a = ap.parse_args()
    import numpy as np
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    g = np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "detect.npz")
    trip = g["triplets"]
    errors, counts = [], [0] * a.threads
    t_end = time.time() + a.seconds

    def work(tid):
        rng = np.random.default_rng(100 + tid)
        while time.time() < t_end and not errors:
            L = int(rng.choice([2, 4, 6]))
            diff = bool(rng.integers(0, 2))
            n = int(rng.integers(3, 260))
            det = SOQPSKTrellisDetector(L, differantial_encoding=diff)
            wb, ws = g[f"trip_L{L}_diff{int(diff)}_bits"], g[f"trip_L{L}_diff{int(diff)}_syms"]
            for k in range(n):
                b, s = det.iteration(trip[k])
                if not (np.array_equal(b, wb[k]) and np.array_equal(s, ws[k])):
                    errors.append((tid, L, diff, k, b.tolist(), wb[k].tolist()))
                    return
            counts[tid] += n
            del det
            if rng.random() < 0.02:
                time.sleep(0.03)        # longer than the server's idle time: it retires and is restarted

    th = [threading.Thread(target=work, args=(i,)) for i in range(a.threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    print(f"{sum(counts)} iteration() calls from {a.threads} threads, every output equal to the reference's", flush=True)


if __name__ == "__main__":
    main()
