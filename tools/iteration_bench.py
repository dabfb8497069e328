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
    # the C-ABI call alone (preallocated operands, no numpy work per call): what the boundary itself costs
    import ctypes

    from waveforms_amd import _hip

    cdet = SOQPSKTrellisDetector(length=2)
    cdet._ensure_state()
    fn, ctx, st, stream = _hip.lib().wf_viterbi4_iteration_host, _hip.ctx(), cdet._d_state_ptr, _hip.stream()
    flat = np.ascontiguousarray(rows).view(np.float64).reshape(-1, 6)
    bits, syms = np.empty(2), np.empty(2)
    pb, ps, base = bits.ctypes.data, syms.ctypes.data, flat.ctypes.data
    for k in range(200):
        fn(ctx, st, 2, 1, base + 48 * k, pb, ps, stream)
    t2 = time.perf_counter()
    for k in range(a.calls):
        fn(ctx, st, 2, 1, base + 48 * k, pb, ps, stream)
    dc = time.perf_counter() - t2
    us4 = (ctypes.c_double * 4)()
    _hip.lib().wf_viterbi4_iteration_server_timing(ctx, us4)
    t3 = time.perf_counter()
    ver = _hip.lib().wf_version
    for k in range(a.calls):
        ver()
    dv = time.perf_counter() - t3
    print(json.dumps({"calls": a.calls, "iteration_us_per_call": round(dt / a.calls * 1e6, 2),
                      "c_abi_call_us": round(dc / a.calls * 1e6, 2),
                      "server_us_last_request": dict(zip(("request_read", "cache_check", "iteration", "answer_stores_issued"), (round(v, 2) for v in us4))), "empty_ctypes_call_us": round(dv / a.calls * 1e6, 2),
                      "reference_us_per_call": 57.0, "batch_detect_us_per_call_incl_h2d_d2h": round(db / a.calls * 1e6, 4),
                      "iteration_equals_batch": bool(np.array_equal(want.astype(np.uint8), bb[:2000]))}))


if __name__ == "__main__":
    main()
