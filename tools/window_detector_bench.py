"""Throughput of the batch SOQPSK detector per traceback window length (device-resident rows).
    python tools/window_detector_bench.py [--n 10000000] [--lengths 2,4,6,8,16]"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--lengths", default="2,4,6,8,16")
    a = ap.parse_args()
    import torch

    from waveforms_amd import device as dev

    rows = torch.randn((a.n, 3, 2), dtype=torch.float64, device="cuda")
    out = {}
    for L in [int(x) for x in a.lengths.split(",")]:
        run = (lambda: dev.viterbi_detect(rows)) if L == 2 else (lambda: dev.viterbi_detect_window(rows, L))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[f"L{L}"] = {"ms_per_1e7_rows": round(ms * 1e7 / a.n, 4), "Msym_per_s": round(a.n / ms / 1e3, 1),
                        "unproven_chunks": dev.viterbi_unmerged(reset=True)}
    print(json.dumps({"rows": a.n, "row_bytes": 48, "noise_rows": "unit Gaussian (no signal: the slowest merging there is)", **out}))


if __name__ == "__main__":
    main()
