"""How short can the SOQPSK detector's chunk warm-up be?  For each warm-up length: chunks whose
start metrics were NOT bitwise those of the sequential detector (the launch's own proof), over
Eb/N0 0 .. 12 dB, and the detector stage time at 1e7 symbols.

    python tools/warmup_scan.py [--symbols-per-point 1e8]
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols-per-point", type=float, default=1e8)
    a = ap.parse_args()
    import torch

    from waveforms_amd import device as dev
    from waveforms_amd.link import SOQPSKLink

    nsym = 10_000_000
    blocks = max(1, int(a.symbols_per_point / nsym))
    out = []
    for w in (11, 15, 19, 23, 31, 47):
        link = SOQPSKLink(nsym, 8, warmup=w)
        row = {"warmup_rows": w + 1, "unmerged_by_ebn0": {}, "chunks_per_point": None}
        for e in range(0, 13, 2):
            dev.viterbi_unmerged(reset=True, ctx=link._ctx)
            for b in range(blocks):
                link.run_block(float(e), seed=3, stream_id=(e << 20) | b, skip_bits=b * nsym)
            torch.cuda.synchronize()
            row["unmerged_by_ebn0"][e] = dev.viterbi_unmerged(reset=True, ctx=link._ctx)
        link.run_block(10.0, event_slot=0)
        row["viterbi_ms"] = round(link.stage_ms(0)["viterbi"], 4)
        row["chunks_per_point"] = blocks * (nsym // 160)
        out.append(row)
        print(json.dumps(row), flush=True)
        del link
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/r02_warmup_scan.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
