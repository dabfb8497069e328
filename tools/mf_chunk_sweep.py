"""Round 6: chunk length of the lane detector with the matched filters inside (ARTM link, fuse bit 7): detector alone and the
pipelined link's steady state per chunk length.   python tools/mf_chunk_sweep.py [--fuse 175] [--chunks 128,160,192] [--warmup 48]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from waveforms_amd import _hip
from waveforms_amd.link import CPMLink


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuse", type=int, default=175)
    ap.add_argument("--chunks", default="128,144,160,176,192,208,256")
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--nsym", type=int, default=10_000_000)
    ap.add_argument("--blocks", type=int, default=300)
    ap.add_argument("--waveform", default="multih")
    a = ap.parse_args()
    for ch in [int(c) for c in a.chunks.split(",")]:
        _hip.set_default_option(_hip.WF_OPT_CPM_CHUNK_CALLS, ch)
        alone = CPMLink(a.nsym, 8, waveform=a.waveform, fuse=a.fuse & ~32, warmup=a.warmup, private_ctx=True)
        best = 1e9
        for _ in range(6):
            alone.run_block(10.0, seed=1, stream_id=2, event_slot=0)
            torch.cuda.synchronize()
            best = min(best, alone.stage_ms(0)["viterbi"])
        res1 = alone.result()
        del alone
        link = CPMLink(a.nsym, 8, waveform=a.waveform, fuse=a.fuse | 32, warmup=a.warmup, private_ctx=True)
        for _ in range(5):
            link.run_block(10.0, seed=1, stream_id=2)
        torch.cuda.synchronize()
        ts = []
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(a.blocks):
                link.run_block(10.0, seed=1, stream_id=2 + i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / a.blocks * 1e3)
        print(f"chunk {ch:4d} warm-up {a.warmup}: detector alone {best:.4f} ms; pipelined steady {ts[0]:.4f} / {ts[1]:.4f} ms; counts {res1[:2]} {link.result()[:2]}", flush=True)
        del link


if __name__ == "__main__":
    main()
