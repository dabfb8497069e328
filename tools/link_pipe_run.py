"""A pipelined CPM link for N blocks and nothing else (the program a rocprofv3 kernel trace wraps: tools/link_timeline.sh).
    python3 tools/link_pipe_run.py [--waveform multih] [--fuse 175] [--blocks 40] [--opt cpm_chunk_calls=160]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from waveforms_amd import _hip
from waveforms_amd.link import CPMLink, operating_point_warmup


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waveform", default="multih")
    ap.add_argument("--fuse", type=int, default=175)
    ap.add_argument("--blocks", type=int, default=40)
    ap.add_argument("--nsym", type=int, default=10_000_000)
    ap.add_argument("--ebn0", type=float, default=10.0)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    _hip.apply_option_args(a.opt)
    if a.waveform == "soqpsk":
        from waveforms_amd.link import SOQPSKLink, soqpsk_warmup_param

        link = SOQPSKLink(a.nsym, 8, detector="PT", fuse=a.fuse, warmup=soqpsk_warmup_param(operating_point_warmup("soqpsk", a.ebn0)))
    else:
        link = CPMLink(a.nsym, 8, waveform=a.waveform, fuse=a.fuse, warmup=operating_point_warmup(a.waveform, a.ebn0))
    for _ in range(100):
        link.run_block(a.ebn0, seed=1, stream_id=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.blocks):
        link.run_block(a.ebn0, seed=1, stream_id=2 + i)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / a.blocks * 1e3:.4f} ms per block; {link.result()}")


if __name__ == "__main__":
    main()
