"""Round 6: the pipelined SOQPSK link with each block's prologue on its own stream beside the previous front end, the front end on a
CU-masked stream: steady state per WF_OPT_PIPE_RESERVE_CUS value, from the NULL stream and from a stream of the caller's.
    python tools/pipe_ahead_probe.py [--reserve=0,-1,8,4,16] [--detector PT] [--sps 8] [--blocks 600]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from waveforms_amd import _hip
from waveforms_amd.link import SOQPSKLink, operating_point_warmup, soqpsk_warmup_param


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", default="0,-1,8,4,16,32")
    ap.add_argument("--detector", default="PT")
    ap.add_argument("--sps", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=600)
    ap.add_argument("--nsym", type=int, default=10_000_000)
    a = ap.parse_args()
    wu = soqpsk_warmup_param(operating_point_warmup("soqpsk", 10.0))
    for own_stream in (False, True):
        for rsv in [int(v) for v in a.reserve.split(",")]:
            _hip.set_default_option(_hip.WF_OPT_PIPE_RESERVE_CUS, rsv)
            ctxm = torch.cuda.stream(torch.cuda.Stream()) if own_stream else torch.cuda.stream(torch.cuda.current_stream())
            with ctxm:
                link = SOQPSKLink(a.nsym, a.sps, detector=a.detector, fuse=47, private_ctx=True, warmup=wu)
                for i in range(100):
                    link.run_block(10.0, seed=1, stream_id=i)
                torch.cuda.synchronize()
                link.reset_counts()
                ts = []
                for rep in range(3):
                    t0 = time.perf_counter()
                    for i in range(a.blocks):
                        link.run_block(10.0, seed=1, stream_id=i, skip_bits=(i % 4096) * a.nsym)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / a.blocks * 1e3)
                res = link.result()
                print(f"{'own stream ' if own_stream else 'NULL stream'} reserve {rsv:3d}: steady {ts[0]:.4f} / {ts[1]:.4f} / {ts[2]:.4f} ms per block; bit errors {res[1]} of {res[2]}", flush=True)
                del link
    _hip.set_default_option(_hip.WF_OPT_PIPE_RESERVE_CUS, 0)


if __name__ == "__main__":
    main()
