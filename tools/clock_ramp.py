"""Front-end kernel time of each of the first N blocks after process start (HIP events per block), to see how long the
chip takes to reach the clock it then holds.    python tools/clock_ramp.py [--steps 80]"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=80)
    a = ap.parse_args()
    import torch
    from waveforms_amd.link import SOQPSKLink, operating_point_warmup, soqpsk_warmup_param
    link = SOQPSKLink(10_000_000, 8, pn_degree=23, warmup=soqpsk_warmup_param(operating_point_warmup("soqpsk", 10.0)), fuse=15)
    out = []
    for k in range(a.steps):
        link.run_block(10.0, seed=1, stream_id=k, event_slot=0)
        torch.cuda.synchronize()
        out.append(round(link.stage_ms(0)["fir"], 4))
    print("front-end ms per block:", out, flush=True)


if __name__ == "__main__":
    main()
