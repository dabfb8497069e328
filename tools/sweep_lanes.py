"""The 13-point sweep by lanes (trial blocks in flight on separate streams) and link fuse value: seconds of the second pass.
    python tools/sweep_lanes.py [--waveform soqpsk|multih|pcmfm]"""
import argparse, functools, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from waveforms_amd import bert

ap = argparse.ArgumentParser()
ap.add_argument("--waveform", default="soqpsk")
ap.add_argument("--block", type=int, default=10_000_000)
ap.add_argument("--detector", default="PT")
a = ap.parse_args()
cpm = a.waveform != "soqpsk"
plan = bert.SweepPlan(ebn0_db=list(range(13)), blocks_per_point=max(1, round(1e8 / a.block)), nsym=a.block, waveform=a.waveform, detector=a.detector)
ref = None
for streams, fuse in ((3, None), (1, 42 if cpm else 47), (2, 42 if cpm else 47), (3, 42 if cpm else 47), (2, None), (1, None)):
    runner = functools.partial(bert.gpu_block_runner, streams=streams, fuse=fuse)
    secs = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = bert.ber_sweep(plan, rank=0, world=1, runner=runner, reduce=False)
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
    ref = out if ref is None else ref
    print(a.waveform, a.detector, "block", a.block, "streams", streams, "fuse", fuse, [round(s, 4) for s in secs], "same counts" if (out == ref).all() else "COUNTS DIFFER", flush=True)
