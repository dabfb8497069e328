"""Round 6: the ARTM link with the matched filters inside the detector (fuse bit 7) against the paired one-kernel front end
(rows in HBM): decisions bit for bit, error counts, stage times.   python tools/mf_form_probe.py [nsym]"""
import ctypes
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from waveforms_amd import _hip, device as dev
from waveforms_amd.link import CPMLink


def form(link):
    info = (ctypes.c_int * 4)()
    _hip.check(_hip.lib().wf_cpm_link_form(link._ctx, ctypes.byref(link.cfg), info))
    return list(info)


def timed(link, ebn0, reps=6):
    ts = []
    for r in range(reps):
        link.reset_counts()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        link.run_block(ebn0, seed=1, stream_id=2, event_slot=0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), link.stage_ms(0)


def main():
    nsym = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    states = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from waveforms_amd.viterbi import cpm as _cpm
    spec = {16: None, 64: _cpm.ARTM_64, 256: _cpm.ARTM_256}[states]
    for ebn0, warm in ((10.0, 48), (10.0, 0), (2.0, 16), (None, 48)):
        a = CPMLink(nsym, 8, waveform="multih", spec=spec, fuse=10, warmup=warm, private_ctx=True)
        b = CPMLink(nsym, 8, waveform="multih", spec=spec, fuse=10 | 128, warmup=warm, private_ctx=True)
        fa, fb = form(a), form(b)
        ta, sa = timed(a, ebn0)
        tb, sb = timed(b, ebn0)
        ra, rb = a.result(), b.result()
        la, lb = a.layout(), b.layout()
        da = a.workspace[la["off_decisions"]:la["off_decisions"] + la["calls"]].cpu().numpy()
        db = b.workspace[lb["off_decisions"]:lb["off_decisions"] + lb["calls"]].cpu().numpy()
        rep_a = (dev.viterbi_repaired(reset=True, ctx=a._ctx), dev.viterbi_cascaded(reset=True, ctx=a._ctx))
        rep_b = (dev.viterbi_repaired(reset=True, ctx=b._ctx), dev.viterbi_cascaded(reset=True, ctx=b._ctx))
        print(f"ebn0 {ebn0} warmup {warm}: forms {fa} | {fb}; results {ra} | {rb}; decisions equal {np.array_equal(da, db)} "
              f"(differ at {int(np.count_nonzero(da != db))}); repairs {rep_a} | {rep_b}")
        print(f"   rows form   {ta:.3f} ms  stages {{{', '.join(f'{k}: {v:.3f}' for k, v in sa.items())}}}")
        print(f"   samples form {tb:.3f} ms  stages {{{', '.join(f'{k}: {v:.3f}' for k, v in sb.items())}}}")
        del a, b
    # pipelined (what bench.py runs): steady state over 200 blocks
    for fuse in (47, 47 | 128):
        link = CPMLink(nsym, 8, waveform="multih", spec=spec, fuse=fuse, warmup=48 if states == 16 else 0, private_ctx=True)
        for _ in range(5):
            link.run_block(10.0, seed=1, stream_id=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 200 if states == 16 else 40
        for i in range(nb):
            link.run_block(10.0, seed=1, stream_id=2 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb * 1e3
        print(f"pipelined fuse {fuse}: form {form(link)} steady {dt:.4f} ms per block, result {link.result()}")
        del link


if __name__ == "__main__":
    main()
