"""Random bursts through the one-kernel front end (fuse 15) against the separate kernels (fuse 7): packed rows bitwise
(pulse-truncation bank) or to 2e-12 (PAM bank), decisions and counts identical — burst lengths from one symbol to a few
million (run partition, single-tile tail, tile edges, ragged ends), every decimation phase, both precoder forms.
    python tools/fuzz_front_end.py [--seconds 60] [--seed 1]"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import numpy as np
    import torch
    from waveforms_amd.link import SOQPSKLink
    rng = np.random.default_rng(a.seed)
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        kind = rng.integers(0, 4)
        nsym = int([rng.integers(40, 3000), rng.integers(3000, 200_000), rng.integers(200_000, 3_000_000), rng.integers(1000, 1100) * 1024 + rng.integers(-3, 4)][kind])
        det = "PAM" if rng.random() < 0.4 else "PT"
        off = int(rng.integers(-4, 4))
        diff = bool(rng.integers(0, 2))
        ebn0 = float(rng.choice([2.0, 6.0, 10.0]))
        kw = dict(detector=det, timing_offset=off, differential=diff)
        try:
            ref, fus = SOQPSKLink(nsym, 8, fuse=7, **kw), SOQPSKLink(nsym, 8, fuse=15, **kw)
        except ValueError:
            continue            # burst shorter than the matched filter
        seed, sid, skip = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 22))
        for l in (ref, fus):
            l.run_block(ebn0, seed=seed, stream_id=sid, skip_bits=skip)
        lr, lf = ref.layout(), fus.layout()
        calls = lr["calls"]
        assert lf["one_kernel_front_end"] == 1, (nsym, kw)
        x = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch.float64).cpu().numpy()
        y = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch.float64).cpu().numpy()
        if det == "PT":
            assert np.array_equal(x.view(np.int64), y.view(np.int64)), ("rows", nsym, kw)
        else:
            assert np.abs(x - y).max() <= 2e-12, ("rows", nsym, kw, float(np.abs(x - y).max()))
        for key in ("off_bits", "off_syms"):
            assert np.array_equal(ref.workspace[lr[key]:lr[key] + calls].cpu().numpy(), fus.workspace[lf[key]:lf[key] + calls].cpu().numpy()), (key, nsym, kw)
        assert ref.result() == fus.result(), (nsym, kw)
        n += 1
    print(f"{n} random bursts: one-kernel front end == separate kernels", flush=True)


if __name__ == "__main__":
    main()
