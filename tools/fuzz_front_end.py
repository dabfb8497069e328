"""Random bursts through the one-kernel front end (fuse 15) against the separate kernels (fuse 7): packed rows bitwise
(pulse-truncation bank at 8 samples per symbol) or to 2e-12 (PAM bank; 10 samples per symbol), decisions and counts identical — burst lengths from one symbol to a few
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
        sps = 10 if rng.random() < 0.3 else 8            # 10: the reference example's own rate (rows of 51 columns, 48-byte rows from the separate kernels)
        off = int(rng.integers(-(sps // 2), sps // 2))
        diff = bool(rng.integers(0, 2))
        ebn0 = float(rng.choice([2.0, 6.0, 10.0]))
        kw = dict(detector=det, timing_offset=off, differential=diff)
        factored = bool(rng.integers(0, 2))      # PAM: the bank as two real filters + weights (what a link runs by default), or as three complex filters
        try:
            ref, fus = SOQPSKLink(nsym, sps, fuse=7, **kw), SOQPSKLink(nsym, sps, fuse=15, factor_bank=factored, **kw)
        except ValueError:
            continue            # burst shorter than the matched filter
        seed, sid, skip = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 22))
        for l in (ref, fus):
            l.run_block(ebn0, seed=seed, stream_id=sid, skip_bits=skip)
        lr, lf = ref.layout(), fus.layout()
        calls = lr["calls"]
        assert lf["one_kernel_front_end"] == 1, (nsym, kw)
        x = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * ref.row_bytes].view(torch.float64).cpu().numpy()
        y = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch.float64).cpu().numpy()
        if ref.row_bytes == 48:     # full rows [k][3] complex -> the detector-packed {Re z1, Im z1, odd ? Im z0 : Re z0, odd ? Re z2 : Im z2}
            z = x.reshape(calls, 3, 2)
            odd = (np.arange(calls) & 1) == 1
            x = np.stack([z[:, 1, 0], z[:, 1, 1], np.where(odd, z[:, 0, 1], z[:, 0, 0]), np.where(odd, z[:, 2, 0], z[:, 2, 1])], axis=1).reshape(-1)
        kw = dict(kw, sps=sps, factored=factored)
        if det == "PT" and sps == 8:     # (at sps 10 the separate bank kernel runs plain fma chains, the one-kernel form the shared sums: rounding apart)
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
