"""Time ONE stage kernel at bench size with HIP events (torch stream = launch stream).

    python tools/stage_bench.py <stage> [--nsym 10000000] [--reps 10] [--lib path/to/libwfhip.so]

Stages: fir phase awgn mfbank viterbi modulate.  Prints ms per launch and algorithmic GB/s.
Used for A/B ablations of kernel variants (see tools/ablate.sh).
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage")
    ap.add_argument("--nsym", type=int, default=10_000_000)
    ap.add_argument("--sps", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    import torch

    from waveforms_amd import _hip

    if a.lib:
        _hip._LIB_PATH = Path(a.lib).resolve()
    from waveforms_amd import device as dev
    from waveforms_amd.cpm.soqpsk import freq_pulse_soqpsk_tg
    from waveforms_amd.filters.matched import pt_matched_filter_taps

    n, sps = a.nsym, a.sps
    npts = (n + 1) * sps
    pulse = freq_pulse_soqpsk_tg(sps)
    g = torch.Generator(device="cuda").manual_seed(1)
    sym = (torch.randint(-1, 2, (n,), device="cuda", generator=g, dtype=torch.int8) * 2).contiguous()
    h = _hip.to_device(np.array([0.25]))
    d_pulse = _hip.to_device(pulse)
    taps = _hip.to_device(pt_matched_filter_taps(pulse, 0.25, sps))
    freq = dev.upsample_fir(sym, h, d_pulse, sps)
    sig = dev.phase_cexp(freq, sps, np.pi / 4)
    first, ncols = dev.decimation(npts, sps, 2, -1)
    rows = dev.mf_bank(sig, taps, first, sps, ncols)
    bytes_per_sym = {"fir": 1 + 8 * sps, "phase": 24 * sps, "awgn": 32 * sps, "mfbank": 16 * sps + 48,
                     "viterbi": 50, "chan": 16 * sps + 48, "modulate": 1 + 16 * sps, "modfused": 1 + 16 * sps}[a.stage]
    out = torch.empty_like(sig)

    def run():
        if a.stage == "fir":
            dev.upsample_fir(sym, h, d_pulse, sps)
        elif a.stage == "phase":
            dev.phase_cexp(freq, sps, np.pi / 4)
        elif a.stage == "awgn":
            dev.awgn(sig, npts, 0.6, 1, 0, 0, np.exp(-1j * np.pi / 4), out)
        elif a.stage == "mfbank":
            dev.mf_bank(sig, taps, first, sps, ncols)
        elif a.stage == "chan":
            dev.awgn_mf_bank(sig, taps, first, sps, ncols, 0.6, 1, 0, 0, np.exp(-1j * np.pi / 4))
        elif a.stage == "viterbi":
            dev.viterbi_detect(rows)
        elif a.stage == "modulate":
            dev.cpm_modulate(sym, h, d_pulse, sps, np.pi / 4, fused=False)
        elif a.stage == "modfused":
            dev.cpm_modulate(sym, h, d_pulse, sps, np.pi / 4, fused=True)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
    ev[0].record()
    for k in range(a.reps):
        run()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(a.reps))
    med = ms[len(ms) // 2]
    _hip.device_check()
    print(f"{a.stage:8s} median {med:8.4f} ms  min {ms[0]:8.4f}  {bytes_per_sym * n / med / 1e6:8.1f} GB/s algorithmic")


if __name__ == "__main__":
    main()
