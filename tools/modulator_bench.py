"""Time the fused modulator on the BASELINE modulator configurations (device-resident in and out):

    python tools/modulator_bench.py        # SOQPSK-TG, ARTM multi-h (config 3), PCM/FM at 1e7 symbols
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from waveforms_amd import _hip, device as dev                      # noqa: E402
from waveforms_amd.cpm.multih import freq_pulse_multih_irig     # noqa: E402
from waveforms_amd.cpm.pcmfm import freq_pulse_pcmfm               # noqa: E402
from waveforms_amd.cpm.soqpsk import freq_pulse_soqpsk_tg          # noqa: E402

n, sps = 10_000_000, 8
g = torch.Generator(device="cuda").manual_seed(1)
cases = {
    "SOQPSK-TG (h 1/4, 65 taps, ternary)": ((torch.randint(-1, 2, (n,), device="cuda", generator=g, dtype=torch.int8) * 2), [0.25], freq_pulse_soqpsk_tg(sps)),
    "ARTM multi-h (h 4/16 5/16, 3RC, quaternary)": ((torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int8) * 2 - 3), [4 / 16, 5 / 16], freq_pulse_multih_irig(sps)),
    "PCM/FM (h 0.7, binary)": ((torch.randint(0, 2, (n,), device="cuda", generator=g, dtype=torch.int8) * 2 - 1), [0.7], freq_pulse_pcmfm(sps)),
}
for name, (sym, h, pulse) in cases.items():
    d_h, d_p = _hip.to_device(np.asarray(h, dtype=np.float64)), _hip.to_device(np.asarray(pulse, dtype=np.float64))
    sym = sym.contiguous()
    for _ in range(3):
        out = dev.cpm_modulate(sym, d_h, d_p, sps, np.pi / 4, fused=True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for k in range(10):
        out = dev.cpm_modulate(sym, d_h, d_p, sps, np.pi / 4, fused=True)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(10))[5]
    print(f"{name:46s} {pulse.size:3d} taps  {ms:7.3f} ms  {n / ms / 1e3:8.1f} Msym/s  {(1 + 16 * sps) * n / ms / 1e6:7.0f} GB/s")
