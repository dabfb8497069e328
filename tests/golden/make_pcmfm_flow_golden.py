"""tests/golden/pcmfm_flow.npz: what the reference's examples/pcmfm_test.py computes (not its drawing) —
PN15 bits -> SimpleTrellis2 symbols -> NRZ (*) Bessel order 4..8 frequency pulses at sps 20 (built inline there
with scipy besselap/impulse, examples/pcmfm_test.py:45-51) and the unfiltered NRZ pulse (:78) -> cpm_modulate,
h = 7/10 -> Axes.psd(NFFT=1024, Fs=20) = matplotlib.mlab.psd with its defaults (:64-69, :73-84).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tests/golden/make_pcmfm_flow_golden.py
"""
from pathlib import Path

import numpy as np
from matplotlib import mlab
from scipy.signal import besselap, impulse

import waveforms
from waveforms.cpm.helpers import normalize_cpm_filter
from waveforms.cpm.modulate import cpm_modulate
from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER
from waveforms.cpm.trellis.encoder import TrellisEncoder
from waveforms.cpm.trellis.model import SimpleTrellis2
from waveforms.glfsr import PNSequence

assert "/root/reference" in waveforms.__file__
OUT = Path(__file__).resolve().parent
sps, fft_size, length = 20, 2**10, 3
bit_array = np.unpackbits(np.packbits(PNSequence(15).generate_sequence()))
symbols = TrellisEncoder(SimpleTrellis2)(bit_array)
mod_index = PCMFM_NUMER / PCMFM_DENOM
out = {"nbits": np.array([bit_array.size]), "symbols_sum": np.array([int(symbols.astype(np.int64).sum())])}
pulses = {}
for order in (4, 5, 6, 7, 8):
    t, y = impulse(besselap(order, norm="mag"), T=np.linspace(0, length * 2 / 0.7, num=(length - 1) * sps + 1))
    pulses[f"o{order}"] = normalize_cpm_filter(sps, np.convolve(y, np.ones(sps)))
pulses["nrz"] = normalize_cpm_filter(sps, np.ones(sps))
for name, pulse in pulses.items():
    _t, sig = cpm_modulate(symbols=symbols, mod_index=mod_index, pulse_filter=pulse, sps=sps)
    pxx, freqs = mlab.psd(sig, NFFT=fft_size, Fs=sps)
    out[f"pulse_{name}"], out[f"pxx_{name}"] = pulse, pxx
    out[f"sig_head_{name}"], out[f"sig_sum_{name}"] = sig[:32], np.array([sig.sum()])
out["freqs"] = freqs
np.savez_compressed(OUT / "pcmfm_flow.npz", **out)
print("wrote pcmfm_flow.npz", {k: v.shape for k, v in out.items()})
