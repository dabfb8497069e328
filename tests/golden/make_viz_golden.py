"""tests/golden/viz.npz: the arrays behind the reference's plots, from the third-party routines the
reference calls (matplotlib.mlab.psd behind Axes.psd, numpy angle/unwrap) on a signal produced
by the REFERENCE modulator.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 /root/repo/tests/golden/make_viz_golden.py
"""
from pathlib import Path

import numpy as np
from matplotlib import mlab

import waveforms
from waveforms.cpm.modulate import cpm_modulate
from waveforms.cpm.soqpsk import freq_pulse_soqpsk_tg
from waveforms.cpm.trellis.encoder import TrellisEncoder
from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded
from waveforms.glfsr import PNSequence

assert "/root/reference" in waveforms.__file__
OUT = Path(__file__).resolve().parent
bits = np.array(PNSequence(11).generate_sequence(), dtype=np.uint8)[:2046]
sym = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)(bits)
t, sig = cpm_modulate(sym, 0.25, freq_pulse_soqpsk_tg(8), 8)
out = {"bits": bits, "signal_sum": np.array([sig.sum()]), "signal_head": sig[:64]}   # the signal itself is re-made from the bits
for nfft, bps in ((256, 1), (1024, 2)):
    pxx, freqs = mlab.psd(sig * np.sqrt(bps), NFFT=nfft, Fs=8 / bps, scale_by_freq=False)
    out[f"psd_{nfft}_{bps}_pxx"], out[f"psd_{nfft}_{bps}_freqs"] = pxx, freqs
phase = np.angle(sig)
L = 8 * 4
out["tree_first"] = np.array([np.unwrap(phase[c * L:(c + 1) * L]) - np.unwrap(phase[c * L:(c + 1) * L])[0] for c in range(phase.size // L)])
out["tree_off"] = np.array([np.unwrap(phase[c * L:(c + 1) * L]) - 0.25 for c in range(phase.size // L)])
np.savez_compressed(OUT / "viz.npz", **out)
print("wrote viz.npz", {k: v.shape for k, v in out.items()})
