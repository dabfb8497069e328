"""tests/golden/example_flows.npz: the arrays the reference's examples/irig_comparison.py and
examples/soqpsk_example.py compute and hand to matplotlib — not the drawing.

* irig_comparison.py:44-135: PN15 bits -> {SimpleTrellis2, SOQPSKTrellis4x2DiffEncoded, SimpleTrellis4} ->
  cpm_modulate at sps 20 with the PCM/FM (Bessel order 6), SOQPSK-TG and multi-h ARTM pulses ->
  Axes.psd(signal * sqrt(bpsym), NFFT=1024, Fs=sps / bpsym, scale_by_freq=False) = matplotlib.mlab.psd.
* soqpsk_example.py:41-176: PN13 bits -> SOQPSK precoder -> cpm_modulate at sps 8 with the four SOQPSK pulses
  (B, TG, A, MIL) -> Axes.psd(NFFT=1024, Fs=8); the eye diagram of the first quarter of each signal (time / 2,
  modulo 4, t_offset 1/32, 0 for MIL); the "QPSK-esque" constellation samples of the TG signal; the phase tree.

The eye traces and the phase tree are read back from the Line2D objects the reference's OWN plot functions create
(waveforms/viz/eye.py, tree.py) on an Agg canvas.  Those two files are loaded by path: the reference's
waveforms/viz/__init__.py imports a name constellation.py does not define, so the package itself cannot be
imported (and examples/soqpsk_example.py does not start) — an ordinary ImportError of the reference, not of this build.
The phase tree is taken for the MIL pulse (one symbol: 4 sequences) and for SOQPSK-A's first 16 taps as a 2-symbol
pulse (256 sequences); the example's TG tree enumerates 65 536 sequences.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference python3 /root/repo/tests/golden/make_example_flows_golden.py
"""
import importlib.util
from pathlib import Path

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt
import numpy as np
from matplotlib import mlab

import waveforms
from waveforms.cpm.modulate import cpm_modulate
from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, freq_pulse_multih_irig
from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER, freq_pulse_pcmfm
from waveforms.cpm.soqpsk import (SOQPSK_DENOM, SOQPSK_NUMER, freq_pulse_soqpsk_a, freq_pulse_soqpsk_b,
                                  freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg)
from waveforms.cpm.trellis.encoder import TrellisEncoder
from waveforms.cpm.trellis.model import SimpleTrellis2, SimpleTrellis4, SOQPSKTrellis4x2DiffEncoded
from waveforms.glfsr import PNSequence

assert "/root/reference" in waveforms.__file__
OUT = Path(__file__).resolve().parent
REF_VIZ = Path(waveforms.__file__).resolve().parent / "viz"


def load(name):
    spec = importlib.util.spec_from_file_location(f"_ref_viz_{name}", REF_VIZ / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_eye, ref_tree = load("eye"), load("tree")
out = {}

# ---- examples/irig_comparison.py
sps, fft_size = 20, 2**10
bit_array = np.unpackbits(np.frombuffer(np.packbits(PNSequence(15).generate_sequence()), dtype=np.uint8))
out["irig_nbits"] = np.array([bit_array.size])
for name, trellis, mod_index, pulse, bpsym in (
        ("pcmfm", SimpleTrellis2, PCMFM_NUMER / PCMFM_DENOM, freq_pulse_pcmfm(sps=sps, order=6), 1),
        ("soqpsk", SOQPSKTrellis4x2DiffEncoded, SOQPSK_NUMER / SOQPSK_DENOM, freq_pulse_soqpsk_tg(sps=sps), 1),
        ("multih", SimpleTrellis4, MULTIH_IRIG_NUMER / MULTIH_IRIG_DENOM, freq_pulse_multih_irig(sps=sps), 2)):
    symbols = TrellisEncoder(trellis)(bit_array)
    _t, sig = cpm_modulate(symbols=symbols, mod_index=mod_index, pulse_filter=pulse, sps=sps)
    pxx, freqs = mlab.psd(sig * np.sqrt(bpsym), NFFT=fft_size, Fs=sps / bpsym, scale_by_freq=False)
    out[f"irig_{name}_mod_index"] = np.atleast_1d(np.asarray(mod_index, dtype=np.float64))
    out[f"irig_{name}_pulse"], out[f"irig_{name}_q"] = pulse, np.cumsum(pulse) / sps
    out[f"irig_{name}_nsym"], out[f"irig_{name}_symsum"] = np.array([symbols.size]), np.array([int(np.asarray(symbols, dtype=np.int64).sum())])
    out[f"irig_{name}_sig_head"], out[f"irig_{name}_sig_tail"], out[f"irig_{name}_sig_sum"] = sig[:48], sig[-48:], np.array([sig.sum()])
    out[f"irig_{name}_pxx"], out[f"irig_{name}_freqs"] = pxx, freqs

# ---- examples/soqpsk_example.py
sps, mod_index = 8, 1 / 4
bit_array = np.unpackbits(np.packbits(PNSequence(13).generate_sequence()))
precoder = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)
symbols = precoder(bit_array)
out["sq_nbits"], out["sq_symsum"] = np.array([bit_array.size]), np.array([int(np.asarray(symbols, dtype=np.int64).sum())])
signals = {}
for label, pulse in (("B", freq_pulse_soqpsk_b(sps=sps)), ("TG", freq_pulse_soqpsk_tg(sps=sps)),
                     ("A", freq_pulse_soqpsk_a(sps=sps)), ("MIL", freq_pulse_soqpsk_mil(sps=sps))):
    normalized_time, sig = cpm_modulate(symbols=symbols, mod_index=mod_index, pulse_filter=pulse, sps=sps)
    signals[label] = sig[:]
    normalized_time /= 2
    fig, (ax_re, ax_im) = plt.subplots(2)
    n4 = normalized_time.size // 4
    ref_eye.plot_eye_diagram(normalized_time[:n4], sig[:n4], sps=sps, modulo=4, t_offset=0 if label == "MIL" else 1 / sps / 4,
                             axes=(ax_re, ax_im))
    tr_t = np.array([ln.get_xdata() for ln in ax_re.lines])
    tr_re = np.array([ln.get_ydata() for ln in ax_re.lines])
    tr_im = np.array([ln.get_ydata() for ln in ax_im.lines])
    plt.close(fig)
    pxx, freqs = mlab.psd(sig, NFFT=fft_size, Fs=sps)
    out[f"sq_{label}_pulse"] = pulse
    out[f"sq_{label}_sig_head"], out[f"sq_{label}_sig_sum"] = sig[:48], np.array([sig.sum()])
    out[f"sq_{label}_pxx"], out["sq_freqs"] = pxx, freqs
    out[f"sq_{label}_eye_shape"] = np.array(tr_re.shape)
    out[f"sq_{label}_eye_t0"] = tr_t[0]                                    # every trace has the same abscissa ...
    out[f"sq_{label}_eye_t_spread"] = np.array([np.abs(tr_t - tr_t[0]).max()])   # ... to rounding
    out[f"sq_{label}_eye_re_head"], out[f"sq_{label}_eye_im_head"] = tr_re[:12], tr_im[:12]
    out[f"sq_{label}_eye_re_tail"], out[f"sq_{label}_eye_im_tail"] = tr_re[-4:], tr_im[-4:]
    out[f"sq_{label}_eye_re_colsum"], out[f"sq_{label}_eye_im_colsum"] = tr_re.sum(axis=0), tr_im.sum(axis=0)
tg = signals["TG"]
qpsk = np.zeros_like(tg)
qpsk[sps:] += tg.real[:-sps]
qpsk[:] += tg.imag * 1j
out["sq_constellation"] = qpsk[sps::sps * 2][1:][:1024]      # what constellation.py:36-37 draws: the first 1024 points

for label, pulse in (("MIL", freq_pulse_soqpsk_mil(sps=sps)), ("A16", freq_pulse_soqpsk_a(sps=sps)[:16])):
    fig, ax = plt.subplots(1)
    ref_tree.generate_cpm_phase_tree(pulse, 1 / 4, encoder=precoder, sps=sps, axis=ax)
    out[f"sq_tree_{label}_pulse"] = pulse
    out[f"sq_tree_{label}_t"] = np.asarray(ax.lines[0].get_xdata())
    out[f"sq_tree_{label}"] = np.array([ln.get_ydata() for ln in ax.lines])
    plt.close(fig)

np.savez_compressed(OUT / "example_flows.npz", **out)
print("wrote example_flows.npz", {k: v.shape for k, v in out.items()})
