"""The three figure scripts of the reference on the MI355X path, in one file.

What the reference's examples/irig_comparison.py, examples/pcmfm_test.py and examples/soqpsk_example.py draw —
spectra of the three IRIG-106 waveforms, PCM/FM spectra per pre-modulation filter order, and the SOQPSK family's
pulses / spectra / eye diagrams / constellation / phase tree — written against the same ``waveforms`` API
(``cpm_modulate`` and the ``waveforms.viz`` functions run on the GPU; matplotlib only draws the arrays that come back).
The arrays themselves are pinned against the reference's by tests/test_example_flows.py.

    python examples/waveform_figures.py [--out DIR] [irig] [pcmfm] [soqpsk]      (default: all three, PNGs under DIR)
"""
import argparse
import sys
from pathlib import Path

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
import numpy as np  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from waveforms.cpm.modulate import cpm_modulate  # noqa: E402
from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, freq_pulse_multih_irig  # noqa: E402
from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER, freq_pulse_pcmfm  # noqa: E402
from waveforms.cpm.soqpsk import (SOQPSK_DENOM, SOQPSK_NUMER, freq_pulse_soqpsk_a, freq_pulse_soqpsk_b,  # noqa: E402
                                  freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg)
from waveforms.cpm.trellis.encoder import TrellisEncoder  # noqa: E402
from waveforms.cpm.trellis.model import SimpleTrellis2, SimpleTrellis4, SOQPSKTrellis4x2DiffEncoded  # noqa: E402
from waveforms.glfsr import PNSequence  # noqa: E402
from waveforms.viz import (constellation, generate_cpm_phase_tree, plot_eye_diagram,  # noqa: E402
                           plot_power_spectral_density)


def pn_bits(degree: int) -> np.ndarray:
    return np.unpackbits(np.packbits(PNSequence(degree).generate_sequence()))


def irig(out: Path, sps: int = 20) -> Path:
    """PCM/FM, SOQPSK-TG and ARTM multi-h over the same PN15 bits: pulses and spectra per bit rate."""
    bits = pn_bits(15)
    fig, (ax_pulse, ax_psd) = plt.subplots(2, figsize=(9, 8))
    for label, trellis, h, pulse, bps in (
            ("PCM/FM", SimpleTrellis2, PCMFM_NUMER / PCMFM_DENOM, freq_pulse_pcmfm(sps=sps, order=6), 1),
            ("SOQPSK-TG", SOQPSKTrellis4x2DiffEncoded, SOQPSK_NUMER / SOQPSK_DENOM, freq_pulse_soqpsk_tg(sps=sps), 1),
            ("ARTM CPM", SimpleTrellis4, MULTIH_IRIG_NUMER / MULTIH_IRIG_DENOM, freq_pulse_multih_irig(sps=sps), 2)):
        symbols = TrellisEncoder(trellis)(bits)
        _t, signal = cpm_modulate(symbols=symbols, mod_index=h, pulse_filter=pulse, sps=sps)
        ax_pulse.plot(np.arange(pulse.size) / sps, pulse, label=label)
        plot_power_spectral_density(signal, sps=sps, bps=bps, nfft=1024, axis=ax_psd)
    ax_pulse.set_xlabel("symbol times")
    ax_pulse.legend()
    ax_psd.set_xlabel("frequency / bit rate")
    path = out / "irig_comparison.png"
    fig.tight_layout()
    fig.savefig(path, dpi=80)
    plt.close(fig)
    return path


def pcmfm(out: Path, sps: int = 20) -> Path:
    """PCM/FM spectrum against the order of its Bessel pre-modulation filter (4 ... 8)."""
    bits = pn_bits(13)
    symbols = TrellisEncoder(SimpleTrellis2)(bits)
    fig, ax = plt.subplots(1, figsize=(9, 5))
    for order in range(4, 9):
        _t, signal = cpm_modulate(symbols=symbols, mod_index=PCMFM_NUMER / PCMFM_DENOM,
                                  pulse_filter=freq_pulse_pcmfm(sps=sps, order=order), sps=sps)
        plot_power_spectral_density(signal, sps=sps, nfft=1024, axis=ax)
    ax.set_xlabel("frequency / bit rate")
    path = out / "pcmfm_filter_orders.png"
    fig.tight_layout()
    fig.savefig(path, dpi=80)
    plt.close(fig)
    return path


def soqpsk(out: Path, sps: int = 8) -> Path:
    """The four SOQPSK pulses: spectra, eye diagrams, the TG constellation and the MIL phase tree."""
    bits = pn_bits(13)
    encoder = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)
    symbols = encoder(bits)
    h = SOQPSK_NUMER / SOQPSK_DENOM
    pulses = {"B": freq_pulse_soqpsk_b(sps=sps), "TG": freq_pulse_soqpsk_tg(sps=sps),
              "A": freq_pulse_soqpsk_a(sps=sps), "MIL": freq_pulse_soqpsk_mil(sps=sps)}
    fig = plt.figure(figsize=(12, 10))
    grid = fig.add_gridspec(3, 4)
    ax_psd = fig.add_subplot(grid[0, :2])
    ax_const = fig.add_subplot(grid[0, 2])
    ax_tree = fig.add_subplot(grid[0, 3])
    for row, (label, pulse) in enumerate(pulses.items()):
        time, signal = cpm_modulate(symbols=symbols, mod_index=h, pulse_filter=pulse, sps=sps)
        plot_power_spectral_density(signal, sps=sps, nfft=1024, axis=ax_psd)
        quarter = time.size // 4
        ax_re = fig.add_subplot(grid[1 + row // 2, 2 * (row % 2)])        # two pulses per figure row: (real, imaginary) each
        ax_im = fig.add_subplot(grid[1 + row // 2, 2 * (row % 2) + 1])
        plot_eye_diagram(time[:quarter] / 2, signal[:quarter], sps=sps, modulo=4, t_offset=0 if label == "MIL" else 1 / sps / 4,
                         axes=(ax_re, ax_im))
        ax_re.set_title(f"SOQPSK-{label}")
        if label == "TG":
            # offset QPSK view: the real rail delayed by one symbol, one point per two symbols
            shifted = np.zeros_like(signal)
            shifted[sps:] += signal.real[:-sps]
            shifted += 1j * signal.imag
            constellation(shifted[sps::2 * sps][1:], axis=ax_const)
    generate_cpm_phase_tree(pulses["MIL"], h, encoder=TrellisEncoder(SOQPSKTrellis4x2DiffEncoded), sps=sps, axis=ax_tree)
    path = out / "soqpsk_family.png"
    fig.tight_layout()
    fig.savefig(path, dpi=80)
    plt.close(fig)
    return path


FIGURES = {"irig": irig, "pcmfm": pcmfm, "soqpsk": soqpsk}


def main(argv=None) -> list[Path]:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="figures")
    ap.add_argument("which", nargs="*", help="irig | pcmfm | soqpsk (default: all)")
    a = ap.parse_args(argv)
    for name in a.which:
        if name not in FIGURES:
            ap.error(f"unknown figure {name!r}")
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    made = [FIGURES[name](out) for name in (a.which or sorted(FIGURES))]
    for p in made:
        print(p)
    return made


if __name__ == "__main__":
    main()
