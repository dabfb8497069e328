"""ARTM multi-h CPM and PCM/FM detection on the MI355X path.

The reference ships modulators for these IRIG-106 waveforms (``waveforms.cpm.multih``,
``waveforms.cpm.pcmfm``; its ``examples/irig_comparison.py`` only plots their spectra) but no
detector.  This is the counterpart of ``examples/soqpsk_detection.py`` for them, written against
the same API plus the build's generic CPM trellis detector: PN bits -> symbol mapper ->
``cpm_modulate`` -> AWGN (numpy ``Generator``, like the reference's noise helper) -> matched-filter
rows -> ``CPMTrellisDetector`` -> error counts.

    python examples/cpm_detection.py [--ebn0 9] [--nsym 32768]
"""
import argparse
import logging
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from waveforms.cpm.modulate import cpm_modulate  # noqa: E402
from waveforms.cpm.multih import MultiHSymbolMapper, freq_pulse_multih_irig  # noqa: E402
from waveforms.cpm.pcmfm import PCMFMSymbolMapper, freq_pulse_pcmfm  # noqa: E402
from waveforms.glfsr import PNSequence  # noqa: E402
from waveforms.noise import generate_complex_awgn  # noqa: E402
from waveforms.viterbi.cpm import (ARTM_16, PCMFM_10, CPMTrellisDetector, filter_geometry,  # noqa: E402
                                   matched_filter_templates, sigma_for_ebn0)

_logger = logging.getLogger("cpm_detection")


def run(ebn0_db: float = 9.0, nsym: int = 32768, sps: int = 8, pn_degree: int = 17, seed: int = 1) -> dict:
    from waveforms_amd import _hip, device as dev

    rng = np.random.Generator(np.random.PCG64(seed=seed))
    pn = np.array(PNSequence(pn_degree).generate_sequence(), dtype=np.uint8)
    out = {}
    for label, spec, mapper, pulse in (("ARTM multi-h", ARTM_16, MultiHSymbolMapper(), freq_pulse_multih_irig(sps)),
                                       ("PCM/FM", PCMFM_10, PCMFMSymbolMapper(), freq_pulse_pcmfm(sps))):
        bits = np.resize(pn, nsym * spec.bits_per_symbol)
        symbols = mapper(bits)
        _t, signal = cpm_modulate(symbols, spec.mod_index if len(spec.K) > 1 else float(spec.mod_index[0]), pulse, sps)
        signal[:] *= np.exp(-1j * np.pi / 4)                     # remove the start phase, as examples/soqpsk_detection.py:85
        received = signal + generate_complex_awgn(sigma_for_ebn0(ebn0_db, sps, spec.bits_per_symbol), signal.size, rng)
        geo = filter_geometry(pulse.size, sps, spec, symbols.size)
        rows = dev.cpm_mf_rows(_hip.to_device(received), _hip.to_device(matched_filter_templates(pulse, sps, spec)),
                               geo["start0"], sps, geo["ncalls"])
        decided = _hip.to_host(CPMTrellisDetector(spec).detect_device(rows))[spec.D - 1:]     # call k decides symbol k - D + 1
        truth = ((symbols.astype(np.int64) + spec.M - 1) // 2)[:decided.size]
        wrong = decided.astype(np.int64) ^ truth
        sym_err = int(np.count_nonzero(wrong))
        bit_err = int(sum(bin(int(v)).count("1") for v in wrong[wrong != 0]))
        n = decided.size
        _logger.info("%-13s (%2d states): Eb/N0 = %.2f dB, SER = %.3E BER = %.3E", label, spec.nstates, ebn0_db,
                     sym_err / n, bit_err / (n * spec.bits_per_symbol))
        out[label] = (sym_err, bit_err, n)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ebn0", type=float, default=9.0)
    ap.add_argument("--nsym", type=int, default=32768)
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    run(a.ebn0, a.nsym)
