"""SOQPSK-MIL / SOQPSK-TG detection run on the MI355X path.

Same experiment as the reference's examples/soqpsk_detection.py (PN15 bits, differential
trellis encoder, CPM modulator, AWGN from PCG64(seed=1) shared across the two waveforms,
pulse-truncation and PAM matched-filter banks, length-2 trellis detector, error counts)
written against the same ``waveforms`` API, without the matplotlib figure.  Logs the same
four lines; ``run()`` returns the counts so tests can compare them with the reference's.

    python examples/soqpsk_detection.py [--per-symbol]

``--per-symbol`` drives the detector through the reference's one-call-per-symbol
``iteration()`` API instead of the batch ``detect()``.
"""
import logging
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from waveforms.cpm.modulate import cpm_modulate  # noqa: E402
from waveforms.cpm.soqpsk import freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg  # noqa: E402
from waveforms.cpm.trellis.encoder import TrellisEncoder  # noqa: E402
from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded  # noqa: E402
from waveforms.filters.matched import (MatchedFilterBank, pam_matched_filter_taps,  # noqa: E402
                                       pt_matched_filter_taps)
from waveforms.glfsr import PNSequence  # noqa: E402
from waveforms.noise import generate_complex_awgn  # noqa: E402
from waveforms.viterbi.algorithm import SOQPSKTrellisDetector  # noqa: E402

_logger = logging.getLogger("soqpsk_detection")

TIMING_OFFSET = {("MIL", "PT"): -1, ("TG", "PT"): -1, ("MIL", "PAM"): -3, ("TG", "PAM"): 0}


def run(sps: int = 10, sigma: float = np.sqrt(2) / 2, pn_degree: int = 15, per_symbol: bool = False,
        labels=("MIL", "TG")) -> dict:
    rng = np.random.Generator(np.random.PCG64(seed=1))
    bit_array = np.unpackbits(np.packbits(PNSequence(pn_degree).generate_sequence()))
    symbols = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)(bit_array)
    mod_index = 1 / 4
    pulses = {"MIL": freq_pulse_soqpsk_mil(sps=sps), "TG": freq_pulse_soqpsk_tg(sps=sps)}
    ebn0 = 10 * np.log10(sps / (2 * sigma**2))
    results = {}
    for label in labels:
        pulse = pulses[label]
        _time, signal = cpm_modulate(symbols=symbols, mod_index=mod_index, pulse_filter=pulse, sps=sps)
        noise = generate_complex_awgn(sigma, signal.size, rng)
        signal[:] *= np.exp(-1j * np.pi / 4)
        received = signal + noise
        banks = {"PT": MatchedFilterBank(pt_matched_filter_taps(pulse, mod_index, sps)),
                 "PAM": MatchedFilterBank(pam_matched_filter_taps(pulse, mod_index, sps))}
        for kind, bank in banks.items():
            det = SOQPSKTrellisDetector(length=2, differantial_encoding=True)
            offset = TIMING_OFFSET[(label, kind)]
            first = (-offset) % sps
            ncols = len(range(first, received.size - det.length * sps, sps))
            rows = bank(received, first=first, step=sps, ncols=ncols)
            if per_symbol:
                pairs = [det.iteration(row) for row in rows]
                out_bits = np.array([b[0] for b, _ in pairs])
                out_syms = np.array([s[0] for _, s in pairs])
            else:
                out_bits, out_syms = det.detect(rows)
            det_syms = np.asarray(out_syms[det.length:], dtype=np.int8)
            det_bits = np.asarray(out_bits[det.length:], dtype=np.uint8)
            m = min(symbols.size, det_syms.size)
            sym_err = int(np.count_nonzero(det_syms[:m] - symbols[:m]))
            bit_err = int(np.count_nonzero(det_bits[:m] - bit_array[:m]))
            _logger.info("SOQPSK-%s %s: Eb/N0 = %.2f dB, SER = %.3E BER = %.3E", label, kind, ebn0,
                         sym_err / m, bit_err / m)
            results[(label, kind)] = (sym_err, bit_err, m)
    return results


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    run(per_symbol="--per-symbol" in sys.argv)
