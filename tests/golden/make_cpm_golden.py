"""tests/golden/cpm_detect.npz — regression pin of the BUILD-DEFINED generic CPM detector.

There is no reference detector for these waveforms, so unlike the other fixtures this one is NOT
produced by /root/reference: it freezes what oracle/cpm_oracle.c (the definition) decided at the
commit that introduced it, on inputs made by the reference-pinned modulator restatement, so that a
later change to the oracle's arithmetic or tie-breaks cannot go unnoticed.

    python tests/golden/make_cpm_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

out = {}
bits = oracle.pn_sequence(15)[:6000]
for name, spec, pulse, sym, bps in (
        ("artm16", oracle.ARTM_16, oracle.freq_pulse_multih_irig(8), oracle.multih_mapper(bits)[0], 2),
        ("pcmfm10", oracle.PCMFM_SPEC, oracle.freq_pulse_pcmfm(8), oracle.pcmfm_mapper(bits[:3000]), 1)):
    # noise by the counter-based device spec (oracle.philox_awgn: seed 11, stream 3) — re-made in the test, not stored
    noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(5.0, 8, bps), 11, 3, 0, (sym.size + 1) * 8)
    res = oracle.cpm_detection_run(sym, pulse, 8, spec, noise=noise)
    out[f"{name}_rows_head"] = res["rows"][:64]
    out[f"{name}_decisions"] = res["decisions"]
    out[f"{name}_errors"] = np.array([res["sym_errors"], res["bit_errors"], res["compared"]])
out["bits"] = bits
out["d2_artm"] = np.array([oracle.cpm_min_distance(oracle.freq_pulse_multih_irig(8), 8, 4, (4, 5), 16, 6)])
np.savez_compressed(ROOT / "tests" / "golden" / "cpm_detect.npz", **out)
print({k: (v.shape, v.dtype) for k, v in out.items()}, out["artm16_errors"], out["pcmfm10_errors"], out["d2_artm"])
