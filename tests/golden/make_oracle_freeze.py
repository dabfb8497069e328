"""Freeze the outputs of every golden-pinned oracle function (rows a1-a11 of SURVEY 8) on fixed inputs.

    python tests/golden/make_oracle_freeze.py            # rewrites tests/golden/oracle_freeze.json

The reference-generated goldens pin WHAT these functions must return; this digest pins that they KEEP returning it:
the one place where a statement of the oracle was ever reordered to follow a kernel is the build-defined CPM
detector (oracle/cpm_oracle.c, which says so) — the functions below may never be.  A change to any of them shows up
as a changed digest in review (tests/test_oracle_golden.py::test_golden_pinned_oracle_outputs_are_frozen).
Regenerate ONLY together with a golden that motivates the change.

Integer outputs are hashed exactly; floating-point outputs after rounding to 11 significant digits, so that the
last-ulp spread of libm / numpy builds does not trip the test while any change of formula, order or constant does.
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.asarray(a)
        if a.dtype.kind in "fc":
            v = np.ascontiguousarray(a).view(np.float64).ravel()
            with np.errstate(divide="ignore", invalid="ignore"):
                e = np.where(v == 0, 0.0, np.floor(np.log10(np.abs(v))))
            q = np.where(np.isfinite(v), np.round(v / 10.0 ** e, 10) * 10.0 ** e, v)
            q = np.where(q == 0, 0.0, q)                      # no -0.0
            h.update(np.array2string(q, threshold=q.size + 1, formatter={"float_kind": lambda x: f"{x:.10e}"}).encode())
        else:
            h.update(str(a.dtype).encode())
            h.update(np.ascontiguousarray(a).tobytes())
        h.update(str(a.shape).encode())
    return h.hexdigest()[:24]


def compute() -> dict:
    import oracle as o

    o.build_c_oracle()
    out = {}
    pn9 = o.pn_sequence(9)
    bits = np.unpackbits(np.packbits(pn9))
    out["pn_sequence"] = digest(pn9, o.pn_sequence(15), o.pn_sequence(7))
    out["glfsr_bits"] = digest(*[o.glfsr_bits(o.lfsr_mask(d), (1 << d) - 1, 5000)[0] for d in (23, 31, 47)],
                               np.array([o.lfsr_mask(k) for k in range(2, 65)], dtype=np.uint64))
    for name in ("SOQPSKTrellis8x1", "SOQPSKTrellis4x2", "SOQPSKTrellis4x2DiffEncoded", "SimpleTrellis2", "SimpleTrellis4"):
        out[f"fsm_encode[{name}]"] = digest(o.fsm_encode(name, bits)[0])
    out["mappers"] = digest(o.soqpsk_precoder(bits)[0], o.multih_mapper(bits)[0], o.pcmfm_mapper(bits))
    tg8, tg10, mil8 = o.freq_pulse_soqpsk_tg(8), o.freq_pulse_soqpsk_tg(10), o.freq_pulse_soqpsk_mil(8)
    out["pulses"] = digest(tg8, tg10, mil8, o.freq_pulse_soqpsk_a(8), o.freq_pulse_soqpsk_b(8), o.freq_pulse_multih_irig(8),
                           o.freq_pulse_pcmfm(8, 4), o.kaiser_fir_lpf(8, 0.5), *o.rho_pulses(tg8, 0.25, 8), *o.rho_pulses(tg10, 0.25, 10))
    sym = o.fsm_encode("SOQPSKTrellis4x2DiffEncoded", bits)[0]
    t, sig = o.cpm_modulate(sym, 0.25, tg8, 8)
    out["cpm_modulate"] = digest(t, sig, o.cpm_modulate(o.multih_mapper(bits)[0], np.array([4 / 16, 5 / 16]), o.freq_pulse_multih_irig(8), 8)[1],
                                 o.cpm_modulate(o.pcmfm_mapper(bits), 0.7, o.freq_pulse_pcmfm(8, 4), 8)[1])
    fp = o.upsample_fir(sym, 0.25, tg8, 8)
    out["fir_and_phase"] = digest(fp, o.upsample_fir_direct(sym[:64], 0.25, tg8, 8), o.frequency_modulate(fp, 8), o.phase_modulate(fp[:256], 0.3))
    out["numpy_awgn"] = digest(o.numpy_awgn(np.sqrt(2) / 2, 300, np.random.Generator(np.random.PCG64(1))))
    out["philox_awgn"] = digest(o.philox_awgn(0.6324555320336759, 1, 3, 0, 4096), o.philox_awgn(0.5, 7, 0, 12345, 33))
    r = sig * np.exp(-1j * np.pi / 4) + o.philox_awgn(0.5, 2, 0, 0, sig.size)
    pt, pam = o.pt_bank(r, tg8, 0.25, 8), o.pam_bank(r, tg8, 0.25, 8)
    out["mf_banks"] = digest(pt, pam, o.mf_bank_decim_direct(r, o.pt_taps(tg8, 0.25, 8), 7, 8, 500), o.decimate_columns(r.size, 8, 2, -1))
    rng = np.random.default_rng(5)
    trip = rng.standard_normal((600, 3)) + 1j * rng.standard_normal((600, 3))
    out["viterbi_detect"] = digest(*[x for L in (2, 4, 6) for d in (True, False) for x in o.ViterbiOracle(L, d).run(trip, full=True)])
    res = o.detection_run(np.asarray(o.pn_sequence(15)), tg8, 0.25, 8, None, noise=o.philox_awgn(o.sigma_for_ebn0(6.0, 8), 1, 0, 0, (32767 + 1) * 8))
    out["detection_run"] = digest(np.array([res["sym_errors"], res["bit_errors"], res["compared"]], dtype=np.int64))
    return out


if __name__ == "__main__":
    out = compute()
    path = Path(__file__).resolve().parent / "oracle_freeze.json"
    path.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(json.dumps(out, indent=1, sort_keys=True))
