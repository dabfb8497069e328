"""Generate tests/golden/*.npz by IMPORTING the reference (mcdiarmid/waveforms).

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_golden.py

Every array stored here is either an input we chose or an output the reference
computed for it; no reference source text is stored.  numpy 2.2.6 / scipy 1.15.3.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent

import waveforms  # noqa: E402  (must resolve to /root/reference)

assert "/root/reference" in waveforms.__file__, waveforms.__file__

from waveforms.cpm.helpers import normalize_cpm_filter  # noqa: E402
from waveforms.cpm.modulate import cpm_modulate, frequency_modulate, phase_modulate  # noqa: E402
from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, MultiHSymbolMapper, freq_pulse_multih_irig  # noqa: E402
from waveforms.cpm.pamapprox import pam_unit_pulse, pam_unit_pulse2, rho_pulses  # noqa: E402
from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER, PCMFMSymbolMapper, freq_pulse_pcmfm  # noqa: E402
from waveforms.cpm.soqpsk import (SOQPSKPrecoder, freq_pulse_soqpsk_a, freq_pulse_soqpsk_b,  # noqa: E402
                                  freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg)
from waveforms.cpm.trellis.encoder import TrellisEncoder  # noqa: E402
from waveforms.cpm.trellis import model as tm  # noqa: E402
from waveforms.filters.lpf import kaiser_fir_lpf  # noqa: E402
from waveforms.glfsr import PNSequence  # noqa: E402
from waveforms.glfsr.pn import GALOIS_LFSR_POLYS, generate_mask  # noqa: E402
from waveforms.noise import generate_complex_awgn  # noqa: E402
from waveforms.viterbi.algorithm import SOQPSKTrellisDetector  # noqa: E402

TRELLIS_NAMES = ["SOQPSKTrellis8x1", "SOQPSKTrellis4x2", "SOQPSKTrellis4x2DiffEncoded",
                 "SimpleTrellis2", "SimpleTrellis4"]


def pn_bits(degree, pad=True):
    bits = np.array(PNSequence(degree).generate_sequence(), dtype=np.uint8)
    return np.unpackbits(np.packbits(bits)) if pad else bits


def glfsr():
    d = {}
    for deg in (2, 3, 7, 9, 15, 16):
        d[f"pn{deg}_packed"] = np.packbits(np.array(PNSequence(deg).generate_sequence(), dtype=np.uint8))
    d["masks"] = np.array([generate_mask(k) for k in range(2, len(GALOIS_LFSR_POLYS))], dtype=np.uint64)
    # first 200 bits + state after them for a few long registers
    for deg in (23, 31, 47, 64):
        p = PNSequence(deg)
        d[f"pn{deg}_first200"] = np.array([p.next_bit() for _ in range(200)], dtype=np.uint8)
        d[f"pn{deg}_state200"] = np.array([p.state], dtype=np.uint64)
    np.savez_compressed(OUT / "glfsr.npz", **d)


def encode():
    rng = np.random.Generator(np.random.PCG64(20240601))
    rand = rng.integers(0, 2, size=4096, dtype=np.uint8)
    pn9, pn15 = pn_bits(9), pn_bits(15)
    d = {"rand_bits": rand}
    for name in TRELLIS_NAMES:
        tr = getattr(tm, name)
        d[f"{name}__rand"] = TrellisEncoder(tr)(rand)
        d[f"{name}__pn9"] = TrellisEncoder(tr)(pn9)
        d[f"{name}__pn15"] = TrellisEncoder(tr)(pn15)
        # chunked (stateful) call: 1002 + rest must equal one shot
        enc = TrellisEncoder(tr)
        d[f"{name}__rand_chunked"] = np.concatenate((enc(rand[:1002]), enc(rand[1002:])))
        d[f"{name}__final_i_state"] = np.array([enc.i, enc.state], dtype=np.int64)
        d[f"{name}__dims"] = np.array([tr.columns, tr.states, tr.input_cardinality,
                                       tr.output_cardinality, tr.branches_per_column], dtype=np.int64)
        d[f"{name}__branches"] = np.array([[b.inp, b.out, b.start, b.end] for col in tr.branches
                                           for b in col], dtype=np.int8)
    pre = SOQPSKPrecoder()
    d["precoder__rand"] = pre(rand)
    pre = SOQPSKPrecoder()
    d["precoder__rand_chunked"] = np.concatenate((pre(rand[:1001]), pre(rand[1001:])))
    mh = MultiHSymbolMapper()
    d["multih__rand"] = mh(rand)
    d["pcmfm__rand"] = PCMFMSymbolMapper()(rand)
    np.savez_compressed(OUT / "encode.npz", **d)


def pulses():
    d = {}
    for sps in (4, 8, 10):
        d[f"tg_{sps}"] = freq_pulse_soqpsk_tg(sps)
        d[f"mil_{sps}"] = freq_pulse_soqpsk_mil(sps)
        d[f"a_{sps}"] = freq_pulse_soqpsk_a(sps)
        d[f"b_{sps}"] = freq_pulse_soqpsk_b(sps)
        d[f"multih_{sps}"] = freq_pulse_multih_irig(sps)
        for k, r in enumerate(rho_pulses(freq_pulse_soqpsk_tg(sps), 0.25, sps, 2)):
            d[f"rho{k}_tg_{sps}"] = r
        for k, r in enumerate(rho_pulses(freq_pulse_soqpsk_mil(sps), 0.25, sps, 2)):
            d[f"rho{k}_mil_{sps}"] = r
    for sps, order in ((8, 4), (8, 6), (20, 4), (20, 8)):
        d[f"pcmfm_{sps}_{order}"] = freq_pulse_pcmfm(sps, order)
    d["kaiser_8_0p5"] = kaiser_fir_lpf(8, 0.5)
    d["kaiser_10_0p7_w0p2_r60"] = kaiser_fir_lpf(10, 0.7, 0.2, 60.0)
    q = np.cumsum(freq_pulse_soqpsk_tg(8)) / 8
    d["unit_pulse_tg_8"] = pam_unit_pulse(q, 0.25)
    d["unit_pulse2_tg_8"] = pam_unit_pulse2(q, 0.25)
    d["normalize_in"] = np.linspace(0.1, 2.0, 17)
    d["normalize_out"] = normalize_cpm_filter(8, d["normalize_in"])
    from waveforms.cpm.soqpsk import SOQPSK_DENOM, SOQPSK_NUMER
    d["consts"] = np.array([SOQPSK_NUMER, SOQPSK_DENOM, PCMFM_NUMER, PCMFM_DENOM, MULTIH_IRIG_DENOM,
                            *MULTIH_IRIG_NUMER], dtype=np.int64)
    np.savez_compressed(OUT / "pulses.npz", **d)


def modulate():
    d = {}
    pn9 = pn_bits(9)
    sym_soq = TrellisEncoder(tm.SOQPSKTrellis4x2DiffEncoded)(pn9)
    sym_mh = TrellisEncoder(tm.SimpleTrellis4)(pn9)
    sym_pcm = TrellisEncoder(tm.SimpleTrellis2)(pn9)
    cases = {
        "tg8": (sym_soq, 0.25, freq_pulse_soqpsk_tg(8), 8),
        "tg10": (sym_soq, 0.25, freq_pulse_soqpsk_tg(10), 10),
        "mil8": (sym_soq, 0.25, freq_pulse_soqpsk_mil(8), 8),
        "mh8": (sym_mh, np.array([4, 5]) / 16, freq_pulse_multih_irig(8), 8),
        "pcm8": (sym_pcm, 0.7, freq_pulse_pcmfm(8, 4), 8),
        "pcm5": (sym_pcm[:100], 0.7, freq_pulse_pcmfm(5, 4), 5),   # odd sps
        "tiny": (sym_soq[:3], 0.25, freq_pulse_soqpsk_tg(8), 8),    # shorter than the pulse
        "one": (sym_soq[5:6], 0.25, freq_pulse_soqpsk_mil(8), 8),
    }
    for name, (sym, h, g, sps) in cases.items():
        t, s = cpm_modulate(sym, h, g, sps)
        d[f"{name}__symbols"] = sym
        d[f"{name}__h"] = np.atleast_1d(np.asarray(h, dtype=np.float64))
        d[f"{name}__pulse"] = g
        d[f"{name}__sps"] = np.array([sps])
        d[f"{name}__time"] = t
        d[f"{name}__signal"] = s
    # the FIR stage alone (np.convolve 'same' of the zero-stuffed train)
    sym, h, g, sps = cases["tg8"]
    x = np.zeros((sym.size + 1) * sps)
    x[sps:-1:sps] = sym * h
    d["tg8__freq_pulses"] = np.convolve(x, g, mode="same")
    # frequency_modulate / phase_modulate on arbitrary input
    rng = np.random.Generator(np.random.PCG64(7))
    fp = rng.normal(0, 0.3, size=3001)
    d["fm_in"] = fp
    d["fm_out_sps8"] = frequency_modulate(fp, 8, 0.25)
    d["fm_out_sps5"] = frequency_modulate(fp, 5)
    d["pm_out"] = phase_modulate(fp, 1.7)
    # checksums of the long PN15 cases quoted in SURVEY 8(c)-4
    pn15 = pn_bits(15)
    s15 = TrellisEncoder(tm.SOQPSKTrellis4x2DiffEncoded)(pn15)
    _t, sig = cpm_modulate(s15, 0.25, freq_pulse_soqpsk_tg(8), 8)
    d["pn15_tg8_sum"] = np.array([sig.sum()])
    d["pn15_tg8_every997"] = sig[::997].copy()
    np.savez_compressed(OUT / "modulate.npz", **d)


def awgn():
    rng = np.random.Generator(np.random.PCG64(seed=1))
    d = {"seed1_sigma_sqrt_half_4104": generate_complex_awgn(np.sqrt(2) / 2, 4104, rng)}
    np.savez_compressed(OUT / "awgn.npz", **d)


def run_chain(bits, pulse, h, sps, sigma, rng, offsets, length=2):
    """Same flow as the reference example's per-waveform body (the example itself is
    exercised verbatim in e2e()); returns intermediates for the PN9 fixtures."""
    symbols = TrellisEncoder(tm.SOQPSKTrellis4x2DiffEncoded)(bits)
    _t, sig = cpm_modulate(symbols, h, pulse, sps)
    noise = generate_complex_awgn(sigma, sig.size, rng)
    sig[:] *= np.exp(-1j * np.pi / 4)
    r = sig + noise
    L = int(pulse.size / sps)
    q = np.cumsum(pulse) / sps
    qt = q[int((L - 1) * sps / 2):int((L + 1) * sps / 2) + 1]
    pt = np.array([np.convolve(r, np.exp(-2j * np.pi * h * a * qt), mode="same") for a in (-2, 0, 2)])
    rho = rho_pulses(pulse, h, sps, k_max=2)
    dmax = max(x.size for x in rho)
    pseudo = np.array([[-1j, 1, 1j], [np.sqrt(2) / 2 * (1 - 1j), np.sqrt(2) / 2, np.sqrt(2) / 2 * (1 + 1j)]])
    pam = np.zeros((3, r.size), dtype=np.complex128)
    for s in range(3):
        for k in range(2):
            rk = np.concatenate((rho[k], np.zeros(dmax - rho[k].size)))
            pam[s, :] += np.convolve(r, rk, mode="same") * np.conj(pseudo[k, s])
    res = dict(symbols=symbols, noise=noise, received=r, pt_full=pt, pam_full=pam)
    for kind, mf in (("PT", pt), ("PAM", pam)):
        det = SOQPSKTrellisDetector(length=length, differantial_encoding=True)
        ob, osym, cols = [], [], []
        for n in range(r.size - det.length * sps):
            if (n + offsets[kind]) % sps:
                continue
            rb, rs = det.iteration(mf[:, n])
            ob.append(rb[0]); osym.append(rs[0]); cols.append(n)
        ds = np.array(osym[det.length:], dtype=np.int8)
        db = np.array(ob[det.length:], dtype=np.uint8)
        m = min(symbols.size, ds.size)
        res[f"{kind}_cols"] = np.array(cols)
        res[f"{kind}_det_bits"] = np.array(ob)
        res[f"{kind}_det_syms"] = np.array(osym)
        res[f"{kind}_errors"] = np.array([np.count_nonzero(ds[:m] - symbols[:m]),
                                          np.count_nonzero(db[:m] - bits[:m]), m])
    return res


def detect():
    d = {}
    pn9 = pn_bits(9)
    rng = np.random.Generator(np.random.PCG64(seed=1))
    # noisy enough (Eb/N0 = 4 dB) that the PN9 fixture contains detector errors
    sigma = float(np.sqrt(8 / (2 * 10 ** 0.4)))
    res = run_chain(pn9, freq_pulse_soqpsk_tg(8), 0.25, 8, sigma, rng, {"PT": -1, "PAM": 0})
    d["pn9_bits"] = pn9
    d["pn9_sigma"] = np.array([sigma])
    for k, v in res.items():
        d[f"pn9_tg8__{k}"] = v
    # detector alone on random triplets: pins tie-break, normalisation, traceback
    rng = np.random.Generator(np.random.PCG64(99))
    trip = (rng.normal(size=(4000, 3)) + 1j * rng.normal(size=(4000, 3)))
    # a few exact ties and zeros
    trip[100:110] = 0
    trip[200:210] = 1 + 1j
    d["triplets"] = trip
    for length in (2, 4, 6):
        for diff in (True, False):
            det = SOQPSKTrellisDetector(length=length, differantial_encoding=diff)
            fb, fs = [], []
            for z in trip:
                b, s = det.iteration(z)
                fb.append(b); fs.append(s)
            d[f"trip_L{length}_diff{int(diff)}_bits"] = np.array(fb)
            d[f"trip_L{length}_diff{int(diff)}_syms"] = np.array(fs)
    np.savez_compressed(OUT / "detect.npz", **d)


def e2e():
    """Error counts of the reference pipeline on PN15 (+ the verbatim example)."""
    out = {}
    pn15 = pn_bits(15)
    # (1) the example's configuration: sps 10, sigma sqrt(2)/2, one rng shared MIL -> TG
    rng = np.random.Generator(np.random.PCG64(seed=1))
    for label, pulse, offs in (("MIL", freq_pulse_soqpsk_mil(10), {"PT": -1, "PAM": -3}),
                               ("TG", freq_pulse_soqpsk_tg(10), {"PT": -1, "PAM": 0})):
        res = run_chain(pn15, pulse, 0.25, 10, np.sqrt(2) / 2, rng, offs)
        for kind in ("PT", "PAM"):
            out[f"example_sps10_{label}_{kind}"] = [int(v) for v in res[f"{kind}_errors"]]
    # (2) sps 8, Eb/N0 = 10 dB, TG only, fresh seed 1
    rng = np.random.Generator(np.random.PCG64(seed=1))
    res = run_chain(pn15, freq_pulse_soqpsk_tg(8), 0.25, 8, float(np.sqrt(0.4)), rng, {"PT": -1, "PAM": 0})
    for kind in ("PT", "PAM"):
        out[f"sps8_10dB_TG_{kind}"] = [int(v) for v in res[f"{kind}_errors"]]
    (OUT / "e2e.json").write_text(json.dumps(out, indent=1) + "\n")
    print(out)


def offset_scan():
    """Row a9: error counts of the reference chain for every timing offset (sps 8, Eb/N0 = 10 dB,
    TG, PN15, fresh seed 1 per offset so each run sees the same noise)."""
    out = {}
    pn15 = pn_bits(15)
    for off in range(-4, 4):
        rng = np.random.Generator(np.random.PCG64(seed=1))
        res = run_chain(pn15, freq_pulse_soqpsk_tg(8), 0.25, 8, float(np.sqrt(0.4)), rng, {"PT": off, "PAM": off})
        out[str(off)] = {k: [int(v) for v in res[f"{k}_errors"]] for k in ("PT", "PAM")}
    (OUT / "offset_scan.json").write_text(json.dumps(out, indent=1) + "\n")
    print(out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["glfsr", "encode", "pulses", "modulate", "awgn", "detect", "e2e", "offset_scan"]
    for name in which:
        globals()[name]()
        print("done", name)
