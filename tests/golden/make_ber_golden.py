"""BER-vs-Eb/N0 golden curve from the REFERENCE pipeline.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_ber_golden.py [blocks_per_point] [procs]

Trial block = 2**17 consecutive PN23 bits (block b starts at bit b * 2**17 of the
sequence), encoded / modulated / detected on its own exactly like the per-waveform
body of the reference example (sps 8, SOQPSK-TG, h = 1/4, detector length 2,
PT offset -1, PAM offset 0), noise from PCG64(seed = 1000 * ebn0_db + block).
Writes ber_golden.csv: one row per (Eb/N0, block) with symbol/bit error counts.
"""
from __future__ import annotations

import sys
from multiprocessing import Pool
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent
BLOCK = 1 << 17
SPS = 8


def job(args):
    ebn0, block = args
    import waveforms
    assert "/root/reference" in waveforms.__file__
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.pamapprox import rho_pulses
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_tg
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded
    from waveforms.glfsr.pn import generate_mask
    from waveforms.noise import generate_complex_awgn
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    # PN23 bits [block*BLOCK, (block+1)*BLOCK): step the reference's own recurrence
    # (glfsr.py:15-19) with plain ints — fast enough and identical by construction.
    mask, s = generate_mask(23), (1 << 23) - 1
    bits = np.empty(BLOCK, dtype=np.uint8)
    for _ in range(block * BLOCK):
        b = s & 1
        s >>= 1
        if b:
            s ^= mask
    for k in range(BLOCK):
        b = s & 1
        s >>= 1
        if b:
            s ^= mask
        bits[k] = b
    h, pulse = 0.25, freq_pulse_soqpsk_tg(SPS)
    sigma = float(np.sqrt(SPS / (2.0 * 10.0 ** (ebn0 / 10.0))))
    rng = np.random.Generator(np.random.PCG64(seed=1000 * ebn0 + block))
    symbols = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)(bits)
    _t, sig = cpm_modulate(symbols, h, pulse, SPS)
    noise = generate_complex_awgn(sigma, sig.size, rng)
    sig[:] *= np.exp(-1j * np.pi / 4)
    r = sig + noise
    L = int(pulse.size / SPS)
    q = np.cumsum(pulse) / SPS
    qt = q[int((L - 1) * SPS / 2):int((L + 1) * SPS / 2) + 1]
    pt = np.array([np.convolve(r, np.exp(-2j * np.pi * h * a * qt), mode="same") for a in (-2, 0, 2)])
    rho = rho_pulses(pulse, h, SPS, k_max=2)
    dmax = max(x.size for x in rho)
    pseudo = np.array([[-1j, 1, 1j], [np.sqrt(2) / 2 * (1 - 1j), np.sqrt(2) / 2, np.sqrt(2) / 2 * (1 + 1j)]])
    pam = np.zeros((3, r.size), dtype=np.complex128)
    for si in range(3):
        for k in range(2):
            rk = np.concatenate((rho[k], np.zeros(dmax - rho[k].size)))
            pam[si, :] += np.convolve(r, rk, mode="same") * np.conj(pseudo[k, si])
    row = [ebn0, block]
    for mf, off in ((pt, -1), (pam, 0)):
        det = SOQPSKTrellisDetector(length=2, differantial_encoding=True)
        ob, osym = [], []
        for n in range(r.size - det.length * SPS):
            if (n + off) % SPS:
                continue
            rb, rs = det.iteration(mf[:, n])
            ob.append(rb[0]); osym.append(rs[0])
        ds = np.array(osym[det.length:], dtype=np.int8)
        db = np.array(ob[det.length:], dtype=np.uint8)
        m = min(symbols.size, ds.size)
        row += [m, int(np.count_nonzero(ds[:m] - symbols[:m])), int(np.count_nonzero(db[:m] - bits[:m]))]
    return row


if __name__ == "__main__":
    # usage: make_ber_golden.py [n_blocks] [procs] [first_block] [first_ebn0] [out.csv]
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    blk0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    e0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out_name = sys.argv[5] if len(sys.argv) > 5 else "ber_golden.csv"
    jobs = [(e, b) for b in range(blk0, blk0 + nblk) for e in range(e0, 13)]
    rows = []
    with Pool(procs) as pool:
        for k, row in enumerate(pool.imap_unordered(job, jobs)):
            rows.append(row)
            if k % 13 == 0:
                print(k, "/", len(jobs), flush=True)
    rows.sort()
    hdr = "ebn0_db,block,pt_compared,pt_sym_err,pt_bit_err,pam_compared,pam_sym_err,pam_bit_err"
    (OUT / out_name).write_text(hdr + "\n" + "\n".join(",".join(map(str, r)) for r in rows) + "\n")
    print("wrote", len(rows), "rows")
