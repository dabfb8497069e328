"""Generic CPM trellis detector — oracle side (ARTM multi-h, PCM/FM).

TEST INFRASTRUCTURE ONLY.  Parity status: UNPINNED / BUILD-DEFINED — the reference has no
detector for these waveforms (see the header of ``cpm_oracle.c``, which is the executable
definition).  This module builds the detector's constant data (matched-filter templates,
rotation table) from the reference's own pulse and modulation-index definitions, drives the
sequential C detector, and carries a second, independent pure-Python statement of the same
recursion (``viterbi_py``) that the C is checked against on small inputs.

Citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import numpy_ref as _nr

__all__ = ["CPMDetectorSpec", "ARTM_16", "ARTM_64", "ARTM_256", "PCMFM_SPEC", "cpm_templates", "cpm_rot_table",
           "cpm_mf_rows", "cpm_viterbi", "cpm_viterbi_py", "cpm_geometry", "cpm_detection_run", "cpm_sigma_for_ebn0",
           "symbols_to_u", "u_to_bits", "cpm_min_distance"]


@dataclass(frozen=True)
class CPMDetectorSpec:
    """Waveform + detector design.  ``K/p`` are the modulation indices
    (waveforms/cpm/multih/pulse_filters.py:7-8: (4, 5)/16; waveforms/cpm/pcmfm/__init__.py:5-6:
    7/10), ``M`` the alphabet size, ``Lp`` the matched-filter length in symbols (pulse truncation),
    ``NC`` the number of phase classes in the trellis state (NC == p: every phase state;
    NC < p: the phase index is carried per survivor), ``D`` the decision delay."""
    M: int
    p: int
    K: tuple
    Lp: int
    NC: int
    D: int

    @property
    def lgM(self) -> int:
        return self.M.bit_length() - 1

    @property
    def nstates(self) -> int:
        return self.NC * self.M ** (self.Lp - 1)

    @property
    def nfilt(self) -> int:
        return self.M ** self.Lp


# ARTM multi-h CPM (IRIG-106 Tier II): M = 4, 3RC, h = {4/16, 5/16}.
ARTM_256 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=3, NC=16, D=32)   # full trellis p * M^(L-1) (notes/cpm/cpm.md:128-140)
ARTM_64 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=2, NC=16, D=32)    # pulse truncated to 2 symbols
ARTM_16 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=2, NC=4, D=32)     # BASELINE configs[2]: 16 states
PCMFM_SPEC = CPMDetectorSpec(M=2, p=10, K=(7,), Lp=2, NC=5, D=32)    # PCM/FM: h = 7/10, 10 states (padded to 16 lanes)


class _Cfg(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("lgM", ctypes.c_int), ("p", ctypes.c_int), ("nh", ctypes.c_int),
                ("K", ctypes.c_int * 8), ("Lp", ctypes.c_int), ("NC", ctypes.c_int), ("D", ctypes.c_int)]


def _cfg(spec: CPMDetectorSpec) -> _Cfg:
    c = _Cfg()
    c.M, c.lgM, c.p, c.nh, c.Lp, c.NC, c.D = spec.M, spec.lgM, spec.p, len(spec.K), spec.Lp, spec.NC, spec.D
    for i, k in enumerate(spec.K):
        c.K[i] = k
    return c


def symbols_to_u(symbols, M: int) -> np.ndarray:
    """alpha in {-(M-1), ..., M-1} -> U = (alpha + M - 1) / 2 (notes/cpm/cpm.md:96-100)."""
    return ((np.asarray(symbols, dtype=np.int64) + (M - 1)) // 2).astype(np.uint8)


def u_to_bits(u, M: int) -> np.ndarray:
    """Inverse of the reference's natural-binary mappers (waveforms/cpm/multih/precoder.py:22-23:
    alpha = 2 (2 b0 + b1) - 3; waveforms/cpm/pcmfm/precoder.py: alpha = 2 b - 1): MSB first."""
    u = np.asarray(u, dtype=np.uint8)
    lg = M.bit_length() - 1
    return np.stack([(u >> (lg - 1 - i)) & 1 for i in range(lg)], axis=-1).reshape(-1).astype(np.uint8)


def cpm_geometry(pulse, sps: int, spec: CPMDetectorSpec, nsym: int) -> dict:
    """Sample alignment.  The modulator (waveforms/cpm/modulate.py:95-99) puts symbol m's impulse
    at sample (m+1)*sps and np.convolve(..., "same") centres the pulse there, so the phase ramp of
    symbol m starts at sample s_m = (m+1)*sps - (ntaps-1)//2.  A matched filter of Lp symbols
    looks at the middle Lp symbols of the pulse: window offset o = ((ntaps-1) - Lp*sps)//2, and
    spans sps+1 samples (both ends, like q_t in examples/soqpsk_detection.py:135-140)."""
    ntaps = int(np.asarray(pulse).size)
    c = (ntaps - 1) // 2
    o = ((ntaps - 1) - spec.Lp * sps) // 2
    if o < -(sps // 2):
        raise ValueError("matched filter longer than the pulse")
    start0 = sps - c + o
    npts = max((nsym + 1) * sps, ntaps)
    ntm = sps + 1
    ncalls = max(0, min(nsym, (npts - ntm - start0) // sps + 1))
    return dict(start0=start0, ntm=ntm, offset=o, ncalls=int(ncalls), npts=int(npts))


def cpm_templates(pulse, sps: int, spec: CPMDetectorSpec) -> np.ndarray:
    """complex128[nh][NF][sps+1]: T[c][f][k] = exp(j 2 pi sum_i h_{c-i} alpha_i Q[o + k + i sps]),
    f = u_0 + M u_1 + M^2 u_2 (u_0 = the symbol of column c, u_i = i symbols older),
    Q = cumsum(g)/sps (examples/soqpsk_detection.py:136)."""
    g = np.asarray(pulse, dtype=np.float64)
    Q = np.cumsum(g) / sps
    o = ((g.size - 1) - spec.Lp * sps) // 2
    nh, M, Lp = len(spec.K), spec.M, spec.Lp
    k = np.arange(sps + 1)
    out = np.empty((nh, spec.nfilt, sps + 1), dtype=np.complex128)
    for c in range(nh):
        for f in range(spec.nfilt):
            ph = np.zeros(sps + 1)
            for i in range(Lp):
                u = (f // M ** i) % M
                alpha = 2 * u - (M - 1)
                h = spec.K[(c - i) % nh] / spec.p
                idx = o + k + i * sps
                ph += h * alpha * np.where(idx < 0, 0.0, Q[np.clip(idx, 0, g.size - 1)])
            out[c, f] = np.exp(2j * np.pi * ph)
    return out


def cpm_rot_table(spec: CPMDetectorSpec) -> np.ndarray:
    """float64[2p][2]: (cos, sin)(pi r / p) — shared bit for bit with the device."""
    r = np.arange(2 * spec.p)
    return np.ascontiguousarray(np.stack([np.cos(np.pi * r / spec.p), np.sin(np.pi * r / spec.p)], axis=1))


def cpm_mf_rows(r, templates, start0: int, sps: int, ncalls: int) -> np.ndarray:
    r = np.ascontiguousarray(r, dtype=np.complex128)
    T = np.ascontiguousarray(templates, dtype=np.complex128)
    nh, NF, ntm = T.shape
    out = np.empty((ncalls, NF), dtype=np.complex128)
    _nr._c().orc_cpm_mf_rows(_nr._p(r), ctypes.c_int64(r.size), _nr._p(T), nh, NF, ntm, ctypes.c_int64(start0), sps,
                             ctypes.c_int64(ncalls), _nr._p(out))
    return out


class cpm_viterbi:
    """Sequential detector (cpm_oracle.c), state carried across run() calls."""

    def __init__(self, spec: CPMDetectorSpec):
        lib = _nr._c()
        lib.orc_cpm_viterbi.restype = ctypes.c_int
        lib.orc_cpm_state_size.restype = ctypes.c_int
        self.spec, self._cfg = spec, _cfg(spec)
        self._st = ctypes.create_string_buffer(lib.orc_cpm_state_size())
        lib.orc_cpm_state_init(ctypes.byref(self._cfg), self._st)
        self._rot = cpm_rot_table(spec)
        self.n = 0

    def run(self, rows) -> np.ndarray:
        """rows complex128[ncalls][NF] -> decided U for symbols n-D+1 of the calls made (uint8);
        the first D-1 calls of a fresh detector decide nothing."""
        rows = np.ascontiguousarray(rows, dtype=np.complex128)
        ncalls = rows.shape[0]
        out = np.zeros(self.n + ncalls + 1, dtype=np.uint8)
        rc = _nr._c().orc_cpm_viterbi(ctypes.byref(self._cfg), _nr._p(self._rot), _nr._p(rows), ctypes.c_int64(ncalls),
                                      self._st, _nr._p(out))
        if rc:
            raise ValueError("unsupported detector configuration")
        D = self.spec.D
        lo, hi = max(self.n - D + 1, 0), max(self.n + ncalls - D + 1, 0)
        self.n += ncalls
        return out[lo:hi].copy()


def cpm_viterbi_py(spec: CPMDetectorSpec, rows) -> np.ndarray:
    """The same recursion written independently in plain Python (dicts of candidates per end
    state, candidates listed by (start state, input) ascending) — the check on cpm_oracle.c."""
    M, p, Lp, NC, D, K = spec.M, spec.p, spec.Lp, spec.NC, spec.D, spec.K
    S, lg = spec.nstates, spec.lgM
    rot = cpm_rot_table(spec)
    metric = [0.0] * S
    v = [s % NC for s in range(S)]
    hist = [0] * S
    msub = M ** max(Lp - 2, 0)
    out = []
    import math

    for n, Z in enumerate(np.asarray(rows)):
        m_old = n - Lp + 1
        K_old = K[m_old % len(K)] if m_old >= 0 else 0
        ksum = sum(K[i % len(K)] for i in range(max(m_old, 0)))
        tilt = ((M - 1) * ksum) % (2 * p)
        cands = {}
        for s in range(S):
            corr = s // NC
            r = (2 * v[s] - tilt) % (2 * p)
            for u in range(M):
                z = Z[u + M * corr]
                inc = -_fma(rot[r, 0], z.real, rot[r, 1] * z.imag)
                u_old = u if Lp == 1 else corr // msub
                corr2 = 0 if Lp == 1 else u + M * (corr % msub)
                v2 = (v[s] + K_old * u_old) % p
                cands.setdefault(v2 % NC + NC * corr2, []).append((metric[s] + inc, v2, (hist[s] << lg | u) & (2 ** 64 - 1)))
        new = []
        for s2 in range(S):
            best = None
            for cand in cands.get(s2, []):
                if best is None or cand[0] < best[0]:
                    best = cand
            new.append(best if best is not None else (math.inf, 0, 0))
        mn = min(c[0] for c in new)
        metric = [c[0] - mn for c in new]
        v = [c[1] for c in new]
        hist = [c[2] for c in new]
        if n >= D - 1:
            best_state = next(s for s in range(S) if new[s][0] == mn)
            out.append((hist[best_state] >> (lg * (D - 1))) & (M - 1))
    return np.array(out, dtype=np.uint8)


def _fma(a, b, c):
    """Correctly rounded a*b + c (math.fma arrives in Python 3.13): exact rational arithmetic,
    rounded once."""
    from fractions import Fraction

    return float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def cpm_sigma_for_ebn0(ebn0_db: float, sps: int, bits_per_symbol: int) -> float:
    """Es/N0 = sps / (2 sigma^2) (examples/soqpsk_detection.py:132 states it for 1 bit/symbol),
    Eb = Es / bits_per_symbol."""
    return float(np.sqrt(sps / (2.0 * bits_per_symbol * 10.0 ** (ebn0_db / 10.0))))


def cpm_detection_run(symbols, pulse, sps: int, spec: CPMDetectorSpec, sigma=None, rng=None, noise=None) -> dict:
    """Modulate (reference modulator) -> derotate by the pi/4 start phase + AWGN
    (examples/soqpsk_detection.py:85-89) -> matched-filter rows -> sequential detector ->
    symbol / bit error counts over the decided range."""
    symbols = np.asarray(symbols, dtype=np.int8)
    h = np.asarray(spec.K, dtype=np.float64) / spec.p
    _t, sig = _nr.cpm_modulate(symbols, h if h.size > 1 else float(h[0]), pulse, sps)
    sig = sig * np.exp(-1j * np.pi / 4)
    if noise is None and sigma:
        noise = _nr.numpy_awgn(sigma, sig.size, rng)
    r = sig if noise is None else sig + noise
    geo = cpm_geometry(pulse, sps, spec, symbols.size)
    rows = cpm_mf_rows(r, cpm_templates(pulse, sps, spec), geo["start0"], sps, geo["ncalls"])
    dec = cpm_viterbi(spec).run(rows)
    u = symbols_to_u(symbols, spec.M)[:dec.size]
    x = dec ^ u
    bit_err = int(np.unpackbits(x[:, None], axis=1).sum())
    return dict(received=r, rows=rows, decisions=dec, truth=u, sym_errors=int(np.count_nonzero(x)), bit_errors=bit_err,
                compared=int(dec.size), geometry=geo)


def cpm_min_distance(pulse, sps: int, M: int, K, p: int, max_len: int = 8) -> float:
    """Normalised squared minimum Euclidean distance of the CPM scheme (Anderson, Aulin & Sundberg):
        d^2 = log2(M) * min over difference sequences gamma (gamma_0 != 0, gamma_i in 2 * {-(M-1)..M-1})
              of (1/T) * integral (1 - cos dphi(t)) dt,   dphi(t) = 2 pi sum_i h_i gamma_i q(t - iT),
    over both alignments of the modulation-index cycle.  Depth-first over sequences of up to
    ``max_len`` symbols with the running integral as the bound; a sequence ends (merges) when the
    accumulated phase difference is a multiple of 2 pi after its last pulse has finished.
    Theory anchor for the build-defined detector: the published value for ARTM CPM is 1.29."""
    g = np.asarray(pulse, dtype=np.float64)
    q = np.cumsum(g) / sps
    L = -(-(g.size - 1) // sps)
    ramp = np.concatenate([q[:L * sps], np.full(1, q[-1])])          # q at samples 0 .. L*sps
    ramp = np.concatenate([ramp, np.full(sps * (max_len + L + 2), q[-1])])
    gammas = [2 * d for d in range(-(M - 1), M)]
    nh = len(K)
    best = [np.inf]
    lg = np.log2(M)

    def dfs(seq, align, acc):
        n = len(seq)
        # integral over symbol interval n-1 is final once symbol n-1's successors are known only
        # through later intervals; evaluate interval (n-1) contribution with the symbols so far
        if n:
            k = np.arange(sps) + (n - 1) * sps
            dphi = np.zeros(sps)
            for i, gi in enumerate(seq):
                if gi:
                    dphi += (K[(i + align) % nh] / p) * gi * ramp[k - i * sps]
            acc = acc + np.mean(1.0 - np.cos(2 * np.pi * dphi))
            if lg * acc >= best[0]:
                return
        # try to close: all remaining symbols zero -> add the tails of the last L-1 intervals
        if n and seq[-1] != 0 or n > 1:
            total = sum((K[(i + align) % nh]) * gi for i, gi in enumerate(seq))
            if n and total % (2 * p) == 0 and any(seq):
                tail = acc
                for j in range(n, n + L - 1):
                    k = np.arange(sps) + j * sps
                    dphi = np.zeros(sps)
                    for i, gi in enumerate(seq):
                        if gi:
                            dphi += (K[(i + align) % nh] / p) * gi * ramp[k - i * sps]
                    tail += np.mean(1.0 - np.cos(2 * np.pi * dphi))
                best[0] = min(best[0], lg * tail)
        if n >= max_len:
            return
        for gi in (gammas if n else [x for x in gammas if x > 0]):
            dfs(seq + [gi], align, acc)

    for align in range(nh):
        dfs([], align, 0.0)
    return float(best[0])
