"""NumPy + C restatement of the reference's hot path (see ``oracle/__init__.py``).

TEST INFRASTRUCTURE ONLY — never imported by the product package.
Citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

__all__ = [
    "build_c_oracle", "lfsr_mask", "glfsr_bits", "pn_sequence", "TRELLISES", "trellis_tables",
    "fsm_encode", "soqpsk_precoder", "multih_mapper", "pcmfm_mapper", "normalize_cpm_filter",
    "freq_pulse_soqpsk", "freq_pulse_soqpsk_tg", "freq_pulse_soqpsk_mil", "freq_pulse_soqpsk_a",
    "freq_pulse_soqpsk_b", "freq_pulse_multih_irig", "freq_pulse_pcmfm", "kaiser_fir_lpf",
    "pam_unit_pulse", "pam_unit_pulse2", "rho_pulses", "upsample_fir", "frequency_modulate",
    "phase_modulate", "cpm_modulate", "numpy_awgn", "philox4x32_10", "philox_awgn", "box_muller32",
    "pt_taps", "pt_bank", "pam_bank", "PSEUDO_SYMBOLS", "decimate_columns", "ViterbiOracle",
    "viterbi_detect", "count_errors", "detection_run", "sigma_for_ebn0", "upsample_fir_direct",
    "mf_bank_decim_direct",
]

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libwforacle.so"
_lib = None


def build_c_oracle(force: bool = False) -> Path:
    """Compile the C restatements with gcc (building the checker is not using it).
    WF_ORACLE_SANITIZE=1 (tools/sanitize.py, CPU box): an AddressSanitizer + UBSan build in its
    own file; the process must then run with gcc's libasan preloaded."""
    srcs = [_HERE / "wf_oracle.c"] + ([_HERE / "cpm_oracle.c"] if (_HERE / "cpm_oracle.c").exists() else [])
    san = os.environ.get("WF_ORACLE_SANITIZE") == "1"
    so = _SO.with_name("libwforacle_san.so") if san else _SO
    if force or not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        so.parent.mkdir(exist_ok=True)
        tmp = so.with_suffix(f".{os.getpid()}.tmp")
        # -ffp-contract=off: no fused multiply-adds, the reference's numpy/libm do none.
        opt = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if san else ["-O2"]
        subprocess.check_call(
            ["gcc", *opt, "-ffp-contract=off", "-fPIC", "-shared", "-o", str(tmp), *map(str, srcs), "-lm"]
        )
        os.replace(tmp, so)
    return so


def _c():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(str(build_c_oracle()))
        lib.orc_viterbi_state_size.restype = ctypes.c_int
        lib.orc_fsm_encode.restype = ctypes.c_int
        lib.orc_viterbi_run.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------- K1
# Tap table of waveforms/glfsr/pn.py:6-72 (Xilinx XAPP052 maximal-length taps),
# index = degree.  Held as a compact string so this stays data, not code.
_TAPS = """0;1;2,1;3,2;4,3;5,3;6,5;7,6;8,6,5,4;9,5;10,7;11,9;12,6,4,1;13,4,3,1;14,5,3,1;15,14;
16,15,13,4;17,14;18,11;19,6,2,1;20,17;21,19;22,21;23,18;24,23,22,17;25,22;26,6,2,1;27,5,2,1;
28,25;29,27;30,6,4,1;31,28;32,22,2,1;33,20;34,27,2,1;35,33;36,25;37,5,4,3,2,1;38,6,5,1;39,35;
40,38,21,19;41,38;42,41,20,19;43,42,38,37;44,43,18,17;45,44,42,41;46,45,26,25;47,42;
48,47,21,20;49,40;50,49,24,23;51,50,36,35;52,49;53,52,38,37;54,53,18,17;55,31;56,55,35,34;
57,50;58,39;59,58,38,37;60,59;61,60,46,45;62,61,6,5;63,62;64,63,61,60"""
LFSR_TAPS = [[int(t) for t in row.split(",")] for row in _TAPS.replace("\n", "").split(";")]


def lfsr_mask(degree: int) -> int:
    """waveforms/glfsr/pn.py:75-90."""
    if not 1 < degree < len(LFSR_TAPS):
        raise KeyError(f"PRBS Polynomial Not Defined for {degree}.")
    return sum(1 << (t - 1) for t in LFSR_TAPS[degree])


def glfsr_bits(mask: int, state: int, n: int) -> tuple[np.ndarray, int]:
    """waveforms/glfsr/glfsr.py:15-19, n steps.  Returns (bits u8[n], new state)."""
    bits = np.empty(n, dtype=np.uint8)
    st = ctypes.c_uint64(state)
    _c().orc_lfsr_generate(ctypes.c_uint64(mask), ctypes.byref(st), _p(bits), ctypes.c_int64(n))
    return bits, st.value


def pn_sequence(degree: int, n: int | None = None) -> np.ndarray:
    """waveforms/glfsr/pn.py:93-107: state 2**degree-1, 2**degree-1 bits by default."""
    n = (1 << degree) - 1 if n is None else n
    return glfsr_bits(lfsr_mask(degree), (1 << degree) - 1, n)[0]


# ----------------------------------------------------------------------------- K2
# Branch lists of waveforms/cpm/trellis/model.py:179-293 as (inp, out, start, end).
def _soqpsk_4x2(diff: bool):
    cols = []
    # column 0 (even / I): next = (inp ? start|2 : start&1)...  written out explicitly:
    ev = [(0, 0, 0, 0), (1, 2, 0, 2), (0, 0, 1, 1), (1, -2, 1, 3),
          (0, -2, 2, 0), (1, 0, 2, 2), (0, 2, 3, 1), (1, 0, 3, 3)]
    od = [(0, 0, 0, 0), (1, -2, 0, 1), (0, 2, 1, 0), (1, 0, 1, 1),
          (0, 0, 2, 2), (1, 2, 2, 3), (0, -2, 3, 2), (1, 0, 3, 3)]
    if diff:
        # model.py:233-258: input label flipped on branches leaving states {2,3} (even)
        # and {1,3} (odd); list order kept as in the reference.
        ev = [(i ^ (s >> 1), o, s, e) for (i, o, s, e) in ev]
        od = [(i ^ (s & 1), o, s, e) for (i, o, s, e) in od]
    cols.append(ev)
    cols.append(od)
    return cols


def _soqpsk_8x1():
    a = [(0, 0, 0, 4), (1, 2, 0, 6), (0, 0, 1, 5), (1, -2, 1, 7), (0, -2, 2, 4), (1, 0, 2, 6),
         (0, 2, 3, 5), (1, 0, 3, 7), (0, 0, 4, 0), (1, -2, 4, 1), (0, 2, 5, 0), (1, 0, 5, 1),
         (0, 0, 6, 2), (1, 2, 6, 3), (0, -2, 7, 2), (1, 0, 7, 3)]
    return [a]


TRELLISES = {
    "SOQPSKTrellis8x1": _soqpsk_8x1(),
    "SOQPSKTrellis4x2": _soqpsk_4x2(False),
    "SOQPSKTrellis4x2DiffEncoded": _soqpsk_4x2(True),
    "SimpleTrellis2": [[(i, 2 * i - 1, s, i) for s in range(2) for i in range(2)]],
    "SimpleTrellis4": [[(i, 2 * i - 3, s, i) for s in range(4) for i in range(4)]],
}


def trellis_tables(name: str) -> dict:
    """Dense tables for a trellis: forward (next/out by [col][state][inp], as
    model.py:127-137 forward_map) and the flat per-branch arrays the detector uses."""
    cols = TRELLISES[name]
    ncol = len(cols)
    states = max(b[2] for c in cols for b in c) + 1                       # model.py:74-84
    n_inputs = len({b[0] for c in cols for b in c})                        # model.py:22-36
    if 1 << (n_inputs.bit_length() - 1) != n_inputs:
        raise ValueError("Number of unique inputs should be multiple of 2.")
    card = n_inputs.bit_length() - 1
    nxt = np.zeros((ncol, states, n_inputs), dtype=np.uint8)
    out = np.zeros((ncol, states, n_inputs), dtype=np.int8)
    for c, col in enumerate(cols):
        for (i, o, s, e) in col:
            nxt[c, s, i], out[c, s, i] = e, o
    alphabet = sorted({b[1] for c in cols for b in c})                     # model.py:175-176
    flat = [b for c in cols for b in c]
    return dict(
        columns=ncol, states=states, card=card, bpc=len(cols[0]), next=nxt, out=out,
        alphabet=alphabet,
        br_inp=np.array([b[0] for b in flat], dtype=np.int8),
        br_out=np.array([b[1] for b in flat], dtype=np.int8),
        br_out_idx=np.array([alphabet.index(b[1]) for b in flat], dtype=np.uint8),
        br_start=np.array([b[2] for b in flat], dtype=np.uint8),
        br_end=np.array([b[3] for b in flat], dtype=np.uint8),
    )


def fsm_encode(name: str, bits: np.ndarray, i: int = 0, state: int = 0):
    """waveforms/cpm/trellis/encoder.py:17-48.  Returns (symbols i8, i, state)."""
    t = trellis_tables(name)
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    if bits.size % t["card"]:
        raise ValueError("Input length must be a multiple of FSM cardinality.")
    sym = np.empty(bits.size // t["card"], dtype=np.int8)
    ii, st = ctypes.c_int64(i), ctypes.c_int32(state)
    rc = _c().orc_fsm_encode(_p(t["next"]), _p(t["out"]), t["columns"], t["states"], t["card"],
                             _p(bits), ctypes.c_int64(bits.size), _p(sym), ctypes.byref(ii),
                             ctypes.byref(st))
    assert rc == 0
    return sym, ii.value, st.value


def soqpsk_precoder(bits: np.ndarray, i: int = 0, mem=(0, 0)):
    """waveforms/cpm/soqpsk/precoder.py:10-24.  Returns (symbols, i, mem)."""
    a = np.concatenate((np.asarray(mem), bits)).astype(np.int8)
    sign = np.ones(bits.shape, dtype=np.int8)
    sign[i::2] = -1
    out = sign * (2 * a[1:-1] - 1) * (a[:-2] - a[2:])
    return out, (i + len(bits)) % 2, tuple(int(v) for v in a[-2:])


def multih_mapper(bits: np.ndarray, i: int = 0):
    """waveforms/cpm/multih/precoder.py:9-23 (note: `i` advances *before* slicing)."""
    if bits.size % 2:
        raise ValueError("Odd length bit array passed into quaternary mapper.")
    i = (i + len(bits)) % 2
    return 2 * (2 * bits[i::2] + bits[(i + 1) % 2::2]).astype(np.int8) - 3, i


def pcmfm_mapper(bits: np.ndarray):
    """waveforms/cpm/pcmfm/precoder.py:6-15."""
    return 2 * bits.astype(np.int8) - 1


# ----------------------------------------------------------------------------- a3 pulses
def normalize_cpm_filter(sps: int, g: np.ndarray) -> np.ndarray:
    """waveforms/cpm/helpers.py:5-19."""
    return sps / (np.sum(g) * 2) * g


def freq_pulse_soqpsk(t1=1.5, t2=0.5, rho=0.7, b=1.25, sps=8) -> np.ndarray:
    """waveforms/cpm/soqpsk/pulse_filters.py:9-50."""
    span = (t1 + t2) * 2
    t = np.linspace(-span, span, num=int(span * sps * 2) + 1, dtype=np.float64, endpoint=True)
    shape = np.cos(np.pi * rho * b * t / 2) / (1 - np.power(rho * b * t, 2)) * np.sinc(b * t / 2)
    win = np.ones(t.shape, dtype=np.float64)
    if t2 > 0:
        edge = np.where((np.abs(t) >= 2 * t1) & (np.abs(t) <= span))
        win[edge] = (1 + np.cos(np.pi * (t[edge] / 2 - t1) / t2)) / 2
        win[np.where(np.abs(t) > span)] = 0
    scale = sps / (np.sum(shape * win) * 2)
    return scale * shape * win


def freq_pulse_soqpsk_tg(sps=8):
    return freq_pulse_soqpsk(sps=sps)                                      # pulse_filters.py:107-116


def freq_pulse_soqpsk_a(sps=8):
    return freq_pulse_soqpsk(b=1.35, t1=1.4, t2=0.6, rho=1.0, sps=sps)     # :53-62


def freq_pulse_soqpsk_b(sps=8):
    return freq_pulse_soqpsk(b=1.45, t1=2.8, t2=1.2, rho=0.5, sps=sps)     # :65-74


def freq_pulse_soqpsk_mil(sps=8):
    g = np.full(sps + 1, 0.5)                                              # :102-104
    g[0] = 0
    return g


def freq_pulse_multih_irig(sps=8, length=3):
    """waveforms/cpm/multih/pulse_filters.py:11-23."""
    t = np.linspace(0, length, num=length * sps + 1)
    return normalize_cpm_filter(sps, (1 - np.cos(2 * np.pi * t / length)) / (2 * length))


def freq_pulse_pcmfm(sps=8, order=4):
    """waveforms/cpm/pcmfm/pulse_filters.py:8-25 (scipy besselap + impulse)."""
    from scipy.signal import besselap, impulse

    nrz = np.ones(sps) / (2 * sps)
    _t, bessel = impulse(besselap(order, norm="mag"),
                         T=np.linspace(0, 3 * 2 / 0.7, num=2 * sps + 1, endpoint=False))
    return normalize_cpm_filter(sps, np.convolve(nrz, bessel, mode="full"))


def kaiser_fir_lpf(sps, f_cutoff, width=None, ripple_db=80.0):
    """waveforms/filters/lpf.py:13-34."""
    from scipy.signal import firwin, kaiserord

    numtaps, beta = kaiserord(ripple_db, width or 1 / sps)
    return firwin(numtaps=numtaps, cutoff=f_cutoff / (sps / 2), window=("kaiser", beta))


def pam_unit_pulse(q, h):
    """waveforms/cpm/pamapprox.py:12-29."""
    r = np.zeros(q.size * 2 + 1)
    r[1:q.size + 1] = np.sin(2 * h * np.pi * q) / np.sin(h * np.pi)
    r[q.size + 1:] = np.sin(h * np.pi - 2 * h * np.pi * q) / np.sin(h * np.pi)
    return r


def pam_unit_pulse2(q, h):
    """waveforms/cpm/pamapprox.py:32-49."""
    pih = h * np.pi
    r = np.zeros(q.size * 2 - 1)
    r[:q.size] = np.sin(2 * pih * q) / np.sin(pih)
    r[q.size:] = np.sin(pih - 2 * pih * q[1:]) / np.sin(pih)
    return r


def rho_pulses(g, h, sps, k_max=2):
    """waveforms/cpm/pamapprox.py:52-102.  Shift rows: k = 0 uses each of 0..L-1
    twice; k >= 1 uses (x, x+1) pairs; scale k+1; slice [last_shift*sps : -L*sps]."""
    L = int(g.size / sps)
    u = pam_unit_pulse2(np.cumsum(g) / sps, h)
    total = L * sps + u.size
    out = []
    for k in range(k_max):
        shifts = [x + (c if k > 0 else 0) for x in range(L) for c in (0, 1)]
        prod = None
        for s in shifts:
            seg = np.zeros(total)
            seg[s * sps:s * sps + u.size] = u
            prod = seg if prod is None else prod * seg
        out.append((float(k + 1) * prod)[shifts[-1] * sps:total - L * sps])
    return out


# ----------------------------------------------------------------------------- K3/K4
def upsample_fir(symbols, mod_index, g, sps):
    """waveforms/cpm/modulate.py:75-99 — returns freq_pulses f64[(N+1)*sps]."""
    if isinstance(mod_index, (float, int)):
        mod_index = [float(mod_index)]
    h = np.asarray(mod_index, dtype=np.float64)
    n = symbols.size
    x = np.zeros((n + 1) * sps, dtype=np.float64)
    x[sps:-1:sps] = symbols * h.take(np.arange(n) % h.size)
    return np.convolve(x, g, mode="same")


def upsample_fir_direct(symbols, mod_index, g, sps):
    """C direct-sum form of upsample_fir (CPU-baseline helper; checked against it)."""
    h = np.atleast_1d(np.asarray(mod_index, dtype=np.float64))
    symbols = np.ascontiguousarray(symbols, dtype=np.int8)
    g = np.ascontiguousarray(g, dtype=np.float64)
    out = np.empty(max((symbols.size + 1) * sps, g.size), dtype=np.float64)
    _c().orc_upsample_fir(_p(symbols), ctypes.c_int64(symbols.size), _p(h), int(h.size), _p(g),
                          int(g.size), int(sps), _p(out))
    return out


def frequency_modulate(freq_pulses, sps, initial_phase=0.0):
    """waveforms/cpm/modulate.py:28-54 (sequential accumulate-with-modulo)."""
    fp = np.ascontiguousarray(freq_pulses, dtype=np.float64)
    out = np.empty(fp.size, dtype=np.complex128)
    _c().orc_frequency_modulate(_p(fp), ctypes.c_int64(fp.size), int(sps),
                                ctypes.c_double(initial_phase), _p(out))
    return out


def phase_modulate(phase, sensitivity):
    """waveforms/cpm/modulate.py:12-25."""
    return np.exp(1j * sensitivity * phase)


def cpm_modulate(symbols, mod_index, g, sps=8):
    """waveforms/cpm/modulate.py:57-101 -> (normalized_time, signal)."""
    n = symbols.size
    t = np.linspace(0, n + 1, num=(n + 1) * sps, dtype=np.float64, endpoint=False)
    return t, frequency_modulate(upsample_fir(symbols, mod_index, g, sps), sps, np.pi / 4)


# ----------------------------------------------------------------------------- K5
def numpy_awgn(sigma, size, rng):
    """waveforms/noise.py:24-32 with an explicit numpy Generator."""
    return rng.normal(loc=0, scale=sigma, size=(size, 2)).view(np.complex128).flatten()


def sigma_for_ebn0(ebn0_db: float, sps: int) -> float:
    """Inverse of examples/soqpsk_detection.py:132: Eb/N0 = 10 log10(sps / (2 sigma^2))."""
    return float(np.sqrt(sps / (2.0 * 10.0 ** (ebn0_db / 10.0))))


def philox4x32_10(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    _c().orc_philox4x32_10(_p(np.asarray(ctr, dtype=np.uint32)),
                           _p(np.asarray(key, dtype=np.uint32)), _p(out))
    return out


def box_muller32(words, sigma):
    """Box-Muller of uint32 word pairs (n x 2), the transform inside philox_awgn."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(w.shape[0], dtype=np.complex128)
    _c().orc_box_muller32(_p(w), ctypes.c_int64(w.shape[0]), ctypes.c_double(sigma), _p(out))
    return out


def philox_awgn(sigma, seed, stream, first_index, n, signal=None):
    """Build-defined device noise spec (Philox4x32-10 + Box-Muller), see wf_oracle.c."""
    out = np.empty(n, dtype=np.complex128)
    sig = None
    if signal is not None:
        sig = np.ascontiguousarray(signal, dtype=np.complex128)
        assert sig.size == n
    _c().orc_philox_awgn(ctypes.c_double(sigma), ctypes.c_uint64(seed), ctypes.c_uint64(stream),
                         ctypes.c_uint64(first_index), ctypes.c_int64(n),
                         _p(sig) if sig is not None else None, _p(out))
    return out


# ----------------------------------------------------------------------------- K6/K7
PSEUDO_SYMBOLS = np.array(                                   # examples/soqpsk_detection.py:57-63
    [[-1j, 1, 1j],
     [np.sqrt(2) / 2 * (1 - 1j), np.sqrt(2) / 2, np.sqrt(2) / 2 * (1 + 1j)]], dtype=np.complex128)


def pt_taps(g, h, sps, alphas=(-2, 0, 2)):
    """examples/soqpsk_detection.py:135-156 — pulse-truncation taps, one row per alpha."""
    L = int(g.size / sps)
    q = np.cumsum(g) / sps
    qt = q[int((L - 1) * sps / 2):int((L + 1) * sps / 2) + 1]
    return np.array([np.exp(-2j * np.pi * h * a * qt) for a in alphas])


def pt_bank(r, g, h, sps):
    """Full-rate PT bank: 3 x len(r) complex (examples/soqpsk_detection.py:140-156)."""
    return np.array([np.convolve(r, t, mode="same") for t in pt_taps(g, h, sps)])


def pam_bank(r, g, h, sps, pseudo=PSEUDO_SYMBOLS):
    """Full-rate PAM bank (examples/soqpsk_detection.py:158-173)."""
    rho = rho_pulses(g, h, sps, k_max=2)
    d_max = max(p.size for p in rho)
    k_max, nsym = pseudo.shape
    out = np.zeros((nsym, r.size), dtype=np.complex128)
    for s in range(nsym):
        for k in range(k_max):
            rk = np.concatenate((rho[k], np.zeros(d_max - rho[k].size)))
            out[s, :] += np.convolve(r, rk, mode="same") * np.conj(pseudo[k, s])
    return out


def mf_bank_decim_direct(r, taps, first, sps, ncols):
    """C decimating bank == np.convolve(r, taps[f], 'same')[first + k*sps] (CPU baseline)."""
    r = np.ascontiguousarray(r, dtype=np.complex128)
    taps = np.ascontiguousarray(taps, dtype=np.complex128)
    out = np.empty((ncols, taps.shape[0]), dtype=np.complex128)
    _c().orc_mf_bank_decim(_p(r), ctypes.c_int64(r.size), _p(taps), int(taps.shape[0]),
                           int(taps.shape[1]), ctypes.c_int64(first), int(sps),
                           ctypes.c_int64(ncols), _p(out))
    return out


def decimate_columns(size, sps, length, timing_offset):
    """Sample indices the detector consumes (examples/soqpsk_detection.py:189-192):
    n in range(size - length*sps) with (n + timing_offset) % sps == 0."""
    n = np.arange(size - length * sps)
    return n[(n + timing_offset) % sps == 0]


# ----------------------------------------------------------------------------- K8-K10
class ViterbiOracle:
    """waveforms/viterbi/algorithm.py:18-101 (state carried across calls)."""

    def __init__(self, length=2, differential=True):
        self.length = length
        self.t = trellis_tables("SOQPSKTrellis4x2DiffEncoded" if differential
                                else "SOQPSKTrellis4x2")
        self._st = ctypes.create_string_buffer(_c().orc_viterbi_state_size())

    def run(self, mf_rows: np.ndarray, full=False):
        """mf_rows: (n, 3) complex — one row per call.  Returns element [0] of the
        reference's two return arrays per call (f64), or all `length` if full."""
        mf = np.ascontiguousarray(mf_rows, dtype=np.complex128)
        n, L, t = mf.shape[0], self.length, self.t
        b0, s0 = np.empty(n), np.empty(n)
        fb = np.empty((n, L)) if full else None
        fs = np.empty((n, L)) if full else None
        rc = _c().orc_viterbi_run(self._st, t["columns"], t["states"], t["bpc"], _p(t["br_inp"]),
                                  _p(t["br_out"]), _p(t["br_out_idx"]), _p(t["br_start"]),
                                  _p(t["br_end"]), L, _p(mf), ctypes.c_int64(n), _p(b0), _p(s0),
                                  _p(fb) if full else None, _p(fs) if full else None)
        if rc:
            raise KeyError("traceback hit a non-existent branch")
        return (fb, fs) if full else (b0, s0)


def viterbi_detect(mf_rows, length=2, differential=True):
    return ViterbiOracle(length, differential).run(mf_rows)


# ----------------------------------------------------------------------------- K11
def count_errors(det_syms, det_bits, symbols, bits, length, delay=0):
    """examples/soqpsk_detection.py:200-209 -> (sym_errors, bit_errors, min_size)."""
    s = np.asarray(det_syms[length:], dtype=np.int8)
    b = np.asarray(det_bits[length:], dtype=np.uint8)
    ref = symbols[delay:]
    m = min(ref.size, s.size)
    return (int(np.count_nonzero(s[:m] - ref[:m])), int(np.count_nonzero(b[:m] - bits[:m])), m)


def detection_run(bits, g, h, sps, sigma, rng=None, noise=None, detector="PT", timing_offset=-1,
                  length=2, trellis="SOQPSKTrellis4x2DiffEncoded"):
    """The per-waveform body of examples/soqpsk_detection.py:78-216 without plotting.
    Returns dict with every intermediate the parity tests compare."""
    symbols, _, _ = fsm_encode(trellis, bits)
    _t, sig = cpm_modulate(symbols, h, g, sps)
    if noise is None:
        noise = numpy_awgn(sigma, sig.size, rng)
    sig = sig * np.exp(-1j * np.pi / 4)
    r = sig + noise
    mf = pt_bank(r, g, h, sps) if detector == "PT" else pam_bank(r, g, h, sps)
    cols = decimate_columns(r.size, sps, length, timing_offset)
    rows = np.ascontiguousarray(mf[:, cols].T)
    db, ds = viterbi_detect(rows, length, True)
    se, be, m = count_errors(ds, db, symbols, bits, length)
    return dict(symbols=symbols, signal=sig, received=r, mf_rows=rows, det_bits=db, det_syms=ds,
                sym_errors=se, bit_errors=be, compared=m)
