"""Generic CPM trellis detector — ARTM multi-h CPM and PCM/FM.

The reference has no detector for these waveforms: it ships their modulator side
(waveforms/cpm/multih, waveforms/cpm/pcmfm) and the state-space theory
(notes/cpm/cpm.md:52-140, N_S = p M^(L-1)).  This module is the host side of the build's own
detector, in the reference's style: ``CPMTrellisDetector`` mirrors the shape of
``SOQPSKTrellisDetector`` (waveforms/viterbi/algorithm.py:18-101) — matched-filter outputs in,
decisions out, state carried across calls — and the numeric work is two HIP kernels
(csrc/wf_cpm_detect.hip) reached through the C ABI.

Detector = tilted-phase trellis (notes/cpm/cpm.md:100-140): state = (phase index mod NC, the
Lp - 1 previous symbols), matched filters over Lp symbols of the phase pulse (pulse truncation,
as examples/soqpsk_detection.py:134-156 does for SOQPSK), branch increment
Re(rotation * mf) minimised with the reference detector's tie-breaks.

Designs (bit-error rates of the sequential statement of the detector (cpm_oracle.c), 4e6 bits per point
for the first three rows, 4e5 for the last two; the 16-state row is reproduced by the GPU link at
2e8 bits per point: 7.6e-3, 1.9e-3, 3.7e-4, 5.4e-5):

    ARTM multi-h, Eb/N0 (dB)        8        9        10       11
    256 states (Lp 3, NC 16)     5.3e-3   1.3e-3   2.6e-4   4.0e-5     full trellis p M^(L-1)
     64 states (Lp 2, NC 16)     5.8e-3   1.4e-3   2.6e-4   3.7e-5
     16 states (Lp 2, NC  4)     7.5e-3   1.9e-3   3.3e-4   5.6e-5     <- ARTM_16 (BASELINE configs[2]), 0.2 dB
     16 states (Lp 1, NC 16)     4.1e-2   2.0e-2   9.4e-3   3.7e-3     one-symbol filters: unusable
     16 states (Lp 3, NC  1)     6.6e-2   3.5e-2   1.3e-2   3.9e-3     no phase state at all: unusable
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

__all__ = ["CPMDetectorSpec", "ARTM_16", "ARTM_64", "ARTM_256", "PCMFM_10", "CPMTrellisDetector", "matched_filter_templates", "rotation_table", "detector_kernel_name",
           "filter_geometry", "sigma_for_ebn0"]


@dataclass(frozen=True)
class CPMDetectorSpec:
    M: int          # alphabet size, alpha = 2 U - (M - 1)
    p: int          # modulation indices K[i] / p
    K: tuple
    Lp: int         # symbols per matched filter
    NC: int         # phase classes in the trellis state (NC < p: phase index carried per survivor)
    D: int          # decision delay (calls)

    @property
    def nstates(self) -> int:
        return self.NC * self.M ** (self.Lp - 1)

    @property
    def nfilt(self) -> int:
        return self.M ** self.Lp

    @property
    def bits_per_symbol(self) -> int:
        return self.M.bit_length() - 1

    @property
    def mod_index(self) -> np.ndarray:
        return np.asarray(self.K, dtype=np.float64) / self.p

    def c_config(self):
        from waveforms_amd import _hip

        c = _hip.CPMDetectorConfig()
        c.M, c.p, c.nh, c.Lp, c.NC, c.D = self.M, self.p, len(self.K), self.Lp, self.NC, self.D
        for i, k in enumerate(self.K):
            c.K[i] = k
        return c


# MULTIH_IRIG_NUMER / DENOM (waveforms/cpm/multih/pulse_filters.py:7-8), quaternary, 3RC
ARTM_16 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=2, NC=4, D=32)
# ... the same filters (pulse truncated to two symbols) with EVERY phase state of notes/cpm/cpm.md:128-140 in the trellis:
# N_S = p M^(Lp-1) = 64, 0.2 dB ahead of ARTM_16 (table above); lane = state, one wave per detector (csrc/wf_cpm_wide.hip)
ARTM_64 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=2, NC=16, D=32)
# ... and the FULL trellis of notes/cpm/cpm.md:128-140: the 3RC pulse kept to its three symbols, N_S = p M^(L-1) = 256 states,
# 64 matched filters per symbol; thread = state, one workgroup of four waves per detector (csrc/wf_cpm_quad.hip)
ARTM_256 = CPMDetectorSpec(M=4, p=16, K=(4, 5), Lp=3, NC=16, D=32)
# PCMFM_NUMER / DENOM = 7 / 10 (waveforms/cpm/pcmfm/__init__.py:5-6), binary
PCMFM_10 = CPMDetectorSpec(M=2, p=10, K=(7,), Lp=2, NC=5, D=32)


def detector_kernel_name(spec: "CPMDetectorSpec", ncalls: int = 10_000_000, warmup: int = 0, ctx=None, info4: list | None = None,
                         beside: bool = False) -> str:
    """Name (as rocprofv3 prints it) of the kernel ``wf_cpm_viterbi_detect`` runs for ``spec`` on a burst of ``ncalls``
    calls on context ``ctx`` (default: the device's): asked from the library (``wf_cpm_detector_form``), so that bench.py
    and the profile tools price the kernel that really runs.  ``info4`` (a list) receives {form, ring slots, calls per
    chunk, warm-up calls}.  ``beside``: the launch shares the chip with a front end (a pipelined link, ``fuse`` bit 5): the lane
    form then runs its plain instantiation, alone the one that claims a whole SIMD's registers (wf_cpm_lanes.hip, SOLO)."""
    import ctypes

    from waveforms_amd import _hip

    info = (ctypes.c_int * 4)()
    cfg = spec.c_config()
    _hip.check(_hip.lib().wf_cpm_detector_form(ctx if ctx is not None else _hip.ctx(), ctypes.byref(cfg), int(ncalls), int(warmup), info))
    if info4 is not None:
        info4[:] = [int(v) for v in info]
    if info[0] == 1:
        k1 = spec.K[1] if len(spec.K) > 1 else spec.K[0]
        hi = "true" if spec.bits_per_symbol * (spec.D - 1) >= 32 else "false"
        solo = "false" if beside else "true"
        return f"cpm_lane_kernel<lane_spec<{spec.M}, {spec.Lp}, {spec.NC}, {spec.p}, {len(spec.K)}, {spec.K[0]}, {k1}>, {info[1]}, {hi}, {solo}, false>"
    if info[0] == 3:
        return f"cpm_quad_kernel<{spec.M}, {spec.Lp}>"
    if info[0] == 2:
        return f"cpm_wide_kernel<{spec.M}, {spec.Lp}>"
    return f"cpm_viterbi_kernel<{spec.M}, {spec.Lp}>"


def filter_geometry(ntaps: int, sps: int, spec: CPMDetectorSpec, nsym: int) -> dict:
    """Where the matched-filter windows sit.  cpm_modulate (waveforms/cpm/modulate.py:95-99) centres
    symbol m's pulse on sample (m+1) sps, so its phase ramp starts at (m+1) sps - (ntaps-1)//2; a
    filter of Lp symbols covers the middle of the pulse and has sps + 1 taps (both ends, like q_t
    in examples/soqpsk_detection.py:135-140)."""
    offset = ((ntaps - 1) - spec.Lp * sps) // 2
    if offset < -(sps // 2):
        raise ValueError("matched filter longer than the frequency pulse")
    start0 = sps - (ntaps - 1) // 2 + offset
    npts = max((nsym + 1) * sps, ntaps)
    room = npts - (sps + 1) - start0
    ncalls = min(nsym, room // sps + 1) if room >= 0 else 0
    return {"offset": offset, "start0": start0, "ntm": sps + 1, "ncalls": int(ncalls), "npts": int(npts)}


def matched_filter_templates(pulse, sps: int, spec: CPMDetectorSpec) -> np.ndarray:
    """complex128[nh][M^Lp][sps+1]: the phase trajectory of (u_0, u_1, ...) over one symbol time,
    filter index f = u_0 + M u_1 + M^2 u_2, u_i = the symbol i places before the one being
    clocked in; column c serves the symbols with modulation index K[c]/p."""
    g = np.asarray(pulse, dtype=np.float64)
    q = np.cumsum(g) / sps
    nh, M = len(spec.K), spec.M
    off = ((g.size - 1) - spec.Lp * sps) // 2
    k = np.arange(sps + 1)
    digits = (np.arange(spec.nfilt)[:, None] // M ** np.arange(spec.Lp)[None, :]) % M       # [f][i] = u_i
    alpha = 2 * digits - (M - 1)
    out = np.empty((nh, spec.nfilt, sps + 1), dtype=np.complex128)
    for c in range(nh):
        turns = np.zeros((spec.nfilt, sps + 1))
        for i in range(spec.Lp):
            idx = off + k + i * sps
            ramp = np.where(idx < 0, 0.0, q[np.clip(idx, 0, g.size - 1)])
            turns += (spec.K[(c - i) % nh] / spec.p) * alpha[:, i:i + 1] * ramp[None, :]
        out[c] = np.exp(2j * np.pi * turns)
    return out


def rotation_table(spec: CPMDetectorSpec) -> np.ndarray:
    """float64[2p][2] = (cos, sin)(pi r / p): phase state 2 pi v / p minus the phase tilt
    (notes/cpm/cpm.md:108-121), both multiples of pi / p."""
    ang = np.pi * np.arange(2 * spec.p) / spec.p
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))


def sigma_for_ebn0(ebn0_db: float, sps: int, bits_per_symbol: int) -> float:
    """Es/N0 = sps / (2 sigma^2) (examples/soqpsk_detection.py:132), Eb = Es / bits_per_symbol."""
    return float(np.sqrt(sps / (2.0 * bits_per_symbol * 10.0 ** (ebn0_db / 10.0))))


class CPMTrellisDetector:
    """Batch detector, stateful like ``SOQPSKTrellisDetector``: successive ``detect`` calls continue
    one burst through a device-resident carry.  ``detect(rows)`` takes the matched-filter rows of
    consecutive symbols (complex128[n][M^Lp]) and returns the symbols decided by those calls: call
    k of the burst decides symbol k - D + 1 (as U = (alpha + M - 1)/2), so a fresh detector returns
    nothing for its first D - 1 calls."""

    def __init__(self, spec: CPMDetectorSpec = ARTM_16) -> None:
        self.spec = spec
        self.i = 0
        self._cfg = spec.c_config()
        self._d_rot = None
        self._d_state = None
        self._ctx = None

    def __del__(self):
        if getattr(self, "_ctx", None) is not None:
            from waveforms_amd import _hip

            _hip.free_ctx(self._ctx)
            self._ctx = None

    def detect_device(self, rows, warmup: int = 0):
        """``rows``: float64[n, nfilt, 2] device tensor -> uint8[n] device tensor (entry k = the
        decision of call k; entries of calls before the D-th of the burst are 0)."""
        from waveforms_amd import _hip, device as dev

        if self._ctx is None:
            self._ctx = _hip.new_ctx()          # private proof counter, like SOQPSKTrellisDetector
            self._d_rot = _hip.to_device(rotation_table(self.spec))
            self._d_state = _hip.zeros(_hip.WF_CPM_STATE_BYTES // 8, "int64")
        n = int(rows.shape[0])
        if tuple(rows.shape[1:]) != (self.spec.nfilt, 2):
            raise ValueError(f"rows must be [n, {self.spec.nfilt}, 2] float64")
        out = _hip.zeros(n + 16, "uint8")
        # (chunk-parallel; chunks that miss their warm-up are repaired on the device, cascading where needed: the
        #  decisions are the sequential detector's whatever `warmup` is — it only sets how many chunks get repaired)
        # (repairs switched off — tests of the proof — is the one way this call can raise behind a launch: the carry is put back then)
        keep = self._d_state.clone() if _hip.get_option(self._ctx, _hip.WF_OPT_DET_REPAIR) else None
        _hip.check(_hip.lib().wf_cpm_viterbi_detect(self._ctx, ctypes.byref(self._cfg), _hip.ptr(self._d_rot), _hip.ptr(rows), n, int(warmup),
                                                    _hip.ptr(out), _hip.ptr(self._d_state), _hip.stream()))
        unproven = dev.viterbi_unmerged(reset=True, ctx=self._ctx)
        if unproven:        # only with the context's WF_OPT_DET_REPAIR option switched off (tests of the proof itself)
            if keep is not None:
                self._d_state.copy_(keep)
            raise RuntimeError(f"{unproven} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        return out[:n]

    def detect(self, rows, warmup: int = 0) -> np.ndarray:
        from waveforms_amd import _hip

        rows = np.ascontiguousarray(rows, dtype=np.complex128).reshape(-1, self.spec.nfilt)
        dec = _hip.to_host(self.detect_device(_hip.to_device(rows), warmup))
        lo = max(self.spec.D - 1 - self.i, 0)
        self.i += rows.shape[0]
        return dec[lo:].copy()

    def detect_samples(self, received, templates, start0: int, sps: int, ncalls: int, warmup: int = 0) -> np.ndarray:
        """The matched filters and the detector over the received SAMPLES (``wf_cpm_viterbi_detect_samples``: the filters
        run inside the detector, no rows in HBM) — call k's window is ``received[start0 + k sps : start0 + k sps + sps + 1]``
        against ``templates[k % nh]`` (complex128[nh][M^Lp][sps + 1], ``matched_filter_templates``).  Where that launch
        does not serve the configuration (``self.samples_form`` says which path ran) the rows are made by
        ``wf_cpm_mf_rows_c128`` and detected as :meth:`detect` does.  Returns what :meth:`detect` returns for those rows."""
        from waveforms_amd import _hip, device as dev

        r = np.ascontiguousarray(received, dtype=np.complex128).ravel()
        t = np.ascontiguousarray(templates, dtype=np.complex128)
        if self._ctx is None:
            self._ctx = _hip.new_ctx()
            self._d_rot = _hip.to_device(rotation_table(self.spec))
            self._d_state = _hip.zeros(_hip.WF_CPM_STATE_BYTES // 8, "int64")
        d_r, d_t = _hip.to_device(r), _hip.to_device(t)
        out = _hip.zeros(int(ncalls) + 16, "uint8")
        rc = _hip.lib().wf_cpm_viterbi_detect_samples(self._ctx, ctypes.byref(self._cfg), _hip.ptr(self._d_rot), _hip.ptr(d_t), int(t.shape[0]),
                                                      int(t.shape[1]), int(t.shape[2]), _hip.ptr(d_r), int(r.size), int(start0), int(sps),
                                                      int(ncalls), int(warmup), _hip.ptr(out), _hip.ptr(self._d_state), _hip.stream())
        self.samples_form = rc == 0
        if rc == 1:         # not this launch's configuration: rows, then the detector over them
            return self.detect(_hip.to_host(dev.cpm_mf_rows(d_r, d_t, int(start0), int(sps), int(ncalls)), complex_pairs=True), warmup)
        _hip.check(rc)
        unproven = dev.viterbi_unmerged(reset=True, ctx=self._ctx)
        if unproven:
            raise RuntimeError(f"{unproven} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        dec = _hip.to_host(out[:int(ncalls)])
        lo = max(self.spec.D - 1 - self.i, 0)
        self.i += int(ncalls)
        return dec[lo:].copy()
