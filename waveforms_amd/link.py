"""Device-resident SOQPSK link: PRBS -> encode -> modulate -> AWGN -> MF bank -> Viterbi ->
error count in one C-ABI call (``wf_link_run``), all intermediates in one HBM workspace.

This is the unit of work of the benchmark (one step = one block) and of the Monte-Carlo
BER sweep (one trial block per (Eb/N0 point, block index)).  It restates the
per-waveform body of reference examples/soqpsk_detection.py:45-216.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _hip
from .cpm.soqpsk import freq_pulse_soqpsk_tg
from .filters.matched import pam_matched_filter_taps, pt_matched_filter_taps
from .glfsr.pn import generate_mask


def sigma_for_ebn0(ebn0_db: float, sps: int) -> float:
    """Inverse of Eb/N0[dB] = 10 log10(sps / (2 sigma^2)) (examples/soqpsk_detection.py:132)."""
    return math.sqrt(sps / (2.0 * 10.0 ** (ebn0_db / 10.0)))


class SOQPSKLink:
    def __init__(self, nsym: int, sps: int = 8, pulse=None, mod_index: float = 0.25, detector: str = "PT",
                 pn_degree: int = 23, differential: bool = True, timing_offset: int | None = None,
                 warmup: int = 0, fuse: int = 3, private_ctx: bool = False) -> None:
        self.nsym, self.sps = int(nsym), int(sps)
        # a link that runs on its own stream next to other links needs its own scratch
        self._ctx = _hip.new_ctx() if private_ctx else _hip.ctx()
        pulse = freq_pulse_soqpsk_tg(sps) if pulse is None else np.asarray(pulse, dtype=np.float64)
        if detector == "PT":
            taps = pt_matched_filter_taps(pulse, mod_index, sps)
            off = -1 if timing_offset is None else timing_offset
        elif detector == "PAM":
            taps = pam_matched_filter_taps(pulse, mod_index, sps)
            off = 0 if timing_offset is None else timing_offset
        else:
            raise ValueError(f"unknown detector {detector!r}")
        self.detector = detector
        self.pn_degree = pn_degree
        self._d_h = _hip.to_device(np.array([mod_index], dtype=np.float64))
        self._d_pulse = _hip.to_device(pulse)
        self._d_taps = _hip.to_device(np.ascontiguousarray(taps))
        cfg = _hip.LinkConfig()
        cfg.nsym, cfg.sps = self.nsym, self.sps
        cfg.degree, cfg.mask, cfg.state, cfg.skip = pn_degree, generate_mask(pn_degree), (1 << pn_degree) - 1, 0
        cfg.differential = int(differential)
        cfg.d_h, cfg.d_pulse, cfg.ntaps = self._d_h.data_ptr(), self._d_pulse.data_ptr(), pulse.size
        cfg.d_mf_taps, cfg.mf_ntaps, cfg.mf_nfilt = self._d_taps.data_ptr(), taps.shape[1], taps.shape[0]
        cfg.timing_offset, cfg.warmup = off, warmup
        cfg.sigma, cfg.seed, cfg.stream_id = 0.0, 1, 0
        cfg.event_slot = -1
        cfg.fuse = int(fuse)
        self.cfg = cfg
        self.workspace_bytes = _hip.lib().wf_link_workspace_bytes(ctypes.byref(cfg))
        if self.workspace_bytes < 0:
            raise ValueError("invalid link configuration")
        self.workspace = _hip.empty(self.workspace_bytes, "uint8")
        self.counts = _hip.zeros(2, "int64")
        self.compared = 0

    def reset_counts(self) -> None:
        self.counts.zero_()
        self.compared = 0

    STAGES = ("prbs", "encode", "fir", "phase", "awgn", "mfbank", "viterbi", "count")

    def stage_ms(self, event_slot: int) -> dict[str, float]:
        """Per-stage HIP-event times (ms) of the last run that used ``event_slot``."""
        buf = (ctypes.c_float * len(self.STAGES))()
        _hip.check(_hip.lib().wf_link_stage_ms(self._ctx, event_slot, buf))
        return dict(zip(self.STAGES, (float(v) for v in buf)))

    def run_block(self, ebn0_db: float, seed: int = 1, stream_id: int = 0, skip_bits: int = 0,
                  event_slot: int = -1) -> None:
        """Queue one trial block on the current stream; error counts accumulate on the
        device (``self.counts``), the compared-symbol count on the host."""
        c = self.cfg
        c.sigma, c.seed, c.stream_id, c.skip = sigma_for_ebn0(ebn0_db, self.sps), seed, stream_id, skip_bits
        c.event_slot = event_slot
        m = ctypes.c_int64(0)
        _hip.check(_hip.lib().wf_link_run(self._ctx, ctypes.byref(c), self.workspace.data_ptr(),
                                          self.workspace_bytes, self.counts.data_ptr(), ctypes.byref(m),
                                          _hip.stream()))
        self.compared += m.value

    def result(self) -> tuple[int, int, int]:
        """(symbol errors, bit errors, symbols compared) — synchronises."""
        _hip.check(_hip.lib().wf_ctx_check(self._ctx, _hip.stream()))
        se, be = (int(v) for v in self.counts.cpu().tolist())
        return se, be, self.compared
