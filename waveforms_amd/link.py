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
from .filters.matched import pack_bank_factors, pam_bank_factors, pam_matched_filter_taps, pt_matched_filter_taps
from .glfsr.pn import generate_mask


def sigma_for_ebn0(ebn0_db: float, sps: int) -> float:
    """Inverse of Eb/N0[dB] = 10 log10(sps / (2 sigma^2)) (examples/soqpsk_detection.py:132)."""
    return math.sqrt(sps / (2.0 * 10.0 ** (ebn0_db / 10.0)))


def operating_point_warmup(waveform: str, ebn0_db: float | None) -> int:
    """Detector chunk warm-up in ROWS (= detector calls run before a chunk's own first call) for a link at a
    known operating point; 0 = the library default.  A matter of SPEED only: every launch proves on the device
    that each chunk started from the sequential detector's state and runs the chunks that did not again from the
    true state, cascading into the following chunks where needed (wf_viterbi_repaired / wf_viterbi_cascaded), so
    decisions and counts are the sequential detector's at any warm-up and any Eb/N0 — a short warm-up just moves
    work from every chunk's warm-up to the repair of the few that needed more.  One table for bench.py and the
    timing tools (tools/link_stage_time.py, tools/stream_bench.py), from the failure-rate scans
    (tools/warmup_scan.py, tools/cpm_warmup_scan.py; profiles/r02_warmup_scan.json, r03_cpm_repair_scan_*.json,
    r04_lowsnr_scan.log): SOQPSK 4-state 16 rows from 6 dB up; ARTM 16-state 48 calls from 8 dB up (0.3 / 0.05 /
    0.01 % of the chunks repaired at 8 / 10 / 12 dB), 64 from 6 dB; binary PCM/FM 64 from 3 dB up (1.2 - 3 %
    repaired; below, the lane form's 320-call chunks hand on often enough for the default to be faster)."""
    if ebn0_db is None:
        return 0
    if waveform == "soqpsk":
        return 16 if ebn0_db >= 6.0 else 0
    if waveform == "multih":
        return 48 if ebn0_db >= 8.0 else (64 if ebn0_db >= 6.0 else 0)
    return 64 if ebn0_db >= 3.0 else 0


def soqpsk_warmup_param(rows: int) -> int:
    """``wf_link_config.warmup`` of the SOQPSK link counts the rows AFTER the priming row (rows - 1; 0 = default)."""
    return max(int(rows) - 1, 0)


class SOQPSKLink:
    def __init__(self, nsym: int, sps: int = 8, **kw) -> None:
        self._configure(nsym, sps, **kw)
        self.workspace_bytes = _hip.lib().wf_link_workspace_bytes(ctypes.byref(self.cfg))
        if self.workspace_bytes < 0:
            raise ValueError("invalid link configuration")
        self.workspace = _hip.empty(self.workspace_bytes, "uint8")
        self.counts = _hip.zeros(2, "int64")
        self.compared = 0

    def _configure(self, nsym: int, sps: int = 8, pulse=None, mod_index: float = 0.25, detector: str = "PT",
                   pn_degree: int = 23, differential: bool = True, timing_offset: int | None = None,
                   warmup: int = 0, fuse: int = 15, private_ctx: bool = False, factor_bank: bool = True) -> None:
        self.nsym, self.sps = int(nsym), int(sps)
        # a link that runs on its own stream next to other links needs its own scratch
        self._ctx = _hip.new_ctx() if private_ctx else _hip.ctx()
        _hip.check(_hip.lib().wf_ctx_forget_promises(self._ctx))    # (this link's tables may sit where a dead link's did: every promise is checked afresh)
        self._owns_ctx = bool(private_ctx)
        pulse = freq_pulse_soqpsk_tg(sps) if pulse is None else np.asarray(pulse, dtype=np.float64)
        factors = None
        if detector == "PT":
            taps = pt_matched_filter_taps(pulse, mod_index, sps)
            off = -1 if timing_offset is None else timing_offset
        elif detector == "PAM":
            taps = pam_matched_filter_taps(pulse, mod_index, sps)
            off = 0 if timing_offset is None else timing_offset
            if factor_bank:
                # the bank as the reference computes it: two real rho filters, outputs weighted by conj(pseudo symbols)
                # (examples/soqpsk_detection.py:158-173) — the one-kernel front end then runs two real filters, not three complex ones
                factors = pam_bank_factors(pulse, mod_index, sps)
                if np.abs(factors[1] @ factors[0] - taps).max() > 1e-12 * np.abs(taps).max():
                    factors = None
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
        self._d_factor = _hip.to_device(pack_bank_factors(*factors)) if factors is not None else None
        cfg.d_mf_factor = self._d_factor.data_ptr() if self._d_factor is not None else None
        self.cfg = cfg

    def __del__(self):
        try:
            if getattr(self, "cfg", None) is not None and (self.cfg.fuse & 32) and getattr(self, "workspace", None) is not None:
                # back ends of the last blocks may still be reading the workspace on the side stream
                _hip.lib().wf_link_join(self._ctx, _hip.stream())
                _hip.torch().cuda.current_stream().synchronize()
        except Exception:          # noqa: BLE001 — interpreter teardown
            pass
        if getattr(self, "_owns_ctx", False):
            _hip.free_ctx(self._ctx)
            self._owns_ctx = False

    @property
    def row_bytes(self) -> int:
        """Bytes per matched-filter row in the workspace: 48 (3 complex128), or 32 when fuse bit 2 is in effect
        (detector-packed rows: 3-filter bank with the fused channel at 8 samples per symbol, or the one-kernel
        front end at 8 / 10 / 20)."""
        return self.layout()["row_bytes"]            # the library's own rule (wf_link_layout)

    def layout(self) -> dict:
        """Byte offsets of the intermediates inside ``self.workspace``."""
        info = (ctypes.c_int64 * 8)()
        _hip.check(_hip.lib().wf_link_layout(ctypes.byref(self.cfg), info))
        keys = ("calls", "one_kernel_front_end", "off_bits", "off_syms", "off_signal", "row_bytes", "signal_len", "off_mf")
        return dict(zip(keys, (int(v) for v in info)))

    def reset_counts(self) -> None:
        # (fuse bit 5: earlier blocks' counters may still be running on the context's side stream)
        _hip.check(_hip.lib().wf_link_join(self._ctx, _hip.stream()))
        self.counts.zero_()
        self.compared = 0

    STAGES = ("prbs", "encode", "fir", "phase", "awgn", "mfbank", "viterbi", "count")

    def stage_ms(self, event_slot: int) -> dict[str, float]:
        """Per-stage HIP-event times (ms) of the last run that used ``event_slot``."""
        buf = (ctypes.c_float * len(self.STAGES))()
        _hip.check(_hip.lib().wf_link_stage_ms(self._ctx, event_slot, buf))
        d = dict(zip(self.STAGES, (float(v) for v in buf)))
        if self.prologue_ahead:
            # (wf_link_run, round 6: the prologue of such a block runs on its own stream beside the previous front end; the event
            #  pair of the "fir" slot then brackets the WAIT between the prologue's end and the main kernel's start, and the main
            #  kernel sits in the "phase" slot — reported here under the name every other form uses)
            d["fir"], d["phase"] = d["phase"], 0.0
        return d

    @property
    def prologue_ahead(self) -> bool:
        """A pipelined link (``fuse`` bit 5 with the one-kernel front end) whose context runs each block's prologue ahead on its
        own stream (``WF_OPT_PIPE_RESERVE_CUS`` other than 0: a measured alternative, not the default)."""
        return bool(self.cfg.fuse & 32) and bool(self.layout()["one_kernel_front_end"]) and _hip.get_option(self._ctx, _hip.WF_OPT_PIPE_RESERVE_CUS) != 0

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
        """(symbol errors, bit errors, symbols compared) — synchronises.  The detector's chunks are proven (and, where
        a warm-up fell short, repaired) on the device, so this never depends on ``warmup`` or Eb/N0; it raises only
        if a chunk was left unproven because the context's WF_OPT_DET_REPAIR option switched the repairs off."""
        _hip.check(_hip.lib().wf_ctx_check(self._ctx, _hip.stream()))
        from waveforms_amd import device as dev

        unmerged = dev.viterbi_unmerged(reset=True, ctx=self._ctx)
        if unmerged:
            raise RuntimeError(f"{unmerged} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        se, be = (int(v) for v in self.counts.cpu().tolist())
        return se, be, self.compared


class SOQPSKStream:
    """The same chain over a continuous stream of ``total_symbols`` symbols, processed in
    chunks of ``chunk_symbols`` detector calls (``wf_link_stream_chunk``; BASELINE config 5).
    Neighbouring context is re-generated as a halo or carried in a 512-byte device block;
    decisions and error counts equal a one-shot :class:`SOQPSKLink` over the whole stream,
    while HBM use is that of one chunk."""

    @property
    def row_bytes(self) -> int:
        """Bytes per matched-filter row of a chunk: 32 (detector-packed) with fuse bits 1 + 2 at 8 samples per
        symbol, else 48 (the streaming windows of the one-kernel front end are an sps-8 path)."""
        c = self.cfg
        return 32 if (c.fuse & 4) and (c.fuse & 2) and c.sps == 8 and c.mf_nfilt == 3 else 16 * c.mf_nfilt

    def __init__(self, total_symbols: int, chunk_symbols: int, sps: int = 8, **kw) -> None:
        # reuse the link's configuration (taps, pulse, PRBS ...) without its one-shot workspace
        self._proto = SOQPSKLink.__new__(SOQPSKLink)
        SOQPSKLink._configure(self._proto, int(total_symbols), sps, **kw)
        self.cfg, self._ctx, self.sps = self._proto.cfg, self._proto._ctx, int(sps)
        self.total_symbols, self.chunk_symbols = int(total_symbols), int(chunk_symbols)
        lib = _hip.lib()
        nbytes = lib.wf_link_stream_workspace_bytes(ctypes.byref(self.cfg), self.chunk_symbols)
        if nbytes < 0:
            tl, spt, nt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            lib.wf_mod_tile_geometry(self.sps, self.cfg.ntaps, self.total_symbols, ctypes.byref(tl), ctypes.byref(spt),
                                     ctypes.byref(nt))
            raise ValueError(f"chunk_symbols must be a multiple of {spt.value} (one modulator tile) and of 128, and at "
                             f"least {4 * (spt.value + 48)}")
        self.workspace_bytes = nbytes
        self.workspace = _hip.empty(nbytes, "uint8")
        self.state = _hip.zeros(64, "float64")          # WF_LINK_STREAM_STATE_BYTES = 512
        self.counts = _hip.zeros(2, "int64")
        self.compared = 0
        info = (ctypes.c_int64 * 8)()
        _hip.check(lib.wf_link_layout(ctypes.byref(self.cfg), info))
        self.total_calls = int(info[0])
        self.nchunks = -(-self.total_calls // self.chunk_symbols)

    def reset(self) -> None:
        self.state.zero_()
        self.counts.zero_()
        self.compared = 0

    def chunk_info(self, c: int) -> dict:
        info = (ctypes.c_int64 * 8)()
        _hip.check(_hip.lib().wf_link_stream_layout(ctypes.byref(self.cfg), self.chunk_symbols, c, info))
        keys = ("calls", "first_call", "off_bits", "off_syms", "off_signal", "signal_origin", "signal_len", "off_mf")
        return dict(zip(keys, (int(v) for v in info)))

    def run_chunk(self, c: int, ebn0_db: float, seed: int = 1, stream_id: int = 0) -> None:
        cfg = self.cfg
        cfg.sigma, cfg.seed, cfg.stream_id, cfg.event_slot = sigma_for_ebn0(ebn0_db, self.sps), seed, stream_id, -1
        m = ctypes.c_int64(0)
        _hip.check(_hip.lib().wf_link_stream_chunk(self._ctx, ctypes.byref(cfg), self.chunk_symbols, c,
                                                   self.state.data_ptr(), self.workspace.data_ptr(), self.workspace_bytes,
                                                   self.counts.data_ptr(), ctypes.byref(m), _hip.stream()))
        self.compared += m.value

    def run(self, ebn0_db: float, seed: int = 1, stream_id: int = 0) -> tuple[int, int, int]:
        """All chunks in order; returns (symbol errors, bit errors, symbols compared)."""
        self.reset()
        for c in range(self.nchunks):
            self.run_chunk(c, ebn0_db, seed, stream_id)
        return self.result()

    def run_pipelined(self, ebn0_db: float, seed: int = 1, stream_id: int = 0) -> tuple[int, int, int]:
        """Like :meth:`run` with consecutive chunks alternating between two HIP streams, each with its own
        workspace and library context.  A chunk is issued as three parts (``wf_link_stream_chunk_phase``)
        and each part waits only for the same part of the previous chunk, which is where its carry comes
        from: PRBS / encoder / modulator carries (encoder state, modulator phase), the modulator + channel
        + bank kernel, detector + error count (detector state).  So the small serial kernels of chunk c + 1
        and the detector of chunk c run beside a main kernel instead of between two.  Same decisions and
        counts as :meth:`run`."""
        torch = _hip.torch()
        self.reset()
        cfg = self.cfg
        cfg.sigma, cfg.seed, cfg.stream_id, cfg.event_slot = sigma_for_ebn0(ebn0_db, self.sps), seed, stream_id, -1
        if getattr(self, "_ws2", None) is None:
            self._ws2 = _hip.empty(self.workspace_bytes, "uint8")
            self._ctx2 = _hip.new_ctx()
            self._lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        main = torch.cuda.current_stream()
        for lane in self._lanes:
            lane.wait_stream(main)                       # the carry block / counters were zeroed on the caller's stream
        lib, m = _hip.lib(), ctypes.c_int64(0)
        done = {1: None, 4: None, 2: None}
        for c in range(self.nchunks):
            lane, ws, ctx = self._lanes[c & 1], (self.workspace, self._ws2)[c & 1], (self._ctx, self._ctx2)[c & 1]
            with torch.cuda.stream(lane):
                for part in (1, 4, 2):
                    if done[part] is not None:
                        lane.wait_event(done[part])
                    _hip.check(lib.wf_link_stream_chunk_phase(ctx, ctypes.byref(cfg), self.chunk_symbols, c, self.state.data_ptr(),
                                                              ws.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                              ctypes.byref(m), part, lane.cuda_stream))
                    done[part] = torch.cuda.Event()
                    done[part].record(lane)
                self.compared += m.value
        for lane in self._lanes:
            main.wait_stream(lane)
        from waveforms_amd import device as dev

        _hip.check(lib.wf_ctx_check(self._ctx2, _hip.stream()))
        late = dev.viterbi_unmerged(reset=True, ctx=self._ctx2)     # the first context's count is read by result()
        if late:
            dev.viterbi_unmerged(reset=True, ctx=self._ctx)
            raise RuntimeError(f"{late} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        return self.result()

    def __del__(self):
        if getattr(self, "_ctx2", None):
            _hip.free_ctx(self._ctx2)
            self._ctx2 = None

    def interior_chunks(self) -> int:
        """Number of consecutive chunks 1, 2, ... that issue exactly the launches of chunk 1."""
        n = 0
        while 1 + n < self.nchunks and _hip.lib().wf_link_stream_interior(ctypes.byref(self.cfg), self.chunk_symbols, 1 + n):
            n += 1
        return n

    def run_graph(self, ebn0_db: float, seed: int = 1, stream_id: int = 0) -> tuple[int, int, int]:
        """Like :meth:`run`, with the steady state as ONE captured hipGraph replayed per interior
        chunk (the PRBS position and the noise counter live in the device carry block)."""
        torch = _hip.torch()
        self.reset()
        cfg = self.cfg
        cfg.sigma, cfg.seed, cfg.stream_id, cfg.event_slot = sigma_for_ebn0(ebn0_db, self.sps), seed, stream_id, -1
        self.run_chunk(0, ebn0_db, seed, stream_id)
        n_int = self.interior_chunks()
        if n_int:
            m = ctypes.c_int64(0)

            def steady():
                _hip.check(_hip.lib().wf_link_stream_steady(self._ctx, ctypes.byref(cfg), self.chunk_symbols,
                                                            self.state.data_ptr(), self.workspace.data_ptr(),
                                                            self.workspace_bytes, self.counts.data_ptr(), ctypes.byref(m),
                                                            _hip.stream()))

            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                steady()
            for _ in range(n_int):
                graph.replay()
            self.compared += n_int * m.value
            self.graph_replays = n_int
        for c in range(1 + n_int, self.nchunks):
            self.run_chunk(c, ebn0_db, seed, stream_id)
        return self.result()

    def run_graph_pipelined(self, ebn0_db: float, seed: int = 1, stream_id: int = 0, chunks_per_graph: int = 8) -> tuple[int, int, int]:
        """The two-stream chunk pipeline of :meth:`run_pipelined` with its steady state captured as ONE hipGraph:
        ``chunks_per_graph`` consecutive interior chunks, alternating between two streams (workspace and wf_ctx
        each), every part (1: PRBS / encoder / carries, 4: modulator + channel + bank, 2: detector + count) ordered
        behind the same part of the previous chunk by an event inside the capture
        (``wf_link_stream_steady_phase``: each part advances its own position word on the device)."""
        torch = _hip.torch()
        self.reset()
        cfg = self.cfg
        cfg.sigma, cfg.seed, cfg.stream_id, cfg.event_slot = sigma_for_ebn0(ebn0_db, self.sps), seed, stream_id, -1
        if not hasattr(self, "_ws2"):
            self._ws2 = _hip.empty(self.workspace_bytes, "uint8")
            self._ctx2 = _hip.new_ctx()
            self._lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        lib, m = _hip.lib(), ctypes.c_int64(0)
        K = max(2, int(chunks_per_graph) // 2 * 2)
        n_int = self.interior_chunks()
        self.run_chunk(0, ebn0_db, seed, stream_id)                 # context 1: first use outside any capture
        done = 1
        self.graph_replays = 0
        if n_int >= 1 + K:
            # chunk 1 = the first steady call, eagerly on context 2 (its first use; tables, scratch)
            _hip.check(lib.wf_link_stream_steady_phase(self._ctx2, ctypes.byref(cfg), self.chunk_symbols, self.state.data_ptr(),
                                                       self._ws2.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                       ctypes.byref(m), 7, _hip.stream()))
            self.compared += m.value
            done = 2
            torch.cuda.current_stream().synchronize()
            key = (cfg.sigma, seed, stream_id, K)               # what the captured launches have baked in
            cached = getattr(self, "_gp_cache", None)
            graph = cached[1] if cached and cached[0] == key else None
            if graph is not None:
                m.value = cached[2]
            else:
              graph = torch.cuda.CUDAGraph()
              with torch.cuda.graph(graph):
                  a, b = torch.cuda.current_stream(), self._lanes[1]
                  b.wait_stream(a)                                    # fork the second lane into the capture
                  ev = {1: None, 4: None, 2: None}
                  for i in range(K):
                      lane, ws, ctx = (a, self.workspace, self._ctx) if i % 2 == 0 else (b, self._ws2, self._ctx2)
                      with torch.cuda.stream(lane):
                          for part in (1, 4, 2):
                              if ev[part] is not None:
                                  lane.wait_event(ev[part])
                              _hip.check(lib.wf_link_stream_steady_phase(ctx, ctypes.byref(cfg), self.chunk_symbols, self.state.data_ptr(),
                                                                         ws.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                                         ctypes.byref(m), part, lane.cuda_stream))
                              ev[part] = torch.cuda.Event()
                              ev[part].record(lane)
                  a.wait_stream(b)                                    # join
              self._gp_cache = (key, graph, m.value)
            per_chunk = m.value
            while done + K <= 1 + n_int:
                graph.replay()
                self.compared += K * per_chunk
                done += K
                self.graph_replays += 1
        for c in range(done, self.nchunks):
            if c <= n_int and done > 1:                             # a leftover interior chunk: the position words are live
                _hip.check(lib.wf_link_stream_steady_phase(self._ctx, ctypes.byref(cfg), self.chunk_symbols, self.state.data_ptr(),
                                                           self.workspace.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                           ctypes.byref(m), 7, _hip.stream()))
                self.compared += m.value
            else:
                self.run_chunk(c, ebn0_db, seed, stream_id)
        from waveforms_amd import device as dev

        _hip.check(lib.wf_ctx_check(self._ctx2, _hip.stream()))
        late = dev.viterbi_unmerged(reset=True, ctx=self._ctx2)
        if late:
            dev.viterbi_unmerged(reset=True, ctx=self._ctx)
            raise RuntimeError(f"{late} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        return self.result()

    def result(self) -> tuple[int, int, int]:
        """Like :meth:`SOQPSKLink.result`."""
        return SOQPSKLink.result(self)


class CPMLink:
    """Device-resident link for the waveforms served by the generic CPM trellis detector
    (``wf_cpm_link_run``): PRBS -> mapper -> cpm_modulate -> AWGN -> matched-filter rows ->
    detector -> error count.  ``waveform``: "multih" (ARTM multi-h CPM, BASELINE configs[2]) or
    "pcmfm"."""

    STAGES = ("prbs", "map", "modulate", "-", "awgn", "mfbank", "viterbi", "count")

    def __init__(self, nsym: int, sps: int = 8, waveform: str = "multih", spec=None, pn_degree: int = 23, warmup: int = 0,
                 skip_head: int = 64, private_ctx: bool = False, fuse: int = 10, _no_workspace: bool = False,
                 paired_templates: bool = True) -> None:
        from .viterbi import cpm

        if waveform == "multih":
            from .cpm.multih import freq_pulse_multih_irig

            pulse, kind, spec = freq_pulse_multih_irig(sps), 1, spec or cpm.ARTM_16
        elif waveform == "pcmfm":
            from .cpm.pcmfm import freq_pulse_pcmfm

            pulse, kind, spec = freq_pulse_pcmfm(sps), 2, spec or cpm.PCMFM_10
        else:
            raise ValueError(f"unknown waveform {waveform!r}")
        self.nsym, self.sps, self.spec, self.waveform = int(nsym), int(sps), spec, waveform
        self._ctx = _hip.new_ctx() if private_ctx else _hip.ctx()
        _hip.check(_hip.lib().wf_ctx_forget_promises(self._ctx))    # (this link's tables may sit where a dead link's did: every promise is checked afresh)
        self._owns_ctx = bool(private_ctx)
        self._d_h = _hip.to_device(spec.mod_index)
        self._d_pulse = _hip.to_device(np.asarray(pulse, dtype=np.float64))
        templates = cpm.matched_filter_templates(pulse, sps, spec)
        self._d_templates = _hip.to_device(templates)
        # A symmetric alphabet makes the templates of a symbol pattern and of its negation exact conjugates (filter f and
        # nfilt - 1 - f): the one-kernel front end then forms each pair of filters from four real sums (wf_cpm_link_config.fuse
        # bit 6, which vouches for exactly this identity — checked here, bit for bit)
        self.paired_templates = bool(paired_templates and templates.shape[1] in (4, 16) and np.array_equal(templates[:, ::-1, :], np.conj(templates)))
        if self.paired_templates:
            fuse = int(fuse) | 64
        self._d_rot = _hip.to_device(cpm.rotation_table(spec))
        cfg = _hip.CPMLinkConfig()
        cfg.nsym, cfg.sps = self.nsym, self.sps
        cfg.degree, cfg.mask, cfg.state, cfg.skip = pn_degree, generate_mask(pn_degree), (1 << pn_degree) - 1, 0
        cfg.mapper_kind, cfg.det = kind, spec.c_config()
        cfg.d_h, cfg.d_pulse, cfg.ntaps = self._d_h.data_ptr(), self._d_pulse.data_ptr(), int(np.asarray(pulse).size)
        cfg.d_templates, cfg.d_rot_cs = self._d_templates.data_ptr(), self._d_rot.data_ptr()
        cfg.sigma, cfg.seed, cfg.stream_id = 0.0, 1, 0
        cfg.warmup, cfg.skip_head, cfg.event_slot, cfg.fuse = warmup, skip_head, -1, int(fuse)
        self.cfg = cfg
        if _no_workspace:        # (CPMStream: the one-shot workspace of a long stream is exactly what it avoids)
            return
        self.workspace_bytes = _hip.lib().wf_cpm_link_workspace_bytes(ctypes.byref(cfg))
        if self.workspace_bytes < 0:
            raise ValueError("invalid link configuration")
        self.workspace = _hip.empty(self.workspace_bytes, "uint8")
        self.counts = _hip.zeros(2, "int64")
        self.compared = 0

    def __del__(self):
        try:
            if getattr(self, "cfg", None) is not None and (self.cfg.fuse & 32) and getattr(self, "workspace", None) is not None:
                _hip.lib().wf_link_join(self._ctx, _hip.stream())
                _hip.torch().cuda.current_stream().synchronize()
        except Exception:          # noqa: BLE001 — interpreter teardown
            pass
        if getattr(self, "_owns_ctx", False):
            _hip.free_ctx(self._ctx)
            self._owns_ctx = False

    def layout(self) -> dict:
        info = (ctypes.c_int64 * 8)()
        _hip.check(_hip.lib().wf_cpm_link_layout(ctypes.byref(self.cfg), info))
        keys = ("calls", "start0", "off_decisions", "off_syms", "off_signal", "one_kernel_front_end", "signal_len", "off_rows")
        return dict(zip(keys, (int(v) for v in info)))

    def reset_counts(self) -> None:
        _hip.check(_hip.lib().wf_link_join(self._ctx, _hip.stream()))   # (fuse bit 5: counters of earlier blocks on the side stream)
        self.counts.zero_()
        self.compared = 0

    def stage_ms(self, event_slot: int) -> dict[str, float]:
        buf = (ctypes.c_float * len(self.STAGES))()
        _hip.check(_hip.lib().wf_link_stage_ms(self._ctx, event_slot, buf))
        return {k: float(v) for k, v in zip(self.STAGES, buf) if k != "-"}

    def run_block(self, ebn0_db: float | None, seed: int = 1, stream_id: int = 0, skip_bits: int = 0, event_slot: int = -1) -> None:
        """Queue one trial block; ``ebn0_db=None`` = no noise."""
        from .viterbi.cpm import sigma_for_ebn0 as cpm_sigma

        c = self.cfg
        c.sigma = 0.0 if ebn0_db is None else cpm_sigma(ebn0_db, self.sps, self.spec.bits_per_symbol)
        c.seed, c.stream_id, c.skip, c.event_slot = seed, stream_id, skip_bits, event_slot
        m = ctypes.c_int64(0)
        _hip.check(_hip.lib().wf_cpm_link_run(self._ctx, ctypes.byref(c), self.workspace.data_ptr(), self.workspace_bytes,
                                              self.counts.data_ptr(), ctypes.byref(m), _hip.stream()))
        self.compared += m.value

    def result(self) -> tuple[int, int, int]:
        """(symbol errors, bit errors, symbols compared), like :meth:`SOQPSKLink.result`."""
        return SOQPSKLink.result(self)


class CPMStream:
    """The CPM link (:class:`CPMLink`) over a continuous stream of ``total_symbols`` symbols in chunks of
    ``chunk_symbols`` detector calls (``wf_cpm_link_stream_chunk``): HBM use of one chunk, the detector state
    and the modulator phase carried on the device, everything else re-generated as a halo.  Decisions and
    error counts equal a one-shot :class:`CPMLink` over the whole stream."""

    def __init__(self, total_symbols: int, chunk_symbols: int, sps: int = 8, **kw) -> None:
        self._proto = CPMLink.__new__(CPMLink)
        CPMLink.__init__(self._proto, int(total_symbols), sps, _no_workspace=True, **kw)
        self.cfg, self._ctx, self.sps, self.spec = self._proto.cfg, self._proto._ctx, int(sps), self._proto.spec
        self.total_symbols, self.chunk_symbols = int(total_symbols), int(chunk_symbols)
        lib = _hip.lib()
        nbytes = lib.wf_cpm_link_stream_workspace_bytes(ctypes.byref(self.cfg), self.chunk_symbols)
        if nbytes < 0:
            tl, spt, nt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            lib.wf_mod_tile_geometry(self.sps, self.cfg.ntaps, self.total_symbols, ctypes.byref(tl), ctypes.byref(spt), ctypes.byref(nt))
            raise ValueError(f"chunk_symbols must be a multiple of {spt.value} (one modulator tile) and of 128 and at least "
                             f"{4 * (spt.value + 48 + self.spec.D)}, with a configuration the one-kernel front end takes")
        self.workspace_bytes = nbytes
        self.workspace = _hip.empty(nbytes, "uint8")
        self.state = _hip.zeros(_hip.WF_CPM_STREAM_STATE_BYTES // 8, "int64")
        self.counts = _hip.zeros(2, "int64")
        self.compared = 0
        self.total_calls = self.chunk_info(0)["stream_calls"]
        self.nchunks = -(-self.total_calls // self.chunk_symbols)

    def reset(self) -> None:
        self.state.zero_()
        self.counts.zero_()
        self.compared = 0

    def chunk_info(self, c: int) -> dict:
        info = (ctypes.c_int64 * 8)()
        _hip.check(_hip.lib().wf_cpm_link_stream_layout(ctypes.byref(self.cfg), self.chunk_symbols, c, info))
        keys = ("calls", "first_call", "off_decisions", "off_syms", "syms_origin", "stream_calls", "sym_per_tile", "off_rows")
        return dict(zip(keys, (int(v) for v in info)))

    def run_chunk(self, c: int, ebn0_db: float | None, seed: int = 1, stream_id: int = 0) -> None:
        from .viterbi.cpm import sigma_for_ebn0 as cpm_sigma

        cfg = self.cfg
        cfg.sigma = 0.0 if ebn0_db is None else cpm_sigma(ebn0_db, self.sps, self.spec.bits_per_symbol)
        cfg.seed, cfg.stream_id, cfg.event_slot = seed, stream_id, -1
        m = ctypes.c_int64(0)
        _hip.check(_hip.lib().wf_cpm_link_stream_chunk(self._ctx, ctypes.byref(cfg), self.chunk_symbols, c, self.state.data_ptr(),
                                                       self.workspace.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                       ctypes.byref(m), _hip.stream()))
        self.compared += m.value

    def run(self, ebn0_db: float | None, seed: int = 1, stream_id: int = 0) -> tuple[int, int, int]:
        self.reset()
        for c in range(self.nchunks):
            self.run_chunk(c, ebn0_db, seed, stream_id)
        return self.result()

    def run_pipelined(self, ebn0_db: float | None, seed: int = 1, stream_id: int = 0) -> tuple[int, int, int]:
        """Like :meth:`run` with consecutive chunks alternating between two HIP streams, each with its own workspace and
        library context (``wf_cpm_link_stream_chunk_phase``, as :meth:`SOQPSKStream.run_pipelined` does it): PRBS + mapper
        depend on nothing, a chunk's front end waits for the previous chunk's front end (phase carry), its detector for
        the previous chunk's detector (detector carry) — so the detector of chunk c runs beside the front end of chunk
        c + 1.  Same decisions and counts as :meth:`run`."""
        from .viterbi.cpm import sigma_for_ebn0 as cpm_sigma

        torch = _hip.torch()
        self.reset()
        cfg = self.cfg
        cfg.sigma = 0.0 if ebn0_db is None else cpm_sigma(ebn0_db, self.sps, self.spec.bits_per_symbol)
        cfg.seed, cfg.stream_id, cfg.event_slot = seed, stream_id, -1
        if getattr(self, "_ws2", None) is None:
            self._ws2 = _hip.empty(self.workspace_bytes, "uint8")
            self._ctx2 = _hip.new_ctx()
            self._lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        main = torch.cuda.current_stream()
        for lane in self._lanes:
            lane.wait_stream(main)                       # the carry block / counters were zeroed on the caller's stream
        lib, m = _hip.lib(), ctypes.c_int64(0)
        done = {4: None, 2: None}
        for c in range(self.nchunks):
            lane, ws, ctx = self._lanes[c & 1], (self.workspace, self._ws2)[c & 1], (self._ctx, self._ctx2)[c & 1]
            with torch.cuda.stream(lane):
                for part in (1, 4, 2):
                    if part != 1 and done[part] is not None:
                        lane.wait_event(done[part])
                    _hip.check(lib.wf_cpm_link_stream_chunk_phase(ctx, ctypes.byref(cfg), self.chunk_symbols, c, self.state.data_ptr(),
                                                                  ws.data_ptr(), self.workspace_bytes, self.counts.data_ptr(),
                                                                  ctypes.byref(m), part, lane.cuda_stream))
                    if part != 1:
                        done[part] = torch.cuda.Event()
                        done[part].record(lane)
                self.compared += m.value
        for lane in self._lanes:
            main.wait_stream(lane)
        from waveforms_amd import device as dev

        _hip.check(lib.wf_ctx_check(self._ctx2, _hip.stream()))
        late = dev.viterbi_unmerged(reset=True, ctx=self._ctx2)     # the first context's count is read by result()
        if late:
            dev.viterbi_unmerged(reset=True, ctx=self._ctx)
            raise RuntimeError(f"{late} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        return self.result()

    def __del__(self):
        if getattr(self, "_ctx2", None):
            _hip.free_ctx(self._ctx2)
            self._ctx2 = None

    def result(self) -> tuple[int, int, int]:
        """(symbol errors, bit errors, symbols compared), like :meth:`SOQPSKLink.result`."""
        return SOQPSKLink.result(self)
