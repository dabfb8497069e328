"""SOQPSK trellis detector — API of reference waveforms/viterbi/algorithm.py:18-101.

``iteration`` is the reference's per-symbol call: one literal window recomputation on
the GPU with the detector state resident in HBM (csrc/wf_viterbi.hip,
viterbi_iteration_kernel), any window ``length``.  ``detect`` is the batch form the
MI355X path is built for: a whole burst of matched-filter rows in one launch
(chunk-parallel ACS + traceback), any ``length`` from 1 to 64 like ``iteration`` (length 2: the kernel the link
runs; the others: a window kernel with ``length`` - 1 stages of look-ahead per row), returning for
every row exactly what ``iteration(row)[...][0]`` would — for odd lengths too, where the reference pairs a
row's increments with the other trellis section's branches (algorithm.py:57-63 against :69-87): restated
literally; it is stateful like ``iteration`` (successive calls continue the same burst through a device-resident carry).
"""
from __future__ import annotations

from typing import TYPE_CHECKING

import numpy as np

from waveforms_amd.cpm.trellis.model import (
    FiniteStateMachine,
    SOQPSKTrellis4x2,
    SOQPSKTrellis4x2DiffEncoded,
)

if TYPE_CHECKING:
    from numpy.typing import NDArray


class SOQPSKTrellisDetector:
    def __init__(
        self,
        length: int = 2,
        *,
        differantial_encoding: bool = True,  # (sic) the reference's keyword
    ) -> None:
        self.i = 0
        self.length = length
        self.differential = bool(differantial_encoding)
        self.fsm = FiniteStateMachine(
            trellis=SOQPSKTrellis4x2DiffEncoded if differantial_encoding else SOQPSKTrellis4x2,
        )
        self.state_exp_term = [+1j, -1, +1, -1j]
        self._d_state = None
        self._d_io = None
        self._d_carry = None      # batch API: [i, metrics[4], increments[8], ...] on the device
        self._ctx = None          # batch API: private wf_ctx (own proof counter), created on first use
        self._mode = None         # "iteration" or "batch" once the first call has been made

    def __del__(self):
        if getattr(self, "_mode", None) == "iteration" and getattr(self, "_d_state", None) is not None:
            # the per-symbol server answers before it has written the state back: let that land before the
            # allocator may hand this memory to somebody else
            try:
                self._iter_quiesce(self._iter_ctx)
            except Exception:      # noqa: BLE001 — interpreter teardown
                pass
        if getattr(self, "_ctx", None) is not None:
            from waveforms_amd import _hip

            _hip.free_ctx(self._ctx)
            self._ctx = None

    # ------------------------------------------------------------------ the reference's public state arrays
    # algorithm.py:25-42 gives every instance ``bi_history`` float64[branches_per_column, length], ``metrics``
    # float64[states, length] and ``path`` uint8[states, length]; iteration() (:57-88) rewrites them on every call.  Here the
    # state lives in HBM (one block per detector: wf_viterbi4_state_bytes), so the three names are properties that READ it —
    # after the per-symbol server's write-through has landed (wf_viterbi4_state_read) — into fresh arrays of the reference's
    # shapes and dtypes: a script that looks at ``det.metrics`` between calls sees what the reference's would hold.  They are
    # snapshots (marked read-only: assigning into one would not reach the device).  A detector driven through the batch form
    # detect() keeps a different, smaller carry (no window arrays exist): the properties then raise AttributeError.
    def _state_arrays(self):
        shape_b = (self.fsm.branches_per_column, int(self.length))
        shape_s = (self.fsm.states, int(self.length))
        if self._mode == "batch":
            raise AttributeError("bi_history / metrics / path are the window arrays of the per-symbol API (iteration()); "
                                 "this detector was driven through detect(), whose device carry holds no window")
        if self._mode is None or self._d_state is None:          # no call made yet: the reference's initial zeros
            out = (np.zeros(shape_b, dtype=np.float64), np.zeros(shape_s, dtype=np.float64), np.zeros(shape_s, dtype=np.uint8))
        else:
            from waveforms_amd import _hip

            out = (np.empty(shape_b, dtype=np.float64), np.empty(shape_s, dtype=np.float64), np.empty(shape_s, dtype=np.uint8))
            _hip.check(_hip.lib().wf_viterbi4_state_read(self._iter_ctx, self._d_state_ptr, int(self.length), None, out[0].ctypes.data,
                                                         out[1].ctypes.data, out[2].ctypes.data, _hip.stream()))
        for a in out:
            a.setflags(write=False)
        return out

    @property
    def bi_history(self) -> "NDArray[np.float64]":
        return self._state_arrays()[0]

    @property
    def metrics(self) -> "NDArray[np.float64]":
        return self._state_arrays()[1]

    @property
    def path(self) -> "NDArray[np.uint8]":
        return self._state_arrays()[2]

    # ------------------------------------------------------------------ per-symbol API
    def _ensure_state(self):
        from waveforms_amd import _hip

        if self._d_state is None:
            nbytes = _hip.lib().wf_viterbi4_state_bytes(int(self.length))
            if nbytes < 0:
                raise ValueError(f"unsupported traceback length {self.length}")
            self._d_state = _hip.zeros(nbytes, "uint8")
            self._d_state_ptr = self._d_state.data_ptr()
            # The zero fill must have LANDED before the first request: the iteration server reads the state from its
            # own stream, and the allocator may hand out the address of a detector it served before (the C side only
            # synchronises when the address changes; the server tells a new detector from a cached one by its call
            # counter, which it can only do once the zeros are there).
            _hip.torch().cuda.current_stream().synchronize()
            self._iter_fn, self._iter_ctx = _hip.lib().wf_viterbi4_iteration_host, _hip.ctx()
            self._iter_quiesce = _hip.lib().wf_viterbi4_iteration_quiesce

    def iteration(
        self,
        mf_outputs: NDArray[np.complex128],
    ) -> tuple[NDArray[np.float64], NDArray[np.float64]]:
        """One detector step on the matched-filter outputs of one symbol time
        (rows ordered alpha = -2, 0, +2).  Returns (bits, symbols), ``length`` each,
        oldest first, float64 like the reference."""
        if self._mode != "iteration":
            from waveforms_amd import _hip

            if self._mode == "batch":
                raise ValueError("this detector has been driven through detect(); use one API per burst")
            self._mode = "iteration"
            self._ensure_state()
            self._iter_stream = _hip.stream()
            self._iter_diff = int(self.fsm.trellis is SOQPSKTrellis4x2DiffEncoded)
            # operands at fixed host addresses (an ndarray's .ctypes.data builds a helper object per access: three
            # of them were 1.5 us of a 7.5 us call): the input is copied in, the outputs are copied out
            self._iter_z = np.zeros(3, dtype=np.complex128)
            self._iter_out = np.zeros((2, int(self.length)), dtype=np.float64)
            self._iter_bits, self._iter_syms = self._iter_out[0], self._iter_out[1]      # (views kept: no indexing per call)
            self._iter_args = (self._iter_ctx, self._d_state_ptr, int(self.length), self._iter_diff, self._iter_z.ctypes.data,
                               self._iter_bits.ctypes.data, self._iter_syms.ctypes.data, self._iter_stream)
        elif (self.fsm.trellis is SOQPSKTrellis4x2DiffEncoded) != bool(self._iter_diff) or self.length != self._iter_args[2]:
            # The reference fixes the trellis in __init__ (self.fsm, algorithm.py:27-29) and reads self.fsm and self.length
            # on every call (:62, :67, :71, :94-95): a caller that swaps `det.fsm` between the two 4x2 trellises mid-burst
            # is followed, like there.  (`differential` is this build's own read-only note of the constructor argument.)
            if int(self.length) != self._iter_args[2]:
                raise ValueError("the traceback length of a detector cannot change inside a burst (its state arrays are sized by it)")
            if self.fsm.trellis is not SOQPSKTrellis4x2DiffEncoded and self.fsm.trellis is not SOQPSKTrellis4x2:
                raise ValueError("SOQPSKTrellisDetector serves the two 4-state, 2-column SOQPSK trellises (state_exp_term has four entries)")
            self._iter_diff = int(self.fsm.trellis is SOQPSKTrellis4x2DiffEncoded)
            self.differential = bool(self._iter_diff)
            self._iter_args = self._iter_args[:3] + (self._iter_diff,) + self._iter_args[4:]
        # (the HIP stream is the one current at the FIRST call of the burst: the detector's state is private to this
        # object and only its own methods touch it, so there is nothing on another stream to order against)
        # one C-ABI call per symbol: host operands in, host results out (a persistent kernel serves the calls
        # through a pinned mailbox: no launch, no torch op, no separate copies); the wrapper itself is kept to a
        # few attribute reads — at ~5 us per C call Python's share is what is left to trim
        try:
            self._iter_z[:] = mf_outputs                       # the usual caller: an ndarray of 3 values
        except ValueError:
            self._iter_z[:] = np.reshape(mf_outputs, 3)
        rc = self._iter_fn(*self._iter_args)
        bits, syms = self._iter_bits.copy(), self._iter_syms.copy()
        if rc:
            from waveforms_amd import _hip

            _hip.check(rc)                   # WF_ERR_KEY -> KeyError: the traceback met a state pair with no connecting branch
        self.i += 1
        return bits, syms

    # ------------------------------------------------------------------ batch API
    def detect_device(self, mf_rows, warmup: int = 0):
        """``mf_rows``: float64[n, 3, 2] device tensor -> (bits u8[n], symbols i8[n]) on device."""
        from waveforms_amd import device as dev

        from waveforms_amd import _hip

        L = int(self.length)
        if L < 1 or L > 64:
            raise ValueError(f"unsupported traceback length {L} (1 .. 64, as for iteration())")
        if self._mode == "iteration":
            raise ValueError("this detector has been driven through iteration(); use one API per burst")
        self._mode = "batch"
        if self._d_carry is None:
            self._d_carry = _hip.zeros(32 if L == 2 else _hip.lib().wf_viterbi4_window_state_bytes() // 8, "float64")
        # The kernel is chunk-parallel; every launch proves on the device that each chunk started from bitwise the
        # state its predecessor ended with and runs the chunks that did not again from the true state, cascading
        # where needed (csrc/wf_viterbi.hip: viterbi_fixup_kernel) — the output is the sequential detector's
        # (algorithm.py:44-101) whatever `warmup` is.  The proof counter lives in the wf_ctx: the detector owns a
        # private context so that its read-with-reset cannot disturb a SOQPSKLink or a BER sweep on the default one.
        if self._ctx is None:
            self._ctx = _hip.new_ctx()
        n = int(mf_rows.shape[0])
        # the trellis as it stands NOW (iteration() follows a swap of det.fsm the same way; algorithm.py:62 reads self.fsm per call)
        if self.fsm.trellis is not SOQPSKTrellis4x2DiffEncoded and self.fsm.trellis is not SOQPSKTrellis4x2:
            raise ValueError("SOQPSKTrellisDetector serves the two 4-state, 2-column SOQPSK trellises (state_exp_term has four entries)")
        self.differential = self.fsm.trellis is SOQPSKTrellis4x2DiffEncoded
        # (repairs switched off — tests of the proof — is the one way this call can raise behind a launch: the carry is put back then)
        keep = self._d_carry.clone() if _hip.get_option(self._ctx, _hip.WF_OPT_DET_REPAIR) else None
        out = (dev.viterbi_detect(mf_rows, self.differential, warmup, self._d_carry, ctx=self._ctx) if L == 2 else
               dev.viterbi_detect_window(mf_rows, L, self.differential, warmup, self._d_carry, ctx=self._ctx))
        unproven = dev.viterbi_unmerged(reset=True, ctx=self._ctx)
        if unproven:        # only with the context's WF_OPT_DET_REPAIR option switched off (tests of the proof itself)
            if keep is not None:
                self._d_carry.copy_(keep)
            raise RuntimeError(f"{unproven} detector chunk(s) were left unproven (the repairs are switched off on this context)")
        self.i += n                         # like iteration(): one call per row, state carried
        return out

    def detect(self, mf_rows: NDArray[np.complex128], warmup: int = 0):
        """Host in / host out batch form: complex128[n, 3] -> (bits u8[n], symbols i8[n])."""
        from waveforms_amd import _hip

        rows = np.ascontiguousarray(mf_rows, dtype=np.complex128).reshape(-1, 3)
        bits, syms = self.detect_device(_hip.to_device(rows), warmup)
        return _hip.to_host(bits), _hip.to_host(syms)
