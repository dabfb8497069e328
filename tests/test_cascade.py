"""Equality with the sequential detector does not depend on the chunk warm-up (round-4 verdict, item 1).

The reference's detector is one sequential loop and is right at any Eb/N0 (waveforms/viterbi/algorithm.py:44-101).  The
chunk-parallel kernels prove every chunk boundary on the device and run the chunks that missed their warm-up again from
the true state; a chunk whose END changed hands on to the next chunk, round after round, until no boundary differs.
These tests take the warm-up down to 2 rows — nearly every chunk misses — at 0 and 2 dB (and below), in every detector
family and form, with chunks short enough that repairs must cross chunk boundaries many times over, and ask for:
decisions == the sequential oracle, nothing unproven (WF_OPT_DET_FINAL_VERIFY re-checks every boundary at the end),
nothing raised.
"""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SPS = 8


def _soqpsk_rows(oracle, n, ebn0, length, seed):
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    noise = oracle.philox_awgn(oracle.sigma_for_ebn0(ebn0, SPS), 7, seed, 0, (n + 1) * SPS)
    return oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(SPS), 0.25, SPS, None, noise=noise, length=length)


def _cpm_rows(oracle, spec, pulse, nsym, ebn0, seed):
    rng = np.random.default_rng(seed)
    sym = (2 * rng.integers(0, spec.M, nsym) - (spec.M - 1)).astype(np.int8)
    sigma = oracle.cpm_sigma_for_ebn0(ebn0, SPS, spec.lgM)
    return oracle.cpm_detection_run(sym, pulse, SPS, spec, sigma=sigma, rng=np.random.Generator(np.random.PCG64(seed)))


def test_ctx_options_cabi_without_a_gpu():
    """wf_ctx_set_option / wf_ctx_get_option validate before they touch the context."""
    from waveforms_amd import _hip

    lib = _hip.lib()
    assert lib.wf_ctx_set_option(None, _hip.WF_OPT_CPM_FORM, 1) == _hip.WF_ERR_VALUE
    v = ctypes.c_int64(5)
    assert lib.wf_ctx_get_option(None, _hip.WF_OPT_CPM_FORM, ctypes.byref(v)) == _hip.WF_ERR_VALUE
    fake = ctypes.create_string_buffer(1 << 16)                   # (only its option fields are touched on these paths)
    assert lib.wf_ctx_set_option(fake, 99, 1) == _hip.WF_ERR_VALUE
    assert lib.wf_ctx_set_option(fake, -1, 1) == _hip.WF_ERR_VALUE
    assert lib.wf_ctx_set_option(fake, _hip.WF_OPT_CPM_FORM, 3) == _hip.WF_ERR_VALUE
    assert lib.wf_ctx_set_option(fake, _hip.WF_OPT_DET_REPAIR, 2) == _hip.WF_ERR_VALUE
    assert lib.wf_ctx_set_option(fake, _hip.WF_OPT_CPM_CHUNK_CALLS, 1 << 20) == _hip.WF_ERR_VALUE
    assert lib.wf_ctx_set_option(fake, _hip.WF_OPT_CPM_CHUNK_CALLS, 320) == 0
    assert lib.wf_ctx_get_option(fake, _hip.WF_OPT_CPM_CHUNK_CALLS, ctypes.byref(v)) == 0 and v.value == 320
    assert lib.wf_ctx_get_option(fake, _hip.WF_OPT_CPM_FORM, ctypes.byref(v)) == 0 and v.value == 0


def test_library_reads_no_environment_variable():
    """no hidden globals.  Every knob is a wf_ctx option; no translation unit calls getenv."""
    offenders = [p.name for p in (ROOT / "waveforms_amd" / "csrc").glob("*.hip") if "getenv" in p.read_text()]
    offenders += [p.name for p in (ROOT / "waveforms_amd" / "csrc").glob("*.h") if "getenv" in p.read_text()]
    assert not offenders, offenders


@pytest.mark.gpu
def test_ctx_options_round_trip_on_a_context():
    from waveforms_amd import _hip

    ctx = _hip.new_ctx()
    try:
        for key, val in ((_hip.WF_OPT_CPM_FORM, 2), (_hip.WF_OPT_CPM_CHUNK_CALLS, 384), (_hip.WF_OPT_DET_FINAL_VERIFY, 1),
                         (_hip.WF_OPT_MCB_TAIL_PERMILLE, -1)):
            assert _hip.get_option(ctx, key) == 0
            _hip.set_option(ctx, key, val)
            assert _hip.get_option(ctx, key) == val
        with pytest.raises(ValueError):
            _hip.set_option(ctx, _hip.WF_OPT_CPM_FORM, 7)
    finally:
        _hip.free_ctx(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("length", [2, 8])
@pytest.mark.parametrize("ebn0", [0.0, 2.0, -8.0])
def test_soqpsk_detector_warmup_2_equals_sequential_at_low_snr(oracle, ctx_options, length, ebn0):
    """SOQPSKTrellisDetector.detect, length 2 (the batch kernel: 32-call chunks here) and 8 (the window kernel: 8-call
    chunks with this warm-up — a repair has to travel through several chunks before it meets the old trajectory)."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    n = 400_000
    res = _soqpsk_rows(oracle, n, ebn0, length, 40 + length)
    rows = _hip.to_device(np.ascontiguousarray(res["mf_rows"]))
    for k in (dev.viterbi_unmerged, dev.viterbi_repaired, dev.viterbi_cascaded):
        k(reset=True)
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        b, s = (dev.viterbi_detect(rows, warmup=2) if length == 2 else dev.viterbi_detect_window(rows, length, warmup=2))
        assert dev.viterbi_unmerged(reset=True) == 0
        repaired, handed_on = dev.viterbi_repaired(reset=True), dev.viterbi_cascaded(reset=True)
        assert np.array_equal(_hip.to_host(b), res["det_bits"]) and np.array_equal(_hip.to_host(s), res["det_syms"])
        assert repaired > 1000, repaired
        if length == 8:
            assert handed_on > 0, "chunks this short were expected to hand repairs on to their successors"
        # the object API, state carried across ragged pieces, the same warm-up: never raises, same decisions
        det = SOQPSKTrellisDetector(length)
        cuts = [0, 1, 77, 100_001, 250_000, n]
        parts = [det.detect(res["mf_rows"][a:b_], warmup=2) for a, b_ in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), res["det_bits"])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), res["det_syms"])


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["rows", "lanes"])
@pytest.mark.parametrize("waveform", ["pcmfm", "multih"])
@pytest.mark.parametrize("ebn0,chunk", [(0.0, 0), (2.0, 0), (0.0, 64), (2.0, 128)])
def test_cpm_detector_warmup_2_equals_sequential_at_low_snr(oracle, ctx_options, form, waveform, ebn0, chunk):
    """PCM/FM (10 states) and ARTM (16 states) in the row form and in the lane form, at the library's chunk length and
    with chunks of 64 / 128 calls — shorter than the merge depth at these Eb/N0, so repairs cross chunk boundaries."""
    from waveforms_amd import _hip, device as dev
    from waveforms_amd.viterbi import cpm

    spec_o = oracle.ARTM_16 if waveform == "multih" else oracle.PCMFM_SPEC
    spec_p = cpm.ARTM_16 if waveform == "multih" else cpm.PCMFM_10
    pulse = oracle.freq_pulse_multih_irig(SPS) if waveform == "multih" else oracle.freq_pulse_pcmfm(SPS)
    res = _cpm_rows(oracle, spec_o, pulse, 300_000, ebn0, 3 + chunk)
    with ctx_options(WF_OPT_CPM_FORM={"rows": 1, "lanes": 2}[form], WF_OPT_CPM_CHUNK_CALLS=chunk, WF_OPT_DET_FINAL_VERIFY=1):
        det = cpm.CPMTrellisDetector(spec_p)
        got = det.detect(res["rows"], warmup=2)                  # (raises if anything was left unproven)
        repaired, handed_on = dev.viterbi_repaired(reset=True, ctx=det._ctx), dev.viterbi_cascaded(reset=True, ctx=det._ctx)
        assert np.array_equal(got, res["decisions"])
        assert repaired > 100, repaired
        if chunk == 64:
            assert handed_on > 0, "chunks this short were expected to hand repairs on to their successors"
        # state carried across pieces (a 1-call piece first: the lane form's fresh-burst corner), same warm-up
        det = cpm.CPMTrellisDetector(spec_p)
        n = res["rows"].shape[0]
        cuts = [0, 1, 2, 40_001, 170_000, n]
        got = np.concatenate([det.detect(res["rows"][a:b], warmup=2) for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(got, res["decisions"])


@pytest.mark.gpu
@pytest.mark.parametrize("ebn0,chunk", [(0.0, 0), (2.0, 0), (0.0, 64)])
def test_cpm_wide_detector_warmup_2_equals_sequential_at_low_snr(oracle, ctx_options, ebn0, chunk):
    """The 64-state ARTM design (one wave per chunk, a pair of waves per repair)."""
    from waveforms_amd import device as dev
    from waveforms_amd.viterbi import cpm

    res = _cpm_rows(oracle, oracle.ARTM_64, oracle.freq_pulse_multih_irig(SPS), 120_000, ebn0, 17 + chunk)
    with ctx_options(WF_OPT_CPM_CHUNK_CALLS=chunk, WF_OPT_DET_FINAL_VERIFY=1):
        det = cpm.CPMTrellisDetector(cpm.ARTM_64)
        got = det.detect(res["rows"], warmup=2)
        repaired, handed_on = dev.viterbi_repaired(reset=True, ctx=det._ctx), dev.viterbi_cascaded(reset=True, ctx=det._ctx)
        assert np.array_equal(got, res["decisions"])
        assert repaired > 100
        if chunk:
            assert handed_on > 0
        det = cpm.CPMTrellisDetector(cpm.ARTM_64)
        n = res["rows"].shape[0]
        cuts = [0, 1, 30_001, n]
        got = np.concatenate([det.detect(res["rows"][a:b], warmup=2) for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(got, res["decisions"])


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,ebn0,warmup", [("pcmfm", 2.0, 64), ("pcmfm", 0.0, 8), ("multih", 0.0, 8), ("soqpsk", 0.0, 2)])
def test_links_at_low_snr_with_a_short_warmup_equal_the_default(ctx_options, waveform, ebn0, warmup):
    """The links: error counts with a warm-up far below the merge depth == counts with the library default, pipelined
    (fuse bit 5) and sequential; result() never raises.  (PCM/FM at 2 dB with 64 rows is the configuration that
    ended in a proof failure in round 4: profiles/r04_lowsnr_scan.log.)"""
    from waveforms_amd import device as dev
    from waveforms_amd.link import CPMLink, SOQPSKLink

    nsym = 2_000_000
    out = {}
    # (the lane form — what a 1e7-symbol block of these waveforms runs — forced at this size too)
    with ctx_options(WF_OPT_CPM_FORM=2, WF_OPT_DET_FINAL_VERIFY=1):
        for wu in (0, warmup):
            for piped in (False, True):
                if waveform == "soqpsk":
                    link = SOQPSKLink(nsym, SPS, warmup=wu, fuse=47 if piped else 15, private_ctx=True)
                else:
                    link = CPMLink(nsym, SPS, waveform=waveform, warmup=wu, fuse=42 if piped else 10, private_ctx=True)
                for blk in range(3):
                    link.run_block(ebn0, seed=1, stream_id=blk, skip_bits=blk * nsym)
                out[(wu, piped)] = link.result()
                if wu:
                    assert dev.viterbi_repaired(reset=True, ctx=link._ctx) > 0
                del link
    assert len(set(out.values())) == 1, out
    assert out[(0, False)][1] > 0


@pytest.mark.gpu
def test_bench_pcmfm_at_2_db_prints_a_line():
    """`bench.py --waveform pcmfm --ebn0 2` — the command that died with a traceback in round 4 — and the same with the
    64-row warm-up forced (what the table then chose at 2 dB): a JSON line, nothing unproven."""
    for extra in ([], ["--vit-warmup", "64"]):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--waveform", "pcmfm", "--ebn0", "2", "--steps", "3", "--warmup", "1",
               "--steady-steps", "40", "--no-cpu-baseline", "--overlap-streams", "0", *extra]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["ber"]["detector_chunks_unproven"] == 0 and line["steady_state"]["detector_chunks_unproven"] == 0
        assert line["ber"]["bit_errors"] > 0 and line["steady_state"]["bit_errors"] > 0


def _survivor_permutation_rows(n, col0_first, amplitude=8.0, seed=5):
    """Matched-filter rows under which no two survivors of the 4-state trellis ever share an ancestor: in each of the two trellis
    sections ONE incoming branch of every end state is preferred by 2 x amplitude over the other, and the preferred branches
    start in four different states (branch lists of waveforms/cpm/trellis/model.py:205-258; increments Re(state_exp_term *
    mf), waveforms/viterbi/algorithm.py:57-63).  Metric differences between the states then travel along a permutation for
    ever: a chunk warmed up from a fresh detector never meets the true trajectory bitwise — every repair changes its chunk's
    end and hands on to the next chunk.  The first rows are random, so that the true metrics are not all equal."""
    rng = np.random.default_rng(seed)
    a = np.array([-1 - 1j, 1 - 1j, -1 + 1j]) * amplitude        # section 0: survivors 0 <- 2, 1 <- 1, 2 <- 0, 3 <- 3
    b = np.array([-1 - 1j, 1 + 1j, -1 - 1j]) * amplitude        # section 1: survivors 0 <- 0, 1 <- 1, 2 <- 3, 3 <- 2
    rows = np.empty((n, 3), dtype=np.complex128)
    rows[0::2] = a if col0_first else b
    rows[1::2] = b if col0_first else a
    rows[:9] = rng.normal(size=(9, 3)) + 1j * rng.normal(size=(9, 3))
    rows[9:] += 1e-3 * (rng.normal(size=(n - 9, 3)) + 1j * rng.normal(size=(n - 9, 3)))     # (no exact ties)
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("diff", [True, False])
def test_soqpsk_batch_detector_cascades_through_every_chunk(oracle, ctx_options, diff):
    """The cascade branch of the length-2 batch detector (viterbi_fixup_kernel: vit_rerun_chunk returns `changed`, the
    chunk behind is listed for the next round, the carry of a stream is rewritten when the last chunk's end changes):
    rows under which metrics never re-merge (above), so that EVERY repair hands on — hundreds of rounds, each reading
    its predecessor's end while other chunks are being repaired in the same round.  Decisions == the sequential oracle
    in one burst, in two bursts with the carry between them (the repaired chunk owns the first burst's last call), and
    through the link-style packed rows' twin (the object API); every boundary proven once more behind the repairs."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    n = 40_001
    seen_cascade = 0
    for col0_first in (True, False):
        rows = _survivor_permutation_rows(n, col0_first)
        want_b, want_s = oracle.viterbi_detect(rows, 2, diff)
        with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
            for k in (dev.viterbi_unmerged, dev.viterbi_repaired, dev.viterbi_cascaded):
                k(reset=True)
            b, s = dev.viterbi_detect(_hip.to_device(rows), differential=diff, warmup=16)
            assert dev.viterbi_unmerged(reset=True) == 0
            repaired, handed_on = dev.viterbi_repaired(reset=True), dev.viterbi_cascaded(reset=True)
            assert np.array_equal(_hip.to_host(b), want_b) and np.array_equal(_hip.to_host(s), want_s), (col0_first, repaired, handed_on)
            seen_cascade = max(seen_cascade, handed_on)
            # the carry: two bursts, the first ending inside the permutation rows (its last chunk is repaired, and its end changes)
            det = SOQPSKTrellisDetector(2, differantial_encoding=diff)
            cut = 20_017
            p0 = det.detect(rows[:cut], warmup=16)
            p1 = det.detect(rows[cut:], warmup=16)
            assert dev.viterbi_cascaded(reset=True, ctx=det._ctx) >= 0
            assert np.array_equal(np.concatenate([p0[0], p1[0]]), want_b) and np.array_equal(np.concatenate([p0[1], p1[1]]), want_s)
    assert seen_cascade > 500, f"the crafted rows were expected to make every repair hand on (saw {seen_cascade})"
