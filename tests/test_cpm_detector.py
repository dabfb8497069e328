"""Generic CPM trellis detector (BASELINE configs[2]: ARTM multi-h, 16 states).

There is no reference implementation of this detector, so parity here is BUILD-DEFINED: the
sequential C detector in oracle/cpm_oracle.c is the definition, pinned by (a) a second, independent
plain-Python statement of the same recursion, (b) theory — zero errors without noise, the
published minimum distance of ARTM CPM (d^2 = 1.29) recomputed from the reference's own pulse, and
the bit-error rate against the minimum-distance bound — and the HIP kernels must then reproduce
it bit for bit through the C ABI.
"""
import ctypes
import os
import math

import numpy as np
import pytest

SPS = 8


def torch_f64():
    import torch

    return torch.float64


def _spec(oracle, M, p, K, Lp, NC, D=32):
    return oracle.CPMDetectorSpec(M=M, p=p, K=tuple(K), Lp=Lp, NC=NC, D=D)


def _random_symbols(rng, n, M):
    return (2 * rng.integers(0, M, n) - (M - 1)).astype(np.int8)


DESIGNS = [  # (M, p, K, Lp, NC): full and reduced trellises, both alphabets, every filter length
    (4, 16, (4, 5), 2, 4), (4, 16, (4, 5), 2, 16), (4, 16, (4, 5), 3, 16), (4, 16, (4, 5), 1, 16), (4, 16, (4, 5), 3, 1),
    (4, 16, (4, 5), 2, 2), (2, 10, (7,), 2, 5), (2, 10, (7,), 3, 2), (2, 10, (7,), 1, 10), (2, 2, (1,), 1, 2),
    # 17 .. 64 states (the wide form of the GPU detector: lane = state, one wave per detector)
    (4, 16, (4, 5), 2, 8), (4, 16, (4, 5), 3, 4), (2, 10, (7,), 3, 10), (2, 16, (7,), 2, 16), (2, 20, (7,), 1, 20),
    # 65 .. 256 states (the quad form: thread = state, one workgroup per detector; pulses of 2 or 3 symbols)
    (4, 16, (4, 5), 3, 8), (2, 64, (9,), 2, 64), (2, 32, (7,), 3, 32), (2, 64, (9, 11), 3, 64),
]
GPU_DESIGNS = [d for d in DESIGNS if d[4] * d[0] ** (d[3] - 1) <= 64 or d[3] >= 2]


@pytest.fixture(params=["auto", "lanes", "rows"])
def detector_form(request):
    """wf_cpm_viterbi_detect has two forms — one 16-lane row per chunk, one lane per chunk — and picks by burst length
    (the lane form pays from ~9e6 / ~6.5e6 calls).  The tests below run every size in BOTH (the context option
    WF_OPT_CPM_FORM forces a form where a lane specialisation exists) and as shipped.  WF_OPT_DET_FINAL_VERIFY is on for
    all of them: after the repairs every chunk boundary is compared once more and any difference counted as unproven."""
    from waveforms_amd import _hip

    _hip.set_default_option(_hip.WF_OPT_CPM_FORM, {"auto": 0, "rows": 1, "lanes": 2}[request.param])
    _hip.set_default_option(_hip.WF_OPT_DET_FINAL_VERIFY, 1)
    yield request.param
    _hip.set_default_option(_hip.WF_OPT_CPM_FORM, 0)
    _hip.set_default_option(_hip.WF_OPT_DET_FINAL_VERIFY, 0)


# ------------------------------------------------------------------ oracle side (CPU)
@pytest.mark.parametrize("design", DESIGNS)
def test_c_detector_equals_independent_python_statement(oracle, design):
    spec = _spec(oracle, *design, D=7)
    rng = np.random.default_rng(hash(design) & 0xFFFF)
    rows = rng.standard_normal((260, spec.nfilt)) + 1j * rng.standard_normal((260, spec.nfilt))
    rows[::17] = 0.0                                      # exact ties: first listed branch / first arg-min must win
    rows[5::23] = np.round(rows[5::23])
    want = oracle.cpm_viterbi_py(spec, rows)
    det = oracle.cpm_viterbi(spec)
    got = np.concatenate([det.run(rows[:101]), det.run(rows[101:])])      # state carried across calls
    assert got.size == 260 - 6 and np.array_equal(got, want)


@pytest.mark.parametrize("name,errors_allowed_at_start", [("ARTM_16", 0), ("ARTM_64", 1), ("ARTM_256", 1)])
def test_noiseless_multih_is_error_free(oracle, name, errors_allowed_at_start):
    """Every design decodes the reference modulator's own output without error (the full trellises
    may miss symbol 0: their hypothesised pre-start symbols are not what was (not) sent)."""
    bits = oracle.glfsr_bits(oracle.lfsr_mask(23), (1 << 23) - 1, 2 * 20000)[0]
    sym = oracle.multih_mapper(bits)[0]
    res = oracle.cpm_detection_run(sym, oracle.freq_pulse_multih_irig(SPS), SPS, getattr(oracle, name))
    assert res["compared"] == 20000 - 31 and res["sym_errors"] <= errors_allowed_at_start
    assert np.array_equal(res["decisions"][8:], res["truth"][8:])
    assert np.array_equal(oracle.u_to_bits(res["truth"], 4), bits[:2 * res["compared"]])


def test_noiseless_pcmfm_is_error_free(oracle):
    bits = oracle.pn_sequence(15)[:12000]
    res = oracle.cpm_detection_run(oracle.pcmfm_mapper(bits), oracle.freq_pulse_pcmfm(SPS), SPS, oracle.PCMFM_SPEC)
    assert res["sym_errors"] == 0 and np.array_equal(res["truth"], bits[:res["compared"]])


def test_minimum_distance_matches_published_values(oracle):
    """Theory anchor: d^2_min of the waveform the detector is built for, recomputed from the
    reference's own pulse and modulation indices — MSK 2.0 (textbook) and ARTM CPM 1.29 (the
    figure quoted for IRIG-106 ARTM CPM in the reduced-complexity detection literature the
    reference cites, README.md:65-76)."""
    assert abs(oracle.cpm_min_distance([0.0] + [0.5] * 8, 8, 2, (1,), 2, 4) - 2.0) < 1e-12
    d2 = oracle.cpm_min_distance(oracle.freq_pulse_multih_irig(SPS), SPS, 4, (4, 5), 16, 6)
    assert 1.285 < d2 < 1.30, d2


def test_oracle_ber_sits_on_the_minimum_distance_bound(oracle):
    """Full-trellis detector at 9 dB: BER within a small factor of Q(sqrt(d^2 Eb/N0)) (the
    minimum-distance term of the union bound; multiplicities and neighbours add a factor of a
    few), and the 16-state design within 0.5 dB-equivalent of it."""
    rng = np.random.default_rng(9)
    sym = _random_symbols(rng, 60000, 4)
    sigma = oracle.cpm_sigma_for_ebn0(9.0, SPS, 2)
    noise = oracle.numpy_awgn(sigma, (sym.size + 1) * SPS, np.random.Generator(np.random.PCG64(5)))
    pulse = oracle.freq_pulse_multih_irig(SPS)
    ber = {}
    for name in ("ARTM_256", "ARTM_16"):
        res = oracle.cpm_detection_run(sym, pulse, SPS, getattr(oracle, name), noise=noise)
        ber[name] = res["bit_errors"] / (2 * res["compared"])
    q = 0.5 * math.erfc(math.sqrt(1.2957 * 10 ** 0.9) / math.sqrt(2))
    assert q < ber["ARTM_256"] < 6 * q, (ber, q)
    assert ber["ARTM_256"] <= ber["ARTM_16"] < 2.5 * ber["ARTM_256"]


def test_oracle_detector_regression_pin(oracle, golden):
    """tests/golden/cpm_detect.npz freezes the build-defined detector's decisions (made by the oracle
    itself at the commit that introduced it — not by the reference, which has no such detector)."""
    g = golden("cpm_detect")
    bits = g["bits"]
    for name, spec, pulse, sym in (("artm16", oracle.ARTM_16, oracle.freq_pulse_multih_irig(SPS), oracle.multih_mapper(bits)[0]),
                                   ("pcmfm10", oracle.PCMFM_SPEC, oracle.freq_pulse_pcmfm(SPS), oracle.pcmfm_mapper(bits[:3000]))):
        noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(5.0, SPS, spec.lgM), 11, 3, 0, (sym.size + 1) * SPS)
        res = oracle.cpm_detection_run(sym, pulse, SPS, spec, noise=noise)
        np.testing.assert_allclose(res["rows"][:64], g[f"{name}_rows_head"], rtol=0, atol=1e-12)
        assert np.array_equal(res["decisions"], g[f"{name}_decisions"])
        assert [res["sym_errors"], res["bit_errors"], res["compared"]] == g[f"{name}_errors"].tolist()
        assert res["bit_errors"] > 0
    assert abs(g["d2_artm"][0] - 1.2957297551846658) < 1e-12


def test_templates_of_a_symmetric_alphabet_pair_off_as_conjugates(oracle):
    """What wf_cpm_link_config.fuse bit 6 vouches for: filter nf - 1 - f (the negated symbol pattern) is, bit for bit, the
    conjugate of filter f — for every shipped design, in every modulation-index column."""
    from waveforms_amd.viterbi import cpm

    for spec, pulse in ((cpm.ARTM_16, oracle.freq_pulse_multih_irig(SPS)), (cpm.ARTM_64, oracle.freq_pulse_multih_irig(SPS)),
                        (cpm.PCMFM_10, oracle.freq_pulse_pcmfm(SPS))):
        t = cpm.matched_filter_templates(pulse, SPS, spec)
        assert t.shape[1] in (4, 16)
        assert np.array_equal(t[:, ::-1, :], np.conj(t))


@pytest.mark.gpu
def test_gpu_cpm_paired_front_end_with_an_odd_centre_tap_takes_the_plain_form(monkeypatch):
    """The paired 16-filter front end reads a sample pair's symbols once (both samples under the same symbols: even sps AND an
    even centre tap, ARTM's 25 taps).  A 23-tap pulse of the same shape has an odd one: the library must run the plain form
    for it — bit for bit the rows of a link that never asked for pairs — and the staged kernels must agree with it."""
    import waveforms_amd.cpm.multih as mh
    from waveforms_amd.link import CPMLink

    full = mh.freq_pulse_multih_irig(SPS)
    short = full[1:-1] / full[1:-1].sum() * full.sum()
    assert short.size == 23
    monkeypatch.setattr(mh, "freq_pulse_multih_irig", lambda sps: short)
    nsym = 60_001
    plain = CPMLink(nsym, SPS, waveform="multih", fuse=10, paired_templates=False)
    pairs = CPMLink(nsym, SPS, waveform="multih", fuse=10)
    staged = CPMLink(nsym, SPS, waveform="multih", fuse=2)
    for link in (plain, pairs, staged):
        link.run_block(7.0, seed=3, stream_id=9, skip_bits=5)
    la, lb, lc = plain.layout(), pairs.layout(), staged.layout()
    assert la["one_kernel_front_end"] == 1 and lb["one_kernel_front_end"] == 1 and lc["one_kernel_front_end"] == 0
    calls = la["calls"]
    rows = [l.workspace[y["off_rows"]:y["off_rows"] + calls * 16 * 16].view(torch_f64()).cpu().numpy() for l, y in ((plain, la), (pairs, lb), (staged, lc))]
    assert np.abs(rows[0]).max() > 1.0 and np.array_equal(rows[0], rows[1])
    np.testing.assert_allclose(rows[2], rows[0], rtol=0, atol=1e-11)
    assert plain.result() == pairs.result() == staged.result()


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,nsym", [("multih", 150_001), ("pcmfm", 150_001), ("multih", 900), ("pcmfm", 70)])
def test_gpu_cpm_front_end_conjugate_pairs_equal_the_plain_form(waveform, nsym):
    """The one-kernel front end with the templates taken as conjugate pairs (four real sums per pair: what a link runs) against
    its plain form (every filter on its own, the k-ascending chain cpm_oracle.c states): rows equal to rounding — the sums
    are added in another order —, decisions and counts identical; tile edges, both ends of the burst, a burst shorter
    than a row."""
    from waveforms_amd.link import CPMLink

    plain = CPMLink(nsym, SPS, waveform=waveform, fuse=10, paired_templates=False)
    pairs = CPMLink(nsym, SPS, waveform=waveform, fuse=10)
    assert not (plain.cfg.fuse & 64) and (pairs.cfg.fuse & 64) and pairs.paired_templates and not plain.paired_templates
    nf = 16 if waveform == "multih" else 4
    for ebn0, sid in ((3.0, 5), (9.0, 6)):
        for link in (plain, pairs):
            link.reset_counts()
            link.run_block(ebn0, seed=3, stream_id=sid, skip_bits=17)
        la, lb = plain.layout(), pairs.layout()
        assert la == lb and la["one_kernel_front_end"] == 1
        calls = la["calls"]
        a = plain.workspace[la["off_rows"]:la["off_rows"] + calls * nf * 16].view(torch_f64()).cpu().numpy()
        b = pairs.workspace[lb["off_rows"]:lb["off_rows"] + calls * nf * 16].view(torch_f64()).cpu().numpy()
        assert np.abs(a).max() > 1.0
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-12)
        assert not np.array_equal(a, b) or nsym < 100            # (another order of additions: the last bits do differ)
        da = plain.workspace[la["off_decisions"]:la["off_decisions"] + calls].cpu().numpy()
        db = pairs.workspace[lb["off_decisions"]:lb["off_decisions"] + calls].cpu().numpy()
        assert np.array_equal(da, db)
        assert plain.result() == pairs.result()


def test_host_mirror_builds_the_same_constants(oracle):
    """waveforms_amd.viterbi.cpm (product side) and the oracle derive identical templates,
    rotation table and window geometry — the product never imports the oracle."""
    from waveforms_amd.viterbi import cpm

    for spec_o, spec_p, pulse in ((oracle.ARTM_16, cpm.ARTM_16, oracle.freq_pulse_multih_irig(SPS)),
                                  (oracle.PCMFM_SPEC, cpm.PCMFM_10, oracle.freq_pulse_pcmfm(SPS))):
        assert (spec_o.M, spec_o.p, spec_o.K, spec_o.Lp, spec_o.NC, spec_o.D) == \
            (spec_p.M, spec_p.p, spec_p.K, spec_p.Lp, spec_p.NC, spec_p.D)
        assert np.array_equal(cpm.matched_filter_templates(pulse, SPS, spec_p), oracle.cpm_templates(pulse, SPS, spec_o))
        assert np.array_equal(cpm.rotation_table(spec_p), oracle.cpm_rot_table(spec_o))
        for nsym in (1, 5, 1000):
            a, b = cpm.filter_geometry(pulse.size, SPS, spec_p, nsym), oracle.cpm_geometry(pulse, SPS, spec_o, nsym)
            assert (a["start0"], a["ntm"], a["ncalls"], a["npts"]) == (b["start0"], b["ntm"], b["ncalls"], b["npts"])
    assert cpm.sigma_for_ebn0(7.0, 8, 2) == oracle.cpm_sigma_for_ebn0(7.0, 8, 2)
    assert cpm.ARTM_16.nstates == 16 and cpm.PCMFM_10.nstates == 10


def test_cabi_rejects_unsupported_detectors_without_a_gpu():
    from waveforms_amd import _hip

    lib = _hip.lib()
    # the context pointer is checked first, the configuration before any device work
    for bad in (dict(M=3), dict(Lp=4), dict(NC=3), dict(p=32, NC=32, Lp=3), dict(D=33), dict(nh=3), dict(p=65)):
        c = _hip.CPMDetectorConfig()
        c.M, c.p, c.nh, c.Lp, c.NC, c.D = 4, 16, 2, 2, 4, 32
        c.K[0], c.K[1] = 4, 5
        for k, v in bad.items():
            setattr(c, k, v)
        fake_ctx = ctypes.create_string_buffer(4096)      # never dereferenced for a device call on these paths
        rc = lib.wf_cpm_viterbi_detect(fake_ctx, ctypes.byref(c), None, None, 0, 0, None, None, None)
        assert rc == _hip.WF_ERR_VALUE, bad
    cfg = _hip.CPMLinkConfig()
    assert lib.wf_cpm_link_workspace_bytes(ctypes.byref(cfg)) == -1
    cfg.nsym, cfg.sps, cfg.ntaps, cfg.mapper_kind = 1000, 8, 25, 1
    cfg.det.M, cfg.det.p, cfg.det.nh, cfg.det.Lp, cfg.det.NC, cfg.det.D = 4, 16, 2, 2, 4, 32
    assert lib.wf_cpm_link_workspace_bytes(ctypes.byref(cfg)) > 1000 * (16 * 16 + 8 * 16)
    cfg.mapper_kind = 2                                   # binary mapper with a quaternary detector
    assert lib.wf_cpm_link_workspace_bytes(ctypes.byref(cfg)) == -1


# ------------------------------------------------------------------ HIP kernels (GPU)
def _noisy_rows(oracle, spec, pulse, nsym, ebn0, seed, sps=SPS):
    rng = np.random.default_rng(seed)
    sym = _random_symbols(rng, nsym, spec.M)
    sigma = oracle.cpm_sigma_for_ebn0(ebn0, sps, spec.lgM)
    res = oracle.cpm_detection_run(sym, pulse, sps, spec, sigma=sigma, rng=np.random.Generator(np.random.PCG64(seed)))
    return sym, res


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,sps", [("multih", 8), ("multih", 4), ("pcmfm", 8), ("pcmfm", 10)])
def test_gpu_matched_filter_rows_equal_oracle(oracle, waveform, sps):
    from waveforms_amd import _hip, device as dev

    spec = oracle.ARTM_16 if waveform == "multih" else oracle.PCMFM_SPEC
    pulse = oracle.freq_pulse_multih_irig(sps) if waveform == "multih" else oracle.freq_pulse_pcmfm(sps)
    sym, res = _noisy_rows(oracle, spec, pulse, 3001, 6.0, 3, sps)
    geo = res["geometry"]
    T = oracle.cpm_templates(pulse, sps, spec)
    rows = dev.cpm_mf_rows(_hip.to_device(res["received"]), _hip.to_device(T), geo["start0"], sps, geo["ncalls"])
    got = _hip.to_host(rows, complex_pairs=True)
    assert got.shape == res["rows"].shape
    np.testing.assert_allclose(got, res["rows"], rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("ebn0", [0.0, 4.0, 10.0])
def test_gpu_artm16_decisions_equal_sequential_oracle(oracle, ebn0, detector_form):
    """BASELINE configs[2]'s detector: 4.2e5 symbols, chunk-parallel on the GPU vs the sequential
    C detector on the same matched-filter rows — every decision identical."""
    from waveforms_amd.viterbi.cpm import ARTM_16, CPMTrellisDetector

    sym, res = _noisy_rows(oracle, oracle.ARTM_16, oracle.freq_pulse_multih_irig(SPS), 420_000, ebn0, int(ebn0) + 11)
    got = CPMTrellisDetector(ARTM_16).detect(res["rows"])
    assert got.size == res["decisions"].size == 420_000 - 31
    assert np.array_equal(got, res["decisions"])
    if ebn0 >= 10.0:
        assert res["bit_errors"] < 2e-3 * res["compared"]
    else:
        assert res["bit_errors"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("ebn0", [0.0, 4.0, 10.0])
def test_gpu_artm64_decisions_equal_sequential_oracle(oracle, ebn0):
    """The 64-state ARTM design (Lp 2, NC = p = 16: N_S = p M^(Lp-1) of notes/cpm/cpm.md:128-140 for the two-symbol
    pulse) on the GPU — one wave per detector — against the sequential C detector on the same 16 matched-filter rows per
    symbol: every decision identical; and it beats the 16-state design on the same received samples."""
    from waveforms_amd.viterbi.cpm import ARTM_64, CPMTrellisDetector

    sym, res = _noisy_rows(oracle, oracle.ARTM_64, oracle.freq_pulse_multih_irig(SPS), 420_000, ebn0, int(ebn0) + 11)
    got = CPMTrellisDetector(ARTM_64).detect(res["rows"])
    assert got.size == res["decisions"].size == 420_000 - 31
    assert np.array_equal(got, res["decisions"])
    _, res16 = _noisy_rows(oracle, oracle.ARTM_16, oracle.freq_pulse_multih_irig(SPS), 420_000, ebn0, int(ebn0) + 11)
    assert res["bit_errors"] < res16["bit_errors"]


@pytest.mark.gpu
@pytest.mark.parametrize("ebn0", [0.0, 4.0, 10.0])
def test_gpu_artm256_decisions_equal_sequential_oracle(oracle, ebn0, ctx_options):
    """The FULL ARTM trellis (notes/cpm/cpm.md:128-140: N_S = p M^(L-1) = 256, 64 matched filters per symbol) on the GPU:
    thread = state, one workgroup per chunk — every decision identical to the sequential C detector, in one call and with
    the state carried across ragged pieces; also with chunks of 512 calls and a 2-call warm-up (repairs that cross chunks)."""
    from waveforms_amd import device as dev
    from waveforms_amd.viterbi.cpm import ARTM_256, CPMTrellisDetector

    sym, res = _noisy_rows(oracle, oracle.ARTM_256, oracle.freq_pulse_multih_irig(SPS), 150_000, ebn0, int(ebn0) + 31)
    n = res["rows"].shape[0]
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        got = CPMTrellisDetector(ARTM_256).detect(res["rows"])
        assert got.size == res["decisions"].size and np.array_equal(got, res["decisions"])
        det = CPMTrellisDetector(ARTM_256)
        cuts = [0, 1, 2, 3, 40_001, n]
        got = np.concatenate([det.detect(res["rows"][a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(got, res["decisions"])
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_CHUNK_CALLS=512):
        det = CPMTrellisDetector(ARTM_256)
        got = det.detect(res["rows"], warmup=2)
        assert np.array_equal(got, res["decisions"])
        assert dev.viterbi_repaired(reset=True, ctx=det._ctx) > 100
    if ebn0 >= 10.0:
        assert res["bit_errors"] < 2e-3 * res["compared"]


@pytest.mark.gpu
def test_gpu_artm256_link_equals_oracle_chain(oracle):
    """The ARTM link with the 256-state detector (modulator, channel, 64-filter rows, detector, count — the staged front
    end: the one-kernel form holds 4 or 16 filters): error counts equal the sequential oracle chain's on the same bits and
    the same Philox noise."""
    from waveforms_amd.link import CPMLink
    from waveforms_amd.viterbi import cpm

    nsym, ebn0 = 200_000, 8.0
    link = CPMLink(nsym, SPS, "multih", spec=cpm.ARTM_256, fuse=10)
    link.run_block(ebn0, seed=1, stream_id=4)
    se, be, m = link.result()
    bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, nsym * 2)[0]
    sym = oracle.multih_mapper(bits)[0]
    noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(ebn0, SPS, 2), 1, 4, 0, (nsym + 1) * SPS)
    res = oracle.cpm_detection_run(sym, oracle.freq_pulse_multih_irig(SPS), SPS, oracle.ARTM_256, noise=noise)
    x = (res["decisions"] ^ res["truth"])[64:]
    assert m == x.size
    assert (se, be) == (int(np.count_nonzero(x)), int(np.unpackbits(x[:, None], axis=1).sum()))
    assert be > 0


@pytest.mark.gpu
def test_gpu_artm64_link_equals_oracle_chain_and_streams(oracle):
    """The ARTM link with the 64-state detector: same front end (16 filters per symbol), counts equal the oracle chain fed
    the same noise; pipelined blocks equal sequential ones; the stream in chunks equals the one-shot link; fewer bit
    errors than the 16-state design on the same block."""
    from waveforms_amd.link import CPMLink, CPMStream
    from waveforms_amd.viterbi import cpm

    nsym, ebn0 = 100_000, 6.0
    link = CPMLink(nsym, SPS, "multih", spec=cpm.ARTM_64, fuse=10)
    link.run_block(ebn0, seed=1, stream_id=0)
    se, be, m = link.result()
    bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, 2 * nsym)[0]
    sym = oracle.multih_mapper(bits)[0]
    noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(ebn0, SPS, 2), 1, 0, 0, (nsym + 1) * SPS)
    res = oracle.cpm_detection_run(sym, oracle.freq_pulse_multih_irig(SPS), SPS, oracle.ARTM_64, noise=noise)
    x = (res["decisions"] ^ res["truth"])[64:]
    assert m == x.size and (se, be) == (int(np.count_nonzero(x)), int(np.unpackbits(x[:, None], axis=1).sum())) and be > 0
    link16 = CPMLink(nsym, SPS, "multih", fuse=10)
    link16.run_block(ebn0, seed=1, stream_id=0)
    assert be < link16.result()[1]
    # two pipelined blocks (detector of block 0 beside the front end of block 1) == two sequential ones
    seq, pip = CPMLink(nsym, SPS, "multih", spec=cpm.ARTM_64, fuse=10), CPMLink(nsym, SPS, "multih", spec=cpm.ARTM_64, fuse=42)
    for lk in (seq, pip):
        for blk in range(2):
            lk.run_block(ebn0, seed=1, stream_id=blk, skip_bits=blk * nsym * 2)
    assert seq.result() == pip.result() and seq.result()[2] == 2 * m
    # the stream in chunks
    n2 = 4 * 5120 + 1777
    one = CPMLink(n2, SPS, "multih", spec=cpm.ARTM_64)
    one.run_block(ebn0, seed=4, stream_id=3)
    st = CPMStream(n2, 5120, SPS, waveform="multih", spec=cpm.ARTM_64)
    for c in range(st.nchunks):
        st.run_chunk(c, ebn0, seed=4, stream_id=3)
    assert st.result() == one.result() and one.result()[1] > 0
    assert st.run_pipelined(ebn0, seed=4, stream_id=3) == one.result()


@pytest.mark.gpu
@pytest.mark.parametrize("design", GPU_DESIGNS)
def test_gpu_every_supported_design_equals_oracle(oracle, design, detector_form):
    """Every trellis shape the kernel family accepts (M 2 / 4, Lp 1..3, full and reduced phase
    state, 2 .. 64 states: up to 16 in one DPP row or one lane, 17 .. 64 one wave per detector) on noisy rows and on
    unstructured random rows with exact ties."""
    from waveforms_amd.viterbi import cpm

    spec_o = _spec(oracle, *design, D=32 if design[0] == 2 else 20)
    spec_p = cpm.CPMDetectorSpec(M=spec_o.M, p=spec_o.p, K=spec_o.K, Lp=spec_o.Lp, NC=spec_o.NC, D=spec_o.D)
    rng = np.random.default_rng(77)
    n = 70_001
    rows = rng.standard_normal((n, spec_o.nfilt)) + 1j * rng.standard_normal((n, spec_o.nfilt))
    rows[:, 0] += 2.0                                               # a drift so that survivors merge
    rows[::501] = 0.0
    want = oracle.cpm_viterbi(spec_o).run(rows)
    det = cpm.CPMTrellisDetector(spec_p)
    got = np.concatenate([det.detect(rows[:30_000]), det.detect(rows[30_000:])])   # carried across calls
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_detector_reports_and_repairs_unmerged_chunks(oracle, detector_form):
    from waveforms_amd import _hip, device as dev
    from waveforms_amd.viterbi import cpm

    sym, res = _noisy_rows(oracle, oracle.ARTM_16, oracle.freq_pulse_multih_irig(SPS), 40_000, 3.0, 21)
    rows = _hip.to_device(res["rows"])
    out = _hip.zeros(40_000, "uint8")
    cfg, rot = cpm.ARTM_16.c_config(), _hip.to_device(cpm.rotation_table(cpm.ARTM_16))
    dev.viterbi_unmerged(reset=True)
    dev.viterbi_repaired(reset=True)
    n = res["rows"].shape[0]

    def launch(warmup):
        _hip.check(_hip.lib().wf_cpm_viterbi_detect(_hip.ctx(), ctypes.byref(cfg), _hip.ptr(rot), _hip.ptr(rows), n, warmup,
                                                    _hip.ptr(out), None, _hip.stream()))

    # 4 rows of warm-up cannot even fill the decision register.  With the repairs switched off (a context option) the call
    # must say so: the proof counts the chunks, and the host API refuses to return decisions
    _hip.set_option(_hip.ctx(), _hip.WF_OPT_DET_REPAIR, 1)
    _hip.set_default_option(_hip.WF_OPT_DET_REPAIR, 1)
    try:
        launch(4)
        assert dev.viterbi_unmerged(reset=True) > 0
        assert dev.viterbi_repaired(reset=True) == 0
        with pytest.raises(RuntimeError, match="unproven"):
            cpm.CPMTrellisDetector(cpm.ARTM_16).detect(res["rows"], warmup=4)
    finally:
        _hip.set_default_option(_hip.WF_OPT_DET_REPAIR, 0)
    got = cpm.CPMTrellisDetector(cpm.ARTM_16).detect(res["rows"], warmup=4)
    assert np.array_equal(got, res["decisions"])
    # The call as shipped repairs those chunks itself: their own calls again from the state the previous chunk ended with
    # until that trajectory meets the first launch's.  Every decision is then the sequential detector's, and proven.
    launch(4)
    assert dev.viterbi_unmerged(reset=True) == 0
    repaired = dev.viterbi_repaired(reset=True)
    assert repaired >= n // 512, repaired                         # practically every chunk
    D = cpm.ARTM_16.D
    assert np.array_equal(_hip.to_host(out)[D - 1:n], res["decisions"])


@pytest.mark.gpu
@pytest.mark.parametrize("design", GPU_DESIGNS)
def test_gpu_short_warmup_repaired_on_device_equals_oracle(oracle, design, detector_form):
    """Every trellis shape with a warm-up of 8 calls — far below the merge depth, so most chunks miss it: the
    repair launch makes the output the sequential detector's all the same (noisy rows with a drift, exact ties,
    carried state across two calls), or the call says which chunks it could not prove."""
    from waveforms_amd import _hip, device as dev
    from waveforms_amd.viterbi import cpm

    spec_o = _spec(oracle, *design, D=32 if design[0] == 2 else 20)
    spec_p = cpm.CPMDetectorSpec(M=spec_o.M, p=spec_o.p, K=spec_o.K, Lp=spec_o.Lp, NC=spec_o.NC, D=spec_o.D)
    rng = np.random.default_rng(5)
    n = 50_001
    rows = rng.standard_normal((n, spec_o.nfilt)) + 1j * rng.standard_normal((n, spec_o.nfilt))
    rows[:, 0] += 2.0
    rows[::333] = 0.0
    want = oracle.cpm_viterbi(spec_o).run(rows)
    det = cpm.CPMTrellisDetector(spec_p)
    got = np.concatenate([det.detect(rows[:20_000], warmup=8), det.detect(rows[20_000:], warmup=8)])
    assert np.array_equal(got, want)
    assert dev.viterbi_repaired(reset=True, ctx=det._ctx) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_repair_fuzz_random_designs_warmups_and_splits(oracle, seed, detector_form):
    """Seeded fuzz of the chunk-parallel detector + repair launch against the sequential oracle: a random trellis shape,
    burst length (ragged ends), warm-up from 8 calls up, drift of the first filter output (how fast survivors merge),
    exact-tie rows, and the burst cut into 1 .. 3 calls with the state carried between them."""
    from waveforms_amd.viterbi import cpm

    rng = np.random.default_rng(1000 + seed)
    design = GPU_DESIGNS[int(rng.integers(len(GPU_DESIGNS)))]
    spec_o = _spec(oracle, *design, D=int(rng.choice([8, 20, 32])))
    spec_p = cpm.CPMDetectorSpec(M=spec_o.M, p=spec_o.p, K=spec_o.K, Lp=spec_o.Lp, NC=spec_o.NC, D=spec_o.D)
    n = int(rng.integers(3_000, 90_000))
    rows = rng.standard_normal((n, spec_o.nfilt)) + 1j * rng.standard_normal((n, spec_o.nfilt))
    rows[:, 0] += float(rng.choice([0.0, 0.5, 2.0, 4.0]))
    rows[::int(rng.integers(50, 700))] = 0.0
    warm = int(rng.choice([8, 16, 24, 32, 48, 64]))
    cuts = sorted(int(c) for c in rng.integers(1, n, size=int(rng.integers(0, 3))))
    want = oracle.cpm_viterbi(spec_o).run(rows)
    det = cpm.CPMTrellisDetector(spec_p)
    got = np.concatenate([det.detect(part, warmup=warm) for part in np.split(rows, cuts) if part.shape[0]])
    assert np.array_equal(got, want), (design, n, warm, cuts)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [10, 2, 0])
@pytest.mark.parametrize("waveform,nsym", [("multih", 100_000), ("pcmfm", 60_000), ("multih", 777)])
def test_gpu_cpm_link_equals_oracle_chain(oracle, waveform, nsym, fuse, detector_form):
    """wf_cpm_link_run (PRBS -> mapper -> modulate -> Philox AWGN -> rows -> detector -> count)
    against the oracle chain fed the same noise: identical symbol and bit error counts."""
    from waveforms_amd.link import CPMLink

    link = CPMLink(nsym, SPS, waveform=waveform, fuse=fuse)      # fuse 10: modulator + channel + filters in one kernel; 2: channel inside the filter kernel
    spec = oracle.ARTM_16 if waveform == "multih" else oracle.PCMFM_SPEC
    pulse = oracle.freq_pulse_multih_irig(SPS) if waveform == "multih" else oracle.freq_pulse_pcmfm(SPS)
    bps = spec.lgM
    for ebn0, block in ((4.0, 0), (7.0, 3)):
        link.reset_counts()
        link.run_block(ebn0, seed=1, stream_id=block, skip_bits=block * nsym * bps)
        se, be, m = link.result()
        bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, (block + 1) * nsym * bps)[0][block * nsym * bps:]
        sym = oracle.multih_mapper(bits)[0] if waveform == "multih" else oracle.pcmfm_mapper(bits)
        noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(ebn0, SPS, bps), 1, block, 0, (nsym + 1) * SPS)
        res = oracle.cpm_detection_run(sym, pulse, SPS, spec, noise=noise)
        x = (res["decisions"] ^ res["truth"])[64:]
        assert m == x.size == res["compared"] - 64
        assert (se, be) == (int(np.count_nonzero(x)), int(np.unpackbits(x[:, None], axis=1).sum()))
        assert be > 0 or nsym < 1000 or ebn0 > 5.0


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,ebn0", [("pcmfm", 4.0), ("multih", 8.0)])
def test_gpu_cpm_link_at_operating_point_warmup_equals_oracle_chain(oracle, waveform, ebn0, detector_form):
    """The link as bench.py runs it — chunk warm-up of the operating point (PCM/FM 64 calls, ARTM 48 from 8 dB), where
    dozens of chunks per block miss the warm-up and are repaired by the detector's second launch — over 2e6 symbols:
    symbol and bit error counts equal the sequential oracle chain's on the same bits and the same Philox noise."""
    from waveforms_amd import device as dev
    from waveforms_amd.link import CPMLink, operating_point_warmup

    nsym = 2_000_000
    w = operating_point_warmup(waveform, ebn0)
    assert 0 < w <= 64
    link = CPMLink(nsym, SPS, waveform=waveform, warmup=w, private_ctx=True)
    spec = oracle.ARTM_16 if waveform == "multih" else oracle.PCMFM_SPEC
    pulse = oracle.freq_pulse_multih_irig(SPS) if waveform == "multih" else oracle.freq_pulse_pcmfm(SPS)
    bps = spec.lgM
    dev.viterbi_repaired(reset=True, ctx=link._ctx)
    link.run_block(ebn0, seed=1, stream_id=7)
    se, be, m = link.result()                                   # (raises if a chunk was left unproven)
    assert dev.viterbi_repaired(reset=True, ctx=link._ctx) > 0
    bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, nsym * bps)[0]
    sym = oracle.multih_mapper(bits)[0] if waveform == "multih" else oracle.pcmfm_mapper(bits)
    noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(ebn0, SPS, bps), 1, 7, 0, (nsym + 1) * SPS)
    res = oracle.cpm_detection_run(sym, pulse, SPS, spec, noise=noise)
    x = (res["decisions"] ^ res["truth"])[64:]
    assert m == x.size
    want = (int(np.count_nonzero(x)), int(np.unpackbits(x[:, None], axis=1).sum()))
    assert (se, be) == want
    assert be > 0


@pytest.mark.gpu
def test_gpu_multih_full_size_noiseless_and_ber(oracle):
    """BASELINE configs[2] at full size (1e7 quaternary symbols = 2e7 PN23 bits): no noise -> zero errors;
    at 10 dB the link's symbol and bit error counts EQUAL the sequential oracle chain's on the same PN23 bits
    and the same Philox noise, count for count (one ~20 s CPU run of the C oracle), and the BER sits above the
    minimum-distance bound."""
    from waveforms_amd.link import CPMLink

    nsym = 10_000_000
    link = CPMLink(nsym, SPS, waveform="multih")
    link.run_block(None)
    se, be, m = link.result()
    assert (se, be) == (0, 0) and m == nsym - 31 - 64
    link.reset_counts()
    link.run_block(10.0, seed=1, stream_id=2)
    se, be, m = link.result()
    ber = be / (2 * m)
    q = 0.5 * math.erfc(math.sqrt(1.2957 * 10.0) / math.sqrt(2))
    assert q < ber, (ber, q)
    spec = oracle.ARTM_16
    bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, nsym * 2)[0]
    sym = oracle.multih_mapper(bits)[0]
    noise = oracle.philox_awgn(oracle.cpm_sigma_for_ebn0(10.0, SPS, 2), 1, 2, 0, (nsym + 1) * SPS)
    res = oracle.cpm_detection_run(sym, oracle.freq_pulse_multih_irig(SPS), SPS, spec, noise=noise)
    del noise
    x = (res["decisions"] ^ res["truth"])[64:]
    assert m == x.size
    want = (int(np.count_nonzero(x)), int(np.unpackbits(x[:, None], axis=1).sum()))
    assert (se, be) == want
    del link
    # ... and the same in the configuration bench.py times (fuse 42: blocks software-pipelined, the detector of a block
    # beside the next block's front end; two blocks so that the overlap happens), at bench.py's operating-point warm-up
    from waveforms_amd.link import operating_point_warmup

    for warmup in (0, operating_point_warmup("multih", 10.0)):
        piped = CPMLink(nsym, SPS, waveform="multih", fuse=42, warmup=warmup)
        piped.run_block(10.0, seed=1, stream_id=2)
        piped.run_block(10.0, seed=1, stream_id=2)
        se2, be2, m2 = piped.result()                 # (raises if a detector chunk was left unproven)
        assert (se2, be2, m2) == (2 * want[0], 2 * want[1], 2 * m), warmup
        del piped


@pytest.mark.gpu
def test_gpu_cpm_ber_sweep_through_the_bert_harness():
    """waveforms.bert serves the generic CPM detector's waveforms too (SweepPlan.waveform): three blocks
    in flight on separate streams, library-default warm-up (safe at any Eb/N0), curve monotone and
    above the minimum-distance term Q(sqrt(1.2957 Eb/N0)) by a small factor."""
    from waveforms.bert import SweepPlan, ber_sweep

    ebn0 = [6.0, 8.0, 10.0]
    plan = SweepPlan(ebn0_db=ebn0, blocks_per_point=6, nsym=1 << 21, waveform="multih")
    assert plan.bits_per_symbol == 2 and plan.skip_bits(3) == 3 * (1 << 22)
    counts = ber_sweep(plan, rank=0, world=1)
    ber = counts[:, 1] / (2.0 * counts[:, 2])
    assert (counts[:, 2] == 6 * ((1 << 21) - 31 - 64)).all()
    assert ber[0] > ber[1] > ber[2] > 0
    q = [0.5 * math.erfc(math.sqrt(1.2957 * 10 ** (e / 10) / 2)) for e in ebn0]
    for b, lo in zip(ber, q):
        assert lo < b < 6 * lo, (ber, q)
    pc = ber_sweep(SweepPlan(ebn0_db=[4.0, 7.0], blocks_per_point=3, nsym=1 << 20, waveform="pcmfm"), rank=0, world=1)
    assert pc[0, 1] > pc[1, 1] > 0 and (pc[:, 0] == pc[:, 1]).all()          # binary: symbol errors = bit errors


@pytest.mark.gpu
def test_gpu_cpm_detection_example(oracle):
    """examples/cpm_detection.py (the soqpsk_detection example's counterpart for the waveforms the
    reference only modulates) reproduces the oracle chain's counts through the public API."""
    import importlib.util
    from pathlib import Path

    path = Path(__file__).resolve().parent.parent / "examples" / "cpm_detection.py"
    spec_ = importlib.util.spec_from_file_location("cpm_detection_example", path)
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    got = mod.run(ebn0_db=7.0, nsym=20000, pn_degree=17, seed=1)
    rng = np.random.Generator(np.random.PCG64(seed=1))
    pn = oracle.pn_sequence(17)
    for label, spec, pulse, bps in (("ARTM multi-h", oracle.ARTM_16, oracle.freq_pulse_multih_irig(SPS), 2),
                                    ("PCM/FM", oracle.PCMFM_SPEC, oracle.freq_pulse_pcmfm(SPS), 1)):
        bits = np.resize(pn, 20000 * bps)
        sym = oracle.multih_mapper(bits)[0] if bps == 2 else oracle.pcmfm_mapper(bits)
        noise = oracle.numpy_awgn(oracle.cpm_sigma_for_ebn0(7.0, SPS, bps), (sym.size + 1) * SPS, rng)
        res = oracle.cpm_detection_run(sym, pulse, SPS, spec, noise=noise)
        assert got[label] == (res["sym_errors"], res["bit_errors"], res["compared"]), label
    assert got["ARTM multi-h"][1] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,chunk", [("multih", 5120), ("multih", 8192), ("pcmfm", 6144)])
def test_gpu_cpm_stream_in_chunks_equals_one_shot(waveform, chunk, detector_form):
    """wf_cpm_link_stream_chunk: the CPM link over a stream in chunks (detector state and modulator phase carried,
    everything else re-generated as a halo) makes exactly the one-shot link's decisions and counts — ragged
    last chunk, with and without noise."""
    from waveforms_amd.link import CPMLink, CPMStream

    nsym = 4 * chunk + 1777
    one = CPMLink(nsym, SPS, waveform=waveform)
    st = CPMStream(nsym, chunk, SPS, waveform=waveform)
    assert st.nchunks == 5
    for ebn0, sid in ((None, 0), (6.0, 3)):
        one.reset_counts()
        one.run_block(ebn0, seed=4, stream_id=sid)
        want = one.result()
        lo = one.layout()
        want_dec = one.workspace[lo["off_decisions"]:lo["off_decisions"] + lo["calls"]].cpu().numpy()
        st.reset()
        got_dec = []
        for c in range(st.nchunks):
            st.run_chunk(c, ebn0, seed=4, stream_id=sid)
            info = st.chunk_info(c)
            got_dec.append(st.workspace[info["off_decisions"]:info["off_decisions"] + info["calls"]].cpu().numpy())
        assert st.result() == want
        assert np.array_equal(np.concatenate(got_dec), want_dec)
        assert ebn0 is None or want[1] > 0
        # ... and as the two-stream chunk pipeline (wf_cpm_link_stream_chunk_phase: the detector of chunk c beside the
        # front end of chunk c + 1, carries handed over by events)
        assert st.run_pipelined(ebn0, seed=4, stream_id=sid) == want
    with pytest.raises(ValueError):
        CPMStream(nsym, 1000, SPS, waveform=waveform)


@pytest.mark.gpu
@pytest.mark.parametrize("waveform", ["multih", "pcmfm"])
def test_gpu_cpm_link_pipelined_blocks_equal_sequential_blocks(waveform, detector_form):
    """wf_cpm_link_config.fuse bit 5: a block's detector and error count on the context's side stream, beside the front
    end of the next block (two sets of intermediates).  Block for block and in total the counts are those of the
    sequential link."""
    from waveforms_amd.link import CPMLink

    nsym = 150_001
    seq = CPMLink(nsym, SPS, waveform=waveform, fuse=10)
    pip = CPMLink(nsym, SPS, waveform=waveform, fuse=42, private_ctx=True)
    two = CPMLink(nsym, SPS, waveform=waveform, fuse=26, private_ctx=True)      # bit 4: PRBS and mapper as two kernels, not the one launch
    assert pip.workspace_bytes >= 2 * seq.workspace_bytes
    out = []
    for link in (seq, pip, two):
        got = []
        for k in range(5):
            link.reset_counts()
            link.run_block(7.0, seed=2, stream_id=k, skip_bits=31 * k)
            got.append(link.result())
        link.reset_counts()
        for k in range(8):
            link.run_block(6.0 + (k % 3), seed=9, stream_id=50 + k, skip_bits=k, event_slot=0 if k == 7 else -1)
        got.append(link.result())
        out.append(got)
    assert out[0] == out[1] == out[2]
    assert out[0][-1][1] > 0


# ------------------------------------------------------------------ round 6: the matched filters inside the detector
def _link_form(link):
    from waveforms_amd import _hip

    info = (ctypes.c_int * 4)()
    _hip.check(_hip.lib().wf_cpm_link_form(link._ctx, ctypes.byref(link.cfg), info))
    return list(info)


@pytest.mark.gpu
def test_gpu_samples_form_link_equals_rows_form(ctx_options):
    """wf_cpm_link_config.fuse bit 7: the front end stores the noisy samples (128 B per symbol) and the detector's lanes run
    the 16 matched filters themselves — against the paired one-kernel front end with its rows in HBM (fuse bits 1 + 3 + 6):
    every decision identical, without noise, at 10 dB and at 3 dB with a warm-up short enough that thousands of chunks go to
    the repair (which rebuilds its rows from the same samples), chunk boundaries proven once more behind the repairs; a
    chunk length that is not a multiple of 64; and two blocks through the pipelined form (bit 5: two side streams)."""
    from waveforms_amd import device as dev
    from waveforms_amd.link import CPMLink

    nsym = 2_400_001
    if True:
        for ebn0, warm, chunk in ((None, 48, 0), (10.0, 48, 0), (10.0, 0, 0), (3.0, 16, 0), (6.0, 32, 80), (6.0, 32, 208)):
            with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_CHUNK_CALLS=chunk, WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):
                a = CPMLink(nsym, SPS, waveform="multih", fuse=10, warmup=warm, private_ctx=True)
                b = CPMLink(nsym, SPS, waveform="multih", fuse=10 | 128, warmup=warm, private_ctx=True)
                assert a.paired_templates and b.paired_templates
                fa, fb = _link_form(a), _link_form(b)
                assert fa[0] == 1 and fb[0] == 2 and fb[1] == 1, (fa, fb)
                if chunk:
                    assert fb[2] == chunk
                for link in (a, b):
                    link.run_block(ebn0, seed=5, stream_id=3, skip_bits=11)
                la, lb = a.layout(), b.layout()
                da = a.workspace[la["off_decisions"]:la["off_decisions"] + la["calls"]].cpu().numpy()
                db = b.workspace[lb["off_decisions"]:lb["off_decisions"] + lb["calls"]].cpu().numpy()
                assert np.array_equal(da, db), (ebn0, warm, chunk, int(np.count_nonzero(da != db)))
                ra, rb = a.result(), b.result()          # (raises if a chunk boundary was left unproven)
                assert ra == rb and ra[2] == nsym - 31 - 64
                if ebn0 is None:
                    assert ra[:2] == (0, 0)
                rep = dev.viterbi_repaired(reset=True, ctx=b._ctx)
                if warm == 16:
                    assert rep > 1000, rep
                del a, b
    # the pipelined form bench.py times: consecutive blocks' detectors on two side streams, each with its own proof records
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):
        one = CPMLink(nsym, SPS, waveform="multih", fuse=10 | 128, warmup=48, private_ctx=True)
        piped = CPMLink(nsym, SPS, waveform="multih", fuse=42 | 128, warmup=48, private_ctx=True)
        assert _link_form(piped)[0] == 2
        for blk in range(5):
            one.run_block(8.0, seed=2, stream_id=blk)
            piped.run_block(8.0, seed=2, stream_id=blk)
        assert one.result() == piped.result() and piped.result()[1] > 0


@pytest.mark.gpu
def test_gpu_samples_front_end_stores_the_samples_the_paired_bank_saw(ctx_options):
    """mod_chan_samples_kernel against the staged kernels: its samples equal wf_cpm_modulate_c128 + wf_awgn_c128 to 1e-11, and
    wf_cpm_mf_rows_c128 over them gives the rows of the one-kernel front end (both forms) to 1e-11."""
    with ctx_options(WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):           # (the library's own floor is 6e6 calls)
        _samples_front_end_body(2_200_000)


def _samples_front_end_body(nsym):
    from waveforms_amd import _hip, device as dev
    from waveforms_amd.link import CPMLink
    from waveforms_amd.viterbi import cpm

    rows_link = CPMLink(nsym, SPS, waveform="multih", fuse=10, private_ctx=True)
    samp_link = CPMLink(nsym, SPS, waveform="multih", fuse=10 | 128, private_ctx=True)
    staged = CPMLink(nsym, SPS, waveform="multih", fuse=0, private_ctx=True)
    for link in (rows_link, samp_link, staged):
        link.run_block(7.0, seed=3, stream_id=9, skip_bits=5)
    assert _link_form(samp_link)[0] == 2 and _link_form(staged)[0] == 0
    ls, lt, lr = samp_link.layout(), staged.layout(), rows_link.layout()
    n = ls["signal_len"]
    got = samp_link.workspace[ls["off_signal"]:ls["off_signal"] + 16 * n].view(torch_f64()).cpu().numpy()
    ref = staged.workspace[lt["off_signal"]:lt["off_signal"] + 16 * n].view(torch_f64()).cpu().numpy()
    assert np.abs(ref).max() > 1.0
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    calls = ls["calls"]
    t = cpm.matched_filter_templates(__import__("waveforms_amd.cpm.multih", fromlist=["x"]).freq_pulse_multih_irig(SPS), SPS, cpm.ARTM_16)
    sig = samp_link.workspace[ls["off_signal"]:ls["off_signal"] + 16 * n].view(torch_f64()).view(n, 2)
    rows = dev.cpm_mf_rows(sig, _hip.to_device(t), ls["start0"], SPS, calls).cpu().numpy().reshape(-1)
    want = rows_link.workspace[lr["off_rows"]:lr["off_rows"] + calls * 16 * 16].view(torch_f64()).cpu().numpy()
    np.testing.assert_allclose(rows, want, rtol=0, atol=1e-11)
    assert samp_link.result() == rows_link.result()


@pytest.mark.gpu
@pytest.mark.parametrize("ebn0", [2.0, 9.0])
def test_gpu_detect_samples_equals_sequential_oracle(oracle, ebn0, ctx_options):
    """wf_cpm_viterbi_detect_samples (the C-ABI form of the same launch) on the oracle's own received samples against the
    sequential C detector over the oracle's rows: every decision identical — the library's warm-up and a short one
    (hundreds of chunks repaired from the samples); and the entry point declines (returns 1: rows + detector) what its
    launch does not serve."""
    from waveforms_amd import device as dev
    from waveforms_amd.viterbi.cpm import ARTM_16, PCMFM_10, CPMTrellisDetector, matched_filter_templates

    nsym = 2_300_000
    pulse = oracle.freq_pulse_multih_irig(SPS)
    sym, res = _noisy_rows(oracle, oracle.ARTM_16, pulse, nsym, ebn0, int(ebn0) + 21)
    geo = res["geometry"]
    T = matched_filter_templates(pulse, SPS, ARTM_16)
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):
        det = CPMTrellisDetector(ARTM_16)
        got = det.detect_samples(res["received"], T, geo["start0"], SPS, geo["ncalls"])
        assert det.samples_form and got.size == res["decisions"].size
        assert np.array_equal(got, res["decisions"])
        det = CPMTrellisDetector(ARTM_16)
        got = det.detect_samples(res["received"], T, geo["start0"], SPS, geo["ncalls"], warmup=8)
        assert np.array_equal(got, res["decisions"]) and dev.viterbi_repaired(reset=True, ctx=det._ctx) > 100
        # a burst below the lane form's reach, and a trellis without the specialisation: declined, same decisions through the rows
        det = CPMTrellisDetector(ARTM_16)
        small = det.detect_samples(res["received"], T, geo["start0"], SPS, 300_000)
        assert not det.samples_form and np.array_equal(small, res["decisions"][:300_000 - 31])
    p2 = oracle.freq_pulse_pcmfm(SPS)
    sym2, res2 = _noisy_rows(oracle, oracle.PCMFM_SPEC, p2, 50_000, ebn0, 5)
    det = CPMTrellisDetector(PCMFM_10)
    got2 = det.detect_samples(res2["received"], matched_filter_templates(p2, SPS, PCMFM_10), res2["geometry"]["start0"], SPS, res2["geometry"]["ncalls"])
    assert not det.samples_form and np.array_equal(got2, res2["decisions"])


@pytest.mark.gpu
def test_gpu_detect_samples_carries_its_state(oracle, ctx_options):
    """One burst through wf_cpm_viterbi_detect_samples in two launches (d_state carries metrics, phase indices and decision
    registers; the second piece's first call is even, so its calls take the template columns of their places in the burst):
    the decisions of the whole burst in one sequential pass."""
    from waveforms_amd.viterbi.cpm import ARTM_16, CPMTrellisDetector, matched_filter_templates

    nsym, k0 = 4_300_000, 2_150_000
    pulse = oracle.freq_pulse_multih_irig(SPS)
    sym, res = _noisy_rows(oracle, oracle.ARTM_16, pulse, nsym, 5.0, 77)
    geo = res["geometry"]
    assert geo["start0"] == 0
    T = matched_filter_templates(pulse, SPS, ARTM_16)
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):
        det = CPMTrellisDetector(ARTM_16)
        first = det.detect_samples(res["received"], T, 0, SPS, k0)
        assert det.samples_form
        second = det.detect_samples(res["received"][SPS * k0:], T, 0, SPS, geo["ncalls"] - k0)
        assert det.samples_form
    assert np.array_equal(np.concatenate([first, second]), res["decisions"])


@pytest.mark.gpu
def test_gpu_samples_form_of_the_256_state_link_equals_rows_form(ctx_options):
    """The full ARTM trellis (256 states, 64 matched filters: a row is 1 KB per symbol) with fuse bit 7: the front end stores the
    noisy samples and every detector workgroup forms the filter outputs of a batch itself (quad form, the k-ascending chain of
    wf_cpm_mf_rows_c128) — the same decisions and counts as the link that writes rows, at 10 dB and at 3 dB with a short
    warm-up (repairs run from the samples too), chunk boundaries proven once more; and through the pipelined form."""
    from waveforms_amd import device as dev
    from waveforms_amd.link import CPMLink
    from waveforms_amd.viterbi import cpm

    nsym = 600_001
    for ebn0, warm in ((10.0, 0), (3.0, 16), (None, 0)):
        with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_CHUNK_CALLS=512):
            a = CPMLink(nsym, SPS, waveform="multih", spec=cpm.ARTM_256, fuse=10, warmup=warm, private_ctx=True)
            b = CPMLink(nsym, SPS, waveform="multih", spec=cpm.ARTM_256, fuse=10 | 128, warmup=warm, private_ctx=True)
            fa, fb = _link_form(a), _link_form(b)
            assert fa[0] == 0 and fb[0] == 2 and fb[1] == 3, (fa, fb)
            for link in (a, b):
                link.run_block(ebn0, seed=5, stream_id=3, skip_bits=11)
            la, lb = a.layout(), b.layout()
            da = a.workspace[la["off_decisions"]:la["off_decisions"] + la["calls"]].cpu().numpy()
            db = b.workspace[lb["off_decisions"]:lb["off_decisions"] + lb["calls"]].cpu().numpy()
            assert np.array_equal(da, db), (ebn0, warm, int(np.count_nonzero(da != db)))
            assert a.result() == b.result()
            if warm == 16:
                assert dev.viterbi_repaired(reset=True, ctx=b._ctx) > 50
            del a, b
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        one = CPMLink(nsym, SPS, waveform="multih", spec=cpm.ARTM_256, fuse=10 | 128, private_ctx=True)
        piped = CPMLink(nsym, SPS, waveform="multih", spec=cpm.ARTM_256, fuse=42 | 128, private_ctx=True)
        for blk in range(3):
            one.run_block(8.0, seed=2, stream_id=blk)
            piped.run_block(8.0, seed=2, stream_id=blk)
        assert one.result() == piped.result() and piped.result()[1] > 0


@pytest.mark.gpu
def test_gpu_cpm_stream_in_the_samples_form_equals_one_shot(ctx_options):
    """The streaming ARTM link with fuse bit 7: a chunk's noisy samples take its rows' place in the workspace and the detector
    (state carried from chunk to chunk) runs the matched filters — decisions and counts of the one-shot link over the whole
    stream, chunk by chunk and as the two-stream chunk pipeline; the last, short chunk falls back to rows (below the lane
    form's reach), so both forms hand the same carries to each other."""
    from waveforms_amd.link import CPMLink, CPMStream

    chunk = 3 << 20
    nsym = 2 * chunk + 1_200_000
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1, WF_OPT_CPM_SAMPLES_MIN_CALLS=1 << 21):
        one = CPMLink(nsym, SPS, waveform="multih", fuse=10 | 128)
        st = CPMStream(nsym, chunk, SPS, waveform="multih", fuse=10 | 128)
        rows_st = CPMStream(nsym, chunk, SPS, waveform="multih", fuse=10)
        assert st.nchunks == 3 and _link_form(one)[0] == 2
        for ebn0, sid in ((7.0, 3), (None, 0)):
            one.reset_counts()
            one.run_block(ebn0, seed=4, stream_id=sid)
            want = one.result()
            lo = one.layout()
            want_dec = one.workspace[lo["off_decisions"]:lo["off_decisions"] + lo["calls"]].cpu().numpy()
            st.reset()
            got_dec = []
            for c in range(st.nchunks):
                st.run_chunk(c, ebn0, seed=4, stream_id=sid)
                info = st.chunk_info(c)
                got_dec.append(st.workspace[info["off_decisions"]:info["off_decisions"] + info["calls"]].cpu().numpy())
            assert st.result() == want
            assert np.array_equal(np.concatenate(got_dec), want_dec)
            assert st.run_pipelined(ebn0, seed=4, stream_id=sid) == want
            assert rows_st.run(ebn0, seed=4, stream_id=sid) == want
            assert ebn0 is None or want[1] > 0
