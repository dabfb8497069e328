"""The flows of the reference's examples/irig_comparison.py and examples/soqpsk_example.py, each as one test:
what those scripts compute and hand to matplotlib (modulated signals of the three IRIG-106 waveforms at sps 20
and of the four SOQPSK pulses at sps 8, the PSDs, the eye traces, the constellation samples, the phase tree).
Golden arrays come from the reference itself (tests/golden/make_example_flows_golden.py: its modulator, and
the Line2D data its own viz/eye.py and viz/tree.py create)."""
import numpy as np
import pytest

NFFT = 1024
IRIG = (("pcmfm", "SimpleTrellis2", 1), ("soqpsk", "SOQPSKTrellis4x2DiffEncoded", 1), ("multih", "SimpleTrellis4", 2))
_W = np.hanning(NFFT)


def _density(fs):      # Axes.psd's default (scale_by_freq=True) against the "spectrum" scaling of the PSD product
    return _W.sum() ** 2 / (_W ** 2).sum() / fs


def _irig_pulse(name, sps):
    from waveforms.cpm.multih import freq_pulse_multih_irig
    from waveforms.cpm.pcmfm import freq_pulse_pcmfm
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_tg

    return {"pcmfm": lambda: freq_pulse_pcmfm(sps=sps, order=6), "soqpsk": lambda: freq_pulse_soqpsk_tg(sps=sps),
            "multih": lambda: freq_pulse_multih_irig(sps=sps)}[name]()


def _sq_pulses(sps=8):
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_a, freq_pulse_soqpsk_b, freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg

    return (("B", freq_pulse_soqpsk_b(sps=sps)), ("TG", freq_pulse_soqpsk_tg(sps=sps)), ("A", freq_pulse_soqpsk_a(sps=sps)),
            ("MIL", freq_pulse_soqpsk_mil(sps=sps)))


def _eye_reference(time, sig, sps, modulo, t_offset):
    """Slicing of waveforms/viz/eye.py:40-55 in NumPy (checked against the golden traces below, then used as the
    full-size yardstick for the device kernel)."""
    n = (time.size - 1) // (sps * modulo)
    idx = np.arange(n)[:, None] * sps * modulo + np.arange(sps * modulo + 1)[None, :]
    return time[idx] - time[idx[:, :1]] + t_offset, sig.real[idx], sig.imag[idx]


# ------------------------------------------------------------------ CPU: oracle + host design code vs the reference
def test_irig_comparison_flow_oracle(oracle, golden):
    g = golden("example_flows")
    sps = 20
    bits = np.unpackbits(np.packbits(oracle.pn_sequence(15)))
    assert bits.size == int(g["irig_nbits"][0])
    for name, trellis, bpsym in IRIG:
        pulse = _irig_pulse(name, sps)
        np.testing.assert_allclose(pulse, g[f"irig_{name}_pulse"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.cumsum(pulse) / sps, g[f"irig_{name}_q"], rtol=0, atol=1e-15)
        sym = oracle.fsm_encode(trellis, bits)[0]
        assert sym.size == int(g[f"irig_{name}_nsym"][0]) and int(sym.astype(np.int64).sum()) == int(g[f"irig_{name}_symsum"][0])
        h = g[f"irig_{name}_mod_index"]
        _t, sig = oracle.cpm_modulate(sym, h if h.size > 1 else float(h[0]), pulse, sps)
        np.testing.assert_allclose(sig[:48], g[f"irig_{name}_sig_head"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sig[-48:], g[f"irig_{name}_sig_tail"], rtol=0, atol=1e-9)
        assert abs(sig.sum() - g[f"irig_{name}_sig_sum"][0]) < 1e-7
        f, p = oracle.psd_welch(sig, sps, bpsym, NFFT)
        np.testing.assert_allclose(f, g[f"irig_{name}_freqs"], rtol=0, atol=1e-13)
        want = g[f"irig_{name}_pxx"]
        np.testing.assert_allclose(p, want, rtol=1e-9, atol=1e-13 * want.max())


def test_soqpsk_example_flow_oracle(oracle, golden):
    g = golden("example_flows")
    sps = 8
    bits = np.unpackbits(np.packbits(oracle.pn_sequence(13)))
    assert bits.size == int(g["sq_nbits"][0])
    sym = oracle.fsm_encode("SOQPSKTrellis4x2DiffEncoded", bits)[0]
    assert int(sym.astype(np.int64).sum()) == int(g["sq_symsum"][0])
    for label, pulse in _sq_pulses(sps):
        np.testing.assert_allclose(pulse, g[f"sq_{label}_pulse"], rtol=0, atol=1e-15)
        t, sig = oracle.cpm_modulate(sym, 0.25, pulse, sps)
        np.testing.assert_allclose(sig[:48], g[f"sq_{label}_sig_head"], rtol=0, atol=1e-12)
        assert abs(sig.sum() - g[f"sq_{label}_sig_sum"][0]) < 1e-7
        f, p = oracle.psd_welch(sig, sps, 1, NFFT)
        want = g[f"sq_{label}_pxx"]
        np.testing.assert_allclose(f, g["sq_freqs"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(p * _density(sps), want, rtol=1e-9, atol=1e-13 * want.max())
        # the eye traces the reference's plot_eye_diagram drew
        t = t / 2
        n4 = t.size // 4
        et, ere, eim = _eye_reference(t[:n4], sig[:n4], sps, 4, 0 if label == "MIL" else 1 / sps / 4)
        assert tuple(ere.shape) == tuple(int(v) for v in g[f"sq_{label}_eye_shape"])
        np.testing.assert_allclose(et[0], g[f"sq_{label}_eye_t0"], rtol=0, atol=1e-12)
        for got, key in ((ere, "re"), (eim, "im")):
            np.testing.assert_allclose(got[:12], g[f"sq_{label}_eye_{key}_head"], rtol=0, atol=1e-11)
            np.testing.assert_allclose(got[-4:], g[f"sq_{label}_eye_{key}_tail"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(got.sum(axis=0), g[f"sq_{label}_eye_{key}_colsum"], rtol=0, atol=1e-7)
        if label == "TG":
            q = np.zeros_like(sig)
            q[sps:] += sig.real[:-sps]
            q[:] += sig.imag * 1j
            np.testing.assert_allclose(q[sps::sps * 2][1:][:1024], g["sq_constellation"], rtol=0, atol=1e-9)


# ------------------------------------------------------------------ GPU: the same flows through the public API
@pytest.mark.gpu
def test_gpu_irig_comparison_flow_equals_reference(golden):
    """PNSequence -> TrellisEncoder -> cpm_modulate at sps 20 (60-, 161- and 61-tap pulses, one and two modulation
    indices) -> Welch PSD kernel with the example's scaling (signal * sqrt(bpsym), Fs = sps / bpsym, spectrum)."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER
    from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER
    from waveforms.cpm.soqpsk import SOQPSK_DENOM, SOQPSK_NUMER
    from waveforms.cpm.trellis import model
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.glfsr import PNSequence
    from waveforms.viz import power_spectral_density

    g = golden("example_flows")
    sps = 20
    bits = np.unpackbits(np.frombuffer(np.packbits(PNSequence(15).generate_sequence()), dtype=np.uint8))
    index = {"pcmfm": PCMFM_NUMER / PCMFM_DENOM, "soqpsk": SOQPSK_NUMER / SOQPSK_DENOM, "multih": MULTIH_IRIG_NUMER / MULTIH_IRIG_DENOM}
    for name, trellis, bpsym in IRIG:
        np.testing.assert_allclose(np.atleast_1d(np.asarray(index[name], dtype=np.float64)), g[f"irig_{name}_mod_index"], rtol=0, atol=0)
        sym = TrellisEncoder(getattr(model, trellis))(bits)
        assert int(np.asarray(sym, dtype=np.int64).sum()) == int(g[f"irig_{name}_symsum"][0])
        _t, sig = cpm_modulate(symbols=sym, mod_index=index[name], pulse_filter=_irig_pulse(name, sps), sps=sps)
        np.testing.assert_allclose(sig[:48], g[f"irig_{name}_sig_head"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sig[-48:], g[f"irig_{name}_sig_tail"], rtol=0, atol=1e-9)
        assert abs(sig.sum() - g[f"irig_{name}_sig_sum"][0]) < 1e-7
        f, p = power_spectral_density(sig, sps, bpsym, NFFT)
        want = g[f"irig_{name}_pxx"]
        np.testing.assert_allclose(f, g[f"irig_{name}_freqs"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(p, want, rtol=1e-9, atol=1e-13 * want.max())


@pytest.mark.gpu
def test_gpu_soqpsk_example_flow_equals_reference(golden):
    """The four SOQPSK pulses (129, 65, 65 and 9 taps at sps 8) through cpm_modulate, the PSD, the eye-trace kernel
    (against the traces the reference's plot_eye_diagram drew), the constellation samples and the phase tree (against
    the curves the reference's generate_cpm_phase_tree drew)."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded
    from waveforms.glfsr import PNSequence
    from waveforms.viz import constellation_data, cpm_phase_tree_signal, eye_diagram_data, phase_tree_data, power_spectral_density

    g = golden("example_flows")
    sps = 8
    bits = np.unpackbits(np.packbits(PNSequence(13).generate_sequence()))
    precoder = TrellisEncoder(SOQPSKTrellis4x2DiffEncoded)
    sym = precoder(bits)
    assert int(np.asarray(sym, dtype=np.int64).sum()) == int(g["sq_symsum"][0])
    for label, pulse in _sq_pulses(sps):
        t, sig = cpm_modulate(symbols=sym, mod_index=1 / 4, pulse_filter=pulse, sps=sps)
        np.testing.assert_allclose(sig[:48], g[f"sq_{label}_sig_head"], rtol=0, atol=1e-12)
        assert abs(sig.sum() - g[f"sq_{label}_sig_sum"][0]) < 1e-7
        f, p = power_spectral_density(sig, sps, 1, NFFT)
        want = g[f"sq_{label}_pxx"]
        np.testing.assert_allclose(p * _density(sps), want, rtol=1e-9, atol=1e-13 * want.max())
        t = t / 2
        n4 = t.size // 4
        off = 0 if label == "MIL" else 1 / sps / 4
        et, ere, eim = eye_diagram_data(t[:n4], sig[:n4], sps=sps, modulo=4, t_offset=off)
        assert tuple(ere.shape) == tuple(int(v) for v in g[f"sq_{label}_eye_shape"])
        np.testing.assert_allclose(et[0], g[f"sq_{label}_eye_t0"], rtol=0, atol=1e-12)
        assert np.abs(et - et[0]).max() <= float(g[f"sq_{label}_eye_t_spread"][0]) + 1e-12
        for got, key in ((ere, "re"), (eim, "im")):
            np.testing.assert_allclose(got[:12], g[f"sq_{label}_eye_{key}_head"], rtol=0, atol=1e-11)
            np.testing.assert_allclose(got[-4:], g[f"sq_{label}_eye_{key}_tail"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(got.sum(axis=0), g[f"sq_{label}_eye_{key}_colsum"], rtol=0, atol=1e-7)
        wt, wre, wim = _eye_reference(t[:n4], sig[:n4], sps, 4, off)           # every trace, against NumPy on the same signal
        assert np.array_equal(ere, wre) and np.array_equal(eim, wim)
        np.testing.assert_allclose(et, wt, rtol=0, atol=1e-12)
        if label == "TG":
            q = np.zeros_like(sig)
            q[sps:] += sig.real[:-sps]
            q[:] += sig.imag * 1j
            re, im = constellation_data(q[sps::sps * 2][1:])
            assert re.size == 1024
            np.testing.assert_allclose(re + 1j * im, g["sq_constellation"], rtol=0, atol=1e-9)
    for label in ("MIL", "A16"):
        pulse = g[f"sq_tree_{label}_pulse"]
        tree_sig, length = cpm_phase_tree_signal(pulse, 1 / 4, precoder, sps)
        tt, traces = phase_tree_data(tree_sig, sps, modulo=length)
        np.testing.assert_allclose(tt, g[f"sq_tree_{label}_t"], rtol=0, atol=1e-13)
        assert traces.shape == g[f"sq_tree_{label}"].shape
        np.testing.assert_allclose(traces, g[f"sq_tree_{label}"], rtol=0, atol=1e-10)


@pytest.mark.gpu
def test_gpu_figure_examples_render(tmp_path):
    """examples/waveform_figures.py — the reference's three figure scripts against the same API on the GPU path — runs
    end to end on a headless box and writes its three PNGs (the arrays behind them are compared above)."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("waveform_figures", Path(__file__).resolve().parent.parent / "examples" / "waveform_figures.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    made = mod.main(["--out", str(tmp_path)])
    assert [p.name for p in made] == ["irig_comparison.png", "pcmfm_filter_orders.png", "soqpsk_family.png"]
    assert all(p.stat().st_size > 10_000 for p in made)
