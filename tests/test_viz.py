"""the arrays behind the reference's plots (Welch PSD, phase tree, eye traces) as
GPU data products.  Pinned by tests/golden/viz.npz = matplotlib.mlab.psd / numpy angle+unwrap on a
reference-modulated signal."""
import numpy as np
import pytest

SPS = 8


def _signal(oracle, golden):
    g = golden("viz")
    sym = oracle.fsm_encode("SOQPSKTrellis4x2DiffEncoded", g["bits"])[0]
    t, sig = oracle.cpm_modulate(sym, 0.25, oracle.freq_pulse_soqpsk_tg(SPS), SPS)
    np.testing.assert_allclose(sig[:64], g["signal_head"], rtol=0, atol=1e-13)
    assert abs(sig.sum() - g["signal_sum"][0]) < 1e-9
    return g, t, sig


def test_oracle_viz_arrays_equal_matplotlib_and_numpy(oracle, golden):
    g, t, sig = _signal(oracle, golden)
    for nfft, bps in ((256, 1), (1024, 2)):
        f, p = oracle.psd_welch(sig, SPS, bps, nfft)
        np.testing.assert_allclose(f, g[f"psd_{nfft}_{bps}_freqs"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(p, g[f"psd_{nfft}_{bps}_pxx"], rtol=1e-10, atol=1e-30)
    tt, tr = oracle.phase_tree_traces(sig, SPS, None, 4)
    np.testing.assert_allclose(tr, g["tree_first"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(oracle.phase_tree_traces(sig, SPS, 0.25, 4)[1], g["tree_off"], rtol=0, atol=1e-10)
    te, re, im = oracle.eye_traces(t, sig, SPS, 4, 0.5)
    assert te.shape == re.shape == im.shape == ((t.size - 1) // 32, 33)
    assert te[3, 0] == 0.5 and re[2, 32] == re[3, 0] == sig.real[96]


@pytest.mark.gpu
def test_gpu_welch_psd_equals_mlab(oracle, golden):
    from waveforms.viz import power_spectral_density

    g, t, sig = _signal(oracle, golden)
    for nfft, bps in ((256, 1), (1024, 2)):
        f, p = power_spectral_density(sig, SPS, bps, nfft)
        assert np.array_equal(f, g[f"psd_{nfft}_{bps}_freqs"])
        want = g[f"psd_{nfft}_{bps}_pxx"]         # bins 140 dB under the peak carry the FFT's own rounding: bound them by the peak
        np.testing.assert_allclose(p, want, rtol=1e-9, atol=1e-13 * want.max())
    rng = np.random.default_rng(4)
    x = rng.standard_normal(70_000) + 1j * rng.standard_normal(70_000)
    for nfft in (16, 64, 4096):                     # every supported size class, many segments per workgroup
        f, p = power_spectral_density(x, 10, 1, nfft)
        fo, po = oracle.psd_welch(x, 10, 1, nfft)
        assert np.array_equal(f, fo)
        np.testing.assert_allclose(p, po, rtol=1e-9, atol=1e-13 * po.max())
    f, p = power_spectral_density(x[:100], 8, 1, 256)     # shorter than one segment: zero-padded like mlab
    po = oracle.psd_welch(x[:100], 8, 1, 256)[1]
    np.testing.assert_allclose(p, po, rtol=1e-9, atol=1e-13 * po.max())
    with pytest.raises(ValueError):
        power_spectral_density(x, 8, 1, 1000)


@pytest.mark.gpu
def test_gpu_phase_tree_and_eye_traces(oracle, golden):
    from waveforms.viz import eye_diagram_data, phase_tree_data

    g, t, sig = _signal(oracle, golden)
    tt, tr = phase_tree_data(sig, SPS, None, 4)
    assert np.array_equal(tt, np.linspace(0, 4, 32, endpoint=False))
    np.testing.assert_allclose(tr, g["tree_first"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(phase_tree_data(sig, SPS, 0.25, 4)[1], g["tree_off"], rtol=0, atol=1e-10)
    noisy = sig + 0.8 * oracle.numpy_awgn(1.0, sig.size, np.random.Generator(np.random.PCG64(2)))   # many wraps
    np.testing.assert_allclose(phase_tree_data(noisy, SPS, None, 3)[1], oracle.phase_tree_traces(noisy, SPS, None, 3)[1], rtol=0, atol=1e-10)
    for modulo, off in ((4, 0.0), (2, 0.5)):
        got = eye_diagram_data(t, sig, SPS, modulo, off)
        want = oracle.eye_traces(t, sig, SPS, modulo, off)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    assert phase_tree_data(sig[:10], SPS, None, 4)[1].shape == (0, 32)


@pytest.mark.gpu
def test_gpu_cpm_phase_tree_signal(oracle):
    """generate_cpm_phase_tree's signal (waveforms/viz/tree.py:99-146) for the one-symbol MIL pulse
    and the 4x2 trellis: every distinct symbol sequence, modulated on the GPU."""
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_mil
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SOQPSKTrellis4x2
    from waveforms.viz import cpm_phase_tree_signal, phase_tree_data

    pulse = freq_pulse_soqpsk_mil(SPS)
    sig, length = cpm_phase_tree_signal(pulse, 0.25, TrellisEncoder(SOQPSKTrellis4x2), SPS)
    assert length == 1
    seqs = sorted({tuple(int(v) for v in oracle.fsm_encode("SOQPSKTrellis4x2", np.array([0, b], dtype=np.uint8))[0]) for b in (0, 1)})
    want = np.zeros(len(seqs) * SPS + 1, dtype=np.complex128)
    for i, s in enumerate(seqs):
        want[i * SPS:(i + 1) * SPS] = oracle.cpm_modulate(np.array(s, dtype=np.int8), 0.25, oracle.freq_pulse_soqpsk_mil(SPS), SPS)[1][SPS - 1:2 * SPS - 1]
    np.testing.assert_allclose(sig, want, rtol=0, atol=1e-12)
    assert phase_tree_data(sig, SPS, None, length)[1].shape == (len(seqs), SPS)
