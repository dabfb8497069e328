"""The flow of the reference's examples/pcmfm_test.py as ONE test (BASELINE configs[0] is this example's
waveform): PN15 -> SimpleTrellis2 -> NRZ (*) Bessel order 4..8 frequency pulses at sps 20 (built inline in the
example with scipy, :45-51) and the unfiltered NRZ pulse (:78) -> cpm_modulate with h = 7/10 -> the PSD the
example draws (Axes.psd(NFFT=1024, Fs=20) = matplotlib.mlab.psd with its defaults).  Golden arrays were
computed by importing the reference (tests/golden/make_pcmfm_flow_golden.py)."""
import numpy as np
import pytest

SPS, NFFT, LENGTH = 20, 1024, 3
ORDERS = (4, 5, 6, 7, 8)
# Axes.psd's defaults (scale_by_freq=True) against the "spectrum" scaling the PSD product uses, as
# waveforms/viz/psd.py asks for it (scale_by_freq=False): the window norm sum(w^2) instead of sum(w)^2, and / Fs
_W = np.hanning(NFFT)
DENSITY = _W.sum() ** 2 / (_W ** 2).sum() / SPS


def _pulses():
    from scipy.signal import besselap, impulse

    from waveforms.cpm.helpers import normalize_cpm_filter

    out = {}
    for order in ORDERS:       # the example's own third-party calls; normalize_cpm_filter is this build's
        _t, y = impulse(besselap(order, norm="mag"), T=np.linspace(0, LENGTH * 2 / 0.7, num=(LENGTH - 1) * SPS + 1))
        out[f"o{order}"] = normalize_cpm_filter(SPS, np.convolve(y, np.ones(SPS)))
    out["nrz"] = normalize_cpm_filter(SPS, np.ones(SPS))
    return out


def test_flow_pulses_and_oracle_modulator_match_reference(oracle, golden):
    g = golden("pcmfm_flow")
    bits = np.unpackbits(np.packbits(oracle.pn_sequence(15)))
    assert bits.size == int(g["nbits"][0]) == 32768
    sym = oracle.fsm_encode("SimpleTrellis2", bits)[0]
    assert int(sym.astype(np.int64).sum()) == int(g["symbols_sum"][0])
    for name, pulse in _pulses().items():
        np.testing.assert_allclose(pulse, g[f"pulse_{name}"], rtol=0, atol=1e-15)
        _t, sig = oracle.cpm_modulate(sym, 0.7, pulse, SPS)
        np.testing.assert_allclose(sig[:32], g[f"sig_head_{name}"], rtol=0, atol=1e-12)
        assert abs(sig.sum() - g[f"sig_sum_{name}"][0]) < 1e-8
        f, p = oracle.psd_welch(sig, SPS, 1, NFFT)
        np.testing.assert_allclose(f, g["freqs"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(p * DENSITY, g[f"pxx_{name}"], rtol=1e-9, atol=1e-13 * g[f"pxx_{name}"].max())


@pytest.mark.gpu
def test_gpu_pcmfm_test_flow_equals_reference(golden):
    """Through the public API on the device: PNSequence -> TrellisEncoder(SimpleTrellis2) -> cpm_modulate (sps 20,
    59/60-tap and 20-tap pulses: the general-sps modulator) -> Welch PSD kernel, rescaled to Axes.psd's default density scaling."""
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER
    from waveforms.cpm.trellis.encoder import TrellisEncoder
    from waveforms.cpm.trellis.model import SimpleTrellis2
    from waveforms.glfsr import PNSequence
    from waveforms.viz import power_spectral_density

    g = golden("pcmfm_flow")
    bits = np.unpackbits(np.packbits(PNSequence(15).generate_sequence()))
    sym = TrellisEncoder(SimpleTrellis2)(bits)
    assert int(np.asarray(sym, dtype=np.int64).sum()) == int(g["symbols_sum"][0])
    for name, pulse in _pulses().items():
        _t, sig = cpm_modulate(symbols=sym, mod_index=PCMFM_NUMER / PCMFM_DENOM, pulse_filter=pulse, sps=SPS)
        assert sig.size == (sym.size + 1) * SPS
        np.testing.assert_allclose(sig[:32], g[f"sig_head_{name}"], rtol=0, atol=1e-12)
        assert abs(sig.sum() - g[f"sig_sum_{name}"][0]) < 1e-8
        f, p = power_spectral_density(sig, SPS, 1, NFFT)
        np.testing.assert_allclose(f, g["freqs"], rtol=0, atol=1e-13)
        want = g[f"pxx_{name}"]
        np.testing.assert_allclose(p * DENSITY, want, rtol=1e-9, atol=1e-13 * want.max())
