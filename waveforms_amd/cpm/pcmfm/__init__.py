"""PCM/FM (IRIG-106 Tier 0): h = 7/10, premodulation Bessel low-pass, antipodal symbols.

Same public names as the reference package ``waveforms.cpm.pcmfm``.
"""
from waveforms_amd.cpm.pcmfm.precoder import PCMFMSymbolMapper
from waveforms_amd.cpm.pcmfm.pulse_filters import freq_pulse_pcmfm

# modulation index h = PCMFM_NUMER / PCMFM_DENOM
PCMFM_NUMER, PCMFM_DENOM = 7, 10

__all__ = ["PCMFMSymbolMapper", "freq_pulse_pcmfm", "PCMFM_NUMER", "PCMFM_DENOM"]
