from .precoder import PCMFMSymbolMapper
from .pulse_filters import freq_pulse_pcmfm

PCMFM_NUMER = 7
PCMFM_DENOM = 10
