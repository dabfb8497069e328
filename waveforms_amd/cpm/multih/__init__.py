from .precoder import MultiHSymbolMapper
from .pulse_filters import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, freq_pulse_multih_irig
