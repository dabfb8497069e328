"""ARTM multi-h CPM (IRIG-106 Tier II): quaternary symbols, h alternating 4/16 and 5/16, 3RC.

Same public names as the reference package ``waveforms.cpm.multih``.
"""
from waveforms_amd.cpm.multih import pulse_filters as _pulses
from waveforms_amd.cpm.multih.precoder import MultiHSymbolMapper

MULTIH_IRIG_NUMER, MULTIH_IRIG_DENOM = _pulses.MULTIH_IRIG_NUMER, _pulses.MULTIH_IRIG_DENOM
freq_pulse_multih_irig = _pulses.freq_pulse_multih_irig

__all__ = ["MultiHSymbolMapper", "MULTIH_IRIG_NUMER", "MULTIH_IRIG_DENOM", "freq_pulse_multih_irig"]
