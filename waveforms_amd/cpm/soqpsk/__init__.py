"""SOQPSK family (MIL-STD 188-181, IRIG-106 TG/A/B): ternary precoder, h = 1/4, pulses.

Same public names as the reference package ``waveforms.cpm.soqpsk``.
"""
from waveforms_amd.cpm.soqpsk import pulse_filters as _pulses
from waveforms_amd.cpm.soqpsk.precoder import SOQPSKPrecoder

SOQPSK_NUMER, SOQPSK_DENOM = _pulses.SOQPSK_NUMER, _pulses.SOQPSK_DENOM
freq_pulse_soqpsk = _pulses.freq_pulse_soqpsk
freq_pulse_soqpsk_tg = _pulses.freq_pulse_soqpsk_tg
freq_pulse_soqpsk_mil = _pulses.freq_pulse_soqpsk_mil
freq_pulse_soqpsk_a = _pulses.freq_pulse_soqpsk_a
freq_pulse_soqpsk_b = _pulses.freq_pulse_soqpsk_b

__all__ = ["SOQPSKPrecoder", "SOQPSK_NUMER", "SOQPSK_DENOM", "freq_pulse_soqpsk", "freq_pulse_soqpsk_tg",
           "freq_pulse_soqpsk_mil", "freq_pulse_soqpsk_a", "freq_pulse_soqpsk_b"]
