from .precoder import SOQPSKPrecoder
from .pulse_filters import (
    SOQPSK_DENOM,
    SOQPSK_NUMER,
    freq_pulse_soqpsk,
    freq_pulse_soqpsk_a,
    freq_pulse_soqpsk_b,
    freq_pulse_soqpsk_mil,
    freq_pulse_soqpsk_tg,
)
