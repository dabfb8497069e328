"""waveforms/viz/tree.py of the reference: ``plot_phase_tree``, ``generate_cpm_phase_tree`` and the tick formatter
``pi_fraction_formatter`` (the curves themselves: ``phase_tree_data``, ``cpm_phase_tree_signal``)."""
from fractions import Fraction

import numpy as np

from waveforms_amd.viz import cpm_phase_tree_signal, generate_cpm_phase_tree, phase_tree_data, plot_phase_tree

__all__ = ["plot_phase_tree", "generate_cpm_phase_tree", "pi_fraction_formatter", "phase_tree_data", "cpm_phase_tree_signal"]


def _pi_fraction(x: float, _pos: int) -> str:
    """A tick value as a multiple of pi in lowest terms, denominators up to 16 (waveforms/viz/tree.py:21-40):
    0, "$\\pi$", "$-\\pi$", "$3\\pi$", "$\\frac{3\\pi}{4}$"."""
    frac = Fraction.from_float(float(x) / np.pi).limit_denominator(16)
    num, den = frac.numerator, frac.denominator
    if num == 0:
        return "0"
    if den == 1:
        return rf"${'' if num > 0 else '-'}\pi$" if abs(num) == 1 else rf"${num}\pi$"
    return rf"$\frac{{{num}\pi}}{{{den}}}$"


try:                                   # a matplotlib FuncFormatter, as in the reference, when matplotlib is there
    import matplotlib.ticker as _ticker

    pi_fraction_formatter = _ticker.FuncFormatter(_pi_fraction)
except ImportError:                    # pragma: no cover
    pi_fraction_formatter = _pi_fraction
