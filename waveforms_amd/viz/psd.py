"""waveforms/viz/psd.py of the reference: ``plot_power_spectral_density`` (the spectrum: ``power_spectral_density``)."""
from waveforms_amd.viz import plot_power_spectral_density, power_spectral_density

__all__ = ["plot_power_spectral_density", "power_spectral_density"]
