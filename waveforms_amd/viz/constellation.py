"""waveforms/viz/constellation.py of the reference: ``constellation`` (the points: ``constellation_data``)."""
from waveforms_amd.viz import constellation, constellation_data

__all__ = ["constellation", "constellation_data"]
