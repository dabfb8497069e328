"""waveforms/viz/eye.py of the reference: ``plot_eye_diagram`` (the traces themselves: ``eye_diagram_data``)."""
from waveforms_amd.viz import eye_diagram_data, plot_eye_diagram

__all__ = ["plot_eye_diagram", "eye_diagram_data"]
