"""Empty in the reference (waveforms/calculations.py has no content at the surveyed commit); kept so the module imports."""
