"""Empty in the reference (waveforms/encoding/__init__.py has no content at the surveyed commit); kept so the package imports."""
