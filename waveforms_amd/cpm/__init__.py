"""CPM waveform layer — API of reference waveforms/cpm/."""
