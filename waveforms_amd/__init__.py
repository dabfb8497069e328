"""waveforms_amd — MI355X-native CPM modulate -> AWGN -> matched-filter -> Viterbi path.

Same Python API as mcdiarmid/waveforms for that path (``import waveforms`` resolves to
this package through the thin alias in ``waveforms/__init__.py``); every numeric stage
is a hand-written gfx950 HIP kernel in ``csrc/`` reached through the C ABI of
``include/wfhip.h``.  There is no CPU fallback.
"""
__version__ = "0.1.0"
