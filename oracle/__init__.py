"""CPU oracle for the CPM modulate -> AWGN -> matched-filter -> Viterbi path.

TEST INFRASTRUCTURE ONLY.  The product package (``waveforms_amd`` / its ``waveforms``
alias) never imports this; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / timed baseline.

Parity status: PINNED against outputs of the reference itself — the fixtures in
``tests/golden`` are produced by ``tests/golden/make_golden.py`` importing
``/root/reference`` in the build container, and ``tests/test_oracle_golden.py``
checks every function here against them.  (Device AWGN is the one build-defined
piece: see ``oracle/wf_oracle.c`` header.)

Citations are relative to ``/root/reference``.
"""
from .numpy_ref import *  # noqa: F401,F403
from .numpy_ref import __all__  # noqa: F401
from .cpm_detect import *  # noqa: F401,F403,E402  (generic CPM detector: build-defined, see cpm_oracle.c)
from .viz_ref import *  # noqa: F401,F403,E402  (arrays behind the reference's plots)
