"""Object lifetime at interpreter exit, and the error path of the per-symbol detector call on both of its transports.

The reference's own harness (examples/soqpsk_detection.py:176-198) keeps its detector at ``__main__`` level and drives
``iteration()``; such objects are finalised AFTER the library's atexit hook has run.  The hook therefore retires the
default contexts (wf_ctx_retire: the persistent iteration server leaves the device, side streams drain) without freeing
them, so the finalisers' calls — wf_viterbi4_iteration_quiesce, wf_link_join — still find a live context.
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(code: str, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


MODULE_LEVEL_OBJECTS = r"""
import numpy as np
from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
from waveforms_amd.link import SOQPSKLink, CPMLink

det = SOQPSKTrellisDetector(2)                      # module level, iteration mode: finalised after atexit
rng = np.random.default_rng(3)
for k in range(50):
    det.iteration(rng.normal(size=3) + 1j * rng.normal(size=3))
link = SOQPSKLink(200_000, 8, fuse=47)              # pipelined: its back end runs on the context's side stream
for blk in range(3):
    link.run_block(8.0, seed=1, stream_id=blk)
cpm = CPMLink(100_000, 8, waveform="pcmfm", fuse=42)
for blk in range(2):
    cpm.run_block(8.0, seed=1, stream_id=blk)
print("alive", det.i, flush=True)
"""


def test_module_level_detector_and_pipelined_links_survive_interpreter_exit():
    r = _run(MODULE_LEVEL_OBJECTS)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "alive 50" in r.stdout
    assert "Segmentation" not in r.stderr and "core dumped" not in r.stderr


SOME_TRIPLETS = r"""
import json
import numpy as np
from waveforms.viterbi.algorithm import SOQPSKTrellisDetector
from waveforms_amd import _hip

_hip.set_default_option(_hip.WF_OPT_ITERATION_SERVER, OPTION_VALUE)
rng = np.random.default_rng(11)
x = rng.normal(size=(8, 3)) + 1j * rng.normal(size=(8, 3))
det = SOQPSKTrellisDetector(4)
out = []
for row in x:
    try:
        b, s = det.iteration(row)
        out.append([b.tolist(), s.tolist()])
    except KeyError:
        out.append("KeyError")
print("RESULT " + json.dumps(out), flush=True)
"""


@pytest.mark.parametrize("server", ["0", "1"])
def test_iteration_transports_agree_with_the_oracle(oracle, server):
    """Both forms of the per-symbol call — the persistent server and the one-launch-per-call form
    (context option WF_OPT_ITERATION_SERVER = 1) — go through the same traceback, including its KeyError exit
    (waveforms/cpm/trellis/model.py:171-174: a state pair with no connecting branch; the C ABI's WF_ERR_KEY), which no
    input reaches for this trellis (a search over finite and non-finite triplets with the oracle found none)."""
    import json

    rng = np.random.default_rng(11)
    x = rng.normal(size=(8, 3)) + 1j * rng.normal(size=(8, 3))
    wb, ws = oracle.ViterbiOracle(4, True).run(x, full=True)
    r = _run(SOME_TRIPLETS.replace("OPTION_VALUE", server))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    assert "KeyError" not in got
    for k, (b, s_) in enumerate(got):
        assert np.array_equal(np.asarray(b), wb[k]), k
        assert np.array_equal(np.asarray(s_), ws[k]), k


def test_iteration_follows_a_swap_of_the_fsm_attribute(golden):
    """The reference fixes the trellis in __init__ (self.fsm, waveforms/viterbi/algorithm.py:27-29) and reads self.fsm on
    every call (:62, :71, :94-95): swapping `det.fsm` between the two SOQPSK trellises mid-burst takes effect at the next
    call, as it does there.  The window length cannot change (the state arrays are sized by it)."""
    from waveforms.cpm.trellis.model import FiniteStateMachine, SOQPSKTrellis4x2
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detect")
    a, b = SOQPSKTrellisDetector(2, differantial_encoding=True), SOQPSKTrellisDetector(2, differantial_encoding=True)
    c = SOQPSKTrellisDetector(2, differantial_encoding=False)
    for k in range(40):
        if k == 20:
            a.fsm = FiniteStateMachine(trellis=SOQPSKTrellis4x2)
        ba, sa = a.iteration(g["triplets"][k])
        bb, sb = b.iteration(g["triplets"][k])
        bc, sc = c.iteration(g["triplets"][k])
        assert np.array_equal(sa, sb) and np.array_equal(sa, sc)      # symbols do not depend on the input labelling
        assert np.array_equal(ba, bb if k < 20 else bc)
    with pytest.raises(ValueError):
        a.length = 4
        a.iteration(g["triplets"][0])
