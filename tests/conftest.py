import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Lazy loader for tests/golden/*.npz (outputs of the reference itself)."""
    cache = {}

    def load(name):
        if name not in cache:
            if name.endswith(".json"):
                cache[name] = json.loads((GOLDEN / name).read_text())
            else:
                with np.load(GOLDEN / f"{name}.npz") as z:
                    cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc

    orc.build_c_oracle()
    return orc


@pytest.fixture
def ctx_options():
    """Context manager factory: ``with ctx_options(WF_OPT_DET_REPAIR=1): ...`` sets wf_ctx options (include/wfhip.h,
    wf_option) on the device's default context and on every context created inside the block, and restores the defaults."""
    import contextlib

    from waveforms_amd import _hip

    @contextlib.contextmanager
    def scope(**opts):
        keys = {name: getattr(_hip, name) for name in opts}
        for name, value in opts.items():
            _hip.set_default_option(keys[name], value)
        try:
            yield
        finally:
            for name in opts:
                _hip.set_default_option(keys[name], 0)

    return scope
