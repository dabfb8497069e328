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
