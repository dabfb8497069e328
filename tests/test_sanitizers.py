"""Host-side AddressSanitizer + UBSan run: the oracle's C restatements and the
host halves of the HIP shim, on the CPU build only (GPU sanitizers are not available on this pool)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_sanitizer_run_is_clean():
    if os.environ.get("LD_PRELOAD") or os.environ.get("WF_HIP_LIBRARY"):
        pytest.skip("already inside a sanitizer run")
    r = subprocess.run([sys.executable, "tools/sanitize.py", "--quick"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "[sanitize] clean" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr
