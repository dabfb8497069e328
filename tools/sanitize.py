"""Host-side sanitizer run (CPU box only; GPU AddressSanitizer does not exist on this pool).

    python tools/sanitize.py [--quick]

1. oracle/*.c built with gcc -fsanitize=address,undefined, exercised by the oracle-vs-golden
   tests (tests/test_oracle_golden.py) with gcc's libasan preloaded into python;
2. the HOST halves of waveforms_amd/csrc/*.hip (argument validation, layouts, geometry, table
   construction: everything the C ABI does before a launch) built with
   hipcc --cuda-host-only -fsanitize=address,undefined, exercised by tests/test_cabi.py with the
   clang ASan runtime preloaded (WF_HIP_LIBRARY selects that build).
Exit status is non-zero if a test fails or a sanitizer reports.
"""
import glob
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(label, env_extra, tests, extra=()):
    env = dict(os.environ)
    env.update(env_extra)
    # python itself is not instrumented: leak reports would be CPython's own allocations
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *extra, *tests]
    print(f"[sanitize] {label}: {' '.join(cmd)}", flush=True)
    r = subprocess.run(cmd, cwd=ROOT, env=env)
    return r.returncode


def main():
    quick = "--quick" in sys.argv
    rc = 0
    # 1. the oracle's C restatements
    env = {"WF_ORACLE_SANITIZE": "1"}
    subprocess.check_call([sys.executable, "-c", "import oracle; print(oracle.build_c_oracle())"], cwd=ROOT,
                          env={**os.environ, **env})
    libasan = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libasan.so.*"))
    if not libasan:
        raise SystemExit("gcc libasan not found")
    sel = ["-k", "glfsr or encoder or modulate or triplets or philox or pn9 or cpm_detector"] if quick else []
    rc |= run("oracle (gcc ASan+UBSan)", {**env, "LD_PRELOAD": libasan[0]},
              ["tests/test_oracle_golden.py"] + ([] if quick else ["tests/test_ber_curve.py"]), sel)
    # 2. the host side of the HIP shim
    from waveforms_amd.csrc.build import asan_runtime, build_host_sanitized

    so = build_host_sanitized(verbose=False)
    rc |= run("C-ABI host code (clang ASan+UBSan, --cuda-host-only)",
              {"WF_HIP_LIBRARY": str(so), "LD_PRELOAD": asan_runtime()}, ["tests/test_cabi.py"])
    print("[sanitize] " + ("FAILED" if rc else "clean"))
    return rc


if __name__ == "__main__":
    sys.exit(main())
