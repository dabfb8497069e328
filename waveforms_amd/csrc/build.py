"""Build libwfhip.so for gfx950 with hipcc (in-tree; the .so travels with the snapshot).

    python -m waveforms_amd.csrc.build [--force] [--save-temps]
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SO = HERE / "libwfhip.so"
SOURCES = ["wf_ctx.hip", "wf_lfsr.hip", "wf_encode.hip", "wf_fir.hip", "wf_phase.hip", "wf_modulate.hip", "wf_awgn.hip",
           "wf_mfbank.hip", "wf_viterbi.hip", "wf_count.hip", "wf_pipeline.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall",
         "-Wno-unused-function"]


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(p.name for p in HERE.iterdir() if p.suffix in (".hip", ".h")) + ["../../include/wfhip.h"]:
        h.update(name.encode())
        h.update((HERE / name).read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    stamp = HERE / ".build_digest"
    digest = _digest()
    if not force and SO.exists() and stamp.exists() and stamp.read_text() == digest:
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        if not (HERE / src).exists():
            continue
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        if save_temps:
            cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
        procs.append((src, subprocess.Popen(cmd, cwd=objdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    ok = True
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            ok = False
            print(f"[build] {src} FAILED\n{out}", file=sys.stderr)
        elif verbose and out.strip():
            print(f"[build] {src}\n{out}")
    if not ok:
        raise RuntimeError("libwfhip.so: compilation failed")
    tmp = SO.with_suffix(f".{os.getpid()}.tmp")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *objs])
    os.replace(tmp, SO)
    stamp.write_text(digest)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv))
