"""Static VALU class mix of every kernel of libwfhip.so (CPU box: hipcc -S of the in-tree sources).

    python tools/valu_mix.py [--out profiles/r02_valu_mix.json]

For the kernels whose HBM traffic equals their algorithmic bytes but whose HBM fraction is low,
the bound is vector-instruction ISSUE.  bench.py prices a launch as
    issue cycles = SQ_INSTS_VALU (measured, tools/sq_profile.sh) x average cycles per VALU instruction
with the average taken over this static mix and the per-class costs measured on the chip by
tools/clock_probe.hip (profiles/r02_clock_probe.json): fp64 arithmetic 8 cycles per wave64
instruction, v_mad_u64_u32 / 32-bit multiplies 8, every other VALU instruction 4, at 2.35 GHz.
"""
import argparse
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CSRC = ROOT / "waveforms_amd" / "csrc"
COST = {"f64": 8, "mul32": 8, "other": 4}


def classify(op: str) -> str:
    if re.search(r"_f64|_u64_u32|_i64_i32", op) and not op.startswith("v_mov") and "cndmask" not in op:
        return "mul32" if "u64_u32" in op or "i64_i32" in op else "f64"
    if re.match(r"v_mul_(lo|hi)_[ui]32", op):
        return "mul32"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r02_valu_mix.json"))
    a = ap.parse_args()
    from waveforms_amd.csrc.build import FLAGS, SOURCES, _digest

    kernels = {}
    for src in SOURCES:
        asm = subprocess.run(["/opt/rocm/bin/hipcc", *[f for f in FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", str(CSRC / src), "-o", "-"],
                             capture_output=True, text=True, check=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                cur = m.group(1)
                kernels[cur] = {"f64": 0, "mul32": 0, "other": 0}
                continue
            if cur and re.match(r"^\s+s_endpgm", line):
                cur = None
                continue
            if cur:
                mm = re.match(r"^\s+(v_\w+)", line)
                if mm:
                    kernels[cur][classify(mm.group(1))] += 1
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for mangled, dem in zip(kernels, names):
        short = re.sub(r"^void ", "", dem).split("(")[0]
        c = kernels[mangled]
        n = sum(c.values())
        if n < 20:
            continue
        out[short] = {**c, "valu_static": n,
                      "avg_cycles_per_valu": round(sum(COST[k] * v for k, v in c.items()) / n, 3)}
    doc = {"note": "static VALU class mix per kernel (hipcc -S of the in-tree sources); costs per wave64 instruction: "
                   f"{COST} cycles (tools/clock_probe.hip on MI355X: fp64 fma 7.7, v_mad_u64_u32 6.8, fp32 / integer 4.5-4.9)",
           "build_digest": _digest(), "clock_hz": 2.35e9, "simds": 1024, "kernels": out}
    Path(a.out).write_text(json.dumps(doc, indent=1) + "\n")
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["valu_static"])[:12]:
        print(f"{k[:60]:60s} {v}")


if __name__ == "__main__":
    main()
