"""Static VALU class mix of every kernel of libwfhip.so (CPU box: hipcc -S of the in-tree sources).

    python tools/valu_mix.py [--out profiles/r03_valu_mix.json]

For the kernels whose HBM traffic equals their algorithmic bytes but whose HBM fraction is low, the
bound is vector-instruction ISSUE.  bench.py prices a launch as
    issue cycles = SQ_INSTS_VALU (measured, tools/profile.sh) x average cycles per VALU instruction
with the average taken over this static mix and the per-class costs measured on the chip by
tools/valu_probe.hip (profiles/r02_valu_probe.json; exact instruction counts, inline asm), in
shader cycles per wave64 instruction per SIMD at 4-8 waves per SIMD:
    full  4.2   everything not listed below: fp64 add / mul / fma / min / cmp / cvt / floor / ldexp,
                32 x 32 multiplies and v_mad_u64_u32, every 64-bit integer op and v_mov_b64, every
                three-operand 32-bit integer op (v_add3, v_xad, v_lshl_add, v_bfe, v_perm, v_and_or,
                v_alignbit, v_mad_u32_u24), compares, v_cndmask with an SGPR mask, DPP moves, v_readlane
    fast  2.3   two-operand 32-bit integer / logic / shift ops and v_mov_b32 in the e32 encoding,
                v_cndmask_b32 on vcc, fp32 add / mul / fma
    trans64 16.2  v_rcp_f64, v_rsq_f64, v_sqrt_f64
    trans32 8.2   fp32 transcendentals
(An earlier probe, tools/clock_probe.hip, let the compiler add three v_mov_b64 per iteration to its
fp64 loop and so read 7.7 cycles for v_fma_f64; the figures priced with that — fp64 8, mul32 8,
other 4 — overstated every issue fraction by about a third.)
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
COST = {"full": 4.2, "fast": 2.3, "trans64": 16.2, "trans32": 8.2}
LOOP_WEIGHT = 16     # assumed trips per loop level when weighting the static mix by loop depth


def classify(op: str) -> str:
    if re.match(r"v_(rsq|rcp|sqrt)_f64", op):
        return "trans64"
    if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "trans32"
    if op.endswith("_e32") and not re.search(r"f64|b64|u64|i64|mul_lo|mul_hi|mul_u32|mul_i32|cmp", op):
        return "fast"
    if re.match(r"v_(fma|fmac|add|mul|sub)_f32", op):
        return "fast"
    return "full"


PROBE_CLASS = {"full": ("v_fma_f64 (8 chains)", "v_mul_f64", "v_add_f64", "v_min_f64", "v_mul_lo_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_cmp_lt_f64",
                        "v_cndmask_b32 (sgpr mask)", "v_add3_u32", "v_cvt_f64_u32"),
               "fast": ("v_fma_f32 (8 chains)", "v_xor_b32", "v_add_u32", "v_bitop3_b32 (xor3)"),
               "trans64": ("v_rsq_f64", "v_rcp_f64", "v_sqrt_f64"), "trans32": ("v_log_f32",)}


def costs_from_newest_probe():
    """Per-class issue cost = median of the event-time column of the newest profiles/r*_valu_probe.json over the class's named
    instructions (tools/valu_probe.hip); the built-in figures when no probe record carries that column."""
    import statistics

    for path in sorted((ROOT / "profiles").glob("r*_valu_probe.json"), reverse=True):
        try:
            rows = {r["instruction"]: r for r in json.loads(path.read_text()) if r.get("waves_per_simd", 8) == 8}    # (the costs are those of a full SIMD: 8 waves)
        except (OSError, ValueError, TypeError, KeyError):
            continue
        out = {}
        for cls, names in PROBE_CLASS.items():
            vals = [rows[n]["from_event_time"] for n in names if n in rows and "from_event_time" in rows[n]]
            if vals:
                out[cls] = round(statistics.median(vals), 2)
        if len(out) == len(PROBE_CLASS):
            return out, path.name
    return dict(COST), "r02_valu_probe.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r03_valu_mix.json"))
    a = ap.parse_args()
    probe_cost, probe_name = costs_from_newest_probe()
    COST.update(probe_cost)
    from waveforms_amd.csrc.build import FLAGS, SOURCES, _digest

    kernels, loopw = {}, {}
    for src in SOURCES:
        asm = subprocess.run(["/opt/rocm/bin/hipcc", *[f for f in FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", str(CSRC / src), "-o", "-"],
                             capture_output=True, text=True, check=True).stdout
        cur = None
        depth, in_label = 0, False
        for line in asm.splitlines():
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                cur = m.group(1)
                kernels[cur] = {c: 0 for c in COST}
                loopw[cur] = {c: 0.0 for c in COST}
                depth, in_label = 0, False
                continue
            if cur and re.match(r"^\s+s_endpgm", line):
                cur = None
                continue
            if cur:
                # basic-block labels carry the loop depth LLVM computed: "; in Loop: Header=BBx_y Depth=N" or
                # "; =>This [Inner] Loop Header: Depth=N" (possibly on the comment lines that follow the label);
                # "Parent Loop" / "Child Loop" lines describe OTHER loops
                if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", line):
                    depth, in_label = 0, True
                if in_label and (line.startswith(".LBB") or line.lstrip().startswith(";")):
                    if "Child Loop" not in line and "Parent Loop" not in line:
                        dm = re.search(r"(?:in Loop: Header=\w+|Loop Header:) Depth=(\d+)", line)
                        if dm:
                            depth = max(depth, int(dm.group(1)))
                else:
                    in_label = False
                mm = re.match(r"^\s+(v_\w+)", line)
                if mm:
                    cls = classify(mm.group(1))
                    kernels[cur][cls] += 1
                    loopw[cur][cls] += float(LOOP_WEIGHT ** depth)
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for mangled, dem in zip(kernels, names):
        short = re.sub(r"^void ", "", dem).split("(")[0]
        c = kernels[mangled]
        n = sum(c.values())
        if n < 20:
            continue
        lw = loopw[mangled]
        out[short] = {**c, "valu_static": n,
                      "avg_cycles_per_valu": round(sum(COST[k] * v for k, v in c.items()) / n, 3),
                      # each instruction weighted LOOP_WEIGHT ** (loop depth of its basic block): what the loops execute
                      "avg_cycles_per_valu_loop": round(sum(COST[k] * v for k, v in lw.items()) / max(sum(lw.values()), 1e-9), 3)}
    doc = {"note": "static VALU class mix per kernel (hipcc -S of the in-tree sources; avg_cycles_per_valu_loop weights every instruction "
                   f"by {LOOP_WEIGHT} ** loop depth of its basic block, i.e. it is the mix of what the inner loops execute); issue cost per wave64 instruction per SIMD "
                   f"by class: {COST} shader cycles (tools/valu_probe.hip on MI355X, profiles/{probe_name})",
           "probe": probe_name, "class_cost": dict(COST), "build_digest": _digest(), "simds": 1024, "kernels": out}
    Path(a.out).write_text(json.dumps(doc, indent=1) + "\n")
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["valu_static"])[:12]:
        print(f"{k[:60]:60s} {v}")


if __name__ == "__main__":
    main()
