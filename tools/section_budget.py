"""Instruction budget of the headline front-end kernel's row loop BY SECTION (round-4 verdict, item 2).

    python tools/section_budget.py [--kernel 'mod_chan_bank_kernel<9, 0, 8>'] [--out profiles/r05_section_budget.json]

Differential: wf_modulate.hip is compiled to assembly (hipcc -S, gfx950, the build's own flags) once as shipped and once per
section with that section of the row body stubbed (WF_MCB_SECTION_OFF, an analysis-only macro; the stubs keep every
dependence the rest of the row needs, so nothing else is removed with them); a section's budget is what disappears from
the row loop — the instructions of the basic blocks LLVM marks "Depth=2" and deeper in that kernel — when it is stubbed:

    philox     counter words, 10 rounds (v_mad_u64_u32 + v_xor_b32) and the scalar key schedule
    gaussian   2 x (word -> double, log table + series, rsq + Goldschmidt, sector table + series, radius x angle)
    modulator  amplitude / count reads, 18 phase FMAs, base, floor / wrap, 2 x table sincos
    bank       9-tap pulse-truncation bank (read x, read tap, fma, add) x 9, quad swap, packed-row store
    rest       what no stub removes: loop control, derotate + add noise, the ring-slot stores, barriers, cold paths (row 16, tile edges)

Every VALU instruction is priced by issue class (tools/valu_mix.py: full 4.2, fast 2.3, trans64 16.2 shader cycles per
wave64 instruction per SIMD, measured by tools/valu_probe.hip); scalar, LDS and vector-memory instructions are counted.
The loop holds TWO unrolled rows per trip: figures are per row.  (A position-based split of the shipped binary does not
work: the scheduler interleaves the Box-Muller arithmetic with the modulator's across the s_setprio markers.)
"""
import argparse
import collections
import json
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from valu_mix import COST, classify  # noqa: E402

SECTIONS = {"philox": 1, "gaussian": 2, "modulator": 4, "bank": 8}


def row_loop_counts(kernel: str, off: int) -> dict:
    from waveforms_amd.csrc.build import FLAGS

    cmd = ["/opt/rocm/bin/hipcc", *[f for f in FLAGS if f != "-fPIC"], f"-DWF_MCB_SECTION_OFF={off}", "-S", "--cuda-device-only",
           str(ROOT / "waveforms_amd" / "csrc" / "wf_modulate.hip"), "-o", "-"]
    asm = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    names = {}
    for m in re.finditer(r"^(_Z\w+):", asm, flags=re.M):
        names[m.group(1)] = None
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    want = next(mg for mg, d in zip(names, dem) if re.sub(r"^void ", "", d).split("(")[0] == kernel)
    body = asm.split(f"\n{want}:", 1)[1].split("s_endpgm", 1)[0]
    c = collections.Counter()
    depth, in_label = 0, False
    for line in body.splitlines():
        if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", line):
            depth, in_label = 0, True
        if in_label and (line.startswith(".LBB") or line.lstrip().startswith(";")):
            if "Child Loop" not in line and "Parent Loop" not in line:
                dm = re.search(r"(?:in Loop: Header=\w+|Loop Header:) Depth=(\d+)", line)
                if dm:
                    depth = max(depth, int(dm.group(1)))
        else:
            in_label = False
        mm = re.match(r"^\s+([a-z]\w+)", line)
        if not mm or depth < 2:
            continue
        op = mm.group(1)
        if op.startswith("v_"):
            cls = classify(op)
            c["valu"] += 1
            c["valu_" + cls] += 1
            c["valu_issue_cycles"] += COST[cls]
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_")):
            c["vmem"] += 1
        elif op.startswith("s_load"):
            c["smem"] += 1
        elif op in ("s_waitcnt", "s_barrier", "s_nop", "s_setprio"):
            c["wait_or_marker"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return dict(c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="mod_chan_bank_kernel<9, 0, 8>")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    variants = {"shipped": 0, **SECTIONS}
    with ThreadPoolExecutor(len(variants)) as ex:
        res = dict(zip(variants, ex.map(lambda off: row_loop_counts(a.kernel, off), variants.values())))
    base = res["shipped"]
    keys = ("valu", "valu_full", "valu_fast", "valu_trans64", "valu_issue_cycles", "salu", "lds", "vmem", "smem", "branch", "wait_or_marker")
    rows_per_trip = 2
    table = {}
    for s in SECTIONS:
        table[s] = {k: round((base.get(k, 0) - res[s].get(k, 0)) / rows_per_trip, 1) for k in keys}
    table["rest"] = {k: round(base.get(k, 0) / rows_per_trip - sum(table[s][k] for s in SECTIONS), 1) for k in keys}
    tot = {k: round(base.get(k, 0) / rows_per_trip, 1) for k in keys}
    for s, t in table.items():
        t["share_of_valu_instructions"] = round(t["valu"] / tot["valu"], 3)
        t["share_of_valu_issue_cycles"] = round(t["valu_issue_cycles"] / tot["valu_issue_cycles"], 3)
    from waveforms_amd.csrc.build import _digest

    doc = {"kernel": a.kernel, "build_digest": _digest(), "rows_per_trip": rows_per_trip,
           "method": "differential: row-loop instructions (basic blocks at loop depth >= 2) of the kernel as shipped minus the same with one section stubbed (WF_MCB_SECTION_OFF)",
           "issue_cost_by_class": COST, "per_row": table, "total_per_row": tot}
    print(f"{'section':10s} {'VALU':>6s} {'full':>6s} {'fast':>6s} {'trans':>6s} {'cycles':>8s} {'% cyc':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s}")
    for s in (*SECTIONS, "rest"):
        t = table[s]
        print(f"{s:10s} {t['valu']:6.1f} {t['valu_full']:6.1f} {t['valu_fast']:6.1f} {t['valu_trans64']:6.1f} {t['valu_issue_cycles']:8.1f} "
              f"{100 * t['share_of_valu_issue_cycles']:6.1f} {t['salu']:6.1f} {t['lds']:5.1f} {t['vmem']:5.1f}")
    print(f"{'total':10s} {tot['valu']:6.1f} {tot['valu_full']:6.1f} {tot['valu_fast']:6.1f} {tot['valu_trans64']:6.1f} {tot['valu_issue_cycles']:8.1f} "
          f"{100.0:6.1f} {tot['salu']:6.1f} {tot['lds']:5.1f} {tot['vmem']:5.1f}")
    if a.out:
        Path(a.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
