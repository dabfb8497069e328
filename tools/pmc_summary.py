"""Summarise rocprofv3 output of `bench.py` into profiles/<tag>_summary.json.

    python tools/pmc_summary.py <tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> \
        --nsym 10000000 --sps 8

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is exact for 16 B-per-lane stores.  FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes (they do not fit one pass).
"""
import argparse
import csv
import json
import re
import statistics
from collections import defaultdict
from pathlib import Path


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def counters(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return {k: statistics.median(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("kernel_stats")
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--nsym", type=int, required=True)
    ap.add_argument("--sps", type=int, default=8)
    ap.add_argument("--out", default=None, help="output directory (default: profiles/)")
    ap.add_argument("--sq", default=None, help="counter_collection.csv of the SQ_* pass")
    ap.add_argument("--trace", default=None, help="kernel_trace.csv: per-dispatch durations (launch-to-launch spread)")
    ap.add_argument("--mfma", default=None, help="counter_collection.csv of the matrix-pipe pass (SQ_INSTS_VALU_MFMA_F64, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)")
    a = ap.parse_args()
    stats = {}
    for r in csv.DictReader(open(a.kernel_stats)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                   "pct": float(r["Percentage"])}
    fetch, write = counters(a.fetch_csv, "FETCH_SIZE"), counters(a.write_csv, "WRITE_SIZE")
    sq = defaultdict(dict)
    if a.sq:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(a.sq)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            sq[k] = {c: statistics.median(v) for c, v in cs.items()}
    kernels = {}
    for k, st in stats.items():
        if not (k.startswith(("lfsr", "enc_", "fir_", "phase_", "mod_", "awgn", "mf_bank", "viterbi", "count_", "cpm_", "symbol_map"))):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {**st, "fetch_size_bytes_raw": int(f), "write_size_bytes": int(w),
                      "hbm_traffic_bytes": int(2 * f + w)}
        if k in sq:
            m = sq[k]
            kernels[k].update(waves=int(m.get("SQ_WAVES", 0)), valu_insts=int(m.get("SQ_INSTS_VALU", 0)),
                              salu_insts=int(m.get("SQ_INSTS_SALU", 0)), lds_insts=int(m.get("SQ_INSTS_LDS", 0)))
            # shader cycles of the launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs (SQ_BUSY_CYCLES over the
            # 32 shader engines; the two agree within 2-4 % on the probe kernels of tools/valu_probe.hip)
            if m.get("GRBM_GUI_ACTIVE"):
                kernels[k]["shader_cycles"] = int(m["GRBM_GUI_ACTIVE"] / 8)
            elif m.get("SQ_BUSY_CYCLES"):
                kernels[k]["shader_cycles"] = int(m["SQ_BUSY_CYCLES"] / 32)
            if m.get("SQ_ACTIVE_INST_VALU"):
                kernels[k]["valu_active_quad_cycles"] = int(m["SQ_ACTIVE_INST_VALU"])   # 1 per instruction (4 per fp64 transcendental)
            if m.get("SQ_WAVE_CYCLES"):
                kernels[k]["frac_wave_cycles_valu_active"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"], 3)
                kernels[k]["frac_wave_cycles_waiting"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
    if a.mfma:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(a.mfma)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            m = {c: statistics.median(v) for c, v in cs.items()}
            if k in kernels and m.get("SQ_INSTS_VALU_MFMA_F64"):
                # matrix pipe of this launch, from its OWN counter pass: instructions, the cycles the SIMDs' matrix pipes
                # were busy (summed over SIMDs) and that pass's shader cycles
                kernels[k].update(mfma_f64_insts=int(m["SQ_INSTS_VALU_MFMA_F64"]),
                                  mfma_busy_cycles=int(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)),
                                  mfma_pass_shader_cycles=int(m.get("GRBM_GUI_ACTIVE", 0) / 8))
    if a.trace:
        per = defaultdict(list)
        for r in csv.DictReader(open(a.trace)):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            if k in kernels:
                v.sort()
                d = [x[1] for x in v]
                kernels[k].update(dispatch_ns_min=min(d), dispatch_ns_median=int(statistics.median(d)), dispatch_ns_max=max(d),
                                  dispatch_ns_in_order=d[:32])
    digest = Path(__file__).resolve().parent.parent / "waveforms_amd" / "csrc" / ".build_digest"
    out = {"tag": a.tag, "nsym": a.nsym, "sps": a.sps, "build_digest": digest.read_text().strip() if digest.exists() else None,
           "note": "traffic = 2*FETCH_SIZE + WRITE_SIZE per launch (gfx950 FETCH_SIZE correction), medians over launches",
           "kernels": kernels}
    outdir = Path(a.out) if a.out else Path(__file__).resolve().parent.parent / "profiles"
    path = outdir / f"{a.tag}_summary.json"
    path.write_text(json.dumps(out, indent=1) + "\n")
    print(path)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["avg_ns"]):
        print(f"{k:34s} {v['avg_ns'] / 1e6:8.4f} ms  traffic {v['hbm_traffic_bytes'] / 1e9:7.3f} GB")


if __name__ == "__main__":
    main()
