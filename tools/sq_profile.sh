#!/bin/bash
# SQ instruction / activity counters per kernel of `bench.py` (own --pmc passes, no other tracing):
#   tools/sq_profile.sh <tag> [bench.py flags...]  ->  gpurun_out/<tag>/profiles/<tag>_sq_counters.json
set -e
tag=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
mkdir -p $out/profiles
flags="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/sq1 -o sq1 -- python3 bench.py $flags > /dev/null 2> $out/sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq2 -o sq2 -- python3 bench.py $flags > /dev/null 2> $out/sq2.err
python3 - <<PY
import csv, glob, json, re, statistics
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$out/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if not k.startswith(("lfsr", "enc_", "fir_", "phase_", "mod_", "awgn", "mf_bank", "viterbi", "count_")):
        continue
    m = {c: statistics.median(v) for c, v in cs.items()}
    d = {"waves": int(m.get("SQ_WAVES", 0)), "valu_insts": int(m.get("SQ_INSTS_VALU", 0)),
         "salu_insts": int(m.get("SQ_INSTS_SALU", 0)), "lds_insts": int(m.get("SQ_INSTS_LDS", 0))}
    if m.get("SQ_WAVE_CYCLES"):
        d["frac_wave_cycles_issuing"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
        d["frac_wave_cycles_valu_active"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"], 3)
    if d["waves"]:
        d["valu_insts_per_wave"] = round(d["valu_insts"] / d["waves"], 1)
    res[k] = d
json.dump({"note": "rocprofv3 --pmc SQ_* (two passes) over python3 bench.py $flags; medians per launch",
           "kernels": res}, open("$out/profiles/${tag}_sq_counters.json", "w"), indent=1)
for k, d in res.items():
    print(k.ljust(44), d)
PY
find $out -name '*counter_collection.csv' -delete
