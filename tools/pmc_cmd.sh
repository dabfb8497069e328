#!/bin/bash
# SQ / LDS counters of ONE kernel family of any python3 command (own --pmc passes, no other tracing):
#   tools/pmc_cmd.sh <tag> <kernel-substring> <script.py> [args...]  -> gpurun_out/<tag>/pmc.json
set -e
tag=$1; shift
pat=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/p1" -o p1 -- python3 "$@" > /dev/null 2> "$out/p1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d "$out/p2" -o p2 -- python3 "$@" > /dev/null 2> "$out/p2.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d "$out/p3" -o p3 -- python3 "$@" > /dev/null 2> "$out/p3.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/p4" -o p4 -- python3 "$@" > /dev/null 2> "$out/p4.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/p5" -o p5 -- python3 "$@" > /dev/null 2> "$out/p5.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 "$@" > /dev/null 2> "$out/kt.err"
PMC_OUT="$out" PMC_PAT="$pat" python3 - <<'PY'
import csv, glob, json, os, statistics
out, pat = os.environ["PMC_OUT"], os.environ["PMC_PAT"]      # by environment, never pasted into the source: a pattern may hold quotes
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: statistics.median(v) for c, v in acc.items()}
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            res["avg_ns"] = float(r["AverageNs"]); res["calls"] = int(r["Calls"]); res["kernel"] = r["Name"][:120]
if "FETCH_SIZE" in res: res["hbm_read_bytes"] = 2 * 1024 * res["FETCH_SIZE"]     # gfx950: FETCH_SIZE counts half of a wide streaming read
if "WRITE_SIZE" in res: res["hbm_write_bytes"] = 1024 * res["WRITE_SIZE"]
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$out" -name '*counter_collection.csv' -delete; find "$out" -name '*kernel_trace.csv' -delete
