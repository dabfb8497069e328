#!/bin/bash
# SQ / LDS counters of ONE kernel family of a bench.py run (own --pmc passes, no other tracing):
#   tools/pmc_kernel.sh <tag> <kernel-substring> [bench.py flags...]  -> gpurun_out/<tag>/pmc.json
set -e
tag=$1; shift
pat=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
mkdir -p $out
flags="--steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --steady-steps 0 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -o p1 -- python3 bench.py $flags > /dev/null 2> $out/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $out/p2 -o p2 -- python3 bench.py $flags > /dev/null 2> $out/p2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F --output-format csv -d $out/p3 -o p3 -- python3 bench.py $flags > /dev/null 2> $out/p3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $flags > /dev/null 2> $out/kt.err
python3 - <<PY
import csv, glob, json, statistics
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: statistics.median(v) for c, v in acc.items()}
for f in glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Name"]:
            res["avg_ns"] = float(r["AverageNs"]); res["calls"] = int(r["Calls"])
json.dump(res, open("$out/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name '*counter_collection.csv' -delete; find $out -name '*kernel_trace.csv' -delete
