#!/bin/bash
# Same-box A/B of compile-time variants of ONE source file (GPU box; timing aid, never shipped):
#   tools/ab_kernel.sh wf_modulate.hip "" "-DWF_MCB_WAVES=3" ...      [BENCH_FLAGS="--waveform multih"]
# Every variant is built into /tmp, loaded through WF_HIP_LIBRARY and timed by bench.py's stage events;
# the list is run twice (A B C A B C) so clock drift shows up as a difference between the two passes.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
src=$1; shift
n=0
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $flags -c waveforms_amd/csrc/$src -o /tmp/ab_$n.o 2>/dev/null
  objs=""
  for o in waveforms_amd/csrc/build/*.hip.o; do
    if [ "$(basename $o)" = "$src.o" ]; then objs="$objs /tmp/ab_$n.o"; else objs="$objs $o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libwfhip_ab_$n.so $objs
  n=$((n+1))
done
for pass in 1 2; do
  i=0
  for flags in "$@"; do
    WF_HIP_LIBRARY=/tmp/libwfhip_ab_$i.so python3 bench.py --no-cpu-baseline --overlap-streams 0 --steps 30 --warmup 5 $BENCH_FLAGS > /tmp/ab_out.json
    python3 - "$flags" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_out.json").read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:40s}] step {d['ms_per_step']:.4f} ms  stages " + " ".join(f"{k}={v['ms']:.4f}" for k, v in d["stages"].items()), flush=True)
PY
    i=$((i+1))
  done
done
