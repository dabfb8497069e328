#!/bin/bash
# Same-box A/B of compile-time variants of ONE source file, timed by bench.py's driver-timed and steady-state figures
# (the pipelined link: what a variant does to the kernels SHARING the chip, which per-kernel stage times cannot show):
#   BENCH_FLAGS="--waveform multih" tools/ab_bench.sh wf_modulate.hip "" "-DWF_MCB_RUNS_PER_SLOT=8" ...
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
src=$1; shift
n=0
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iwaveforms_amd/csrc $flags -c waveforms_amd/csrc/$src -o /tmp/abb_$n.o 2>/dev/null
  objs=""
  for o in waveforms_amd/csrc/build/*.hip.o; do
    if [ "$(basename $o)" = "$src.o" ]; then objs="$objs /tmp/abb_$n.o"; else objs="$objs $o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libwfhip_abb_$n.so $objs
  n=$((n+1))
done
for pass in 1 2; do
  i=0
  for flags in "$@"; do
    echo -n "[$flags] $BENCH_FLAGS: "
    WF_HIP_LIBRARY=/tmp/libwfhip_abb_$i.so python3 bench.py --no-cpu-baseline --overlap-streams 0 $BENCH_FLAGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(d['ms_per_step'], 'steady', s['ms_per_step'], s['bit_errors'], s['detector_chunks_unproven'])"
    i=$((i+1))
  done
done
