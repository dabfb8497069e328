#!/bin/bash
# Same-box A/B of compile-time variants of ONE source file (GPU box; timing aid, never shipped):
#   tools/ab_kernel.sh wf_modulate.hip "" "-DWF_MCB_WAVES=3" ...      [BENCH_FLAGS="--waveform multih"]
# Every variant is built into /tmp, loaded through WF_HIP_LIBRARY and timed by the link's stage events (tools/link_stage_time.py: no result checks, so ablation builds run);
# the list is run twice (A B C A B C) so clock drift shows up as a difference between the two passes.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
src=$1; shift
n=0
for flags in "$@"; do
  # a variant "SRC=<path> [flags]" compiles another copy of the file (e.g. an older revision parked
  # under waveforms_amd/csrc/build/, which travels with the snapshot but is not tracked)
  file=waveforms_amd/csrc/$src
  case "$flags" in SRC=*) file=${flags%% *}; file=${file#SRC=}; case "$flags" in *" "*) flags=${flags#* };; *) flags="";; esac;; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iwaveforms_amd/csrc $flags -c $file -o /tmp/ab_$n.o 2>/dev/null
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
    WF_HIP_LIBRARY=/tmp/libwfhip_ab_$i.so python3 tools/link_stage_time.py "--label=$flags" $BENCH_FLAGS
    i=$((i+1))
  done
done
