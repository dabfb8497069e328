#!/bin/bash
# Compile-time ablations of the CPM detector kernel (results are WRONG by construction; timing only):
#   tools/ablate_cpm.sh "-DWF_ABL_CPM_NOBPERM" "-DWF_ABL_CPM_NOROT" ...
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
cp waveforms_amd/csrc/libwfhip.so /tmp/libwfhip.keep.so
for flags in "" "$@"; do
  objs=""
  for f in waveforms_amd/csrc/*.hip; do
    o=/tmp/abl_$(basename $f).o
    if [ "$(basename $f)" = "wf_cpm_detect.hip" ] || [ ! -f $o ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $flags -c $f -o $o 2>/dev/null; fi
    objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o waveforms_amd/csrc/libwfhip.so $objs
  echo "== flags: [$flags]"; python3 tools/cpm_vit_time.py
done
cp /tmp/libwfhip.keep.so waveforms_amd/csrc/libwfhip.so
