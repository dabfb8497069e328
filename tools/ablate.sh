#!/bin/bash
# Build kernel variants with extra -D flags ON THE GPU BOX and time one stage for each.
#   tools/ablate.sh <stage> "<flags variant 1>" "<flags variant 2>" ...
# e.g. tools/ablate.sh awgn "" "-DWF_ABL_NO_PHILOX" "-DWF_ABL_NO_LOG"
set -e
stage=$1; shift
cd "$(dirname "$0")/.."
src=waveforms_amd/csrc
for flags in "$@"; do
  out=/tmp/libwfhip_variant.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -shared $flags \
      $src/wf_ctx.hip $src/wf_lfsr.hip $src/wf_encode.hip $src/wf_fir.hip $src/wf_phase.hip $src/wf_modulate.hip $src/wf_awgn.hip \
      $src/wf_mfbank.hip $src/wf_viterbi.hip $src/wf_count.hip $src/wf_pipeline.hip -o $out 2>/dev/null
  printf "%-40s " "[$flags]"
  python tools/stage_bench.py $stage --lib $out
done
