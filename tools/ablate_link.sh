#!/bin/bash
# Like tools/ablate.sh, but times the stages INSIDE the link (tools/link_stage_times.py) for each variant.
#   [NSYM=1e7] [FUSE=7] tools/ablate_link.sh "<flags variant 1>" "<flags variant 2>" ...
set -e
cd "$(dirname "$0")/.."
src=waveforms_amd/csrc
for flags in "$@"; do
  out=/tmp/libwfhip_variant.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -shared $flags \
      $src/wf_ctx.hip $src/wf_lfsr.hip $src/wf_encode.hip $src/wf_fir.hip $src/wf_phase.hip $src/wf_modulate.hip $src/wf_awgn.hip \
      $src/wf_mfbank.hip $src/wf_viterbi.hip $src/wf_count.hip $src/wf_pipeline.hip -o $out 2>/dev/null
  printf "%-32s " "[$flags]"
  python tools/link_stage_times.py $out ${NSYM:-10000000} ${FUSE:-7} 2>/dev/null
done
