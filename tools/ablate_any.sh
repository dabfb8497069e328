#!/bin/bash
# Build a variant of the library with extra -D flags and run any command against it:
#   tools/ablate_any.sh "<flags>" python tools/link_stage_times.py /tmp/libwfhip_variant.so
set -e
cd "$(dirname "$0")/.."
src=waveforms_amd/csrc
flags=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -shared $flags \
    $src/wf_ctx.hip $src/wf_lfsr.hip $src/wf_encode.hip $src/wf_fir.hip $src/wf_phase.hip $src/wf_modulate.hip $src/wf_awgn.hip \
    $src/wf_mfbank.hip $src/wf_viterbi.hip $src/wf_count.hip $src/wf_pipeline.hip -o /tmp/libwfhip_variant.so 2>/dev/null
"$@"
