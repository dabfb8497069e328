#!/bin/bash
# bench.py steady state under values of one wf_ctx option (include/wfhip.h, wf_option):  tools/ab_env.sh KEY "v1 v2 ..." [bench flags]
#   tools/ab_env.sh cpm_chunk_calls "256 320 384" --waveform pcmfm
key=$1; vals=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for v in $vals; do
    echo -n "$key=$v: "
    python3 bench.py --no-cpu-baseline --overlap-streams 0 --ber-points none --opt "$key=$v" "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(d['ms_per_step'], 'steady', s['ms_per_step'], s['bit_errors'], 'repairs', s['detector_chunk_repairs'], s['detector_chunk_repairs_handed_on'])"
done
