#!/bin/bash
# The lane-form detector ALONE (link fuse 15: one kernel after the other) over chunk lengths (GPU box):
#   tools/lane_alone_sweep.sh pcmfm "128 160 192 256" 64
wf=$1; chs=$2; w=$3; shift 3
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for ch in $chs; do
  echo -n "$wf ch=$ch W=$w: "
  python3 bench.py --waveform $wf --fuse 15 --no-cpu-baseline --overlap-streams 0 --steady-steps 100 --ber-points none --opt cpm_chunk_calls=$ch --vit-warmup $w "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print('steady', s['ms_per_step'], 'repairs/block', s['detector_chunk_repairs']/s['steps'], d['config']['detector_form'], {k: v['ms'] for k, v in d['stages'].items()})"
done
