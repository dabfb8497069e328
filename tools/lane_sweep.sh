#!/bin/bash
# Chunk length x warm-up sweep of the lane-form CPM detectors inside the pipelined links (GPU box):
#   tools/lane_sweep.sh multih "160 192 224 256" "32 48"      -> steady-state ms per 1e7-symbol block, repairs per block
wf=$1; chs=$2; ws=$3; shift 3
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for ch in $chs; do for w in $ws; do
  echo -n "$wf ch=$ch W=$w: "
  python3 bench.py --waveform $wf --no-cpu-baseline --overlap-streams 0 --steady-steps 600 --opt cpm_chunk_calls=$ch --vit-warmup $w "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(d['ms_per_step'], 'steady', s['ms_per_step'], s['bit_errors'], 'repairs/block', s['detector_chunk_repairs']/s['steps'], 'handed on', s['detector_chunk_repairs_handed_on'], d['config']['detector_form'], d['stages']['viterbi']['ms'])"
done; done
