#!/bin/bash
# Chunk-length scan of the generic CPM detector inside the link (GPU box; tuning aid):
#   tools/cpm_ch_scan.sh multih 512 576 640 704 768
wf=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for ch in "$@"; do
  python3 tools/link_stage_time.py --waveform $wf --steps 20 --opt cpm_chunk_calls=$ch "--label=$wf CH=$ch" 2>/dev/null
done
