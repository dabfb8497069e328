#!/bin/bash
# bench.py steady state under values of one environment variable:  tools/ab_env.sh VAR "v1 v2 ..." [bench flags]
var=$1; vals=$2; shift 2
for pass in 1 2; do
  for v in $vals; do
    echo -n "$var=$v $*: "
    env $var=$v python3 bench.py --no-cpu-baseline --overlap-streams 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(d['ms_per_step'], 'steady', s['ms_per_step'], s['bit_errors'], s['detector_chunks_unproven'])"
  done
done
