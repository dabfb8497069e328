#!/bin/bash
# Same-box A/B of two source trees (this one and another checkout inside it, e.g. `git worktree add _ab_r04 <commit>` with its
# library built): bench.py steady state, alternating, PASSES times.   tools/ab_trees.sh _ab_r04 [PASSES] [bench flags]
other=$1; passes=${2:-2}; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["steady_state"]; print(d["ms_per_step"], "steady", s["ms_per_step"], s["bit_errors"], {k: v["ms"] for k, v in d["stages"].items()})'
for p in $(seq $passes); do
  for tree in "$root/$other" "$root"; do
    echo -n "$(basename $tree) [$*]: "
    (cd $tree && python3 bench.py --no-cpu-baseline --overlap-streams 0 "$@" 2>/dev/null | python3 -c "$line")
  done
done
