# what a shorter detector warm-up would buy (timing only: the short ones leave chunks unproven)
for a in "pcmfm 320" "pcmfm 128" "pcmfm 96" "multih 128" "multih 64" "multih 48"; do
  set -- $a
  for f in 15 47; do
    echo -n "$1 warmup $2 fuse $f: "
    python3 bench.py --no-cpu-baseline --overlap-streams 0 --fuse $f --waveform $1 --vit-warmup $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steady_state']['ms_per_step'], d['steady_state']['detector_chunks_unproven'], d['stages']['viterbi']['ms'])"
  done
done
