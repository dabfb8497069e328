for wf in multih pcmfm; do
for ch in 128 192 256 320; do
  WF_CPM_LANE_CH=$ch python bench.py --waveform $wf --fuse 15 --no-cpu-baseline --steady-steps 200 --overlap-streams 0 > gpurun_out/r04_j_${wf}_ch$ch.json 2>/dev/null
  WF_CPM_LANE_CH=$ch python bench.py --waveform $wf --fuse 47 --no-cpu-baseline --steady-steps 200 --overlap-streams 0 > gpurun_out/r04_j_${wf}_ch${ch}_f47.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_j_${wf}_ch$ch.json").read().strip().splitlines()[-1])
e=json.loads(open("gpurun_out/r04_j_${wf}_ch${ch}_f47.json").read().strip().splitlines()[-1])
print("$wf CH $ch f15:", d["ms_per_step"], d["steady_state"]["ms_per_step"], {k:v["ms"] for k,v in d["stages"].items() if k in ("viterbi","mod+awgn+mfbank")}, "unproven", d["steady_state"]["detector_chunks_unproven"], "| f47 steady", e["steady_state"]["ms_per_step"], "unproven", e["steady_state"]["detector_chunks_unproven"])
PY
done; done
