set -e
mkdir -p gpurun_out
for w in ${WUS:-256 192 128 96 64}; do
  for e in ${EBS:-10 6}; do
    timeout -k 10 200 python3 bench.py --waveform multih --states ${STATES:-256} --fuse ${FUSE:-143} --steps 8 --warmup 3 --steady-steps 60 --vit-warmup $w --ebn0 $e --no-cpu-baseline > gpurun_out/wu${STATES:-256}_${w}_${e}.json 2> gpurun_out/wu${STATES:-256}_${w}_${e}.err || true
    python3 - <<P
import json
try:
    d=json.loads(open('gpurun_out/wu${STATES:-256}_${w}_${e}.json').read().strip().splitlines()[-1])
    s=d['steady_state']
    print('W',$w,'ebn0',$e,'ms',d['ms_per_step'],'steady',s['ms_per_step'],'repairs',s.get('detector_chunk_repairs'),'unproven',s.get('detector_chunks_unproven'),'handed',s.get('detector_chunk_repairs_handed_on'),'errs',s.get('bit_errors'), flush=True)
except Exception as ex:
    print('W',$w,'ebn0',$e,'failed',ex, open('gpurun_out/wu${STATES:-256}_${w}_${e}.err').read()[-400:], flush=True)
P
  done
done
