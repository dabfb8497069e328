#!/bin/bash
# Kernel timelines of the pipelined links at steady state (rocprofv3 kernel trace of a short bench run, last kernels listed):
#   bash tools/timeline_run.sh r05      -> gpurun_out/<tag>_timeline_{soqpsk,pcmfm,multih}.txt
tag=$1
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
for wf in soqpsk pcmfm multih; do
  flags="--steps 60 --warmup 200 --no-cpu-baseline --overlap-streams 0 --steady-steps 0 --ber-points none"
  [ $wf != soqpsk ] && flags="$flags --waveform $wf"
  out=gpurun_out/${tag}_tl_$wf
  rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 bench.py $flags > /dev/null 2> $out.err
  kt=$(find $out -name '*kernel_trace.csv' | head -1)
  python3 tools/timeline.py $kt 36 > gpurun_out/${tag}_timeline_$wf.txt
  rm -rf $out
done
echo TLDONE
